#!/bin/bash
# Instruction counts and VALU cycles of the step kernel by iteration class: one counter pass over tools/class_probe.py,
# split by grid size.   tools/class_pmc.sh <outdir under gpurun_out> [library suffix ...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/$1; shift
mkdir -p "$out"
[ $# -eq 0 ] && set -- ""
for v in "$@"; do
  L=$PWD/t-route_amd/libtrmc$v.so
  echo "== variant '$v'" | tee -a "$out/summary.txt"
  TRMC_LIB_PATH=$L python tools/class_probe.py 2>&1 | tee -a "$out/summary.txt"
  TRMC_LIB_PATH=$L timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$out/p$v" -o r -- python tools/class_probe.py --reps 1 > "$out/p$v.log" 2>&1
  db=$(find "$out/p$v" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_dispatches.py "$db" k_mc_step | tee -a "$out/summary.txt"
  find "$out/p$v" -name '*.db' -delete
done
