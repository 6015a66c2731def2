#!/bin/bash
# Counter passes over tools/flow_probe.py (counter-only rocprofv3 runs, --kernel-trace only); per-dispatch table of the
# routing kernels.   usage: tools/flow_pmc.sh <outdir under gpurun_out> [probe args...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/$1; shift
mkdir -p "$out"
: > "$out/summary.txt"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d "$out/p$i" -o r -- python tools/flow_probe.py "$@" > "$out/p$i.log" 2>&1
  db=$(find "$out/p$i" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_dispatches.py "$db" "k_mc_" >> "$out/summary.txt"
  find "$out/p$i" -name '*.db' -delete
done
cat "$out/summary.txt"
