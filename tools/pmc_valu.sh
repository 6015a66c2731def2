#!/bin/bash
# Per library variant: kernel durations + VALU/SALU counts (pass 1) and L2 fetch/write sizes (pass 2) of the step kernel.
#   tools/pmc_valu.sh <outdir under gpurun_out> "" _x ...   (suffixes of t-route_amd/libtrmc<suffix>.so)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/$1; shift
mkdir -p "$out"
cmd="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-ts"
for v in "$@"; do
  L=$PWD/t-route_amd/libtrmc$v.so
  echo "== variant '$v'" >> "$out/summary.txt"
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    TRMC_LIB_PATH=$L timeout 600 rocprofv3 --pmc $set --kernel-trace -d "$out/p$v$i" -o r -- $cmd > "$out/p$v$i.log" 2>&1
    db=$(find "$out/p$v$i" -name '*.db' | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py "$db" | grep -E "k_mc_step|k_class_order|k_emit" | cut -c1-40,60-200 >> "$out/summary.txt"
    find "$out/p$v$i" -name '*.db' -delete
  done
done
cat "$out/summary.txt"
