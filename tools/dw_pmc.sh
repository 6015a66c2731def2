#!/bin/bash
# Instruction mix of k_dw_solve (counter-only passes): tools/dw_pmc.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/${1:-dwpmc}
mkdir -p "$out"
: > "$out/summary.txt"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d "$out/p$i" -o r -- python tools/dw_time.py 12 > "$out/p$i.log" 2>&1
  db=$(find "$out/p$i" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" | grep -E "k_dw_solve" | cut -c1-30,60-200 >> "$out/summary.txt"
  find "$out/p$i" -name '*.db' -delete
done
cat "$out/summary.txt"
