#!/bin/bash
# Dynamic instruction mix of the step kernel: several counter-only rocprofv3 passes (no trace domains besides
# --kernel-trace), summarised by tools/rocpd_summary.py.   usage: tools/pmc_mix.sh <outdir under gpurun_out>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/${1:-mix}
mkdir -p "$out"
cmd="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-ts"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_I8" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d "$out/p$i" -o r -- $cmd > "$out/p$i.log" 2>&1
  db=$(find "$out/p$i" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" | grep -A40 -- "-- counters" | grep -E "k_mc_step|counters" >> "$out/summary.txt"
  find "$out/p$i" -name '*.db' -delete
done
cat "$out/summary.txt"
