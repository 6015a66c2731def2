#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/chainpmc
mkdir -p $out
: > $out/summary.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_WAVE_CYCLES SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o r -- python tools/chain_probe.py --rows 64 --reps 1 > $out/p$i.log 2>&1
  db=$(find $out/p$i -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_dispatches.py "$db" "k_mc_flow" >> $out/summary.txt
  find $out/p$i -name '*.db' -delete
done
cat $out/summary.txt
