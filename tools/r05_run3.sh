#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05c
mkdir -p $out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=1
python tools/tol_probe.py > $out/tol_probe.txt 2>&1
tail -40 $out/tol_probe.txt
for w in 8 4 2; do
  ( time python tools/sim_ranks.py --world $w --retune --sequence 6 --reps 2 ) > $out/sim_seq_$w.txt 2>&1
  grep -v Warning $out/sim_seq_$w.txt | tail -14
done
