#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05d
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/tolerance_report.json
python tools/tol_probe.py 2>&1 | tail -30 > $out/tol_probe.txt
cat $out/tol_probe.txt | tail -25 | cut -c1-400
( time python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1
grep -n "passed\|failed\|FAILED" $out/pytest.log | tail -12
cp gpurun_out/tolerance_report.json $out/ 2>/dev/null
export GPU_MAX_HW_QUEUES=1
for w in 8 4 2; do
  ( time python tools/sim_ranks.py --world $w --retune --sequence 6 --reps 2 ) > $out/sim_seq_$w.txt 2>&1
  grep -v Warning $out/sim_seq_$w.txt | tail -12
done
