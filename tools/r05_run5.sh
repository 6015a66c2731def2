#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05e
mkdir -p $out
export TMPDIR=/tmp
( time bash tools/profile_round.sh r05 ) > $out/profile.log 2>&1
tail -30 $out/profile.log | cut -c1-200
( time python bench.py ) > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05e/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'roof', d['roofline']['frac'], 'dom', d['roofline'].get('dominant_kernel',{}).get('frac'), 'tol', (d.get('value_tolerance') or {}).get('ms_per_step'), 'hourly', (d.get('hourly_output') or {}).get('ms_per_step'), 'parity', d['parity_full'].get('bit_identical'))
PY
export GPU_MAX_HW_QUEUES=1
for w in 8 4; do
  ( time python tools/sim_ranks.py --world $w --retune --sequence 6 --reps 2 --rebalance 2 ) > $out/sim_seq_$w.txt 2>&1
  grep -v Warning $out/sim_seq_$w.txt | grep "slowest\|pace\|single GPU under"
done
( time python tools/fuzz_parity.py --seconds 300 --nseg 30000 --seed 51 ) > $out/fuzz_parity.txt 2>&1
tail -3 $out/fuzz_parity.txt
( time python tools/fuzz_sequence.py --seconds 240 --nseg 100000 --seed 52 ) > $out/fuzz_sequence.txt 2>&1
tail -3 $out/fuzz_sequence.txt
