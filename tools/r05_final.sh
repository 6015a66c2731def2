#!/bin/bash
# round 5: the evidence run -- the whole GPU suite, smoke, the bench line, the rocprofv3 trace + counter passes, the N = 2 / 8 rehearsals
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05z
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/tolerance_report.json
( time python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1
grep -n "passed\|failed\|FAILED" $out/pytest.log | tail -8
cp gpurun_out/tolerance_report.json $out/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05z/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'roof', d['roofline']['frac'], 'dom', d['roofline'].get('dominant_kernel',{}).get('frac'), 'tol', (d.get('value_tolerance') or {}).get('ms_per_step'), 'hourly', (d.get('hourly_output') or {}).get('ms_per_step'), 'parity', d['parity_full'].get('bit_identical'), 'untuned', d['untuned']['ms_per_step'], 'persist', d['forcing_persistence'])
PY
( time bash tools/profile_round.sh r05 ) > $out/profile.log 2>&1
head -30 gpurun_out/r05_rocprofv3_summary.txt | cut -c1-180
for n in 2 8; do
  ( time timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n --steps 3 --warmup 1 --no-full-ts ) > $out/bench_gpus$n.json 2> $out/bench_gpus$n.err
  echo "gpus $n rc=$? $(tail -c 200 $out/bench_gpus$n.json)"
done
