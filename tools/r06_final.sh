#!/bin/bash
# round 6: the evidence run -- the whole GPU suite, smoke, the bench line, a rocprofv3 kernel trace of the timed configuration
# (bench.py --headline-only: the stream of days) with its summary table and the stream's steady state, the counter passes of the
# bench itself (roofline.traffic / valu in the line), and bench.py --gpus 2 / 8 rehearsed on this one box
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r06z
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/tolerance_report.json
( time python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1
grep -n "passed\|failed\|FAILED" $out/pytest.log | tail -8
cp gpurun_out/tolerance_report.json $out/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --steps ${BENCH_STEPS:-20} ) > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06z/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'roof', d['roofline']['frac'], 'dom', d['roofline'].get('dominant_kernel',{}).get('frac'), 'two plans', (d.get('pipeline_two_plans') or {}).get('ms_per_step'),
      'tol', (d.get('value_tolerance') or {}).get('ms_per_step'), 'hourly', {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in (d.get('hourly_output') or {}).items() if k in ('ms_per_step','in_sequence','in_stream')},
      'parity', (d.get('parity_full') or {}).get('bit_identical'), 'untuned', d['untuned']['ms_per_step'], 'persist', d['forcing_persistence'], 'dropin', (d.get('dropin') or {}).get('steady_state_call_ms'),
      'full_ts', {k: d['full_ts'].get(k) for k in ('ms_per_step', 'days_as_one_window')}, 'velocity_on_demand', d.get('velocity_on_demand'), 'valu', d['roofline'].get('valu_instructions_per_window'))
PY
lean="--headline-only --no-traffic --no-parity-full"
timeout 900 rocprofv3 --kernel-trace -d $out/trace -o trace -- python bench.py --steps 6 --warmup 1 $lean > $out/trace.log 2>&1
tdb=$(find $out/trace -name '*.db' | head -1)
{
  echo "# rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 1 --headline-only   (every kernel of the process: the untuned plan's windows, the"
  echo "# tuning window, the spin-up of the tuned plan, then the timed STREAM of days: k_mc_tile + k_mc_ctile, 18 launches a day each)"
  python tools/rocpd_summary.py "$tdb" | cut -c1-170
  echo
  echo "# the steady state of that stream"
  python tools/stream_timeline.py "$tdb"
} > gpurun_out/r06_rocprofv3_summary.txt
find $out -name '*.db' -delete
head -30 gpurun_out/r06_rocprofv3_summary.txt | cut -c1-180
tail -8 gpurun_out/r06_rocprofv3_summary.txt | cut -c1-220
for n in 2 8; do
  ( time timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n --steps 3 --warmup 1 --no-full-ts ) > $out/bench_gpus$n.json 2> $out/bench_gpus$n.err
  echo "gpus $n rc=$? $(tail -c 300 $out/bench_gpus$n.json)"
done
