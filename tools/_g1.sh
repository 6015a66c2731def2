cd /root/repo
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|FAILED" gpurun_out/t.log | tail -5
( time python bench.py --steps 20 ) > gpurun_out/r06z/bench.json 2> gpurun_out/r06z/bench.err; tail -3 gpurun_out/r06z/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06z/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'roof', d['roofline']['frac'], 'dom', d['roofline'].get('dominant_kernel',{}).get('frac'), 'two plans', (d.get('pipeline_two_plans') or {}).get('ms_per_step'),
      'tol', (d.get('value_tolerance') or {}).get('ms_per_step'), 'hourly', {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in (d.get('hourly_output') or {}).items() if k in ('ms_per_step','in_sequence','in_stream')},
      'parity', (d.get('parity_full') or {}).get('bit_identical'), 'untuned', d['untuned']['ms_per_step'], 'full_ts', d['full_ts'], 'dropin', (d.get('dropin') or {}).get('steady_state_call_ms'))
PY
