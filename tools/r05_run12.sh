#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05k
mkdir -p $out
export TMPDIR=/tmp
( time python bench.py --no-traffic ) > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05k/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'tol', (d.get('value_tolerance') or {}).get('ms_per_step'), 'hourly', (d.get('hourly_output') or {}).get('ms_per_step'), 'parity_mode', d['parity_mode']['ms_per_step'], 'resident', d['value_resident']['ms_per_step'], 'parity', d['parity_full'].get('bit_identical'), 'untuned', d['untuned']['ms_per_step'])
PY
tail -3 $out/bench.err
