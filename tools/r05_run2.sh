#!/bin/bash
# round 5, GPU run 2: the whole GPU suite (tolerance distributions recorded), tolerance-arithmetic tuning of the tiers
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05b
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/tolerance_report.json
( time python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1
tail -5 $out/pytest.log
cp gpurun_out/tolerance_report.json $out/ 2>/dev/null
for cfg in 0:16 65536:16 49152:16 32768:16 65536:12 32768:8; do
  IFS=: read rows k <<< "$cfg"
  e="TRMC_ARITHMETIC=tolerance TRMC_WIDE_K=$k"
  [ "$rows" != "0" ] && e="$e TRMC_WIDE_MIN_ROWS=$rows"
  env $e python bench.py --headline-only --steps 9 --warmup 1 > $out/tol_${rows}_${k}.json 2> $out/tol_${rows}_${k}.err
  echo "tol $cfg: $(head -c 420 $out/tol_${rows}_${k}.json)"
done
( time python bench.py --no-traffic --no-cpu-baseline --no-diffusive --no-full-ts --no-persistence-sweep --no-parity-full ) > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('hourly_output'), d.get('parity_mode',{}).get('ms_per_step'), d.get('value_tolerance',{}).get('ms_per_step'))
PY
