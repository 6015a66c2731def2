#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05l
mkdir -p $out
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_window_api.py tests/test_gpu_sequence.py tests/test_reservoirs.py tests/test_nudging.py tests/test_gpu_distributed_sim.py -m gpu -q -x ) > $out/pytest.log 2>&1
grep -n "passed\|failed\|FAILED" $out/pytest.log | tail -5
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --headline-only --steps 12 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err
  python -c "import json,sys; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
for rep in 1 2; do
  run untuned_tp1_$rep TRMC_TAIL_PARTITION=1 -- --no-retune
  run untuned_tp0_$rep TRMC_TAIL_PARTITION=0 -- --no-retune
  run tuned_tp1_$rep TRMC_TAIL_PARTITION=1 --
  run tuned_tp0_$rep TRMC_TAIL_PARTITION=0 --
  run persist0_tp1_$rep TRMC_TAIL_PARTITION=1 -- --persistence 0.0
  run persist0_tp0_$rep TRMC_TAIL_PARTITION=0 -- --persistence 0.0
done
run tol_tp1 TRMC_TAIL_PARTITION=1 TRMC_ARITHMETIC=tolerance --
run tol_tp0 TRMC_TAIL_PARTITION=0 TRMC_ARITHMETIC=tolerance --
