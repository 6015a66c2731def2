#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05f
mkdir -p $out
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_window_api.py tests/test_gpu_sequence.py tests/test_reservoirs.py tests/test_nudging.py -m gpu -q -x ) > $out/pytest.log 2>&1
grep -n "passed\|failed\|FAILED" $out/pytest.log | tail -5
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --headline-only --steps 9 --warmup 1 "$@" > $out/$name.json 2> $out/$name.err
  echo "$name: $(head -c 330 $out/$name.json)"
}
run part_on X=1 --
run part_off TRMC_TILE_PERM=0 --
run part_on_w7 TRMC_WIDE_MIN_ROWS=65536 --
run untuned_off TRMC_TILE_PERM=0 -- --no-retune
run untuned_on TRMC_TILE_PERM=1 -- --no-retune
run tol_on TRMC_ARITHMETIC=tolerance --
run tol_off TRMC_ARITHMETIC=tolerance TRMC_TILE_PERM=0 --
run persist0_on TRMC_TILE_PERM=1 -- --persistence 0.0
run persist0_off TRMC_TILE_PERM=0 -- --persistence 0.0
for n in 2 8; do
  ( time timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 3 --warmup 1 --no-full-ts ) > $out/bench_gpus$n.json 2> $out/bench_gpus$n.err
  echo "gpus $n rc=$? $(tail -c 400 $out/bench_gpus$n.json)"
  tail -3 $out/bench_gpus$n.err | cut -c1-300
done
