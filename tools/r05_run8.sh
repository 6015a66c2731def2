#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05h
mkdir -p $out
export TMPDIR=/tmp
n=2
( time TRMC_BENCH_STACKS_S=75 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus $n --steps 3 --warmup 1 --no-full-ts ) > $out/bench_gpus$n.json 2> $out/bench_gpus$n.err
echo "gpus $n rc=$? $(tail -c 300 $out/bench_gpus$n.json)"
grep -v "Warn\|warn" $out/bench_gpus$n.err | grep -n "Thread\|File\|line" | head -60 | cut -c1-200
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --headline-only --steps 12 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err
  python -c "import json,sys; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['day_ms'])"; }
L=$PWD/t-route_amd
for rep in 1 2; do
  run tb128_$rep X=1 --
  run tb256_$rep TRMC_LIB_PATH=$L/libtrmc_tb256.so --
  run tb64_$rep TRMC_LIB_PATH=$L/libtrmc_tb64.so --
  run un128_$rep X=1 -- --no-retune
  run un256_$rep TRMC_LIB_PATH=$L/libtrmc_tb256.so -- --no-retune
done
