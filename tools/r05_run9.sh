#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05i
mkdir -p $out
export TMPDIR=/tmp TRMC_COMM_TIMEOUT_S=50 TRMC_BENCH_STACKS_S=45
for i in 1 2 3 4 5 6; do
  ( TRMC_ENGINE=levels TRMC_WIDE_MIN_ROWS=8192 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 2 --steps 4 --warmup 1 --no-full-ts --nseg 500000 --no-parity-full ) > $out/small_$i.json 2> $out/small_$i.err
  echo "small $i rc=$? $(tail -c 120 $out/small_$i.json)"
done
for i in 1 2 3; do
  ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 2 --steps 3 --warmup 1 --no-full-ts $([ $i != 1 ] && echo --no-parity-full) ) > $out/full_$i.json 2> $out/full_$i.err
  echo "full $i rc=$? $(tail -c 120 $out/full_$i.json)"
done
grep -l "did not reach" $out/*.err
