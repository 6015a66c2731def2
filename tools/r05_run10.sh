#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05j
mkdir -p $out
export TMPDIR=/tmp TRMC_COMM_TIMEOUT_S=40 TRMC_BENCH_STACKS_S=12
for i in 1 2 3 4 5 6; do
  mkdir -p $out/st$i
  ( TRMC_BENCH_STACKS_DIR=$out/st$i timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2956$i bench.py --gpus 2 --steps 3 --warmup 1 --no-full-ts --no-parity-full ) > $out/full_$i.json 2> $out/full_$i.err
  rc=$?
  echo "full $i rc=$rc"
  if [ $rc != 0 ]; then
    for r in 0 1; do echo "--- rank $r last stacks"; grep -n "Timeout" $out/st$i/bench_stacks_rank$r.txt | tail -2; tail -25 $out/st$i/bench_stacks_rank$r.txt | cut -c1-200; done
    break
  fi
done
