#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05g
mkdir -p $out
export TMPDIR=/tmp
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, '.')
from troute_amd import synthetic
from troute_amd.plan import topology_levels
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
up_ptr, up_idx = synthetic.upstream_csr(net["to"])
lvl, pos, nl = topology_levels(up_ptr, up_idx)
print("level widths 0..19:", np.bincount(lvl)[:20].tolist())
PY
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --headline-only --steps 12 --warmup 2 "$@" > $out/$name.json 2> $out/$name.err
  python -c "import json,sys; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), d['day_ms'])"; }
for rep in 1 2 3; do
  run w5_$rep X=1 --
  run w6_$rep TRMC_WIDE_MIN_ROWS=75000 --
  run w7_$rep TRMC_WIDE_MIN_ROWS=65536 --
  run w8_$rep TRMC_WIDE_MIN_ROWS=57000 --
done
for n in 2 8; do
  ( time timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --no-full-ts ) > $out/bench_gpus$n.json 2> $out/bench_gpus$n.err
  echo "gpus $n rc=$? $(tail -c 300 $out/bench_gpus$n.json)"
  grep -v "Warn\|warn" $out/bench_gpus$n.err | tail -4 | cut -c1-300
done
