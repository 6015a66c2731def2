#!/bin/bash
# round 5, GPU run 1: the whole GPU suite, the bench line, and the second tier of tiles A/B on the headline sequence
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05a
mkdir -p $out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1
tail -5 $out/pytest.log
( time python bench.py --no-traffic ) > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
for cfg in 0:4:12 16384:4:12 16384:8:12 32768:4:8 8192:4:16 16384:2:12; do
  IFS=: read rows k lv <<< "$cfg"
  TRMC_MID_MIN_ROWS=$rows TRMC_MID_K=$k TRMC_MID_LEVELS=$lv python bench.py --headline-only --steps 9 --warmup 1 \
      > $out/mid_${rows}_${k}_${lv}.json 2> $out/mid_${rows}_${k}_${lv}.err
  echo "mid $cfg: $(cat $out/mid_${rows}_${k}_${lv}.json | head -c 400)"
done
