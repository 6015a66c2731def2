#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05t
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace -d $out/trace -o trace -- python bench.py --steps 4 --warmup 1 --headline-only --no-traffic --no-parity-full > $out/trace.log 2>&1
tdb=$(find $out/trace -name '*.db' | head -1)
python tools/window_timeline.py "$tdb" | tail -16
find $out -name '*.db' -delete
