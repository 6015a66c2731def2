#!/bin/bash
# The round's profile evidence: rocprofv3 kernel trace of the default bench command, then separate counter-only
# passes for FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU (never combined with other trace domains), summarised by
# tools/rocpd_summary.py into gpurun_out/<tag>_summary.txt.   usage: tools/profile_round.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p "$out"
sum=gpurun_out/${tag}_summary.txt
: > "$sum"
timeout 900 rocprofv3 --kernel-trace -d "$out/trace" -o trace -- python bench.py --steps 3 --warmup 1 > "$out/trace.log" 2>&1
tail -1 "$out/trace.log" > "gpurun_out/${tag}_bench_under_trace.json"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d "$out/$c" -o $c -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-full-ts > "$out/$c.log" 2>&1
done
for d in trace FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  db=$(find "$out/$d" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" >> "$sum"
done
find "$out" -name '*.db' -delete
python bench.py --steps 5 --warmup 1 > "gpurun_out/${tag}_bench.json" 2> "$out/bench.err"
cat "$sum" | cut -c1-160
tail -1 "gpurun_out/${tag}_bench.json" | cut -c1-400
