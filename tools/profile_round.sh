#!/bin/bash
# The round's profile evidence: rocprofv3 kernel trace of the default bench command, then separate counter-only
# passes for FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU (never combined with other trace domains), summarised by
# tools/rocpd_summary.py into gpurun_out/<tag>_summary.txt, plus gpurun_out/<tag>_pmc_traffic.json
# (copy both into profiles/).   usage: tools/profile_round.sh <tag> <engine: levels|flow> <round>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r}
engine=${2:-levels}
rnd=${3:-0}
out=gpurun_out/$tag
mkdir -p "$out"
sum=gpurun_out/${tag}_summary.txt
: > "$sum"
lean="--no-cpu-baseline --no-full-ts --no-diffusive --no-parity-mode --no-traffic"
timeout 900 rocprofv3 --kernel-trace -d "$out/trace" -o trace -- python bench.py --steps 3 --warmup 1 $lean > "$out/trace.log" 2>&1
grep "^{\"metric\"" "$out/trace.log" | tail -1 > "gpurun_out/${tag}_bench_under_trace.json"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d "$out/$c" -o $c -- python bench.py --steps 2 --warmup 0 $lean > "$out/$c.log" 2>&1
done
dbs=""
for d in trace FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  db=$(find "$out/$d" -name '*.db' | head -1)
  [ -n "$db" ] && dbs="$dbs $db"
done
python tools/rocpd_summary.py $dbs --json > "$out/all.txt"
grep -v '^\[{' "$out/all.txt" > "$sum"
grep '^\[{' "$out/all.txt" > "$out/summary.json"
python tools/make_pmc_json.py "$out/summary.json" "$engine" "$rnd" > "gpurun_out/${tag}_pmc_traffic.json"
find "$out" -name '*.db' -delete
cut -c1-150 "$sum" | head -60
cat "gpurun_out/${tag}_pmc_traffic.json"
