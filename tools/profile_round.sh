#!/bin/bash
# The round's profile evidence, all of the TIMED configuration (bench.py --headline-only: the process ends with the
# headline's windows): a rocprofv3 kernel trace (every kernel of the process, then the timeline of the last window), and
# separate counter-only passes -- FETCH_SIZE; WRITE_SIZE; the SQ instruction / cycle counters -- never combined with other
# trace domains, read for the LAST window's launches.   usage: tools/profile_round.sh <tag>     (copy gpurun_out/<tag>_* into profiles/)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p "$out"
lean="--headline-only --no-traffic --no-parity-full"
timeout 900 rocprofv3 --kernel-trace -d "$out/trace" -o trace -- python bench.py --steps 4 --warmup 1 $lean > "$out/trace.log" 2>&1
sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES")
i=0
for c in "${sets[@]}"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d "$out/pmc$i" -o p -- python bench.py --steps 1 --warmup 0 $lean > "$out/pmc$i.log" 2>&1
done
tdb=$(find "$out/trace" -name '*.db' | head -1)
sum=gpurun_out/${tag}_rocprofv3_summary.txt
{
  echo "# rocprofv3 --kernel-trace -- python bench.py --steps 4 --warmup 1 --headline-only   (every kernel of the process: the untuned plan's"
  echo "# windows, the tuning window, the spin-up of the tuned plan, then the timed sequence of days on the plan and its clone)"
  python tools/rocpd_summary.py "$tdb" | cut -c1-170
  echo
  echo "# the last window of that trace (a timed headline window)"
  python tools/window_timeline.py "$tdb"
} > "$sum"
dbs=""
for d in pmc1 pmc2 pmc3; do
  db=$(find "$out/$d" -name '*.db' | head -1)
  [ -n "$db" ] && dbs="$dbs $db"
done
python tools/pmc_last_window.py $dbs --json > "gpurun_out/${tag}_pmc_last_window.json"
find "$out" -name '*.db' -delete
cat "$sum" | head -40
cat "gpurun_out/${tag}_pmc_last_window.json" | head -60
