#!/bin/bash
# Kernel trace of one rank of an N-way CONUS partition (tools/sim_ranks.py): the last window's timeline.
#    tools/trace_rank.sh <world> <rank> [VAR=a ...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
world=$1; rank=$2; shift 2
out=gpurun_out/trace_rank_${world}_${rank}
mkdir -p "$out"
env "$@" timeout 600 rocprofv3 --kernel-trace -d "$out/t" -o trace -- python tools/sim_ranks.py --world $world --retune --ranks $rank --reps 1 > "$out/log" 2>&1
tail -2 "$out/log"
db=$(find "$out/t" -name '*.db' | head -1)
python tools/window_timeline.py "$db"
find "$out" -name '*.db' -delete
