#!/bin/bash
# Counters per kernel of tools/cluster_probe.py's last window: tools/pmc_cluster.sh <tag> "<counters>" [probe args]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
ctr=${1:-"SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"}; shift
out=gpurun_out/pmc_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d "$out/p" -o p -- python tools/cluster_probe.py --windows 1 "$@" > "$out/log" 2>&1
grep cluster_rows "$out/log"
db=$(find "$out/p" -name '*.db' | head -1)
python tools/pmc_last_window.py "$db" | cut -c1-700
find "$out" -name '*.db' -delete
