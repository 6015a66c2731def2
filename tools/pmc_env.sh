#!/bin/bash
# Counters per kernel of the last headline window under run-time knobs: tools/pmc_env.sh <tag> "<counters>" [VAR=a VAR2=b ...]
# (one counter-only rocprofv3 pass, --kernel-trace only).  Default counters: VALU instructions and activity.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
ctr=${1:-"SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"}; shift
out=gpurun_out/pmc_$tag
mkdir -p "$out"
env "$@" timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d "$out/p" -o p -- python bench.py --steps 1 --warmup 0 --headline-only --no-traffic --no-parity-sample ${BENCH_ARGS:-} > "$out/log" 2>&1
db=$(find "$out/p" -name '*.db' | head -1)
python tools/pmc_last_window.py "$db" | cut -c1-900
find "$out" -name '*.db' -delete
