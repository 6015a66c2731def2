#!/bin/bash
# Kernel trace of the last headline window under run-time knobs: tools/trace_env.sh <tag> [VAR=a ...] -- the window's
# timeline (tools/window_timeline.py) and the first wide launches with their queue and stream.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/trace_$tag
mkdir -p "$out"
env "$@" timeout 600 rocprofv3 --kernel-trace -d "$out/t" -o trace -- python bench.py --steps 2 --warmup 1 --headline-only --no-traffic --no-parity-sample > "$out/log" 2>&1
db=$(find "$out/t" -name '*.db' | head -1)
python tools/window_timeline.py "$db"
python - "$db" <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id, d.grid_size_x from rocpd_kernel_dispatch d "
                   "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
st = [i for i, r in enumerate(rows) if "k_init_state" in r[0]][-1]
t0 = rows[st][1]
n = 0
for r in rows[st:]:
    if "k_mc_tile" in r[0]:
        print("tile queue %s stream %s grid %7d start %8.3f ms dur %7.1f us" % (r[3], r[4], r[5], (r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3))
        n += 1
        if n > 45:
            break
P
find "$out" -name '*.db' -delete
