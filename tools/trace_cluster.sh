#!/bin/bash
# Kernel trace of tools/cluster_probe.py (one timed window per setting): per-kernel totals and the timeline of the last window's
# cluster tiles.  tools/trace_cluster.sh <tag> [probe args]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/trace_$tag
mkdir -p "$out"
timeout 900 rocprofv3 --kernel-trace -d "$out/t" -o trace -- python tools/cluster_probe.py --windows 1 "$@" > "$out/log" 2>&1
tail -4 "$out/log"
db=$(find "$out/t" -name '*.db' | head -1)
python tools/rocpd_summary.py "$db" | cut -c1-150 | head -24
python - "$db" <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id, d.grid_size_x from rocpd_kernel_dispatch d "
                   "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
st = [i for i, r in enumerate(rows) if "k_init_state" in r[0]][-1]
t0 = rows[st][1]
for r in rows[st:]:
    if "k_mc_ctile" in r[0] or "k_mc_tile" in r[0]:
        print("%-10s stream %s grid %8d start %8.3f ms dur %7.1f us" % (r[0][:10], r[4], r[5], (r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3))
P
find "$out" -name '*.db' -delete
