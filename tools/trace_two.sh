#!/bin/bash
# Kernel trace of bench.py's two-member leg: when do the tiles and the tails of the two plans run?
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/trace_two
mkdir -p "$out"
env TRMC_BENCH_STOP_AFTER_TWO=1 "$@" timeout 600 rocprofv3 --kernel-trace -d "$out/t" -o trace -- python bench.py --steps 2 --warmup 1 --no-traffic --no-parity-sample --no-cpu-baseline > "$out/log" 2>&1
tail -1 "$out/log" | cut -c1-300
db=$(find "$out/t" -name '*.db' | head -1)
python - "$db" <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from rocpd_kernel_dispatch d "
                   "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
inits = [i for i, r in enumerate(rows) if "k_init_state" in r[0]]
first = inits[-6]            # the last six windows
t0 = rows[first][1]
def short(n):
    for k in ("k_mc_tile", "k_mc_step", "k_emit", "k_init_state", "k_prep_qlat"):
        if k in n: return k
    return None
spans = {}
for r in rows[first:]:
    k = short(r[0])
    if k is None: continue
    key = (k, r[4])
    # split a stream's launches into windows by gaps: new window when k_init_state of that plan appears -- simpler: bucket by launch count
    spans.setdefault(key, []).append(((r[1] - t0) / 1e6, (r[2] - t0) / 1e6, r[3]))
for (k, stream), v in sorted(spans.items(), key=lambda kv: kv[1][0][0]):
    per = {"k_mc_tile": 22, "k_mc_step": 288, "k_emit": 9, "k_init_state": 1, "k_prep_qlat": 1}[k]
    for w in range(0, len(v), per):
        c = v[w:w + per]
        print("%-12s stream %3s queue %s  window %d: %8.3f -> %8.3f ms (%d launches)" % (k, stream, c[0][2], w // per, c[0][0], c[-1][1], len(c)))
P
find "$out" -name '*.db' -delete
