#!/bin/bash
# Kernel + memory-copy trace of bench.py's timed sequence of days (plan + clone): who runs when in the last windows?
#     tools/trace_seq.sh [VAR=value ...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/trace_seq
rm -rf "$out"; mkdir -p "$out"
env TRMC_BENCH_DEBUG=1 "$@" timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d "$out/t" -o trace -- python bench.py --steps 6 --warmup 1 --headline-only --no-traffic --no-parity-full > "$out/log" 2> "$out/err"
tail -1 "$out/log" | cut -c1-200
grep "\[sequence\]" "$out/err" | cut -c1-1500
db=$(find "$out/t" -name '*.db' | head -1)
python - "$db" <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from rocpd_kernel_dispatch d "
                   "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
copies = []
for t in tabs:
    if "memory_copy" in t:
        cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
        try:
            copies = con.execute(f"select start, end, size from {t} order by start").fetchall()
        except Exception as e:
            print("copy table", t, cols, e)
        break
def short(n):
    for k in ("k_mc_tile", "k_mc_step", "k_emit", "k_prep_qlat", "k_chain_state", "k_gather_rows", "k_final_state", "k_init_state", "k_decimate"):
        if k in n: return k
    return None
# the last three windows: find the last three k_prep_qlat launches
preps = [i for i, r in enumerate(rows) if "k_prep_qlat" in r[0]]
first = preps[-4]
t0 = rows[first][1]
ev = []
cur = {}
for r in rows[first:]:
    k = short(r[0])
    if k is None: continue
    key = (k, r[4])
    s, e = (r[1] - t0) / 1e6, (r[2] - t0) / 1e6
    if key in cur and s - cur[key][1] < 1.0:       # same burst
        cur[key][1] = e; cur[key][2] += 1; cur[key][3] += e - s
    else:
        if key in cur: ev.append((cur[key][0], cur[key][1], key, cur[key][2], cur[key][3]))
        cur[key] = [s, e, 1, e - s]
for key, v in cur.items(): ev.append((v[0], v[1], key, v[2], v[3]))
for c in copies:
    s, e = (c[0] - t0) / 1e6, (c[1] - t0) / 1e6
    if e > 0: ev.append((s, e, ("COPY %.0f MB" % (c[2] / 1e6), -1), 1, e - s))
print("start ms   end ms    what (stream)            launches  busy ms")
for s, e, key, n, busy in sorted(ev):
    if e < 0: continue
    print(f"{s:8.2f} {e:8.2f}   {key[0]:18s} ({key[1]:3d}) {n:6d} {busy:8.2f}")
P
