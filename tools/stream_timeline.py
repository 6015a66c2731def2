#!/usr/bin/env python3
"""The steady state of a STREAM of windows (trmc_stream_*) in a rocprofv3 --kernel-trace database: the tile launches of the last
full days (before the flush that brings the last days to their end), per kernel of the stream their number, mean duration and
the gaps between them, and the period of a day.
    python tools/stream_timeline.py results.db [tiles_per_day, default 18] [lag_max tiles, default: found from the trace]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
tpd = int(sys.argv[2]) if len(sys.argv) > 2 else 18
rows = con.execute(
    "select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
    "on d.kernel_id = s.id order by d.start").fetchall()


def short(n):
    for k in ("k_mc_ctile", "k_mc_tile", "k_mc_step", "k_emit", "k_init_state", "k_prep_qlat", "k_gather_rows", "k_final_state"):
        if k in n:
            return k
    return n[:24]


rows = [(short(n), s, e, q, st, g) for n, s, e, q, st, g in rows]
# the last stream of the process: everything after the last k_init_state (trmc_stream_begin)
i0 = max(i for i, r in enumerate(rows) if r[0] == "k_init_state")
win = rows[i0:]
tiles = [r for r in win if r[0] == "k_mc_tile"]
ctiles = [r for r in win if r[0] == "k_mc_ctile"]
preps = [r for r in win if r[0] == "k_prep_qlat"]
days = len(preps)
lag = (len(tiles) - days * tpd) if len(sys.argv) <= 3 else int(sys.argv[3])
print(f"last stream of the trace: {days} days pushed, {len(tiles)} launches of k_mc_tile and {len(ctiles)} of k_mc_ctile "
      f"({tpd} per day each + {lag} that bring the last days to their end)")
if days < 3:
    sys.exit("too few days for a steady state")
# the steady state: the launches of the days after the stream has filled (lag tiles in) and before the flush
lo, hi = min(len(tiles) - lag - tpd, max(lag, tpd)), len(tiles) - lag
for name, ks in (("k_mc_tile", tiles), ("k_mc_ctile", ctiles)):
    if len(ks) < hi:
        continue
    sel = ks[lo:hi]
    dur = [(e - s) / 1e3 for _, s, e, _, _, _ in sel]
    gaps = [(sel[i + 1][1] - sel[i][2]) / 1e3 for i in range(len(sel) - 1)]
    per = (sel[-1][2] - sel[0][1]) / 1e6 / (len(sel) / tpd)
    print(f"  {name:11s} launches {lo}..{hi - 1} (full: every row has work): mean {sum(dur) / len(dur):7.1f} us  min {min(dur):7.1f}  max {max(dur):7.1f}"
          f"  gap to the next one mean {sum(gaps) / max(len(gaps), 1):6.1f} us  -> {per:6.2f} ms per day on its stream"
          f"  queue/stream {sorted(set((q, st) for _, _, _, q, st, _ in sel))}")
    print("      durations of the last day's launches, us: " + " ".join(f"{d:.0f}" for d in dur[-tpd:]))
fl = tiles[hi:]
if fl:
    print(f"  the {len(fl)} launches of the flush (fewer and fewer rows have work): " + " ".join(f"{(e - s) / 1e3:.0f}" for _, s, e, _, _, _ in fl[:40]) + " us")
