#!/usr/bin/env python3
"""Timeline of the LAST routing window in a rocprofv3 --kernel-trace database: per kernel the launches, summed and mean
duration, the span they cover, how much of the wide tiles' time the tail launches overlap, and the gaps of each stream.
    python tools/window_timeline.py results.db [first-kernel-of-a-window pattern, default k_prep_qlat|k_init_state]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute(
    "select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
    "on d.kernel_id = s.id order by d.start").fetchall()


def short(n):
    for k in ("k_mc_tile", "k_mc_step", "k_emit", "k_init_state", "k_prep_qlat", "k_gather_rows", "k_final_state", "k_mc_flow"):
        if k in n:
            return k
    return n[:24]


rows = [(short(n), s, e, q, st) for n, s, e, q, st in rows]
# a window starts with its forcing transpose (the windows of a sequence handed on in HBM have no k_init_state); in such a
# sequence the next-to-last window is the one shown, between its transpose and the last window's
starts = [i for i, r in enumerate(rows) if r[0] == "k_prep_qlat"] or [i for i, r in enumerate(rows) if r[0] == "k_init_state"]
# (a sequence of two plans in sequence mode transposes a day's forcing a day EARLY, behind its copy: there the windows are
# told apart by the tile stream, which alternates between the plans)
tiles_at = [i for i, r in enumerate(rows) if r[0] == "k_mc_tile"]
groups = [i for j, i in enumerate(tiles_at) if j == 0 or rows[tiles_at[j - 1]][4] != rows[i][4]]
if len(groups) >= 3:
    starts = groups
if not starts:
    sys.exit("no window found")
i0 = starts[-1]
win = rows[i0:]
if len(starts) >= 2 and not any(r[0] == "k_init_state" for r in win):
    # The last window of a pipelined sequence (two plans taking turns) is followed by nothing: the one before it shows the
    # steady state.  Its launches are those on ITS plan's streams from its transpose on: the tile stream of the tiles that
    # follow the transpose, the stream of the last tail / transposing launch before the next window's transpose (the other
    # plan's trailing launches come first in that interval).
    i0, i1 = starts[-2], starts[-1]
    inter = rows[i0:i1]
    mine = set()
    for kind in ("k_mc_tile", "k_mc_step", "k_emit") + (() if starts is groups else ("k_prep_qlat",)):
        on = [r[4] for r in inter if r[0] == kind]
        if on:
            mine.add((kind, on[-1]))
    win = [r for r in rows[i0:] if (r[0], r[4]) in mine]
    print("(pipelined sequence: the next-to-last window, by its plan's streams)")
t0 = win[0][1]
print(f"last window: {len(win)} dispatches, span {(max(r[2] for r in win) - t0) / 1e6:.3f} ms")
by = {}
for n, s, e, q, st in win:
    by.setdefault(n, []).append((s - t0, e - t0, q, st))
for n, v in by.items():
    dur = [b - a for a, b, _, _ in v]
    print(f"  {n:14s} launches {len(v):4d}  sum {sum(dur) / 1e6:8.3f} ms  mean {sum(dur) / len(v) / 1e3:8.1f} us  first start {v[0][0] / 1e6:7.3f}  last end {v[-1][1] / 1e6:7.3f} ms"
          f"  queue/stream {sorted(set((q, st) for _, _, q, st in v))}")
if "k_mc_tile" in by and "k_mc_step" in by:
    tiles = by["k_mc_tile"]
    ov = 0
    for a, b, _, _ in by["k_mc_step"]:
        for c, d, _, _ in tiles:
            ov += max(0, min(b, d) - max(a, c))
    tot = sum(b - a for a, b, _, _ in by["k_mc_step"])
    print(f"  tail launches: {ov / max(tot, 1):.2f} of their time lies inside a wide tile's; idle between wide tiles "
          f"{sum(max(0, tiles[i + 1][0] - tiles[i][1]) for i in range(len(tiles) - 1)) / 1e6:.3f} ms")
    k = by["k_mc_step"]
    for j in (0, len(k) // 4, len(k) // 2, 3 * len(k) // 4, len(k) - 1):
        print(f"    tail launch {j:3d}: start {k[j][0] / 1e6:7.3f} dur {(k[j][1] - k[j][0]) / 1e3:7.1f} us")
    for j in (0, len(tiles) // 2, len(tiles) - 1):
        print(f"    wide tile  {j:3d}: start {tiles[j][0] / 1e6:7.3f} dur {(tiles[j][1] - tiles[j][0]) / 1e3:7.1f} us")
    gaps = [max(0, tiles[i + 1][0] - tiles[i][1]) / 1e3 for i in range(len(tiles) - 1)]
    print("    gap behind each wide tile, us: " + " ".join(f"{g:.0f}" for g in gaps))
    print("    duration of each wide tile, us: " + " ".join(f"{(b - a) / 1e3:.0f}" for a, b, _, _ in tiles))
    # the tail's chain: gaps between its consecutive launches (launch latency, or a wait for a tile), and where it stands when
    # the last tile ends
    tg = [(k[i + 1][0] - k[i][1]) / 1e3 for i in range(len(k) - 1)]
    srt = sorted(tg)
    print(f"    gaps between consecutive tail launches, us: median {srt[len(srt) // 2]:.1f}  p10 {srt[len(srt) // 10]:.1f}  p90 {srt[9 * len(srt) // 10]:.1f}"
          f"  sum {sum(tg) / 1e3:.3f} ms; the {sum(1 for g in tg if g > 3 * srt[len(srt) // 2])} above three medians sum to "
          f"{sum(g for g in tg if g > 3 * srt[len(srt) // 2]) / 1e3:.3f} ms")
    last_tile_end = tiles[-1][1]
    behind = sum(1 for a, b, _, _ in k if a >= last_tile_end)
    alone = [(b - a) / 1e3 for a, b, _, _ in k if a >= last_tile_end]
    ag = [tg[i] for i in range(len(tg)) if k[i][1] >= last_tile_end]
    print(f"    tail launches that start after the last tile has ended: {behind} (mean {sum(alone) / max(len(alone), 1):.1f} us each, gaps between them "
          f"mean {sum(ag) / max(len(ag), 1):.1f} us); the tail ends {(k[-1][1] - last_tile_end) / 1e6:.3f} ms after the last tile")
