#!/usr/bin/env python3
"""Per-dispatch table (duration + PMC counters) of the kernels whose name contains a pattern, from a rocprofv3 rocpd db.
    python tools/rocpd_dispatches.py results.db k_mc_"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = con.cursor()
disp = cur.execute(
    "select d.id, d.event_id, s.kernel_name, d.end - d.start, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d "
    "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
pm = {}
try:
    for ev, name, val in cur.execute(
            "select e.event_id, p.name, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
            "group by e.event_id, p.name"):
        pm.setdefault(ev, {})[name] = val
except sqlite3.Error:
    pass
agg = {}
sel = [d for d in disp if pat in d[2]]
if len(sel) <= 12:   # few dispatches: one line each, in launch order
    for did, ev, name, dur, gx, wx in sel:
        short = name[name.find("k_mc_"):][:28] if "k_mc_" in name else name[:28]
        c = pm.get(ev, {})
        print(f"{short:28s} grid {gx:9d} us {dur / 1e3:10.2f}  " + "  ".join(f"{k}={c[k]:.4g}" for k in sorted(c)))
for did, ev, name, dur, gx, wx in disp:
    if pat not in name:
        continue
    short = name[name.find("k_mc_"):][:28] if "k_mc_" in name else name[:28]
    key = (short, gx)
    a = agg.setdefault(key, {"n": 0, "dur": 0, "c": {}})
    a["n"] += 1
    a["dur"] += dur
    for k, v in pm.get(ev, {}).items():
        a["c"][k] = a["c"].get(k, 0) + v
for (short, gx), a in agg.items():
    print(f"{short:28s} grid {gx:9d} dispatches {a['n']:5d} avg_us {a['dur'] / a['n'] / 1e3:10.2f}")
    for k in sorted(a["c"]):
        print(f"    {k:24s} per dispatch {a['c'][k] / a['n']:18.1f}")
