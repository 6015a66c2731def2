#!/usr/bin/env python3
"""Counters of the LAST routing window in rocprofv3 --pmc databases: per kernel of the window the launches, their mean
duration and every counter's mean per launch (summed over the hardware instances that report it).
    python tools/pmc_last_window.py a.db [b.db ...] [--json]"""
import json
import sqlite3
import sys


def short(n):
    for k in ("k_mc_ctile", "k_mc_tile", "k_mc_step", "k_emit", "k_init_state", "k_prep_qlat", "k_gather_rows", "k_final_state", "k_mc_flow_lean", "k_mc_flow"):
        if k in n:
            return k
    return n[:32]


out = {}
for path in [p for p in sys.argv[1:] if p != "--json"]:
    con = sqlite3.connect(path)
    rows = con.execute("select d.id, d.event_id, s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join "
                       "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    # a window starts with its forcing transpose (windows of a sequence handed on in HBM have no k_init_state)
    starts = [i for i, r in enumerate(rows) if "k_prep_qlat" in r[2]] or [i for i, r in enumerate(rows) if "k_init_state" in r[2] or "k_flow_init" in r[2]]
    win = rows[starts[-1]:] if starts else rows
    pm = {}
    for ev, name, val, n in con.execute("select e.event_id, p.name, sum(e.value), count(*) from rocpd_pmc_event e join rocpd_info_pmc p "
                                        "on e.pmc_id = p.id group by e.event_id, p.name"):
        pm.setdefault(ev, {})[name] = (val, n)
    for _, ev, name, s, e in win:
        k = out.setdefault(short(name), {"launches": {}, "us": {}, "counters": {}, "instances": {}})
        k["launches"][path] = k["launches"].get(path, 0) + 1
        k["us"][path] = k["us"].get(path, 0.0) + (e - s) / 1e3
        for c, (v, n) in pm.get(ev, {}).items():
            k["counters"][c] = k["counters"].get(c, 0.0) + v
            k["instances"][c] = n
            k.setdefault("n_" + c, 0)
            k["n_" + c] += 1
res = {}
for kname, k in out.items():
    n = max(k["launches"].values())
    r = {"launches_in_window": n, "mean_us": {p.split("/")[-2] if "/" in p else p: round(k["us"][p] / k["launches"][p], 2) for p in k["us"]}}
    for c, v in k["counters"].items():
        r[c + "_per_launch"] = v / k["n_" + c]
        r[c + "_instances"] = k["instances"][c]
    res[kname] = r
if "--json" in sys.argv:
    print(json.dumps(res, indent=1))
else:
    for kname, r in res.items():
        print(kname, json.dumps(r))
