#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel duration stats and PMC counters.

    python tools/rocpd_summary.py gpurun_out/prof_trace/trace_results.db [more.db ...] > profiles/rNN_x.txt

Equivalent of the `--stats` CSV (this image's rocprofv3 writes rocpd databases).
"""
import json
import sqlite3
import sys


def summarise(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
        "max(d.end-d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), "
        "max(d.private_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = {"db": path, "kernels": []}
    print(f"== {path}")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr sgpr lds scratch")
    for name, n, tot, avg, mn, mx, vg, sg, lds, scr in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:70s} {n:7d} {tot/1e6:10.3f} {avg/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f} {vg} {sg} {lds} {scr}")
        out["kernels"].append({"name": name, "calls": n, "total_ms": tot / 1e6, "avg_us": avg / 1e3,
                               "min_us": mn / 1e3, "max_us": mx / 1e3, "vgpr": vg, "sgpr": sg})
    pm = cur.execute(
        "select s.kernel_name, p.name, count(distinct d.id), sum(e.value), sum(e.value) / count(distinct d.id) "
        "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
        "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name, p.name order by s.kernel_name, p.name").fetchall()
    if pm:
        print("-- counters (per kernel: dispatches, sum over dispatches and hardware instances, per dispatch)")
        out["counters"] = []
        for k, c, n, sm, av in pm:
            short = k if len(k) <= 60 else k[:57] + "..."
            print(f"{short:60s} {c:22s} {n:7d} {sm:18.1f} {av:16.2f}")
            out["counters"].append({"kernel": k, "counter": c, "dispatches": n, "sum": sm, "mean": av})
    return out


if __name__ == "__main__":
    res = [summarise(p) for p in sys.argv[1:] if p != "--json"]
    if "--json" in sys.argv:
        print(json.dumps(res))
