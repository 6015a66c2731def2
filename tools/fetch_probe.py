#!/usr/bin/env python3
"""Where a throughput-mode step's wall time goes: route_device / fetch_wait / fetch_begin per window on the bench network.
    python tools/fetch_probe.py [--steps 6]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
n = net["to"].shape[0]
r = ShardedRouter(net["to"], net["params"], assume_short_ts=True)
r.upload(288, net["qlat"], np.zeros((n, 3), np.float32))
r.route_resident(12, True)
r.upload(288, net["qlat"], None)
rs = r.plan0.rowset(r.my_out0_local)
P = r.plan0
for mode in ("resident", "fetch-async", "fetch-sync"):
    rows = []
    t_all = time.perf_counter()
    for k in range(a.steps):
        t0 = time.perf_counter()
        st = P.route_device(288, 12, True)
        t1 = time.perf_counter()
        if mode == "fetch-async":
            P.fetch_wait()
            t2 = time.perf_counter()
            P.fetch_begin(rs, True)
            t3 = time.perf_counter()
        elif mode == "fetch-sync":
            P.fetch_begin(rs, True)
            t2 = time.perf_counter()
            P.fetch_wait()
            t3 = time.perf_counter()
        else:
            t2 = t3 = t1
        rows.append((t1 - t0, t2 - t1, t3 - t2, st["ms_main"], st["ms_total"]))
    if mode == "fetch-async":
        P.fetch_wait()
    el = (time.perf_counter() - t_all) / a.steps
    m = np.array(rows[1:]).mean(axis=0) * [1e3, 1e3, 1e3, 1, 1]
    print(f"{mode:12s} per window {el * 1e3:7.2f} ms | route_device {m[0]:6.2f}  then {m[1]:5.2f}  then {m[2]:5.2f} | device ms_main {m[3]:6.2f} ms_total {m[4]:6.2f}", flush=True)
r.close()
