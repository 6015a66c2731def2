#!/usr/bin/env python3
"""The drop-in's plan cache on the bench network when the forcing's regime changes (developer tool, GPU box): days of one
regime (every day derived from the one before, synthetic.forcing), then days of an unrelated one -- through
mc_reach._PLANS.lease exactly as compute_network_structured takes its plan.  Prints, per call, the cache entry's stage
(0 collecting costs, 1 costs taken: the next call rebuilds, 2 tuned), the window's device time and the call's wall time.
    python tools/retune_probe.py [--nseg N] [--days-a 5] [--days-b 8]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from troute_amd import synthetic  # noqa: E402
from troute_amd.routing.fast_reach import mc_reach as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nseg", type=int, default=None)
ap.add_argument("--days-a", type=int, default=5)
ap.add_argument("--days-b", type=int, default=8)
ap.add_argument("--nsteps", type=int, default=288)
a = ap.parse_args()
kw = {"nseg": a.nseg, "nnet": max(3, a.nseg // 185)} if a.nseg else {}
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), **kw)
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg, nq = to.shape[0], qlat.shape[1]
up_ptr, up_idx = synthetic.upstream_csr(to)
qts = 12
state = np.zeros((nseg, 3), np.float32)


def days(n, seed, first):
    out, prev = [], first
    for i in range(n):
        prev = synthetic.forcing(nseg, nq, seed + i, previous=prev)
        out.append(prev)
    return out


regimes = [("A", days(a.days_a, 100, qlat)), ("B", days(a.days_b, 200, synthetic.forcing(nseg, nq, 999)))]
for name, ds in regimes:
    for i, q in enumerate(ds):
        t0 = time.perf_counter()
        with M._PLANS.lease(up_ptr, up_idx, params, None, 32, 0, True, ()) as plan:
            plan.upload_forcing(a.nsteps, q, state)
            plan.route_device(a.nsteps, qts, True)
            state = plan.download_final_state()
            ms = plan.stats()["ms_main"]
        (e,) = M._PLANS._d.values()
        pol = e["policy"]
        print(f"regime {name} day {i}: window {ms:6.2f} ms  call {1e3 * (time.perf_counter() - t0):7.1f} ms  -> stage {e['stage']}"
              f"  (fastest on this order {pol.best if pol else None}, re-tunes {pol.retunes if pol else 0})", flush=True)
M._PLANS.clear()
