#!/usr/bin/env python3
"""Probe of the routing engines on the bench network: tuning window, then `--reps` hinted windows; prints ms per window.
    python tools/flow_probe.py [--nseg N] [--reps R] [--full]     (TRMC_ENGINE=levels|flow selects the engine)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nseg", type=int, default=None)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--nsteps", type=int, default=288)
ap.add_argument("--full", action="store_true", help="also the general mode (assume_short_ts=False)")
ap.add_argument("--no-retune", action="store_true")
ap.add_argument("--only-full", action="store_true")
a = ap.parse_args()
kw = {}
if a.nseg:
    kw = {"nseg": a.nseg, "nnet": max(3, a.nseg // 185)}
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), **kw)
to, params, qlat = net["to"], net["params"], net["qlat"]
q0 = np.zeros((to.shape[0], 3), np.float32)


def make(hint, short=True):
    r = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=short, engine=os.environ.get("PROBE_ENGINE", "auto"))
    r.upload(a.nsteps, qlat, q0)
    return r


r = make(None)
if not a.no_retune:
    r.collect_cost(True)
    r.route_resident(12, True)
    print("tuning window ms_main", r.last_stats["phase0"]["ms_main"], flush=True)
    hint = r.iteration_hint()
    r.close()
    r = make(hint)
for mode in ([False] if a.only_full else [True, False] if a.full else [True]):
    if not mode:           # a router built for the general mode
        r.close()
        r = make(None, False)
    print("engine", r.plan0.engine, flush=True)
    for i in range(a.reps):
        t0 = time.perf_counter()
        r.route_resident(12, mode)
        st = r.last_stats["phase0"]
        print(f"short_ts={mode} rep {i}: ms_main {st['ms_main']:.3f} wall {1e3 * (time.perf_counter() - t0):.3f} launches {st['main_launches']}",
              flush=True)
r.close()
