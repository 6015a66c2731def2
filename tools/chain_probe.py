#!/usr/bin/env python3
"""Latency of one dependent segment-step in the general mode (assume_short_ts=False): a single chain of N rows routed for
`nsteps` steps takes (N + nsteps) dependent steps.   python tools/chain_probe.py [--rows 4096] [--nsteps 288]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4096)
ap.add_argument("--nsteps", type=int, default=288)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), nseg=20000, nnet=100)
n = a.rows
to = np.arange(1, n + 1, dtype=np.int64)
to[-1] = -1
params = net["params"][:n].copy()
qlat = np.abs(net["qlat"][:n]).copy() + 0.01
q0 = np.zeros((n, 3), np.float32)
for short in (False, True):
    r = ShardedRouter(to, params, assume_short_ts=short, engine=os.environ.get("PROBE_ENGINE", "auto"))
    r.upload(a.nsteps, qlat, q0)
    for k in range(a.reps):
        r.route_resident(12, short)
    ms = r.last_stats["phase0"]["ms_main"]
    dep = (n + a.nsteps) if not short else a.nsteps
    print(f"engine {r.plan0.engine} short_ts={short}: ms_main {ms:.3f}  launches {r.last_stats['phase0']['main_launches']}  "
          f"-> {ms * 1e3 / dep:.2f} us per dependent step ({dep})", flush=True)
    r.close()
