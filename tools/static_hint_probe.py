#!/usr/bin/env python3
"""Experiment: a row-order hint from the topology alone (drainage size) against no hint and the measured hint."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
to, params, qlat = net["to"], net["params"], net["qlat"]
n = to.shape[0]
q0 = np.zeros((n, 3), np.float32)
# drainage size (rows draining through a row, itself included) by a pass in topological order
indeg = np.zeros(n, np.int64)
m = to >= 0
np.add.at(indeg, to[m], 1)
drain = np.ones(n, np.int64)
deg = indeg.copy()
front = np.flatnonzero(indeg == 0)
while front.size:
    t = to[front]
    ok = t >= 0
    np.add.at(drain, t[ok], drain[front[ok]])
    np.subtract.at(deg, t[ok], 1)
    cand = np.unique(t[ok])
    front = cand[deg[cand] == 0]


def run(hint, label):
    r = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True, engine="levels")
    r.upload(288, qlat, q0)
    best = 1e9
    for _ in range(3):
        r.route_resident(12, True)
        best = min(best, r.last_stats["phase0"]["ms_main"])
    print(f"{label:40s} ms_main {best:.3f}", flush=True)
    return r


r = run(None, "no hint (per-block class partition)")
r.collect_cost(True)
r.route_resident(12, True)
measured = r.iteration_hint()
r.close()
run(measured, "measured hint").close()
for scale in (2, 4, 8):
    h = np.minimum(scale * np.log2(drain.astype(np.float64)), 255).astype(np.uint8)
    run(h, f"hint = {scale} * log2(drainage size)").close()
run(np.zeros(n, np.uint8), "constant hint (plan order, no partition)").close()
h = (params[:, 2] > np.median(params[:, 2])).astype(np.uint8) * 16
run(h, "hint = bottom width above the median").close()
