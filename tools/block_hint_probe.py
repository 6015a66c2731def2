#!/usr/bin/env python3
"""Experiment: how much of what a cost-ordered plan gains is the ORDER OF BLOCKS inside a launch (costly blocks first: the
launch does not end on a straggler) and how much the grouping of ROWS into wavefronts of one cost?  The plan built from the
topology alone is rebuilt with a hint that is constant over every block of 128 consecutive positions (the block's mean
measured cost): whole blocks move, their rows stay together."""
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


def run(hint, label):
    r = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True, engine="levels")
    r.upload(288, qlat, q0)
    best = 1e9
    for _ in range(3):
        r.route_resident(12, True)
        best = min(best, r.last_stats["phase0"]["ms_main"])
    print(f"{label:60s} ms_main {best:.3f}", flush=True)
    return r


r = run(None, "no hint")
r.collect_cost(True)
r.route_resident(12, True)
measured = r.iteration_hint()
lvl, pos = r.plan0.levels()
r.close()
run(measured, "measured hint (rows)").close()
for B in (128, 512, 2048):
    order = np.argsort(pos, kind="stable")                     # rows in plan order
    cost = measured[order].astype(np.float64)
    blk = np.arange(n) // B
    mean = np.bincount(blk, cost) / np.bincount(blk)
    for q in (16, 4):
        h = np.empty(n, np.uint8)
        h[order] = np.minimum(np.round(mean[blk] / q) * q, 255).astype(np.uint8)
        run(h, f"hint = mean of the block of {B} positions, quantised to {q}").close()
