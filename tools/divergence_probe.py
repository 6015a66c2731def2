#!/usr/bin/env python3
"""How uniform are wavefronts in cost?  Routes a tuning window on the GPU, takes every row's mean iteration class, and
reports, for the level-major order and for the block order (with and without the hint), the lane mean against the mean
over wavefronts of the costliest lane -- a proxy of the instruction inflation divergence causes.
    python tools/divergence_probe.py [--nseg N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402
from troute_amd.plan import topology_blocks, topology_levels  # noqa: E402
from troute_amd.synthetic import upstream_csr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nseg", type=int, default=None)
a = ap.parse_args()
kw = {"nseg": a.nseg, "nnet": max(3, a.nseg // 185)} if a.nseg else {}
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), **kw)
to, params, qlat = net["to"], net["params"], net["qlat"]
n = to.shape[0]
r = ShardedRouter(to, params)
r.upload(288, qlat, np.zeros((n, 3), np.float32))
r.collect_cost(True)
r.route_resident(12, True)
hint = r.iteration_hint()
cost = hint.astype(np.float64) / 16.0          # mean of min(iterations, 3) over the window
r.close()
print("lane mean class %.3f; histogram of hint/16 rounded:" % cost.mean(), np.bincount(np.round(cost).astype(int)))
up_ptr, up_idx = upstream_csr(to)


def report(name, pos):
    order = np.argsort(pos)
    c = cost[order]
    pad = (-len(c)) % 64
    c = np.concatenate([c, np.zeros(pad)]).reshape(-1, 64)
    print(f"{name:38s} wave-max mean {c.max(1).mean():.3f}  (x{c.max(1).mean() / cost.mean():.2f} of the lane mean)")
    if "block" in name:
        b = np.concatenate([cost[order], np.zeros((-len(order)) % 256)]).reshape(-1, 256)
        print(f"{'':38s} block-max mean {b.max(1).mean():.3f}")


_, pos, _ = topology_levels(up_ptr, up_idx)
report("level order, no hint", pos)
_, pos, _ = topology_levels(up_ptr, up_idx, cost_hint=hint)
report("level order, hinted", pos)
pos, rank, br, nb = topology_blocks(up_ptr, up_idx)
report("block order, no hint", pos)
print("   max block rank", rank.max(), "mean", rank.mean())
pos, rank, br, nb = topology_blocks(up_ptr, up_idx, cost_hint=hint)
report("block order, hinted", pos)
print("   max block rank", rank.max(), "mean", rank.mean())
