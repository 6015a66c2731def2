#!/usr/bin/env python3
"""Experiment (numpy on measured classes): what would the tiles of an UNORDERED plan execute if the rows of class >= 3 (three
or more secant iterations, over-bank) were taken out of their blocks and routed by blocks of their own?  Classes of the
last step of a CONUS day on the plan built from the topology alone; cost of a wavefront-step by its costliest lane
(instructions per class from the counters of the tuned / lean kernels: dry 130, one iteration 600, two 890, three 1 150,
over bank twice that)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
to, params, qlat = net["to"], net["params"], net["qlat"]
n = to.shape[0]
r = ShardedRouter(to, params, assume_short_ts=True, engine="levels")
r.upload(288, qlat, np.zeros((n, 3), np.float32))
r.route_resident(12, True)
r.upload(288, qlat, None)
r.collect_cost(True)
r.route_resident(12, True)
it = r.plan0.download_iterations().astype(np.int64)          # iterations of the last step, per row
cost16, nst = r.plan0.download_cost()                        # sum over the window of min(it, 3) + 4 * over, per row
lvl, pos = r.plan0.levels()
st = r.plan0.stats()
W = st["wide_levels"]
r.close()
order = np.argsort(pos, kind="stable")
cls = np.minimum(it[order], 3)
mean_cls = cost16[order].astype(np.float64) / max(nst, 1)
over = mean_cls > 3.5                                        # mostly over bank through the day
wide = lvl[order] < W
cls, over = cls[wide], over[wide]
print(f"wide levels {W}, rows {cls.size}; classes 0..3: {np.bincount(cls, minlength=4) / cls.size}; over bank {over.mean():.4f}")
INS = np.array([130.0, 600.0, 890.0, 1150.0])
lane_cost = INS[cls] * np.where(over, 2.0, 1.0)
m = cls.size // 128 * 128
lc = lane_cost[:m].reshape(-1, 128)


def waves(lc):
    s = -np.sort(-lc, axis=1)                                # the in-block partition: descending by class
    return np.concatenate([s[:, :64].max(1), s[:, 64:].max(1)])


base = waves(lc).mean()
plain = np.concatenate([lc[:, :64].max(1), lc[:, 64:].max(1)]).mean()
hot = (cls[:m] >= 3) | over[:m]
cold = np.where(hot.reshape(-1, 128), 0.0, lc)               # the hot rows' lanes idle at home
nhot = int(hot.sum())
hot_cost = np.sort(lane_cost[:m][hot])[::-1]
pad = (-nhot) % 64
hw = np.concatenate([hot_cost, np.zeros(pad)]).reshape(-1, 64).max(1)
total_waves = m // 64
after = (waves(cold).sum() + hw.sum()) / total_waves
ideal = np.sort(lane_cost[:m])[::-1].reshape(-1, 64).max(1).mean()
print(f"per wavefront-step: plan order {plain:.0f}; in-block partition {base:.0f}; hot rows ({nhot}, {nhot / m:.4f}) in blocks of their own "
      f"{after:.0f}; rows sorted over the whole range {ideal:.0f}")
