"""Probe: per-launch time of narrow wet launches (trunk of an 8-way partition with constant inflows)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic, sharding
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
q0 = np.zeros((nseg, 3), np.float32)
part = sharding.partition(to, 8)
r = ShardedRouter(to, params, rank=0, world=8, partition=part)
for inflow in (0.0, 0.5):
    bf = np.zeros((int(r.boundary1.sum()), 288, 3), np.float32)
    bf[:, :, 0] = inflow
    r.plan1.upload_forcing(288, qlat[r.rows1], q0[r.rows1], bf)
    for _ in range(3):
        st = r.plan1.route_device(288, 12, True)
    print(f"trunk {r.rows1.size} rows, inflow {inflow}: {st['ms_main'] / 288 * 1e3:.1f} us per launch")
r.upload(288, qlat, q0)
for _ in range(3):
    st = r.plan0.route_device(288, 12, True)
print(f"sub-basins {r.rows0.size} rows: {st['ms_main'] / 288 * 1e3:.1f} us per launch")
