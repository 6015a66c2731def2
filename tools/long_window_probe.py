#!/usr/bin/env python3
"""What does a routing window of SEVERAL days cost per day?  (A forecast cycle has the forcing of its whole horizon; the
reference routes it in loops of max_loop_size hours, 24 by default.)  The level skew's ramps and the tail's lonely end are
paid once per window, not once per day.   python tools/long_window_probe.py [days ...]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

days_list = [int(x) for x in sys.argv[1:]] or [1, 2, 3]
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
to, params = net["to"], net["params"]
nseg = to.shape[0]
nsteps, qts = 288, 12
qlat_s = net["qlat"]
qlat_a = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 1, previous=qlat_s)
ring, prev = [], qlat_a
for i in range(max(days_list)):
    d = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2 + i, previous=prev)
    ring.append(d)
    prev = d
q0 = np.zeros((nseg, 3), np.float32)
r = ShardedRouter(to, params, assume_short_ts=True)
r.upload(nsteps, qlat_s, q0)
r.route_resident(qts, True)
r.upload(nsteps, qlat_a, None)
r.collect_cost(True)
r.route_resident(qts, True)
hint = r.iteration_hint()
r.close()
r = ShardedRouter(to, params, assume_short_ts=True, cost_hint=hint)
r.upload(nsteps, qlat_s, q0)
r.route_resident(qts, True)
r.upload(nsteps, qlat_a, None)
r.route_resident(qts, True)
state = r.plan0.download_final_state()
for D in days_list:
    # D days as one window: 24 hourly columns of every day but the last, 25 of the last
    ql = np.ascontiguousarray(np.concatenate([ring[i][:, :24] for i in range(D - 1)] + [ring[D - 1]], axis=1))
    ms = []
    for _ in range(3):
        r.upload(nsteps * D, ql, state)
        t0 = time.perf_counter()
        r.route_resident(qts, True)
        ms.append(r.last_stats["phase0"]["ms_main"])
    print(f"{D} day(s) per window: ms_main {min(ms):8.2f}  = {min(ms) / D:6.2f} ms per day", flush=True)
r.close()
