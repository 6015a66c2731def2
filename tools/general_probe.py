"""The general mode (no assume_short_ts: a row reads its upstream rows at the SAME step, mc_reach.pyx:499-502) over several days:
D days routed one by one (D windows of 288 steps) against the same D days as ONE window of 288 D steps on the same plan -- the
final state and sampled hydrographs bit for bit, and what a day costs either way.
  python tools/general_probe.py [--nseg N] [--days 1,2,4,8] [--hint]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic as S                     # noqa: E402
from troute_amd.plan import RoutingPlan                   # noqa: E402
from troute_amd import _lib                               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nseg", type=int, default=S.CONUS_NSEG)
ap.add_argument("--days", default="1,2,4,8")
ap.add_argument("--hint", action="store_true")
ap.add_argument("--engine", default="auto")
a = ap.parse_args()
nnet = S.CONUS_NNET if a.nseg == S.CONUS_NSEG else max(1, a.nseg // 185)
net = S.generate(a.nseg, nnet, cache_dir=os.environ.get("TRMC_SYNTH_CACHE", "/tmp"))
n = a.nseg
up_ptr, up_idx = S.upstream_csr(net["to"])
nsteps, qts = 288, 12
Ds = [int(x) for x in a.days.split(",")]
days = [np.ascontiguousarray(net["qlat"][:, :nsteps // qts])]
for d in range(1, min(max(Ds), 4)):
    days.append(np.ascontiguousarray(S.forcing(n, previous=net["qlat"] if d == 1 else prev, seed=S.DEFAULT_SEED + 1 + d)))
    prev = days[-1]
    days[-1] = np.ascontiguousarray(days[-1][:, :nsteps // qts])
q0 = np.zeros((n, 3), np.float32)
sample = np.sort(np.random.default_rng(5).choice(n, min(n, 3000), replace=False))
hint = None
if a.hint:
    with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=False, engine=a.engine) as p:
        p.upload_forcing(nsteps, days[0], q0)
        p.collect_cost(True)
        p.route_device(nsteps, qts, False)
        cost, ns = p.download_cost()
        hint = np.minimum(255, (cost.astype(np.float64) * 16 / ns).round()).astype(np.uint8)
with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=False, engine=a.engine, cost_hint=hint) as p:
    print("engine", p.engine, flush=True)
    for D in Ds:
        # one by one
        hyd1, t1 = [], 0.0
        for d in range(D):
            p.upload_forcing(nsteps, days[d % len(days)], q0 if d == 0 else None)
            st = p.route_device(nsteps, qts, False)
            t1 += st["ms_main"]
            hyd1.append(p.gather_flow_rows(sample))
        fin1 = p.download_final_state()
        # as one window
        q = np.ascontiguousarray(np.concatenate([days[d % len(days)] for d in range(D)], axis=1))
        p.upload_forcing(nsteps * D, q, q0)
        t0 = time.perf_counter()
        st = p.route_device(nsteps * D, qts, False)
        wall = (time.perf_counter() - t0) * 1e3
        finD = p.download_final_state()
        hydD = p.gather_flow_rows(sample)
        same = np.array_equal(fin1.view(np.uint32), finD.view(np.uint32)) and np.array_equal(
            np.concatenate(hyd1, axis=1).view(np.uint32), hydD.view(np.uint32))
        print(f"D = {D}: one by one {t1 / D:.2f} ms per day; one window of {nsteps * D} steps {st['ms_main'] / D:.2f} ms per day "
              f"(ms_main {st['ms_main']:.1f}, wall {wall:.1f}, launches {st.get('launches')}); final state and {sample.shape[0]} hydrographs "
              f"bit-identical: {same}", flush=True)
