"""Cluster tiles (k_mc_ctile) against the one-step tail (k_mc_step) on the bench network: launches, window time, and the
results of both compared bit for bit (final state of every row, the flow / velocity / depth series of sampled rows).
  python tools/cluster_probe.py [--nseg N] [--rows 128,64] [--windows 3]"""
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
ap.add_argument("--rows", default="128")
ap.add_argument("--windows", type=int, default=3)
ap.add_argument("--hint", action="store_true")
ap.add_argument("--nsteps", type=int, default=288)
ap.add_argument("--wide-min-rows", type=int, default=0)
ap.add_argument("--wide-k", type=int, default=0)
ap.add_argument("--skip-ref", action="store_true")
a = ap.parse_args()
_lib.single_hw_queue_per_priority("cluster_probe")
nnet = S.CONUS_NNET if a.nseg == S.CONUS_NSEG else max(1, a.nseg // 185)
net = S.generate(a.nseg, nnet, cache_dir=os.environ.get("TRMC_SYNTH_CACHE", "/tmp"))
up_ptr, up_idx = S.upstream_csr(net["to"])
nsteps = a.nsteps
qts = -(-nsteps // 24)
q0 = np.zeros((a.nseg, 3), np.float32)
ref = None
hint = None
for rows in ([] if a.skip_ref else [0]) + [int(x) for x in a.rows.split(",")]:
    with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=True, engine="levels", cost_hint=hint,
                     options={"cluster_rows": rows, "wide_min_rows": a.wide_min_rows, "wide_k": a.wide_k}) as p:
        p.upload_forcing(nsteps, net["qlat"], q0)
        if a.hint and hint is None:
            p.collect_cost(True)
        st = p.route_device(nsteps, qts, True)
        if a.hint and hint is None:
            cost, ns = p.download_cost()
            hint_next = np.minimum(255, (cost.astype(np.float64) * 16 / ns).round()).astype(np.uint8)
        ms = []
        for _ in range(a.windows):
            p.upload_forcing(nsteps, net["qlat"], None)
            t0 = time.perf_counter()
            st = p.route_device(nsteps, qts, True)
            ms.append(round(st["ms_main"], 2))
        fin = p.download_final_state()
        rng = np.random.default_rng(5)
        sample = np.sort(rng.choice(a.nseg, 2000, replace=False))
        fvd = p.gather_flow_rows(sample) if hasattr(p, "gather_flow_rows") else None
        print(f"cluster_rows={rows}: launches {st['main_launches']} wide {st['wide_levels']} ms_main {ms} ms_emit {st['ms_emit']:.2f}", flush=True)
        if a.hint and hint is None:
            hint = hint_next
        if ref is None:
            ref = (fin, fvd)
        else:
            same_f = np.array_equal(ref[0].view(np.uint32), fin.view(np.uint32))
            same_h = fvd is None or np.array_equal(ref[1].view(np.uint32), fvd.view(np.uint32))
            print(f"   final state of every row bit-identical: {same_f}; sampled hydrographs: {same_h}", flush=True)
