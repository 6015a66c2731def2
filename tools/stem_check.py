#!/usr/bin/env python3
"""The general mode (assume_short_ts=False) on the bench network, dataflow engine, with the plain block order
(TRMC_STEM_MIN_ROWS=0) and with long stems laid out last and started first (the default, csrc/topology.hpp stem_min_rows):
device time of the second window, and the two results compared bit for bit -- the final state of every row, the flow series
of 30 000 random rows, every outlet hydrograph.  (Developer tool, GPU box; the arithmetic itself is pinned to the oracle by
the tests.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from troute_amd import synthetic
from troute_amd.plan import RoutingPlan
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
up_ptr, up_idx = synthetic.upstream_csr(to)
q0 = np.zeros((nseg, 3), np.float32)
rng = np.random.default_rng(1)
rows = np.sort(rng.choice(nseg, 30000, replace=False))
outlets = np.flatnonzero(to < 0)
res = {}
variants = sys.argv[1:] or ["1024"]                                 # python tools/stem_check.py [stem_min_rows ...]
for stem in ["0"] + variants:
    os.environ["TRMC_STEM_MIN_ROWS"] = stem
    with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=False, engine="flow") as plan:
        plan.upload_forcing(288, qlat, q0)
        for rep in range(2):
            t0 = time.perf_counter()
            plan.route_device(288, 12, False)
            ms = plan.stats()["ms_main"]
        res[stem] = (plan.download_final_state(), plan.gather_flow_rows(rows), plan.gather_flow_rows(outlets), ms)
        print("stem", stem, "ms_main", ms, flush=True)
a = res["0"]
for stem in variants:
    b = res[stem]
    for k, name in enumerate(("final state of every row", "flow series of 30 000 rows", "outlet hydrographs")):
        same = np.array_equal(np.ascontiguousarray(a[k]).view(np.uint32), np.ascontiguousarray(b[k]).view(np.uint32))
        print(f"stems of at least {stem} rows: {name}", "identical" if same else "DIFFERENT")
