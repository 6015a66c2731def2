#!/usr/bin/env python3
"""Experiment: how many rows the hot lists of k_mc_tile carry per launch (trmc_plan_hot_rows) on the plan built from the
topology alone and on the cost-ordered plan, and what a resident CONUS day takes with and without them."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
to, params, qlat = net["to"], net["params"], net["qlat"]
n = to.shape[0]
q0 = np.zeros((n, 3), np.float32)
os.environ["TRMC_HOT_ROWS"] = "1"
r = ShardedRouter(to, params, assume_short_ts=True, engine="levels")
r.upload(288, qlat, q0); r.route_resident(12, True); r.upload(288, qlat, None)
r.collect_cost(True); r.route_resident(12, True)
hint = r.iteration_hint()
a = r.plan0.hot_rows(); r.route_resident(12, True); b = r.plan0.hot_rows()
print("unhinted: rows from the list per day", b - a, "per launch", (b - a) / 23, "ms", r.last_stats["phase0"]["ms_main"])
r.close()
for hot in ("1", "0"):
    os.environ["TRMC_HOT_ROWS"] = hot
    r = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True, engine="levels")
    r.upload(288, qlat, q0); r.route_resident(12, True); r.upload(288, qlat, None); r.route_resident(12, True)
    a = r.plan0.hot_rows(); r.route_resident(12, True); b = r.plan0.hot_rows()
    print("hinted hot", hot, ": rows from the list per day", b - a, "per launch", (b - a) / 23, "ms", r.last_stats["phase0"]["ms_main"])
    r.close()
