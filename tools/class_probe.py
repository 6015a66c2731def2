#!/usr/bin/env python3
"""Cost of a segment-step by secant-iteration class on the level engine (k_mc_step), free of divergence: for every class a
plan of N independent one-segment networks whose forcing puts EVERY row in that class (dry: no flow, 0 iterations; trickle:
the depth stays under the 1 cm floor, 1 iteration; wet: 2 iterations; flood: over bank), routed warm for `nsteps` steps.
Prints the iteration histogram reached, microseconds per launch and nanoseconds per thousand rows; the plans have different
sizes (N + 4096 k), so that `tools/rocpd_dispatches.py db k_mc_step` separates the classes of a counter pass by grid size.
    python tools/class_probe.py [--rows 1048576] [--nsteps 48] [--reps 3]      (TRMC_LIB_PATH selects a library variant)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic  # noqa: E402
from troute_amd.plan import RoutingPlan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--nsteps", type=int, default=48)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--classes", default="dry,trickle,wet,wet2,flood,mixed,sorted2,inter2,inter2w")
a = ap.parse_args()
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
rng = np.random.default_rng(1)
FORCING = {   # lateral inflow per row [m3/s]
    "dry": lambda n: np.zeros(n, np.float32),
    "trickle": lambda n: np.exp(rng.uniform(np.log(1e-8), np.log(2e-6), n)).astype(np.float32),
    "wet": lambda n: np.exp(rng.uniform(np.log(0.02), np.log(0.8), n)).astype(np.float32),
    "flood": lambda n: np.exp(rng.uniform(np.log(300.0), np.log(3000.0), n)).astype(np.float32),
    "wet2": lambda n: np.exp(rng.uniform(np.log(0.05), np.log(0.12), n)).astype(np.float32),   # typical channel: all rows 2 iterations
    "mixed": None,
    # half the rows two-iteration, half one-iteration: the two classes one after the other (the order a cost-sorted level
    # has), dealt out in turn by blocks of 128 rows (a step-kernel workgroup), by wavefronts of 64
    "sorted2": None, "inter2": None, "inter2w": None,
}
TYPICAL = np.array([300.0, 1500.0, 3.0, 5.0, 15.0, 0.06, 0.12, 0.6, 0.006], np.float32)
for k, name in enumerate(a.classes.split(",")):
    n = a.rows + 4096 * k
    rows = rng.integers(0, net["params"].shape[0], n)
    params = net["params"][rows].copy()
    if name == "mixed":      # the bench network's own forcing on unconnected rows: classes as they come, unsorted
        ql = net["qlat"][rows, 3].copy()
    elif name in ("sorted2", "inter2", "inter2w"):
        params[:] = TYPICAL
        wet = FORCING["wet2"](n)
        tri = FORCING["trickle"](n)
        chunk = {"sorted2": n // 2, "inter2": 128, "inter2w": 64}[name]
        pick = (np.arange(n) // chunk) % 2 == 0
        ql = np.where(pick, wet, tri)
    else:
        ql = FORCING[name](n)
        if name == "wet2":
            params[:] = TYPICAL
    nq = (a.nsteps - 1) // 12 + 1
    qlat = np.repeat(ql[:, None], nq, axis=1) * (1.0 + 0.05 * np.sin(np.arange(nq)))[None, :].astype(np.float32)
    up_ptr = np.zeros(n + 1, np.int64)
    up_idx = np.zeros(0, np.int64)
    with RoutingPlan(up_ptr, up_idx, params, engine="levels", assume_short_ts=True) as plan:
        plan.upload_forcing(a.nsteps, qlat, np.zeros((n, 3), np.float32))
        plan.route_device(a.nsteps, 12, True)                       # spin-up from a cold start
        plan.upload_forcing(a.nsteps, qlat, None)
        plan.collect_cost(True)
        ms = []
        for _ in range(a.reps):
            st = plan.route_device(a.nsteps, 12, True)
            ms.append(st["ms_main"] / st["main_launches"])
        cost, ns = plan.download_cost()
        it = plan.download_iterations()
    hist = np.bincount(np.minimum(it, 4), minlength=5) / n
    us = min(ms) * 1e3
    print(f"{name:8s} rows {n:8d}  last-step iterations 0/1/2/3/4+: " + " ".join(f"{h:.3f}" for h in hist)
          + f"  mean cost per step {cost.mean() / ns:.3f}  -> {us:7.2f} us per launch, {us * 1e3 / (n / 1e3):7.2f} ns per 1000 rows", flush=True)
