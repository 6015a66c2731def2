#!/usr/bin/env python3
"""Where the time of the drop-in callable goes (LowerColorado, 11 248 segments x 288 steps): wall time of
compute_network_structured called repeatedly on the same network, and a cProfile of one call."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args  # noqa: E402

lc = H.LowerColorado()
args = mc_only_args(lc.nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols, lc.data_values, lc.q0, lc.qlat,
                    assume_short_ts=True)
for k in range(4):
    t0 = time.perf_counter()
    r = compute_network_structured(*args)
    print(f"call {k}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
compute_network_structured(*args)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
