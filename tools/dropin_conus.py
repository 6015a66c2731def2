#!/usr/bin/env python3
"""bench.py's `dropin` leg by itself (compute_nhd_routing_v02 from DataFrames at the bench network's size), with a cProfile of
the steady-state call on stderr:  TRMC_DROPIN_PROFILE=1 python tools/dropin_conus.py [--nseg N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TRMC_DROPIN_PROFILE", "1")
import bench  # noqa: E402
from troute_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nseg", type=int, default=None)
a = ap.parse_args()
kw = {"nseg": a.nseg, "nnet": max(3, a.nseg // 185)} if a.nseg else {}
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), **kw)
n = net["to"].shape[0]
days = [net["qlat"], synthetic.forcing(n, previous=net["qlat"], seed=synthetic.DEFAULT_SEED + 1)]
print(json.dumps(bench.dropin_leg(net, 288, 12, days), indent=1))
