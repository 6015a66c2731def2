#!/usr/bin/env python3
"""CPU baseline alone (bench.py's cpu_baseline) at several thread counts:  python tools/cpu_probe.py 64 128 256"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("INTERLEAVE"):          # set_mempolicy(MPOL_INTERLEAVE, nodes 0-1): pages of later allocations alternate
    import ctypes
    mask = ctypes.c_ulong(3)
    rc = ctypes.CDLL(None, use_errno=True).syscall(238, 3, ctypes.byref(mask), 65)
    print("set_mempolicy rc", rc, flush=True)
import bench  # noqa: E402
from troute_amd import synthetic  # noqa: E402
kw = {}
if os.environ.get("NSEG"):
    kw = {"nseg": int(os.environ["NSEG"]), "nnet": max(3, int(os.environ["NSEG"]) // 185)}
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), **kw)
for th in [int(x) for x in sys.argv[1:]] or [0]:
    r = bench.cpu_baseline(net, net["qlat"], 288, 12, True, float(os.environ.get("CPU_SECONDS", "6")), th)
    r.pop("_check", None)
    print(th, "%.3e" % r["value"], "per thread %.3e" % r["per_thread"], r["order_seconds"], r["sample"][:60], flush=True)
