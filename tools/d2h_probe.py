import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from troute_amd import synthetic
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
n = net["to"].shape[0]
r = ShardedRouter(net["to"], net["params"])
r.upload(288, net["qlat"], np.zeros((n, 3), np.float32))
for k in range(3):
    t0 = time.perf_counter(); r.route_resident(12, True); t1 = time.perf_counter()
    h = r.outlet_hydrographs(); t2 = time.perf_counter()
    s = r.plan0.download_final_state(); t3 = time.perf_counter()
    print(f"route {1e3*(t1-t0):.2f} ms  outlets {1e3*(t2-t1):.2f} ms  final state {1e3*(t3-t2):.2f} ms", flush=True)
    del h, s
