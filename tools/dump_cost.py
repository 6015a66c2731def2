import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from troute_amd import synthetic as S, _lib
from troute_amd.plan import RoutingPlan
_lib.single_hw_queue_per_priority("dump")
net = S.generate(cache_dir="/tmp")
up_ptr, up_idx = S.upstream_csr(net["to"])
n = net["to"].shape[0]
q0 = np.zeros((n, 3), np.float32)
with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=True, engine="levels") as p:
    p.upload_forcing(288, net["qlat"], q0)
    p.route_device(288, 12, True)
    p.upload_forcing(288, S.forcing(n, previous=net["qlat"]), None)
    p.collect_cost(True)
    p.route_device(288, 12, True)
    cost, ns = p.download_cost()
    it = p.download_iterations()
np.savez_compressed("gpurun_out/cost_day2.npz", cost=cost, it=it, nsteps=ns)
print(cost.dtype, cost.shape, ns, np.bincount(it)[:8])
