import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from troute_amd import synthetic
from troute_amd.plan import RoutingPlan
from troute_amd.synthetic import upstream_csr
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
q0 = np.zeros((nseg, 3), np.float32)
up_ptr, up_idx = upstream_csr(to)
plan = RoutingPlan(up_ptr, up_idx, params)
for n in (12, 100, 288):
    plan.upload_forcing(n, qlat, q0)
    plan.route_device(n, 12, True)
    it = plan.download_iterations()
    print(n, "max", it.max(), "hist", np.bincount(np.minimum(it, 12), minlength=13))
