import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from troute_amd import synthetic
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
q0 = np.zeros((nseg, 3), np.float32)
r = ShardedRouter(to, params)
for nsteps in (12, 48, 144, 288):
    r.upload(nsteps, qlat, q0)
    r.route_resident(12, True)
    it = r.plan0.download_iterations()
    h = np.bincount(it, minlength=8)
    print(nsteps, "hist0-7", h[:8], "max", it.max(), ">=10:", int((it >= 10).sum()), ">=100:", int((it >= 100).sum()))
