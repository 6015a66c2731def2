"""Diagnostic: how many CONUS rows are over bankfull / which secant class, at a few times of the bench window."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
q0 = np.zeros((nseg, 3), np.float32)
bw, tw, twcc, cs = params[:, 2], params[:, 3], params[:, 4], params[:, 7]
z = np.where(cs == 0, 1.0, 1.0 / cs)
bfd = np.where(bw > tw, bw / 1e-5, np.where(bw == tw, bw / (2 * z), (tw - bw) / (2 * z)))
r = ShardedRouter(to, params)
for nsteps in (24, 96, 288):
    r.upload(nsteps, qlat, q0)
    r.route_resident(12, True)
    st = r.plan0.download_final_state()
    it = r.plan0.download_iterations()
    d = st[:, 2]
    over = d > bfd
    wet = it >= 2
    print(f"t={nsteps}: over bankfull {over.mean():.3f} of all rows, {over[wet].mean():.3f} of rows with >=2 iterations; "
          f"class hist {np.bincount(np.minimum(it, 3), minlength=4)}")
    # how mixed are waves in plan order?  (positions of the level-major order, 64 at a time)
    lvl, pos = r.plan0.levels()
    order = np.argsort(pos)
    o = over[order][: nseg // 64 * 64].reshape(-1, 64)
    w = wet[order][: nseg // 64 * 64].reshape(-1, 64)
    both = ((o & w).any(1) & (~o & w).any(1)).mean()
    print(f"      waves (plan order) holding both in-bank and over-bank wet lanes: {both:.3f}; only in-bank {((~o & w).any(1) & ~(o & w).any(1)).mean():.3f}; only over {((o & w).any(1) & ~(~o & w).any(1)).mean():.3f}")
