import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import helpers as H
from oracle import oracle as O
from troute_amd.plan import RoutingPlan, csr_from_lists, topology_levels
O.build()
def bits(a): return np.ascontiguousarray(a).view(np.uint32)
tot = 0
for seed in range(1, 9):
    rng = np.random.default_rng(1000 + seed)
    nseg = 40000
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    lvl, _, _ = topology_levels(up_ptr, up_idx)
    logu = lambda lo, hi, n: np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    lo = 2.0 ** -14
    bw = logu(lo * 1.01, 2.0 ** 16.9, nseg)
    tw = bw * rng.uniform(0.8, 3.0, nseg)
    twcc = np.where(rng.random(nseg) < 0.2, 0.0, tw * rng.uniform(1.0, 4.0, nseg))
    n = logu(lo * 1.01, 2.0, nseg)
    ncc = np.where(rng.random(nseg) < 0.1, 0.0, n * rng.uniform(0.5, 3.0, nseg))
    cs = np.where(rng.random(nseg) < 0.05, 0.0, logu(lo * 1.01, 2.0 ** 16.9, nseg))
    params = np.stack([np.full(nseg, rng.choice([60.0, 300.0, 3600.0])), logu(1.0, 9e4, nseg), bw, tw, twcc, n, ncc, cs,
                       logu(1e-6, 4.0, nseg)], 1).astype(np.float32)
    qlat = (logu(1e-12, 50.0, (nseg, 4)) * (rng.random((nseg, 4)) > 0.2)).astype(np.float32)
    q0 = np.stack([logu(1e-12, 500.0, nseg), logu(1e-12, 500.0, nseg), logu(1e-14, 2e5, nseg)], 1).astype(np.float32)
    q0[rng.random(nseg) < 0.2] = 0
    for short in (True, False):
        nsteps, qts = 36, 9
        want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, short, det=True)[:, 1:, :]
        fin = np.isfinite(want).all(axis=(1, 2))
        with RoutingPlan(up_ptr, up_idx, params) as plan:
            got = plan.route(nsteps, qts, short, qlat, q0)
            it = plan.download_iterations()
        with RoutingPlan(up_ptr, up_idx, params, cost_hint=np.minimum(it, 3)) as plan:
            got2 = plan.route(nsteps, qts, short, qlat, q0)
        ok = np.array_equal(bits(got[fin]), bits(want[fin])) and np.array_equal(bits(got2[fin]), bits(want[fin]))
        tot += 1
        print(seed, short, "finite rows", fin.mean(), "max iters", it.max(), "OK" if ok else "MISMATCH", flush=True)
        assert ok
print("all", tot, "cases bit-identical")
