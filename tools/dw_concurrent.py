"""Several tailwater domains at once: ONE launch whose blocks are the domains (trdw_diffnw_batch) against the same
domains one call after the other."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd.routing.fast_reach import diffusive as D
z = np.load("tests/golden/diffusive_lowercolorado.npz")
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
ins["timestep_ar_g"] = ins["timestep_ar_g"].copy(); ins["timestep_ar_g"][2] = 300.0 * nsteps / 3600.0
ins["ntss_ev_g"] = np.array(nsteps + 1)
ref = D.compute_diffusive(ins)
t0 = time.perf_counter(); D.compute_diffusive(ins); one = time.perf_counter() - t0
for n in (2, 8, 32, 128, 256):
    t0 = time.perf_counter(); outs = D.compute_diffusive_batch([ins] * n); el = time.perf_counter() - t0
    same = all(np.array_equal(o[2], ref[2]) and np.array_equal(o[0], ref[0]) for o in outs)
    print(f"{n:4d} domains in one launch: {el:.2f} s  (one alone {one:.2f} s, one after the other {n * one:.1f} s) identical: {same}; solve {D.last_timing()[1]:.0f} ms")
