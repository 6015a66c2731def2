import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from troute_amd.routing.fast_reach import diffusive as D
z = np.load("tests/golden/diffusive_lowercolorado.npz")
ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
# shorten the window: tfin and the recording count
ins["timestep_ar_g"] = ins["timestep_ar_g"].copy(); ins["timestep_ar_g"][2] = 300.0 * nsteps / 3600.0
ins["ntss_ev_g"] = np.array(nsteps + 1)
t0 = time.time(); got = D.compute_diffusive(ins); el = time.time() - t0
print("steps", nsteps, "wall", round(el, 3), "tables/solve ms", D.last_timing())
