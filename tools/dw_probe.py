import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from troute_amd.routing.fast_reach import diffusive as D
def load(path, prefix=None):
    z = np.load(path)
    if prefix is None:
        return {k[3:]: z[k] for k in z.files if k.startswith("in_")}, (z["out_q"], z["out_elv"], z["out_depth"])
    d = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(prefix + "__")}
    return {k[3:]: v for k, v in d.items() if k.startswith("in_")}, (d["out_q"], d["out_elv"], d["out_depth"])
for nm in ("chain1", "y3", "comb"):
    ins, want = load("tests/golden/diffusive_small.npz", nm)
    t0 = time.time(); got = D.compute_diffusive(ins); el = time.time() - t0
    for g, w, lab in zip(got, want, "qed"):
        diff = np.abs(g - w)
        print(nm, lab, "max abs", diff.max(), "max rel", (diff / np.maximum(np.abs(w), 1e-9)).max(), "bitwise", np.array_equal(g, w), "time", round(el, 3), D.last_timing())
ins, want = load("tests/golden/diffusive_lowercolorado.npz")
t0 = time.time(); got = D.compute_diffusive(ins); el = time.time() - t0
for g, w, lab in zip(got, want, "qed"):
    diff = np.abs(g - w)
    print("LC", lab, "max abs", diff.max(), "n diff", (diff > 0).sum(), "n >1e-9", (diff > 1e-9).sum(), "of", (w != 0).sum(), "time", round(el, 3), D.last_timing())
