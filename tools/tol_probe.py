#!/usr/bin/env python3
"""Where does the tolerance arithmetic leave the exact one?  Kernel vectors of the reference's test suite: the vectors whose
tolerance result is not finite where the exact one is, and the error quantiles by iteration-count agreement."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H  # noqa: E402
from troute_amd.plan import segments  # noqa: E402

kv = H.load_kernel_vectors()
x = np.ascontiguousarray(kv["inputs_f64"].astype(np.float32))
names = "dt qup quc qdp ql dx bw tw twcc n ncc cs s0 velp depthp".split()
ex, ie = segments(x, arithmetic="exact", with_iterations=True)
tl, it = segments(x, arithmetic="tolerance", with_iterations=True)
bad = np.isfinite(ex) & ~np.isfinite(tl)
print("vectors", len(x), "non-finite in tolerance only:", int(bad.any(axis=1).sum()), "by column", bad.sum(axis=0))
for i in np.flatnonzero(bad.any(axis=1))[:12]:
    print(i, "tag", kv["tags"][i] if "tags" in kv.files else "", dict(zip(names, x[i].tolist())))
    print("   exact", ex[i], "iters", ie[i], "\n   tol  ", tl[i], "iters", it[i])
ok = np.isfinite(ex).all(axis=1) & np.isfinite(tl).all(axis=1)
same = ok & (ie == it)
print("finite both", int(ok.sum()), "same iteration count", int(same.sum()), "different", int((ok & (ie != it)).sum()))
for col, nm in enumerate(("q", "vel", "depth", "ck", "cn", "X")):
    e = np.abs(tl[same, col].astype(np.float64) - ex[same, col])
    rel = e / np.maximum(np.abs(ex[same, col]), 1e-6)
    print(f"  {nm:6s} same-count rel quantiles 50/90/99/99.9/max: " + " ".join(f"{np.quantile(rel, q):.2e}" for q in (0.5, 0.9, 0.99, 0.999, 1.0))
          + f"   abs max {e.max():.3e}")
worst = np.argsort(-(np.abs(tl[:, 0].astype(np.float64) - ex[:, 0]) / np.maximum(np.abs(ex[:, 0]), 1e-6)) * same)[:8]
for i in worst:
    print("worst same-count", i, dict(zip(names, x[i].tolist())), "\n   exact", ex[i], ie[i], "\n   tol  ", tl[i], it[i])
flip = ok & (ie != it)
for col, nm in enumerate(("q", "vel", "depth")):
    if flip.any():
        e = np.abs(tl[flip, col].astype(np.float64) - ex[flip, col])
        rel = e / np.maximum(np.abs(ex[flip, col]), 1e-6)
        print(f"  {nm:6s} other-count rel quantiles 50/90/99/max: " + " ".join(f"{np.quantile(rel, q):.2e}" for q in (0.5, 0.9, 0.99, 1.0)))

# the same vectors with their state and inflows scaled (a wetter / drier channel): anything not finite in tolerance only?
rng = np.random.default_rng(5)
more = x[rng.integers(0, len(x), 200000)].copy()
more[:, [1, 2, 3, 4]] *= rng.lognormal(0.0, 1.0, (more.shape[0], 1)).astype(np.float32)
more[:, 14] *= rng.lognormal(0.0, 0.5, more.shape[0]).astype(np.float32)
ex2, ie2 = segments(more, arithmetic="exact", with_iterations=True)
tl2, it2 = segments(more, arithmetic="tolerance", with_iterations=True)
bad2 = np.isfinite(ex2) & ~np.isfinite(tl2)
print("perturbed vectors", len(more), "non-finite in tolerance only:", int(bad2.any(axis=1).sum()), "by column", bad2.sum(axis=0),
      "non-finite in exact:", int((~np.isfinite(ex2)).any(axis=1).sum()))
for i in np.flatnonzero(bad2.any(axis=1))[:10]:
    print(i, dict(zip(names, more[i].tolist())))
    print("   exact", ex2[i], "iters", ie2[i], "\n   tol  ", tl2[i], "iters", it2[i])
