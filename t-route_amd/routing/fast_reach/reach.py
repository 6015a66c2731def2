"""Drop-in for ``troute.routing.fast_reach.reach`` (single-segment entry points).

``compute_reach_kernel`` mirrors the dict-returning test entry of the
reference (src/troute-routing/troute/routing/fast_reach/reach.pyx:66-103):
one Muskingum-Cunge segment-timestep, evaluated by the HIP kernel.
``muskingcunge_batch`` is the array form (n independent segment-steps per
launch) -- the GPU counterpart of calling c_muskingcungenwm n times
(src/kernel/muskingum/pyMCsingleSegStime_NoLoop.f90:8-21).
"""
import numpy as np

from ...plan import segments

_IN = ("dt", "qup", "quc", "qdp", "ql", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0", "velp", "depthp")


def muskingcunge_batch(inputs, device=0):
    """inputs [n,15] (float32 or float64, column order ``_IN``) -> [n,6] qdc velc depthc ck cn X."""
    return segments(inputs, device=device)


def compute_reach_kernel(dt, qup, quc, qdp, ql, dx, bw, tw, twcc, n, ncc, cs, s0, velp, depthp,
                         precision=32, device=0):
    """One segment, one timestep; returns the reference's QVD dict
    (reach.pxd:1-7: qdc, velc, depthc, cn, ck, X)."""
    dtype = np.float32 if precision == 32 else np.float64
    row = np.array([[dt, qup, quc, qdp, ql, dx, bw, tw, twcc, n, ncc, cs, s0, velp, depthp]], dtype=dtype)
    o = segments(row, device=device)[0]
    return {"qdc": float(o[0]), "velc": float(o[1]), "depthc": float(o[2]), "cn": float(o[4]),
            "ck": float(o[3]), "X": float(o[5])}
