"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle.

Wraps oracle/libmc_oracle.so (the C restatement, mc_oracle_impl.inc) and, when
present, the oracle/_ref/*.so builds of the reference Fortran
(oracle/Makefile `make ref`).  Never imported by the product package.

Column order of a segment input row (15): dt qup quc qdp ql dx bw tw twcc n ncc
cs s0 velp depthp ; output row (6): qdc velc depthc ck cn X  -- the argument
order of c_muskingcungenwm
(/root/reference/src/kernel/muskingum/pyMCsingleSegStime_NoLoop.f90:8-21).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

IN_COLS = ("dt", "qup", "quc", "qdp", "ql", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0",
           "velp", "depthp")
OUT_COLS = ("qdc", "velc", "depthc", "ck", "cn", "X")
# column order of the params block handed to network(): column_mapper order,
# /root/reference/src/troute-routing/troute/routing/fast_reach/mc_reach.pyx:150-162
PARAM_COLS = ("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")

_CT = {np.dtype("float32"): (C.c_float, "f32"), np.dtype("float64"): (C.c_double, "f64")}


def build(force=False):
    """Compile the C restatement (and, if /root/reference exists, oracle/_ref)."""
    so = os.path.join(_HERE, "libmc_oracle.so")
    if force or not os.path.exists(so) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
        for f in ("mc_oracle.c", "mc_oracle_impl.inc")
    ):
        subprocess.check_call(["make", "-C", _HERE, "libmc_oracle.so"], stdout=subprocess.DEVNULL)
    # host instantiation of the diffusive-wave solver (make decides whether it is stale)
    subprocess.check_call(["make", "-C", _HERE, "libdw_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/kernel/muskingum"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _sfx(dtype, det):
    ct, sfx = _CT[np.dtype(dtype)]
    if det:
        if sfx != "f32":
            raise ValueError("the bit-reproducible power exists for float32 only")
        sfx = "f32det"
    return ct, sfx


def det_powf(x, y):
    """t-route_amd/csrc/det_pow.h evaluated on the host (elementwise, float32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(np.broadcast_to(np.asarray(y, dtype=np.float32), x.shape))
    out = np.empty_like(x)
    fn = lib().mc_oracle_det_powf
    fn.restype = None
    fn(C.c_long(x.size), _ptr(x, C.c_float), _ptr(y, C.c_float), _ptr(out, C.c_float))
    return out


def segments(inputs, return_iters=False, det=False):
    """Restated kernel over rows of `inputs` [n,15] (float32 or float64).
    det=True: use the bit-reproducible power of det_pow.h instead of libm powf."""
    inputs = np.ascontiguousarray(inputs)
    ct, sfx = _sfx(inputs.dtype, det)
    n = inputs.shape[0]
    out = np.zeros((n, 6), dtype=inputs.dtype)
    iters = np.zeros(n, dtype=np.int32)
    fn = getattr(lib(), f"mc_oracle_segments_{sfx}")
    fn.restype = None
    fn(C.c_long(n), _ptr(inputs, ct), _ptr(out, ct), _ptr(iters, C.c_int))
    return (out, iters) if return_iters else out


def ref_path(name):
    return os.path.join(_HERE, "_ref", name)


def have_ref(name="libmc_ref_qj0_f32.so"):
    return os.path.exists(ref_path(name))


_REFS = {}


def ref_symbol(name, symbol):
    """Address of `symbol` in oracle/_ref/<name> as a void pointer."""
    key = (name, symbol)
    if key not in _REFS:
        h = C.CDLL(ref_path(name))
        _REFS[key] = (h, C.cast(getattr(h, symbol), C.c_void_p))
    return _REFS[key][1]


F32_SYMBOL = "c_muskingcungenwm"
F64_SYMBOL = "_QMmuskingcunge_modulePmuskingcungenwm"


def ref_segments(inputs, name=None):
    """A reference build (c_muskingcungenwm signature) over rows of `inputs`."""
    inputs = np.ascontiguousarray(inputs)
    ct, sfx = _CT[inputs.dtype]
    if name is None:
        name = f"libmc_ref_qj0_{sfx}.so"
    sym = ref_symbol(name, F32_SYMBOL if sfx == "f32" else F64_SYMBOL)
    n = inputs.shape[0]
    out = np.zeros((n, 6), dtype=inputs.dtype)
    fn = getattr(lib(), f"mc_oracle_ref_segments_{sfx}")
    fn.restype = None
    fn(sym, C.c_long(n), _ptr(inputs, ct), _ptr(out, ct))
    return out


def wrf_segments(inputs):
    """Unmodified WRF-Hydro submuskingcunge over rows of `inputs` [n,15] f32 -> [n,3]."""
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    sym = ref_symbol("libmc_wrf_f32.so", "c_submuskingcunge")
    n = inputs.shape[0]
    out = np.zeros((n, 3), dtype=np.float32)
    fn = lib().mc_oracle_wrf_segments_f32
    fn.restype = None
    fn(sym, C.c_long(n), _ptr(inputs, C.c_float), _ptr(out, C.c_float))
    return out


def network(nsteps, qts_subdivisions, reaches, upstreams, params, q0, qlat, assume_short_ts,
            prefilled=None, fvd_init=None, ref_name=None, return_iters=False, det=False, da=None, res=None):
    """Reference network loop (mc_reach.pyx:492-750 restated).

    reaches   : list of int arrays (row positions, upstream->downstream), list order
    upstreams : list of int arrays (row positions summed at the head of each reach)
    params    : [nseg, 9] in PARAM_COLS order;  q0 [nseg,3];  qlat [nseg,nq]
    returns   : fvd [nseg, nsteps+1, 3]   (column 0 of axis 1 = initial state)
    """
    reach_ptr = np.zeros(len(reaches) + 1, dtype=np.int64)
    reach_ptr[1:] = np.cumsum([len(r) for r in reaches])
    reach_seg = (np.concatenate([np.asarray(r, dtype=np.int64) for r in reaches])
                 if reaches else np.zeros(0, np.int64))
    up_ptr = np.zeros(len(upstreams) + 1, dtype=np.int64)
    up_ptr[1:] = np.cumsum([len(u) for u in upstreams])
    up_idx = (np.concatenate([np.asarray(u, dtype=np.int64) for u in upstreams] + [np.zeros(0, np.int64)]))
    return network_arrays(nsteps, qts_subdivisions, reach_ptr, reach_seg, up_ptr, up_idx, params, q0, qlat,
                          assume_short_ts, prefilled, fvd_init, ref_name, return_iters, det, da, res)


class DA(C.Structure):
    """mirror of da_t (float instantiations) in mc_oracle_impl.inc"""
    _fields_ = [("ngage", C.c_long), ("gage_maxtimestep", C.c_long), ("usgs_values", C.POINTER(C.c_float)),
                ("gage_row", C.POINTER(C.c_long)), ("gage_of_reach", C.POINTER(C.c_long)),
                ("decay_coeff", C.c_float), ("routing_period", C.c_float),
                ("lastobs_time", C.POINTER(C.c_float)), ("lastobs_val", C.POINTER(C.c_float)),
                ("nudge", C.POINTER(C.c_float))]


class RES(C.Structure):
    """mirror of res_t (float instantiations) in mc_oracle_impl.inc"""
    _fields_ = [("nres", C.c_long), ("res_of_reach", C.POINTER(C.c_long)), ("par", C.POINTER(C.c_float)),
                ("water_elevation", C.POINTER(C.c_float)), ("routing_period", C.c_float),
                ("inflow_out", C.POINTER(C.c_float))]


LP_PAR = ("area", "max_depth", "orifice_area", "orifice_coefficient", "orifice_elevation", "weir_coefficient",
          "weir_elevation", "weir_length", "dam_length")


def levelpool(inflow, dt, water_elevation, par, lateral=0.0):
    """One routing period of the restated LEVELPOOL_PHYSICS (float32): (outflow, new water elevation)."""
    par = np.ascontiguousarray(par, dtype=np.float32)
    h = C.c_float(float(water_elevation))
    fn = lib().mc_oracle_levelpool_flat_f32
    fn.restype = C.c_float
    q = fn(C.c_float(float(inflow)), C.c_float(float(lateral)), C.c_float(float(dt)), C.byref(h), _ptr(par, C.c_float))
    return np.float32(q), np.float32(h.value)


def simple_da(timestep, routing_period, decay_coeff, gage_maxtimestep, target, model, lastobs_time, lastobs_val):
    """Restated simple_da (simple_da.pyx:22-95), float32: (replacement, nudge, lastobs_time, lastobs_val)."""
    out = (C.c_float * 4)()
    fn = lib().mc_oracle_simple_da_f32
    fn.restype = None
    fn(*(C.c_float(float(v)) for v in (timestep, routing_period, decay_coeff, gage_maxtimestep, target, model,
                                       lastobs_time, lastobs_val)), out)
    return tuple(np.float32(v) for v in out)


def simple_da_with_decay(last_valid_obs, model_val, minutes, decay_coeff):
    fn = lib().mc_oracle_simple_da_with_decay_f32
    fn.restype = C.c_float
    return np.float32(fn(C.c_float(last_valid_obs), C.c_float(model_val), C.c_float(minutes), C.c_float(decay_coeff)))


def network_arrays(nsteps, qts_subdivisions, reach_ptr, reach_seg, up_ptr, up_idx, params, q0, qlat,
                   assume_short_ts, prefilled=None, fvd_init=None, ref_name=None, return_iters=False,
                   det=False, da=None, res=None):
    """As network(), with the reach lists already flattened to CSR arrays (int64).

    res (float32 only): dict(res_of_reach [nreach] (-1 = not a reservoir), par [nres, 9] (LP_PAR order),
    water_elevation [nres], routing_period); on return also 'inflow' [nres, nsteps+1] and the final
    water elevations (mc_reach.pyx:507-716, level-pool branch).

    da (float32 only): dict(usgs_values [ngage, nobs] (NaN = missing), gage_row [ngage], gage_of_reach
    [nreach] (-1 = none), decay_coeff, routing_period, lastobs_time [ngage], lastobs_val [ngage]); on
    return it also holds 'nudge' [ngage, nsteps+1] and the final lastobs arrays (mc_reach.pyx:761-796)."""
    params = np.ascontiguousarray(params)
    dt = params.dtype
    ct, sfx = _sfx(dt, det)
    q0 = np.ascontiguousarray(q0, dtype=dt)
    qlat = np.ascontiguousarray(qlat, dtype=dt)
    nseg = params.shape[0]
    reach_ptr = np.ascontiguousarray(reach_ptr, dtype=np.int64)
    reach_seg = np.ascontiguousarray(reach_seg, dtype=np.int64)
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int64)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int64)
    nreach = reach_ptr.shape[0] - 1
    fvd = np.zeros((nseg, nsteps + 1, 3), dtype=dt) if fvd_init is None else np.ascontiguousarray(fvd_init, dtype=dt)
    pre = None if prefilled is None else np.ascontiguousarray(prefilled, dtype=np.uint8)
    ref = C.c_void_p(0)
    if ref_name is not None:
        ref = ref_symbol(ref_name, F64_SYMBOL if sfx == "f64" else F32_SYMBOL)
    iters = C.c_long(0)
    fn = getattr(lib(), f"mc_oracle_network_{sfx}")
    fn.restype = None
    fn(C.c_long(nseg), C.c_int(nsteps), C.c_int(qts_subdivisions), C.c_long(nreach),
       _ptr(reach_ptr, C.c_long), _ptr(reach_seg, C.c_long), _ptr(up_ptr, C.c_long),
       _ptr(up_idx, C.c_long), _ptr(params, ct), _ptr(q0, ct), _ptr(qlat, ct),
       C.c_long(qlat.shape[1]), C.c_int(int(bool(assume_short_ts))),
       None if pre is None else _ptr(pre, C.c_ubyte), _ptr(fvd, ct), ref, C.byref(iters), _da_struct(da, nsteps, dt),
       _res_struct(res, nsteps, dt))
    return (fvd, iters.value) if return_iters else fvd


def _res_struct(res, nsteps, dtype):
    if res is None:
        return None
    if np.dtype(dtype) != np.float32:
        raise ValueError("reservoirs are restated for float32 only")
    res["res_of_reach"] = np.ascontiguousarray(res["res_of_reach"], dtype=np.int64)
    res["par"] = np.ascontiguousarray(res["par"], dtype=np.float32)
    res["water_elevation"] = np.array(res["water_elevation"], dtype=np.float32, copy=True)
    nres = res["par"].shape[0]
    res["inflow"] = np.zeros((nres, nsteps + 1), dtype=np.float32)
    st = RES(nres, _ptr(res["res_of_reach"], C.c_long), _ptr(res["par"], C.c_float),
             _ptr(res["water_elevation"], C.c_float), float(res["routing_period"]), _ptr(res["inflow"], C.c_float))
    res["_struct"] = st
    return C.byref(st)


def _da_struct(da, nsteps, dtype):
    if da is None:
        return None
    if np.dtype(dtype) != np.float32:
        raise ValueError("nudging is restated for float32 only")
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)  # noqa: E731
    da["usgs_values"] = f32(da["usgs_values"])
    da["gage_row"], da["gage_of_reach"] = i64(da["gage_row"]), i64(da["gage_of_reach"])
    da["lastobs_time"], da["lastobs_val"] = f32(da["lastobs_time"]).copy(), f32(da["lastobs_val"]).copy()
    ngage = da["gage_row"].shape[0]
    da["nudge"] = np.zeros((ngage, nsteps + 1), dtype=np.float32)
    st = DA(ngage, da["usgs_values"].shape[1] if da["usgs_values"].ndim == 2 else 0,
            _ptr(da["usgs_values"], C.c_float), _ptr(da["gage_row"], C.c_long), _ptr(da["gage_of_reach"], C.c_long),
            float(da["decay_coeff"]), float(da["routing_period"]), _ptr(da["lastobs_time"], C.c_float),
            _ptr(da["lastobs_val"], C.c_float), _ptr(da["nudge"], C.c_float))
    da["_struct"] = st
    return C.byref(st)


def network_by_segment(nsteps, qts_subdivisions, up_ptr, up_idx, level, params, q0, qlat, assume_short_ts,
                       routed=None, **kw):
    """Drive the loop with every segment as its own one-segment reach, visited in level order.
    Identical arithmetic to reach-wise execution: inside a reach the reference hands segment i+1
    exactly (q[i,t-1], q[i,t]) (mc_reach.pyx:133-138), which is what a one-element upstream sum gives."""
    order = np.argsort(level, kind="stable").astype(np.int64)
    if routed is not None:                      # rows with a prescribed hydrograph are in no reach
        order = order[np.asarray(routed, dtype=bool)[order]]
    n = order.shape[0]
    reach_ptr = np.arange(n + 1, dtype=np.int64)
    cnt = (up_ptr[1:] - up_ptr[:-1])[order]
    r_up_ptr = np.zeros(n + 1, dtype=np.int64)
    r_up_ptr[1:] = np.cumsum(cnt)
    # gather the upstream lists in visiting order
    starts = up_ptr[:-1][order]
    rep = np.repeat(starts - r_up_ptr[:-1], cnt)
    r_up_idx = up_idx[np.arange(r_up_ptr[-1]) + rep] if r_up_ptr[-1] else np.zeros(0, np.int64)
    return network_arrays(nsteps, qts_subdivisions, reach_ptr, order, r_up_ptr, r_up_idx, params, q0, qlat,
                          assume_short_ts, **kw)


# ---- bench.py's CPU baseline: the reference's ordered sub-network decomposition, C + OpenMP (oracle/cpu_baseline.c) ----
def ordered_subnetworks(to, target=10000):
    """Cut a forest (``to[row]`` = downstream row, -1 at outlets) the way the reference's by-subnetwork methods do
    (build_subnetworks, nhd_network.py:691-771: sub-networks of about ``target`` segments, in orders that are routed one
    after the other, deepest first): a sub-network is a maximal sub-tree of at most ``target`` remaining segments; what
    is left after an order (the rows more than ``target`` segments drain through) is cut again.  Returns
    (order_ptr [norders+1], job_ptr [njobs+1], rows) with the rows of a job in topological order."""
    to = np.asarray(to, dtype=np.int64)
    n = to.shape[0]
    # distance to the outlet, groups from the outlets upward
    idx = np.arange(n, dtype=np.int64)
    anc = np.where(to >= 0, to, idx)
    dist = (to >= 0).astype(np.int64)
    while True:
        nxt = anc[anc]
        dist = dist + dist[anc]
        if np.array_equal(nxt, anc):
            break
        anc = nxt
    by_dist = np.argsort(dist, kind="stable")
    cuts = np.flatnonzero(np.diff(dist[by_dist])) + 1
    groups = np.split(by_dist, cuts)                       # outlets first
    alive = np.ones(n, dtype=bool)
    order_of = np.full(n, -1, dtype=np.int64)
    root_of = np.full(n, -1, dtype=np.int64)
    order = 0
    while alive.any():
        size = alive.astype(np.int64)                      # alive rows draining through each row
        for g in groups[::-1]:
            g = g[alive[g]]
            t = to[g]
            ok = t >= 0
            np.add.at(size, t[ok], size[g[ok]])
        small = alive & (size <= target)
        down_big = np.where(to >= 0, ~small[np.maximum(to, 0)] | ~alive[np.maximum(to, 0)], True)
        is_root = small & down_big
        for g in groups:                                   # roots hand their label upstream
            g = g[small[g]]
            t = np.maximum(to[g], 0)
            root_of[g] = np.where(is_root[g], g, root_of[t])
        order_of[small] = order
        alive &= ~small
        order += 1
    # by order; inside an order the largest jobs first (they bound the makespan of a dynamic schedule); upstream rows first
    job_size = np.bincount(root_of, minlength=n)
    key = np.lexsort((-dist, root_of, -job_size[root_of], order_of))
    rows = key.astype(np.int64)
    jobs = root_of[rows]
    job_start = np.flatnonzero(np.concatenate([[True], jobs[1:] != jobs[:-1]]))
    job_ptr = np.concatenate([job_start, [n]]).astype(np.int64)
    job_order = order_of[rows[job_start]]
    order_ptr = np.concatenate([[0], np.flatnonzero(np.diff(job_order)) + 1, [job_start.shape[0]]]).astype(np.int64)
    return order_ptr, job_ptr, rows


_CPU = None


def cpu_baseline_route(nsteps, qts, short_ts, order_ptr, job_ptr, rows, up_ptr, up_idx, params, qlat, q0,
                       ref_name=None, nthreads=0, checksums=False):
    """One routing window over ordered sub-networks in C + OpenMP (oracle/cpu_baseline.c); the kernel is the reference
    Fortran symbol of oracle/_ref/<ref_name>, or the oracle's restatement when ref_name is None.
    Returns (flows [nseg, nsteps+1], depths [nseg], segment-timesteps routed, threads); with checksums=True the
    position-weighted checksums of every row's velocity and depth series (cpu_baseline.c; `series_checksum` forms the
    same sums from a [row][t] array) are left in cpu_baseline_route.chk_v / .chk_d [nseg]."""
    global _CPU
    if _CPU is None:
        _CPU = C.CDLL(os.path.join(_HERE, "libcpu_baseline.so"))
        _CPU.cpu_baseline_route.restype = C.c_long
    kern = ref_symbol(ref_name, F32_SYMBOL) if ref_name else C.cast(lib().mc_oracle_kernel_f32, C.c_void_p)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)  # noqa: E731
    order_ptr, job_ptr, rows, up_ptr, up_idx = map(i64, (order_ptr, job_ptr, rows, up_ptr, up_idx))
    n = rows.shape[0]
    # The tables are handed over in JOB ORDER (row k of the C side = rows[k]): what a job reads and writes is then one
    # contiguous range of every array -- the reference gives each job its own sliced copies (compute.py:741-760); with
    # the caller's row order the depths of neighbouring rows, written at every step by different threads, would share
    # cache lines (measured on 2 x 64 cores: 256 threads slower than 32).
    inv = np.empty(n, dtype=np.int64)
    inv[rows] = np.arange(n, dtype=np.int64)
    cnt = (up_ptr[1:] - up_ptr[:-1])[rows]
    up_ptr_j = np.zeros(n + 1, dtype=np.int64)
    up_ptr_j[1:] = np.cumsum(cnt)
    rep = np.repeat(up_ptr[:-1][rows] - up_ptr_j[:-1], cnt)
    up_idx_j = inv[up_idx[np.arange(up_ptr_j[-1]) + rep]] if up_ptr_j[-1] else np.zeros(0, np.int64)
    caller_rows, rows = rows, np.arange(n, dtype=np.int64)
    up_ptr, up_idx = up_ptr_j, np.ascontiguousarray(up_idx_j)
    params = np.ascontiguousarray(np.asarray(params, dtype=np.float32)[caller_rows])
    qlat = np.ascontiguousarray(np.asarray(qlat, dtype=np.float32)[caller_rows])
    # flows: one time-major block per job, q[job_ptr[j] * (nsteps + 1) + t * m + k]
    m_of_job = np.diff(job_ptr)
    job_of_row = np.repeat(np.arange(m_of_job.shape[0], dtype=np.int64), m_of_job)
    k_in_job = np.arange(n, dtype=np.int64) - job_ptr[:-1][job_of_row]
    slot0 = job_ptr[:-1][job_of_row] * (nsteps + 1) + k_in_job          # element of (row, t = 0)
    q = np.zeros(n * (nsteps + 1), dtype=np.float32)
    q[slot0] = q0[caller_rows, 0]
    d = np.ascontiguousarray(q0[caller_rows, 2], dtype=np.float32).copy()
    import time as _time
    order_s = np.zeros(order_ptr.shape[0] - 1, dtype=np.float64)
    q += 0                                              # every page touched before the clock starts
    chk_v = np.zeros(n if checksums else 1, dtype=np.float64)
    chk_d = np.zeros(n if checksums else 1, dtype=np.float64)
    t_c = _time.perf_counter()
    done = _CPU.cpu_baseline_route(kern, C.c_int(nsteps), C.c_int(qts), C.c_int(int(bool(short_ts))),
                                   C.c_long(order_ptr.shape[0] - 1), _ptr(order_ptr, C.c_long), _ptr(job_ptr, C.c_long),
                                   _ptr(job_of_row, C.c_long), _ptr(up_ptr, C.c_long), _ptr(up_idx, C.c_long),
                                   _ptr(params, C.c_float), _ptr(qlat, C.c_float), C.c_long(qlat.shape[1]),
                                   _ptr(q, C.c_float), _ptr(d, C.c_float), C.c_int(int(nthreads)),
                                   _ptr(order_s, C.c_double), _ptr(chk_v, C.c_double) if checksums else None,
                                   _ptr(chk_d, C.c_double) if checksums else None)
    cpu_baseline_route.order_seconds = order_s.tolist()
    cpu_baseline_route.last_seconds = _time.perf_counter() - t_c    # the C call alone
    q_j, d_j = q, d                                                  # back to [row][t], the caller's row order
    q = np.empty((n, nsteps + 1), dtype=np.float32)
    step = m_of_job[job_of_row]
    for t in range(nsteps + 1):
        q[caller_rows, t] = q_j[slot0 + t * step]
    d = np.empty_like(d_j)
    d[caller_rows] = d_j
    cpu_baseline_route.chk_v = cpu_baseline_route.chk_d = None
    if checksums:
        cpu_baseline_route.chk_v = np.empty(n, np.float64)
        cpu_baseline_route.chk_d = np.empty(n, np.float64)
        cpu_baseline_route.chk_v[caller_rows] = chk_v
        cpu_baseline_route.chk_d[caller_rows] = chk_d
    return q, d, int(done), int(nthreads) if nthreads else int(_CPU.cpu_baseline_max_threads())


def series_checksum(x, chunk=100000):
    """Position-weighted checksum of every row of a float32 [row][t] array: sum over t = 1..T of (bit pattern as an
    integer) * t, exact in float64 (cpu_baseline.c forms the same sum while it routes).  `x` may be a strided view."""
    n, T = x.shape
    w = np.arange(1, T + 1, dtype=np.float64)
    out = np.empty(n, dtype=np.float64)
    for lo in range(0, n, chunk):
        b = np.ascontiguousarray(x[lo:lo + chunk]).view(np.uint32).astype(np.float64)
        out[lo:lo + chunk] = b @ w
    return out


def reference_windows(to, params, days, q0, nsteps, qts, short_ts=True, nthreads=0, deterministic=True):
    """EVERY row of a network through consecutive routing windows on the CPU (checker for full-size runs): the
    reference Fortran kernel with the canonical `Qj_0 = 0` (oracle/_ref/libmc_ref_qj0_f32.so, built from the reference's
    own sources) when it is there -- else the oracle's restatement, which is pinned to it bit for bit
    (tests/test_oracle_pinning.py) -- driven by oracle/cpu_baseline.c over the reference's own decomposition into ordered
    sub-networks (compute.py:553-1209), window after window with the reference's warm start between them
    (q0 <- (q_T, q_T, depth_T), AbstractNetwork.py:177-191; loop semantics mc_reach.pyx:492-505,:719-750).
    days: forcing arrays [nseg][nq].  Returns a dict for the LAST window: q [nseg][nsteps+1] (column 0 = initial flow),
    d_final [nseg], chk_v / chk_d (series_checksum of the velocity / depth series), state [nseg][3] = the next window's
    q0, kind ("reference" | "port"), seconds."""
    import time as _time
    from troute_amd.synthetic import upstream_csr
    ref_name = "libmc_ref_qj0_f32.so" if (deterministic and have_ref("libmc_ref_qj0_f32.so")) else None
    t0 = _time.perf_counter()
    order_ptr, job_ptr, rows = ordered_subnetworks(to, 10000)
    up_ptr, up_idx = upstream_csr(to)
    state = np.ascontiguousarray(q0, dtype=np.float32)
    q = d = None
    for i, ql in enumerate(days):
        last = i == len(days) - 1
        q, d, _, _ = cpu_baseline_route(nsteps, qts, short_ts, order_ptr, job_ptr, rows, up_ptr, up_idx, params, ql, state,
                                        ref_name=ref_name, nthreads=nthreads, checksums=last)
        state = np.ascontiguousarray(np.stack([q[:, -1], q[:, -1], d], axis=1))
        if not last:
            del q
    return {"q": q, "d_final": d, "chk_v": cpu_baseline_route.chk_v, "chk_d": cpu_baseline_route.chk_d, "state": state,
            "kind": "reference" if ref_name else "port", "seconds": _time.perf_counter() - t0}
