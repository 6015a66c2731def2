"""Drop-in for ``troute.routing.fast_reach.diffusive`` (the Cython wrapper of the Fortran diffusive-wave solver,
src/troute-routing/troute/routing/fast_reach/diffusive.pyx:129-262): ``compute_diffusive(diff_inputs)`` takes the
dictionary ``diffusive_input_data_v02`` builds (diffusive_utils_v02.py:659-1155) and returns
``(out_q, out_elv, out_depth)``, each ``[ntss_ev_g, mxncomp_g, nrch_g]`` float64, computed on the GPU by
``trdw_diffnw`` (include/trdw.h; c_diffnw's argument list).  No CPU fallback."""
import ctypes as C

import numpy as np

from ... import _lib

ARG_ORDER = ("timestep_ar_g", "nts_ql_g", "nts_ub_g", "nts_db_g", "ntss_ev_g", "nts_qtrib_g", "nts_da_g", "mxncomp_g",
             "nrch_g", "z_ar_g", "bo_ar_g", "traps_ar_g", "tw_ar_g", "twcc_ar_g", "mann_ar_g", "manncc_ar_g", "so_ar_g",
             "dx_ar_g", "iniq", "frnw_col", "frnw_g", "qlat_g", "ubcd_g", "dbcd_g", "qtrib_g", "paradim", "para_ar_g",
             "mxnbathy_g", "x_bathy_g", "z_bathy_g", "mann_bathy_g", "size_bathy_g", "usgs_da_g", "usgs_da_reach_g",
             "rdx_ar_g", "cwnrow_g", "cwncol_g", "crosswalk_g", "z_thalweg_g")
_INT_SCALARS = {"nts_ql_g", "nts_ub_g", "nts_db_g", "ntss_ev_g", "nts_qtrib_g", "nts_da_g", "mxncomp_g", "nrch_g",
                "frnw_col", "paradim", "mxnbathy_g", "cwnrow_g", "cwncol_g"}
_INT_ARRAYS = {"frnw_g", "size_bathy_g", "usgs_da_reach_g"}

TRDW_EINVAL, TRDW_EUNSUPPORTED, TRDW_ENOMEM = -1, -2, -5


def _marshal(diff_inputs):
    """(keep-alive list, the 42 argument pointers in c_diffnw's order, the three Fortran-ordered output arrays)"""
    keep, args = [], []
    for k in ARG_ORDER:
        v = diff_inputs[k]
        if k in _INT_SCALARS:
            c = C.c_int(int(v))
            keep.append(c)
            args.append(C.cast(C.pointer(c), C.c_void_p))
        else:
            a = np.asfortranarray(v, dtype=np.int32 if k in _INT_ARRAYS else np.float64)
            if a.size == 0:
                a = np.zeros(1, dtype=a.dtype)
            keep.append(a)
            args.append(a.ctypes.data_as(C.c_void_p))
    shape = (int(diff_inputs["ntss_ev_g"]), int(diff_inputs["mxncomp_g"]), int(diff_inputs["nrch_g"]))
    outs = [np.zeros(shape, dtype=np.float64, order="F") for _ in range(3)]
    args += [o.ctypes.data_as(C.c_void_p) for o in outs]
    return keep, args, outs


def _raise(lib, rc):
    if rc != 0:
        msg = lib.trdw_last_error().decode("utf-8", "replace")
        if rc == TRDW_EUNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == TRDW_EINVAL:
            raise ValueError(msg)
        if rc == TRDW_ENOMEM:
            raise MemoryError(msg)
        raise RuntimeError(f"trdw error {rc}: {msg}")


class _Options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("solver", C.c_int32), ("chain_global", C.c_int32), ("window_rows", C.c_int32),
                ("phase_ticks", C.c_int32)]


def _configure(lib):
    """trdw_configure from the test / measurement variables of the host layer (the library reads no environment):
    TRDW_SOLVER=serial, TRDW_CHAIN_GLOBAL, TRDW_WINDOW_ROWS=<n>, TRDW_PHASES"""
    import os
    o = _Options(C.sizeof(_Options), 1 if os.environ.get("TRDW_SOLVER") == "serial" else 0,
                 1 if os.environ.get("TRDW_CHAIN_GLOBAL") else 0, int(os.environ.get("TRDW_WINDOW_ROWS") or 0),
                 1 if os.environ.get("TRDW_PHASES") else 0)
    _raise(lib, lib.trdw_configure(C.byref(o)))


def compute_diffusive(diff_inputs, device=0):
    lib = _lib.lib()
    keep, args, outs = _marshal(diff_inputs)
    _configure(lib)
    rc = lib.trdw_select_device(int(device))
    if rc == 0:
        rc = lib.trdw_diffnw(*args)
    _raise(lib, rc)
    # the reference hands back C-ordered copies (diffusive.pyx:124-126)
    return tuple(np.ascontiguousarray(o) for o in outs)


def compute_diffusive_batch(diff_inputs_list, device=0):
    """Several tailwater domains in one launch (trdw_diffnw_batch): one (out_q, out_elv, out_depth) per domain.
    The reference calls compute_diffusive once per tailwater, one after the other (compute.py:1762-1850)."""
    lib = _lib.lib()
    n = len(diff_inputs_list)
    if n == 0:
        return []
    arr = ((C.c_void_p * 42) * n)()
    keep_all, outs_all = [], []
    for b, ins in enumerate(diff_inputs_list):
        keep, args, outs = _marshal(ins)
        keep_all.append(keep)
        outs_all.append(outs)
        for k, a in enumerate(args):
            arr[b][k] = a.value if isinstance(a, C.c_void_p) else C.cast(a, C.c_void_p).value
    _configure(lib)
    rc = lib.trdw_select_device(int(device))
    if rc == 0:
        rc = lib.trdw_diffnw_batch(n, C.cast(arr, C.c_void_p))
    _raise(lib, rc)
    return [tuple(np.ascontiguousarray(o) for o in outs) for outs in outs_all]


def last_timing():
    """(tables_ms, solve_ms) of this thread's last compute_diffusive call (HIP events)."""
    a, b = C.c_double(0), C.c_double(0)
    _lib.lib().trdw_last_timing(C.byref(a), C.byref(b))
    return a.value, b.value
