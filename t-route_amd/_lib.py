"""ctypes binding of libtrmc.so (include/trmc.h).

The library is the product: if it is missing or does not load, importing this
module's ``lib()`` raises -- there is no Python or CPU fallback for the
routing arithmetic.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRMC_LIB_PATH: developer hook for A/B timing of experimental builds of the SAME library
LIB_PATH = os.environ.get("TRMC_LIB_PATH") or os.path.join(_HERE, "libtrmc.so")

TRMC_OK, TRMC_EINVAL, TRMC_ECYCLE, TRMC_ENODEVICE, TRMC_EHIP, TRMC_ENOMEM, TRMC_ESTATE = 0, -1, -2, -3, -4, -5, -6
ENGINE_AUTO, ENGINE_LEVELS, ENGINE_FLOW, PLAN_SHORT_TS, PLAN_FULL_TS = 0, 1, 2, 4, 8   # trmc.h plan flags
NPARAM = 9
PARAM_COLS = ("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")  # trmc.h TRMC_P_*


class Stats(C.Structure):
    _fields_ = [
        ("nseg", C.c_int64), ("nseg_routed", C.c_int64), ("nlevels", C.c_int32), ("nsteps", C.c_int32),
        ("assume_short_ts", C.c_int32), ("main_launches", C.c_int32), ("segment_steps", C.c_int64),
        ("ms_prep", C.c_double), ("ms_main", C.c_double), ("ms_emit", C.c_double), ("ms_total", C.c_double),
        ("wide_levels", C.c_int32), ("wide_k", C.c_int32), ("wide_launches", C.c_int32), ("mid_levels", C.c_int32),
        ("wide_segment_steps", C.c_int64), ("ms_wide", C.c_double), ("mid_k", C.c_int32), ("mid_launches", C.c_int32),
        ("arithmetic", C.c_int32), ("reserved0", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PlanOptions(C.Structure):
    """trmc_plan_options (include/trmc.h): a zero-filled struct means the defaults"""
    _fields_ = [
        ("struct_size", C.c_int32), ("arithmetic", C.c_int32), ("wide_min_rows", C.c_int64), ("wide_levels", C.c_int32),
        ("wide_k", C.c_int32), ("mid_min_rows", C.c_int64), ("mid_levels", C.c_int32), ("mid_k", C.c_int32),
        ("tile_perm_group", C.c_int32), ("tail_sort", C.c_int32), ("stem_min_rows", C.c_int32), ("sequence_mode", C.c_int32),
        ("flow_watchdog_ms", C.c_int32), ("flow_overlap", C.c_int32), ("flow_lean", C.c_int32), ("flow_debug", C.c_int32), ("hot_rows", C.c_int32),
        ("cluster_rows", C.c_int32), ("cluster_late_lag", C.c_int32), ("stream_split", C.c_int32), ("hot_wave_rows", C.c_int32), ("velocity_on_demand", C.c_int32), ("reserved", C.c_int32 * 2),
    ]


ARITH_EXACT, ARITH_TOLERANCE = 0, 1
# Environment variables the HOST layer maps onto trmc_plan_options when a plan is created (tests and A/B measurements; the
# library itself reads none of them).  name -> (field, how): "off0" = the variable's 0 means "off" (the field's < 0), any
# other number is the value; "int" = the number as it is; "flag" = 1 when the variable is "1".
OPTION_ENV = {
    "TRMC_WIDE_MIN_ROWS": ("wide_min_rows", "off0"), "TRMC_WIDE_LEVELS": ("wide_levels", "int"), "TRMC_WIDE_K": ("wide_k", "int"),
    "TRMC_MID_MIN_ROWS": ("mid_min_rows", "off0"), "TRMC_MID_LEVELS": ("mid_levels", "int"), "TRMC_MID_K": ("mid_k", "int"),
    "TRMC_TILE_PERM": ("tile_perm_group", "off0"), "TRMC_HOT_ROWS": ("hot_rows", "off0"), "TRMC_CLUSTER_ROWS": ("cluster_rows", "int"), "TRMC_STREAM_SPLIT": ("stream_split", "int"), "TRMC_HOT_WAVE_ROWS": ("hot_wave_rows", "int"), "TRMC_VELOCITY_ON_DEMAND": ("velocity_on_demand", "int"), "TRMC_TAIL_SORT": ("tail_sort", "off0"),
    "TRMC_STEM_MIN_ROWS": ("stem_min_rows", "off0"), "TRMC_SETUP_ASIDE": ("sequence_mode", "flag"),
    "TRMC_FLOW_WATCHDOG_MS": ("flow_watchdog_ms", "int"), "TRMC_FLOW_OVERLAP": ("flow_overlap", "flag"),
    "TRMC_FLOW_LEAN": ("flow_lean", "lean"), "TRMC_FLOW_DEBUG": ("flow_debug", "flag"),
}


def plan_options(options=None):
    """A PlanOptions from the environment (OPTION_ENV, TRMC_ARITHMETIC=exact|tolerance) overridden by `options` (a dict of
    field names; ``arithmetic`` may be "exact" / "tolerance")."""
    o = PlanOptions()
    o.struct_size = C.sizeof(PlanOptions)
    for name, (field, how) in OPTION_ENV.items():
        v = os.environ.get(name)
        if v is None or v == "":
            continue
        n = int(v)
        if how == "off0":
            n = -1 if n == 0 else n
        elif how == "flag":
            n = 1 if n == 1 else 0
        elif how == "lean":
            n = 1 if n == 1 else -1
        setattr(o, field, n)
    a = os.environ.get("TRMC_ARITHMETIC")
    if a:
        o.arithmetic = {"exact": ARITH_EXACT, "tolerance": ARITH_TOLERANCE}[a]
    for k, v in (options or {}).items():
        if k == "arithmetic" and isinstance(v, str):
            v = {"exact": ARITH_EXACT, "tolerance": ARITH_TOLERANCE}[v]
        if not hasattr(o, k) or k in ("struct_size", "reserved"):
            raise ValueError(f"unknown plan option {k!r}")
        setattr(o, k, int(v))
    return o


_vp, _i64, _i32, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_int
_P = C.POINTER
# name -> (restype, argtypes); must list every symbol include/trmc.h declares
SIGNATURES = {
    "trmc_last_error": (C.c_char_p, []),
    "trmc_abi_version": (_int, []),
    "trmc_device_count": (_int, [_P(_int)]),
    "trmc_plan_create": (_int, [_i64, _vp, _vp, _vp, _vp, _int, _int, _P(_vp)]),
    "trmc_plan_create_hinted": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _int, _int, _P(_vp)]),
    "trmc_plan_create_ex": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _P(_vp)]),
    "trmc_plan_create_opt": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _P(PlanOptions), _P(_vp)]),
    "trmc_plan_set_sequence_mode": (_int, [_vp, _int]),
    "trmc_stream_begin": (_int, [_vp, _int, _int, _int, _int, _int]),
    "trmc_stream_push": (_int, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "trmc_stream_gather": (_int, [_vp, _i64, _i32, _vp, _vp]),
    "trmc_stream_boundary": (_int, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "trmc_stream_advance": (_int, [_vp, _int]),
    "trmc_stream_flush": (_int, [_vp]),
    "trmc_stream_wait": (_int, [_vp, _i64]),
    "trmc_stream_info": (_int, [_vp, _P(_i32), _P(_i32), _P(_i32), _P(_i32), _P(_i32), _P(_i64), _P(_i64), _P(_i64)]),
    "trmc_stream_day_ms": (_int, [_vp, _i64, _P(C.c_double)]),
    "trmc_stream_end": (_int, [_vp]),
    "trmc_plan_arithmetic": (_int, [_vp, _P(_i32)]),
    "trmc_plan_engine": (_int, [_vp, _P(_i32)]),
    "trmc_plan_destroy": (None, [_vp]),
    "trmc_topology_levels": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _P(_i32)]),
    "trmc_topology_levels_hinted": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _P(_i32)]),
    "trmc_topology_clusters": (_int, [_i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _P(_i32), _P(_i32), _P(_i32)]),
    "trmc_topology_blocks": (_int, [_i64, _vp, _vp, _vp, _vp, _int, _vp, _vp, _P(_i32), _P(_i32)]),
    "trmc_topology_blocks_general": (_int, [_i64, _vp, _vp, _vp, _i32, _vp, _vp, _P(_i32), _P(_i32), _vp, _i32, _P(_i32)]),
    "trmc_plan_info": (_int, [_vp, _P(_i64), _P(_i64), _P(_i32), _P(_i32), _P(_i32)]),
    "trmc_plan_levels": (_int, [_vp, _vp, _vp]),
    "trmc_plan_lags": (_int, [_vp, _vp, _P(_i32), _P(_i32)]),
    "trmc_upload_forcing": (_int, [_vp, _int, _vp, _i64, _vp, _vp]),
    "trmc_upload_forcing_packed": (_int, [_vp, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trmc_set_boundary_flow_device": (_int, [_vp, _int, _vp]),
    "trmc_set_reservoirs": (_int, [_vp, _i64, _vp, _vp, C.c_double]),
    "trmc_download_reservoir_inflow": (_int, [_vp, _vp]),
    "trmc_set_nudging": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _vp]),
    "trmc_set_nudging_successors": (_int, [_vp, C.c_int64, _vp]),
    "trmc_download_nudge": (_int, [_vp, _vp]),
    "trmc_route_device": (_int, [_vp, _int, _int, _int]),
    "trmc_route_begin": (_int, [_vp, _int, _int, _int]),
    "trmc_route_advance": (_int, [_vp, _int]),
    "trmc_route_end": (_int, [_vp]),
    "trmc_plan_stream": (_int, [_vp, _P(_vp)]),
    "trmc_rowset_create": (_int, [_vp, _vp, _i64, _P(_i32)]),
    "trmc_gather_flow_range": (_int, [_vp, _i32, _int, _int, _vp, _i64]),
    "trmc_set_boundary_flow_range": (_int, [_vp, _int, _int, _vp, _i64, _vp]),
    "trmc_set_boundary_flow_range_indexed": (_int, [_vp, _int, _int, _vp, _i64, _vp, _vp]),
    "trmc_plan_set_lag": (_int, [_vp, _vp]),
    "trmc_download_fvd": (_int, [_vp, _vp]),
    "trmc_download_fvd_strided": (_int, [_vp, _int, _vp]),
    "trmc_download_fvd_rowset": (_int, [_vp, _int, C.c_int32, _vp]),
    "trmc_plan_set_nan_is_zero": (_int, [_vp, _int]),
    "trmc_host_alloc": (_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "trmc_host_free": (_int, [_vp]),
    "trmc_download_final_state": (_int, [_vp, _vp]),
    "trmc_download_iterations": (_int, [_vp, _vp]),
    "trmc_plan_collect_cost": (_int, [_vp, _int]),
    "trmc_download_cost": (_int, [_vp, _vp, _P(_i32)]),
    "trmc_gather_flow_rows": (_int, [_vp, _vp, _i64, _vp, _int]),
    "trmc_download_gathered": (_int, [_vp, _vp]),
    "trmc_plan_set_stamps": (_int, [_vp, _vp, _i32]),
    "trmc_plan_set_output_stride": (_int, [_vp, _i32]),
    "trmc_plan_hot_rows": (_int, [_vp, _P(_i64)]),
    "trmc_fetch_begin": (_int, [_vp, _i32, _vp, _vp]),
    "trmc_fetch_begin_fvd": (_int, [_vp, _i32, _vp, _vp, _int, _vp]),
    "trmc_fetch_wait": (_int, [_vp]),
    "trmc_get_stats": (_int, [_vp, _P(Stats)]),
    "trmc_route": (_int, [_vp, _int, _int, _int, _vp, _i64, _vp, _vp, _vp]),
    "trmc_segments": (_int, [_int, _int, _i64, _vp, _vp]),
    "trmc_segments_ex": (_int, [_int, _int, _int, _i64, _vp, _vp, _vp]),
    "trmc_muskingcungenwm": (None, [_P(C.c_float)] * 21),
    "trmc_plan_chain_from": (_int, [_vp, _vp]),
    "trmc_plan_clone": (_int, [_vp, _P(_vp)]),
    "trmc_stage_forcing": (_int, [_vp, _int, _vp, _i64]),
    "trmc_selfcheck_fast_arith": (_int, [_int, _int, _i64, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    # communicator + device plumbing of the multi-GPU path (csrc/comm.hip)
    "trmc_comm_unique_id": (_int, [_vp]),
    "trmc_comm_init": (_int, [_int, _int, _vp, _int, _P(_vp)]),
    "trmc_comm_init_shm": (_int, [_int, _int, C.c_char_p, _int, _i64, _P(_vp)]),
    "trmc_comm_info": (_int, [_vp, _P(_i32), _P(_i32), _P(_i32)]),
    "trmc_comm_all_gather": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "trmc_comm_all_gather_host": (_int, [_vp, _vp, _vp, _i64]),
    "trmc_comm_barrier": (_int, [_vp]),
    "trmc_comm_destroy": (None, [_vp]),
    "trmc_dev_alloc": (_int, [_int, _i64, _P(_vp)]),
    "trmc_dev_free": (_int, [_int, _vp]),
    "trmc_dev_upload": (_int, [_int, _vp, _vp, _i64]),
    "trmc_dev_copy": (_int, [_int, _vp, _vp, _i64, _vp]),
    "trmc_dev_download": (_int, [_int, _vp, _vp, _i64, _vp]),
    "trmc_dev_download_async": (_int, [_int, _vp, _vp, _i64, _vp]),
    "trmc_dev_gather_rows": (_int, [_int, _vp, _vp, _i64, _i64, _vp, _vp]),
    "trmc_stream_create": (_int, [_int, _P(_vp)]),
    "trmc_stream_create_prio": (_int, [_int, _int, _P(_vp)]),
    "trmc_stream_destroy": (_int, [_int, _vp]),
    "trmc_stream_synchronize": (_int, [_int, _vp]),
    "trmc_device_synchronize": (_int, [_int]),
    "trmc_event_create": (_int, [_int, _P(_vp)]),
    "trmc_event_destroy": (_int, [_int, _vp]),
    "trmc_event_record": (_int, [_int, _vp, _vp]),
    "trmc_stream_wait_event": (_int, [_int, _vp, _vp]),
}

# include/trdw.h (the diffusive-wave mainstem solver, same shared library)
SIGNATURES_DW = {
    "trdw_last_error": (C.c_char_p, []),
    "trdw_select_device": (_int, [_int]),
    "trdw_diffnw": (_int, [_vp] * 42),
    "trdw_diffnw_batch": (_int, [_int, _vp]),
    "trdw_last_timing": (_int, [_P(C.c_double), _P(C.c_double)]),
    "trdw_configure": (_int, [_vp]),
}

_LIB = None
# entry points that never touch the HIP runtime (everything else may initialise it: single_hw_queue_per_priority must know)
_HOST_ONLY = {"trmc_last_error", "trmc_abi_version", "trmc_topology_levels", "trmc_topology_levels_hinted", "trmc_topology_blocks", "trmc_topology_clusters",
              "trmc_topology_blocks_general", "trmc_get_stats", "trmc_plan_info", "trmc_plan_levels", "trmc_plan_lags", "trmc_plan_engine",
              "trmc_plan_arithmetic", "trmc_stream_info", "trdw_last_error", "trdw_last_timing", "trdw_configure", "trmc_comm_info"}


class _Marking:
    """The loaded library; any call that may start the HIP runtime of the process is noted (``_hip_started``) before it is
    made, whoever makes it -- a plan, a page-locked allocation, the single-segment entry point."""

    def __init__(self, cdll):
        self.__dict__["_cdll"] = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name in _HOST_ONLY:
            self.__dict__[name] = fn
            return fn

        def call(*args, _fn=fn):
            global _hip_started
            _hip_started = True
            return _fn(*args)
        self.__dict__[name] = call
        return call


def lib():
    """Load libtrmc.so (once).  Raises if the HIP library is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C t-route_amd/csrc).  There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        lenient = bool(os.environ.get("TRMC_LIB_PATH")) and os.environ.get("TRMC_LIB_LENIENT") == "1"   # A/B against an older build
        for name, (res, args) in list(SIGNATURES.items()) + list(SIGNATURES_DW.items()):
            if lenient and not hasattr(h, name):
                continue
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = _Marking(h)
    return _LIB


def check(rc):
    """Map a trmc_status to the exception the reference would raise."""
    if rc == TRMC_OK:
        return
    msg = lib().trmc_last_error().decode("utf-8", "replace")
    if rc in (TRMC_EINVAL, TRMC_ECYCLE):
        raise ValueError(msg)
    if rc == TRMC_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(f"trmc error {rc}: {msg}")


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- page-locked result arrays ----------------------------------------------------------------------------------
# A result array the size of a CONUS day (9.4 GB) crosses PCIe four times faster into page-locked memory than into
# pageable memory, but locking that many pages costs more than one copy saves -- so the buffers are pooled: the array
# handed out owns its buffer, and when the caller drops it the buffer goes back to the pool for the next window
# (TRMC_PINNED_RESULTS=0 switches the pool off; an allocation that fails falls back to ordinary memory).
_PINNED_MIN = 8 << 20
_PINNED_KEEP = 2          # free buffers kept per size
_PINNED_CAP = int(os.environ.get("TRMC_PINNED_POOL_MB", "24576")) << 20   # free page-locked memory kept in all, bytes
_pinned_free = {}         # nbytes -> [addresses]; guarded by _pinned_lock
_pinned_order = []        # sizes in the order they were last given back (oldest first): what goes when the pool is full
import threading as _threading
import contextlib as _contextlib
# The pool is entered from weakref finalizers too, and a finalizer can run on ANY allocation -- including one made while
# this very thread is inside the pool code (a garbage-collection pass triggered by `setdefault(nbytes, [])`, say).  So the
# lock is re-entrant, and a release that arrives while its thread is already inside the pool is only noted
# (`_pinned_pending`, list.append is atomic) and carried out by the frame that holds the lock, before it leaves.
_pinned_lock = _threading.RLock()
_pinned_pending = []
_pinned_inside = _threading.local()


def _pool_put(address, nbytes, drop):
    """(lock held) give one buffer back: into the pool, or onto `drop` (to be freed outside the lock)"""
    pool = _pinned_free.setdefault(nbytes, [])
    if len(pool) < (_PINNED_KEEP if nbytes <= (4 << 30) else 1):     # (one spare only of the multi-gigabyte buffers)
        pool.append(address)
        if nbytes in _pinned_order:
            _pinned_order.remove(nbytes)
        _pinned_order.append(nbytes)
    else:
        drop.append(address)
    # a process that routes windows of many sizes must not keep page-locked memory of every one of them: beyond
    # the cap the sizes given back longest ago are released first
    total = sum(k * len(v) for k, v in _pinned_free.items())
    while total > _PINNED_CAP and _pinned_order:
        k = _pinned_order[0]
        if _pinned_free.get(k):
            drop.append(_pinned_free[k].pop())
            total -= k
        if not _pinned_free.get(k):
            _pinned_order.pop(0)


@_contextlib.contextmanager
def _pool():
    """the pool's critical section; releases noted meanwhile by finalizers of this thread are carried out before it ends"""
    drop = []
    _pinned_inside.depth = getattr(_pinned_inside, "depth", 0) + 1
    try:
        with _pinned_lock:
            try:
                yield drop
            finally:                         # (also when the body raises: what finalizers noted meanwhile is still carried out)
                while _pinned_pending:
                    a, n = _pinned_pending.pop()
                    _pool_put(a, n, drop)
    finally:
        _pinned_inside.depth -= 1
        if _LIB is not None:
            for a in drop:
                _LIB.trmc_host_free(C.c_void_p(a))


def _pinned_release(address, nbytes):
    try:
        _pinned_live.pop(address, None)
        if getattr(_pinned_inside, "depth", 0) > 0:      # a finalizer run by a collection inside the pool code of this thread
            _pinned_pending.append((address, nbytes))
            return
        with _pool() as drop:
            _pool_put(address, nbytes, drop)
    except Exception:        # interpreter shutdown
        pass


def result_empty(shape, dtype, always_pinned=False):
    """An uninitialised array for a device-to-host copy: page-locked when it is large (or when an asynchronous copy needs
    it to be: always_pinned), ordinary memory otherwise."""
    import weakref
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes == 0 or ((nbytes < _PINNED_MIN and not always_pinned) or os.environ.get("TRMC_PINNED_RESULTS", "1") == "0"):
        return np.empty(shape, dtype=dtype)
    with _pool():
        pool = _pinned_free.get(nbytes)
        address = pool.pop() if pool else None
    if address is None:
        p = C.c_void_p(0)
        if lib().trmc_host_alloc(nbytes, C.byref(p)) != TRMC_OK or not p.value:
            return np.empty(shape, dtype=dtype)
        address = p.value
    buf = (C.c_char * nbytes).from_address(address)
    _pinned_live[address] = nbytes
    weakref.finalize(buf, _pinned_release, address, nbytes)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


_pinned_live = {}         # address -> nbytes of the page-locked buffers handed out and still alive


def is_pinned(array):
    """True if the array's memory lies inside a page-locked buffer this module handed out (``result_empty``): an asynchronous
    copy can then read or write it where it is."""
    if not isinstance(array, np.ndarray) or array.size == 0:
        return False
    lo = array.ctypes.data
    hi = lo + array.nbytes
    return array.flags.c_contiguous and any(a <= lo and hi <= a + n for a, n in list(_pinned_live.items()))


def pinned_pool_clear():
    """Free the page-locked buffers that are not in use."""
    with _pool():
        addrs = [a for pool in _pinned_free.values() for a in pool]
        _pinned_free.clear()
        del _pinned_order[:]
    for a in addrs:
        lib().trmc_host_free(C.c_void_p(a))


_hip_started = False     # a call that initialises the HIP runtime has been made through this module


def single_hw_queue_per_priority(who):
    """GPU_MAX_HW_QUEUES=1 for the HIP runtime of this process, unless the caller has set the variable: the multi-GPU path
    (troute_amd.distributed, troute_amd.comm) wants one hardware queue per stream priority.  The
    runtime reads the variable when it initialises, so this is done when a communicator or a sharded router is BUILT --
    not as a side effect of importing a module -- and if the runtime is already up by then the setting cannot take effect
    any more: said aloud instead of silently changing (or silently not changing) what other HIP users of the process get."""
    if os.environ.get("GPU_MAX_HW_QUEUES"):
        return
    if _hip_started:
        import warnings
        warnings.warn(f"{who}: the HIP runtime of this process was initialised before GPU_MAX_HW_QUEUES could be set to 1; "
                      "export GPU_MAX_HW_QUEUES=1 before the first HIP call for the stream-to-queue mapping the multi-GPU "
                      "path is tuned for (results are unaffected)", RuntimeWarning, stacklevel=3)
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "1"


def mark_hip_started():
    """called by whoever is about to make a call that initialises the HIP runtime"""
    global _hip_started
    _hip_started = True


def device_count():
    mark_hip_started()
    n = C.c_int(0)
    rc = lib().trmc_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def np_dtype(precision):
    if precision == 32:
        return np.float32
    if precision == 64:
        return np.float64
    raise ValueError("precision must be 32 or 64")
