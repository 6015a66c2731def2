"""RoutingPlan -- Python handle of a trmc_plan (include/trmc.h).

A plan owns the flattened topology and the channel parameters in HBM; one plan
serves any number of routing windows (the reference rebuilds its MC_Segment /
MC_Reach objects on every compute_network_structured call,
mc_reach.pyx:283-378).
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def csr_from_lists(lists):
    """[[rows...], ...] -> (ptr int64[n+1], idx int64[nnz])"""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    if lists:
        ptr[1:] = np.cumsum([len(x) for x in lists])
    idx = np.fromiter((v for x in lists for v in x), dtype=np.int64, count=int(ptr[-1]))
    return ptr, idx


def topology_levels(up_ptr, up_idx, boundary=None, cost_hint=None):
    """Host-only level flattening (no GPU).  Returns (level_of_row, plan_pos_of_row, nlevels)."""
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int64)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int64)
    nseg = up_ptr.shape[0] - 1
    b = None if boundary is None else np.ascontiguousarray(boundary, dtype=np.uint8)
    lvl = np.empty(nseg, dtype=np.int32)
    pos = np.empty(nseg, dtype=np.int64)
    nl = C.c_int32(0)
    h = None if cost_hint is None else np.ascontiguousarray(cost_hint, dtype=np.uint8)
    _lib.check(_lib.lib().trmc_topology_levels_hinted(nseg, _lib.ptr(up_ptr), _lib.ptr(up_idx), _lib.ptr(b), _lib.ptr(h),
                                                      _lib.ptr(lvl), _lib.ptr(pos), C.byref(nl)))
    return lvl, pos, nl.value


def topology_blocks(up_ptr, up_idx, boundary=None, cost_hint=None, cost_tiers=True):
    """Host-only block order of the dataflow engine (no GPU).  Returns (plan_pos_of_row, rank_of_row, block_rows, nblocks)."""
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int64)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int64)
    nseg = up_ptr.shape[0] - 1
    b = None if boundary is None else np.ascontiguousarray(boundary, dtype=np.uint8)
    h = None if cost_hint is None else np.ascontiguousarray(cost_hint, dtype=np.uint8)
    pos = np.empty(nseg, dtype=np.int64)
    rank = np.empty(nseg, dtype=np.int32)
    br, nb = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().trmc_topology_blocks(nseg, _lib.ptr(up_ptr), _lib.ptr(up_idx), _lib.ptr(b), _lib.ptr(h),
                                               int(bool(cost_tiers)), _lib.ptr(pos), _lib.ptr(rank), C.byref(br),
                                               C.byref(nb)))
    return pos, rank, br.value, nb.value


def topology_clusters(up_ptr, up_idx, boundary=None, cost_hint=None, wide_min_rows=0, wide_max_levels=16, cluster_rows=128):
    """Host-only cluster order of a short-timestep plan of the level engine (no GPU; csrc/topology.hpp).  Returns
    (plan_pos_of_row, lag_of_row, block_of_row, wide_levels, cluster_levels, cluster_blocks)."""
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int64)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int64)
    nseg = up_ptr.shape[0] - 1
    b = None if boundary is None else np.ascontiguousarray(boundary, dtype=np.uint8)
    h = None if cost_hint is None else np.ascontiguousarray(cost_hint, dtype=np.uint8)
    pos = np.empty(nseg, dtype=np.int64)
    lag = np.empty(nseg, dtype=np.int32)
    blk = np.empty(nseg, dtype=np.int32)
    w, c, nb = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().trmc_topology_clusters(nseg, _lib.ptr(up_ptr), _lib.ptr(up_idx), _lib.ptr(b), _lib.ptr(h),
                                                 int(wide_min_rows), int(wide_max_levels), int(cluster_rows), _lib.ptr(pos),
                                                 _lib.ptr(lag), _lib.ptr(blk), C.byref(w), C.byref(c), C.byref(nb)))
    return pos, lag, blk, w.value, c.value, nb.value


def topology_blocks_general(up_ptr, up_idx, boundary=None, stem_min_rows=1024):
    """Host-only block order of a dataflow plan built for the general mode (no GPU): long stems last in their basin, their
    side tributaries from the top down.  Returns (plan_pos_of_row, rank_of_row, block_rows, nblocks, early_blocks)."""
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int64)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int64)
    nseg = up_ptr.shape[0] - 1
    b = None if boundary is None else np.ascontiguousarray(boundary, dtype=np.uint8)
    pos = np.empty(nseg, dtype=np.int64)
    rank = np.empty(nseg, dtype=np.int32)
    early = np.empty(256, dtype=np.int32)
    br, nb, ne = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().trmc_topology_blocks_general(nseg, _lib.ptr(up_ptr), _lib.ptr(up_idx), _lib.ptr(b), int(stem_min_rows),
                                                       _lib.ptr(pos), _lib.ptr(rank), C.byref(br), C.byref(nb),
                                                       _lib.ptr(early), early.shape[0], C.byref(ne)))
    return pos, rank, br.value, nb.value, early[:min(ne.value, early.shape[0])].copy()


class RoutingPlan:
    def __init__(self, up_ptr, up_idx, params, boundary=None, precision=32, device=0, cost_hint=None,
                 assume_short_ts=None, engine="auto", options=None):
        """
        options       : dict of trmc_plan_options fields (include/trmc.h), e.g. {"arithmetic": "tolerance"} or
                        {"wide_min_rows": 32, "wide_k": 8}; the TRMC_* variables of ``_lib.OPTION_ENV`` (tests, A/B runs)
                        fill what the dict leaves out.  ``engine="auto"`` also honours TRMC_ENGINE=levels|flow.
        assume_short_ts : the timestep mode the plan will be routed with, if known (None: unknown) -- it routes
                        correctly either way, the value picks the row order that is fast for the mode
        engine        : "auto" | "levels" | "flow"  (include/trmc.h, trmc_plan_create_ex)
        up_ptr/up_idx : CSR of upstream rows per row, reference summation order
        params        : float32 [nseg, 9] in _lib.PARAM_COLS order
        boundary      : optional bool/uint8 [nseg], rows with prescribed hydrographs
        cost_hint     : optional uint8 [nseg], e.g. ``download_iterations()`` of a plan of the same network after
                        a window: rows of equal cost are placed together inside their level (results unchanged)
        """
        self._h = C.c_void_p(0)
        up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int64)
        up_idx = np.ascontiguousarray(up_idx, dtype=np.int64)
        params = np.ascontiguousarray(params, dtype=np.float32)
        nseg = up_ptr.shape[0] - 1
        if params.shape != (nseg, _lib.NPARAM):
            raise ValueError("data_values shape mismatch")
        b = None if boundary is None else np.ascontiguousarray(boundary, dtype=np.uint8)
        if b is not None and b.shape != (nseg,):
            raise ValueError("boundary mask shape mismatch")
        self.nseg = nseg
        self.precision = precision
        self.dtype = _lib.np_dtype(precision)
        self.nboundary = 0 if b is None else int(np.count_nonzero(b == 1))   # (2 = a routed row kept out of the leading levels)
        hint = None if cost_hint is None else np.ascontiguousarray(cost_hint, dtype=np.uint8)
        if hint is not None and hint.shape != (nseg,):
            raise ValueError("cost_hint shape mismatch")
        h = C.c_void_p(0)
        if engine == "auto" and precision == 32 and os.environ.get("TRMC_ENGINE"):       # (A/B runs; fp64 plans: level engine)
            engine = os.environ["TRMC_ENGINE"]
            if engine not in ("levels", "flow"):
                raise ValueError("TRMC_ENGINE must be 'flow' or 'levels'")
        flags = {"auto": _lib.ENGINE_AUTO, "levels": _lib.ENGINE_LEVELS, "flow": _lib.ENGINE_FLOW}[engine]
        if assume_short_ts is not None:
            flags |= _lib.PLAN_SHORT_TS if assume_short_ts else _lib.PLAN_FULL_TS
        opt = _lib.plan_options(options)
        _lib.mark_hip_started()
        _lib.check(_lib.lib().trmc_plan_create_opt(nseg, _lib.ptr(up_ptr), _lib.ptr(up_idx), _lib.ptr(params),
                                                   _lib.ptr(b), _lib.ptr(hint), precision, device, flags, C.byref(opt),
                                                   C.byref(h)))
        self._h = h
        self.arithmetic = "tolerance" if opt.arithmetic == _lib.ARITH_TOLERANCE else "exact"
        self.tile_steps = int(opt.wide_k) if opt.wide_k > 0 else 16          # steps per tile launch (trmc_plan_options.wide_k)
        f = C.c_int32(0)
        _lib.check(_lib.lib().trmc_plan_engine(self._h, C.byref(f)))
        self.engine = "flow" if f.value else "levels"
        self._nsteps = None
        self._stream_keep = {}
        self.maxlag = 0

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().trmc_plan_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- facts ------------------------------------------------------------------------
    def info(self):
        nseg, nr = C.c_int64(0), C.c_int64(0)
        nl, pr, dev = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.lib().trmc_plan_info(self._h, C.byref(nseg), C.byref(nr), C.byref(nl), C.byref(pr),
                                             C.byref(dev)))
        return {"nseg": nseg.value, "nseg_routed": nr.value, "nlevels": nl.value, "precision": pr.value,
                "device": dev.value}

    def levels(self):
        lvl = np.empty(self.nseg, dtype=np.int32)
        pos = np.empty(self.nseg, dtype=np.int64)
        _lib.check(_lib.lib().trmc_plan_levels(self._h, _lib.ptr(lvl), _lib.ptr(pos)))
        return lvl, pos

    def lags(self):
        """A plan in cluster order: (tiles every row runs behind the headwaters [-1: boundary row], levels kept as slices,
        cluster levels) -- include/trmc.h, trmc_plan_lags."""
        lag = np.empty(self.nseg, dtype=np.int32)
        w, c = C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.lib().trmc_plan_lags(self._h, _lib.ptr(lag), C.byref(w), C.byref(c)))
        return lag, w.value, c.value

    def stats(self):
        s = _lib.Stats()
        _lib.check(_lib.lib().trmc_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    # -- one routing window -------------------------------------------------------------
    def upload_forcing(self, nsteps, qlat, q0, boundary_fvd=None):
        qlat = np.ascontiguousarray(qlat, dtype=self.dtype)
        if qlat.ndim != 2 or qlat.shape[0] != self.nseg:
            raise ValueError(f"Number of rows in Qlat is incorrect: expected ({self.nseg}), got ({qlat.shape[0]})")
        if q0 is not None:                       # None: continue from the state resident in HBM
            q0 = np.ascontiguousarray(q0, dtype=self.dtype)
            if q0.shape != (self.nseg, 3):
                raise ValueError("initial_conditions shape mismatch")
        bf = None
        if self.nboundary and boundary_fvd is not None:   # None: set_boundary_flow_device() follows
            bf = np.ascontiguousarray(boundary_fvd, dtype=self.dtype)
            if bf.shape != (self.nboundary, nsteps, 3):
                raise ValueError("boundary hydrograph shape mismatch")
        _lib.check(_lib.lib().trmc_upload_forcing(self._h, nsteps, _lib.ptr(qlat), qlat.shape[1], _lib.ptr(q0),
                                                  _lib.ptr(bf)))
        self._nsteps = nsteps

    def upload_forcing_packed(self, nsteps, packed, row_ids, q0, boundary_fvd=None):
        """Forcing straight from packed CHRTOUT columns (``nhd_io.chrtout_packed``): the device decodes, joins
        on feature id and lays the forcing out.  row_ids: int64 [nseg] id of every row of the table."""
        feat = np.asarray(packed["feature_id"], dtype=np.int64)
        row_ids = np.asarray(row_ids, dtype=np.int64)
        if row_ids.shape != (self.nseg,):
            raise ValueError("row_ids must be [nseg]")
        order = np.argsort(feat, kind="stable")
        where = np.searchsorted(feat[order], row_ids)
        where = np.clip(where, 0, max(0, feat.shape[0] - 1))
        hit = feat[order][where] == row_ids if feat.size else np.zeros(self.nseg, dtype=bool)
        feat_of_row = np.where(hit, order[where], -1).astype(np.int64)
        raw_a = np.ascontiguousarray(packed["raw_a"], dtype=np.int32)
        raw_b = None if packed["raw_b"] is None else np.ascontiguousarray(packed["raw_b"], dtype=np.int32)
        pa = np.ascontiguousarray(packed["pack_a"], dtype=np.float64)
        pb = None if packed["pack_b"] is None else np.ascontiguousarray(packed["pack_b"], dtype=np.float64)
        if q0 is not None:
            q0 = np.ascontiguousarray(q0, dtype=self.dtype)
            if q0.shape != (self.nseg, 3):
                raise ValueError("initial_conditions shape mismatch")
        bf = None
        if self.nboundary and boundary_fvd is not None:
            bf = np.ascontiguousarray(boundary_fvd, dtype=self.dtype)
        _lib.check(_lib.lib().trmc_upload_forcing_packed(
            self._h, nsteps, raw_a.shape[0], raw_a.shape[1], _lib.ptr(raw_a), _lib.ptr(raw_b), _lib.ptr(pa),
            _lib.ptr(pb), _lib.ptr(feat_of_row), _lib.ptr(q0), _lib.ptr(bf)))
        self._nsteps = nsteps
        return feat_of_row

    def set_boundary_flow_device(self, nsteps, device_ptr):
        """Boundary rows' flow hydrographs from a device buffer [nboundary][nsteps] (plan precision)."""
        _lib.check(_lib.lib().trmc_set_boundary_flow_device(self._h, nsteps, C.c_void_p(device_ptr)))

    def set_reservoirs(self, res_rows, par, routing_period):
        """Level-pool reservoirs: rows [nres], parameters [nres, 9] (include/trmc.h trmc_set_reservoirs)."""
        res_rows = np.ascontiguousarray(res_rows, dtype=np.int64)
        par = np.ascontiguousarray(par, dtype=self.dtype)
        if par.shape != (res_rows.shape[0], 9):
            raise ValueError("reservoir parameters must be [nres, 9]")
        self._nres = res_rows.shape[0]
        _lib.check(_lib.lib().trmc_set_reservoirs(self._h, self._nres, _lib.ptr(res_rows), _lib.ptr(par),
                                                  float(routing_period)))

    def download_reservoir_inflow(self):
        out = np.zeros((getattr(self, "_nres", 0), self._nsteps), dtype=self.dtype)
        _lib.check(_lib.lib().trmc_download_reservoir_inflow(self._h, _lib.ptr(out)))
        return out

    def set_nudging(self, nsteps, gage_rows, mode, a, w):
        """Nudging tables [ngage, nsteps] for the staged window (see include/trmc.h trmc_set_nudging)."""
        gage_rows = np.ascontiguousarray(gage_rows, dtype=np.int64)
        mode = np.ascontiguousarray(mode, dtype=np.uint8)
        a = np.ascontiguousarray(a, dtype=self.dtype)
        w = np.ascontiguousarray(w, dtype=self.dtype)
        ng = gage_rows.shape[0]
        if mode.shape != (ng, nsteps) or a.shape != (ng, nsteps) or w.shape != (ng, nsteps):
            raise ValueError("nudging tables must be [ngage, nsteps]")
        self._ngage = ng
        _lib.check(_lib.lib().trmc_set_nudging(self._h, nsteps, ng, _lib.ptr(gage_rows), _lib.ptr(mode),
                                               _lib.ptr(a), _lib.ptr(w)))

    def set_nudging_successors(self, successor_rows):
        """Rows directly below gages that sit INSIDE their reach (-1: the gage ends its reach), for a window routed without
        assume_short_ts on a level-engine plan (include/trmc.h trmc_set_nudging_successors)."""
        succ = np.ascontiguousarray(successor_rows, dtype=np.int64)
        _lib.check(_lib.lib().trmc_set_nudging_successors(self._h, succ.shape[0], _lib.ptr(succ)))

    def download_nudge(self):
        out = np.zeros((getattr(self, "_ngage", 0), self._nsteps), dtype=self.dtype)
        _lib.check(_lib.lib().trmc_download_nudge(self._h, _lib.ptr(out)))
        return out

    def route_device(self, nsteps, qts_subdivisions, assume_short_ts):
        _lib.check(_lib.lib().trmc_route_device(self._h, nsteps, qts_subdivisions, int(bool(assume_short_ts))))
        self._nsteps = nsteps
        return self.stats()

    # -- the same window in parts (asynchronous; see include/trmc.h) ------------------------------
    def clone(self):
        """A second set of window buffers on this plan's static data (trmc_plan_clone): same topology, parameters and
        order in HBM, its own forcing / state / result / streams -- for a sequence of windows that takes turns on the two
        (``chain_from``, ``stage_forcing``)."""
        other = object.__new__(RoutingPlan)
        for k in ("nseg", "precision", "dtype", "nboundary", "engine", "maxlag", "arithmetic", "tile_steps"):
            setattr(other, k, getattr(self, k))
        h = C.c_void_p(0)
        other._h = C.c_void_p(0)
        _lib.check(_lib.lib().trmc_plan_clone(self._h, C.byref(h)))
        other._h = h
        other._nsteps = None
        other._stream_keep = {}
        return other

    def set_sequence_mode(self, on=True):
        """The plan is one of several that take turns on the device (``clone``, ``chain_from``): a window's set-up goes to
        the tile stream and its end is queued with its last launch (include/trmc.h, trmc_plan_options.sequence_mode)."""
        _lib.check(_lib.lib().trmc_plan_set_sequence_mode(self._h, int(bool(on))))

    def stage_forcing(self, nsteps, qlat):
        """The next window's forcing on its way to the device without waiting for anything (trmc_stage_forcing): `qlat`
        [nseg, nq] should live in page-locked memory (``_lib.result_empty(..., always_pinned=True)``) for the copy to run
        beside the window another plan is routing.  The array must stay alive and unchanged until that window has begun."""
        if qlat.dtype != self.dtype or not qlat.flags.c_contiguous or qlat.ndim != 2 or qlat.shape[0] != self.nseg:
            raise ValueError(f"qlat must be a C-contiguous {np.dtype(self.dtype).name} array of shape ({self.nseg}, nq)")
        self._staged_qlat = qlat                      # (kept alive while the copy may be in flight)
        _lib.check(_lib.lib().trmc_stage_forcing(self._h, int(nsteps), _lib.ptr(qlat), qlat.shape[1]))
        # (the STAGED window's length: the window in progress, whose products a fetch queued right after this call sizes its
        # arrays for, keeps its own until route_begin starts the staged one)
        self._staged_nsteps = nsteps
        if getattr(self, "_nsteps", None) is None:
            self._nsteps = nsteps

    def chain_from(self, source):
        """This plan's NEXT window starts from the state `source`'s window (queued to its end) leaves, handed over on the
        device in two parts so that this plan's wide levels can start before the source's tail has finished (trmc.h)."""
        _lib.check(_lib.lib().trmc_plan_chain_from(self._h, source._h))

    def route_begin(self, nsteps, qts_subdivisions, assume_short_ts):
        _lib.check(_lib.lib().trmc_route_begin(self._h, nsteps, qts_subdivisions, int(bool(assume_short_ts))))
        self._nsteps = nsteps
        self._staged_nsteps = None

    def route_advance(self, t_end):
        _lib.check(_lib.lib().trmc_route_advance(self._h, int(t_end)))

    def route_end(self):
        _lib.check(_lib.lib().trmc_route_end(self._h))
        return self.stats()

    # -- a stream of windows (include/trmc.h, trmc_stream_*) ------------------------------------------------
    def stream_begin(self, nsteps, qts_subdivisions, slots=0, full_output=False, output_stride=0):
        """After ``upload_forcing`` (the state, the shape of the forcing): days are then ``stream_push``-ed one after the other."""
        _lib.check(_lib.lib().trmc_stream_begin(self._h, int(nsteps), int(qts_subdivisions), int(slots), int(bool(full_output)),
                                                int(output_stride or 0)))
        self._nsteps = nsteps
        self._stream_keep = {}

    def stream_push(self, qlat, boundary_q_ptr=None, rowset=None, hyd=None, q0=None, fvd=None):
        """The next day: ``qlat`` [nseg, nq] (page-locked: ``_lib.result_empty(..., always_pinned=True)``), where its products go
        (page-locked arrays or None), the device pointer of its boundary rows' flows.  Returns the day's number in the stream."""
        if qlat.dtype != self.dtype or not qlat.flags.c_contiguous or qlat.ndim != 2 or qlat.shape[0] != self.nseg:
            raise ValueError(f"qlat must be a C-contiguous {np.dtype(self.dtype).name} array of shape ({self.nseg}, nq)")
        day = self.stream_info()["days_pushed"]
        self._stream_keep[day] = (qlat, hyd, q0, fvd)          # (alive while the copies may be in flight)
        for old in [k for k in self._stream_keep if isinstance(k, int) and k < day - 8]:
            del self._stream_keep[old]
        _lib.check(_lib.lib().trmc_stream_push(self._h, _lib.ptr(qlat), qlat.shape[1], C.c_void_p(boundary_q_ptr or 0),
                                               -1 if rowset is None else int(rowset), _lib.ptr(hyd), _lib.ptr(q0), _lib.ptr(fvd)))
        return day

    def stream_gather(self, day, rowset, device_ptr, stream=0):
        """Flows of a row set over `day` [rows, nsteps] into device memory (trmc_stream_gather)."""
        _lib.check(_lib.lib().trmc_stream_gather(self._h, int(day), int(rowset), C.c_void_p(device_ptr), C.c_void_p(stream or 0)))

    def stream_boundary(self, day, device_ptr, src_row_stride, index_ptr=None, stream=0):
        """The boundary rows' flows of `day` from a block of hydrographs in device memory (trmc_stream_boundary)."""
        _lib.check(_lib.lib().trmc_stream_boundary(self._h, int(day), C.c_void_p(device_ptr), int(src_row_stride),
                                                   C.c_void_p(index_ptr or 0), C.c_void_p(stream or 0)))

    def stream_gather_host(self, day, rowset):
        """The same through the host: [rows, nsteps] (a stream whose ranks exchange with host collectives)."""
        from . import comm as X
        n = self._rowset_n[rowset]
        if n == 0:
            return np.zeros((0, self._nsteps), dtype=self.dtype)
        dev = self.info()["device"]
        buf = X.DeviceBuffer(dev, n * self._nsteps * np.dtype(self.dtype).itemsize)
        self.stream_gather(day, rowset, buf.ptr)
        return buf.download((n, self._nsteps), self.dtype, stream=self.stream())

    def stream_boundary_host(self, day, flows):
        """The boundary rows' flows of `day` [nboundary, nsteps] from a host array."""
        from . import comm as X
        flows = np.ascontiguousarray(flows, dtype=self.dtype)
        if flows.shape != (self.nboundary, self._nsteps):
            raise ValueError("boundary flows must be [nboundary, nsteps]")
        if self.nboundary == 0:
            return
        buf = X.DeviceBuffer.from_array(self.info()["device"], flows)
        self._stream_keep[("boundary", int(day))] = buf          # (alive until the fill has run)
        for k in [k for k in self._stream_keep if isinstance(k, tuple) and k[1] < day - 8]:
            del self._stream_keep[k]
        self.stream_boundary(day, buf.ptr, self._nsteps)

    def stream_advance(self, ntiles):
        """Queue the next ``ntiles`` launches of the stream without a new day (the rows still under way move on)."""
        _lib.check(_lib.lib().trmc_stream_advance(self._h, int(ntiles)))

    def stream_flush(self):
        _lib.check(_lib.lib().trmc_stream_flush(self._h))

    def stream_wait(self, day):
        _lib.check(_lib.lib().trmc_stream_wait(self._h, int(day)))

    def stream_info(self):
        v = [C.c_int32(0) for _ in range(5)] + [C.c_int64(0) for _ in range(3)]
        _lib.check(_lib.lib().trmc_stream_info(self._h, *[C.byref(x) for x in v]))
        names = ("slots", "tiles_per_day", "lag_max", "wide_levels", "cluster_levels", "days_pushed", "days_complete", "launches")
        return {k: x.value for k, x in zip(names, v)}

    def stream_day_ms(self, day):
        """device ms between the first and the last launch of the day's push on the slices' stream (trmc_stream_day_ms)"""
        ms = C.c_double(0)
        _lib.check(_lib.lib().trmc_stream_day_ms(self._h, int(day), C.byref(ms)))
        return ms.value

    def stream_end(self):
        _lib.check(_lib.lib().trmc_stream_end(self._h))
        self._stream_keep = {}

    def stream(self):
        """hipStream_t of the plan as an integer (what troute_amd.comm's event / stream calls and the collectives take)."""
        s = C.c_void_p(0)
        _lib.check(_lib.lib().trmc_plan_stream(self._h, C.byref(s)))
        return s.value or 0

    def rowset(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        rid = C.c_int32(-1)
        _lib.check(_lib.lib().trmc_rowset_create(self._h, _lib.ptr(rows), rows.shape[0], C.byref(rid)))
        if not hasattr(self, "_rowset_n"):
            self._rowset_n = {}
        self._rowset_n[rid.value] = int(rows.shape[0])
        return rid.value

    def gather_flow_range(self, rowset, t_begin, t_end, device_ptr, stride):
        _lib.check(_lib.lib().trmc_gather_flow_range(self._h, rowset, int(t_begin), int(t_end),
                                                     C.c_void_p(device_ptr), int(stride)))

    def set_boundary_flow_range(self, t_begin, t_end, device_ptr, stride, stream=None, index_ptr=None):
        """index_ptr: device int64 [nboundary], the source row of every boundary row (None: row b <- source row b)"""
        _lib.check(_lib.lib().trmc_set_boundary_flow_range_indexed(self._h, int(t_begin), int(t_end),
                                                                   C.c_void_p(device_ptr), int(stride),
                                                                   C.c_void_p(index_ptr) if index_ptr else None,
                                                                   C.c_void_p(stream) if stream else None))

    def set_lag(self, lag_of_row):
        """Rows with lag L are routed L launches behind the others (include/trmc.h trmc_plan_set_lag)."""
        lag = None if lag_of_row is None else np.ascontiguousarray(lag_of_row, dtype=np.int32)
        if lag is not None and lag.shape != (self.nseg,):
            raise ValueError("lag_of_row must be [nseg]")
        _lib.check(_lib.lib().trmc_plan_set_lag(self._h, _lib.ptr(lag)))
        self.maxlag = 0 if lag is None else int(lag.max(initial=0))

    def download_fvd(self, stride=1, rowset=None):
        """fvd [nseg, nsteps // stride, 3]: every `stride`-th step (the steps stride, 2 stride, ... counted from 1), decimated
        on the device (trmc_download_fvd_strided); stride = 1: the whole result.  rowset (``rowset(rows)``): the rows of that
        set only, in its order (trmc_download_fvd_rowset)."""
        stride = int(stride)
        if stride < 1:
            raise ValueError("stride must be >= 1")
        if rowset is not None:
            out = _lib.result_empty((self._rowset_n[rowset], self._nsteps // stride, 3), self.dtype)
            if out.size:
                _lib.check(_lib.lib().trmc_download_fvd_rowset(self._h, stride, int(rowset), _lib.ptr(out)))
            return out
        out = _lib.result_empty((self.nseg, self._nsteps // stride, 3), self.dtype)
        if out.size:
            _lib.check(_lib.lib().trmc_download_fvd_strided(self._h, stride, _lib.ptr(out)))
        return out

    def set_nan_is_zero(self, on=True):
        """uploads of forcing and state: NaN -> 0 on the device (trmc_plan_set_nan_is_zero)"""
        _lib.check(_lib.lib().trmc_plan_set_nan_is_zero(self._h, int(bool(on))))

    def download_final_state(self):
        out = _lib.result_empty((self.nseg, 3), self.dtype)
        _lib.check(_lib.lib().trmc_download_final_state(self._h, _lib.ptr(out)))
        return out

    def download_iterations(self):
        """uint8 [nseg]: secant iterations each row spent on the last routed timestep (diagnostic)."""
        out = np.zeros(self.nseg, dtype=np.uint8)
        _lib.check(_lib.lib().trmc_download_iterations(self._h, _lib.ptr(out)))
        return out

    def collect_cost(self, enable=True):
        """Sum min(secant iterations, 3) per row over the timesteps of the windows routed from now on (the cost
        hint of ``RoutingPlan(cost_hint=...)``); off by default."""
        _lib.check(_lib.lib().trmc_plan_collect_cost(self._h, int(bool(enable))))

    def download_cost(self):
        """(uint16 [nseg] cost sums of the last window, its number of timesteps)"""
        out = np.zeros(self.nseg, dtype=np.uint16)
        n = C.c_int32(0)
        _lib.check(_lib.lib().trmc_download_cost(self._h, _lib.ptr(out), C.byref(n)))
        return out, n.value

    def gather_flow_rows_resident(self, rows):
        """Gather into plan-owned HBM (no host copy); download_gathered() fetches it later."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        self._gathered_shape = (rows.shape[0], self._nsteps)
        _lib.check(_lib.lib().trmc_gather_flow_rows(self._h, _lib.ptr(rows), rows.shape[0], None, 1))

    def download_gathered(self):
        out = _lib.result_empty(self._gathered_shape, self.dtype)
        if out.size:
            _lib.check(_lib.lib().trmc_download_gathered(self._h, _lib.ptr(out)))
        return out

    def hot_rows(self):
        """Diagnosis (trmc_plan_hot_rows): rows the tiles have routed from the hot list so far, all windows together."""
        n = C.c_int64(0)
        _lib.check(_lib.lib().trmc_plan_hot_rows(self._h, C.byref(n)))
        return n.value

    def set_output_stride(self, stride):
        """Windows begun from now on write every `stride`-th step of (q, v, d) aside as they go (trmc_plan_set_output_stride):
        what ``fetch_begin(..., output_stride=stride)`` then copies without another pass over the result; 0 / None: off."""
        _lib.check(_lib.lib().trmc_plan_set_output_stride(self._h, int(stride or 0)))

    def set_stamps(self, nwindows=8):
        """Diagnosis (trmc_plan_set_stamps): returns a page-locked uint64 array [nwindows, 4] that window k since this call
        fills (row k % nwindows) with the device clock (100 MHz) at: tiles begin, last tile ended, tail begins, last step
        launch ended.  nwindows = 0 switches it off."""
        if not nwindows:
            _lib.check(_lib.lib().trmc_plan_set_stamps(self._h, None, 0))
            self._stamps = None
            return None
        ring = _lib.result_empty((int(nwindows), 4), np.uint64, always_pinned=True)
        ring[...] = 0
        _lib.check(_lib.lib().trmc_plan_set_stamps(self._h, _lib.ptr(ring), int(nwindows)))
        self._stamps = ring
        return ring

    def fetch_begin(self, rowset, want_state=True, output_stride=None):
        """Start the asynchronous fetch of a window's products (include/trmc.h trmc_fetch_begin): the hydrographs of a
        registered row set and / or the final state, into page-locked arrays; returns at once.  ``fetch_wait()`` hands the
        arrays over when the copy stream is through -- typically after the NEXT window has been queued.  The arrays come
        from a ring of three sets the plan keeps (all made at the first call: no allocation in a steady pipeline), so what
        ``fetch_wait()`` returned stays valid until the third fetch_begin() after it.

        output_stride = n: also every n-th step of (q, v, d) of every row, [nseg, nsteps // n, 3], decimated on the device
        and copied beside the next window (trmc_fetch_begin_fvd); ``fetch_wait()`` then returns three arrays."""
        nrows = 0 if rowset is None else self._rowset_n[rowset]
        stride = None if output_stride is None else int(output_stride)
        if stride is not None and stride < 1:
            raise ValueError("output_stride must be >= 1")
        shape = ((nrows, self._nsteps) if rowset is not None else None, (self.nseg, 3) if want_state else None,
                 None if stride is None else (self.nseg, self._nsteps // stride, 3))
        ring = getattr(self, "_fetch_ring", None)
        if ring is None or ring["shape"] != shape:
            ring = {"shape": shape, "k": 0,
                    "sets": [tuple(None if sh is None else _lib.result_empty(sh, self.dtype, always_pinned=True) for sh in shape)
                             for _ in range(3)]}
            self._fetch_ring = ring
        hyd, q0, fvd = ring["sets"][ring["k"] % 3]
        ring["k"] += 1
        rs = -1 if rowset is None else int(rowset)
        if stride is None:
            _lib.check(_lib.lib().trmc_fetch_begin(self._h, rs, _lib.ptr(hyd), _lib.ptr(q0)))
            self._fetch = (hyd, q0)
        else:
            _lib.check(_lib.lib().trmc_fetch_begin_fvd(self._h, rs, _lib.ptr(hyd), _lib.ptr(q0), stride,
                                                       _lib.ptr(fvd) if fvd.size else None))
            self._fetch = (hyd, q0, fvd)

    def fetch_wait(self):
        _lib.check(_lib.lib().trmc_fetch_wait(self._h))
        out, self._fetch = getattr(self, "_fetch", (None, None)), (None, None)
        return out

    def gather_flow_rows(self, rows, device_ptr=None):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        if device_ptr is not None:
            _lib.check(_lib.lib().trmc_gather_flow_rows(self._h, _lib.ptr(rows), rows.shape[0],
                                                        C.c_void_p(device_ptr), 1))
            return None
        out = np.empty((rows.shape[0], self._nsteps), dtype=self.dtype)
        _lib.check(_lib.lib().trmc_gather_flow_rows(self._h, _lib.ptr(rows), rows.shape[0], _lib.ptr(out), 0))
        return out

    def route(self, nsteps, qts_subdivisions, assume_short_ts, qlat, q0, boundary_fvd=None):
        """upload + route + download: fvd [nseg, nsteps, 3]."""
        self.upload_forcing(nsteps, qlat, q0, boundary_fvd)
        self.route_device(nsteps, qts_subdivisions, assume_short_ts)
        return self.download_fvd()


def segments(inputs, device=0, arithmetic="exact", with_iterations=False):
    """Batch of independent single-segment steps on the GPU.  inputs [n,15] -> [n,6] (and, with_iterations, the secant
    iterations of every step, int32 [n]).  arithmetic: "exact" | "tolerance" (include/trmc.h, trmc_plan_options)."""
    inputs = np.ascontiguousarray(inputs)
    if inputs.dtype == np.float32:
        precision = 32
    elif inputs.dtype == np.float64:
        precision = 64
    else:
        raise ValueError("inputs must be float32 or float64")
    if inputs.ndim != 2 or inputs.shape[1] != 15:
        raise ValueError("inputs must be [n, 15]")
    out = np.empty((inputs.shape[0], 6), dtype=inputs.dtype)
    iters = np.zeros(inputs.shape[0], dtype=np.int32) if with_iterations else None
    _lib.mark_hip_started()
    _lib.check(_lib.lib().trmc_segments_ex(device, precision, {"exact": _lib.ARITH_EXACT, "tolerance": _lib.ARITH_TOLERANCE}[arithmetic],
                                           inputs.shape[0], _lib.ptr(inputs), _lib.ptr(out), _lib.ptr(iters)))
    return (out, iters) if with_iterations else out
