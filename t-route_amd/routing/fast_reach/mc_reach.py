"""Drop-in for ``troute.routing.fast_reach.mc_reach`` (Muskingum-Cunge branch).

``compute_network_structured`` has the positional signature and the 10-tuple
return of the reference's Cython function
(src/troute-routing/troute/routing/fast_reach/mc_reach.pyx:164-224, return
:811-845) and can be registered in the reference's ``_compute_func_map``
(src/troute-routing/troute/routing/compute.py:21-26) -- see INTEGRATION.md.

Implemented: the MC branch, streamflow nudging at gages (simple_da) and level-pool
reservoir reaches (reach_type 1 with reservoir type 1).  Hybrid-persistence, RFC and
Great Lakes reservoir data assimilation raise NotImplementedError.  All arithmetic
runs in libtrmc.so on the GPU; there is no Python fallback.
"""
import numpy as np

from ... import _lib
from ...plan import RoutingPlan

# mc_reach.pyx:150-162: the kernel's column order inside data_values
_KERNEL_COLS = ("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")


def binary_find(arr, els):
    """Positions of ``els`` in sorted ``arr`` (mc_reach.pyx:36-66); ValueError if absent."""
    arr = np.asarray(arr)
    els = np.asarray(list(els) if not isinstance(els, np.ndarray) else els, dtype=arr.dtype)
    if els.size == 0:
        return []
    if arr.shape[0] == 0:
        raise ValueError(f"element {els[0]} not found in {arr}")
    idx = np.searchsorted(arr, els)
    idx_c = np.minimum(idx, arr.shape[0] - 1)
    bad = arr[idx_c] != els
    if bad.any():
        raise ValueError(f"element {els[bad][0]} not found in {arr}")
    return idx_c.tolist()


class RetunePolicy:
    """When the row order of a tuned plan is worth taking again.

    The order groups rows by the secant-iteration cost they showed in ONE window; it serves for as long as the rows keep
    their classes -- day after day of an ordinary sequence -- and is worth nothing once half of them have changed (then a
    window costs what it costs on the untuned plan, a quarter more).  The cache therefore watches the
    device time of every window on the tuned plan: ``patience`` consecutive windows slower than ``slow_ratio`` x the fastest
    one seen on this order ask for a re-tune -- the next window collects costs again, the call after it rebuilds the plan
    from them (1.6 s on the host for CONUS).  A re-tune that does not bring the time back (a forcing whose classes do not
    persist from window to window cannot be tuned for) makes the new, slower time the yardstick, so the next request needs
    another ``slow_ratio`` on top of it, and doubles the number of windows to sit out before one may be made at all.
    Pure bookkeeping: results never depend on the order.  ``TRMC_RETUNE=0`` switches it off."""

    def __init__(self, slow_ratio=1.12, patience=2, first_hold=8, max_hold=1024, min_gap_ms=0.5):
        self.slow_ratio, self.patience, self.first_hold, self.max_hold = slow_ratio, patience, first_hold, max_hold
        self.min_gap_ms = min_gap_ms  # ... and slower by at least this much (the windows of a small network are all launch latency)
        self.best = None          # fastest window on the current order, ms
        self.slow = 0             # consecutive slow windows
        self.hold = 0             # windows to sit out before the next request
        self.next_hold = first_hold
        self.before = None        # the window time that made the last request
        self.retunes = 0

    def window(self, ms):
        """``ms``: device time of the window just routed on the tuned order.  True: collect costs in the next window."""
        if not (ms > 0):
            return False
        if self.best is None or ms < self.best:
            self.best = ms
        if self.hold > 0:
            self.hold -= 1
            self.slow = 0
            return False
        self.slow = self.slow + 1 if ms > self.slow_ratio * self.best and ms - self.best >= self.min_gap_ms else 0
        if self.slow < self.patience:
            return False
        self.slow, self.before = 0, ms
        return True

    def rebuilt(self, ms):
        """``ms``: the first window on the order a re-tune produced."""
        self.retunes += 1
        if self.before is not None and ms > 0.95 * self.before:    # it did not help
            self.hold, self.next_hold = self.next_hold, min(2 * self.next_hold, self.max_hold)
        else:
            self.next_hold = self.first_hold
        self.best, self.before = (ms if ms > 0 else None), None


class _PlanCache:
    """Plans of the networks this process has routed, by content (``nwm_route`` calls the kernel callable once per loop
    with the same network: topology, order and device copies of the parameters are built once instead of per call).
    The first window of a short-timestep fp32 network is routed on a plan built from the topology alone while its
    secant-iteration costs are collected; the second call rebuilds the plan with those costs as the row-order hint
    (results do not depend on the order), and later calls reuse it -- until its windows have become slower than they were
    (RetunePolicy: the forcing has moved on), when costs are collected and the plan is rebuilt again.
    ``TRMC_PLAN_CACHE=<n>`` sets the number of plans kept (default 2; 0: a fresh plan per call, as the reference's callable
    is stateless)."""

    def __init__(self):
        import collections
        import threading
        self._d = collections.OrderedDict()
        self._lock = threading.RLock()

    def _key(self, up_ptr, up_idx, params, boundary, precision, device, short_ts, res_rows, engine):
        try:                                   # (18 x faster than blake2b on the 120 MB of a CONUS table)
            import xxhash
            h = xxhash.xxh3_128()
        except ImportError:
            import hashlib
            h = hashlib.blake2b(digest_size=16)
        for a in (up_ptr, up_idx, params, boundary):
            if a is None:
                h.update(b"-")
            else:
                a = np.ascontiguousarray(a)
                h.update(str((a.shape, a.dtype.str)).encode())
                h.update(a.view(np.uint8).reshape(-1).data)
        h.update(repr((precision, device, short_ts, res_rows, engine)).encode())
        return h.digest()

    def lease(self, up_ptr, up_idx, params, boundary, precision, device, short_ts, res_rows, engine="auto", token=None):
        """token: None, or a dict the caller keeps with the arrays (compute_network_structured: the entry of its _FlatCache for
        one set of caller objects -- the same entry means the same up_ptr, up_idx, params): the content key is computed once,
        kept in it, and found again on the next call instead of hashing 120 MB of tables per call"""
        import contextlib
        import os
        keep = int(os.environ.get("TRMC_PLAN_CACHE", "2"))

        def build(hint=None):
            return RoutingPlan(up_ptr, up_idx, params, boundary, precision, device, cost_hint=hint, assume_short_ts=short_ts,
                               engine=engine)

        @contextlib.contextmanager
        def fresh():
            with build() as plan:
                yield plan

        if keep <= 0:
            return fresh()

        @contextlib.contextmanager
        def cached():
            with self._lock:
                if token is not None:
                    bkey = None if boundary is None else np.packbits(np.asarray(boundary, dtype=bool)).tobytes()
                    small = (bkey, precision, device, short_ts, res_rows, engine)
                    key = token.get(small)
                    if key is None:
                        key = token[small] = self._key(up_ptr, up_idx, params, boundary, precision, device, short_ts, res_rows, engine)
                else:
                    key = self._key(up_ptr, up_idx, params, boundary, precision, device, short_ts, res_rows, engine)
                e = self._d.pop(key, None)
                tune = short_ts and precision == 32
                retune = tune and os.environ.get("TRMC_RETUNE", "1") != "0"
                if e is None:
                    e = {"plan": build(), "stage": 0 if tune else 2, "policy": RetunePolicy() if retune else None, "fresh": 0}
                    if tune:
                        e["plan"].collect_cost(True)
                elif e["stage"] == 1:                          # costs of the window before -> the tuned plan
                    cost, n = e["plan"].download_cost()
                    hint = np.minimum((cost.astype(np.int64) * 16 + n - 1) // max(n, 1), 255).astype(np.uint8)
                    e["plan"].close()
                    e = {"plan": build(hint), "stage": 2, "policy": e.get("policy"), "fresh": e.get("fresh", 0)}
                ok = False
                try:
                    yield e["plan"]
                    ok = True
                finally:
                    if ok:
                        if e["stage"] == 0:
                            e["stage"] = 1
                        elif e["stage"] == 2 and e.get("policy") is not None:
                            try:
                                ms = float(e["plan"].stats()["ms_main"])
                            except Exception:
                                ms = 0.0
                            pol = e["policy"]
                            if e.get("fresh"):                 # a re-tuned order is judged by its SECOND window (the first
                                e["fresh"] -= 1                # of a new plan also pays for its buffers' first use)
                                if e["fresh"] == 0:
                                    pol.rebuilt(ms)
                            elif pol.window(ms):               # slower than this order used to be: take the costs again
                                e["plan"].collect_cost(True)
                                e["stage"], e["fresh"] = 0, 2
                        self._d[key] = e
                        while len(self._d) > keep:
                            self._d.popitem(last=False)[1]["plan"].close()
                    else:
                        e["plan"].close()
        return cached()

    def clear(self):
        with self._lock:
            while self._d:
                self._d.popitem()[1]["plan"].close()


_PLANS = _PlanCache()


class _FlatCache:
    """The flattened network of the objects a caller keeps handing in.  ``nwm_route`` calls the kernel callable once per run
    set with the SAME reach list, upstream dictionary and id table (nwm_routing/__main__.py:1215-1257); at CONUS size the walk
    over 2.1 million reach lists, the look-up of 2.7 million ids and the hash of the result for the plan cache are seconds
    around a window of milliseconds.  So the last few results are kept by the IDENTITY of those objects (with their lengths and
    end points as a guard against an object that was refilled in place) -- the objects themselves are kept alive by the entry,
    so an id cannot be reused while it is in here.  ``TRMC_FLAT_CACHE=0`` switches it off."""

    def __init__(self, keep=3):
        import collections
        self._d, self._keep = collections.OrderedDict(), keep

    @staticmethod
    def _guard(reaches_wTypes, upstream_connections, data_idx):
        n = len(reaches_wTypes)
        ends = (tuple(reaches_wTypes[0][0]), reaches_wTypes[0][1], tuple(reaches_wTypes[-1][0]), reaches_wTypes[-1][1]) if n else ()
        return (n, len(upstream_connections), data_idx.shape[0], ends,
                int(data_idx[0]) if data_idx.size else 0, int(data_idx[-1]) if data_idx.size else 0)

    def get(self, reaches_wTypes, upstream_connections, data_idx):
        import os
        if os.environ.get("TRMC_FLAT_CACHE", "1") == "0" or not isinstance(reaches_wTypes, list):
            return None, None
        key = (id(reaches_wTypes), id(upstream_connections), id(data_idx))
        e = self._d.get(key)
        if e is not None and e["guard"] == self._guard(reaches_wTypes, upstream_connections, data_idx):
            self._d.move_to_end(key)
            return key, e
        return key, None

    def put(self, key, reaches_wTypes, upstream_connections, data_idx, **content):
        if key is None:
            return None
        e = dict(content, guard=self._guard(reaches_wTypes, upstream_connections, data_idx),
                 refs=(reaches_wTypes, upstream_connections, data_idx), serial=_FlatCache._serial)
        _FlatCache._serial += 1
        self._d[key] = e
        while len(self._d) > self._keep:
            self._d.popitem(last=False)
        return e

    def clear(self):
        self._d.clear()


_FlatCache._serial = 0
_FLAT = _FlatCache()


def column_mapper(src_cols):
    """Map source columns to the kernel's column order (mc_reach.pyx:150-162)."""
    index = {label: i for i, label in enumerate(src_cols)}
    return [index[label] for label in _KERNEL_COLS]


def _flatten_network(reaches_wTypes, upstream_connections, data_idx):
    """Reach lists + upstream dict -> per-row upstream CSR in the reference's summation order.

    Head of a reach: rows of upstream_connections[reach[0]] in dict-list order
    (mc_reach.pyx:288-289, :499-502).  Inside a reach: the previous segment
    (mc_reach.pyx:133-138).  Rows that belong to no reach get no upstreams and
    are reported in ``in_reach``.
    """
    nseg = data_idx.shape[0]
    heads, head_ups = [], []
    flat = []
    starts = []
    for reach, reach_type in reaches_wTypes:
        if reach_type == 1 and len(reach) != 1:
            raise ValueError("a reservoir reach must be the single waterbody node")   # mc_reach.pyx:293
        starts.append(len(flat))
        flat.extend(reach)
        heads.append(reach[0])
        head_ups.append(upstream_connections.get(reach[0], ()))
    flat = np.asarray(flat, dtype=np.int64)
    rows = np.asarray(binary_find(data_idx, flat), dtype=np.int64)
    in_reach = np.zeros(nseg, dtype=bool)
    in_reach[rows] = True
    if rows.shape[0] != np.count_nonzero(in_reach):
        raise ValueError("a segment appears in more than one reach")

    counts = np.zeros(nseg, dtype=np.int64)
    is_head = np.zeros(flat.shape[0], dtype=bool)
    is_head[np.asarray(starts, dtype=np.int64)] = True
    counts[rows[~is_head]] = 1
    head_rows = rows[is_head]
    # the upstream ids of all reach heads in one look-up (one searchsorted instead of one per reach)
    n_up = np.fromiter((len(u) for u in head_ups), dtype=np.int64, count=len(head_ups))
    all_ups = np.fromiter((x for u in head_ups for x in u), dtype=np.int64, count=int(n_up.sum()))
    all_up_rows = np.asarray(binary_find(data_idx, all_ups), dtype=np.int64)
    counts[head_rows] = n_up
    up_ptr = np.zeros(nseg + 1, dtype=np.int64)
    up_ptr[1:] = np.cumsum(counts)
    up_idx = np.empty(up_ptr[-1], dtype=np.int64)
    # inside a reach: previous element of the flattened reach list
    prev = np.empty_like(rows)
    prev[1:] = rows[:-1]
    up_idx[up_ptr[rows[~is_head]]] = prev[~is_head]
    # heads: their upstream rows in dict-list order, written at up_ptr[head] + 0, 1, ...
    if all_up_rows.size:
        first = np.zeros(n_up.shape[0] + 1, dtype=np.int64)
        first[1:] = np.cumsum(n_up)
        within = np.arange(all_up_rows.size, dtype=np.int64) - np.repeat(first[:-1], n_up)
        up_idx[np.repeat(up_ptr[head_rows], n_up) + within] = all_up_rows
    return up_ptr, up_idx, in_reach


def compute_network_structured(
    nsteps,
    dt,
    qts_subdivisions,
    reaches_wTypes,
    upstream_connections,
    data_idx,
    data_cols,
    data_values,
    initial_conditions,
    qlat_values,
    lake_numbers_col,
    wbody_cols,
    data_assimilation_parameters,
    reservoir_types,
    reservoir_type_specified,
    model_start_time,
    usgs_values,
    usgs_positions,
    usgs_positions_reach,
    usgs_positions_gage,
    lastobs_values_init,
    time_since_lastobs_init,
    da_decay_coefficient,
    reservoir_usgs_obs,
    reservoir_usgs_wbody_idx,
    reservoir_usgs_time,
    reservoir_usgs_update_time,
    reservoir_usgs_prev_persisted_flow,
    reservoir_usgs_persistence_update_time,
    reservoir_usgs_persistence_index,
    reservoir_usace_obs,
    reservoir_usace_wbody_idx,
    reservoir_usace_time,
    reservoir_usace_update_time,
    reservoir_usace_prev_persisted_flow,
    reservoir_usace_persistence_update_time,
    reservoir_usace_persistence_index,
    reservoir_rfc_obs,
    reservoir_rfc_wbody_idx,
    reservoir_rfc_totalCounts,
    reservoir_rfc_file,
    reservoir_rfc_use_forecast,
    reservoir_rfc_timeseries_idx,
    reservoir_rfc_update_time,
    reservoir_rfc_da_timestep,
    reservoir_rfc_persist_days,
    great_lakes_idx,
    great_lakes_times,
    great_lakes_discharge,
    great_lakes_param_idx,
    great_lakes_param_prev_assim_flow,
    great_lakes_param_prev_assim_times,
    great_lakes_param_update_times,
    great_lakes_climatology,
    upstream_results={},
    assume_short_ts=False,
    return_courant=False,
    da_check_gage=-1,
    from_files=True,
    *,
    precision=32,
    device=0,
    return_stats=False,
    output_stride=None,
    result_order=None,
    nan_is_zero=False,
):
    """Route one (sub)network for ``nsteps`` timesteps on the GPU.

    Arguments and return value: see the reference docstring,
    mc_reach.pyx:225-240, and SURVEY.md 8(b).  Keyword-only extensions:
    ``precision`` (32 = the reference's arithmetic type, 64 = double),
    ``device`` (HIP ordinal), ``return_stats`` (append the trmc_stats dict),
    ``output_stride`` (None or 1: the reference's result; n > 1: element [1] holds
    every n-th step only -- ``flowveldepth[:, 3 * (n (k + 1) - 1) : ...]``, the steps
    the reference's writers keep when n = qts_subdivisions, nwm_routing/output.py:209-216,
    :232-240 -- decimated on the device, so that a twelfth of the bytes crosses the host
    link; bit-identical to slicing the full result; the other elements are unchanged),
    ``result_order`` (None: rows in the order of ``data_idx``, the reference's; a
    permutation of ``range(len(data_idx))``: elements [0], [1] and [6] come back in that
    row order, permuted on the device -- compute_nhd_routing_v02 asks for its rows grouped
    by tailwater, so that every network's block is a slice; without off-network
    ``upstream_results`` rows only), ``nan_is_zero`` (True: NaN in ``qlat_values`` and
    ``initial_conditions`` -- the waterbody rows of the reference's reindexed tables,
    compute.py:1466-1467, which no Muskingum-Cunge row reads -- become 0 on the device
    behind the upload instead of in a pass over the arrays on the host).
    """
    stride = 1 if output_stride is None else int(output_stride)
    if stride < 1:
        raise ValueError("output_stride must be a positive number of timesteps")
    if int(nsteps) % stride != 0:
        # (with a remainder the last kept step is not the window's last one: new_q0 -- AbstractNetwork.py:182-190 takes the last
        # column -- would restart the next window from an earlier step without a word)
        raise ValueError(f"output_stride ({stride}) must divide nsteps ({nsteps}): the last kept step has to be the window's last")
    data_idx = np.ascontiguousarray(data_idx, dtype=np.int64)
    data_values = np.asarray(data_values)
    qlat_values = np.asarray(qlat_values)
    initial_conditions = np.asarray(initial_conditions)
    nseg = data_idx.shape[0]

    # preconditions of the reference, same messages (mc_reach.pyx:243-250)
    if qlat_values.shape[0] != nseg:
        raise ValueError(
            f"Number of rows in Qlat is incorrect: expected ({nseg}), got ({qlat_values.shape[0]})")
    if qlat_values.shape[1] < nsteps / qts_subdivisions:
        raise ValueError(
            f"Number of columns (timesteps) in Qlat is incorrect: expected at most ({nseg}), "
            f"got ({qlat_values.shape[1]}). The number of columns in Qlat must be equal to or less than "
            "the number of routing timesteps")
    if data_values.shape[0] != nseg or data_values.shape[1] != len(data_cols):
        raise ValueError("data_values shape mismatch")

    usgs_positions = np.asarray(usgs_positions, dtype=np.int64)
    gages_size = usgs_positions.shape[0]

    fkey, fe = _FLAT.get(reaches_wTypes, upstream_connections, data_idx)
    if fe is None:
        up_ptr, up_idx, in_reach = _flatten_network(reaches_wTypes, upstream_connections, data_idx)
        fe = _FLAT.put(fkey, reaches_wTypes, upstream_connections, data_idx, up_ptr=up_ptr, up_idx=up_idx, in_reach=in_reach,
                       any_res=any(rt == 1 for _, rt in reaches_wTypes), params_of=None, params=None)
        any_res = fe["any_res"] if fe is not None else any(rt == 1 for _, rt in reaches_wTypes)
    else:
        up_ptr, up_idx, in_reach, any_res = fe["up_ptr"], fe["up_idx"], fe["in_reach"], fe["any_res"]
    # (the parameter table in kernel column order: kept with the flattened network while the caller hands in the same array)
    plan_token = None
    sample = None
    if fe is not None:      # (a table refilled in place is another table: a strided sample of its rows guards the identity)
        sample = np.asarray(data_values[::max(1, nseg // 257)], dtype=np.float64).sum(axis=0).tobytes() if nseg else b""
    if fe is not None and fe["params_of"] is data_values and fe.get("params_sample") == sample:
        params = fe["params"]
        plan_token = fe["plan_keys"]
    else:
        params = np.ascontiguousarray(np.asarray(data_values, dtype=np.float32)[:, column_mapper(list(data_cols))])
        if fe is not None:
            fe["params_of"], fe["params"], fe["params_sample"] = data_values, params, sample
            plan_token = fe["plan_keys"] = {}     # (content keys of the plans of these tables: filled by the plan cache, found again next call)

    # ---- level-pool reservoirs (mc_reach.pyx:283-356): one-node reaches of type 1 ----------------------
    res_rows, res_par, res_q0 = [], [], []
    if any_res:
        wb = np.asarray(wbody_cols, dtype=np.float64)
        rtypes = np.asarray(reservoir_types)
        for reach, rt in reaches_wTypes:
            if rt != 1:
                continue
            my_id = binary_find(data_idx, reach)[0]
            wbody_index = binary_find(lake_numbers_col, reach)[0]
            if reservoir_type_specified and rtypes.size and int(rtypes[wbody_index][0]) != 1:
                raise NotImplementedError(
                    f"waterbody {reach[0]}: reservoir type {int(rtypes[wbody_index][0])} (hybrid persistence / RFC "
                    "forecast / Great Lakes data assimilation) is outside this engine; level pool (type 1) only")
            a = wb[wbody_index].astype(np.float32)     # levelpool.pyx:63-73: area max_depth orifice_area
            #   orifice_coefficient orifice_elevation weir_coefficient weir_elevation weir_length ifd (qd0) h0
            h0 = a[10]
            if h0 < np.float32(-900000000):              # cold start (levelpool_structs.c init_levelpool_reach)
                h0 = np.float32(a[4] + np.float32(np.float32(a[1] - a[4]) * a[8]))
            res_rows.append(my_id)
            res_par.append([a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], np.float32(10.0)])  # dam_length = 10
            res_q0.append((np.float32(wb[wbody_index, 9]), h0))

    # rows that carry a prescribed hydrograph: upstream_results (mc_reach.pyx:451-469);
    # rows in no reach at all stay zero in the reference -- prescribed zero hydrograph
    fill_index_mask = np.ones(nseg, dtype=bool)
    boundary = ~in_reach
    dtype = _lib.np_dtype(precision)
    bvals = {}
    q0 = np.array(initial_conditions, dtype=dtype, copy=True)
    q0[boundary] = 0
    for r, (qd0, h0) in zip(res_rows, res_q0):           # initial outflow (:297) and water elevation of the pool
        q0[r] = (qd0, 0, h0)
    for _tw, tmp in upstream_results.items():
        fill_index = int(tmp["position_index"])
        fill_index_mask[fill_index] = False
        boundary[fill_index] = True
        res = np.asarray(tmp["results"], dtype=dtype).reshape(-1, 3)
        if res.shape[0] != nsteps:
            raise ValueError("upstream_results hydrograph length does not match nsteps")
        bvals[fill_index] = res
        if len(lake_numbers_col) and int(data_idx[fill_index]) in set(int(x) for x in lake_numbers_col):
            # an off-network waterbody: its initial flow is the reservoir's qd0 column, the depth slot stays zero
            # (mc_reach.pyx:463-465; initial_conditions holds NaN for lake ids in the reference's by-subnetwork driver)
            q0[fill_index, 0] = np.asarray(wbody_cols, dtype=np.float64)[
                binary_find(lake_numbers_col, [int(data_idx[fill_index])])[0], 9]
        else:
            q0[fill_index, 0] = initial_conditions[fill_index, 0]        # mc_reach.pyx:467-468
            q0[fill_index, 2] = initial_conditions[fill_index, 2]
    # ---- streamflow nudging (mc_reach.pyx:380-411): tables resolved on the host, applied on the GPU ----
    nudging = None
    if gages_size:
        from . import simple_da as _da
        usgs_values = np.asarray(usgs_values, dtype=np.float32)
        upr = np.asarray(usgs_positions_reach, dtype=np.int64)
        upg = np.asarray(usgs_positions_gage, dtype=np.int64)
        # reach_has_gage[reach] = gage index; the nudge is applied to usgs_positions[that gage] after the
        # whole reach has been routed, so the engine needs the gage segment to END its reach (which the
        # reference's gage-splitting network builders guarantee, nhd_network.py:319-338)
        reach_gage = {}
        for gi in range(gages_size):
            reach_gage[int(upr[gi])] = int(upg[gi])
        active = sorted(set(reach_gage.values()))
        # With assume_short_ts a segment only ever reads STORED flows of the step before (mc_reach.pyx:133-137: qup = qdp,
        # quc = qup), so it does not matter where in its reach the gage sits.  Without it the segment below a gage inside
        # a reach reads the gage segment's un-nudged flow of the current step (the nudge is applied after the whole reach,
        # :761-796) while everything else sees the nudged one -- two values for one segment-step, which the engine does
        # not carry in one array
        # -- a second value per gage and step, which the level engine carries (trmc_set_nudging_successors).
        successors = None
        for ri, gi in ({} if assume_short_ts else reach_gage).items():
            reach_rows = binary_find(data_idx, reaches_wTypes[ri][0])
            if reach_rows[-1] != int(usgs_positions[gi]):
                if int(usgs_positions[gi]) not in reach_rows:
                    raise ValueError("a gage that is not a segment of the reach it is listed for")
                if successors is None:
                    successors = np.full(gages_size, -1, dtype=np.int64)
                successors[gi] = reach_rows[reach_rows.index(int(usgs_positions[gi])) + 1]
        mode, a_tab, w_tab, lt_fin, lv_fin = _da.resolve_tables(
            nsteps, dt, da_decay_coefficient, usgs_values, lastobs_values_init, time_since_lastobs_init)
        act = np.asarray(active, dtype=np.int64)
        if precision != 32:
            a_tab, w_tab = a_tab.astype(dtype), w_tab.astype(dtype)
        nudging = (usgs_positions[act], mode[act], a_tab[act], w_tab[act], act,
                   None if successors is None else successors[act])
        # gages that own no reach are never visited by the loop: their lastobs stay at the initial values
        idle = np.setdiff1d(np.arange(gages_size), act)
        lt_fin[idle] = np.asarray(time_since_lastobs_init, dtype=np.float32)[idle]
        lv_fin[idle] = np.asarray(lastobs_values_init, dtype=np.float32)[idle]
        if usgs_values.ndim == 2 and usgs_values.shape[1] > 0:      # initial flow <- first observation (:404-411)
            v0 = usgs_values[:, 0]
            ok = ~np.isnan(v0)
            q0[usgs_positions[ok], 0] = v0[ok]
    brow = np.flatnonzero(boundary)
    boundary_fvd = None
    if brow.size:
        boundary_fvd = np.zeros((brow.size, nsteps, 3), dtype=dtype)
        for k, r in enumerate(brow.tolist()):
            if r in bvals:
                boundary_fvd[k] = bvals[r]

    nudge = np.zeros((gages_size, nsteps + 1), dtype="float32")
    # (the timestep mode is known here: the plan picks the engine and the row order that suit it, trmc_plan_create_ex)
    # (gages inside a reach in the general mode need the level engine: it carries the un-nudged flow beside the nudged one)
    engine = "levels" if nudging is not None and nudging[5] is not None else "auto"
    with _PLANS.lease(up_ptr, up_idx, params, boundary if brow.size else None, precision, device,
                      bool(assume_short_ts), tuple(res_rows), engine, token=plan_token) as plan:
        if res_rows:
            plan.set_reservoirs(res_rows, np.asarray(res_par, dtype=dtype), dt)
        plan.set_nan_is_zero(bool(nan_is_zero))
        plan.upload_forcing(nsteps, qlat_values, q0, boundary_fvd)
        if nudging is not None:
            plan.set_nudging(nsteps, nudging[0], nudging[1], nudging[2], nudging[3])
            if nudging[5] is not None:
                plan.set_nudging_successors(nudging[5])
        plan.route_device(nsteps, qts_subdivisions, assume_short_ts)
        oset = None
        if result_order is not None:
            if not fill_index_mask.all():
                raise ValueError("result_order: not with off-network upstream_results rows (they are masked out of the result)")
            result_order = np.ascontiguousarray(result_order, dtype=np.int64)
            if result_order.shape != (nseg,):
                raise ValueError(f"result_order must be a permutation of the {nseg} rows")
            sets = plan.__dict__.setdefault("_order_sets", {})      # (the set lives with the plan: one per caller's order array)
            okey = (result_order.ctypes.data, nseg)
            hit = sets.get(okey)
            if hit is None or hit[0] is not result_order:
                if len(sets) >= 2:
                    sets.clear()
                if nseg and not np.array_equal(np.sort(result_order), np.arange(nseg)):
                    raise ValueError(f"result_order must be a permutation of the {nseg} rows")
                hit = sets[okey] = (result_order, plan.rowset(result_order))
            oset = hit[1]
        fvd = plan.download_fvd(stride, rowset=oset)
        if nudging is not None:
            nudge[nudging[4], 1:] = plan.download_nudge()
        res_inflow = plan.download_reservoir_inflow() if res_rows else None
        stats = plan.stats()

    out_dtype = np.float32 if precision == 32 else np.float64
    flowveldepth = fvd.reshape(nseg, (nsteps // stride) * 3).astype(out_dtype, copy=False)
    if not fill_index_mask.all():          # (no copy of the result when no off-network upstream rows were spliced in)
        flowveldepth = flowveldepth[fill_index_mask]
    upstream = np.zeros((nseg, nsteps), dtype="float32")  # np.empty in the reference (:487), reservoir rows filled (:710)
    if res_rows:
        upstream[res_rows] = res_inflow
        if result_order is not None:
            upstream = upstream[result_order]
    if not fill_index_mask.all():
        upstream = upstream[fill_index_mask]
    t_end = nsteps * dt
    f32 = lambda a: np.asarray(a, dtype="float32")  # noqa: E731
    i32 = lambda a: np.asarray(a, dtype="int32")  # noqa: E731
    result = (
        np.asarray(data_idx, dtype=np.intp)[fill_index_mask] if result_order is None else np.asarray(data_idx, dtype=np.intp)[result_order],
        flowveldepth,
        0,
        (
            np.asarray([data_idx[p] for p in usgs_positions]),
            (lt_fin if gages_size else np.full(0, np.nan, dtype="float32")),
            (lv_fin if gages_size else np.full(0, np.nan, dtype="float32")),
        ),
        (
            i32(reservoir_usgs_wbody_idx),
            f32(reservoir_usgs_update_time) - t_end,
            f32(reservoir_usgs_prev_persisted_flow),
            f32(reservoir_usgs_persistence_index),
            f32(reservoir_usgs_persistence_update_time) - t_end,
        ),
        (
            i32(reservoir_usace_wbody_idx),
            f32(reservoir_usace_update_time) - t_end,
            f32(reservoir_usace_prev_persisted_flow),
            f32(reservoir_usace_persistence_index),
            f32(reservoir_usace_persistence_update_time) - t_end,
        ),
        upstream,
        (
            i32(reservoir_rfc_wbody_idx),
            f32(reservoir_rfc_update_time) - t_end,
            i32(reservoir_rfc_timeseries_idx),
        ),
        nudge,
        (
            i32(great_lakes_param_idx),
            f32(great_lakes_param_prev_assim_flow),
            i32(great_lakes_param_prev_assim_times),
            i32(great_lakes_param_update_times),
        ),
    )
    if return_stats:
        return result + (stats,)
    return result


def mc_only_args(nsteps, dt, qts_subdivisions, reaches, upstream_connections, data_idx, data_cols,
                 data_values, initial_conditions, qlat_values, upstream_results=None,
                 assume_short_ts=False):
    """Positional argument list for an MC-only call: every DA / reservoir array empty, exactly the
    shapes compute_nhd_routing_v02 passes when those features are off (compute.py:1513-1576)."""
    e_f2 = np.zeros((0, 0), dtype="float32")
    e_f1 = np.zeros(0, dtype="float32")
    e_i1 = np.zeros(0, dtype="int32")
    reaches_wTypes = [(list(r), 0) for r in reaches]
    return [
        nsteps, dt, qts_subdivisions, reaches_wTypes, upstream_connections,
        np.asarray(data_idx, dtype="int64"), np.asarray(data_cols, dtype=object),
        np.asarray(data_values, dtype="float32"), np.asarray(initial_conditions, dtype="float32"),
        np.asarray(qlat_values, dtype="float32"),
        [], np.zeros((0, 0), dtype="float64"), {}, np.zeros((0, 0), dtype="int32"), False,
        "2021-08-23_13:00:00",
        e_f2, e_i1, e_i1, e_i1, e_f1, e_f1, 0.0,
        e_f2, e_i1, e_f1, e_f1, e_f1, e_f1, e_f1,
        e_f2, e_i1, e_f1, e_f1, e_f1, e_f1, e_f1,
        e_f2, e_i1, e_i1, [], e_i1, e_i1, e_f1, e_i1, e_i1,
        e_i1, e_i1, e_f1, e_i1, e_f1, e_i1, e_i1, e_f2,
        upstream_results or {}, assume_short_ts,
    ]
