"""Drop-in for the Muskingum-Cunge part of ``troute.routing.compute``.

``compute_nhd_routing_v02`` keeps the reference's signature and return value
(src/troute-routing/troute/routing/compute.py:507-546; results = a list of the
kernel callable's tuples, one per independent network, consumed by
``AbstractNetwork.new_q0`` (AbstractNetwork.py:177-191) and the output writers
(output.py:209-216)).

What is different, on purpose (MI355X-first, SURVEY.md 8b "Threading"):
  * the reference fans the independent networks out to joblib/loky worker
    processes and, for the big basins, to ordered sub-networks that hand
    tailwater hydrographs from one order to the next
    (compute.py:553-1209, :1211-1395).  Here ALL networks of the call are
    flattened into ONE device plan and routed by one sequence of launches: on a
    GPU the levels of every network run side by side, so neither processes nor
    sub-network orders are needed.  Every ``parallel_compute_method`` the
    reference accepts is therefore accepted and produces the same per-network
    result list (``cpu_pool`` and ``subnetwork_target_size`` are ignored).
  * the per-call argument preparation of the reference (pandas ``.loc`` slicing
    per network, compute.py:1399-1467) is done once for the whole table.
Level-pool waterbodies (``waterbodies_df``, reservoir type 1) are routed and gage observations
(``usgs_df`` / ``lastobs_df``, streamflow nudging) are prepared as the reference prepares them
(``_prep_da_dataframes``, ``_prep_da_positions_byreach``); reservoir data-assimilation DataFrames raise
NotImplementedError, as does the diffusive branch.
"""
from collections import defaultdict

import numpy as np

from .fast_reach.mc_reach import _KERNEL_COLS, compute_network_structured  # noqa: F401

_compute_func_map = defaultdict(
    lambda: compute_network_structured,
    {
        "V02-structured": compute_network_structured,   # the reference's only key (compute.py:21-26)
        "V02-hip": compute_network_structured,
    },
)

_PARALLEL_METHODS = ("serial", "by-network", "by-subnetwork", "by-subnetwork-jit",
                     "by-subnetwork-jit-clustered", "bmi")


def _is_empty(df):
    return df is None or getattr(df, "empty", True) or len(df) == 0


def _prep_da_dataframes(usgs_df, lastobs_df, param_df_sub_idx, exclude_segments=None):
    """Gage observation series and last-valid observations of the gages inside one segment table.

    Same name, arguments and return as the reference helper (compute.py:49-123):
    ``(usgs_df_sub, lastobs_df_sub, da_positions_list_byseg)`` for the four presence cases
    (both: analysis & assimilation, gage order = lastobs order; lastobs only: forecast, an observation
    table with no columns; usgs only: cold start, an all-NaN lastobs table; neither: open loop)."""
    import pandas as pd
    table = param_df_sub_idx
    inside = table.difference(set(exclude_segments)) if exclude_segments else table
    have_u, have_l = not _is_empty(usgs_df), not _is_empty(lastobs_df)
    if not have_u and not have_l:
        return pd.DataFrame(), pd.DataFrame(), []
    if have_l:
        gages = [g for g in lastobs_df.index if g in inside]            # lastobs order decides
        lastobs_sub = lastobs_df.loc[gages]
        usgs_sub = usgs_df.loc[gages] if have_u else pd.DataFrame(index=lastobs_sub.index, columns=[])
    else:
        gages = [g for g in usgs_df.index if g in inside]
        usgs_sub = usgs_df.loc[gages]
        lastobs_sub = pd.DataFrame(index=usgs_sub.index, columns=["discharge", "time", "model_discharge"])
    return usgs_sub, lastobs_sub, table.get_indexer(gages)


def _prep_da_positions_byreach(reach_list, gage_index):
    """(reach numbers that hold a gage, the gage number of each) in reach-list order -- compute.py:125-140."""
    where = {g: i for i, g in enumerate(gage_index)}
    reach_key, gage_i = [], []
    for i, reach in enumerate(reach_list):
        for s in reach:
            if s in where:
                reach_key.append(i)
                gage_i.append(where[s])
    return reach_key, np.asarray(gage_i, dtype=np.intp)


def compute_nhd_routing_v02(
    connections,
    rconn,
    wbody_conn,
    reaches_bytw,
    compute_func_name,
    parallel_compute_method,
    subnetwork_target_size,
    cpu_pool,
    t0,
    dt,
    nts,
    qts_subdivisions,
    independent_networks,
    param_df,
    q0,
    qlats,
    usgs_df,
    lastobs_df,
    reservoir_usgs_df,
    reservoir_usgs_param_df,
    reservoir_usace_df,
    reservoir_usace_param_df,
    reservoir_rfc_df,
    reservoir_rfc_param_df,
    great_lakes_df,
    great_lakes_param_df,
    great_lakes_climatology_df,
    da_parameter_dict,
    assume_short_ts,
    return_courant,
    waterbodies_df,
    data_assimilation_parameters,
    waterbody_types_df,
    waterbody_type_specified,
    subnetwork_list,
    flowveldepth_interorder={},
    from_files=True,
    *,
    precision=32,
    device=0,
    output_stride=None,
):
    """Route every independent network of the call for ``nts`` timesteps.

    Returns what the reference returns (compute.py:1738): ``(results, subnetwork_list)``.  ``results`` has one
    10-tuple per tailwater of ``reaches_bytw`` (in its iteration order), each shaped like
    ``compute_network_structured``'s return (mc_reach.pyx:811-845): ``(ids, flowveldepth[n, nts*3], 0, (..), (..),
    (..), upstream[n, nts], (..), nudge, (..))``; ``subnetwork_list`` is handed back unchanged (the reference fills
    it with its sub-network decomposition for the by-subnetwork methods and passes it through for the others; here
    every network of the call is ONE plan whatever the method, so there is nothing to cache -- ``nwm_route`` only
    stores it and passes it back in, nwm_routing/__main__.py:1256-1257).

    ``flowveldepth_interorder`` (the reference consumes the caller's dictionary in its "bmi" method, compute.py:1588-
    1589,:1649-1655,:1729-1732): ``{segment id: {"results": flowveldepth row [nts*3]}}`` -- segments routed elsewhere
    whose hydrographs enter this call's networks.  They join the table as rows with a prescribed hydrograph
    (``upstream_results`` of the kernel callable) and are left out of the results, as in the reference.

    ``output_stride`` (keyword-only; None = the reference's result): n > 1 keeps every n-th timestep of ``flowveldepth``
    only -- the steps n, 2n, ..., which are the ones the reference's writers take when n = ``qts_subdivisions``
    (nwm_routing/output.py:209-216, :232-240) -- decimated on the device (``compute_network_structured``); with ``nts`` a
    multiple of n the last step is among them, so ``new_q0`` works on the result as it is.
    """
    if parallel_compute_method not in _PARALLEL_METHODS and parallel_compute_method is not None:
        raise ValueError(f"unknown parallel_compute_method {parallel_compute_method!r}")
    for name, df in (("reservoir_usgs_df", reservoir_usgs_df), ("reservoir_usace_df", reservoir_usace_df),
                     ("reservoir_rfc_df", reservoir_rfc_df), ("great_lakes_df", great_lakes_df)):
        if not _is_empty(df):
            raise NotImplementedError(
                f"{name} is not empty: reservoir data assimilation (hybrid persistence, RFC forecasts, Great "
                "Lakes) is outside the Muskingum-Cunge path this package replaces")
    offnetwork_upstreams = sorted(int(k) for k in flowveldepth_interorder) if flowveldepth_interorder else []

    # ---- what depends on the NETWORK only is made once per set of caller objects (nwm_route hands the same dictionaries and
    # the same parameter table to every run set, nwm_routing/__main__.py:1215-1257; at CONUS size the walks below are seconds)
    import os
    ckey = None
    if os.environ.get("TRMC_CALL_CACHE", "1") != "0":
        ckey = (id(reaches_bytw), id(independent_networks), id(param_df), id(waterbodies_df), id(waterbody_types_df), float(dt),
                len(reaches_bytw), len(independent_networks), tuple(param_df.shape), tuple(offnetwork_upstreams),
                param_df.iloc[:: max(1, len(param_df) // 257)].values.astype("float64").sum(axis=0).tobytes() if len(param_df) else b"")
    net = _NETWORKS.get(ckey) if ckey is not None else None
    if net is None:
        net = _network_of_call(reaches_bytw, independent_networks, param_df, dt, waterbodies_df, waterbody_types_df, offnetwork_upstreams)
        if ckey is not None:
            net["refs"] = (reaches_bytw, independent_networks, param_df, waterbodies_df, waterbody_types_df)   # (ids stay taken)
            _NETWORKS[ckey] = net
            while len(_NETWORKS) > 2:
                _NETWORKS.pop(next(iter(_NETWORKS)))
    tws, reaches_wTypes, reach_owner, upstream_connections = net["tws"], net["reaches_wTypes"], net["reach_owner"], net["upstream_connections"]
    lake_segs, waterbodies_sub, types_sub, table_index, table_cols, table_values = (
        net["lake_segs"], net["waterbodies_sub"], net["types_sub"], net["index"], net["columns"], net["values"])
    ids = net["ids"]
    nseg = ids.shape[0]

    def rows_of(df):
        """a caller's frame in the table's row order, float32 -- without a copy when it is in that order already"""
        if df.index is table_index or (len(df.index) == nseg and df.index.dtype == table_index.dtype
                                       and np.array_equal(df.index.values, ids)):
            v = df.values
            if v.dtype != np.float32:
                v = v.astype("float32")
            # (NaN -- the rows a reindexed frame does not hold -- become 0 on the device behind the upload: nan_is_zero below)
            return v
        return df.reindex(table_index).fillna(0.0).values.astype("float32")
    q0_v, qlat_v = rows_of(q0), rows_of(qlats)

    upstream_results = {}
    for u in offnetwork_upstreams:                                       # compute.py:1649-1655
        upstream_results[u] = {"results": np.asarray(flowveldepth_interorder[u]["results"]),
                               "position_index": int(table_index.get_loc(u))}

    e_f2, e_f1, e_i1 = np.zeros((0, 0), "float32"), np.zeros(0, "float32"), np.zeros(0, "int32")
    # streamflow nudging tables of the whole call (reference: per tailwater, compute.py:1468-1469)
    import pandas as pd
    if _is_empty(usgs_df) and _is_empty(lastobs_df):
        usgs_sub, lastobs_sub, da_byseg, da_byreach, da_bygage = pd.DataFrame(), pd.DataFrame(), [], [], np.zeros(0, dtype=np.intp)
    else:
        usgs_sub, lastobs_sub, da_byseg = _prep_da_dataframes(
            pd.DataFrame() if usgs_df is None else usgs_df, pd.DataFrame() if lastobs_df is None else lastobs_df, table_index)
        da_byreach, da_bygage = _prep_da_positions_byreach([reach for reach, _ in reaches_wTypes], lastobs_sub.index)
    ngage = len(da_byseg)
    if ngage:
        usgs_v = usgs_sub.values.astype("float32")
        null = pd.Series(index=lastobs_sub.index, name="Null", dtype="float32")
        lastobs_v = lastobs_sub.get("lastobs_discharge", null).values.astype("float32")     # compute.py:1533-1535
        lastobs_t = lastobs_sub.get("time_since_lastobs", null).values.astype("float32")
        gage_args = (usgs_v, np.array(da_byseg, dtype="int32"), np.array(da_byreach, dtype="int32"),
                     np.array(da_bygage, dtype="int32"), lastobs_v, lastobs_t)
    else:
        gage_args = (e_f2, e_i1, e_i1, e_i1, e_f1, e_f1)
    # the result in the order that groups the table's rows by tailwater (a network's block is then a slice of it), permuted on the
    # device as it is decimated -- when every row of the table belongs to a tailwater of the call and is in the result
    grouped = not upstream_results and not ngage and net["order"].shape[0] == nseg and bool(net["bounds"][0] == 0)
    r = compute_network_structured(
        nts, dt, qts_subdivisions, reaches_wTypes, upstream_connections, ids, table_cols,
        table_values, q0_v, qlat_v, lake_segs, waterbodies_sub, data_assimilation_parameters,
        types_sub, bool(waterbody_type_specified),
        t0.strftime('%Y-%m-%d_%H:%M:%S') if hasattr(t0, "strftime") else str(t0),
        *gage_args, da_parameter_dict.get("da_decay_coefficient", 0) if da_parameter_dict else 0,
        e_f2, e_i1, e_f1, e_f1, e_f1, e_f1, e_f1,
        e_f2, e_i1, e_f1, e_f1, e_f1, e_f1, e_f1,
        e_f2, e_i1, e_i1, [], e_i1, e_i1, e_f1, e_i1, e_i1,
        e_i1, e_i1, e_f1, e_i1, e_f1, e_i1, e_i1, e_f2,
        upstream_results, assume_short_ts, return_courant, from_files=from_files, precision=precision, device=device,
        output_stride=output_stride, result_order=net["order"] if grouped else None, nan_is_zero=True)
    rids = r[0].astype("int64")                  # (the rows of upstream_results are masked out, mc_reach.pyx:451,:812)
    fvd, upstream = r[1], r[6]
    gage_ids, lastobs_times, lastobs_values = r[3]
    nudge = r[8]
    if ngage:                                                            # gage rows in the (masked) result order
        gseg = ids[np.asarray(da_byseg, dtype=np.int64)]
        nres = rids.shape[0]
        gage_row = np.minimum(np.searchsorted(rids, gseg), max(nres - 1, 0))
        # a gage whose segment is an off-network upstream row (flowveldepth_interorder: masked out of the result,
        # mc_reach.pyx:451,:812) belongs to no tailwater of this call
        gage_row = np.where(rids[gage_row] == gseg, gage_row, -1) if nres else np.full(ngage, -1, dtype=np.int64)
    else:
        gage_row = np.zeros(0, dtype=np.int64)

    # ---- back to the reference's per-tailwater result list: ONE gather into tailwater order (the rows of a network are then a
    # slice -- a view -- of it; 14 713 boolean masks over 2.7 million rows were most of a minute at CONUS size) -------------
    owner, order, bounds = net["owner"], net["order"], net["bounds"]
    if rids.shape[0] != nseg:                    # (off-network rows masked out of the result: their positions drop out)
        keep = np.ones(nseg, dtype=bool)
        keep[[int(v["position_index"]) for v in upstream_results.values()]] = False
        owner = owner[keep]
        order = np.argsort(owner, kind="stable")
        bounds = np.searchsorted(owner[order], np.arange(len(tws) + 1))
    if grouped:                                  # (compute_network_structured has handed its rows back in that order)
        ids_o, fvd_o, up_o = rids.astype(np.intp), fvd, upstream
    else:
        ids_o, fvd_o = rids[order].astype(np.intp), fvd[order]
        # (the upstream-inflow series are zero except on waterbody rows, mc_reach.pyx:487,:710: without waterbodies any order of
        # an all-zero block is the block itself -- 3.1 GB for a CONUS day that need not be permuted)
        up_o = upstream[order] if len(lake_segs) else upstream
    e5, e3, e4 = (e_i1, e_f1, e_f1, e_f1, e_f1), (e_i1, e_f1, e_i1), (e_i1, e_f1, e_i1, e_i1)
    no_gage = (np.asarray([], dtype=np.int64), np.full(0, np.nan, "float32"), np.full(0, np.nan, "float32"))
    no_nudge = np.zeros((0, nts + 1), dtype="float32")
    gage_owner = owner[np.maximum(gage_row, 0)] if ngage else None
    results = []
    for k in range(len(tws)):
        lo, hi = int(bounds[k]), int(bounds[k + 1])
        if ngage:
            gk = np.flatnonzero((gage_row >= 0) & (gage_owner == k))                          # this network's gages
            gt = (np.asarray(gage_ids)[gk], np.asarray(lastobs_times)[gk], np.asarray(lastobs_values)[gk])
        results.append((ids_o[lo:hi], fvd_o[lo:hi], 0, gt if ngage else no_gage, e5, e5, up_o[lo:hi], e3,
                        nudge[gk] if ngage else no_nudge, e4))
    return results, subnetwork_list


_NETWORKS = {}


def _network_of_call(reaches_bytw, independent_networks, param_df, dt, waterbodies_df, waterbody_types_df, offnetwork_upstreams):
    """Everything compute_nhd_routing_v02 derives from the network alone: the reach list with its types, the merged upstream
    dictionary, the parameter table of the call (compute.py:548-549, :1399-1467: one table for every network of the call instead
    of one per tailwater), which tailwater owns every row, and the order that groups the rows by tailwater."""
    # compute.py:548-549
    param_df = param_df.copy()
    param_df["dt"] = dt
    param_df = param_df.astype("float32")
    cols = ["dt", "bw", "tw", "twcc", "dx", "n", "ncc", "cs", "s0"] + (["alt"] if "alt" in param_df.columns else [])

    # ---- one table for every network of the call (reference: one table per tailwater, compute.py:1399-1467)
    tws = list(reaches_bytw.keys())
    param_ids = set(param_df.index.tolist())
    reaches_wTypes, reach_owner, upstream_connections = [], [], {}
    seg_ids, lake_ids = [], []
    for k, tw in enumerate(tws):
        upstream_connections.update(independent_networks[tw])
        for reach in reaches_bytw[tw]:
            # _build_reach_type_list, compute.py:40-46 (a reach of more than one node never holds a waterbody node)
            is_wb = (reach[0] not in param_ids) if len(reach) == 1 else any(s not in param_ids for s in reach)
            reaches_wTypes.append((list(reach), 1 if is_wb else 0))
            reach_owner.append(k)
            (lake_ids if is_wb else seg_ids).extend(reach)
    have_wb = bool(lake_ids)
    if have_wb and _is_empty(waterbodies_df):
        raise ValueError("reaches contain waterbody nodes but waterbodies_df is empty")
    lake_segs = sorted(lake_ids)
    if have_wb:
        missing = [l for l in lake_segs if l not in waterbodies_df.index]
        if missing:
            raise ValueError(f"element {missing[0]} not found in {list(waterbodies_df.index)}")
        wb_cols = ["LkArea", "LkMxE", "OrificeA", "OrificeC", "OrificeE", "WeirC", "WeirE", "WeirL", "ifd", "qd0", "h0"]
        waterbodies_sub = waterbodies_df.loc[lake_segs, wb_cols].values.astype("float64")       # compute.py:1421-1436
        if not _is_empty(waterbody_types_df):
            types_sub = waterbody_types_df.loc[lake_segs, ["reservoir_type"]].values.astype("int32")
        else:
            types_sub = np.zeros((0, 1), dtype="int32")
    else:
        waterbodies_sub = np.zeros((0, 0), dtype="float64")
        types_sub = np.zeros((0, 0), dtype="int32")

    off_segs = [u for u in offnetwork_upstreams if u in param_ids]      # compute.py:1586-1589: joined to the table
    off_lakes = [u for u in offnetwork_upstreams if u not in param_ids]
    if off_lakes:
        if _is_empty(waterbodies_df) or any(l not in waterbodies_df.index for l in off_lakes):
            raise ValueError(f"flowveldepth_interorder key {off_lakes[0]} is neither a segment of param_df nor a waterbody")
        if not have_wb:
            wb_cols = ["LkArea", "LkMxE", "OrificeA", "OrificeC", "OrificeE", "WeirC", "WeirE", "WeirL", "ifd", "qd0", "h0"]
        lake_segs = sorted(set(lake_segs) | set(off_lakes))
        waterbodies_sub = waterbodies_df.loc[lake_segs, wb_cols].values.astype("float64")
        if not _is_empty(waterbody_types_df):
            types_sub = waterbody_types_df.loc[lake_segs, ["reservoir_type"]].values.astype("int32")
    seg_all = sorted(set(seg_ids) | set(off_segs))
    table = param_df.loc[seg_all, cols].reindex(seg_all + lake_segs).sort_index()                   # :1447-1465
    ids = np.ascontiguousarray(table.index.values.astype("int64"))
    nseg = ids.shape[0]
    # which tailwater owns every row of the table (-1: a row in no reach of the call), and the order that groups them
    owner = np.full(nseg, -1, dtype=np.int64)
    lens = np.fromiter((len(reach) for reach, _ in reaches_wTypes), dtype=np.int64, count=len(reaches_wTypes))
    flat = np.fromiter((s for reach, _ in reaches_wTypes for s in reach), dtype=np.int64, count=int(lens.sum()))
    owner[np.searchsorted(ids, flat)] = np.repeat(np.asarray(reach_owner, dtype=np.int64), lens)
    order = np.argsort(owner, kind="stable")
    bounds = np.searchsorted(owner[order], np.arange(len(tws) + 1))
    return {"tws": tws, "reaches_wTypes": reaches_wTypes, "reach_owner": reach_owner, "upstream_connections": upstream_connections,
            "lake_segs": lake_segs, "waterbodies_sub": waterbodies_sub, "types_sub": types_sub, "index": table.index,
            "columns": table.columns.values, "values": np.ascontiguousarray(table.values.astype("float32")), "ids": ids,
            "owner": owner, "order": order, "bounds": bounds}



def compute_diffusive_routing(results, diffusive_network_data, cpu_pool, t0, dt, nts, q0, qlats, qts_subdivisions, usgs_df,
                              lastobs_df, da_parameter_dict, waterbodies_df, topobathy, refactored_diffusive_domain,
                              refactored_reaches, coastal_boundary_depth_df, unrefactored_topobathy, *, device=0):
    """Diffusive-wave routing of the mainstem domains after the Muskingum-Cunge pass -- same signature and result list
    as the reference (compute.py:1740-1885): per tailwater, the MC flows of the tributary segments
    (``results[...][1][x, ::3]``) become the junction inflows, ``diffusive_input_data_v02`` marshals the solver's
    arguments, the solver runs, ``unpack_output`` turns its arrays into result rows, tributary rows are masked out.
    Different on purpose: every tailwater domain of the call is solved in ONE launch
    (``compute_diffusive_batch``, one compute unit per domain) instead of one after the other."""
    import pandas as pd
    from . import diffusive_utils_v02 as diff_utils
    from .fast_reach import diffusive

    def empty(df):
        return df is None or getattr(df, "empty", True)
    # the gage table goes to the marshalling when the key is present, whatever its value (compute.py:1798-1803); the solver
    # -- the reference's Fortran and this one -- copies the arrays fp_da_map makes of it and computes nothing from them
    # (diffusive.f90:1282-1303, :1316-1319: the nudging branch is commented out)
    if da_parameter_dict and "diffusive_streamflow_nudging" in da_parameter_dict and not empty(usgs_df):
        diffusive_usgs_df = usgs_df
    else:
        diffusive_usgs_df = pd.DataFrame()
    tws = list(diffusive_network_data)
    inputs = []
    for tw in tws:
        dn = diffusive_network_data[tw]
        trib_segs, trib_flow = None, None
        for r in results:                                                    # compute.py:1766-1781
            if r[1].shape[1] != 3 * nts:
                # (a result decimated with output_stride holds every n-th step only: its columns are not the nts junction
                # inflows the solver takes)
                raise ValueError(f"the Muskingum-Cunge results hold {r[1].shape[1] // 3} steps per row, not nts = {nts}: route the "
                                 "windows that feed the diffusive pass without output_stride")
            x = np.isin(r[0], dn["tributary_segments"])
            if x.sum() > 0:
                trib_segs = r[0][x] if trib_segs is None else np.append(trib_segs, r[0][x])
                trib_flow = r[1][x, ::3] if trib_flow is None else np.append(trib_flow, r[1][x, ::3], axis=0)
        junction_inflows = pd.DataFrame(data=trib_flow, index=trib_segs)
        coastal = (coastal_boundary_depth_df.loc[tw].to_frame().T
                   if not empty(coastal_boundary_depth_df) and tw in coastal_boundary_depth_df.index else pd.DataFrame())
        if not empty(topobathy):                                                           # compute.py:1783-1796
            topo = topobathy.loc[refactored_diffusive_domain[tw]["rlinks"] if refactored_diffusive_domain else dn["mainstem_segs"]]
        else:
            topo = pd.DataFrame()
        # the network as refactored for the diffusive solver, by tailwater (compute.py:1805-1812)
        if refactored_diffusive_domain:
            rdomain = refactored_diffusive_domain[tw]
            rreaches = refactored_reaches[rdomain["refac_tw"]]
        else:
            rdomain, rreaches = None, None
        dq = qlats.copy()
        dq.columns = range(dq.shape[1])                                       # compute.py:1822-1823
        inputs.append(diff_utils.diffusive_input_data_v02(
            tw, dn["connections"], dn["rconn"], dn["reaches"], dn["mainstem_segs"], dn["tributary_segments"], None,
            dn["param_df"], dq, q0, junction_inflows, qts_subdivisions, t0, nts, dt, waterbodies_df, topo,
            diffusive_usgs_df, rdomain, rreaches, coastal, pd.DataFrame()))
    outs = diffusive.compute_diffusive_batch(inputs, device=device)
    e = np.asarray([])
    results_diffusive = []
    for tw, ins, (out_q, out_elv, out_depth) in zip(tws, inputs, outs):
        rch_list, dat_all = diff_utils.unpack_output(ins["pynw"], ins["ordered_reaches"], out_q, out_depth)
        x = np.isin(rch_list, diffusive_network_data[tw]["tributary_segments"])    # MC already routed these
        results_diffusive.append((
            rch_list[~x], dat_all[~x, 3:], 0, (e, e, e), (e, e, e, e, e), (e, e, e, e, e),
            np.zeros(dat_all[~x, 3::3].shape), (e, e, e), np.empty(shape=(0, nts + 1), dtype="float32"), (e, e, e, e)))
    return results_diffusive


def new_q0(run_results, index=None):
    """State for the next routing window from a results list: ``[qu0, qd0, h0] = fvd[:, [-3,-3,-1]]``
    (AbstractNetwork.new_q0, AbstractNetwork.py:177-191), as a DataFrame indexed by segment id."""
    import pandas as pd
    ids = np.concatenate([r[0] for r in run_results])
    vals = np.concatenate([r[1][:, [-3, -3, -1]] for r in run_results])
    df = pd.DataFrame(vals, index=ids, columns=["qu0", "qd0", "h0"])
    return df if index is None else df.loc[index]
