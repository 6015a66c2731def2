"""Input marshalling and output unpacking of the diffusive-wave solver -- the host side of SURVEY 8f rank 3.

Same names, arguments and results as the reference module
``src/troute-routing/troute/routing/diffusive_utils_v02.py`` for what the Muskingum-Cunge + diffusive hybrid
configuration uses:

  diffusive_input_data_v02   :659-1155  network of one tailwater -> the argument dictionary of the solver
                             (``troute_amd.routing.fast_reach.diffusive.compute_diffusive`` / ``c_diffnw``)
  unpack_output              :1156-1212 solver outputs -> (segment ids, [flow, nan, depth] x time) rows

Own implementation: the reference fills its arrays one ``DataFrame.loc`` scalar at a time inside four nested loops;
here every reach is one block of rows taken from the parameter / forcing tables.  The arithmetic types are the
reference's (float32 table columns, e.g. ``qlat / dx`` and ``1 / cs`` are formed in float32 and then stored in the
float64 arrays), so the dictionaries are equal to the reference's bit for bit -- pinned on the LowerColorado coastal
subset against the dictionary the reference itself produced (tests/golden/diffusive_lowercolorado.npz).

Natural cross sections (``topobathy_bytw``, fp_naturalxsec_map :394-510) and the coastal depth boundary
(``coastal_boundary_depth_df``, fp_coastal_boundary_input_map :575-657) are marshalled as the reference does.

Refactored hydrofabric (``refactored_diffusive_domain`` / ``refactored_reaches``): forwarded as the reference forwards them
(compute.py:1785-1812).  The reference's own v02 marshalling keeps EMPTY placeholders for the crosswalk arguments
(:1033-1038) and its refactored branch then reads names that are never assigned (``rz_ar_g`` ... :1115-1142), so it
cannot run; here the solver's crosswalk arguments (``rdx_ar_g``, ``crosswalk_g``, ``z_thalweg_g``: what ``trdw_diffnw``
maps results back to the original links with, diffusive.f90:849-920) are taken from the domain dictionary when the caller
supplies them under those names, and without them the call fails with a message that says so.
Gage data for diffusive nudging (``usgs_df``): marshalled by ``fp_da_map`` exactly as the reference does (:512-574) and
handed to the solver, which -- like the reference's Fortran, whose use of ``usgs_da`` is commented out
(diffusive.f90:1282-1303, :1316-1319) -- copies the arrays and computes nothing from them.
"""
import math
from functools import partial

import numpy as np

from .. import nhd_network


def _fake_id(seg):
    """id of the ghost node the reference appends below the last segment of a reach (:771, :828)"""
    return int(str(seg) + str(2))


def _ordered_reaches(tw, connections, rconn, junction_inflows):
    """Reaches by junction order, in the reference's enumeration (:765-806): ``ordered[o]`` = list of
    ``[head, {number_segments, segments_list (with the ghost node), upstream_bottom_segments, downstream_head_segment}]``;
    ``pynw[frj]`` = head segment of Fortran reach frj (orders from the highest down, list order inside an order)."""
    path_func = partial(nhd_network.split_at_waterbodies_and_junctions, set(junction_inflows.index.to_list()), rconn)
    tr = nhd_network.dfs_decomposition_depth_tuple(rconn, path_func)
    jorder_reaches = sorted(tr, key=lambda x: x[0])
    mx_jorder = max(o for o, _ in jorder_reaches)
    ordered, bottoms = {}, {}
    for o, rch in jorder_reaches:
        rch = list(rch) + [_fake_id(rch[-1])]
        meta = {
            "number_segments": len(rch),
            "segments_list": rch,
            "upstream_bottom_segments": [_fake_id(x) for x in rconn[rch[0]]],
            "downstream_head_segment": connections[rch[-2]],
        }
        ordered.setdefault(o, []).append([rch[0], meta])
        bottoms[rch[-1]] = rch
    pynw, frj = {}, -1
    for x in range(mx_jorder, -1, -1):
        for head, _ in ordered[x]:
            frj += 1
            pynw[frj] = head
    return ordered, bottoms, pynw, mx_jorder


def fp_network_map(mainstem_seg_list, trib_seg_list, mx_jorder, ordered_reaches, rchbottom_reaches, nrch_g, frnw_col,
                   dbfksegID, pynw):
    """Fortran network map (:55-165): per reach [node count, downstream reach (1-based; -99 at the tailwater),
    number of upstream reaches, their indices (1-based) ..., 555 mainstem / -555 tributary]."""
    frnw_g = np.zeros((nrch_g, frnw_col), dtype="int32")
    frj_of_head = {head: frj for frj, head in pynw.items()}
    main, trib = set(mainstem_seg_list), set(trib_seg_list)
    frj = -1
    for x in range(mx_jorder, -1, -1):
        for head, reach in ordered_reaches[x]:
            frj += 1
            segs = reach["segments_list"]
            frnw_g[frj, 0] = reach["number_segments"]
            ups = list(reach["upstream_bottom_segments"])
            frnw_g[frj, 2] = len(ups)
            col = 3
            for b in ups:
                # the reach whose ghost node is b: its head names the Fortran index
                frnw_g[frj, col] = frj_of_head[rchbottom_reaches[b][0]] + 1
                col += 1
            if head in main:
                frnw_g[frj, 3 + len(ups)] = 555
            if head in trib:
                frnw_g[frj, 3 + len(ups)] = -555
            if dbfksegID in segs:
                frnw_g[frj, 1] = -100 + 1
            else:
                frnw_g[frj, 1] = frj_of_head[reach["downstream_head_segment"][0]] + 1
    return frnw_g


def fp_naturalxsec_map(ordered_reaches, mainstem_seg_list, topobathy_bytw, param_df, mx_jorder, mxncomp_g, nrch_g,
                       dbfksegID):
    """Station tables of the mainstem nodes (:394-510): ``x/z/mann_bathy_g[station, node, reach]`` and the station
    count per node.  A node carries its segment's cross section; the bottom node of a reach takes the head segment of
    the reach below; the tailwater's bottom node repeats the last segment's section lowered by slope x length."""
    if topobathy_bytw is None or topobathy_bytw.empty:
        z3 = np.array([]).reshape(0, 0, 0)
        return z3, z3.copy(), z3.copy(), np.array([], dtype="i4").reshape(0, 0), 0
    newer = "cs_id" in topobathy_bytw.columns                       # the two column vocabularies of the input file
    cx, cz, cn = ("relative_dist", "Z", "roughness") if newer else ("xid_d", "z", "n")
    mxnbathy_g = int(topobathy_bytw.index.value_counts().max())
    x_bathy_g, z_bathy_g, mann_bathy_g = (np.zeros((mxnbathy_g, mxncomp_g, nrch_g)) for _ in range(3))
    size_bathy_g = np.zeros((mxncomp_g, nrch_g), dtype="i4")
    # one pass over the table: rows of every cross section, in file order
    rows = {}
    for pos, key in enumerate(topobathy_bytw.index.values):
        rows.setdefault(key, []).append(pos)
    xs, zs, ns = (topobathy_bytw[c].values for c in (cx, cz, cn))
    main = set(mainstem_seg_list)
    frj = -1
    for o in range(mx_jorder, -1, -1):
        for head, reach in ordered_reaches[o]:
            frj += 1
            if head not in main:
                continue
            segs, ncomp = reach["segments_list"], reach["number_segments"]
            for k, seg in enumerate(segs):
                if k == ncomp - 1 and o > 0:
                    src = reach["downstream_head_segment"][0]
                elif seg == dbfksegID:
                    src = segs[k - 1]
                else:
                    src = seg
                r = rows[src]
                m = len(r)
                size_bathy_g[k, frj] = m
                x_bathy_g[:m, k, frj] = xs[r]
                z_bathy_g[:m, k, frj] = zs[r]
                mann_bathy_g[:m, k, frj] = ns[r]
                if seg == dbfksegID:
                    z_bathy_g[:m, k, frj] = z_bathy_g[:m, k, frj] - param_df.loc[src].s0 * param_df.loc[src].dx
    return x_bathy_g, z_bathy_g, mann_bathy_g, size_bathy_g, mxnbathy_g


def fp_coastal_boundary_input_map(tw, coastal_boundary_depth_df, nrch_g, t0, t0_g, tfin_g):
    """Water depth series at the tailwater (:575-657) -> (interval [s], boundary option, count, series).  Depths <= 0
    become the smallest positive depth of the row; gaps are bridged linearly over at most 6 records from either side;
    anything still missing in ANY row of the table switches the domain back to the normal-depth boundary (option 2)."""
    import pandas as pd
    if coastal_boundary_depth_df is None or coastal_boundary_depth_df.empty:
        nts_db_g = int((tfin_g - t0_g) * 3600.0 / 3600.0) + 1
        return 3600.0, 2, nts_db_g, np.zeros(nts_db_g)
    c = coastal_boundary_depth_df.columns
    dt_db_g = (c[1] - c[0]).total_seconds()
    nts_db_g = int((tfin_g - t0_g) * 3600.0 / dt_db_g) + 1
    step = pd.Timedelta(minutes=dt_db_g / 60.0)
    stamps = pd.date_range(t0, t0 + step * (nts_db_g - 1), freq=step)
    df = coastal_boundary_depth_df.reindex(columns=stamps).astype(float)
    row = df.loc[tw]
    df.loc[tw, row <= 0] = row.where(row > 0).min()
    df = df.interpolate(axis="columns", limit_direction="both", limit=6)
    if df.isnull().values.any():
        return dt_db_g, 2, nts_db_g, np.zeros(nts_db_g)
    return dt_db_g, 1, nts_db_g, df.loc[tw].values.astype(float)


def fp_da_map(mx_jorder, ordered_reaches, usgs_df, nrch_g, t0, nsteps, dt_da_g, t0_g, tfin_g):
    """Gage observations for the solver (:512-574) -> (number of DA times, ``usgs_da_g[time, reach]``, 1-based index of
    every reach that holds a gage segment, 0 elsewhere).  A reach takes the series of the LAST of its nodes found in the
    table; missing stamps and NaN become -4444."""
    import pandas as pd
    nts_da_g = int((tfin_g - t0_g) * 3600.0 / dt_da_g) + 1
    usgs_da_g = -4444.0 * np.ones((nts_da_g, nrch_g))
    usgs_da_reach_g = np.zeros(nrch_g, dtype="i4")
    if usgs_df is None or usgs_df.empty:
        return nts_da_g, usgs_da_g, usgs_da_reach_g
    step = pd.Timedelta(minutes=dt_da_g / 60.0)
    stamps = pd.date_range(t0, t0 + step * nsteps, freq=step)
    table = usgs_df.reindex(columns=stamps).fillna(-4444.0)
    where = {seg: k for k, seg in enumerate(table.index.values)}
    values = table.values
    frj = -1
    for x in range(mx_jorder, -1, -1):
        for _, reach in ordered_reaches[x]:
            frj += 1
            hits = [where[s] for s in reach["segments_list"] if s in where]
            if hits:
                usgs_da_g[:, frj] = values[hits[-1], :nts_da_g]
                usgs_da_reach_g[frj] = frj + 1
    return nts_da_g, usgs_da_g, usgs_da_reach_g


def diffusive_input_data_v02(tw, connections, rconn, reach_list, mainstem_seg_list, trib_seg_list, diffusive_parameters,
                             param_df, qlat, initial_conditions, junction_inflows, qts_subdivisions, t0, nsteps, dt,
                             waterbodies_df, topobathy_bytw, usgs_df, refactored_diffusive_domain, refactored_reaches,
                             coastal_boundary_depth_df, unrefactored_topobathy_bytw):
    """The solver's argument dictionary for the network draining to tailwater `tw` (reference :659-1155)."""
    def empty(df):
        return df is None or getattr(df, "empty", True)
    crosswalk = None
    if refactored_diffusive_domain:
        need = ("rdx_ar_g", "crosswalk_g", "z_thalweg_g")
        if not all(k in refactored_diffusive_domain for k in need):
            raise NotImplementedError(
                "refactored hydrofabric: the reference's own marshalling leaves the crosswalk arguments empty and its refactored "
                "branch reads unassigned names (diffusive_utils_v02.py:1033-1038, :1115-1142); supply 'rdx_ar_g', 'crosswalk_g' "
                "and 'z_thalweg_g' in refactored_diffusive_domain (the solver maps results back with them, trdw_diffnw cwnrow_g > 0)")
        crosswalk = {k: np.asarray(refactored_diffusive_domain[k], dtype=np.float64) for k in need}

    # ---- time and numerical parameters (:700-745)
    dt_ql_g, saveinterval, tfin_g = 3600.0, dt, (dt * nsteps) / 60 / 60
    timestep_ar_g = np.zeros(10)
    timestep_ar_g[[0, 1, 2, 3, 4, 5, 7, 8, 9]] = [dt, 0.0, tfin_g, saveinterval, dt_ql_g, dt, dt, dt, 10.0]
    para_ar_g = np.array([0.95, 0.5, 10.0, 10000.0, -15.0, -10.0, 1.0, 0.02831, 0.0001, 1.0, 2.0])

    nrch_g = len(reach_list)
    mxncomp_g = max(len(r) + 1 for r in reach_list)
    ordered, bottoms, pynw, mx_jorder = _ordered_reaches(tw, connections, rconn, junction_inflows)
    dbfksegID = _fake_id(tw)
    frnw_col = 20
    frnw_g = fp_network_map(mainstem_seg_list, trib_seg_list, mx_jorder, ordered, bottoms, nrch_g, frnw_col, dbfksegID, pynw)

    # ---- per-reach blocks of the parameter table (fp_chgeo_map :168-240, adj_alt1 :10-52, iniq :858-872,
    #      fp_qlat_map :242-289, tributary hydrographs :905-912)
    shape = (mxncomp_g, nrch_g)
    z_ar_g, bo_ar_g, traps_ar_g, tw_ar_g, twcc_ar_g = (np.zeros(shape) for _ in range(5))
    mann_ar_g, manncc_ar_g, so_ar_g, dx_ar_g, iniq = (np.zeros(shape) for _ in range(5))
    nts_ql_g = math.ceil(tfin_g * 3600.0 / dt_ql_g)
    qlat_g = np.zeros((nts_ql_g, mxncomp_g, nrch_g))
    nts_qtrib_g = int(tfin_g * 3600.0 / dt) + 1
    qtrib_g = np.zeros((nts_qtrib_g, nrch_g))
    main = set(mainstem_seg_list)
    cols = {c: param_df[c] for c in ("bw", "cs", "tw", "twcc", "n", "ncc", "s0", "dx", "alt")}
    qu0 = initial_conditions["qu0"]
    qlat_cols = list(range(nts_ql_g))
    by_frj = [m for x in range(mx_jorder, -1, -1) for _, m in ordered[x]]     # the enumeration pynw follows
    for frj, head in pynw.items():
        reach = by_frj[frj]
        segs = reach["segments_list"]
        n = reach["number_segments"]
        real = segs[:-1]                              # without the ghost node
        src = real + [real[-1]]                       # the bottom node repeats the last segment's geometry
        bo_ar_g[:n, frj] = cols["bw"].loc[src].values
        traps_ar_g[:n, frj] = 1 / cols["cs"].loc[src].values
        tw_ar_g[:n, frj] = cols["tw"].loc[src].values
        twcc_ar_g[:n, frj] = cols["twcc"].loc[src].values
        mann_ar_g[:n, frj] = cols["n"].loc[src].values
        manncc_ar_g[:n, frj] = cols["ncc"].loc[src].values
        so_ar_g[:n, frj] = cols["s0"].loc[src].values
        dx_ar_g[:n, frj] = cols["dx"].loc[src].values
        # node elevations: the segments' altitudes; the bottom node takes the head altitude of the reach below, or,
        # at the tailwater, drops by slope x length of the last segment
        z_ar_g[:n - 1, frj] = cols["alt"].loc[real].values
        if dbfksegID in segs:
            last = real[-1]
            z_ar_g[n - 1, frj] = z_ar_g[n - 2, frj] - cols["s0"].loc[last] * cols["dx"].loc[last]
        else:
            z_ar_g[n - 1, frj] = float(cols["alt"].loc[reach["downstream_head_segment"]].iloc[0])
        q = qu0.loc[src].values.astype(np.float64)
        iniq[:n, frj] = np.where(q < 0.0001, 0.0001, q)
        qlat_g[:, :n - 1, frj] = (qlat.loc[real, qlat_cols].values / cols["dx"].loc[real].values[:, None]).T
        if head not in main:
            qtrib_g[1:, frj] = junction_inflows.loc[head]
            qtrib_g[0, frj] = qu0.loc[head]

    nts_ub_g = int(tfin_g * 3600.0 / dt)
    dt_db_g, dsbd_option, nts_db_g, dbcd_g = fp_coastal_boundary_input_map(tw, coastal_boundary_depth_df, nrch_g, t0, 0.0,
                                                                           tfin_g)
    timestep_ar_g[6] = dt_db_g
    para_ar_g[10] = dsbd_option
    x_bathy_g, z_bathy_g, mann_bathy_g, size_bathy_g, mxnbathy_g = fp_naturalxsec_map(
        ordered, mainstem_seg_list, topobathy_bytw, param_df, mx_jorder, mxncomp_g, nrch_g, dbfksegID)
    nts_da_g, usgs_da_g, usgs_da_reach_g = fp_da_map(mx_jorder, ordered, usgs_df, nrch_g, t0, nsteps, dt, 0.0, tfin_g)  # :1022-1031
    ins = {
        "timestep_ar_g": timestep_ar_g, "nts_ql_g": nts_ql_g, "nts_ub_g": nts_ub_g, "nts_db_g": nts_db_g,
        "nts_qtrib_g": nts_qtrib_g, "ntss_ev_g": int(tfin_g * 3600.0 / dt) + 1, "nts_da_g": nts_da_g,
        "mxncomp_g": mxncomp_g, "nrch_g": nrch_g, "z_ar_g": z_ar_g, "bo_ar_g": bo_ar_g, "traps_ar_g": traps_ar_g,
        "tw_ar_g": tw_ar_g, "twcc_ar_g": twcc_ar_g, "mann_ar_g": mann_ar_g, "manncc_ar_g": manncc_ar_g,
        "so_ar_g": so_ar_g, "dx_ar_g": dx_ar_g, "frnw_col": frnw_col, "frnw_g": frnw_g, "qlat_g": qlat_g,
        "ubcd_g": np.zeros((nts_ub_g, nrch_g)), "dbcd_g": dbcd_g, "qtrib_g": qtrib_g, "paradim": 11,
        "para_ar_g": para_ar_g, "mxnbathy_g": mxnbathy_g, "x_bathy_g": x_bathy_g, "z_bathy_g": z_bathy_g,
        "mann_bathy_g": mann_bathy_g, "size_bathy_g": size_bathy_g, "iniq": iniq, "pynw": pynw, "ordered_reaches": ordered,
        "usgs_da_g": usgs_da_g, "usgs_da_reach_g": usgs_da_reach_g,
        "rdx_ar_g": np.array([]).reshape(0, 0), "cwnrow_g": 0, "cwncol_g": 0, "crosswalk_g": np.array([]).reshape(0, 0),
        "z_thalweg_g": np.array([]).reshape(0, 0),
    }
    if crosswalk is not None:                                       # results mapped back to the original links by the solver
        ins.update(crosswalk)
        ins["cwnrow_g"], ins["cwncol_g"] = (int(x) for x in crosswalk["crosswalk_g"].shape)
    return ins


def unpack_output(pynw, ordered_reaches, out_q, out_elv):
    """(segment ids, rows [flow, nan, depth-or-elevation] x recorded times, float32) -- reference :1156-1212: node k+1
    of a reach carries segment k's result; reaches in the dictionary order of `ordered_reaches`."""
    frj_of_head = {head: frj for frj, head in pynw.items()}
    nts = out_q.shape[0]
    ids, blocks = [], []
    for o in ordered_reaches.keys():
        for head, meta in ordered_reaches[o]:
            segs = meta["segments_list"]
            j = frj_of_head[head]
            ids.extend(segs[:-1])
            blk = np.full((len(segs) - 1, nts * 3), np.nan)
            blk[:, ::3] = np.asarray(out_q)[:, 1:len(segs), j].T
            blk[:, 2::3] = np.asarray(out_elv)[:, 1:len(segs), j].T
            blocks.append(blk)
    return np.asarray(ids, dtype=np.intp), np.asarray(np.concatenate(blocks), dtype="float32")
