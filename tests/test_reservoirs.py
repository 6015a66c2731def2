"""Level-pool reservoirs: reference Fortran -> oracle -> GPU, and the graph collapse."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O
from troute_amd import nhd_network as nn

WB = np.load(os.path.join(H.GOLDEN, "lowercolorado_waterbodies.npz"))


def lake_tables():
    """waterbody arrays as compute_nhd_routing_v02 hands them to the kernel (compute.py:1421-1436): columns
    LkArea LkMxE OrificeA OrificeC OrificeE WeirC WeirE WeirL ifd qd0 h0; cold start: qd0 = 0, h0 = -1e9."""
    lakes = WB["lake_ids"]
    tab = WB["lake_table"]
    wbody_cols = np.concatenate([tab, np.zeros((len(lakes), 1)), np.full((len(lakes), 1), -1.0e9)], 1)
    return lakes, wbody_cols


def collapsed_network():
    lc = H.LowerColorado()
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    wbody_map = {int(s): int(w) for s, w in zip(WB["seg_ids"], WB["wb_of_seg"]) if w != -9999}
    conn_wb, link_lake = nn.replace_waterbodies_connections(conn, wbody_map)
    return lc, conn, wbody_map, conn_wb, link_lake


def test_graph_collapse_equals_reference():
    lc, conn, wbody_map, conn_wb, link_lake = collapsed_network()
    want = {int(n): ([int(t)] if t != 0 else []) for n, t in zip(WB["ref_conn_nodes"], WB["ref_conn_to"])}
    assert conn_wb == want
    assert list(conn_wb) == WB["ref_conn_nodes"].tolist()                 # same node order -> same rconn order
    assert set(link_lake) == set(WB["ref_link_lake_keys"].tolist())
    lakes = set(WB["lake_ids"].tolist())
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn_wb, lakes, set())
    rp, ri = WB["ref_reach_ptr"], WB["ref_reach_ids"]
    want_reaches = {tuple(ri[rp[i]:rp[i + 1]].tolist()) for i in range(len(rp) - 1)}
    got = {tuple(r) for rl in reaches_bytw.values() for r in rl}
    assert got == want_reaches
    assert all([lk] in [list(r) for rl in reaches_bytw.values() for r in rl] for lk in lakes)   # lakes are singletons


@pytest.mark.skipif(not O.have_ref("liblp_ref.so"), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_levelpool_equals_reference_fortran_live():
    lp = C.CDLL(O.ref_path("liblp_ref.so"))
    lp.get_lp_handle.restype = C.c_void_p
    f = C.c_float
    lakes, wbody_cols = lake_tables()
    rng = np.random.default_rng(2)
    for k in range(len(lakes)):
        a = wbody_cols[k].astype(np.float32)
        par = np.array([a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], 10.0], np.float32)
        h0 = np.float32(a[4] + np.float32(np.float32(a[1] - a[4]) * a[8]))
        hv = C.c_void_p(lp.get_lp_handle())
        we_ = f(float(h0))
        lp.init_lp(hv, C.byref(we_), *[C.byref(f(float(v))) for v in (par[0], par[6], par[5], par[7], par[8], par[4], par[3],
                                                                     par[2], par[1])], C.byref(C.c_int(int(lakes[k]))),
                   C.byref(C.c_int(1)))
        Hr, Hm = f(float(h0)), np.float32(h0)
        for _ in range(40):
            inflow = np.float32(rng.lognormal(0, 2))
            out = f(0)
            lp.run_lp(hv, C.byref(f(float(inflow))), C.byref(f(0.0)), C.byref(Hr), C.byref(out), C.byref(f(300.0)))
            q, Hm = O.levelpool(inflow, 300.0, Hm, par)
            assert np.float32(out.value).tobytes() == q.tobytes() and np.float32(Hr.value).tobytes() == Hm.tobytes()
        lp.free_lp(hv)


def reservoir_case(nts=288):
    lc, conn, wbody_map, conn_wb, link_lake = collapsed_network()
    lakes, wbody_cols = lake_tables()
    lakeset = set(lakes.tolist())
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn_wb, lakeset, set())
    tw = next(iter(reaches_bytw))
    assert len(reaches_bytw) == 1
    reaches, net = reaches_bytw[tw], ind[tw]
    segs = sorted(s for r in reaches for s in r if s not in lakeset)
    # table = routed segments + lake rows (param rows of lakes are NaN: compute.py:1447-1450,:1463-1465)
    ids = np.array(sorted(segs + lakes.tolist()), np.int64)
    row_old = {int(s): i for i, s in enumerate(lc.ids)}
    is_lake = np.isin(ids, lakes)
    dv = np.full((len(ids), lc.data_values.shape[1]), np.nan, np.float32)
    ql = np.zeros((len(ids), lc.qlat.shape[1]), np.float32)
    sel = np.array([row_old[int(s)] for s in ids[~is_lake]])
    dv[~is_lake] = lc.data_values[sel]
    ql[~is_lake] = lc.qlat[sel] * 40.0            # wetter than the fixture's forcing so the pools move
    q0 = np.zeros((len(ids), 3), np.float32)
    return lc, ids, dv, ql, q0, reaches, net, lakes, wbody_cols, lakeset, nts


@pytest.mark.gpu
@pytest.mark.parametrize("short,engine", [(True, None), (False, None), (True, "levels"), (True, "levels-wide"), (True, "levels-mid")])
def test_gpu_reservoirs_bit_identical_to_oracle(short, engine, monkeypatch):
    """engine: None = the default (dataflow engine at this size); "levels" = k_mc_step; "levels-wide" = the level engine with
    its wide levels several steps per launch under a level skew (k_mc_tile); "levels-mid" = the same with a second tier
    below the wide levels, fewer steps per launch under its own skew"""
    if engine:
        monkeypatch.setenv("TRMC_ENGINE", "levels")
        monkeypatch.setenv("TRMC_PLAN_CACHE", "0")
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "0" if engine == "levels" else ("64" if engine.endswith("mid") else "32"))
        monkeypatch.setenv("TRMC_WIDE_K", "7")
        monkeypatch.setenv("TRMC_MID_MIN_ROWS", "8" if engine.endswith("mid") else "0")
        monkeypatch.setenv("TRMC_MID_K", "3")
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
    lc, ids, dv, ql, q0, reaches, net, lakes, wbody_cols, lakeset, nts = reservoir_case()
    args = mc_only_args(nts, lc.dt, lc.qts, reaches, net, ids, lc.data_cols, dv, q0, ql, assume_short_ts=short)
    args[3] = [(r, 1 if r[0] in lakeset else 0) for r in reaches]        # _build_reach_type_list, compute.py:40-46
    args[10] = lakes.tolist()
    args[11] = wbody_cols
    args[13] = np.ones((len(lakes), 1), np.int32)
    args[14] = False
    r = compute_network_structured(*args)
    fvd = r[1].reshape(len(ids), nts, 3)

    row = {int(s): i for i, s in enumerate(ids)}
    rl = [np.array([row[s] for s in rr], dtype=np.int64) for rr in reaches]
    ul = [np.array([row[s] for s in net.get(rr[0], [])], dtype=np.int64) for rr in reaches]
    res_of_reach = np.full(len(reaches), -1, np.int64)
    lake_pos = {int(l): k for k, l in enumerate(lakes)}
    for i, rr in enumerate(reaches):
        if rr[0] in lakeset:
            res_of_reach[i] = lake_pos[rr[0]]
    a = wbody_cols.astype(np.float32)
    par = np.concatenate([a[:, :8], np.full((len(lakes), 1), 10.0, np.float32)], 1)
    h0 = (a[:, 4] + ((a[:, 1] - a[:, 4]).astype(np.float32) * a[:, 8]).astype(np.float32)).astype(np.float32)
    res = dict(res_of_reach=res_of_reach, par=par, water_elevation=h0, routing_period=lc.dt)
    params9 = dv[:, [H.DATA_COLS.index(c) for c in ("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")]]
    q0o = q0.copy()
    want = O.network(nts, lc.qts, rl, ul, params9, q0o, ql, short, det=True, res=res)
    assert np.array_equal(fvd.view(np.uint32), np.ascontiguousarray(want[:, 1:, :]).view(np.uint32))
    lake_rows = np.array([row[int(l)] for l in lakes])
    assert np.array_equal(r[6][lake_rows].view(np.uint32), res["inflow"][:, 1:].view(np.uint32))   # upstream_array rows
    assert np.abs(fvd[lake_rows, :, 0]).max() > 0                         # some pool actually spills
    assert (fvd[lake_rows, :, 1] == 0).all()                               # velocity slot of a reservoir row
    assert np.array_equal(fvd[lake_rows, -1, 2], res["water_elevation"])  # final pool elevations


@pytest.mark.gpu
def test_gpu_unsupported_reservoir_type_raises():
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
    lc, ids, dv, ql, q0, reaches, net, lakes, wbody_cols, lakeset, nts = reservoir_case()
    args = mc_only_args(12, lc.dt, lc.qts, reaches, net, ids, lc.data_cols, dv, q0, ql)
    args[3] = [(r, 1 if r[0] in lakeset else 0) for r in reaches]
    args[10], args[11] = lakes.tolist(), wbody_cols
    args[13] = np.full((len(lakes), 1), 2, np.int32)                       # USGS hybrid persistence
    args[14] = True
    with pytest.raises(NotImplementedError, match="reservoir type 2"):
        compute_network_structured(*args)


@pytest.mark.gpu
def test_gpu_compute_nhd_routing_v02_with_waterbodies():
    """The top-level seam with break_network_at_waterbodies: DataFrames in, per-tailwater results out,
    equal to the kernel callable's result for the same table."""
    import pandas as pd
    from troute_amd.routing.compute import compute_nhd_routing_v02
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
    lc, ids, dv, ql, q0, reaches, net, lakes, wbody_cols, lakeset, nts = reservoir_case(nts=48)
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    wbody_map = {int(s): int(w) for s, w in zip(WB["seg_ids"], WB["wb_of_seg"]) if w != -9999}
    conn_wb, _ = nn.replace_waterbodies_connections(conn, wbody_map)
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn_wb, lakeset, set())
    is_lake = np.isin(ids, lakes)
    cols = list(lc.data_cols)
    param_df = pd.DataFrame(dv[~is_lake], index=ids[~is_lake], columns=cols).drop(columns=["dt"])
    q0_df = pd.DataFrame(q0[~is_lake], index=ids[~is_lake], columns=["qu0", "qd0", "h0"])
    ql_df = pd.DataFrame(ql[~is_lake], index=ids[~is_lake])
    wb_df = pd.DataFrame(wbody_cols, index=lakes, columns=["LkArea", "LkMxE", "OrificeA", "OrificeC", "OrificeE", "WeirC",
                                                           "WeirE", "WeirL", "ifd", "qd0", "h0"])
    e = pd.DataFrame()
    res = compute_nhd_routing_v02(conn_wb, rconn, wbody_map, reaches_bytw, "V02-structured", "by-network", 10000, 4,
                                  None, lc.dt, nts, lc.qts, ind, param_df, q0_df, ql_df, e, e, e, e, e, e, e, e, e, e, e,
                                  {}, True, False, wb_df, {}, e, False, [{}, {}])[0]
    assert len(res) == 1 and np.array_equal(res[0][0], ids)
    args = mc_only_args(nts, lc.dt, lc.qts, reaches, net, ids, lc.data_cols, dv, q0, ql, assume_short_ts=True)
    args[3] = [(r, 1 if r[0] in lakeset else 0) for r in reaches]
    args[10], args[11], args[13], args[14] = lakes.tolist(), wbody_cols, np.ones((len(lakes), 1), np.int32), False
    want = compute_network_structured(*args)
    assert np.array_equal(res[0][1].view(np.uint32), want[1].view(np.uint32))
    assert np.array_equal(res[0][6], want[6])
