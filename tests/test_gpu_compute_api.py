"""compute_nhd_routing_v02 drop-in on the GPU: DataFrame in, the reference's results list out."""
import numpy as np
import pandas as pd
import pytest

import helpers as H
from oracle import oracle as O
from troute_amd import nhd_network as nn
from troute_amd.routing.compute import _compute_func_map, compute_nhd_routing_v02, new_q0

pytestmark = pytest.mark.gpu


def frames(ids, params9, q0, qlat):
    cols = ["dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"]
    param_df = pd.DataFrame(params9[:, 1:], index=ids, columns=cols[1:])
    param_df["alt"] = 0.0
    param_df["musk"] = 3600.0                              # extra columns the reference table carries
    q0_df = pd.DataFrame(q0, index=ids, columns=["qu0", "qd0", "h0"])
    qlat_df = pd.DataFrame(qlat, index=ids)
    return param_df, q0_df, qlat_df


def call(conn, param_df, q0_df, qlat_df, nts, qts, short, method="by-network", interorder=None, **kw):
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    e = pd.DataFrame()
    sub_in = [{}, {}]
    out = compute_nhd_routing_v02(
        conn, rconn, {}, reaches_bytw, "V02-structured", method, 10000, 4, None, 300.0, nts, qts, ind,
        param_df, q0_df, qlat_df, e, e, e, e, e, e, e, e, e, e, e, {}, short, False, e, {}, e, False,
        sub_in, {} if interorder is None else interorder, **kw)
    # what the reference returns (compute.py:1738) and how nwm_route unpacks it (nwm_routing/__main__.py:1256-1257)
    assert isinstance(out, tuple) and len(out) == 2
    subnetwork_list = out[1]
    results = out[0]
    assert subnetwork_list is sub_in
    return results, reaches_bytw, ind


@pytest.mark.parametrize("short", [True, False])
def test_lowercolorado_through_compute_nhd_routing_v02(short):
    lc = H.LowerColorado()
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    rng = np.random.default_rng(0)
    shuffle = rng.permutation(lc.nseg)                     # the caller's table need not be sorted
    param_df, q0_df, qlat_df = frames(lc.ids[shuffle], lc.params9[shuffle], lc.q0[shuffle], lc.qlat[shuffle])
    results, reaches_bytw, ind = call(conn, param_df, q0_df, qlat_df, lc.nts, lc.qts, short)
    assert len(results) == 1 and len(results[0]) == 10
    ids, fvd = results[0][0], results[0][1]
    assert np.array_equal(ids, lc.ids) and fvd.shape == (lc.nseg, lc.nts * 3)
    reaches, ups = lc.row_lists()
    want = O.network(lc.nts, lc.qts, reaches, ups, lc.params9, lc.q0, lc.qlat, short, det=True)[:, 1:, :]
    assert np.array_equal(fvd.reshape(lc.nseg, lc.nts, 3).view(np.uint32), want.view(np.uint32))
    nq0 = new_q0(results)
    assert np.array_equal(nq0.values, want[:, -1, :][:, [0, 0, 2]])


@pytest.mark.parametrize("short,engine", [(True, "flow"), (True, "levels"), (False, "flow"), (False, "levels")])
def test_output_stride_is_the_full_result_sliced(short, engine, monkeypatch):
    """``output_stride`` (keyword-only extension of the drop-in): every n-th timestep, decimated on the device -- what the
    reference's writers keep of a window (nwm_routing/output.py:209-216, nhd_io.py:2379-2382) -- equals slicing the full
    ``flowveldepth`` bit for bit for every n that divides nts -- then the last kept step is the window's last and ``new_q0`` is
    the same; an n that does not divide nts is refused (the next window would silently restart from an earlier step)."""
    monkeypatch.setenv("TRMC_ENGINE", engine)
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
    lc = H.LowerColorado()
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    param_df, q0_df, qlat_df = frames(lc.ids, lc.params9, lc.q0, lc.qlat)
    nts = 96
    full, _, _ = call(conn, param_df, q0_df, qlat_df, nts, lc.qts, short)
    want = full[0][1].reshape(lc.nseg, nts, 3)
    for n in (12, 8, 96, 1):
        got, _, _ = call(conn, param_df, q0_df, qlat_df, nts, lc.qts, short, output_stride=n)
        fvd = got[0][1]
        assert fvd.shape == (lc.nseg, (nts // n) * 3) and fvd.dtype == np.float32
        assert np.array_equal(fvd.reshape(lc.nseg, nts // n, 3).view(np.uint32), want[:, n - 1::n, :].view(np.uint32)), n
        assert np.array_equal(got[0][0], full[0][0]) and np.array_equal(got[0][8], full[0][8])
    got, _, _ = call(conn, param_df, q0_df, qlat_df, nts, lc.qts, short, output_stride=12)
    assert np.array_equal(new_q0(got).values, new_q0(full).values)
    with pytest.raises(ValueError, match="output_stride"):
        call(conn, param_df, q0_df, qlat_df, nts, lc.qts, short, output_stride=0)
    for n in (7, 100):
        with pytest.raises(ValueError, match="must divide nsteps"):
            call(conn, param_df, q0_df, qlat_df, nts, lc.qts, short, output_stride=n)


def test_many_networks_one_plan_results_per_tailwater():
    """All independent networks of the call are routed by ONE plan; the result list still has one
    tuple per tailwater, each equal to routing that network alone."""
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    ids = np.array(sorted(conn), np.int64)
    rng = np.random.default_rng(4)
    n = len(ids)
    p = np.stack([np.full(n, 300.0), rng.uniform(300, 3000, n), rng.uniform(1, 9, n), np.zeros(n), np.zeros(n),
                  np.full(n, 0.06), np.full(n, 0.12), rng.uniform(0.2, 1.5, n), rng.uniform(1e-3, 2e-2, n)], 1)
    p[:, 3] = p[:, 2] * 5 / 3
    p[:, 4] = 3 * p[:, 3]
    p = p.astype(np.float32)
    qlat = rng.uniform(0, 0.4, (n, 3)).astype(np.float32)
    q0 = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    param_df, q0_df, qlat_df = frames(ids, p, q0, qlat)
    for method in ("serial", "by-subnetwork-jit-clustered"):
        results, reaches_bytw, ind = call(conn, param_df, q0_df, qlat_df, 24, 12, False, method)
        assert [sorted(r[0].tolist()) for r in results] == [sorted(ind[tw]) for tw in reaches_bytw]
        row = {int(s): i for i, s in enumerate(ids)}
        for r, tw in zip(results, reaches_bytw):
            sel = np.array([row[int(s)] for s in r[0]])
            loc = {int(s): i for i, s in enumerate(r[0])}
            rl = [np.array([loc[s] for s in reach]) for reach in reaches_bytw[tw]]
            ul = [np.array([loc[s] for s in ind[tw].get(reach[0], [])], dtype=np.int64) for reach in reaches_bytw[tw]]
            want = O.network(24, 12, rl, ul, p[sel], q0[sel], qlat[sel], False, det=True)[:, 1:, :]
            assert np.array_equal(r[1].reshape(len(sel), 24, 3).view(np.uint32), want.view(np.uint32))


def test_plugin_seam_and_unsupported_inputs():
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured
    assert _compute_func_map["V02-structured"] is compute_network_structured
    assert _compute_func_map["anything-else"] is compute_network_structured        # defaultdict, compute.py:21
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    ids = np.array(sorted(conn), np.int64)
    n = len(ids)
    p = np.ones((n, 9), np.float32)
    param_df, q0_df, qlat_df = frames(ids, p, np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32))
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    e = pd.DataFrame()
    usgs = pd.DataFrame({"0": [1.0]}, index=[14])
    with pytest.raises(NotImplementedError, match="reservoir_usgs_df"):
        compute_nhd_routing_v02(conn, rconn, {}, reaches_bytw, "V02-structured", "serial", 1, 1, None, 300.0, 12, 12,
                                ind, param_df, q0_df, qlat_df, e, e, usgs, e, e, e, e, e, e, e, e, {}, True, False, e,
                                {}, e, False, [{}, {}])
    with pytest.raises(ValueError, match="Number of columns"):
        compute_nhd_routing_v02(conn, rconn, {}, reaches_bytw, "V02-structured", "serial", 1, 1, None, 300.0, 48, 12,
                                ind, param_df, q0_df, qlat_df, e, e, e, e, e, e, e, e, e, e, e, {}, True, False, e,
                                {}, e, False, [{}, {}])


def test_flowveldepth_interorder_feeds_the_call_like_upstream_results():
    """The caller's flowveldepth_interorder (reference: the "bmi" method, compute.py:1588-1589,:1649-1655,:1729-1732):
    route the upper part of a network, hand its tailwater's flowveldepth row to a second call that routes the rest, and
    get the rows a single call over the whole network gives -- bit for bit -- with the handed-over segment left out."""
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    ids = np.array(sorted(conn), np.int64)
    rng = np.random.default_rng(8)
    n = len(ids)
    p = np.stack([np.full(n, 300.0), rng.uniform(300, 3000, n), rng.uniform(1, 9, n), np.zeros(n), np.zeros(n),
                  np.full(n, 0.06), np.full(n, 0.12), rng.uniform(0.2, 1.5, n), rng.uniform(1e-3, 2e-2, n)], 1)
    p[:, 3] = p[:, 2] * 5 / 3
    p[:, 4] = 3 * p[:, 3]
    p = p.astype(np.float32)
    qlat = rng.uniform(0, 0.4, (n, 3)).astype(np.float32)
    q0 = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    param_df, q0_df, qlat_df = frames(ids, p, q0, qlat)
    nts, qts = 24, 12
    whole, reaches_bytw, ind = call(conn, param_df, q0_df, qlat_df, nts, qts, True)
    whole_rows = {int(s): r[1][i] for r in whole for i, s in enumerate(r[0])}
    # cut at a segment that has upstream segments and a downstream one
    rconn = nn.reverse_network(conn)
    cut = next(s for s in sorted(conn) if rconn.get(s) and conn[s])
    upper = nn.reachable(rconn, [cut])[cut]                 # everything that drains through `cut`
    conn_up = {s: ([d for d in conn[s] if d in upper] if s != cut else []) for s in upper}
    res_up, _, _ = call(conn_up, param_df, q0_df, qlat_df, nts, qts, True)
    rows_up = {int(s): r[1][i] for r in res_up for i, s in enumerate(r[0])}
    assert np.array_equal(rows_up[cut].view(np.uint32), whole_rows[cut].view(np.uint32))
    lower = set(conn) - upper
    conn_lo = {s: conn[s] for s in lower}
    ind_lo, reaches_lo, rconn_lo = nn.organize_independent_networks(conn_lo)
    tw = next(t for t in reaches_lo if conn[cut][0] in ind_lo[t])
    ind_lo[tw] = dict(ind_lo[tw])
    whole_tw = next(t for t in ind if conn[cut][0] in ind[t])
    ind_lo[tw][conn[cut][0]] = list(ind[whole_tw][conn[cut][0]])   # the hand-over edge, in the junction's summation order
    e = pd.DataFrame()
    inter = {cut: {"results": rows_up[cut]}}
    res_lo, sub = compute_nhd_routing_v02(
        conn_lo, rconn_lo, {}, reaches_lo, "V02-structured", "serial", 10000, 4, None, 300.0, nts, qts, ind_lo,
        param_df, q0_df, qlat_df, e, e, e, e, e, e, e, e, e, e, e, {}, True, False, e, {}, e, False, [{}, {}], inter)
    got = {int(s): r[1][i] for r in res_lo for i, s in enumerate(r[0])}
    assert cut not in got and set(got) == lower
    for s in lower:
        assert np.array_equal(got[s].view(np.uint32), whole_rows[s].view(np.uint32)), s


def test_upstream_results_of_a_waterbody_seed_its_outflow_not_the_nan_initial_condition():
    """An off-network upstream row that is a lake: the reference seeds its time-0 flow from the waterbody table's qd0
    column (mc_reach.pyx:463-465) -- the initial_conditions row of a lake id is NaN in its by-subnetwork driver."""
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
    rng = np.random.default_rng(1)
    ids = np.array([10, 20, 30, 40], np.int64)            # 10 (lake, off-network) -> 20 -> 30 -> 40
    n = 4
    dv = np.stack([np.full(n, 300.0), rng.uniform(300, 3000, n), rng.uniform(1, 9, n), rng.uniform(10, 20, n),
                   rng.uniform(30, 60, n), np.full(n, 0.06), np.full(n, 0.12), rng.uniform(0.2, 1.5, n),
                   rng.uniform(1e-3, 2e-2, n)], 1).astype(np.float32)
    cols = np.array(["dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"], dtype=object)
    nts, qts = 12, 12
    qlat = rng.uniform(0, 0.4, (n, 1)).astype(np.float32)
    q0 = rng.uniform(0.1, 1, (n, 3)).astype(np.float32)
    q0[0] = np.nan                                          # what q0_sub.reindex(...) leaves for a lake id
    hyd = rng.uniform(0.5, 2.0, (nts, 3)).astype(np.float32)
    wbody = np.zeros((1, 11))
    wbody[0, 9] = 1.75                                      # qd0
    args = mc_only_args(nts, 300.0, qts, [[20, 30, 40]], {20: [10], 30: [20], 40: [30]}, ids, cols, dv, q0, qlat,
                        upstream_results={10: {"results": hyd.reshape(-1), "position_index": 0}}, assume_short_ts=True)
    args[10], args[11] = [10], wbody
    r = compute_network_structured(*args)
    assert np.array_equal(r[0], ids[1:]) and np.isfinite(r[1]).all()
    # oracle: the lake row prefilled with its hydrograph, its time-0 flow = qd0, depth slot 0 (mc_reach.pyx:458-465)
    q0o = q0.copy()
    q0o[0] = (1.75, 0, 0)
    init = np.zeros((n, nts + 1, 3), np.float32)
    init[0, 1:, :] = hyd
    want = O.network(nts, qts, [np.array([1, 2, 3])], [np.array([0])], dv, q0o, qlat, True, det=True,
                     prefilled=np.array([1, 0, 0, 0], np.uint8), fvd_init=init)[1:, 1:, :]
    assert np.array_equal(r[1].reshape(3, nts, 3).view(np.uint32), np.ascontiguousarray(want).view(np.uint32))


def test_kernel_callable_reuses_and_tunes_its_plan_without_changing_results(monkeypatch):
    """The drop-in callable keeps the plan of a network it has routed (by content) -- the second call rebuilds it once with
    the first window's iteration costs as the row-order hint, later calls reuse it -- and every call returns the bits a
    fresh, untuned plan returns, also for another window's forcing and state."""
    from troute_amd.routing.fast_reach import mc_reach as M
    lc = H.LowerColorado()
    rng = np.random.default_rng(3)
    windows = [(lc.qlat, lc.q0),
               (lc.qlat * rng.uniform(0.5, 1.5, lc.qlat.shape).astype(np.float32), lc.q0),
               (lc.qlat, np.abs(rng.normal(1.0, 0.5, lc.q0.shape)).astype(np.float32))]
    nts = 48

    def call(ql, q0):
        args = M.mc_only_args(nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols, lc.data_values, q0, ql,
                              assume_short_ts=True)
        return M.compute_network_structured(*args)

    monkeypatch.setenv("TRMC_PLAN_CACHE", "0")
    want = [call(ql, q0)[1].copy() for ql, q0 in windows]
    monkeypatch.setenv("TRMC_PLAN_CACHE", "2")
    M._PLANS.clear()
    stages = []
    for k in (0, 1, 2, 0, 1):                                   # untuned, tuning rebuild, tuned, tuned, tuned
        got = call(*windows[k])[1]
        assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), k
        stages.append([e["stage"] for e in M._PLANS._d.values()])
    assert stages == [[1], [2], [2], [2], [2]]
    # ... and when the windows on the tuned plan have become slower than they were (RetunePolicy; here: told so), the next
    # window collects costs again and the call after it routes on a plan rebuilt from them -- the same bits throughout
    verdicts = iter([False, True, False])
    monkeypatch.setattr(M.RetunePolicy, "window", lambda self, ms: next(verdicts))
    stages, plans = [], []
    for k in (2, 0, 1, 2, 0, 1):                               # tuned; tuned, asked; collecting; rebuilt; tuned; tuned
        got = call(*windows[k])[1]
        assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), k
        (e,) = M._PLANS._d.values()
        stages.append(e["stage"])
        plans.append(id(e["plan"]))
    assert stages == [2, 0, 1, 2, 2, 2]
    assert plans[0] == plans[1] == plans[2] and plans[3] != plans[2] and plans[3] == plans[4] == plans[5]
    assert e["policy"].retunes == 1 and e["fresh"] == 0        # (judged by the second window on the new order)
    monkeypatch.setenv("TRMC_RETUNE", "0")                     # switched off: a plan is tuned once and kept
    M._PLANS.clear()
    for k in (0, 1, 2):
        call(*windows[k])
    (e,) = M._PLANS._d.values()
    assert e["stage"] == 2 and e["policy"] is None
    M._PLANS.clear()
    assert not M._PLANS._d


@pytest.mark.parametrize("short", [True, False])
def test_result_order_is_the_result_permuted(short):
    """``result_order`` (keyword-only extension of compute_network_structured; trmc_download_fvd_rowset): ids, flowveldepth and
    the upstream series in a row order of the caller's choice, permuted on the device as the result is decimated -- equal to
    indexing the reference-ordered result, for the full result and a decimated one; anything but a permutation is refused."""
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
    lc = H.LowerColorado()
    nts = 48
    args = mc_only_args(nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols, lc.data_values, lc.q0, lc.qlat, None, short)
    full = compute_network_structured(*args)
    perm = np.random.default_rng(5).permutation(lc.nseg)
    for n in (1, 12):
        got = compute_network_structured(*args, output_stride=n, result_order=perm)
        assert np.array_equal(got[0], full[0][perm])
        want = full[1].reshape(lc.nseg, nts, 3)[perm][:, n - 1::n, :]
        assert np.array_equal(got[1].reshape(lc.nseg, nts // n, 3).view(np.uint32), want.view(np.uint32)), n
        assert got[6].shape == full[6].shape
    with pytest.raises(ValueError, match="permutation"):
        compute_network_structured(*args, result_order=np.zeros(lc.nseg, dtype=np.int64))
    with pytest.raises(ValueError, match="permutation"):
        compute_network_structured(*args, result_order=perm[:-1])


def test_nan_is_zero_scrubs_the_upload_on_the_device():
    """``nan_is_zero`` (keyword-only extension; trmc_plan_set_nan_is_zero): NaN in the forcing and the initial conditions -- what a
    reindexed frame holds on rows the caller's table lacks (compute.py:1466-1467) -- counts as 0, replaced on the device behind the
    upload: the same result as zeros put there on the host; without the option the values reach the kernels as they are."""
    from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
    lc = H.LowerColorado()
    nts = 24
    rng = np.random.default_rng(3)
    hit = rng.choice(lc.nseg, 40, replace=False)
    ql_nan, q0_nan = lc.qlat.copy(), lc.q0.copy()
    ql_nan[hit[:25]] = np.nan
    q0_nan[hit[20:]] = np.nan
    ql_zero, q0_zero = np.nan_to_num(ql_nan, nan=0.0), np.nan_to_num(q0_nan, nan=0.0)
    a_nan = mc_only_args(nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols, lc.data_values, q0_nan, ql_nan, None, True)
    a_zero = mc_only_args(nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols, lc.data_values, q0_zero, ql_zero, None, True)
    want = compute_network_structured(*a_zero)[1]
    got = compute_network_structured(*a_nan, nan_is_zero=True)[1]
    assert np.isfinite(got).all() and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    raw = compute_network_structured(*a_nan)[1]            # (the plan is the same cached one: the policy is set per call)
    assert np.isnan(raw).any()
    again = compute_network_structured(*a_zero, nan_is_zero=True)[1]
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
