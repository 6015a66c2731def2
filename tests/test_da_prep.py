"""Gage tables through the drop-in driver: _prep_da_dataframes / _prep_da_positions_byreach against vectors
made by the reference's own helpers (tests/golden/make_fixtures.py::da_prep_vectors, compute.py:49-140), and
usgs_df / lastobs_df through compute_nhd_routing_v02 on the GPU against the oracle loop with the nudging hook."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import helpers as H
from troute_amd.routing.compute import _prep_da_dataframes, _prep_da_positions_byreach

G = json.load(open(os.path.join(H.GOLDEN, "da_prep_vectors.json")))


@pytest.mark.parametrize("case", range(len(G["cases"])))
def test_da_prep_helpers_equal_reference(case):
    c = G["cases"][case]
    idx = pd.Index(G["seg_ids"])
    usgs = (pd.DataFrame(np.array(c["usgs_values"], dtype="float32"), index=c["usgs_index"])
            if c["usgs_index"] else pd.DataFrame())
    last = (pd.DataFrame(np.array(c["lastobs_values"], dtype="float32"), index=c["lastobs_index"], columns=c["lastobs_cols"])
            if c["lastobs_index"] else pd.DataFrame())
    us, ls, byseg = _prep_da_dataframes(usgs, last, idx, c["exclude"])
    assert [int(x) for x in us.index] == c["out_usgs_index"]
    assert list(us.shape) == c["out_usgs_shape"]
    if us.size:
        assert np.array_equal(np.asarray(us.values, dtype="float64"), np.array(c["out_usgs_values"]))
    assert [int(x) for x in ls.index] == c["out_lastobs_index"]
    assert [str(x) for x in ls.columns] == c["out_lastobs_cols"]
    assert [int(x) for x in byseg] == c["out_byseg"]
    byreach, bygage = _prep_da_positions_byreach(G["reaches"], ls.index)
    assert [int(x) for x in byreach] == c["out_byreach"]
    assert [int(x) for x in bygage] == c["out_bygage"]


def test_da_prep_open_loop():
    us, ls, byseg = _prep_da_dataframes(pd.DataFrame(), pd.DataFrame(), pd.Index([1, 2, 3]))
    assert us.empty and ls.empty and list(byseg) == []


@pytest.mark.gpu
@pytest.mark.parametrize("short", [True, False])
def test_gage_frames_through_compute_nhd_routing_v02(short):
    from oracle import oracle as O
    from test_gpu_compute_api import frames
    from test_nudging import gaged_lowercolorado
    from troute_amd import nhd_network as nn
    from troute_amd.routing.compute import compute_nhd_routing_v02
    lc, reaches, net, gage_ids, upos, upr, upg, usgs, lv0, lt0 = gaged_lowercolorado()
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn, set(), set(gage_ids))
    param_df, q0_df, qlat_df = frames(lc.ids, lc.params9, lc.q0, lc.qlat)
    rng = np.random.default_rng(1)
    order = rng.permutation(len(gage_ids))                     # the lastobs table decides the gage order
    g = np.array(gage_ids)[order]
    usgs_df = pd.DataFrame(usgs[order], index=g)
    lastobs_df = pd.DataFrame({"time_since_lastobs": lt0[order], "lastobs_discharge": lv0[order]}, index=g)
    e = pd.DataFrame()
    decay = 120.0
    results = compute_nhd_routing_v02(
        conn, rconn, {}, reaches_bytw, "V02-structured", "by-network", 10000, 4, None, lc.dt, lc.nts, lc.qts, ind,
        param_df, q0_df, qlat_df, usgs_df, lastobs_df, e, e, e, e, e, e, e, e, e, {"da_decay_coefficient": decay},
        short, False, e, {}, e, False, [{}, {}])[0]
    assert len(results) == 1
    r = results[0]
    row = {int(s): i for i, s in enumerate(lc.ids)}
    rl = [np.array([row[s] for s in rr], dtype=np.int64) for rr in reaches]
    ul = [np.array([row[s] for s in net.get(rr[0], [])], dtype=np.int64) for rr in reaches]
    gage_of_reach = np.full(len(reaches), -1, np.int64)
    gage_of_reach[upr] = upg
    da = dict(usgs_values=usgs, gage_row=upos, gage_of_reach=gage_of_reach, decay_coeff=decay, routing_period=lc.dt,
              lastobs_time=lt0, lastobs_val=lv0)
    want = O.network(lc.nts, lc.qts, rl, ul, lc.params9, lc.q0, lc.qlat, short, det=True, da=da)
    fvd = r[1].reshape(lc.nseg, lc.nts, 3)
    assert np.array_equal(fvd.view(np.uint32), np.ascontiguousarray(want[:, 1:, :]).view(np.uint32))
    # gage-indexed outputs come back in the lastobs table's order
    assert np.array_equal(r[3][0], g)
    assert np.array_equal(r[8].view(np.uint32), da["nudge"][order].view(np.uint32))
    assert np.array_equal(r[3][1].view(np.uint32), da["lastobs_time"][order].view(np.uint32))
    assert np.array_equal(r[3][2].view(np.uint32), da["lastobs_val"][order].view(np.uint32))
