"""Graph utilities against fixtures emitted by the reference's own code
(tests/golden/make_fixtures.py imports src/troute-network/troute/nhd_network.py) and the data
literals of the reference's unit test src/troute-network/troute/test_nhd_network.py."""
import numpy as np
import pandas as pd

import helpers as H
from troute_amd import nhd_network as nn


def toy_frame(toy):
    df = pd.DataFrame(toy["rows_key_dx_downstream_waterbody"], columns=["key", "dx", "downstream", "waterbody"])
    return df.set_index("key")


def test_build_connections_like_reference_test():
    """test_nhd_network.py:test_build_connections (terminal codes = explicit code + off-domain targets)."""
    toy = H.load_toy()
    df = toy_frame(toy)
    terminal = {toy["terminal_code"]} | set(df[~df["downstream"].isin(df.index)]["downstream"].values)
    conn = nn.extract_connections(df, "downstream", terminal)
    assert conn == {int(k): v for k, v in toy["expected_connections"].items()}


def test_reverse_network_like_reference_test():
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    rconn = nn.reverse_network(conn)
    assert rconn == {int(k): v for k, v in toy["expected_rconn"].items()}
    back = nn.reverse_network(rconn)
    assert {k: sorted(v) for k, v in back.items()} == {k: sorted(v) for k, v in conn.items()}
    assert nn.reverse_dict({1: "a", 2: "b"}) == {"a": 1, "b": 2}


def test_headwaters_tailwaters_and_independent_networks():
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    assert sorted(nn.headwaters(conn)) == toy["headwaters"]
    assert sorted(nn.tailwaters(conn)) == toy["tailwaters"]
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn, set(), set())
    want = {int(tw): {int(k): v for k, v in net.items()} for tw, net in toy["independent_networks"].items()}
    assert {tw: set(net) for tw, net in ind.items()} == {tw: set(net) for tw, net in want.items()}
    for tw in want:
        assert ind[tw] == want[tw]                         # upstream lists incl. their order


def reach_contract(reaches, rconn):
    """every node once; chain-connected upstream->downstream; upstream reaches listed earlier."""
    seen_at = {}
    for i, r in enumerate(reaches):
        for a, b in zip(r[:-1], r[1:]):
            assert rconn[b] == [a]
        for s in r:
            assert s not in seen_at
            seen_at[s] = i
    for i, r in enumerate(reaches):
        for u in rconn.get(r[0], ()):
            assert seen_at[u] < i
    return seen_at


def test_dfs_decomposition_toy_equals_reference_reach_set():
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    for tw, want in toy["reaches_bytw"].items():
        got = reaches_bytw[int(tw)]
        assert {tuple(r) for r in got} == {tuple(r) for r in want}
        reach_contract(got, ind[int(tw)])


def test_dfs_decomposition_lowercolorado_equals_reference_reach_set():
    lc = H.LowerColorado()
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    assert list(reaches_bytw) == lc.tailwaters
    got = reaches_bytw[lc.tailwaters[0]]
    assert len(got) == 7734
    assert {tuple(r) for r in got} == {tuple(r) for r in lc.reaches}
    seen = reach_contract(got, ind[lc.tailwaters[0]])
    assert len(seen) == lc.nseg
    # upstream lists carry the reference's order (ascending id for an id-sorted table)
    for s in (lc.ids[100], lc.ids[5000]):
        assert rconn[int(s)] == lc.rconn[int(s)]


def test_gage_breaks_split_reaches():
    toy = H.load_toy()
    conn = {int(k): v for k, v in toy["expected_connections"].items()}
    _, plain, _ = nn.organize_independent_networks(conn)
    _, gaged, _ = nn.organize_independent_networks(conn, set(), {14})
    assert any(14 in r and len(r) > 1 for r in plain[8])
    assert [14] in gaged[8]
    assert sum(len(r) for r in gaged[8]) == sum(len(r) for r in plain[8])


def test_build_subnetworks_contract():
    lc = H.LowerColorado()
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    rconn = nn.reverse_network(conn)
    sub = nn.build_subnetworks(conn, rconn, 1000)
    tw = lc.tailwaters[0]
    orders = sub[tw]
    allseg = [s for k in orders for piece in orders[k].values() for s in piece]
    assert len(allseg) == len(set(allseg)) == lc.nseg      # a partition
    assert list(orders[0]) == [tw]
    order_of = {s: k for k in orders for piece in orders[k].values() for s in piece}
    piece_of = {s: root for k in orders for root, piece in orders[k].items() for s in piece}
    for s, ups in rconn.items():
        for u in ups:                                       # flow only enters from the same piece or order+1
            assert piece_of[u] == piece_of[s] or order_of[u] == order_of[s] + 1
    sizes = [len(p) for k in orders for p in orders[k].values()]
    assert max(sizes) <= 1000 + 3
