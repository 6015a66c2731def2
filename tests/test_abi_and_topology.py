"""CPU-side checks of the C-ABI library: it loads, exports what include/trmc.h declares, refuses to
compute without a GPU, and its native topology flattening is right.  No GPU compute here."""
import os
import re

import numpy as np
import pytest

import helpers as H
from troute_amd import _lib
from troute_amd.plan import RoutingPlan, csr_from_lists, segments, topology_clusters, topology_levels
from troute_amd.routing.fast_reach.mc_reach import _flatten_network, binary_find, column_mapper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "trmc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(trmc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 16
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in trmc.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with trmc.h"
    assert lib.trmc_abi_version() == 19
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "trdw.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(trdw_[a-z0-9_]+)\s*\(", hdr))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in trdw.h but not exported"
    assert declared == set(_lib.SIGNATURES_DW), "ctypes signature table out of sync with trdw.h"


@pytest.mark.skipif(_lib.device_count() > 0, reason="CPU-only check")
def test_no_cpu_fallback():
    """Without a HIP device every computing entry point fails loudly."""
    up_ptr, up_idx = csr_from_lists([[], [0]])
    params = np.ones((2, 9), np.float32)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        RoutingPlan(up_ptr, up_idx, params)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        segments(np.ones((1, 15), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "t-route_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"
                assert "libmc_oracle" not in src, f"{fn} loads the oracle library"
                assert not re.search(r'#\s*include\s*"[^"]*oracle', src), f"{fn} includes oracle code"
    # and the built product library links nothing from it
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in needed


# ---- topology flattening (native, host only) ---------------------------------------------------------
def check_levels(up_ptr, up_idx, lvl, pos, nl, boundary=None):
    n = up_ptr.shape[0] - 1
    b = np.zeros(n, bool) if boundary is None else np.asarray(boundary, bool)
    assert sorted(pos.tolist()) == list(range(n))                      # a permutation
    assert (lvl[b] == -1).all() and (lvl[~b] >= 0).all()
    assert lvl.max() + 1 == nl
    for r in range(n):
        if b[r]:
            continue
        ups = up_idx[up_ptr[r]:up_ptr[r + 1]]
        ups = ups[~b[ups]]
        want = 0 if ups.size == 0 else lvl[ups].max() + 1            # longest path from a headwater
        assert lvl[r] == want
    # every level is one contiguous slice of the plan order, boundary rows first
    order = np.argsort(pos)
    assert (np.diff(lvl[order]) >= 0).all()


def test_levels_toy_network():
    toy = H.load_toy()
    rconn = {int(k): v for k, v in toy["expected_rconn"].items()}
    ids = np.array(sorted(rconn), dtype=np.int64)
    row = {int(s): i for i, s in enumerate(ids)}
    up_ptr, up_idx = csr_from_lists([[row[u] for u in rconn[int(s)]] for s in ids])
    lvl, pos, nl = topology_levels(up_ptr, up_idx)
    check_levels(up_ptr, up_idx, lvl, pos, nl)
    # the long chain 20->19->18->17->16->...->8 of test_nhd_network.py:2-35
    assert lvl[row[20]] == 0 and lvl[row[8]] == 12 and nl == 13
    assert lvl[row[2800]] == 0          # isolated segment


def test_levels_lowercolorado_match_survey_probe():
    lc = H.LowerColorado()
    up_ptr, up_idx, in_reach = _flatten_network([(r, 0) for r in lc.reaches], lc.rconn, lc.ids)
    assert in_reach.all()
    lvl, pos, nl = topology_levels(up_ptr, up_idx)
    assert nl == 649                                                   # longest segment path
    fan = np.bincount(np.diff(up_ptr))
    assert fan[3] == 7                                                 # junction fan-in {2: 3856, 3: 7}
    lv = lvl.astype(np.int64)
    has = up_ptr[1:] > up_ptr[:-1]
    mx = np.maximum.reduceat(lv[up_idx], up_ptr[:-1][has])
    assert np.array_equal(lv[has], mx + 1) and (lv[~has] == 0).all()
    order = np.argsort(pos)
    assert (np.diff(lvl[order]) >= 0).all()


def test_levels_with_boundary_rows():
    ups = [[], [0], [1], [2, 5], [], [4]]
    up_ptr, up_idx = csr_from_lists(ups)
    b = np.array([0, 0, 1, 0, 0, 0], np.uint8)                          # row 2 carries a prescribed hydrograph
    lvl, pos, nl = topology_levels(up_ptr, up_idx, b)
    check_levels(up_ptr, up_idx, lvl, pos, nl, b)
    assert lvl.tolist() == [0, 1, -1, 2, 0, 1] and pos[2] == 0


def test_cycle_and_bad_input_raise_valueerror():
    up_ptr, up_idx = csr_from_lists([[1], [0]])
    with pytest.raises(ValueError, match="cycle"):
        topology_levels(up_ptr, up_idx)
    up_ptr, up_idx = csr_from_lists([[0]])
    with pytest.raises(ValueError):
        topology_levels(up_ptr, up_idx)
    with pytest.raises(ValueError, match="not a row"):
        topology_levels(np.array([0, 1], np.int64), np.array([7], np.int64))
    lvl, pos, nl = topology_levels(np.array([0], np.int64), np.zeros(0, np.int64))      # empty network
    assert nl == 0 and lvl.size == 0


def test_levels_random_forest_large():
    rng = np.random.default_rng(3)
    to = H.random_network(rng, 20000)
    reaches, heads_up, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    lvl, pos, nl = topology_levels(up_ptr, up_idx)
    lv = lvl.astype(np.int64)
    has = up_ptr[1:] > up_ptr[:-1]
    mx = np.maximum.reduceat(lv[up_idx], up_ptr[:-1][has])
    assert np.array_equal(lv[has], mx + 1) and (lv[~has] == 0).all()
    assert sorted(pos.tolist()) == list(range(20000))


def test_plan_order_inside_levels_follows_downstream_rows_and_cost_hint():
    """Order inside a level: by the position of the row flowed into (upstream gathers of consecutive rows are
    consecutive); with a cost hint, descending hint first.  Levels themselves and the set of positions never change."""
    rng = np.random.default_rng(5)
    n = 60000
    to = H.random_network(rng, n)
    reaches, heads_up, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    lvl, pos, nl = topology_levels(up_ptr, up_idx)
    # rows of one level that flow into rows of the next level up sit in the order of those rows
    order = np.argsort(pos)
    dn = np.where(to >= 0, pos[np.maximum(to, 0)], np.iinfo(np.int64).max)
    for l in range(nl - 1):
        rows = order[lvl[order] == l]
        d = dn[rows]
        assert (np.diff(d) >= 0).all(), l
    hint = rng.integers(0, 4, n).astype(np.uint8)
    lvl2, pos2, nl2 = topology_levels(up_ptr, up_idx, cost_hint=hint)
    assert nl2 == nl and np.array_equal(lvl2, lvl) and sorted(pos2.tolist()) == list(range(n))
    # level slices are the same position ranges; inside level 0 (wide) runs of 128 positions are uniform in hint
    for l in range(nl):
        a, b = np.sort(pos[lvl == l]), np.sort(pos2[lvl == l])
        assert np.array_equal(a, b)
    order2 = np.argsort(pos2)
    dn2 = np.where(to >= 0, pos2[np.maximum(to, 0)], np.iinfo(np.int64).max)
    for l in range(nl):
        rows = order2[lvl[order2] == l]
        h = hint[rows].astype(np.int64)
        assert (np.diff(h) <= 0).all(), l                       # costly rows first
        same = np.diff(h) == 0
        assert (np.diff(dn2[rows])[same] >= 0).all(), l          # and inside one cost the downstream order


# ---- host shims of the reference's helpers -----------------------------------------------------------
def test_binary_find_and_column_mapper():
    arr = np.array([3, 5, 9, 12], np.int64)
    assert binary_find(arr, [9, 3]) == [2, 0]
    with pytest.raises(ValueError, match="not found"):
        binary_find(arr, [4])
    with pytest.raises(ValueError, match="not found"):
        binary_find(arr, [99])
    assert column_mapper(H.DATA_COLS) == [0, 4, 1, 2, 3, 5, 6, 7, 8]   # mc_reach.pyx:150-162


def test_flatten_network_matches_reference_structs():
    """Upstream lists: head of a reach <- upstream_connections[reach[0]] in dict order; inside a reach
    <- the previous segment (mc_reach.pyx:288-289, :133-138)."""
    toy = H.load_toy()
    for tw, reaches in toy["reaches_bytw"].items():
        net = {int(k): v for k, v in toy["independent_networks"][tw].items()}
        ids = np.array(sorted(net), dtype=np.int64)
        row = {int(s): i for i, s in enumerate(ids)}
        up_ptr, up_idx, in_reach = _flatten_network([(r, 0) for r in reaches], net, ids)
        assert in_reach.all()
        for r in reaches:
            assert up_idx[up_ptr[row[r[0]]]:up_ptr[row[r[0]] + 1]].tolist() == [row[u] for u in net[r[0]]]
            for a, b in zip(r[:-1], r[1:]):
                assert up_idx[up_ptr[row[b]]:up_ptr[row[b] + 1]].tolist() == [row[a]]
    # a reservoir reach is the single waterbody node (mc_reach.pyx:293)
    up_ptr, up_idx, in_reach = _flatten_network([([1], 0), ([7], 1)], {7: [1]}, np.array([1, 7], np.int64))
    assert up_idx[up_ptr[1]:up_ptr[2]].tolist() == [0] and in_reach.all()
    with pytest.raises(ValueError, match="single waterbody node"):
        _flatten_network([([1, 7], 1)], {}, np.array([1, 7], np.int64))


# ---- block order of the dataflow engine (host only) -------------------------------------------------------
def _check_block_order(up_ptr, up_idx, boundary, hint, tiers):
    from troute_amd.plan import topology_blocks, topology_levels
    n = len(up_ptr) - 1
    pos, rank, B, nb = topology_blocks(up_ptr, up_idx, boundary, hint, tiers)
    lvl, _, _ = topology_levels(up_ptr, up_idx, boundary)
    assert sorted(pos) == list(range(n))                              # a permutation
    nbound = 0 if boundary is None else int(np.count_nonzero(boundary))
    if nbound:
        b_rows = np.flatnonzero(boundary)
        assert np.array_equal(np.sort(pos[b_rows]), np.arange(nbound))     # boundary rows first, ascending row order
        assert np.array_equal(pos[b_rows], np.arange(nbound))
    assert nb == -(-(n - nbound) // B)
    blk = (pos - nbound) // B
    for r in range(n):
        if boundary is not None and boundary[r]:
            continue
        for u in up_idx[up_ptr[r]:up_ptr[r + 1]]:
            if boundary is not None and boundary[u]:
                continue
            assert blk[u] <= blk[r], "a row needs a LATER block"
            if blk[u] == blk[r]:
                assert rank[u] < rank[r], "ranks must grow along the edges inside a block"
    assert (rank[lvl <= 0] == 0).all()                                # headwaters and boundary rows trail nothing
    return pos, rank, B


def test_block_order_is_a_valid_dataflow_order():
    rng = np.random.default_rng(5)
    n = 3000
    to = np.full(n, -1, np.int64)
    for i in range(n):                                                  # random forest, rows in random order
        if rng.random() > 0.003:
            cand = rng.integers(i + 1, min(n, i + 40)) if i + 1 < n else -1
            to[i] = cand
    perm = rng.permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    to_p = np.full(n, -1, np.int64)
    to_p[inv] = np.where(to >= 0, inv[np.maximum(to, 0)], -1)           # relabelled
    ups = [[] for _ in range(n)]
    for r in range(n):
        if to_p[r] >= 0:
            ups[to_p[r]].append(r)
    up_ptr, up_idx = csr_from_lists(ups)
    hint = rng.integers(0, 49, n).astype(np.uint8)
    for boundary in (None, (rng.random(n) < 0.01).astype(np.uint8)):
        if boundary is not None:                                        # boundary rows have no routed upstream of their own here
            up2 = [([] if boundary[r] else ups[r]) for r in range(n)]
            bp, bi = csr_from_lists(up2)
        else:
            bp, bi = up_ptr, up_idx
        for h in (None, hint):
            for tiers in (True, False):
                _check_block_order(bp, bi, boundary, h, tiers)


def _check_cluster_order(up_ptr, up_idx, boundary, hint, wide_min_rows, cluster_rows):
    nseg = up_ptr.shape[0] - 1
    lvl, _, nlevels = topology_levels(up_ptr, up_idx, boundary)
    pos, lag, blk, W, C, nb = topology_clusters(up_ptr, up_idx, boundary, hint, wide_min_rows=wide_min_rows,
                                                wide_max_levels=16, cluster_rows=cluster_rows)
    assert np.array_equal(np.sort(pos), np.arange(nseg))                 # a permutation of the rows
    routed = lvl >= 0
    assert np.all(lag[~routed] == -1) and np.all(blk[~routed] == -1)
    assert np.all(pos[~routed] < (~routed).sum())                        # boundary rows first
    # the leading slices are the levels themselves, in level order, each with at least wide_min_rows rows
    width = np.bincount(lvl[routed], minlength=max(nlevels, 1))
    want_w = 0
    while want_w < min(nlevels, 16) and wide_min_rows > 0 and width[want_w] >= wide_min_rows:
        want_w += 1
    assert W == want_w
    sl = routed & (lvl < W)
    assert np.array_equal(lag[sl], lvl[sl]) and np.all(blk[sl] == -1)
    if W > 0 and sl.any():
        assert np.all(np.diff(lvl[sl][np.argsort(pos[sl])]) >= 0)
        assert pos[sl].max() < pos[routed & (lvl >= W)].min(initial=nseg)
    cl = routed & (lvl >= W)
    assert np.all(blk[cl] >= 0) and np.all(lag[cl] >= W)
    if cl.any():
        assert C == lag[cl].max() - W + 1 and nb == blk[cl].max() + 1
        # a block: consecutive positions, at most cluster_rows of them, one lag; blocks ordered by lag
        order = np.argsort(pos[cl])
        b_sorted, l_sorted = blk[cl][order], lag[cl][order]
        assert np.all(np.diff(b_sorted) >= 0) and np.all(np.diff(l_sorted) >= 0)
        assert np.bincount(b_sorted).max() <= cluster_rows
        first = np.r_[0, np.flatnonzero(np.diff(b_sorted)) + 1]
        assert np.all(np.maximum.reduceat(l_sorted, first) == np.minimum.reduceat(l_sorted, first))
    else:
        assert C == 0 and nb == 0
    # THE promise: a row reads only rows of its own block (same lag: they advance together) or rows that run ahead of it
    down = np.repeat(np.arange(nseg), np.diff(up_ptr))
    keep = routed[up_idx] & routed[down]
    u, d = up_idx[keep], down[keep]
    same = (blk[u] == blk[d]) & (blk[d] >= 0)
    assert np.all(lag[u][same] == lag[d][same])
    assert np.all(lag[u][~same] < lag[d][~same])
    return W, C, nb


def test_cluster_order_is_a_valid_skewed_order():
    """trmc_plan_options.cluster_rows (csrc/topology.hpp): the rows below the wide levels in clusters -- what k_mc_ctile relies on"""
    rng = np.random.default_rng(77)
    for nseg in (1, 2, 65, 700, 5000, 30000):
        to = H.random_network(rng, nseg)
        _, _, ups = H.reaches_from_to(to)
        up_ptr, up_idx = csr_from_lists(ups)
        for wide, rows in ((0, 128), (40, 128), (8, 16), (0, 1), (300, 5)):
            hint = rng.integers(0, 4, nseg).astype(np.uint8) if wide == 40 else None
            _check_cluster_order(up_ptr, up_idx, None, hint, wide, rows)
    # with boundary rows (prescribed hydrographs: they constrain nothing) and the LowerColorado network
    lc = H.LowerColorado()
    up_ptr, up_idx = lc.csr()
    b = np.zeros(lc.nseg, np.uint8)
    b[rng.choice(lc.nseg, 40, replace=False)] = 1
    _check_cluster_order(up_ptr, up_idx, b, None, 0, 128)
    W, C, nb = _check_cluster_order(up_ptr, up_idx, None, None, 0, 128)
    # 649 levels of segments (longest segment path 649) become a few cluster levels of 128 rows, blocks almost full
    lvl, _, nlevels = topology_levels(up_ptr, up_idx)
    assert nlevels == 649 and W == 0 and C <= 24 and nb <= int(np.ceil(lc.nseg / 128 * 1.05))
    with pytest.raises(ValueError):
        topology_clusters(up_ptr, up_idx, cluster_rows=129)


def test_block_order_groups_rows_by_cost_tier_and_keeps_chains_together():
    # one chain of 600 rows: post-order is the chain itself, so every block is a run of consecutive rows and the ranks
    # inside a block count up along it
    n = 600
    ups = [[]] + [[i - 1] for i in range(1, n)]
    up_ptr, up_idx = csr_from_lists(ups)
    pos, rank, B = _check_block_order(up_ptr, up_idx, None, None, False)
    for b in range(-(-n // B)):
        rows = np.flatnonzero(pos // B == b)
        assert rows.min() == b * B and rows.max() == min(n, (b + 1) * B) - 1
        assert np.array_equal(rank[rows], rows - rows.min())
    # with a hint and tiers: rows of a cheap tier never sit behind rows of a costlier one
    rng = np.random.default_rng(2)
    n = 4000
    to = np.array([rng.integers(i + 1, min(n, i + 30)) if i + 1 < n else -1 for i in range(n)])
    ups = [[] for _ in range(n)]
    for r in range(n - 1):
        ups[to[r]].append(r)
    up_ptr, up_idx = csr_from_lists(ups)
    hint = np.where(np.arange(n) > 3000, 48, 16).astype(np.uint8)      # the downstream end is costly
    pos, _, B = _check_block_order(up_ptr, up_idx, None, hint, True)
    blocks_costly = np.unique(pos[hint == 48] // B)
    blocks_cheap_only = np.setdiff1d(np.unique(pos[hint == 16] // B), blocks_costly)
    assert blocks_cheap_only.max() < blocks_costly.min()                # cheap blocks first
    assert blocks_costly.size <= -(-int((hint == 48).sum() + 1) // B) + 1


def test_retune_policy_asks_for_new_costs_when_windows_slow_down_and_backs_off_when_that_does_not_help():
    """The drop-in's plan cache re-tunes the row order when the windows on the tuned plan have become slower than they
    were (mc_reach.RetunePolicy): two consecutive windows 12 % and half a millisecond above the fastest; a re-tune that does
    not bring the time back makes the slower time the yardstick and doubles the number of windows to sit out."""
    from troute_amd.routing.fast_reach.mc_reach import RetunePolicy
    p = RetunePolicy()
    assert [p.window(ms) for ms in (16.0, 16.2, 15.9, 16.4)] == [False] * 4 and p.best == 15.9
    assert [p.window(ms) for ms in (19.0, 16.0, 19.0)] == [False] * 3             # (not consecutive)
    assert p.window(19.5) is True                                                  # the second in a row: the forcing has moved on
    p.rebuilt(16.1)                                                                # the new order serves: no sitting out
    assert p.hold == 0 and p.best == 16.1 and p.retunes == 1
    assert [p.window(ms) for ms in (16.1, 20.3, 20.4)] == [False, False, True]
    p.rebuilt(20.2)                                                                # it did not help: classes that do not persist
    assert p.hold == 8 and p.next_hold == 16 and p.best == 20.2
    assert [p.window(30.0) for _ in range(8)] == [False] * 8 and p.hold == 0       # sat out, however slow
    assert [p.window(ms) for ms in (20.3, 20.1, 22.0, 22.1)] == [False] * 4        # 12 % over the NEW yardstick is 22.5
    assert [p.window(ms) for ms in (23.0, 23.0)] == [False, True]
    p.rebuilt(22.9)
    assert p.hold == 16 and p.next_hold == 32
    # the windows of a small network are launch latency: half a millisecond of noise is no reason
    q = RetunePolicy()
    assert [q.window(ms) for ms in (0.50, 0.70, 0.72, 0.75)] == [False] * 4
    assert [q.window(ms) for ms in (0.0, -1.0, float("nan"))] == [False] * 3       # no timing: no verdict


def _check_general_order(up_ptr, up_idx, boundary, stem_min_rows):
    """validity of the general-mode block order (every upstream row in the row's block or an earlier one, ranks growing
    along the edges inside a block); returns (pos, rank, B, nb, early)"""
    from troute_amd.plan import topology_blocks_general
    n = len(up_ptr) - 1
    pos, rank, B, nb, early = topology_blocks_general(up_ptr, up_idx, boundary, stem_min_rows)
    assert sorted(pos) == list(range(n))
    nbound = 0 if boundary is None else int(np.count_nonzero(boundary))
    assert nb == -(-(n - nbound) // B)
    blk = (pos - nbound) // B
    for r in range(n):
        if boundary is not None and boundary[r]:
            continue
        for u in up_idx[up_ptr[r]:up_ptr[r + 1]]:
            if boundary is not None and boundary[u]:
                continue
            assert blk[u] <= blk[r], "a row needs a LATER block"
            if blk[u] == blk[r]:
                assert rank[u] < rank[r]
    return pos, rank, B, nb, early


def test_general_mode_block_order_puts_long_stems_last_with_their_tributaries_from_the_top_down():
    """A plan built for the general mode (a row needs its upstream rows at the SAME step, mc_reach.pyx:499-505) lays a basin
    with a long stem out as [side tributaries, the top of the stem's first][the stem], the stem's run aligned to blocks,
    and names the stem's blocks for an early start -- still a valid dataflow order."""
    rng = np.random.default_rng(8)
    L = 700                                                            # the stem: rows 0 (top) .. L - 1 (outlet)
    ups = [[] if i == 0 else [i - 1] for i in range(L)]
    side_of = {}                                                       # stem row -> rows of its side tributaries
    for i in range(1, L):
        if rng.random() < 0.4:                                         # a side tributary: a small random tree
            m = int(rng.integers(1, 30))
            base = len(ups)
            for j in range(m):
                ups.append([])
                if j:
                    ups[base + int(rng.integers(0, j))].append(base + j)   # base is the root (feeds the stem row)
            ups[i].append(base)
            side_of[i] = list(range(base, base + m))
    long_side = len(ups)                                               # one side tributary LONGER than the rest of the stem
    for j in range(120):                                               # above it is not: the stem stays the longest path
        ups.append([] if j == 0 else [len(ups) - 1])
    ups[400].append(len(ups) - 1)
    side_of.setdefault(400, []).extend(range(long_side, long_side + 120))
    n_basin = len(ups)
    for _ in range(300):                                               # small networks of one to three rows (the padding)
        k = int(rng.integers(1, 4))
        base = len(ups)
        for j in range(k):
            ups.append([] if j == 0 else [base + j - 1])
    n = len(ups)
    perm = rng.permutation(n)                                          # rows in random order
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    ups_p = [[] for _ in range(n)]
    for r in range(n):
        ups_p[inv[r]] = [int(inv[u]) for u in ups[r]]
    up_ptr, up_idx = csr_from_lists(ups_p)
    pos, rank, B, nb, early = _check_general_order(up_ptr, up_idx, None, 512)
    from troute_amd.plan import topology_levels
    lvl, _, _ = topology_levels(up_ptr, up_idx)
    outlet = int(inv[L - 1])
    drain = np.ones(n, np.int64)
    for r in np.argsort(lvl, kind="stable"):                           # upstream rows before the rows they feed
        drain[r] += drain[up_idx[up_ptr[r]:up_ptr[r + 1]]].sum()
    stem = [outlet]                                                    # the longest path into the outlet (no ties on it here)
    while True:
        us = up_idx[up_ptr[stem[-1]]:up_ptr[stem[-1] + 1]]
        if us.size == 0:
            break
        # (the highest level; among equals the one that drains most rows; among those the first listed)
        stem.append(int(max(us, key=lambda u: (lvl[u], drain[u], -list(us).index(u)))))
    stem = np.asarray(stem[::-1])                                      # top to bottom
    assert stem.size == lvl[outlet] + 1 and stem.size >= 512
    stem_pos = pos[stem]
    lo, hi = stem_pos.min(), stem_pos.max()
    assert hi - lo + 1 == stem.size                                    # one run of positions ...
    assert np.array_equal(stem_pos // B, (lo + np.arange(stem.size)) // B)     # ... block by block from the top down
    assert lo % B == 0                                                 # begins on a block boundary ...
    others_in_last_block = [r for r in range(n) if hi < pos[r] < -(-(hi + 1) // B) * B]
    assert all(perm[r] >= n_basin for r in others_in_last_block)       # ... and shares its last block with small networks only
    assert np.array_equal(early, np.arange(lo // B, hi // B + 1))
    on_stem = np.zeros(n, bool)
    on_stem[stem] = True
    last = -1                                                          # side tributaries: the top of the stem's first
    for v in stem:
        sub, todo = [], [int(u) for u in up_idx[up_ptr[v]:up_ptr[v + 1]] if not on_stem[u]]
        while todo:
            r = todo.pop()
            sub.append(r)
            todo.extend(int(u) for u in up_idx[up_ptr[r]:up_ptr[r + 1]])
        if sub:                                                        # (inside a block the rows are dealt out by size)
            p = pos[np.asarray(sub)]
            assert p.min() // B >= last and p.max() < lo
            last = p.max() // B
    # a stem shorter than asked for: the plain post-order, nothing started early; and the order is valid with boundary rows
    pos0, _, _, _, early0 = _check_general_order(up_ptr, up_idx, None, 4096)
    from troute_amd.plan import topology_blocks
    assert early0.size == 0 and np.array_equal(pos0, topology_blocks(up_ptr, up_idx, None, None, False)[0])
    boundary = np.zeros(n, np.uint8)
    boundary[inv[np.asarray(side_of[400][:1])]] = 1                    # the head of the long side tributary is prescribed
    _check_general_order(up_ptr, up_idx, boundary, 512)
    _check_general_order(up_ptr, up_idx, None, 0)
