"""Partition a river network over GPUs (one process per GPU).

Independent drainage networks share nothing (reference: reachable_network,
src/troute-network/troute/nhd_network.py:245-275; one kernel call per
tailwater, src/troute-routing/troute/routing/compute.py:1214,:1399), so they
are the natural shards.  CONUS has one basin with about half of all segments
(doc/AGU_Poster.md:211-214), which caps by-network speed-up at 2x; that basin
is therefore cut at tributary mouths into sub-basins plus a residual trunk --
the reference's ordered-subnetwork idea (build_subnetworks,
nhd_network.py:691-771; hand-off compute.py:882-897): sub-basins run first
(phase 0), their outlet hydrographs become prescribed boundary rows of the
trunk (phase 1).  No collective is needed inside a phase.
"""
import os

import numpy as np


def outlet_of(to):
    """Outlet row of every row (pointer doubling)."""
    n = to.shape[0]
    idx = np.arange(n, dtype=np.int64)
    anc = np.where(to >= 0, to, idx)
    while True:
        nxt = anc[anc]
        if np.array_equal(nxt, anc):
            return anc
        anc = nxt


def subtree_sizes(to):
    """Number of rows draining through each row (itself included)."""
    n = to.shape[0]
    idx = np.arange(n, dtype=np.int64)
    # distance to outlet, then accumulate farthest-first
    anc = np.where(to >= 0, to, idx)
    d = (to >= 0).astype(np.int64)
    while True:
        nxt = anc[anc]
        d = d + d[anc]
        if np.array_equal(nxt, anc):
            break
        anc = nxt
    size = np.ones(n, dtype=np.int64)
    order = np.argsort(-d, kind="stable")
    dd = d[order]
    cuts = np.flatnonzero(np.diff(dd)) + 1
    for grp in np.split(order, cuts):
        t = to[grp]
        ok = t >= 0
        np.add.at(size, t[ok], size[grp[ok]])
    return size


def lpt_assign(sizes, nparts, initial_load=None):
    """Longest-processing-time bin packing: part index per item (`initial_load`: what the bins hold already)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    part = np.zeros(sizes.shape[0], dtype=np.int32)
    load = np.zeros(nparts, dtype=np.int64) if initial_load is None else np.asarray(initial_load, dtype=np.int64).copy()
    for i in np.argsort(-sizes, kind="stable").tolist():
        p = int(np.argmin(load))
        part[i] = p
        load[p] += sizes[i]
    return part, load


def _trunk_share(nseg, nparts):
    """Fraction of a rank's share a trunk's owner is spared beyond the trunk itself, fitted on the ranks of 8-, 4- and 2-way
    CONUS partitions timed one by one (tools/sim_ranks.py).  Ranks on the dataflow engine (under a million rows each) route a
    window in four time chunks and the owner's skewed trunk drains for two of them: 0.3 where a rank's blocks are all
    resident (under 400 k rows, N = 8: with 0.18 the owner was the slowest rank by 10 %), 0.2 where they run in rounds
    (N = 4: with 0.3 the owner finished 0.6 ms ahead of peers that carried its rows).  0.23 on the level engine, whose
    owner also pays the draining launches of the skewed trunk at full width (N = 2)."""
    env = os.environ.get("TRMC_TRUNK_SHARE")
    if env:
        return float(env)
    per_rank = nseg / max(nparts, 1)
    return 0.23 if per_rank >= 1.0e6 else (0.2 if per_rank >= 4.0e5 else 0.3)


def partition(to, nparts, max_piece_frac=None, row_cost=None):
    """Split rows into pieces for `nparts` workers.

    row_cost: optional [nseg] measured cost of every row (e.g. ``ShardedRouter.iteration_hint()`` of a tuning window:
    the secant iterations a row needs per step, over-bank steps weighted) -- the pieces are then packed by the cost they
    carry instead of by their row counts, and a trunk weighs what its rows were measured to cost.  Without it (the first
    window of a network) rows count equally and a trunk is weighted by constants fitted on timings (below).

    Returns dict:
      piece     int32 [nseg]  piece id of every row
      phase     int32 [npieces]  0 = no upstream piece, 1 = trunk fed by phase-0 pieces
      owner     int32 [npieces]  worker that routes the piece
      cut_rows  int64 [ncut]     rows (outlets of phase-0 sub-basins) whose hydrograph is handed
                                 to the trunk they drain into
      cut_into  int64 [ncut]     trunk row each cut row flows into
      owner_bias int64 [nparts]  rows of sub-basin a worker is spared for the trunks it owns (see below)
    With nparts == 1 everything is one phase-0 piece per independent network (no cuts).
    """
    nseg = to.shape[0]
    outlet = outlet_of(to)
    uniq, piece = np.unique(outlet, return_inverse=True)
    piece = piece.astype(np.int64)
    sizes = np.bincount(piece)
    cut_rows = np.zeros(0, dtype=np.int64)
    phase = np.zeros(uniq.shape[0], dtype=np.int32)
    if nparts > 1:
        if max_piece_frac is None:
            max_piece_frac = 1.0 / (4 * nparts)
        limit = max(1, int(nseg * max_piece_frac))
        big = np.flatnonzero(sizes > 2 * limit)
        if big.size:
            sub = subtree_sizes(to)
            in_big = np.isin(piece, big)
            # trunk = rows of a big network whose own sub-tree is larger than the limit;
            # every other row of it hangs below a cut row (a maximal sub-tree <= limit)
            trunk = in_big & (sub > limit)
            down_is_trunk = np.zeros(nseg, dtype=bool)
            has = to >= 0
            down_is_trunk[has] = trunk[to[has]]
            cut = in_big & ~trunk & down_is_trunk
            cut_rows = np.flatnonzero(cut)
            # label sub-basin rows by the cut row they drain to: stop the climb at cut rows
            to_cut = to.copy()
            to_cut[cut_rows] = -1
            to_cut[trunk] = -1
            root = outlet_of(to_cut)
            sub_rows = in_big & ~trunk
            new_ids, inv = np.unique(root[sub_rows], return_inverse=True)
            nxt = uniq.shape[0]
            piece[sub_rows] = nxt + inv
            phase = np.concatenate([phase, np.zeros(new_ids.shape[0], dtype=np.int32)])
            phase[big] = 1                      # what is left of a big network is its trunk
    npieces = phase.shape[0]
    sizes = np.bincount(piece, minlength=npieces)
    owner = np.zeros(npieces, dtype=np.int32)
    if row_cost is not None and nparts > 1:
        # measured costs: a piece weighs the cost of its rows (in thousandths of the mean row, so that the integer
        # packing below keeps its resolution).  What a trunk costs its owner is NOT the arithmetic of its rows alone: they
        # are the deepest, most tightly coupled rows of the network (every step of theirs waits for the step before, with
        # little else to fill the device), and the time-skewed trunk drains for two time chunks after the rank's other
        # rows are done.  Timed rank by rank (tools/sim_ranks.py, 8-, 4- and 2-way CONUS partitions, both engines) the
        # owner needs to be spared about five times the trunk's measured cost plus a fraction of a rank's share (_trunk_share) -- with the
        # trunk weighed at its measured cost only, the owner was the slowest rank by 15-20 % at every N.
        c = np.maximum(np.asarray(row_cost, dtype=np.float64), 1.0)
        c = c * (1000.0 / c.mean())
        weight = np.bincount(piece, weights=c, minlength=npieces).astype(np.int64)
        bias = np.zeros(nparts, dtype=np.int64)
        p1 = np.flatnonzero(phase == 1)
        if p1.size:
            owner[p1], trunk_load = lpt_assign(weight[p1], nparts)
            bias = 5 * trunk_load + np.where(trunk_load > 0, int(_trunk_share(nseg, nparts) * weight.sum() / nparts), 0)
        p0 = np.flatnonzero(phase == 0)
        owner[p0], _ = lpt_assign(weight[p0], nparts, bias)
        return {
            "piece": piece.astype(np.int32), "phase": phase, "owner": owner,
            "cut_rows": cut_rows, "cut_into": to[cut_rows] if cut_rows.size else np.zeros(0, np.int64),
            "piece_sizes": sizes, "owner_bias": bias, "piece_cost": weight,
        }
    # trunks first; their owners then take fewer sub-basin rows.  What a trunk costs its owner, in rows of sub-basin it
    # should be spared (fitted on ranks of an 8-, 4- and 2-way CONUS partition timed one by one): its own rows,
    # deep in the network and among the costly ones, five times over, plus the launches that drain the time-skewed trunk
    # at the end of a window (2 of the default 24 chunks; a partly filled GPU gains less than proportionally from
    # having fewer rows; the fraction of a rank's share is _trunk_share, refitted when the chunk counts changed).
    bias = np.zeros(nparts, dtype=np.int64)
    p1 = np.flatnonzero(phase == 1)
    if p1.size:
        owner[p1], trunk_load = lpt_assign(sizes[p1], nparts)
        bias = 5 * trunk_load + np.where(trunk_load > 0, int(_trunk_share(nseg, nparts) * nseg / nparts), 0)
    p0 = np.flatnonzero(phase == 0)
    owner[p0], load0 = lpt_assign(sizes[p0], nparts, bias)
    return {
        "piece": piece.astype(np.int32), "phase": phase, "owner": owner,
        "cut_rows": cut_rows, "cut_into": to[cut_rows] if cut_rows.size else np.zeros(0, np.int64),
        "piece_sizes": sizes, "owner_bias": bias,
    }
