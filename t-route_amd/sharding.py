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


def lpt_assign(sizes, nparts, initial_load=None, speed=None):
    """Longest-processing-time bin packing: part index per item (`initial_load`: what the bins hold already; `speed`: relative
    pace of every bin -- an item goes where it would be FINISHED first, (load + size) / speed)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    part = np.zeros(sizes.shape[0], dtype=np.int32)
    load = np.zeros(nparts, dtype=np.int64) if initial_load is None else np.asarray(initial_load, dtype=np.int64).copy()
    inv = None if speed is None else 1.0 / np.maximum(np.asarray(speed, dtype=np.float64), 1e-9)
    for i in np.argsort(-sizes, kind="stable").tolist():
        p = int(np.argmin(load)) if inv is None else int(np.argmin((load + sizes[i]) * inv))
        part[i] = p
        load[p] += sizes[i]
    return part, load


def rank_speeds(loads, times):
    """Relative pace of every rank from a measured window: (cost it carried) / (time it took), mean 1.  What a trunk does to
    its owner -- deep, tightly coupled rows, a skewed drain at the end of the window -- shows up here as a lower pace and
    is handed back to partition(rank_speed=...) instead of being modelled."""
    loads = np.asarray(loads, dtype=np.float64)
    times = np.maximum(np.asarray(times, dtype=np.float64), 1e-9)
    v = np.maximum(loads, 1.0) / times
    return v / v.mean()


def partition(to, nparts, max_piece_frac=None, row_cost=None, rank_speed=None, previous=None):
    """Split rows into pieces for `nparts` workers.

    row_cost: optional [nseg] measured cost of every row (e.g. ``ShardedRouter.iteration_hint()`` of a tuning window:
    the secant iterations a row needs per step, over-bank steps weighted) -- the pieces are then packed by the cost they
    carry instead of by their row counts, and a trunk weighs what its rows were measured to cost.  Without it (the first
    window of a network) rows count equally.
    rank_speed: optional [nparts] relative pace of the ranks MEASURED on a window routed with an earlier partition of the
    same network (``rank_speeds``: cost carried / time taken) -- pieces then go where they are finished first.  A trunk's
    owner is slower than its share of the cost says (deep rows that wait for each other, the drain of the time-skew);
    how much depends on the engine, the device and the network, so it is measured, not assumed: the first partition of a
    network counts a trunk at its cost alone, the tuning window every long run starts with times the ranks, and the
    partition the run continues with is packed by those times (bench.py does exactly that).
    previous: the partition the speeds were measured with -- its trunks keep their owners (a pace belongs to a rank AS the
    owner of its trunk), only sub-basins and small networks move.

    Returns dict:
      piece     int32 [nseg]  piece id of every row
      phase     int32 [npieces]  0 = no upstream piece, 1 = trunk fed by phase-0 pieces
      owner     int32 [npieces]  worker that routes the piece
      cut_rows  int64 [ncut]     rows (outlets of phase-0 sub-basins) whose hydrograph is handed
                                 to the trunk they drain into
      cut_into  int64 [ncut]     trunk row each cut row flows into
      owner_bias int64 [nparts]  what the trunks a worker owns weigh (rows, or cost units with row_cost)
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
        # measured costs: a piece weighs the cost of its rows (in thousandths of the mean row, so that the integer packing
        # below keeps its resolution); trunks first, then the sub-basins and small networks on top of them
        c = np.maximum(np.asarray(row_cost, dtype=np.float64), 1.0)
        c = c * (1000.0 / c.mean())
        weight = np.bincount(piece, weights=c, minlength=npieces).astype(np.int64)
        bias = np.zeros(nparts, dtype=np.int64)
        p1 = np.flatnonzero(phase == 1)
        if p1.size:
            owner[p1], bias = lpt_assign(weight[p1], nparts)
            if previous is not None:
                owner[p1] = previous["owner"][p1]
                bias = np.bincount(owner[p1], weights=weight[p1], minlength=nparts).astype(np.int64)
        p0 = np.flatnonzero(phase == 0)
        owner[p0], _ = lpt_assign(weight[p0], nparts, bias, speed=rank_speed)
        return {
            "piece": piece.astype(np.int32), "phase": phase, "owner": owner,
            "cut_rows": cut_rows, "cut_into": to[cut_rows] if cut_rows.size else np.zeros(0, np.int64),
            "piece_sizes": sizes, "owner_bias": bias, "piece_cost": weight,
        }
    # trunks first (counted at their rows); their owners then take that many fewer sub-basin rows
    bias = np.zeros(nparts, dtype=np.int64)
    p1 = np.flatnonzero(phase == 1)
    if p1.size:
        owner[p1], bias = lpt_assign(sizes[p1], nparts)
        if previous is not None:
            owner[p1] = previous["owner"][p1]
            bias = np.bincount(owner[p1], weights=sizes[p1], minlength=nparts).astype(np.int64)
    p0 = np.flatnonzero(phase == 0)
    owner[p0], load0 = lpt_assign(sizes[p0], nparts, bias, speed=rank_speed)
    return {
        "piece": piece.astype(np.int32), "phase": phase, "owner": owner,
        "cut_rows": cut_rows, "cut_into": to[cut_rows] if cut_rows.size else np.zeros(0, np.int64),
        "piece_sizes": sizes, "owner_bias": bias,
    }
