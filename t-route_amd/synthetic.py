"""Seeded synthetic CONUS-scale river network (SURVEY.md 8(d), configs 3-4 of BASELINE.json).

The real NWM RouteLink (2 729 077 segments) is not in the reference tree, so
the benchmark network is generated: same segment / network counts, the
reference's published shape statistics (doc/AGU_Poster.md:37-41, :120-123,
:211-213, :225-229) and the NWM-2.1 parameter ranges of
src/kernel/muskingum/test_suite_parameters.py:5-14.

Shape
  * independent networks: one dominant basin (~half of all segments), >= 5000
    single-reach networks, the rest heavy-tailed (Pareto alpha = 1.1, clipped);
  * every network is a uniformly random full binary tree of REACHES (Remy's
    algorithm): fan-in 2 everywhere, 0.2 % of junctions get a third tributary;
    the height of such a tree grows like 2*sqrt(pi*junctions), which puts the
    dominant basin's depth in the 2-3 thousand reaches the poster reports (2 218);
  * segments per reach: truncated geometric, mean 1.298 (= 2 729 077 / 2 102 010).
Parameters
  * dx, cs, s0, n: marginal distributions of the NWM table; bw follows the
    downstream hydraulic-geometry scaling bw ~ A^0.4 (A = segments draining
    through), tw = 5/3 bw, twcc = 3 tw, ncc = 2 n (the table's stated relations).
Forcing
  * qlat: 11 % zeros, else lognormal(median 2.3e-4 m3/s, sigma 2.3) clipped at 1,
    times a smooth diurnal factor 1 +- 0.2; 25 hourly columns for a 24 h window.
"""
import os

import numpy as np

CONUS_NSEG = 2_729_077
CONUS_NNET = 14_713
DEFAULT_SEED = 20250117


def _remy_parent(n_junctions, rng):
    """Uniform random full binary tree with n_junctions internal nodes: parent[] (root: -1)."""
    total = 2 * n_junctions + 1
    parent = [-1] * total
    if n_junctions == 0:
        return parent
    picks = rng.random(n_junctions)
    cnt = 1
    for k in range(n_junctions):
        x = int(picks[k] * cnt)          # uniform over the cnt existing nodes
        y = cnt
        parent[y] = parent[x]
        parent[x] = y
        parent[cnt + 1] = y
        cnt += 2
    return parent


def _network_reach_counts(nseg, nnet, rng, mean_len, dominant_frac=0.495, n_single=5000):
    """Reach counts (odd = 2J+1) per network, dominant first."""
    nreach_total = int(round(nseg / mean_len))
    n_single = min(n_single, max(0, nnet - 2))
    dom = int(nreach_total * dominant_frac) | 1
    rest_n = nnet - 1 - n_single
    rest_total = nreach_total - dom - n_single
    if rest_n > 0:
        raw = (rng.pareto(1.1, rest_n) + 1.0) * 3.0
        raw = np.minimum(raw, 250_000 / mean_len)
        raw *= rest_total / raw.sum()
        cnt = np.maximum(3, (raw.astype(np.int64) | 1))
    else:
        cnt = np.zeros(0, dtype=np.int64)
    return np.concatenate([[dom], cnt, np.ones(n_single, dtype=np.int64)])


def generate(nseg=CONUS_NSEG, nnet=CONUS_NNET, seed=DEFAULT_SEED, nq=25, cache_dir=None, dt=300.0):
    """Returns a dict of arrays:
      to        int64 [nseg]  downstream row (-1 at outlets); rows are in random (id) order
      net       int32 [nseg]  independent-network ordinal (0 = dominant basin)
      params    float32 [nseg, 9]  dt dx bw tw twcc n ncc cs s0  (trmc.h column order)
      qlat      float32 [nseg, nq]
    plus 'nreach', 'reach_depth' facts.  Deterministic in (nseg, nnet, seed, nq).
    """
    key = f"synth_{nseg}_{nnet}_{seed}_{nq}.npz"
    if cache_dir:
        path = os.path.join(cache_dir, key)
        if os.path.exists(path):
            d = np.load(path)
            return {k: d[k] for k in d.files}
    rng = np.random.default_rng(seed)
    mean_len = CONUS_NSEG / 2_102_010
    counts = _network_reach_counts(nseg, nnet, rng, mean_len)

    # ---- reach-level forest ---------------------------------------------------------------
    r_parent, r_net = [], []
    base = 0
    for i, c in enumerate(counts.tolist()):
        par = _remy_parent((c - 1) // 2, rng)
        r_parent.extend(p + base if p >= 0 else -1 for p in par)
        r_net.extend([i] * len(par))
        base += len(par)
    r_parent = np.asarray(r_parent, dtype=np.int64)
    r_net = np.asarray(r_net, dtype=np.int32)
    # 0.2 % of junctions receive a third (headwater) tributary
    is_junction = np.zeros(r_parent.shape[0], dtype=bool)
    is_junction[r_parent[r_parent >= 0]] = True
    jn = np.flatnonzero(is_junction)
    extra = jn[rng.random(jn.shape[0]) < 0.002]
    r_parent = np.concatenate([r_parent, extra])
    r_net = np.concatenate([r_net, r_net[extra]])
    nreach = r_parent.shape[0]

    # ---- segments per reach: truncated geometric, then trimmed to exactly nseg -------------
    r = 1.0 - 1.0 / mean_len
    pmf = r ** np.arange(25)
    pmf /= pmf.sum()
    rlen = rng.choice(np.arange(1, 26), size=nreach, p=pmf).astype(np.int64)
    diff = int(rlen.sum()) - nseg
    if diff > 0:                                  # shorten random long reaches
        cand = np.flatnonzero(rlen > 1)
        while diff > 0:
            take = rng.permutation(cand)[:diff]
            rlen[take] -= 1
            diff = int(rlen.sum()) - nseg
            cand = np.flatnonzero(rlen > 1)
    elif diff < 0:
        take = rng.choice(nreach, size=-diff, replace=True)
        np.add.at(rlen, take, 1)
    assert int(rlen.sum()) == nseg

    # ---- expand to segments (topological ids), then shuffle labels ---------------------------
    first = np.zeros(nreach + 1, dtype=np.int64)
    first[1:] = np.cumsum(rlen)
    seg_reach = np.repeat(np.arange(nreach), rlen)
    to = np.arange(1, nseg + 1, dtype=np.int64)            # next segment inside the reach
    last = first[1:] - 1                                     # last segment of each reach
    to[last] = np.where(r_parent >= 0, first[:-1][np.maximum(r_parent, 0)], -1)
    net = r_net[seg_reach]

    # reach depth (longest chain of reaches, headwater -> outlet) for the record
    depth = np.zeros(nreach, dtype=np.int32)
    for grp in _levels_desc(_dist_to_root(r_parent)):
        par = r_parent[grp]
        ok = par >= 0
        np.maximum.at(depth, par[ok], depth[grp[ok]] + 1)

    # drainage count A (segments draining through each segment) for hydraulic geometry
    A = np.ones(nseg, dtype=np.float64)
    sdist = _dist_to_root(to)
    for lvl_idx in _levels_desc(sdist):
        t = to[lvl_idx]
        ok = t >= 0
        np.add.at(A, t[ok], A[lvl_idx[ok]])

    n_ = np.where(rng.random(nseg) < 0.95, 0.06, np.where(rng.random(nseg) < 0.5, 0.05, 0.04))
    bw = np.clip(1.2 * A ** 0.4 * rng.lognormal(0.0, 0.3, nseg), 0.135, 230.0)
    tw = bw * 5.0 / 3.0
    params = np.stack([
        np.full(nseg, dt),
        np.clip(rng.lognormal(np.log(1549.0), 0.85, nseg), 1.0, 95714.0),
        bw, tw, 3.0 * tw, n_, 2.0 * n_,
        np.clip(rng.normal(0.586, 0.195, nseg), 0.085, 2.25),
        np.clip(rng.lognormal(np.log(0.006), 1.6, nseg), 1e-5, 4.6),
    ], 1).astype(np.float32)

    base_q = np.minimum(rng.lognormal(np.log(2.3e-4), 2.3, nseg), 1.0) * (rng.random(nseg) > 0.11)
    hours = np.arange(nq)
    diurnal = 1.0 + 0.2 * np.sin(2 * np.pi * (hours[None, :] / 24.0 + rng.random(nseg)[:, None]))
    qlat = (base_q[:, None] * diurnal).astype(np.float32)

    perm = rng.permutation(nseg)                 # row = perm[topological id]
    to_rows = np.full(nseg, -1, dtype=np.int64)
    has = to >= 0
    to_rows[perm[has]] = perm[to[has]]
    to_rows[perm[~has]] = -1
    inv = np.empty(nseg, dtype=np.int64)
    inv[perm] = np.arange(nseg)
    out = {
        "to": to_rows, "net": net[inv], "params": params[inv], "qlat": qlat[inv],
        "nreach": np.int64(nreach), "reach_depth": np.int64(depth.max() + 1),
        "net_sizes": np.bincount(net, minlength=len(counts)).astype(np.int64),
    }
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        np.savez(os.path.join(cache_dir, key), **out)
    return out


def forcing(nseg, nq=25, seed=DEFAULT_SEED + 1, previous=None, sigma=0.06, redraw=0.001, persistence=None):
    """Another day of lateral inflow with the statistics of ``generate``'s (11 % zeros, else lognormal with
    median 2.3e-4 m3/s, sigma 2.3, clipped at 1, times a smooth diurnal factor 1 +- 0.2) under its own seed -- the
    window a plan is TIMED on after it was tuned on another one.
    previous = None: an independent draw (no row keeps its magnitude: the hardest case for a plan tuned the day before).
    previous = the forcing of the day before [nseg, nq]: the next day of the SAME basin -- every row's daily mean times
    a lognormal factor (sigma), ``redraw`` of the rows drawn anew, new diurnal phases: lateral inflow is mostly
    baseflow, whose spatial pattern persists from day to day.  The defaults are what the reference's own forcing files
    show one day apart (test/LowerColorado_TX/channel_forcing, 2021-08-23 13:00-16:00 against 2021-08-24 13:00-16:00,
    qBucket + qSfcLatRunoff of the 11 248 segments): standard deviation of the log ratio 0.056, rank correlation 0.9998,
    0.05 % of the rows switch between zero and non-zero inflow.
    persistence (optional, 0..1): the share of rows that keep their magnitude, i.e. redraw = 1 - persistence -- how much a
    result depends on the day-to-day persistence is measured by varying it (bench.py --persistence)."""
    if persistence is not None:
        redraw = 1.0 - float(persistence)
    rng = np.random.default_rng(seed)
    fresh = np.minimum(rng.lognormal(np.log(2.3e-4), 2.3, nseg), 1.0) * (rng.random(nseg) > 0.11)
    if previous is None:
        base_q = fresh
    else:
        mean_prev = np.asarray(previous, dtype=np.float64).mean(axis=1)
        base_q = np.minimum(mean_prev * rng.lognormal(0.0, sigma, nseg), 1.0)
        anew = rng.random(nseg) < redraw
        base_q[anew] = fresh[anew]
    hours = np.arange(nq)
    diurnal = 1.0 + 0.2 * np.sin(2 * np.pi * (hours[None, :] / 24.0 + rng.random(nseg)[:, None]))
    return (base_q[:, None] * diurnal).astype(np.float32)


def _dist_to_root(parent):
    """Edges from each node to its root (pointer doubling: log2(height) vectorised sweeps)."""
    n = parent.shape[0]
    idx = np.arange(n, dtype=np.int64)
    anc = np.where(parent >= 0, parent, idx)
    d = (parent >= 0).astype(np.int64)
    while True:
        nxt = anc[anc]
        d = d + d[anc]
        if np.array_equal(nxt, anc):
            return d
        anc = nxt


def _levels_desc(dist):
    """Index arrays grouping nodes by distance-to-root, farthest first."""
    order = np.argsort(-dist, kind="stable")
    d = dist[order]
    cuts = np.flatnonzero(np.diff(d)) + 1
    return np.split(order, cuts)


def upstream_csr(to):
    """to[row] -> (up_ptr, up_idx) with upstream rows in ascending row order
    (= the reference's rconn order for an id-sorted table, nhd_network.py:111-130)."""
    nseg = to.shape[0]
    has = to >= 0
    src = np.flatnonzero(has)
    dst = to[src]
    order = np.argsort(dst, kind="stable")       # stable: ascending src within each dst
    up_idx = src[order].astype(np.int64)
    counts = np.bincount(dst, minlength=nseg)
    up_ptr = np.zeros(nseg + 1, dtype=np.int64)
    up_ptr[1:] = np.cumsum(counts)
    return up_ptr, up_idx
