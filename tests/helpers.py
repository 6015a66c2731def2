"""Shared test helpers: golden fixture loading and reference-shaped inputs."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA_COLS = ["dt", "bw", "tw", "twcc", "dx", "n", "ncc", "cs", "s0", "alt"]  # compute.py:1447-1450


def load_kernel_vectors():
    return np.load(os.path.join(GOLDEN, "kernel_vectors.npz"))


def load_toy():
    return json.load(open(os.path.join(GOLDEN, "toy_network.json")))


class LowerColorado:
    """LowerColorado_TX MC-only inputs in the exact form compute_nhd_routing_v02 hands them to the
    kernel callable (compute.py:1447-1467, :1513-1576)."""

    def __init__(self):
        d = np.load(os.path.join(GOLDEN, "lowercolorado_domain.npz"))
        self.ids = d["ids"]
        self.to = d["to"]
        self.qlat = d["qlat"]
        par = dict(zip(d["param_cols"].tolist(), d["params"].T))
        self.nseg = self.ids.shape[0]
        self.dt = 300.0
        cols = {**par, "dt": np.full(self.nseg, self.dt, np.float32), "alt": np.zeros(self.nseg, np.float32)}
        self.data_cols = np.array(DATA_COLS, dtype=object)
        self.data_values = np.stack([cols[c] for c in DATA_COLS], 1).astype(np.float32)
        self.params9 = np.stack([cols[c] for c in ("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")], 1).astype(np.float32)
        rp, ri = d["ref_reach_ptr"], d["ref_reach_ids"]
        self.reaches = [ri[rp[i]:rp[i + 1]].tolist() for i in range(rp.shape[0] - 1)]
        up, ui = d["ref_rconn_ptr"], d["ref_rconn_ids"]
        self.rconn = {int(s): ui[up[i]:up[i + 1]].tolist() for i, s in enumerate(self.ids)}
        self.tailwaters = d["ref_tailwaters"].tolist()
        self.q0 = np.zeros((self.nseg, 3), np.float32)
        self.nts, self.qts = 288, 12

    def row_lists(self):
        row = {int(s): i for i, s in enumerate(self.ids)}
        reaches = [np.array([row[s] for s in r], dtype=np.int64) for r in self.reaches]
        ups = [np.array([row[s] for s in self.rconn.get(r[0], [])], dtype=np.int64) for r in self.reaches]
        return reaches, ups

    def csr(self):
        """upstream rows per row (reference summation order) as CSR arrays for RoutingPlan"""
        from troute_amd.plan import csr_from_lists
        row = {int(s): i for i, s in enumerate(self.ids)}
        return csr_from_lists([[row[u] for u in self.rconn.get(int(s), [])] for s in self.ids])

    def golden(self):
        return np.load(os.path.join(GOLDEN, "lowercolorado_golden.npz"))


def random_network(rng, nseg, max_chain=6, p_junction=0.45, p_triple=0.1):
    """Random dendritic forest as (to[nseg]) with ids 0..nseg-1 in random order: returns
    reaches (lists of rows, reference contract order), upstream lists per reach head, to-array."""
    to = np.full(nseg, -1, dtype=np.int64)
    # build by attaching each new node (in creation order) downstream-first: node i>0 flows into a
    # random earlier node with few upstreams, or starts a new network
    nup = np.zeros(nseg, dtype=np.int64)
    for i in range(1, nseg):
        if rng.random() < 0.02:
            continue                      # new independent network outlet
        for _ in range(8):
            j = int(rng.integers(max(0, i - 50), i))
            cap = 3 if rng.random() < p_triple else (2 if rng.random() < p_junction else 1)
            if nup[j] < cap:
                to[i] = j
                nup[j] += 1
                break
    perm = rng.permutation(nseg)          # shuffle labels so row order is unrelated to topology
    inv = np.empty(nseg, dtype=np.int64)
    inv[perm] = np.arange(nseg)
    to2 = np.full(nseg, -1, dtype=np.int64)
    for i in range(nseg):
        to2[perm[i]] = perm[to[i]] if to[i] >= 0 else -1
    return to2


def reaches_from_to(to):
    """Reference-contract decomposition of a forest given as to[row] (-1 = outlet): reaches are maximal
    chains broken where the downstream row has != 1 upstream; listed so upstream reaches come first."""
    nseg = to.shape[0]
    ups = [[] for _ in range(nseg)]
    for i in range(nseg):
        if to[i] >= 0:
            ups[to[i]].append(i)
    reaches, heads_up = [], []
    done = np.zeros(nseg, dtype=bool)
    # iterative post-order from each outlet
    for o in np.flatnonzero(to < 0).tolist():
        stack = [(o, False)]
        while stack:
            node, expanded = stack.pop()
            # node is the LAST (most downstream) segment of a reach; walk up while single upstream
            chain = [node]
            while len(ups[chain[-1]]) == 1:
                chain.append(ups[chain[-1]][0])
            head = chain[-1]
            if not expanded:
                stack.append((node, True))
                for u in ups[head]:
                    stack.append((u, False))
            else:
                reaches.append(chain[::-1])
                heads_up.append(list(ups[head]))
                done[chain] = True
    assert done.all()
    return reaches, heads_up, ups


def flow_engine(precision=32):
    """True when plans of this precision run on the dataflow engine (k_mc_flow), False on the level engine."""
    import os
    return precision == 32 and os.environ.get("TRMC_ENGINE", "flow") != "levels"
