"""Graph utilities the Muskingum-Cunge path consumes -- host-side mirror of the functions of
``troute.nhd_network`` that feed ``compute_nhd_routing_v02``
(src/troute-network/troute/nhd_network.py).

Same names, arguments and return structures as the reference, so a caller (or a
test written against the reference) can switch modules; the implementations
are this project's own and array-based inside (a CONUS-size table is ~3 M
nodes -- the reference's dict/set walks take seconds there).

    extract_connections   nhd_network.py:26-53
    reverse_network       :111-130
    headwaters/tailwaters :157-199
    reachable             :201-243      reachable_network :245-275
    split_at_junction     :278-293      (+ gage / waterbody break variants :295-358)
    dfs_decomposition     :503-557      reach lists; contract: every reach is listed
                                        upstream -> downstream and after all reaches above it
    build_subnetworks     :691-771      ordered sub-network decomposition
    organize_independent_networks  nhd_network_utilities_v02.py:133-200

Contract notes
  * Reach *sets* equal the reference's (tests/test_nhd_network.py checks them against
    fixtures emitted by the reference code); the order of reaches inside the list is
    any order honouring the contract above -- the reference's own order depends on
    Python set iteration.
  * ``build_subnetworks`` returns the reference's structure
    {tailwater: {order: {subnetwork_tailwater: set(segments)}}} with order 0 at the
    tailwater; cut points are chosen by sub-tree size, not by the reference's truncated
    breadth-first search, so the pieces differ while the contract (each order only
    depends on higher orders) holds.  Routed results do not depend on the cut
    (tests/test_gpu_parity.py::test_upstream_results_composition_equals_whole_network).
"""
from collections import defaultdict
from functools import partial

import numpy as np


def reverse_dict(d):
    """Reverse a 1-1 mapping."""
    return {v: k for k, v in d.items()}


def extract_connections(rows, target_col, terminal_codes=None):
    """{segment id: [downstream segment ids]} from a DataFrame indexed by segment id."""
    terminal = {0} if terminal_codes is None else set(terminal_codes)
    network = {}
    idx = rows.index.tolist()
    dst = rows[target_col].tolist()
    for s, d in zip(idx, dst):
        lst = network.setdefault(s, [])
        if d not in terminal:
            lst.append(d)
    return network


def extract_waterbody_connections(rows, target_col="waterbody", waterbody_null=-9999):
    """{segment id: lake id} for segments lying inside a waterbody (nhd_network.py:55-78)."""
    col = rows[target_col]
    sel = col != waterbody_null
    return {int(k): int(v) for k, v in zip(rows.index[sel].tolist(), col[sel].tolist())}


def replace_waterbodies_connections(connections, waterbodies):
    """Collapse every waterbody to ONE node named by its lake id (nhd_network.py:637-689).

    connections  {segment: [downstream segments]};  waterbodies {segment: lake id}
    Returns (new_conn, link_lake): the segments inside a waterbody disappear, the lake node drains to the
    segments just outside the footprint, segments flowing into the footprint point at the lake id;
    link_lake maps each lake id to one of its in-waterbody outlet segments."""
    members = {}
    for seg, lake in waterbodies.items():
        members.setdefault(lake, []).append(seg)
    new_conn, link_lake = {}, {}
    for n, dsts in connections.items():
        lake = waterbodies.get(n)
        if lake is not None:
            if lake in new_conn:
                continue
            inside = set(members[lake])
            shore = []                                      # downstream neighbours outside the footprint
            for m in members[lake]:
                for d in connections.get(m, ()):
                    if d not in inside and d not in shore:
                        shore.append(d)
            new_conn[lake] = shore
            if shore:                                       # a member that drains to the first shore segment
                link_lake[lake] = next(m for m in members[lake] if shore[0] in connections.get(m, ()))
        elif any(d in waterbodies for d in dsts):
            new_conn[n] = [waterbodies.get(d, d) for d in dsts]
        else:
            new_conn[n] = dsts
    return new_conn, link_lake


def reverse_network(N):
    """{node: [nodes that list it as a target]} -- sources keep the iteration order of N."""
    rg = {}
    for src, dsts in N.items():
        rg.setdefault(src, [])
        for n in dsts:
            rg.setdefault(n, []).append(src)
    return rg


def headwaters(N):
    """Nodes that are nobody's target (tailwaters when N is a reversed network)."""
    targets = set()
    for v in N.values():
        targets.update(v)
    return N.keys() - targets


def tailwaters(N):
    """Targets that are not keys, plus keys with no target."""
    tw = set()
    for m, v in N.items():
        if not v:
            tw.add(m)
        for n in v:
            if n not in N:
                tw.add(n)
    return tw


def reachable(N, sources=None, targets=None):
    """{source: set of nodes reachable from it through N} (search stops at `targets`)."""
    if sources is None:
        sources = headwaters(N)
    stop = set(targets) if targets is not None else None
    rv = {}
    for h in sources:
        seen = {h}
        frontier = [h]
        while frontier:
            nxt = []
            for x in frontier:
                if stop is not None and x in stop and x != h:
                    continue
                for y in N.get(x, ()):
                    if y not in seen:
                        seen.add(y)
                        nxt.append(y)
            frontier = nxt
        rv[h] = seen
    return rv


def reachable_network(N, sources=None, targets=None, check_disjoint=True):
    """{tailwater: {node: [upstream nodes]}} for every independent network of the reversed graph N."""
    reached = reachable(N, sources=sources, targets=targets)
    if check_disjoint and len(reached) > 1:
        total = sum(len(s) for s in reached.values())
        if total != len(set().union(*reached.values())):
            raise ValueError("Networks not disjoint")
    return {k: {m: N.get(m, []) for m in nodes} for k, nodes in reached.items()}


# ---- reach break predicates: path_func(path, node) -> True if `node` continues `path` ---------------
def split_at_junction(network, path, node):
    return len(network[node]) == 1


def split_at_gages_and_junctions(gage_nodes, network, path, node):
    if (path[-1] in gage_nodes) or (node in gage_nodes):
        return False
    return len(network[node]) == 1


def split_at_waterbodies_and_junctions(waterbody_nodes, network, path, node):
    if (path[-1] in waterbody_nodes) ^ (node in waterbody_nodes):
        return False
    return len(network[node]) == 1


def split_at_gages_waterbodies_and_junctions(gage_nodes, waterbody_nodes, network, path, node):
    if (path[-1] in gage_nodes) or (node in gage_nodes):
        return False
    if (path[-1] in waterbody_nodes) ^ (node in waterbody_nodes):
        return False
    return len(network[node]) == 1


def dfs_decomposition(N, path_func, source_nodes=None):
    """Decompose the reversed network N into reaches.

    A reach starts at a node that does not continue the reach below it... seen from upstream: at
    every headwater and at every node `path_func` refuses to append, and runs downstream for as
    long as `path_func(path, next_downstream_node)` accepts.  Returned list: every reach
    upstream -> downstream, and after every reach that drains into it.
    """
    if source_nodes is None:
        source_nodes = headwaters(N)
    # downstream pointer inside this (sub)network
    down = {}
    for n, ups in N.items():
        for u in ups:
            down[u] = n
    reaches = []
    emitted = set()
    for tw in source_nodes:
        # post-order over the nodes above tw, iteratively
        order = []
        stack = [tw]
        seen = {tw}
        while stack:
            n = stack.pop()
            order.append(n)
            for u in N.get(n, ()):
                if u not in seen and u in N:
                    seen.add(u)
                    stack.append(u)
        # `order` is a pre-order from the tailwater: reversed, every node comes after all nodes above it
        for n in reversed(order):
            if n in emitted:
                continue
            # n starts a reach iff nothing upstream continues into it: it has no upstream, or the
            # path ending at its upstream neighbour was refused -- equivalently n is reached here
            # un-emitted only when no reach from above has absorbed it
            path = [n]
            emitted.add(n)
            cur = n
            while cur != tw and cur in down:
                nxt = down[cur]
                if nxt not in seen or not path_func(path, nxt):
                    break
                path.append(nxt)
                emitted.add(nxt)
                cur = nxt
            reaches.append(path)
    return reaches


def dfs_decomposition_depth_tuple(RN, path_func, source_nodes=None):
    """``dfs_decomposition`` with the junction order of every reach: a list of ``(depth, reach)`` in the order of the
    decomposition, depth 0 for the reach that ends at the tailwater, +1 across every reach break upstream of it.
    Reference: nhd_network.py:362-419 (which walks the coalesced reach graph a second time and zips the two lists)."""
    reaches = dfs_decomposition(RN, path_func, source_nodes)
    reach_of_tail = {r[-1]: k for k, r in enumerate(reaches)}
    down = {}
    for n, ups in RN.items():
        for u in ups:
            down[u] = n
    depth = [None] * len(reaches)

    def depth_of(k):
        chain = []
        while depth[k] is None:
            nxt = down.get(reaches[k][-1])
            if nxt is None:                       # the tailwater reach
                depth[k] = 0
                break
            chain.append(k)
            # the reach that contains the node below this reach's last segment starts with it or holds it inside;
            # find it through its tail by walking down its members
            n = nxt
            while n not in reach_of_tail:
                n = down[n]
            k = reach_of_tail[n]
        base = depth[k]
        for j, c in enumerate(reversed(chain)):
            depth[c] = base + j + 1
        return base

    for k in range(len(reaches)):
        if depth[k] is None:
            depth_of(k)
    return list(zip(depth, reaches))


def organize_independent_networks(connections, wbody_break_segments=None, gage_break_segments=None):
    """(independent_networks, reaches_bytw, rconn), nhd_network_utilities_v02.py:133-200."""
    rconn = reverse_network(connections)
    independent_networks = reachable_network(rconn)
    reaches_bytw = {}
    for tw, net in independent_networks.items():
        if wbody_break_segments and gage_break_segments:
            path_func = partial(split_at_gages_waterbodies_and_junctions, gage_break_segments,
                                wbody_break_segments, net)
        elif gage_break_segments:
            path_func = partial(split_at_gages_and_junctions, gage_break_segments, net)
        elif wbody_break_segments:
            path_func = partial(split_at_waterbodies_and_junctions, wbody_break_segments, net)
        else:
            path_func = partial(split_at_junction, net)
        reaches_bytw[tw] = dfs_decomposition(net, path_func)
    return independent_networks, reaches_bytw, rconn


def build_subnetworks(connections, rconn, min_size, sources=None):
    """Ordered sub-network decomposition: {tailwater: {order: {sub_tailwater: set(segments)}}}.

    Order 0 holds the piece containing the network tailwater; a piece of order k receives flow only
    from pieces of order k+1.  Pieces are cut where a tributary's sub-tree first drops to
    <= min_size segments (and the remaining trunk is cut every ~min_size segments of its own)."""
    if sources is None:
        sources = headwaters(rconn)
    master = {}
    for tw in sources:
        # sub-tree sizes by post-order
        order = []
        stack = [tw]
        while stack:
            n = stack.pop()
            order.append(n)
            stack.extend(rconn.get(n, ()))
        size = {}
        for n in reversed(order):
            size[n] = 1 + sum(size[u] for u in rconn.get(n, ()))
        pieces = defaultdict(dict)
        # grow a piece from its tailwater upstream until it holds min_size nodes; what is left
        # above its frontier becomes new pieces of the next order
        frontier = [(tw, 0)]
        while frontier:
            root, k = frontier.pop()
            members = set()
            queue = [root]
            nxt_roots = []
            while queue:
                n = queue.pop(0)
                if len(members) >= max(1, min_size) and n != root:
                    nxt_roots.append(n)
                    continue
                members.add(n)
                queue.extend(sorted(rconn.get(n, ()), key=lambda u: -size[u]))
            pieces[k][root] = members
            frontier.extend((r, k + 1) for r in nxt_roots)
        master[tw] = dict(pieces)
    return master
