#!/usr/bin/env python3
"""The reference's run-set loop (nwm_routing/__main__.py:195-333: forcing in, compute_nhd_routing_v02, new_q0, output) on the
LowerColorado_TX test domain as ONE stream of days on an MI355X -- `troute_amd.sequence.RouteStream`:

    python examples/stream_lowercolorado.py [--days 5] [--out /tmp/lc_out]

Inputs: the domain of the reference's own test (test/LowerColorado_TX; kept as arrays under tests/golden/): 11 248 segments,
24 h of hourly lateral inflow.  Every run set is that day's forcing scaled by a factor (stand-in for consecutive days of files);
per run set the script writes what `nwm_output_generator` writes for stream output (troute_output_<time>.nc: flow, velocity,
depth at the output interval -- `troute_amd.nhd_io.write_flowveldepth`) and carries the state on in HBM.  The last run set is also
routed by the drop-in `compute_nhd_routing_v02` from DataFrames, restarted from the previous run set's state, and must agree bit
for bit.  Needs a GPU: the library has no CPU path."""
import argparse
import datetime
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from troute_amd import nhd_io, nhd_network as nn                       # noqa: E402
from troute_amd.distributed import ShardedRouter                        # noqa: E402
from troute_amd.routing.compute import compute_nhd_routing_v02          # noqa: E402
from troute_amd.sequence import RouteStream                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--days", type=int, default=5)
    ap.add_argument("--out", default="/tmp/lowercolorado_stream")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    d = np.load(os.path.join(ROOT, "tests", "golden", "lowercolorado_domain.npz"))
    ids, to_id, qlat = d["ids"], d["to"], d["qlat"]
    par = dict(zip(d["param_cols"].tolist(), d["params"].T))
    nseg, dt, nts, qts = ids.shape[0], 300.0, 288, 12
    cols9 = ("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")
    par["dt"] = np.full(nseg, dt, np.float32)
    params = np.stack([par[c] for c in cols9], 1).astype(np.float32)
    row = {int(s): i for i, s in enumerate(ids)}
    to = np.array([row.get(int(t), -1) for t in to_id], dtype=np.int64)              # downstream ROW of every row (-1: outlet)
    q0 = np.zeros((nseg, 3), np.float32)
    factors = [1.0, 0.6, 1.8, 1.2, 0.8, 2.5, 0.4]
    forcing_of = lambda k: np.ascontiguousarray(qlat[:, :nts // qts] * np.float32(factors[k % len(factors)]))   # noqa: E731
    t0 = datetime.datetime(2021, 8, 23, 13, 0)

    # ---- the stream: one plan in cluster order, the days pushed as the iterator yields them, products as they are handed over
    router = ShardedRouter(to, params, stream=True)
    states, tic = {}, time.perf_counter()
    with RouteStream(router, nts, qts, output_stride=qts) as rs:
        for k, hydrographs, state, fvd in rs.route((forcing_of(k) for k in range(a.days)), q0):
            start = t0 + datetime.timedelta(seconds=k * nts * dt)
            frame = pd.DataFrame(fvd.reshape(nseg, -1), index=ids)                   # every qts-th step of (q, v, d), hourly
            nhd_io.write_flowveldepth(a.out, None, frame, pd.DataFrame(), [], start, dt, -1, ".nc", 60, output_stride=qts)
            states[k] = np.array(state, copy=True)
            print(f"run set {k}: {hydrographs.shape[0]} outlet hydrographs, peak outlet flow {float(hydrographs.max()):.3f} m3/s, "
                  f"{len(os.listdir(a.out))} output file(s)")
        outlets = np.array(rs.outlet_rows, copy=True)
    el = time.perf_counter() - tic
    router.close()
    print(f"{a.days} run sets of {nseg} segments x {nts} steps in {el:.2f} s (files included); outlets: rows {outlets.tolist()[:5]} ...")

    # ---- the last run set once more through the reference's call surface, from the state the stream carried into it
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(ids, to_id)}
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    k = a.days - 1
    param_df = pd.DataFrame({c: par[c] for c in cols9[1:]}, index=ids)
    param_df["alt"] = 0.0
    prev = states[k - 1] if k > 0 else q0
    q0_df = pd.DataFrame(prev, index=ids, columns=["qu0", "qd0", "h0"])
    qlat_df = pd.DataFrame(forcing_of(k), index=ids)
    e = pd.DataFrame()
    results, _ = compute_nhd_routing_v02(conn, rconn, {}, reaches_bytw, "V02-structured", "by-network", 10000, 4, None, dt, nts, qts, ind,
                                         param_df, q0_df, qlat_df, e, e, e, e, e, e, e, e, e, e, e, {}, True, False, e, {}, e, False,
                                         [{}, {}])
    got = {}
    for r in results:
        for sid, rowv in zip(r[0], r[1]):
            got[int(sid)] = rowv
    last = np.stack([got[int(s)] for s in ids])                                       # [nseg, nts * 3] in the table's order
    final = np.stack([last[:, -3], last[:, -3], last[:, -1]], 1)
    same = np.array_equal(final.view(np.uint32), states[k].view(np.uint32))
    print(f"run set {k} through compute_nhd_routing_v02 from the stream's state of run set {k - 1}: final state bit-identical: {same}")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
