"""A stream of windows (trmc_stream_*) against the same days routed one by one on the same plan: every day's final state of every
row and the hydrographs of sampled rows bit for bit, and what a day costs either way.
  python tools/stream_probe.py [--nseg N] [--days D] [--wide-min-rows R] [--hint] [--stride n] [--full]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic as S                     # noqa: E402
from troute_amd.plan import RoutingPlan                   # noqa: E402
from troute_amd import _lib                               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nseg", type=int, default=S.CONUS_NSEG)
ap.add_argument("--days", type=int, default=8)
ap.add_argument("--wide-min-rows", type=int, default=0)
ap.add_argument("--wide-k", type=int, default=0)
ap.add_argument("--wide-levels", type=int, default=0)
ap.add_argument("--split", type=int, default=0)
ap.add_argument("--hint", action="store_true")
ap.add_argument("--stride", type=int, default=0)
ap.add_argument("--full", action="store_true")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--slots", type=int, default=0)
ap.add_argument("--velocity-on-demand", type=int, default=0)
ap.add_argument("--later", type=int, default=0, help="hand a day over this many days later than its last row allows (and hold as many more slots)")
a = ap.parse_args()
_lib.single_hw_queue_per_priority("stream_probe")
nnet = S.CONUS_NNET if a.nseg == S.CONUS_NSEG else max(1, a.nseg // 185)
net = S.generate(a.nseg, nnet, cache_dir=os.environ.get("TRMC_SYNTH_CACHE", "/tmp"))
n = a.nseg
up_ptr, up_idx = S.upstream_csr(net["to"])
nsteps, qts = 288, 12
days = [net["qlat"]]
for d in range(1, min(a.days, 4)):
    days.append(S.forcing(n, previous=days[-1], seed=S.DEFAULT_SEED + 1 + d))
pinned = []
for q in days:
    b = _lib.result_empty(q.shape, q.dtype, always_pinned=True)
    b[...] = q
    pinned.append(b)
q0 = np.zeros((n, 3), np.float32)
rng = np.random.default_rng(5)
sample = np.sort(rng.choice(n, min(n, 3000), replace=False))
opts = {"wide_min_rows": a.wide_min_rows, "wide_k": a.wide_k, "cluster_rows": 128, "wide_levels": a.wide_levels, "stream_split": a.split, "velocity_on_demand": a.velocity_on_demand}
hint = None
if a.hint:
    with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=True, engine="levels", options=opts) as p:
        p.upload_forcing(nsteps, days[0], q0)
        p.route_device(nsteps, qts, True)
        p.upload_forcing(nsteps, days[1 % len(days)], None)
        p.collect_cost(True)
        p.route_device(nsteps, qts, True)
        cost, ns = p.download_cost()
        hint = np.minimum(255, (cost.astype(np.float64) * 16 / ns).round()).astype(np.uint8)
with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=True, engine="levels", cost_hint=hint, options=opts) as p:
    rs = p.rowset(sample)
    ref = []
    if not a.no_check:
        t0 = time.perf_counter()
        for d in range(a.days):
            p.upload_forcing(nsteps, days[d % len(days)], q0 if d == 0 else None)
            st = p.route_device(nsteps, qts, True)
            fin = p.download_final_state()
            hyd = p.gather_flow_rows(sample)
            fvd = p.download_fvd(a.stride) if (a.stride or a.full) else None
            ref.append((fin, hyd, fvd))
        print(f"one by one: {(time.perf_counter() - t0) / a.days * 1e3:.2f} ms per day (host loop, downloads included); window ms_main {st['ms_main']:.2f}", flush=True)
    # the stream
    p.upload_forcing(nsteps, days[0], q0)
    p.stream_begin(nsteps, qts, slots=a.slots + a.later if a.slots else (a.later and 2 + -(-(int(p.lags()[0].max(initial=0)) + 1) // (nsteps // p.tile_steps)) + a.later), full_output=a.full and not a.stride, output_stride=a.stride)
    info = p.stream_info()
    print("stream:", info, flush=True)
    D = info["slots"]
    hyds = [_lib.result_empty((sample.shape[0], nsteps), np.float32, always_pinned=True) for _ in range(D)]
    fins = [_lib.result_empty((n, 3), np.float32, always_pinned=True) for _ in range(D)]
    keep = nsteps // a.stride if a.stride else nsteps
    fvds = [_lib.result_empty((n, keep, 3), np.float32, always_pinned=True) for _ in range(D)] if (a.stride or a.full) else [None] * D
    got = []
    behind = (info["lag_max"] + info["tiles_per_day"]) // info["tiles_per_day"] + a.later
    ok = True
    marks = []

    def take(e):
        global ok
        p.stream_wait(e)
        marks.append(time.perf_counter())
        if a.no_check:
            return
        fin, hyd, fvd = ref[e]
        s1 = np.array_equal(fin.view(np.uint32), fins[e % D].view(np.uint32))
        s2 = np.array_equal(hyd.view(np.uint32), hyds[e % D].view(np.uint32))
        s3 = fvd is None or np.array_equal(np.ascontiguousarray(fvd).view(np.uint32), fvds[e % D].view(np.uint32))
        if not (s1 and s2 and s3):
            ok = False
            print(f"   day {e}: final state {s1} hydrographs {s2} fvd {s3}", flush=True)
    t0 = time.perf_counter()
    for d in range(a.days):
        p.stream_push(pinned[d % len(days)], rowset=rs, hyd=hyds[d % D], q0=fins[d % D], fvd=fvds[d % D])
        e = d - behind
        if e >= 0:
            take(e)
    p.stream_flush()
    for e in range(max(0, a.days - behind), a.days):
        take(e)
    el = time.perf_counter() - t0
    p.stream_end()
    per = np.diff(marks) * 1e3
    print(f"stream: {el / a.days * 1e3:.2f} ms per day over {a.days} days (fill and drain included); between deliveries {np.round(per, 2).tolist()}", flush=True)
    print("   launches", p.stream_info()["launches"], " every day bit-identical to the days routed one by one:", ok if not a.no_check else "not checked", flush=True)
