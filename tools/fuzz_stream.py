"""Randomised parity sweep of the STREAM of windows (developer tool, GPU box): trmc_stream_* on random forests against the
oracle's restatement day by day, bit for bit.

    python tools/fuzz_stream.py --seconds 240 [--nseg 8000] [--seed 1]

Every round draws a forest, a plan layout (steps per launch, rows per cluster, where the slices end, hot rows on / off, a cost
hint or none), what the stream hands over (the full result, every n-th step, or hydrographs and states only; velocities formed on
demand or always), a number of days and of ring slots, routes the days as one stream and compares every product of every day with
oracle.network_by_segment(det=True) carried on from day to day.  Prints one line per round; exit code 1 on any difference."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H                                            # noqa: E402
from oracle import oracle as O                                 # noqa: E402
from troute_amd import _lib                                    # noqa: E402
from troute_amd.plan import RoutingPlan, csr_from_lists        # noqa: E402
from troute_amd.sequence import pinned_like                    # noqa: E402
from test_gpu_parity import synth_inputs                       # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--nseg", type=int, default=8000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    rounds = bad = 0
    while time.time() < t_end:
        nseg = int(rng.integers(200, a.nseg))
        to = H.random_network(rng, nseg)
        _, _, ups = H.reaches_from_to(to)
        up_ptr, up_idx = csr_from_lists(ups)
        K = int(rng.choice([2, 4, 8, 16]))
        nsteps = K * int(rng.integers(1, 7))
        qts = int(rng.choice([1, 4, 12, 16]))
        nq = (nsteps - 1) // qts + 1 + int(rng.integers(0, 2))
        params, qlat, q0 = synth_inputs(rng, nseg, nq)
        if rng.random() < 0.3:                                   # (floods: over bank, many iterations -- the hot rows' food)
            qlat = qlat.copy()
            qlat[rng.uniform(0, 1, nseg) < 0.1] *= np.float32(300.0)
        ndays = int(rng.integers(1, 9))
        days = [(qlat * np.float32(rng.uniform(0.2, 2.5))).astype(np.float32) for _ in range(ndays)]
        mode = rng.choice(["full", "stride", "products"])
        divisors = [n for n in range(1, nsteps + 1) if nsteps % n == 0]
        stride = int(rng.choice(divisors)) if mode == "stride" else 0
        opts = {"cluster_rows": int(rng.choice([4, 8, 24, 64, 128])), "wide_min_rows": int(rng.choice([-1, 16, 64, 400])), "wide_k": K,
                "hot_rows": int(rng.choice([-1, 1])), "velocity_on_demand": int(rng.integers(0, 2)),
                "wide_levels": int(rng.choice([0, 2, 8]))}
        hint = rng.integers(0, 4, nseg).astype(np.uint8) if rng.random() < 0.5 else None
        slots = int(rng.choice([0, 0, 12]))
        rows = np.sort(rng.choice(nseg, min(nseg, 50), replace=False))
        ok = True
        with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=True, engine="levels", cost_hint=hint, options=opts) as p:
            lvl, _ = p.levels()
            rs = p.rowset(rows)
            p.upload_forcing(nsteps, days[0], q0)
            p.stream_begin(nsteps, qts, slots=slots, full_output=mode == "full", output_stride=stride)
            info = p.stream_info()
            D = info["slots"]
            keep = nsteps // stride if stride else nsteps
            hyds = [_lib.result_empty((rows.shape[0], nsteps), np.float32, always_pinned=True) for _ in range(D)]
            fins = [_lib.result_empty((nseg, 3), np.float32, always_pinned=True) for _ in range(D)]
            fvds = [_lib.result_empty((nseg, keep, 3), np.float32, always_pinned=True) if mode != "products" else None for _ in range(D)]
            behind = (info["lag_max"] + info["tiles_per_day"]) // info["tiles_per_day"]
            state = q0
            checked = 0

            def check(e):
                nonlocal state, ok, checked
                p.stream_wait(e)
                want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, state, days[e], True, det=True)[:, 1:, :]
                k = e % D
                s1 = np.array_equal(bits(hyds[k]), bits(want[rows, :, 0]))
                fin = np.stack([want[:, -1, 0], want[:, -1, 0], want[:, -1, 2]], 1)
                s2 = np.array_equal(bits(fins[k][:, [0, 2]]), bits(fin[:, [0, 2]]))
                s3 = True
                if mode == "full":
                    s3 = np.array_equal(bits(fvds[k]), bits(want))
                elif mode == "stride":
                    s3 = np.array_equal(bits(fvds[k]), bits(want[:, stride - 1::stride, :]))
                if not (s1 and s2 and s3):
                    ok = False
                    print(f"   day {e}: hydrographs {s1} final state {s2} block {s3}", flush=True)
                state = fin
                checked += 1
            for d in range(ndays):
                p.stream_push(pinned_like(days[d]), rowset=rs, hyd=hyds[d % D], q0=fins[d % D], fvd=fvds[d % D])
                if d - behind >= 0:
                    check(d - behind)
            p.stream_flush()
            for e in range(max(0, ndays - behind), ndays):
                check(e)
            p.stream_end()
        rounds += 1
        bad += 0 if ok else 1
        print(f"round {rounds}: nseg {nseg} nsteps {nsteps} K {K} qts {qts} days {ndays} {mode}{'/' + str(stride) if stride else ''} "
              f"{opts} hint {hint is not None} slots {D} lag {info['lag_max']} ({info['wide_levels']} slices + {info['cluster_levels']}): "
              f"{'ok' if ok else 'DIFFERENT'}", flush=True)
    print(f"{rounds} rounds, {bad} with differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
