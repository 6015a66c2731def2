"""Randomised stress of the sequence pipeline (developer tool, GPU box): the order of bench.py's timed pass -- a plan and its
clone taking turns, every day's forcing staged two days ahead from page-locked memory (trmc_stage_forcing), the state
handed over on the device (trmc_plan_chain_from), the products fetched asynchronously with the window
(trmc_fetch_begin / trmc_fetch_wait; in half of the rounds with every n-th step of every row's (q, v, d) among them,
trmc_fetch_begin_fvd) -- against the same days routed one after the other on ONE plan with synchronous
uploads, bit for bit: every day's outlet hydrographs and final state.

    python tools/fuzz_sequence.py --seconds 300 [--nseg 200000] [--seed 1]

What it is after is ORDERING, not arithmetic (tools/fuzz_parity.py and the tests pin that to the oracle): a copy that
overtakes a kernel, a window that overwrites planes a gather still reads, a staging area refilled too early.  Every round
draws a forest (1/20 .. 1 of --nseg rows), 3-9 days, a window of 8-96 steps, the forcing's subdivision, the wide-level
thresholds (so that some rounds take k_mc_tile with a tail beside it and some the one-step launches alone), K, and
whether the host dawdles between the calls (which moves the moment a call is made relative to the device's progress).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H                                            # noqa: E402
from troute_amd import _lib                                    # noqa: E402
from troute_amd.plan import RoutingPlan, csr_from_lists        # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def one_round(rng, nseg_max):
    nseg = int(nseg_max * rng.uniform(0.05, 1.0))
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    # dt dx bw tw twcc n ncc cs s0 (trmc.h column order); the flood plain three times the channel's top width
    params = np.stack([np.full(nseg, 300.0), rng.uniform(300, 3000, nseg), rng.uniform(1, 9, nseg), np.zeros(nseg), np.zeros(nseg),
                       np.full(nseg, 0.06), np.full(nseg, 0.12), rng.uniform(0.2, 1.5, nseg), rng.uniform(1e-3, 2e-2, nseg)], 1)
    params[:, 3] = params[:, 2] * 5 / 3
    params[:, 4] = 3 * params[:, 3]
    params = params.astype(np.float32)
    qts = int(rng.choice([1, 2, 3, 4, 6, 12]))
    nq = int(rng.integers(2, 9))
    nsteps = nq * qts
    ndays = int(rng.integers(3, 10))
    q0 = np.zeros((nseg, 3), np.float32)
    q0[:, 0] = q0[:, 1] = rng.uniform(0, 2.0, nseg).astype(np.float32)
    q0[:, 2] = rng.uniform(0, 0.5, nseg).astype(np.float32)
    days = []
    for _ in range(ndays):
        d = _lib.result_empty((nseg, nq), np.float32, always_pinned=True)
        d[...] = rng.uniform(0, 0.6, (nseg, nq)).astype(np.float32) * (rng.uniform(0, 1, (nseg, 1)) < 0.8)
        days.append(d)
    outlets = np.flatnonzero(to < 0)
    env = {"TRMC_SETUP_ASIDE": "1", "TRMC_ENGINE": "levels",
           "TRMC_WIDE_MIN_ROWS": str(int(rng.choice([0, 32, 512, 4096]))),
           "TRMC_WIDE_K": str(int(rng.choice([2, 4, 8, 16]))),
           "TRMC_WIDE_LEVELS": str(int(rng.choice([2, 5, 16]))),
           "TRMC_MID_MIN_ROWS": str(int(rng.choice([0, 0, 8, 64]))),      # (a second tier of tiles below the wide levels)
           "TRMC_MID_K": str(int(rng.choice([1, 2, 4]))), "TRMC_MID_LEVELS": str(int(rng.choice([3, 12, 32]))),
           # hot rows (the few rows of three or more iterations in blocks of their own): off / on, also on hinted plans
           "TRMC_HOT_ROWS": str(int(rng.choice([0, 1, 1])))}
    dawdle = float(rng.choice([0.0, 0.0, 0.002, 0.01]))
    # every stride-th step of every row among each day's products (trmc_fetch_begin_fvd), in half of the rounds
    stride = int(rng.choice([0, 0, 1, 2, 3, qts, 7]))
    stride = stride if 0 < stride <= nsteps else None
    hinted = bool(rng.integers(0, 2))
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        hint = None
        want_h, want_s, want_f = [], [], []
        with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=True) as ref:
            for w in range(ndays):
                ref.upload_forcing(nsteps, days[w], q0 if w == 0 else None)
                ref.route_device(nsteps, qts, True)
                want_h.append(ref.gather_flow_rows(outlets))
                want_s.append(ref.download_final_state())
                want_f.append(None if stride is None else ref.download_fvd()[:, stride - 1::stride][:, :nsteps // stride].copy())
            if hinted:
                hint = np.minimum(ref.download_iterations(), 3)
        with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=True, cost_hint=hint) as a:
            b = a.clone()
            plans = [a, b]
            rs = [pl.rowset(outlets) for pl in plans]
            # ... in two of three of those the windows write the kept steps aside as they go (trmc_plan_set_output_stride),
            # now and then told another stride than the fetch asks for (the fetch then decimates the result itself)
            told = {0: 0, 1: stride, 2: stride, 3: (stride or 0) + 1}[int(rng.integers(0, 4))] if stride else 0
            for pl in plans:
                pl.set_output_stride(told)

            def nap():
                if dawdle:
                    time.sleep(dawdle * rng.uniform(0, 1))
            a.upload_forcing(nsteps, days[0], q0)
            a.route_begin(nsteps, qts, True)
            a.route_advance(nsteps)
            if ndays > 1:
                b.stage_forcing(nsteps, days[1])
            a.fetch_begin(rs[0], True, stride)
            if ndays > 2:
                a.stage_forcing(nsteps, days[2])
            got = []
            for w in range(1, ndays):
                cur, prev = plans[w % 2], plans[(w - 1) % 2]
                nap()
                cur.chain_from(prev)
                cur.route_begin(nsteps, qts, True)
                cur.route_advance(nsteps)
                nap()
                cur.fetch_begin(rs[w % 2], True, stride)
                if w + 2 < ndays:
                    cur.stage_forcing(nsteps, days[w + 2])
                nap()
                prev.route_end()
                h, st, *f = prev.fetch_wait()
                got.append((h.copy(), st.copy(), f[0].copy() if f else None))
            last = plans[(ndays - 1) % 2]
            last.route_end()
            h, st, *f = last.fetch_wait()
            got.append((h.copy(), st.copy(), f[0].copy() if f else None))
            wide = a.stats().get("wide_levels", 0)
            b.close()
        bad = [w for w, (h, st, f) in enumerate(got)
               if not (np.array_equal(bits(h), bits(want_h[w])) and np.array_equal(bits(st), bits(want_s[w]))
                       and (stride is None or (f.shape == want_f[w].shape and np.array_equal(bits(f), bits(want_f[w])))))]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return nseg, ndays, nsteps, qts, wide, hinted, dawdle, stride, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--nseg", type=int, default=200000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    t_end = time.time() + a.seconds
    rounds = bad_rounds = days_total = 0
    seed = a.seed
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        nseg, ndays, nsteps, qts, wide, hinted, dawdle, stride, bad = one_round(rng, a.nseg)
        rounds += 1
        days_total += ndays
        if bad:
            bad_rounds += 1
        print(f"seed {seed:4d} nseg {nseg:7d} days {ndays} steps {nsteps:3d} qts {qts:2d} wide levels {wide:2d} hinted {int(hinted)} "
              f"dawdle {dawdle:.3f} stride {stride}  {'DIFFERENT days ' + str(bad) if bad else 'identical'}", flush=True)
        seed += 1
    print(f"fuzz_sequence: {rounds} rounds, {days_total} days, {bad_rounds} rounds with a day that differs")
    sys.exit(1 if bad_rounds else 0)


if __name__ == "__main__":
    main()
