"""Randomised parity sweep (developer tool, GPU box): both engines against the oracle's restatement, bit for bit.

    python tools/fuzz_parity.py --seconds 240 [--nseg 60000] [--seed 1]

Every round draws a random forest and one of several parameter regimes -- the CONUS-like ranges, the whole range the
fast divisions admit ([2**-14, 2**17], DevMathF::fast_ok in csrc/trmc.hip), ranges that straddle it (the plan then divides
plainly), degenerate channels (no flood plain, zero roughness of the flood plain, tw == bw), dry starts, depths from
1e-12 to 1e4, forcing with zeros and spikes -- routes a short window in both timestep modes on both engines and compares
every finite row with oracle.network_by_segment(det=True); each plan is run a second time ordered by the cost hint
of its first run.  Rows the oracle itself drives to NaN / inf must be NaN / inf
on the device too.  Prints one line per round and a total; exit code 1 on any difference.

The tests run one seed of two of these regimes (tests/test_gpu_parity.py::test_extreme_parameters_...); this is the wide
version for after a change to the segment step's control flow.
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
from oracle import oracle as O                                 # noqa: E402
from troute_amd.plan import RoutingPlan, csr_from_lists, topology_levels  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def draw(rng, nseg, regime):
    def logu(lo, hi, n):
        return np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    lo, hi = 2.0 ** -14, 2.0 ** 17
    if regime == "conus":
        bw = logu(0.5, 300.0, nseg)
        n = logu(0.02, 0.2, nseg)
        cs = logu(0.05, 5.0, nseg)
        s0 = logu(1e-5, 0.3, nseg)
        dx = logu(50.0, 2e4, nseg)
        depth = logu(1e-3, 20.0, nseg)
    elif regime == "admitted":
        bw = logu(lo * 1.01, hi * 0.2, nseg)
        n = logu(lo * 1.01, 0.5, nseg)
        cs = logu(lo * 1.01, hi * 0.99, nseg)
        s0 = logu(2.0 ** -30 * 1.01, 2.0 ** 10 * 0.99, nseg)    # (the whole range the fast Muskingum K admits: DevMathF::k_of)
        dx = logu(2.0 ** -10 * 1.01, 2.0 ** 19 * 0.99, nseg)
        depth = logu(1e-12, 1e4, nseg)
    elif regime == "straddle":
        bw = logu(lo * 0.25, hi * 4.0, nseg)
        n = logu(lo * 0.25, 4.0, nseg)
        cs = logu(lo * 0.25, hi * 4.0, nseg)
        s0 = logu(1e-9, 10.0, nseg)
        dx = logu(0.1, 1e7, nseg)
        depth = logu(1e-20, 1e6, nseg)
    else:  # "degenerate"
        bw = logu(0.01, 500.0, nseg)
        n = logu(0.005, 1.0, nseg)
        cs = np.where(rng.random(nseg) < 0.3, 0.0, logu(0.01, 100.0, nseg))
        s0 = logu(1e-6, 1.0, nseg)
        dx = logu(5.0, 1e5, nseg)
        depth = np.where(rng.random(nseg) < 0.3, 0.0, logu(1e-6, 100.0, nseg))
    u = rng.random(nseg)
    tw = np.where(u < 0.15, bw, bw * rng.uniform(1.0, 3.0, nseg))
    twcc = np.where(rng.random(nseg) < 0.2, 0.0, tw * rng.uniform(1.0, 4.0, nseg))
    ncc = np.where(rng.random(nseg) < 0.15, 0.0, n * rng.uniform(1.0, 3.0, nseg))
    dt = 300.0 if rng.random() < 0.7 else float(rng.choice([10.0, 60.0, 3600.0]))
    params = np.stack([np.full(nseg, dt), dx, bw, tw, twcc, n, ncc, cs, s0], 1).astype(np.float32)
    nq = int(rng.integers(1, 4))
    qlat = logu(1e-9, 50.0, (nseg, nq)) * (rng.random((nseg, nq)) > 0.3)
    spikes = rng.random((nseg, nq)) < 0.01
    qlat = np.where(spikes, qlat * 1e3, qlat).astype(np.float32)
    q0 = np.stack([logu(1e-9, 500.0, nseg), logu(1e-9, 500.0, nseg), depth], 1)
    q0[rng.random(nseg) < 0.1] = 0.0
    return params, qlat, q0.astype(np.float32), nq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--nseg", type=int, default=60000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    O.build()
    t_end = time.time() + a.seconds
    rounds = bad_rounds = total = 0
    seed = a.seed
    regimes = ["conus", "admitted", "straddle", "degenerate"]
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        regime = regimes[seed % len(regimes)]
        nseg = int(a.nseg * rng.uniform(0.3, 1.0))
        to = H.random_network(rng, nseg)
        _, _, ups = H.reaches_from_to(to)
        up_ptr, up_idx = csr_from_lists(ups)
        lvl, _, _ = topology_levels(up_ptr, up_idx)
        params, qlat, q0, nq = draw(rng, nseg, regime)
        qts = int(rng.integers(1, 6))
        nsteps = nq * qts - int(rng.integers(0, qts))      # the last forcing column possibly part used
        for short in (True, False):
            want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, short, det=True)[:, 1:, :]
            fin = np.isfinite(want).all(axis=(1, 2))
            # (label, engine, settings for the run): the level engine's one-step launches, the dataflow engine, and -- short
            # timesteps only -- the wide levels K steps per launch (k_mc_tile; on the cost-ordered plan with the rows below
            # them sorted by cost across levels and the tile kernel dealing its rows to threads by class) and a
            # second tier of tiles below them, both with thresholds small enough for these networks to take them
            variants = [("levels", "levels", {"TRMC_WIDE_MIN_ROWS": "0"}), ("flow", "flow", {})]
            if short:
                variants.append(("levels-wide", "levels", {"TRMC_WIDE_MIN_ROWS": "32", "TRMC_WIDE_K": str(int(rng.choice([3, 4, 8, 16]))),
                                                           "TRMC_WIDE_LEVELS": str(int(rng.choice([2, 5, 16]))),
                                                           "TRMC_TILE_PERM": str(int(rng.choice([0, 256, 512, 1024]))),
                                                           "TRMC_HOT_ROWS": str(int(rng.choice([0, 1, 1])))}))
                variants.append(("levels-mid", "levels", {"TRMC_WIDE_MIN_ROWS": "64", "TRMC_WIDE_K": str(int(rng.choice([4, 8, 16]))),
                                                          "TRMC_MID_MIN_ROWS": str(int(rng.choice([2, 8, 16]))),
                                                          "TRMC_MID_K": str(int(rng.choice([1, 2, 3, 4]))),
                                                          "TRMC_MID_LEVELS": str(int(rng.choice([3, 12, 32])))}))
            for label, engine, env in variants:
                saved = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=short, engine=engine) as plan:
                        got = plan.route(nsteps, qts, short, qlat, q0)
                        it = plan.download_iterations()
                    # the same window on the plan ordered by the measured cost of every row
                    with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=short, engine=engine,
                                     cost_hint=np.minimum(it, 3)) as plan:
                        got2 = plan.route(nsteps, qts, short, qlat, q0)
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                engine = label
                ok = np.array_equal(bits(got[fin]), bits(want[fin])) and \
                    np.array_equal(np.isfinite(got).all(axis=(1, 2)), fin) and \
                    np.array_equal(bits(got2[fin]), bits(want[fin]))
                total += 2 * int(fin.sum()) * nsteps
                if not ok:
                    bad_rounds += 1
                    diff = np.argwhere(bits(got[fin]) != bits(want[fin]))
                    print(f"DIFF seed={seed} regime={regime} short={short} engine={engine}: {diff.shape[0]} values, "
                          f"first {diff[:3].tolist()}", flush=True)
        rounds += 1
        print(f"seed {seed:4d} {regime:10s} nseg {nseg:6d} steps {nsteps:2d} qts {qts} finite {fin.mean():.4f} "
              f"differing runs so far {bad_rounds}", flush=True)
        seed += 1
    print(f"fuzz_parity: {rounds} rounds, {total} finite segment-steps compared (one-step launches, dataflow engine, wide tiles with and "
          f"without a second tier x both modes where they apply x plain and cost-ordered plan), {bad_rounds} differing runs")
    sys.exit(1 if bad_rounds else 0)


if __name__ == "__main__":
    main()
