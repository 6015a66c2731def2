"""Randomised parity sweep of the diffusive solver (developer tool, GPU box): the parallel time loop against the host
instantiation of the same source (oracle/libdw_oracle.so -- itself pinned to the reference Fortran on the goldens), bit for
bit, on synthetic mainstems the goldens do not reach.

    python tools/fuzz_diffusive.py --seconds 240 [--seed 1]

Every round draws a domain from tests/test_diffusive.py::long_mainstem -- 3 to 260 mainstem reaches in series with a side
branch and tributary junctions, so both chain-state homes are taken (LDS for the short ones, global memory for the long) --
then scales flows (x 0.05 ... x 40: from nearly dry to far over bank, so that the table windows of the depth solve wander
and are re-centred), roughness and the tributary hydrographs, and every fourth round forces the narrow 5-row windows or the
global-memory chain state (TRDW_WINDOW_ROWS / TRDW_CHAIN_GLOBAL).  Exit code 1 on any differing bit.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_diffusive as TD                                        # noqa: E402
from troute_amd.routing.fast_reach import diffusive as D           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    host = TD.host_oracle()
    t_end = time.time() + a.seconds
    seed, rounds, bad, nodes_total, nonfinite = a.seed, 0, 0, 0, 0
    while time.time() < t_end:
        rng = np.random.default_rng(10_000 + seed)
        nmain = int(rng.choice([3, 5, 8, 13, 21, 40, 80, 150, 220, 260]))
        ins = TD.long_mainstem(nmain=nmain, seed=seed)
        scale = float(np.exp(rng.uniform(np.log(0.05), np.log(40.0))))
        ins["iniq"] = ins["iniq"] * scale
        ins["qtrib_g"] = ins["qtrib_g"] * scale * float(rng.uniform(0.5, 3.0))
        ins["qlat_g"] = ins["qlat_g"] * scale
        f = float(rng.uniform(0.6, 2.5))
        ins["mann_ar_g"] = ins["mann_ar_g"] * f
        ins["manncc_ar_g"] = ins["manncc_ar_g"] * f
        env = {}
        if seed % 4 == 1:
            env["TRDW_WINDOW_ROWS"] = "5"
        elif seed % 4 == 3:
            env["TRDW_CHAIN_GLOBAL"] = "1"
        rc, want = TD.call_c(host, "dw_oracle_diffnw", ins)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            got = D.compute_diffusive(ins)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        ok = rc == 0 and all(TD.same_bits(g, w) for g, w in zip(got, want))
        fin = bool(np.isfinite(want[0]).all())
        nonfinite += 0 if fin else 1
        nodes = int(ins["frnw_g"][:, 0].sum())
        nodes_total += nodes
        rounds += 1
        bad += 0 if ok else 1
        print(f"seed {seed:4d} reaches {int(ins['nrch_g']):4d} nodes {nodes:5d} flow x{scale:7.3f} n x{f:4.2f} "
              f"{' '.join(f'{k}={v}' for k, v in env.items()) or '-':22s} depth max {np.nanmax(want[2]):7.2f} "
              f"finite {fin} {'ok' if ok else 'DIFF'}", flush=True)
        seed += 1
    print(f"fuzz_diffusive: {rounds} domains, {nodes_total} nodes, {nonfinite} with non-finite results in the host run, "
          f"{bad} differing")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
