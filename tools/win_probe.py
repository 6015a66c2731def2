#!/usr/bin/env python3
"""Developer probe of the window kernel (k_mc_window) on the CONUS day: one process, one unhinted or hinted plan, the
run-time knobs varied window by window (they are read at trmc_route_begin).

    python tools/win_probe.py [--hint] "TRMC_WIN_K=8 TRMC_WIN_LEVELS=24" "TRMC_WINDOW=0" ...

Prints ms_main (HIP events around the window) per setting; with TRMC_LIB_PATH=<debug build> the library's own tallies
appear on stderr."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from troute_amd import synthetic  # noqa: E402
from troute_amd.distributed import ShardedRouter  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
hinted = "--hint" in sys.argv
reps = 3
net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"))
to, params = net["to"], net["params"]
nseg = to.shape[0]
nsteps, qts = 288, 12
qlat_s = net["qlat"]
qlat_a = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 1, previous=qlat_s)
qlat_b = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2, previous=qlat_a)
q0 = np.zeros((nseg, 3), np.float32)


def spin(r):
    r.upload(nsteps, qlat_s, q0)
    r.route_resident(qts, True)
    r.upload(nsteps, qlat_a, None)


os.environ["TRMC_WINDOW"] = "0"
r = ShardedRouter(to, params, assume_short_ts=True)
hint = None
if hinted:
    spin(r)
    r.collect_cost(True)
    r.route_resident(qts, True)
    hint = r.iteration_hint()
    r.collect_cost(False)
    r.close()
    r = ShardedRouter(to, params, assume_short_ts=True, cost_hint=hint)
spin(r)
r.route_resident(qts, True)
r.upload(nsteps, qlat_b, None)
r.route_resident(qts, True)
ref_state = r.plan0.download_final_state()
for setting in args or [""]:
    saved = {}
    for kv in setting.split():
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    os.environ.setdefault("TRMC_WINDOW", "1")
    if "TRMC_WINDOW" not in saved:
        saved["TRMC_WINDOW"] = os.environ.get("TRMC_WINDOW")
        os.environ["TRMC_WINDOW"] = "1"
    ms, wall = [], []
    st = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r.route_resident(qts, True)
        wall.append((time.perf_counter() - t0) * 1e3)
        st = r.last_stats["phase0"]
        ms.append(st["ms_main"])
    same = bool(np.array_equal(r.plan0.download_final_state().view(np.uint32), ref_state.view(np.uint32)))
    print(f"{setting or '(defaults)':60s} ms_main {min(ms):8.2f} (max {max(ms):8.2f}) wall {min(wall):8.2f}  window_kernel={st['window_kernel']} "
          f"W={st['wide_levels']} K={st['wide_k']} launches={st['main_launches']} same_bits={same}", flush=True)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
r.close()
