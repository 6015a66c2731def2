#!/usr/bin/env python3
"""Headline benchmark: Muskingum-Cunge routing of a CONUS-scale network on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input: routing the
whole seeded synthetic CONUS network (2 729 077 segments, 14 713 independent networks,
troute_amd/synthetic.py) for one forcing window of 288 x 300 s timesteps with the
reference's configured assume_short_ts=True (test/LowerColorado_TX/test_AnA.yaml:32),
fp32 (the reference's arithmetic type), cold start -- forcing and topology already
resident in HBM when the timed region starts, results (incl. the gathered outlet hydrographs) left in HBM in the reference's
[segment][timestep][q,v,d] layout.

Prints ONE JSON line: BASELINE.json's metric (segment-timesteps/s) plus
  roofline      dominant kernel (k_mc_step) against the 8 TB/s HBM roofline, timed with
                HIP events on the plan's own stream inside this run
  cpu_baseline  the reference Fortran kernel (oracle/_ref, amdflang -O2) driven by the
                restated reference loop on this host's cores, bounded sample
  full_ts       the same workload without the short-timestep assumption (level wavefront)
"""
import argparse
import json
import os
import sys
import time

# Before any HIP runtime is loaded: at most two hardware queues for ordinary-priority streams (the exchange stream,
# RCCL's).  With the ROCm 7.2 default of four, a stream of another hardware queue waiting on events of the plan's
# stream intermittently put that stream's launches in a slow mode (17 ms instead of 6.4 ms per 340 k-row rank,
# depending on which queue the waiting stream happened to get: tools/sim_ranks.py); one or two
# queues never did, and the single-GPU path (priority streams only) is unaffected either way.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SEGSTEP = 64          # 32 params + 4 qlat + 8 own state + 8 upstream + 12 out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseg", type=int, default=None, help="override network size (default CONUS)")
    ap.add_argument("--nnet", type=int, default=None)
    ap.add_argument("--nsteps", type=int, default=288)
    ap.add_argument("--qts", type=int, default=12)
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--chunks", type=int, default=None, help="time chunks of the multi-GPU hand-off pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-ts", action="store_true")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm-start variant of the day")
    ap.add_argument("--no-retune", action="store_true",
                    help="keep the plain topological plan order (skip the untimed tuning window and plan rebuild)")
    ap.add_argument("--no-diffusive", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    return ap.parse_args()


def cpu_baseline(net, nsteps, qts, short_ts, target_s):
    """Reference kernel on host cores over a bounded sample of the SAME workload.

    Sample = independent networks other than the dominant basin, smallest-cost subset
    sized so the run lasts about `target_s` s; one network loop call per worker thread
    (the reference's by-network parallelism, compute.py:1211-1395; ctypes drops the GIL).
    """
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    from troute_amd import sharding
    from troute_amd.plan import topology_levels
    from troute_amd.synthetic import upstream_csr

    kind, ref_name = "port", None
    if O.have_ref("libmc_ref_f32.so"):
        kind, ref_name = "reference", "libmc_ref_f32.so"
    cores = os.cpu_count() or 1
    to = net["to"]
    nseg = to.shape[0]
    outlet = sharding.outlet_of(to)
    uniq, lab = np.unique(outlet, return_inverse=True)
    sizes = np.bincount(lab)
    per_thread = int(1.5e6 * target_s / nsteps)              # ~1.5 M seg-steps/s/core
    order = np.argsort(sizes, kind="stable")                  # small networks first
    order = order[sizes[order] <= max(per_thread, int(sizes.min()))]   # a network never exceeds one thread's budget
    take = order[np.cumsum(sizes[order]) <= per_thread * cores]
    if take.size == 0:
        take = order[:1]
    part, _ = sharding.lpt_assign(sizes[take], cores)
    net_worker = np.full(uniq.shape[0], -1, dtype=np.int64)
    net_worker[take] = part
    row_worker = net_worker[lab]
    up_ptr, up_idx = upstream_csr(to)
    from troute_amd.distributed import restrict_csr

    jobs = []
    for w in range(cores):
        rows = np.flatnonzero(row_worker == w)
        if rows.size == 0:
            continue
        g2l = np.full(nseg, -1, dtype=np.int64)
        g2l[rows] = np.arange(rows.shape[0])
        lp, li = restrict_csr(up_ptr, up_idx, rows, g2l)
        lvl, _, _ = topology_levels(lp, li)
        jobs.append((lp, li, lvl, np.ascontiguousarray(net["params"][rows]),
                     np.zeros((rows.shape[0], 3), np.float32), np.ascontiguousarray(net["qlat"][rows])))

    def run(j):
        lp, li, lvl, par, q0, ql = j
        O.network_by_segment(nsteps, qts, lp, li, lvl, par, q0, ql, short_ts, ref_name=ref_name)
        return par.shape[0]

    O.lib()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        done = sum(ex.map(run, jobs))
    dt = time.perf_counter() - t0
    return {
        "value": done * nsteps / dt, "unit": "segment-timesteps/s", "cores": cores, "kind": kind,
        "sample": f"{int(take.size)} of {int(uniq.shape[0])} independent networks ({done} segments; networks larger than "
                  f"{per_thread} segments, incl. the dominant basin, left out) x "
                  f"{nsteps} steps, {dt:.1f} s wall, {len(jobs)} threads, by-network parallelism",
    }


def _diffusive_inputs(gold, nsteps):
    z = np.load(gold)
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    ins["timestep_ar_g"] = ins["timestep_ar_g"].copy()
    ins["timestep_ar_g"][2] = 300.0 * nsteps / 3600.0
    ins["ntss_ev_g"] = np.array(nsteps + 1)
    return ins


def _diffusive_call(gold, nsteps, path, sym):
    """c_diffnw's argument list on a host library (the reference build or the host restatement): (seconds, outputs)."""
    import ctypes as C
    from troute_amd.routing.fast_reach import diffusive as D
    ins = _diffusive_inputs(gold, nsteps)
    lib = C.CDLL(path)
    keep, args = [], []
    for k in D.ARG_ORDER:
        v = ins[k]
        if k in D._INT_SCALARS:
            c = C.c_int(int(v))
            keep.append(c)
            args.append(C.byref(c))
        else:
            arr = np.asfortranarray(v, dtype=np.int32 if k in D._INT_ARRAYS else np.float64)
            if arr.size == 0:
                arr = np.zeros(1, dtype=arr.dtype)
            keep.append(arr)
            args.append(arr.ctypes.data_as(C.c_void_p))
    shape = (nsteps + 1, int(ins["mxncomp_g"]), int(ins["nrch_g"]))
    outs = [np.zeros(shape, dtype=np.float64, order="F") for _ in range(3)]
    args += [o.ctypes.data_as(C.c_void_p) for o in outs]
    t0 = time.perf_counter()
    getattr(lib, sym)(*args)
    return time.perf_counter() - t0, [np.ascontiguousarray(o) for o in outs]


def diffusive_leg(nsteps=12):
    """Side measurement (BASELINE configs[4]): the diffusive-wave mainstem of the LowerColorado
    coastal subset on the GPU (trdw_diffnw) beside the reference Fortran diffnw on one host core (oracle/_ref, built from
    the reference's own sources) and the host instantiation of the same restatement; inputs = the committed golden
    (marshalled by the reference's diffusive_input_data_v02), shortened to `nsteps` x 300 s."""
    gold = os.path.join(ROOT, "tests", "golden", "diffusive_lowercolorado.npz")
    ins = _diffusive_inputs(gold, nsteps)
    from troute_amd.routing.fast_reach import diffusive as D
    D.compute_diffusive(ins)                                   # warm-up (module load)
    t0 = time.perf_counter()
    got = D.compute_diffusive(ins)
    gpu_s = time.perf_counter() - t0
    tables_ms, solve_ms = D.last_timing()
    out = {"workload": f"LowerColorado_TX coastal subset (hybrid config): {int(ins['nrch_g'])} reaches, "
                       f"{int((ins['frnw_g'] == 555).sum())} diffusive, {nsteps} x 300 s, synthetic cross sections, fp64",
           "gpu_s": gpu_s, "gpu_tables_ms": tables_ms, "gpu_solve_ms": solve_ms}

    # many domains in one launch (one compute unit each): 64 copies of the domain stand in for 64 tailwaters
    nb = 64
    t0 = time.perf_counter()
    many = D.compute_diffusive_batch([ins] * nb)
    out["batch"] = {"domains": nb, "gpu_s": time.perf_counter() - t0, "gpu_solve_ms": D.last_timing()[1],
                    "identical_to_single": bool(all(np.array_equal(m[0], got[0]) and np.array_equal(m[2], got[2]) for m in many))}
    del many
    host = os.path.join(ROOT, "oracle", "libdw_oracle.so")
    if os.path.exists(host):
        out["host_restatement_s"], h = _diffusive_call(gold, nsteps, host, "dw_oracle_diffnw")
        out["gpu_bit_identical_to_host_restatement"] = bool(all(np.array_equal(np.asarray(g), np.asarray(w)) for g, w in zip(got, h)))
    ref = os.path.join(ROOT, "oracle", "_ref", "libdiff_ref.so")
    if os.path.exists(ref):
        # in a child process: the Fortran runtime prints its progress to stdout and flushes it at exit
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            child = (
                "import sys, time, ctypes as C, numpy as np\n"
                f"sys.path.insert(0, {ROOT!r})\n"
                "import bench\n"
                f"t, outs = bench._diffusive_call({gold!r}, {nsteps}, {ref!r}, 'c_diffnw')\n"
                f"np.savez({os.path.join(td, 'r.npz')!r}, t=t, q=outs[0], e=outs[1], d=outs[2])\n")
            subprocess.run([sys.executable, "-c", child], stdout=subprocess.DEVNULL, check=True)
            r = np.load(os.path.join(td, "r.npz"))
            out["reference_fortran_s"] = float(r["t"])
            out["reference_fortran_cores"] = 1
            out["gpu_bit_identical_to_reference"] = bool(all(np.array_equal(np.asarray(g), r[k]) for g, k in zip(got, "qed")))
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or bool(os.environ.get("TRMC_FORCE_DIST"))   # the env var exercises the RCCL path at N=1
    # TRMC_BENCH_BACKEND=gloo: a rehearsal of the multi-rank control flow on a box with ONE GPU -- every rank on device
    # 0, the collectives through host memory.  Not a measurement.
    backend = os.environ.get("TRMC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    from troute_amd import _lib, synthetic
    from troute_amd.distributed import ShardedRouter

    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libtrmc.so has no CPU fallback")

    kw = {}
    if a.nseg:
        kw["nseg"] = a.nseg
        kw["nnet"] = a.nnet or max(3, a.nseg // 185)
    cache = os.environ.get("TRMC_CACHE", "/tmp/trmc_cache")
    t0 = time.perf_counter()
    if rank == 0:
        net = synthetic.generate(cache_dir=cache, **kw)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        net = synthetic.generate(cache_dir=cache, **kw)
    t_gen = time.perf_counter() - t0
    to, params, qlat = net["to"], net["params"], net["qlat"]
    nseg = to.shape[0]
    q0 = np.zeros((nseg, 3), dtype=np.float32)

    def all_gather_into(out, t):
        """out[world, *t.shape] <- every rank's t, over RCCL (xGMI), ordered against the current stream"""
        if backend == "nccl":
            dist.all_gather_into_tensor(out, t)
        else:
            import torch
            torch.cuda.current_stream().synchronize()
            mine = t.cpu()
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            out.copy_(torch.stack(parts).to(out.device))

    def make_router(hint):
        r = ShardedRouter(to, params, rank=rank, world=world, device=local_rank, precision=a.precision, cost_hint=hint)
        r.upload(a.nsteps, qlat, q0)
        if use_dist:
            import torch
            r.enable_device_exchange(torch, torch.device("cuda", local_rank))
            r.upload_trunk()
        return r

    t0 = time.perf_counter()
    router = make_router(None)
    t_plan = time.perf_counter() - t0

    def route_once(short_ts):
        if use_dist:   # every hand-off stays in HBM: gather kernels -> RCCL all-gather -> boundary rows
            return router.route_on_device(a.qts, short_ts, all_gather_into, a.chunks)
        return router.route_resident(a.qts, short_ts), None   # outlet hydrographs stay in HBM

    # Plan tuning, outside the timed region: one window on the plain plan tells which rows are cheap (dry channel:
    # one secant iteration) and which are not; the plan is rebuilt with that as its cost hint, so that the rows of a
    # level sit grouped by cost and the step kernel's wavefronts are uniform.  Same results (tests/test_gpu_parity.py).
    t_tune = 0.0
    if not a.no_retune:
        t0 = time.perf_counter()
        router.collect_cost(True)
        route_once(True)
        hint = router.iteration_hint()
        router.close()
        router = make_router(hint)
        t_tune = time.perf_counter() - t0

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def timed(short_ts, steps, warmup):
        for _ in range(warmup):
            route_once(short_ts)
        sync()
        t0 = time.perf_counter()
        mains, totals, launches = [], [], 0
        for _ in range(steps):
            rows, hyd = route_once(short_ts)
            st = router.last_stats["phase0"]
            mains.append(st["ms_main"])
            totals.append(st["ms_total"])
            launches = st["main_launches"]
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            import torch
            t = torch.tensor([el], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, float(np.mean(mains)), float(np.mean(totals)), launches, router.last_stats, hyd

    el, ms_main, ms_total, launches, stats, hyd = timed(True, a.steps, a.warmup)
    if hyd is None:
        hyd = router.outlet_hydrographs()          # one D2H after the timed region, to report/check
        assert np.isfinite(hyd).all()
    segsteps_job = nseg * a.nsteps
    value = segsteps_job * a.steps / el
    info = router.plan0.info()
    seg0 = stats["phase0"]["segment_steps"]
    achieved = seg0 * ALG_BYTES_PER_SEGSTEP * (a.precision // 32) / (ms_main * 1e-3) / 1e9

    # the warm variant of SURVEY 8(d): the same day started from the state the cold day ended in (kept in HBM)
    warm = None
    if not use_dist and not a.no_warm:
        router.plan0.upload_forcing(a.nsteps, qlat[router.rows0], None)
        wsteps = max(1, min(a.steps, 3))
        el_w, ms_main_w, _, launches_w, stats_w, _ = timed(True, wsteps, 1)
        warm = {"value": segsteps_job * wsteps / el_w, "unit": "segment-timesteps/s", "ms_per_step": el_w / wsteps * 1e3,
                "ms_main": ms_main_w,
                "roofline_frac": stats_w["phase0"]["segment_steps"] * ALG_BYTES_PER_SEGSTEP * (a.precision // 32)
                / (ms_main_w * 1e-3) / 1e9 / HBM_PEAK_GBS}
        router.upload(a.nsteps, qlat, q0)

    full = None
    if not a.no_full_ts:
        fsteps = max(1, min(a.steps, 3))
        el_f, ms_main_f, ms_total_f, launches_f, stats_f, _ = timed(False, fsteps, 1)
        full = {
            "value": segsteps_job * fsteps / el_f, "unit": "segment-timesteps/s", "ms_per_step": el_f / fsteps * 1e3,
            "launches": launches_f, "ms_main": ms_main_f,
            "roofline_frac": stats_f["phase0"]["segment_steps"] * ALG_BYTES_PER_SEGSTEP * (a.precision // 32)
            / (ms_main_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(net, a.nsteps, a.qts, True, a.cpu_seconds)
        except Exception as e:  # the baseline must never take the GPU number down with it
            cpu = {"error": repr(e)}

    diffusive = None
    if rank == 0 and world == 1 and not a.no_diffusive:
        try:
            diffusive = diffusive_leg()
        except Exception as e:
            diffusive = {"error": repr(e)}

    if rank == 0:
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "segment-timesteps/sec, CONUS NHD 2.7M-seg MC",
            "value": value,
            "unit": "segment-timesteps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if a.precision == 32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic CONUS NHDPlus-shaped network, MC-only, 24 h @ 300 s dt (configs[2])",
                "segments": int(nseg), "networks": int(len(net["net_sizes"])), "timesteps": a.nsteps,
                "qts_subdivisions": a.qts, "assume_short_ts": True, "cold_start": True,
                "segment_levels": int(info["nlevels"]), "reach_depth": int(net["reach_depth"]),
                "sharding": "independent networks + dominant basin cut at tributary mouths" if world > 1 else "none",
                "generate_s": round(t_gen, 2), "plan_s": round(t_plan, 2),
                "plan_order": "rows of a level ordered by their secant-iteration cost in a tuning window (untimed)"
                if not a.no_retune else "topological only", "tune_s": round(t_tune, 2),
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_mc_step<float,true>" if a.precision == 32 else "k_mc_step<double,true>",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launches_per_step": launches, "avg_launch_ms": ms_main / max(launches, 1),
                "alg_bytes_per_launch": seg0 / max(launches, 1) * ALG_BYTES_PER_SEGSTEP * (a.precision // 32),
                "ms_main": ms_main, "ms_total_device": ms_total,
            },
            "cpu_baseline": cpu,
            "warm_start": warm,
            "full_ts": full,
            "outlet_hydrographs": list(hyd.shape),
            "diffusive": diffusive,
        }
        print(json.dumps(line))
    router.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
