#!/usr/bin/env python3
"""Headline benchmark: Muskingum-Cunge routing of a CONUS-scale network on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from RANK/WORLD_SIZE)

N > 1: one process per GPU; the ranks exchange cut-edge hydrographs and gather the outlet hydrographs through the engine's own
communicator (include/trmc.h: RCCL over xGMI bound through the C ABI -- no PyTorch anywhere on the path; with fewer devices
than ranks, e.g. `--gpus 2` on a one-GPU box, the ranks share devices and the shared-memory transport stands in: a rehearsal
of the control flow, not a measurement).

One "step" = one pass of the hot path over one batch of synthetic input: routing the
whole seeded synthetic CONUS network (2 729 077 segments, 14 713 independent networks,
troute_amd/synthetic.py) for one forcing window ("day") of 288 x 300 s timesteps with the
reference's configured assume_short_ts=True (test/LowerColorado_TX/test_AnA.yaml:32),
fp32 (the reference's arithmetic type).

What is timed: a SEQUENCE of consecutive days with DISTINCT forcing -- day N-1 spins the network up from a cold start, the
plan is tuned on day N (both untimed), the clock covers days N+1, N+2, ... (a ring of up to ten distinct days, each derived
from the one before like the reference's own forcing files one day apart) routed as ONE STREAM of tile launches
(troute_amd.sequence.RouteStream, include/trmc.h trmc_stream_*: the tile index runs on over the days on a ring of day slots in
HBM; every launch carries every row; a day is nsteps / 16 launches of k_mc_tile and as many of k_mc_ctile).  Inside the clock,
per day: the forcing travels from page-locked host memory to the device on a stream of its own, the state stays in HBM, and the
day's products -- outlet hydrographs and final state, SURVEY 8d's throughput mode -- are copied to page-locked host arrays beside
the launches of the days that follow.  Topology and parameters are resident.  N > 1 is timed by the SAME protocol: every rank
streams its sub-basins (the cut basin's trunk rides in its owner's stream some days behind), the cut-edge hydrographs are
exchanged once per day, rank 0 gathers the outlet block.  TRMC_BENCH_PIPELINE=two-plans times round 5's pipeline
(troute_amd.sequence.DaySequence: a plan and its clone taking turns) as the headline instead.

Prints ONE JSON line: BASELINE.json's metric (segment-timesteps/s) plus
  roofline          the day (all launches of one day of the stream) and, under dominant_kernel, k_mc_tile by itself against the
                    8 TB/s HBM roofline: HIP events on the tile stream around every timed day's launches; traffic and valu: counter
                    passes of this very command (rocprofv3 --pmc) over the last window
  cpu_baseline      the reference Fortran kernel (oracle/_ref, amdflang -O2) over every segment, decomposed like the
                    reference's by-subnetwork-jit method, C + OpenMP, bounded sample of the timesteps
  value             the stream above: INCLUDES every day's forcing host-to-device and the copy of what a throughput-mode caller
                    consumes -- outlet hydrographs + final state -- to the host;  stream: the stream's shape (slots,
                    lag, launches);  pipeline_two_plans: round 5's pipeline on the same days;  value_resident: day N+1 routed again
                    and again on the one plan, forcing resident, everything left in HBM (earlier rounds' protocol)
  untuned           the plan built from the topology alone: a window, round 5's pipeline (in_sequence), the stream (in_stream)
  velocity_on_demand   NOT the headline: the stream on a plan made with trmc_plan_options.velocity_on_demand (a step's velocity
                    formed only where it is handed on), products only and with hourly blocks, every product compared bit for bit
  value_tolerance   the same days on a plan created with TRMC_ARITH_TOLERANCE (hardware log2 / exp2 / reciprocal: NOT
                    bit-comparable; its stated tolerance is tested in tests/test_gpu_tolerance.py) -- beside the headline, never
                    it: round 5's pipeline, the stream (in_stream), the stream with velocities on demand
  forcing_persistence   the stream with days whose rows keep their magnitude with probability 0.5 / 0.0
  parity_full       EVERY segment against the reference Fortran on the CPU (the stream re-run over days N+1, N+2, full result)
  parity_mode       the whole flowveldepth array copied to the host inside the timed region
  hourly_output     every qts-th step of it (what the reference's writers keep), decimated on the device, copied inside the timed
                    region; in_sequence / in_stream: the same block as one more product of each day of either pipeline
  dropin            compute_nhd_routing_v02 from dictionaries and DataFrames at the workload's size: first call, steady state
  tuned_window_warm / cold_start / independent_forcing_cold   the tuned plan on the very window it was tuned on, on a
                    cold start, on an unrelated day
  full_ts           the same workload without the short-timestep assumption (dataflow engine); days_as_one_window: four days
                    routed as one window of 4 x nsteps steps (the general mode's sequence form)
  diffusive         the hybrid configuration's mainstem solver: one domain, a batch of 64, the reference Fortran beside them
  per_rank          (N > 1) every rank's device time
"""
import argparse
import json
import os
import sys
import time

# Before any HIP runtime is loaded: one hardware queue per stream priority (the same setting troute_amd.distributed makes
# at import, where the why is written down: with several queues per priority, some assignments of the streams to them put
# the plan stream's launches in a slow mode whenever another stream waits on its events -- DESIGN.md section 7b).  The
# single-GPU path (one plan, priority streams only) is unaffected either way.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
# the CPU baseline's OpenMP threads: one per physical core, spread over the sockets (read when libgomp initialises)
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SEGSTEP = 64          # 32 params + 4 qlat + 8 own state + 8 upstream + 12 out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseg", type=int, default=None, help="override network size (default CONUS)")
    ap.add_argument("--nnet", type=int, default=None)
    ap.add_argument("--nsteps", type=int, default=288)
    ap.add_argument("--qts", type=int, default=12)
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--chunks", type=int, default=None, help="time chunks of the multi-GPU hand-off pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-ts", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the full-result D2H variant (9.4 GB)")
    ap.add_argument("--no-retune", action="store_true",
                    help="keep the plain topological plan order (skip the untimed tuning window and plan rebuild)")
    ap.add_argument("--no-diffusive", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the leg through the reference's call surface (compute_nhd_routing_v02 from DataFrames)")
    ap.add_argument("--no-tolerance", action="store_true", help="skip the TRMC_ARITH_TOLERANCE leg (a second router of the network)")
    ap.add_argument("--no-parity-full", "--no-parity-sample", dest="no_parity_full", action="store_true",
                    help="skip the post-timing check of every segment against the reference on the CPU")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run counter passes (roofline.traffic / roofline.valu = null)")
    ap.add_argument("--headline-only", action="store_true", help="stop after the headline's timed windows (what the counter passes run)")
    ap.add_argument("--persistence", type=float, default=None,
                    help="share of the rows that keep their forcing magnitude from day to day in the timed sequence "
                         "(default: synthetic.forcing's 0.8; 0 = every day an independent draw)")
    ap.add_argument("--no-persistence-sweep", action="store_true", help="skip the legs with less persistent forcing")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (default: one per CPU the cgroup grants, at most the physical cores)")
    return ap.parse_args()


def cpu_quota():
    """CPUs of run time the cgroup of this process grants (cgroup v2 cpu.max, v1 cfs quota), None when unlimited/unknown."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            return float(txt[0]) / float(txt[1])
        return None
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except Exception:
        return None


def usable_threads():
    """One thread per CPU the cgroup grants, never more than the physical cores.  Given to the checker legs EXPLICITLY:
    torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks, and an OpenMP checker left to that default takes sixteen
    times as long on rank 0 while the other ranks wait for it."""
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        physical = os.cpu_count() or 1
    quota = cpu_quota()
    return physical if quota is None else max(1, min(physical, int(quota + 0.5)))


def cpu_baseline(net, qlat, nsteps, qts, short_ts, target_s, cpu_threads=0):
    """The reference's CPU path on this host's cores, over EVERY segment of the workload.

    Decomposition = the reference's by-subnetwork-jit method (compute.py:553-1209): sub-networks of at most
    subnetwork_target_size = 10 000 segments (compute_parameters.py:51-56) in orders that run one after the other, the
    sub-networks of an order as parallel jobs, tailwater hydrographs handed from order to order -- which is also how the
    dominant basin (half of all segments) gets more than one core.  The time x segment loop and the job scheduling are
    C + OpenMP (oracle/cpu_baseline.c) around the reference Fortran kernel symbol (oracle/_ref, amdflang -O2), so no
    Python sits inside the clock.  Bounded sample: all segments, the first `ns` timesteps of the window, `ns` sized for
    about `target_s` seconds at 3e6 segment-timesteps/s per thread (the box runs 6e6 per core; SURVEY section 6 measured 1.6e6 here).
    """
    from oracle import oracle as O
    from troute_amd.synthetic import upstream_csr

    kind, ref_name = "port", None
    if O.have_ref("libmc_ref_f32.so") and not os.environ.get("TRMC_CPU_PORT"):
        kind, ref_name = "reference", "libmc_ref_f32.so"
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        physical = os.cpu_count() or 1
    # The cores this process may actually use: the container of the GPU box shows 256 hardware threads (2 x 64 cores) but
    # its cgroup grants 16 CPUs of run time (cpu.max = 1600000 100000) -- which is why 64, 128 and 256 threads all gave
    # 9.4e7 segment-timesteps/s or less while 8 threads gave 6.3e6 each: beyond the quota, threads are throttled, not run.
    # One thread per granted CPU (never more than the physical cores).
    quota = cpu_quota()
    usable = physical if quota is None else max(1, min(physical, int(quota + 0.5)))
    threads = cpu_threads or usable
    to = net["to"]
    nseg = to.shape[0]
    ns = int(max(qts, min(nsteps, target_s * threads * 3.0e6 / nseg)))
    t0 = time.perf_counter()
    order_ptr, job_ptr, rows = O.ordered_subnetworks(to, 10000)
    up_ptr, up_idx = upstream_csr(to)
    t_prep = time.perf_counter() - t0
    q0 = np.zeros((nseg, 3), np.float32)
    q, d, done, nthreads = O.cpu_baseline_route(ns, qts, short_ts, order_ptr, job_ptr, rows, up_ptr, up_idx, net["params"],
                                                qlat, q0, ref_name=ref_name, nthreads=threads)
    dt = O.cpu_baseline_route.last_seconds           # the C + OpenMP call alone (arrays allocated and touched before)
    jobs = np.diff(order_ptr)
    return {
        "value": done / dt, "unit": "segment-timesteps/s", "cores": int(nthreads), "physical_cores": int(physical),
        "cpu_quota": None if quota is None else round(float(quota), 2),
        "kind": kind,
        "per_thread": done / dt / max(nthreads, 1),
        "sample": f"all {nseg} segments x the first {ns} of {nsteps} timesteps ({done} segment-timesteps; the as-shipped "
                  f"reference kernel, a bounded sample: {ns} steps, not the whole window), {dt:.1f} s wall; "
                  f"reference decomposition by-subnetwork-jit: {int(job_ptr.shape[0] - 1)} sub-networks of <= 10000 segments in "
                  f"{int(order_ptr.shape[0] - 1)} orders ({', '.join(str(int(j)) for j in jobs)} jobs), OpenMP dynamic, "
                  f"C loop around the reference Fortran kernel; decomposition prepared in {t_prep:.1f} s outside the clock",
        "order_seconds": [round(x, 2) for x in O.cpu_baseline_route.order_seconds],
        "_check": (q[:, ns], d),
    }


def parity_full(net, router, days, q0, nsteps, qts, outlets=None, threads=0, plan=None, final_fetched=None, fvd_fetched=None):
    """Checker, run AFTER the timed region: EVERY segment of the workload -- the dominant basin included -- is routed on the
    CPU by the reference Fortran kernel (canonical Qj_0; oracle/_ref built from the reference's sources, else the pinned
    restatement) through the same sequence of windows the router has been through -- `days`: the forcing of day N-1 (cold
    start from q0), day N, day N+1 -- over the reference's own decomposition into ordered sub-networks
    (oracle.reference_windows), and compared bit for bit with what the timed plan holds after the LAST timed window: the
    flow of every row at every step, the velocity and depth series of every row (exact position-weighted checksums of the
    bit patterns), the final state of every row, the outlet hydrographs the timed windows copied to the host.
    outlets = (rows, hydrographs): the multi-GPU job's product, the gathered outlet hydrographs of EVERY network -- then
    those are what is compared (the ranks hold the rest)."""
    from oracle import oracle as O
    to, params = net["to"], net["params"]
    nseg = to.shape[0]
    t0 = time.perf_counter()
    ref = O.reference_windows(to, params, days, q0, nsteps, qts, True, nthreads=threads or usable_threads())
    u32 = lambda x: np.ascontiguousarray(x).view(np.uint32)   # noqa: E731
    base = {"segments": int(nseg), "networks": int((to < 0).sum()), "windows": len(days), "timesteps": int(nsteps),
            "checker": ("reference Fortran kernel (oracle/_ref/libmc_ref_qj0_f32.so, canonical Qj_0), " if ref["kind"] == "reference"
                        else "oracle/ C restatement (pinned to the reference Fortran), ")
                       + "C + OpenMP over the reference's by-subnetwork-jit decomposition, every segment, all windows",
            "checker_seconds": round(ref["seconds"], 1)}
    o_rows, o_hyd = outlets
    want_o = ref["q"][o_rows, 1:]
    same_o = bool(np.array_equal(u32(o_hyd), u32(want_o)))
    if router is None:
        base.update({"bit_identical": same_o, "compared": "the all-gathered outlet hydrographs of every network",
                     "differing_values": int((u32(o_hyd) != u32(want_o)).sum()), "seconds": round(time.perf_counter() - t0, 1)})
        return base
    if fvd_fetched is not None:                                # (a stream of windows: the last day's products as they arrived)
        fvd, final = np.asarray(fvd_fetched).reshape(nseg, nsteps, 3), final_fetched
    else:
        plan = plan if plan is not None else router.plan0      # (the plan that routed the last window)
        fvd = plan.download_fvd().reshape(nseg, nsteps, 3)
        final = plan.download_final_state()
        if final_fetched is not None and not np.array_equal(u32(final_fetched), u32(final)):
            return dict(base, bit_identical=False, error="the asynchronously fetched final state differs from the plan's")
    diff_q = 0
    for lo in range(0, nseg, 200000):
        diff_q += int((u32(fvd[lo:lo + 200000, :, 0]) != u32(ref["q"][lo:lo + 200000, 1:])).sum())
    diff_v = int((O.series_checksum(fvd[:, :, 1]) != ref["chk_v"]).sum())
    diff_d = int((O.series_checksum(fvd[:, :, 2]) != ref["chk_d"]).sum())
    diff_s = int((u32(final) != u32(ref["state"])).sum())
    base.update({"bit_identical": diff_q == 0 and diff_v == 0 and diff_d == 0 and diff_s == 0 and same_o,
                 "flows_identical": diff_q == 0, "velocity_series_identical": diff_v == 0, "depth_series_identical": diff_d == 0,
                 "final_state_identical": diff_s == 0, "outlet_hydrographs_identical": same_o,
                 "differing_values": diff_q + diff_s, "rows_with_differing_velocity_or_depth": diff_v + diff_d,
                 "compared": "flow of every row at every step; velocity and depth series of every row (exact checksums); final "
                             "state of every row; the outlet hydrographs of the last timed window",
                 "seconds": round(time.perf_counter() - t0, 1)})
    return base


def dropin_leg(net, nsteps, qts, days):
    """The reference's own call surface at the workload's size: ``compute_nhd_routing_v02`` (compute.py:507-546) fed what
    ``nwm_route`` feeds it -- connection / reach dictionaries made by the reference-shaped graph producers
    (nhd_network.organize_independent_networks), DataFrames of parameters, state and lateral inflows -- once per run set
    (nwm_routing/__main__.py:195-333, :1215): the first call (flattening, plan, tuning), then steady-state calls on the next
    days' forcing with the state of the call before (new_q0, AbstractNetwork.py:177-191), every qts-th step of the result kept
    (``output_stride``: what the writers take, output.py:209-216).  Wall time per call, split."""
    import pandas as pd
    from troute_amd import nhd_network as nn
    from troute_amd.routing import compute as RC
    to = net["to"]
    nseg = to.shape[0]
    t0 = time.perf_counter()
    ids = np.arange(1, nseg + 1, dtype=np.int64)                  # (ascending ids in row order: the caller's table is sorted by id)
    dn = np.where(to >= 0, ids[np.maximum(to, 0)], 0)
    conn = {int(s): ([int(t)] if t else []) for s, t in zip(ids.tolist(), dn.tolist())}
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    t_graph = time.perf_counter() - t0
    cols = ["dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"]
    param_df = pd.DataFrame(net["params"][:, 1:], index=ids, columns=cols)
    param_df["alt"] = 0.0
    q0_df = pd.DataFrame(np.zeros((nseg, 3), np.float32), index=ids, columns=["qu0", "qd0", "h0"])
    e = pd.DataFrame()
    calls = []
    device_ms = []
    for k in range(4):
        qlat_df = pd.DataFrame(days[k % len(days)], index=ids)
        prof = None
        if k == 3 and os.environ.get("TRMC_DROPIN_PROFILE"):      # (diagnosis: where the steady-state call's time goes)
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        res, _ = RC.compute_nhd_routing_v02(conn, rconn, {}, reaches_bytw, "V02-structured", "by-network", 10000, 4, None, 300.0, nsteps,
                                            qts, ind, param_df, q0_df, qlat_df, e, e, e, e, e, e, e, e, e, e, e, {}, True, False, e, {},
                                            e, False, [{}, {}], output_stride=qts)
        t1 = time.perf_counter()
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(22)
        # new_q0 (AbstractNetwork.py:182-190): the last kept step of every row, (q, q, depth), in the table's order
        fin = np.concatenate([r[1][:, [-3, -3, -1]] for r in res])
        order = np.concatenate([r[0] for r in res])
        q0_df = pd.DataFrame(fin, index=order, columns=["qu0", "qd0", "h0"]).reindex(ids)
        calls.append({"call_s": round(t1 - t0, 4), "new_q0_s": round(time.perf_counter() - t1, 4)})
        try:
            from troute_amd.routing.fast_reach import mc_reach as MR
            plans = list(MR._PLANS._d.values())
            device_ms.append(round(float(plans[-1]["plan"].stats()["ms_total"]), 2) if plans else None)
        except Exception:
            device_ms.append(None)
    steady = calls[-1]["call_s"]
    return {"graph_s": round(t_graph, 2), "tailwaters": len(reaches_bytw), "reaches": int(sum(len(v) for v in reaches_bytw.values())),
            "calls": calls, "device_window_ms_of_each_call": device_ms, "steady_state_call_ms": round(steady * 1e3, 1),
            "output_stride": qts, "result": f"{len(res)} tuples; flowveldepth [rows, {nsteps // qts} x 3] per tailwater (views of one block)",
            "what": "compute_nhd_routing_v02(connections, rconn, ..., param_df, q0, qlats, ...) at the workload's size: call 0 flattens the "
                    "network and builds the plan, call 1 rebuilds it with call 0's costs as the row-order hint, calls 2-3 are the steady "
                    "state (network, table and plan found by the identity of the caller's objects); each call = table look-up of q0 / "
                    "qlats + upload + one routing window + the decimated result to the host + the per-tailwater result list"}


def _diffusive_inputs(gold, nsteps):
    z = np.load(gold)
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    ins["timestep_ar_g"] = ins["timestep_ar_g"].copy()
    ins["timestep_ar_g"][2] = 300.0 * nsteps / 3600.0
    ins["ntss_ev_g"] = np.array(nsteps + 1)
    return ins


def _diffusive_call(gold, nsteps, path, sym, prepare_only=False):
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
    fn = getattr(lib, sym)
    if prepare_only:                      # (the arguments marshalled, the call left to the caller: a pool of threads times many)
        return lambda: fn(*args), (keep, outs)
    t0 = time.perf_counter()
    fn(*args)
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
        # What the device is FOR in this solver is many tailwater domains at once (one compute unit each); a single domain is
        # a chain of dependent fp64 steps that one CPU core walks faster.  So the batch is priced against the same 64 domains
        # on the host's granted cores: the host restatement (the faster of the two CPU forms) from a pool of threads, one
        # domain per call (ctypes releases the interpreter lock for the duration of a call).
        try:
            import concurrent.futures
            quota = cpu_quota()
            cores = max(1, min(os.cpu_count() or 1, int(quota + 0.5) if quota else (os.cpu_count() or 1)))
            calls = [_diffusive_call(gold, nsteps, host, "dw_oracle_diffnw", prepare_only=True) for _ in range(nb)]
            def unbind():   # (OMP_PROC_BIND has pinned this process's initial thread to one core: the pool's threads inherit that)
                try:
                    os.sched_setaffinity(0, range(os.cpu_count() or 1))
                except Exception:
                    pass
            with concurrent.futures.ThreadPoolExecutor(cores, initializer=unbind) as ex:
                list(ex.map(lambda c: None, range(cores)))      # (threads started before the clock)
                t0 = time.perf_counter()
                list(ex.map(lambda c: c[0](), calls))
                cpu_batch = time.perf_counter() - t0
            out["batch"].update({"host_restatement_s": cpu_batch, "host_cores": cores,
                                 "gpu_over_host": cpu_batch / out["batch"]["gpu_s"],
                                 "what": f"{nb} tailwater domains in one launch on the device against the same {nb} domains, one call each, on "
                                         f"{cores} host threads (the host restatement of the same solver)"})
        except Exception as e:
            out["batch"]["host_error"] = repr(e)
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


def spawn_ranks(a):
    """`--gpus N` without a launcher: start the N ranks as N copies of this script (RANK / LOCAL_RANK / WORLD_SIZE in their
    environment, one communicator key for the launch); rank 0 prints the JSON line on our stdout."""
    import subprocess
    import uuid
    key = f"bench{os.getpid()}_{uuid.uuid4().hex[:8]}"
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), TRMC_COMM_KEY=key,
                   MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    # ONE JSON line on stdout, whatever the libraries loaded below print there (RCCL writes a version banner to stdout when a
    # communicator is made): file descriptor 1 is pointed at stderr for the rest of the process, the line goes out through
    # a private copy of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    from troute_amd import _lib, sharding, synthetic
    from troute_amd import comm as X
    from troute_amd.distributed import ShardedRouter

    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: libtrmc.so has no CPU fallback")
    use_dist = world > 1 or bool(os.environ.get("TRMC_FORCE_DIST"))   # the env var exercises the communicator path at N=1
    # the communicator: RCCL over xGMI when every rank has a device of its own; with fewer devices than ranks the ranks
    # share them and the shared-memory transport stands in (TRMC_BENCH_BACKEND=rccl|shm overrides).  Not a measurement then.
    device = local_rank % ndev
    comm = None
    if world > 1:
        # a rank that hangs (a collective some rank never enters) must not hold the node for ever: the launch is abandoned
        import threading
        limit = float(os.environ.get("TRMC_BENCH_WATCHDOG_S", "1500"))

        def give_up():
            sys.stderr.write(f"bench.py: rank {rank} of {world} still running after {limit:.0f} s -- abandoning the launch\n")
            sys.stderr.flush()
            os._exit(3)
        wd = threading.Timer(limit, give_up)
        wd.daemon = True
        wd.start()
        # ... and one that stands still says where: every thread's stack on stderr (TRMC_BENCH_STACKS_S, default: never)
        if os.environ.get("TRMC_BENCH_STACKS_S"):
            import faulthandler
            _stacks = open(os.path.join(os.environ.get("TRMC_BENCH_STACKS_DIR", "/tmp"), f"bench_stacks_rank{rank}.txt"), "w")
            faulthandler.dump_traceback_later(float(os.environ["TRMC_BENCH_STACKS_S"]), repeat=True, file=_stacks)
    if use_dist:
        comm = X.Comm(rank, world, device, backend=os.environ.get("TRMC_BENCH_BACKEND", "auto"))
    local_rank = device

    kw = {}
    if a.nseg:
        kw["nseg"] = a.nseg
        kw["nnet"] = a.nnet or max(3, a.nseg // 185)
    cache = os.environ.get("TRMC_CACHE", "/tmp/trmc_cache")
    t0 = time.perf_counter()
    if rank == 0:
        net = synthetic.generate(cache_dir=cache, **kw)
    if comm is not None:
        comm.barrier()
    if rank != 0:
        net = synthetic.generate(cache_dir=cache, **kw)
    t_gen = time.perf_counter() - t0
    to, params = net["to"], net["params"]
    nseg = to.shape[0]
    # Three consecutive days of the same basin (synthetic.forcing: yesterday's spatial pattern, a row-wise lognormal
    # day-to-day factor, a fifth of the rows drawn anew).  Day N-1 spins the network up from a cold start; day N, started
    # from the state day N-1 leaves in HBM, is the window every plan is TUNED on (untimed); day N+1, started from the state
    # day N leaves, is the window that is TIMED -- an operational sequence: windows follow each other warm.
    qlat_s = net["qlat"]
    qlat_a = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 1, previous=qlat_s)
    qlat_b = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2, previous=qlat_a)
    q0 = np.zeros((nseg, 3), dtype=np.float32)

    tuned = {"speed": None, "part": None, "part_last": None}

    def make_router(hint, short_ts, qlat, state, options=None, stream=False):
        # with a hint (the measured cost of every row on the tuning day) the partition is packed by cost as well, and by the
        # pace every rank was MEASURED to keep on that day (what a trunk does to its owner is in there; sharding.partition)
        part = (sharding.partition(to, world, row_cost=hint, rank_speed=tuned["speed"], previous=tuned["part"])
                if (hint is not None and world > 1) else None)
        if stream:                                     # (the plans a STREAM of windows runs on: troute_amd.sequence.RouteStream)
            r = ShardedRouter(to, params, rank=rank, world=world, device=local_rank, precision=a.precision, cost_hint=hint,
                              partition=tuned["part_last"] if tuned["part_last"] is not None else part, options=options, stream=True)
            if use_dist:
                r.enable_device_exchange(comm)
            return r
        r = ShardedRouter(to, params, rank=rank, world=world, device=local_rank, precision=a.precision, cost_hint=hint,
                          assume_short_ts=short_ts, partition=part, options=options)
        tuned["part_last"] = r.part if world > 1 else None
        r.upload(a.nsteps, qlat, state)
        if use_dist:
            r.enable_device_exchange(comm)
            r.upload_trunk()
        return r

    def route_once(router, short_ts):
        if use_dist:   # every hand-off stays in HBM: gather kernels -> all-gather (RCCL over xGMI) -> boundary rows
            return router.route_on_device(a.qts, short_ts, a.chunks)
        return router.route_resident(a.qts, short_ts), None   # outlet hydrographs stay in HBM

    def sync():
        if comm is not None:
            X.device_synchronize(local_rank)
            comm.barrier()

    def timed(router, short_ts, steps, warmup, d2h=None):
        """EXACTLY `steps` routing windows between barriers; max over ranks.  d2h: None (results stay in HBM), "state"
        (outlet hydrographs + final state on the host inside the clock: what a caller of the throughput mode consumes,
        SURVEY 8d -- copied on a copy stream beside the NEXT window, the last copy waited for before the clock stops) or
        "full" (the whole flowveldepth array, synchronously: parity mode)."""
        def one():
            if d2h == "state":
                if use_dist:
                    rows, hyd = route_once(router, short_ts)
                    got = router.fetch_wait()                     # window k - 1, copied beside window k
                    router.fetch_begin(hyd, want_hyd=(rank == 0))
                    return got[0]
                return router.route_and_fetch(a.qts, short_ts)[0]
            rows, hyd = route_once(router, short_ts)
            if d2h == "full":
                router.plan0.download_fvd()
            elif d2h == "hourly":
                router.plan0.download_fvd(a.qts)
            return hyd

        def drain():
            return router.fetch_wait()[0] if d2h == "state" else None

        for _ in range(warmup):
            one()                                    # (the page-locked result buffers are taken from the pool and given back)
        drain()
        sync()
        t0 = time.perf_counter()
        mains, totals, launches, hyd = [], [], 0, None
        dbg = []
        for _ in range(steps):
            td = time.perf_counter()
            hyd = one()
            dbg.append((time.perf_counter() - td) * 1e3)
            st = router.last_stats["phase0"]
            mains.append(st["ms_main"])
            totals.append(st["ms_total"])
            launches = st["main_launches"]
        td = time.perf_counter()
        last = drain()
        hyd = last if last is not None else hyd
        sync()
        el = time.perf_counter() - t0
        if os.environ.get("TRMC_BENCH_DEBUG"):
            print(f"[timed d2h={d2h}] per-step wall ms {[round(x, 2) for x in dbg]} drain {(time.perf_counter() - td) * 1e3:.2f} "
                  f"device ms_total {[round(x, 2) for x in totals]}", file=sys.stderr)
        if comm is not None:
            el = float(comm.all_reduce_max_host(np.array([el], dtype=np.float64))[0])
        return {"el": el, "ms_main": float(np.mean(mains)), "ms_total": float(np.mean(totals)), "launches": launches,
                "stats": router.last_stats, "hyd": hyd, "steps": steps}

    segsteps_job = nseg * a.nsteps
    bytes_per = ALG_BYTES_PER_SEGSTEP * (a.precision // 32)

    def rate(t):
        return segsteps_job * t["steps"] / t["el"]

    def frac(t):
        return t["stats"]["phase0"]["segment_steps"] * bytes_per / (t["ms_main"] * 1e-3) / 1e9 / HBM_PEAK_GBS

    def spin_up(router, through_day_n):
        """day N-1 from a cold start, then (optionally) day N from the state it leaves; untimed"""
        router.upload(a.nsteps, qlat_s, q0)
        route_once(router, True)
        if through_day_n:
            router.upload(a.nsteps, qlat_a, None)
            route_once(router, True)

    # ---- 1. the plan as it is built from the topology alone; day N doubles as the tuning window ----------------------
    t0 = time.perf_counter()
    router = make_router(None, True, qlat_s, q0)
    t_plan = time.perf_counter() - t0
    engine = getattr(router.plan0, "engine", "levels")
    usteps = max(1, min(a.steps, 2))
    t0 = time.perf_counter()
    route_once(router, True)                           # day N-1, cold
    router.upload(a.nsteps, qlat_a, None)              # day N, warm: which rows are cheap (dry channel: one secant
    router.collect_cost(not a.no_retune)               # iteration), which are not, which go over bank
    route_once(router, True)
    hint = None if a.no_retune else router.iteration_hint()
    if hint is not None and comm is not None:   # every rank measured its own rows: all of them need the whole vector
        hint = comm.all_reduce_max_host(hint)
    if hint is not None and comm is not None:
        # every rank's pace on the tuning window: the cost its rows were measured to carry over the device time it took
        cost_mine = float(hint[np.concatenate([router.rows0, router.rows1])].astype(np.float64).sum())
        lt = comm.all_gather_host(np.array([cost_mine, router.last_stats["phase0"]["ms_main"]], dtype=np.float64))
        tuned["speed"], tuned["part"] = sharding.rank_speeds(lt[:, 0], lt[:, 1]), router.part
    router.collect_cost(False)
    t_tune = time.perf_counter() - t0
    def state_after_day_n(r):
        """the state after day N, where the timed sequence starts: every row (one GPU), or this rank's rows in the order of its
        merged plan (DaySequence takes either)"""
        return r._state_plans[0].download_final_state() if use_dist else r.plan0.download_final_state()
    state_n = state_after_day_n(router)
    router.upload(a.nsteps, qlat_b, None)              # day N+1, warm
    unt = timed(router, True, usteps, 1)
    untuned = {"value": rate(unt), "unit": "segment-timesteps/s", "ms_per_step": unt["el"] / usteps * 1e3,
               "ms_main": unt["ms_main"], "roofline_frac": frac(unt),
               "window": "day N+1, warm start (the headline's window) on the plan built from the topology alone"}
    # the days of the timed sequence (a ring of distinct days in page-locked memory: see 3.)
    from troute_amd.sequence import DaySequence, pinned_like
    ndays = max(2 if a.headline_only else 4, min(int(os.environ.get("TRMC_BENCH_DAYS", "10")), a.steps + a.warmup))
    t0 = time.perf_counter()
    ring, prev_day = [], qlat_a
    for i in range(ndays):
        day = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2 + i, previous=prev_day, persistence=a.persistence)
        ring.append(pinned_like(day) if not use_dist else day)
        prev_day = day
    assert np.array_equal(ring[0], qlat_b) or a.persistence is not None
    t_days = time.perf_counter() - t0
    # ... and the headline's own protocol -- the sequence of days, plan and clone -- on that plan, before it is rebuilt
    if not use_dist and not a.headline_only and not a.no_retune and a.precision == 32:
        try:
            with DaySequence(router, a.nsteps, a.qts) as us:
                us.run(ring, state_n, max(6, len(ring)), 0)
                ssteps = max(2, min(a.steps, 6))
                s1 = us.run(ring, state_n, ssteps, 1)
            untuned["in_sequence"] = {"ms_per_step": s1["el"] / ssteps * 1e3, "steps": ssteps,
                                      "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / (s1["el"] / ssteps) / 1e9 / HBM_PEAK_GBS,
                                      "what": "the headline's pipeline (DaySequence, plan + clone, distinct days) on the plan built from the "
                                              "topology alone"}
            router.upload(a.nsteps, qlat_b, state_n)
        except Exception as e:
            untuned["in_sequence"] = {"error": repr(e)}
        try:      # ... and the headline's protocol of round 6 -- ONE stream of tile launches over the days -- on a plan in cluster
            # order built from the topology alone (no cost hint: neither the rows' order nor the clusters' packing knows a cost)
            from troute_amd.sequence import RouteStream as _RS
            ur = make_router(None, True, None, None, stream=True)
            try:
                with _RS(ur, a.nsteps, a.qts) as us:
                    us.run(ring[:4], state_n, 2, 0, prepared=True)
                    ssteps = max(2, min(a.steps, 12))
                    s2 = us.run(ring, state_n, ssteps, 1, prepared=True)
                untuned["in_stream"] = {"ms_per_step": s2["el"] / ssteps * 1e3, "steps": ssteps,
                                        "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / (s2["el"] / ssteps) / 1e9 / HBM_PEAK_GBS,
                                        "what": "the headline's stream of days on a plan built from the topology alone (no tuning window)"}
                del s2
            finally:
                ur.close()
        except Exception as e:
            untuned["in_stream"] = {"error": repr(e)}

    # ---- 2. the plan rebuilt with day N's costs as its hint (same results: tests/test_gpu_parity.py), spun up again ----
    if not a.no_retune:
        t0 = time.perf_counter()
        router.close()
        router = make_router(hint, True, qlat_s, q0)
        spin_up(router, True)
        # Multi-GPU: the pace of a rank depends on what it carries (a trunk owner's drain weighs more the fewer sub-basins
        # it has beside it), so the partition is fed back once more -- the paces of the ranks on the HINTED plan, measured
        # on the day-N window the spin-up has just routed -- unless the ranks already finish within 4 % of each other.
        feedback = []
        for _ in range(int(os.environ.get("TRMC_BENCH_REBALANCE", "2")) if comm is not None else 0):
            cost_mine = float(hint[np.concatenate([router.rows0, router.rows1])].astype(np.float64).sum())
            lt = comm.all_gather_host(np.array([cost_mine, router.last_stats["phase0"]["ms_main"]], dtype=np.float64))
            feedback.append([round(float(x), 3) for x in lt[:, 1]])
            if lt[:, 1].max() <= 1.04 * lt[:, 1].mean():
                break
            tuned["speed"], tuned["part"] = sharding.rank_speeds(lt[:, 0], lt[:, 1]), router.part
            router.close()
            router = make_router(hint, True, qlat_s, q0)
            spin_up(router, True)
        tuned["feedback_ms"] = feedback
        state_n = state_after_day_n(router)
        router.upload(a.nsteps, qlat_b, None)
        t_tune += time.perf_counter() - t0

    # ---- 3. the headline: day N+1 on the plan tuned on day N; every window's outlet hydrographs and final state arrive on
    # the host inside the clock (SURVEY 8d's throughput mode), copied beside the next window ---------------------------
    # One protocol for every N (troute_amd.sequence.DaySequence): a SEQUENCE of consecutive days with distinct forcing -- day
    # N+1, N+2, ... -- a ring of `ndays` distinct days in page-locked host memory, each derived from the one before like days
    # N-1 -> N -> N+1 were (synthetic.forcing).  One GPU: on the tuned plan and its clone; a rank of a multi-GPU job: on its
    # merged plan, its rows of every day staged from page-locked memory, the state carried on in HBM.
    persist = None
    # (diagnosis only, with --headline-only: TRMC_BENCH_OUTPUT_STRIDE=n times / traces the pipeline with the decimated result
    # among each day's products -- the `hourly_output.in_sequence` leg -- in place of the headline's)
    ostride = (int(os.environ.get("TRMC_BENCH_OUTPUT_STRIDE", "0")) or None) if a.headline_only else None
    # THE HEADLINE (round 6): the days as ONE STREAM of tile launches (troute_amd.sequence.RouteStream, trmc_stream_*) on a router
    # whose plans are in cluster order -- every launch carries every row, a day costs nsteps / 16 launches on the tile stream and
    # as many on the clusters', nothing between two days; on several ranks the cut-edge hydrographs are exchanged once a day.
    # TRMC_BENCH_PIPELINE=two-plans times round 5's pipeline (DaySequence: a plan and its clone) as the headline instead.
    from troute_amd.sequence import RouteStream
    use_stream = os.environ.get("TRMC_BENCH_PIPELINE", "stream") != "two-plans" and a.precision == 32
    dayseq = DaySequence(router, a.nsteps, a.qts, nchunks=a.chunks, output_stride=ostride,
                         timeline=bool(os.environ.get("TRMC_BENCH_DEBUG")))
    srouter, stream_ring, stream_err = None, None, None
    if use_stream:
        try:
            srouter = make_router(hint, True, None, None, stream=True)
            with RouteStream(srouter, a.nsteps, a.qts, output_stride=ostride) as rs:
                # (this rank's rows of every day, page-locked: outside the clock; one GPU: the ring's arrays as they are)
                stream_ring = rs.prepare_days(ring) if use_dist else ring
                rs.run(stream_ring, state_n, 2, 0, prepared=True)   # (untimed: the ring of day slots, the page-locked product rings)
                sseq = rs.run(stream_ring, state_n, a.steps, a.warmup, prepared=True)
            sinfo = sseq["info"]
            tpd = sinfo["tiles_per_day"]
            nrouted = int(srouter._rowsS.shape[0]) if use_dist else nseg
            push_ms = float(np.mean(sseq["push_ms"])) if sseq["push_ms"] else sseq["el"] / a.steps * 1e3
            sstats = {"segment_steps": nrouted * a.nsteps, "main_launches": sinfo["launches"] // max(1, sseq["days_routed"]),
                      "wide_launches": tpd if sinfo["wide_levels"] > 0 else 0, "wide_levels": sinfo["wide_levels"], "wide_k": a.nsteps // tpd,
                      "cluster_levels": sinfo["cluster_levels"], "lag_max_tiles": sinfo["lag_max"], "slots": sinfo["slots"],
                      "ms_wide": push_ms, "ms_main": sseq["el"] / a.steps * 1e3}
            lagS = srouter.stream_plan(getattr(srouter, "_planS_lag", 0)).lags()[0]
            sstats["wide_segment_steps"] = int((lagS[lagS >= 0] < sinfo["wide_levels"]).sum()) * a.nsteps
            head = {"el": sseq["el"], "ms_main": sseq["el"] / a.steps * 1e3, "ms_total": sseq["el"] / a.steps * 1e3,
                    "launches": sstats["main_launches"], "stats": {"phase0": sstats}, "hyd": sseq["hyd"], "steps": a.steps,
                    "stream": {"warmup_days": sseq["warmup"], **sinfo}}
            seq = {"day_ms": [round(x, 2) for x in sseq["push_ms"]], "hyd": sseq["hyd"], "el": sseq["el"]}
        except Exception as e:
            import traceback
            traceback.print_exc()
            stream_err, use_stream = repr(e), False
            if use_dist:
                raise
    two_plans = None
    if use_dist and not use_stream:
        local_ring = dayseq.prepare_days(ring)          # (this rank's rows of every day, page-locked: outside the clock)
        seq = dayseq.run(local_ring, state_n, a.steps, a.warmup, prepared=True)
        head = {"el": seq["el"], "ms_main": float(np.mean(seq["ms_main"])), "ms_total": seq["el"] / a.steps * 1e3,
                "launches": router.last_stats["phase0"]["main_launches"], "stats": router.last_stats, "hyd": seq["hyd"], "steps": a.steps}
    elif not use_dist and not (use_stream and a.headline_only):
        # round 5's pipeline -- a plan and its clone taking turns (DaySequence) -- beside the stream (`pipeline_two_plans`), or as
        # the headline itself (TRMC_BENCH_PIPELINE=two-plans)
        # (untimed: the clone's window buffers -- 19 GB of planes and result -- and both plans' copy streams and page-locked
        # result rings are made at their first use)
        # ... and every page-locked array -- the ring of days, the three result sets of either plan -- is used by a copy for the
        # first time (a first use costs milliseconds once: the third timed day took 20-22 ms and the fourth 13 without this)
        dayseq.run(ring, state_n, max(6, len(ring)), 0)
        osteps = a.steps if not use_stream else max(2, min(a.steps, 6))
        oseq = dayseq.run(ring, state_n, osteps, a.warmup)
        two_plans = {"ms_per_step": oseq["el"] / osteps * 1e3, "steps": osteps, "value": nseg * a.nsteps * osteps / oseq["el"],
                     "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / (oseq["el"] / osteps) / 1e9 / HBM_PEAK_GBS,
                     "what": "round 5's headline pipeline on the same days: a plan and its clone taking turns (DaySequence), the narrow levels "
                             "one launch per timestep"}
        if not use_stream:
            seq = oseq
            # (ms_main of the sequence: the WALL time per day, every kernel, copy and hand-over of the pipeline in it -- the events
            # around a single window also span what the neighbouring day's kernels take of the device while they overlap it)
            head = {"el": seq["el"], "ms_main": seq["el"] / a.steps * 1e3, "ms_total": seq["el"] / a.steps * 1e3,
                    "ms_window_events": float(np.mean(seq["ms_main"])),
                    "launches": router.plan0.stats()["main_launches"], "stats": {"phase0": seq["last_plan"].stats()},
                    "hyd": seq["hyd"], "steps": a.steps}
        # how much of this rests on the day-to-day persistence of the forcing: the same pipeline with half, and with none, of
        # the rows keeping their magnitude from one day to the next (the plan stays the one tuned on day N)
        if not a.headline_only and not a.no_persistence_sweep:
            persist = {}
            for pv in (0.5, 0.0):
                r2, prev_day = [], qlat_a
                for i in range(4):
                    day = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 40 + i, previous=prev_day, persistence=pv)
                    r2.append(ring[i])                # (the page-locked arrays of the headline's ring are reused)
                    ring[i][...] = day
                    prev_day = day
                if use_stream:                        # (the headline's pipeline: the stream; one GPU: its days are the ring's arrays)
                    with RouteStream(srouter, a.nsteps, a.qts) as rs:
                        s2 = rs.run(r2, state_n, 4, 1, prepared=True)
                else:
                    s2 = dayseq.run(r2, state_n, 4, 1)
                persist[str(pv)] = {"ms_per_day": s2["el"] / 4 * 1e3,
                                    "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / (s2["el"] / 4) / 1e9 / HBM_PEAK_GBS}
            persist["what"] = ("the timed pipeline on the same tuned plan with days whose rows keep their forcing magnitude with "
                               "probability 0.5 / 0.0 from one day to the next (the headline's days: synthetic.forcing's default, "
                               "0.999 -- what the reference's own forcing files show one day apart)")
            prev_day = qlat_a                          # the ring back to the headline's days (the parity pass below uses them)
            for i in range(min(4, ndays)):
                day = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2 + i, previous=prev_day, persistence=a.persistence)
                ring[i][...] = day
                prev_day = day
    if dayseq.timeline and getattr(dayseq, "device_days", None):
        print("[sequence] device clock, ms (day, tiles begin, tiles end, tail begins, tail ends): "
              + " ".join(str(d) for d in dayseq.device_days), file=sys.stderr)
    if dayseq.timeline:
        print("[sequence] host timeline (ms, call returned, day): " + " ".join(f"{t}:{k}:{w}" for t, k, w in dayseq.timeline[-60:]),
              file=sys.stderr)
    hyd = head["hyd"]
    if hyd is None:                                 # (a rank other than 0 of a multi-GPU job does not fetch the outlet block)
        hyd = np.zeros((0, a.nsteps), np.float32)
    assert np.isfinite(hyd).all()
    if a.headline_only:            # (a counter pass of pmc_counters(): the last windows of the process are the headline's)
        if rank == 0:
            json_out.write(json.dumps({"metric": "segment-timesteps/sec, CONUS NHD 2.7M-seg MC", "value": rate(head),
                                       "unit": "segment-timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                                       "ms_per_step": head["el"] / a.steps * 1e3, "ms_main": head["ms_main"],
                                       "headline_only": True, "output_stride": ostride, "day_ms": seq["day_ms"]}) + "\n")
            json_out.flush()
        dayseq.close()
        router.close()
        if srouter is not None:
            srouter.close()
        if comm is not None:
            comm.close()
        return
    parity = None
    # (a multi-rank job: the checker runs on rank 0 alone and takes a minute or two -- AFTER the last leg the ranks take
    # together, below; the other ranks would give up at a barrier in the meantime)
    dist_outlets = None
    if use_dist and not a.no_parity_full and a.precision == 32:
        # the first two days of the timed sequence once more, untimed, from the state after day N: the all-gathered outlet block
        # of day N+2 is what rank 0 hands to the checker (AFTER the last leg the ranks take together)
        if use_stream:
            with RouteStream(srouter, a.nsteps, a.qts) as rs:        # (exactly these two days, from the state after day N)
                for item in rs.route(stream_ring[:2], state_n, prepared=True):
                    chk_hyd = None if item[1] is None else np.array(item[1], copy=True)
                out_rows_s = rs.outlet_rows
            if rank == 0:
                dist_outlets = (np.array(out_rows_s, copy=True), chk_hyd)
        else:
            chk = dayseq.run(local_ring[:2], state_n, 2, 0, prepared=True)
            if rank == 0:
                dist_outlets = (np.array(router._out_rows, copy=True), np.array(chk["hyd"], copy=True))
    router.upload(a.nsteps, qlat_b, None if use_dist else state_n)   # (the legs below route day N+1 again and again on the one plan)
    resident = timed(router, True, max(1, min(a.steps, 3)), 1)
    value = rate(head)
    info = router.plan0.info()
    stats = head["stats"]
    seg0 = stats["phase0"]["segment_steps"]
    achieved = seg0 * bytes_per / (head["ms_main"] * 1e-3) / 1e9
    per_rank = None
    if comm is not None:                            # every rank's device time, so that a scaling run can be diagnosed
        allr = comm.all_gather_host(np.array([head["ms_main"], float(seg0)], dtype=np.float64))
        per_rank = [{"rank": i, "ms_main": float(t[0]), "segment_steps": int(t[1])} for i, t in enumerate(allr)]

    extra = {"value_tolerance": None, "value_resident": {"value": rate(resident), "unit": "segment-timesteps/s",
                                "ms_per_step": resident["el"] / resident["steps"] * 1e3, "ms_main": resident["ms_main"],
                                "what": "the same windows with every result left in HBM (no copy to the host in the clock)"},
             "copied_per_step": f"outlet hydrographs [{len(net['net_sizes'])} x {a.nsteps}] + final state [{nseg} x 3] into page-locked "
                                "host arrays on a copy stream beside the next window; the last window's copy is waited for inside the clock"}
    if not use_dist:
        if not a.no_parity_mode:
            w = timed(router, True, 2, 1, d2h="full")
            extra["parity_mode"] = {"value": rate(w), "unit": "segment-timesteps/s", "ms_per_step": w["el"] / 2 * 1e3,
                                    "copied": f"full flowveldepth [{nseg} x {a.nsteps} x 3] ({nseg * a.nsteps * 12 / 1e9:.1f} GB) into a "
                                              "page-locked array from the library's pool (what compute_network_structured returns; "
                                              "the pool is warm: the first window of a process also pays for locking the pages), "
                                              "inside the timed region"}
            os.environ["TRMC_PINNED_RESULTS"] = "0"
            w = timed(router, True, 1, 0, d2h="full")
            del os.environ["TRMC_PINNED_RESULTS"]
            extra["parity_mode"]["pageable"] = {"value": rate(w), "ms_per_step": w["el"] * 1e3}
            from troute_amd import _lib as _tl
            _tl.pinned_pool_clear()
            # ... and what the reference's WRITERS consume of it: every qts-th step (hourly output of 5-minute steps,
            # nwm_routing/output.py:209-216, nhd_io.py:2379-2382), decimated on the device (trmc_download_fvd_strided: what
            # compute_network_structured(..., output_stride=qts) returns), synchronously inside the timed region
            w = timed(router, True, 3, 1, d2h="hourly")
            extra["hourly_output"] = {"value": rate(w), "unit": "segment-timesteps/s", "ms_per_step": w["el"] / 3 * 1e3,
                                      "copied": f"flowveldepth at every {a.qts}th step [{nseg} x {a.nsteps // a.qts} x 3] "
                                                f"({nseg * (a.nsteps // a.qts) * 12 / 1e9:.2f} GB), decimated on the device, into a "
                                                "page-locked array from the library's pool, inside the timed region"}
            # ... and the same product inside the PIPELINE (DaySequence(output_stride=qts): decimated on the copy stream, copied
            # beside the next day -- trmc_fetch_begin_fvd): the period of a day that also hands every row's hourly (q, v, d) over
            try:
                dayseq.set_output_stride(a.qts)
                dayseq.run(ring[:6], state_n, 6, 0)           # (untimed: the page-locked rings of both plans are made and first used here)
                hsteps = max(2, min(a.steps, 6))
                hs = dayseq.run(ring, state_n, hsteps, 1)
                per = hs["el"] / hsteps
                rows_o = router.my_out0_global
                extra["hourly_output"]["in_sequence"] = {
                    "value": nseg * a.nsteps / per, "ms_per_step": per * 1e3, "steps": hsteps,
                    "outlet_rows_equal_the_fetched_hydrographs": bool(np.array_equal(
                        hs["fvd"][rows_o, :, 0].view(np.uint32), hs["hyd"][:, a.qts - 1::a.qts].view(np.uint32))),
                    "what": "the headline's pipeline with every row's (q, v, d) at every qts-th step among each day's products, "
                            "decimated on the device and copied beside the next day"}
                del hs
            except Exception as e:
                extra["hourly_output"]["in_sequence"] = {"error": repr(e)}
            finally:
                dayseq.set_output_stride(None)
            if use_stream:
                try:      # ... and as a product of the STREAM's days (the tiles write the kept steps aside as they go; nothing else of the result is assembled)
                    with RouteStream(srouter, a.nsteps, a.qts, output_stride=a.qts) as rs:
                        rs.run(ring[:4], state_n, 2, 0, prepared=True)
                        hsteps = max(2, min(a.steps, 12))     # (the clock also holds the last day's block on its way out: 14.5 ms once)
                        hs = rs.run(ring, state_n, hsteps, 1, prepared=True)
                        rows_o = np.array(rs.outlet_rows, copy=True)
                    per = hs["el"] / hsteps
                    extra["hourly_output"]["in_stream"] = {
                        "value": nseg * a.nsteps / per, "ms_per_step": per * 1e3, "steps": hsteps,
                        "outlet_rows_equal_the_fetched_hydrographs": bool(np.array_equal(
                            hs["fvd"][rows_o, :, 0].view(np.uint32), hs["hyd"][:, a.qts - 1::a.qts].view(np.uint32))),
                        "what": "the headline's stream with every row's (q, v, d) at every qts-th step among each day's products"}
                    fvd_ref, hyd_ref, fin_ref = np.array(hs["fvd"], copy=True), np.array(hs["hyd"], copy=True), np.array(hs["final"], copy=True)
                    del hs
                    # ... and with trmc_plan_options.velocity_on_demand: a step's velocity formed only where it is handed on (the kept
                    # steps here; nowhere when the products are hydrographs and states) -- the same days on a plan made with the option
                    vr = make_router(hint, True, None, None, options={"velocity_on_demand": 1}, stream=True)
                    try:
                        with RouteStream(vr, a.nsteps, a.qts, output_stride=a.qts) as rs:
                            rs.run(ring[:4], state_n, 2, 0, prepared=True)
                            hv = rs.run(ring, state_n, hsteps, 1, prepared=True)
                        same_h = bool(np.array_equal(hv["fvd"].view(np.uint32), fvd_ref.view(np.uint32)) and
                                      np.array_equal(hv["hyd"].view(np.uint32), hyd_ref.view(np.uint32)) and
                                      np.array_equal(hv["final"].view(np.uint32), fin_ref.view(np.uint32)))
                        per_h = hv["el"] / hsteps
                        del hv, fvd_ref
                        psteps = max(2, min(a.steps, 12))
                        with RouteStream(srouter, a.nsteps, a.qts) as rs:
                            p0 = rs.run(ring, state_n, psteps, 1, prepared=True)
                        hyd0, fin0, per0 = np.array(p0["hyd"], copy=True), np.array(p0["final"], copy=True), p0["el"] / psteps
                        del p0
                        with RouteStream(vr, a.nsteps, a.qts) as rs:
                            rs.run(ring[:4], state_n, 2, 0, prepared=True)
                            p1 = rs.run(ring, state_n, psteps, 1, prepared=True)
                        per1 = p1["el"] / psteps
                        extra["velocity_on_demand"] = {
                            "products_only": {"ms_per_step": per1 * 1e3, "value": nseg * a.nsteps / per1, "steps": psteps,
                                              "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / per1 / 1e9 / HBM_PEAK_GBS,
                                              "same_run_with_every_velocity_ms": per0 * 1e3,
                                              "hydrographs_and_final_state_bit_identical": bool(
                                                  np.array_equal(p1["hyd"].view(np.uint32), hyd0.view(np.uint32)) and
                                                  np.array_equal(p1["final"].view(np.uint32), fin0.view(np.uint32)))},
                            "hourly_output": {"ms_per_step": per_h * 1e3, "steps": hsteps,
                                              "block_hydrographs_final_state_bit_identical_to_in_stream": same_h},
                            "what": "NOT the headline: the stream on a plan made with trmc_plan_options.velocity_on_demand -- the velocity "
                                    "of a step (it feeds nothing: f90:163-169 forms it from the final depth) is computed only where it "
                                    "is handed on"}
                        del p1
                    finally:
                        vr.close()
                except Exception as e:
                    import traceback
                    traceback.print_exc()
                    extra["hourly_output"].setdefault("in_stream", {"error": repr(e)})
                    extra["velocity_on_demand"] = {"error": repr(e)}
            try:
                pass
            finally:
                for pl_ in dayseq.plans:
                    pl_._fetch_ring = None
                _tl.pinned_pool_clear()
        # the window the plan was tuned on (day N, warm), a cold start (round 1's configuration), and an unrelated day
        spin_up(router, False)
        router.upload(a.nsteps, qlat_a, None)
        w = timed(router, True, usteps, 1)
        extra["tuned_window_warm"] = {"value": rate(w), "unit": "segment-timesteps/s", "ms_per_step": w["el"] / usteps * 1e3,
                                      "ms_main": w["ms_main"], "roofline_frac": frac(w),
                                      "window": "day N itself (the plan was tuned on it), warm start"}
        router.upload(a.nsteps, qlat_s, q0)
        w = timed(router, True, usteps, 1)
        extra["cold_start"] = {"value": rate(w), "unit": "segment-timesteps/s", "ms_per_step": w["el"] / usteps * 1e3,
                               "ms_main": w["ms_main"], "roofline_frac": frac(w), "window": "day N-1, cold start"}
        router.upload(a.nsteps, synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 7), q0)
        w = timed(router, True, usteps, 1)
        extra["independent_forcing_cold"] = {"value": rate(w), "unit": "segment-timesteps/s",
                                             "ms_per_step": w["el"] / usteps * 1e3, "ms_main": w["ms_main"],
                                             "roofline_frac": frac(w),
                                             "window": "an independent draw of the forcing (no row keeps its magnitude), cold start"}
    # ---- the same sequence in TOLERANCE arithmetic (trmc_plan_options.arithmetic; never the headline) -----------------------
    tolerance = None
    if not use_dist and not a.no_tolerance and a.precision == 32:
        try:
            rt = make_router(hint, True, qlat_a, state_n, options={"arithmetic": "tolerance"})
            with DaySequence(rt, a.nsteps, a.qts) as ts:
                ts.run(ring[:6], state_n, 6, 0)
                tsteps = max(2, min(a.steps, 6))
                s3 = ts.run(ring, state_n, tsteps, 1)
            per = s3["el"] / tsteps
            tolerance = {"value": nseg * a.nsteps / per, "unit": "segment-timesteps/s", "ms_per_step": per * 1e3, "steps": tsteps,
                         "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / per / 1e9 / HBM_PEAK_GBS,
                         "outlet_hydrographs_rel_max_vs_exact": None,
                         "what": "the headline's pipeline and days on a plan created with TRMC_ARITH_TOLERANCE (hardware log2 / exp2 power, "
                                 "reciprocal-multiply division): not bit-comparable; stated tolerance and its test: include/trmc.h, "
                                 "tests/test_gpu_tolerance.py, profiles/r05_tolerance_report.json"}
            rt.close()
            if use_stream:      # ... and as the headline's STREAM of days (with and without the velocities nobody is handed)
                for key, opt in (("in_stream", {"arithmetic": "tolerance"}),
                                 ("in_stream_velocity_on_demand", {"arithmetic": "tolerance", "velocity_on_demand": 1})):
                    rt = make_router(hint, True, None, None, options=opt, stream=True)
                    try:
                        with RouteStream(rt, a.nsteps, a.qts) as ts:
                            ts.run(ring[:4], state_n, 2, 0, prepared=True)
                            tsteps = max(2, min(a.steps, 12))
                            s4 = ts.run(ring, state_n, tsteps, 1, prepared=True)
                        per = s4["el"] / tsteps
                        tolerance[key] = {"ms_per_step": per * 1e3, "steps": tsteps, "value": nseg * a.nsteps / per,
                                          "roofline_frac": nseg * a.nsteps * ALG_BYTES_PER_SEGSTEP / per / 1e9 / HBM_PEAK_GBS}
                        del s4
                    finally:
                        rt.close()
        except Exception as e:
            tolerance = {"error": repr(e)} if tolerance is None else dict(tolerance, stream_error=repr(e))
    extra["value_tolerance"] = tolerance
    # ---- the checker, LAST of the legs on this router (it runs OpenMP in this process and pins large host arrays: the copy
    # legs above are timed before it) ------------------------------------------------------------------------------------
    if rank == 0 and not a.no_parity_full and a.precision == 32 and not use_dist:
        try:      # against the reference on the CPU (checker use, outside the clock)
            # the timed pipeline once more, untimed, over the first two days of the ring (days N+1 and N+2 from the state
            # after day N: plan and clone, staged forcing, state handed over on the device, asynchronous fetch), and the
            # reference on the CPU through ALL the days from the cold start: N-1, N, N+1, N+2
            if use_stream:
                # the timed pass's pipeline -- the stream -- once more, untimed, over days N+1 and N+2 from the state after day N,
                # this time with every day's full result assembled and day N+2's handed over with its other products
                with RouteStream(srouter, a.nsteps, a.qts, full_output=True) as rs:
                    for item in rs.route(ring[:2], state_n, prepared=True):       # (exactly these two days; the last item is day N+2)
                        chk = {"hyd": item[1], "final": item[2], "fvd": item[3]}
                    o_rows = np.array(rs.outlet_rows, copy=True)
                parity = parity_full(net, router, (qlat_s, qlat_a, ring[0], ring[1]), q0, a.nsteps, a.qts, outlets=(o_rows, chk["hyd"]),
                                     threads=a.cpu_threads, final_fetched=chk["final"], fvd_fetched=chk["fvd"])
                parity["pipeline"] = ("the timed pass's pipeline (ONE stream of tile launches over the days, forcing staged from page-locked "
                                      "memory, a ring of day slots in HBM, products copied beside the launches that follow) re-run untimed "
                                      "over days N+1, N+2 with the full result of every day assembled; day N+2's compared")
                del chk
            else:
                chk = dayseq.run(ring[:2], state_n, 2, 0)
                parity = parity_full(net, router, (qlat_s, qlat_a, ring[0], ring[1]), q0, a.nsteps, a.qts,
                                     outlets=(router.my_out0_global, chk["hyd"]), threads=a.cpu_threads, plan=chk["last_plan"],
                                     final_fetched=chk["final"])
                parity["pipeline"] = ("the timed pass's pipeline (plan + clone, forcing staged from page-locked memory, state handed "
                                      "over in HBM, asynchronous fetch) re-run untimed over days N+1, N+2")
        except Exception as e:
            parity = {"error": repr(e)}
    dayseq.close()
    router.close()
    if srouter is not None:
        srouter.close()

    full = None
    if not a.no_full_ts:
        fsteps = max(1, min(a.steps, 3))
        rf = make_router(None, False, qlat_b, q0)
        f = timed(rf, False, fsteps, 1)
        full = {"value": rate(f), "unit": "segment-timesteps/s", "ms_per_step": f["el"] / fsteps * 1e3,
                "launches": f["launches"], "ms_main": f["ms_main"], "roofline_frac": frac(f),
                "engine": getattr(rf.plan0, "engine", "levels"), "window": "day N+1 forcing, cold start"}
        if world == 1:
            try:      # the general mode's sequence form: D days as ONE window of D x nsteps steps (the same values: a window's end
                # state is the next one's start, AbstractNetwork.py:177-191) -- the ramp over the network's levels is paid once
                D = 4
                nq1 = a.nsteps // a.qts
                qD = np.ascontiguousarray(np.concatenate([ring[k % len(ring)][:, :nq1] for k in range(D)], axis=1))
                pD = rf.plan0
                pD.upload_forcing(a.nsteps * D, qD, q0)
                stD = pD.route_device(a.nsteps * D, a.qts, False)
                full["days_as_one_window"] = {"days": D, "ms_per_day": stD["ms_main"] / D, "roofline_frac":
                                              nseg * a.nsteps * 64 / (stD["ms_main"] / D * 1e-3) / 8e12,
                                              "what": f"{D} days' forcing side by side, one window of {a.nsteps * D} steps on the same plan "
                                                      "(device time of the window / days)"}
                del qD
            except Exception as e:
                full["days_as_one_window"] = {"error": repr(e)}
        rf.close()

    transport = None
    if comm is not None:      # the last thing the ranks do together: what follows is rank 0's alone (checker, counter passes)
        transport = comm.backend
        comm.barrier()
        comm.close()
        comm = None
    if dist_outlets is not None and not a.no_parity_full and a.precision == 32:
        try:      # the job's product -- the all-gathered outlet block of the last timed window -- against the reference on the CPU
            parity = parity_full(net, None, (qlat_s, qlat_a, ring[0], ring[1]), q0, a.nsteps, a.qts, outlets=dist_outlets,
                                 threads=a.cpu_threads)
            parity["pipeline"] = "the timed pass's pipeline re-run untimed over days N+1, N+2 on every rank; the all-gathered outlet block of day N+2"
        except Exception as e:
            parity = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(net, qlat_s, a.nsteps, a.qts, True, a.cpu_seconds, a.cpu_threads)
            cpu.pop("_check", None)
        except Exception as e:  # the baseline must never take the GPU number down with it
            cpu = {"error": repr(e)}

    dropin = None
    if rank == 0 and world == 1 and not a.no_dropin:
        try:
            dropin = dropin_leg(net, a.nsteps, a.qts, [qlat_s, qlat_a, qlat_b])
            from troute_amd.routing.fast_reach import mc_reach as _MR
            _MR._PLANS.clear()
            _MR._FLAT.clear()
        except Exception as e:
            dropin = {"error": repr(e)}
    diffusive = None
    if rank == 0 and world == 1 and not a.no_diffusive:
        try:
            diffusive = diffusive_leg()
        except Exception as e:
            diffusive = {"error": repr(e)}

    if rank == 0:
        tname = "float" if a.precision == 32 else "double"
        st0 = stats["phase0"]
        launches = max(head["launches"], 1)
        if engine == "levels" and st0.get("wide_launches", 0) > 0:
            # the window has two kernels: the wide levels K steps per launch (k_mc_tile, the dominant one: most of the rows,
            # most of the time) and the narrow tail one step per launch beside it.  achieved / frac are the dominant
            # kernel's -- its algorithmic bytes per launch over its average launch duration, HIP events around every one of
            # its launches -- and `window` is the same arithmetic for the window as a whole (every kernel, ms_main).
            wl, wss = st0["wide_launches"], st0["wide_segment_steps"]
            kernel, pat = f"k_mc_tile<{tname}>", "k_mc_tile"
            k_ms, k_launches, k_bytes = st0["ms_wide"], wl, wss * bytes_per / wl
            tail = {"kernel": f"k_mc_step<{tname},true>", "launches_per_step": launches - wl,
                    "segment_steps": int(seg0 - wss), "runs": "on the tail stream, beside the wide launches"}
            if head.get("stream"):
                tail = {"kernel": f"k_mc_ctile<{tname}>", "launches_per_step": launches - wl, "segment_steps": int(seg0 - wss),
                        "runs": "the rows below the slices, in clusters, on the plan's own stream beside the slices' launches; K steps per launch too"}
        else:
            kernel = {"levels": f"k_mc_step<{tname},true>", "flow": "k_mc_flow_lean / k_mc_flow<true>"}[engine]
            pat = "k_mc_step" if engine == "levels" else "k_mc_flow"
            k_ms, k_launches, k_bytes = head["ms_main"], launches, seg0 * bytes_per / launches
            tail = None
        k_achieved = k_bytes * k_launches / (k_ms * 1e-3) / 1e9
        pmc = pmc_counters(pat, k_launches, a, stream=None if not head.get("stream") else
                           {"lag_max": head["stream"]["lag_max"], "launches_per_day": launches}) if rank == 0 else {}
        valu = pmc.get("valu")
        dominant = {
            "kernel": kernel, "achieved": k_achieved, "frac": k_achieved / HBM_PEAK_GBS,
            "launches_per_step": k_launches, "avg_launch_ms": k_ms / k_launches, "alg_bytes_per_launch": k_bytes,
            "traffic": pmc.get("traffic"), "valu": valu,
            "what": "algorithmic bytes of one launch over its mean duration, HIP events around every launch on its stream",
        }
        if head.get("stream"):
            dominant["what"] = ("algorithmic bytes of one launch over its mean duration: HIP events on the tile stream around the nsteps / K "
                                "launches of each timed day (they follow each other without a gap), / their number")
        if tail is not None:
            # its launches share the device with the tail's: the duration above is that of a kernel that has part of the
            # device; alone (the serialised launches of the counter passes) it is shorter
            dominant["what"] += " -- WHILE the tail's launches run beside it"
            if valu:
                dominant["frac_alone_under_counters"] = k_bytes / (valu["launch_us_under_counters"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        if tail is not None:
            # One pass of the hot path here is one WINDOW, and its launches overlap: two kernels on two streams (the wide
            # levels K steps per launch, the narrow tail one step per launch beside them) plus the transposing pass of the
            # tail's rows.  A launch of one of them has only part of the device while it runs, so no single kernel's
            # launch duration prices the path; the figure that does is the window's -- the algorithmic bytes of ALL its
            # segment-steps over the device time of ALL its kernels (HIP events on the plan's stream around the window) --
            # which no overlap can flatter.  `dominant_kernel` keeps the per-kernel arithmetic beside it.
            roof = {
                "bound": "hbm", "kernel": (f"{kernel} + {tail['kernel']}, concurrent streams: one day of the stream" if head.get("stream") else
                                           f"{kernel} + {tail['kernel']} + k_emit<{tname}>, concurrent streams: one routing window"),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc.get("window_traffic"),
                "launches_per_step": launches, "avg_launch_ms": head["ms_main"], "alg_bytes_per_launch": seg0 * bytes_per,
                "per": "window (all kernels of one pass; traffic = HBM bytes of every dispatch of the last window)",
                "valu": valu, "valu_instructions_per_window": pmc.get("window_valu_instructions"),
                "dominant_kernel": dominant,
            }
        else:
            roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            roof.update(dominant)
        roof.update({
            "window": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "ms_main": head["ms_main"],
                       "launches": launches, "what": "all segment-steps of the window x 64 B over the device time of all its kernels"},
            "tail": tail,
            "wide_levels": st0.get("wide_levels", 0), "wide_k": st0.get("wide_k", 0),
            "ms_main": head["ms_main"], "ms_total_device": head["ms_total"],
        })
        line = {
            "metric": "segment-timesteps/sec, CONUS NHD 2.7M-seg MC",
            "value": value,
            "unit": "segment-timesteps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": head["el"] / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if a.precision == 32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic CONUS NHDPlus-shaped network, MC-only, 24 h @ 300 s dt (configs[2])",
                "segments": int(nseg), "networks": int(len(net["net_sizes"])), "timesteps": a.nsteps,
                "qts_subdivisions": a.qts, "assume_short_ts": True,
                "timed_window": ("consecutive days with distinct forcing after day N-1 (spin-up from cold) and day N (tuning) as ONE stream of "
                                 "tile launches (troute_amd.sequence.RouteStream): forcing host-to-device, the state carried on in HBM, outlet "
                                 "hydrographs + final state of every day to the host, all inside the clock; the clock covers `steps` days of "
                                 "launches between two device synchronisations, the stream full (every launch carries every row)"
                                 if head.get("stream") else
                                 "a sequence of consecutive days with distinct forcing after day N-1 (spin-up from cold) and day N (tuning): forcing "
                                 "host-to-device, state handed on in HBM, outlet hydrographs + final state to the host, all inside the clock, "
                                 "on a plan and its clone") if seq is not None else
                                "day N+1 of three consecutive days of the same basin (N-1 spin-up from cold, N tuning, N+1 timed), warm start from the state day N leaves in HBM",
                "segment_levels": int(info["nlevels"]), "reach_depth": int(net["reach_depth"]),
                "sharding": "independent networks + dominant basin cut at tributary mouths" if world > 1 else "none",
                "transport": transport,
                "rank_pace_on_tuning_day": None if tuned["speed"] is None else [round(float(x), 3) for x in tuned["speed"]],
                "rank_ms_before_each_rebalance": tuned.get("feedback_ms"),
                "engine": engine,
                "generate_s": round(t_gen, 2), "plan_s": round(t_plan, 2),
                "days_in_the_ring": None if seq is None else ndays, "days_made_s": None if seq is None else round(t_days, 2),
                "plan_order": "rows grouped by their secant-iteration cost over day N (untimed tuning window); timed on day N+1"
                if not a.no_retune else "topological only", "tune_s": round(t_tune, 2),
            },
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity_full": parity,
            "forcing_persistence": persist,
            "stream": head.get("stream"),
            "stream_error": stream_err,
            "pipeline_two_plans": two_plans,
            "untuned": untuned,
            "full_ts": full,
            "per_rank": per_rank,
            "outlet_hydrographs": list(hyd.shape),
            "diffusive": diffusive,
            "dropin": dropin,
        }
        line.update(extra)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()


def pmc_counters(pattern, launches_per_window, args, stream=None):
    """Hardware counters of the dominant kernel, MEASURED IN THIS RUN: three counter-only rocprofv3 passes (--kernel-trace
    only, as MI355X_MICROARCH.md prescribes: FETCH_SIZE; WRITE_SIZE; the SQ instruction / cycle counters) over a child run of
    this script that stops after one headline window (--headline-only), read per dispatch for the LAST window's launches of
    the kernel -- the timed configuration, not an average over tuning and warm-up windows.
      traffic  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (units of KB; gfx950: FETCH_SIZE counts half of
               the streamed reads).  Both factors are CALIBRATED for this path's access patterns on this part
               (tools/traffic_calib.hip / .sh, profiles/r04_traffic_calibration.json: kernels that move a known 2 GiB --
               4-byte-per-lane unit-stride reads 2 048 bytes per FETCH_SIZE unit, the same as the guide's 16-byte pattern;
               4-byte-per-lane stores 1 024 bytes per WRITE_SIZE unit; the 96-byte runs of out[row][step][q,v,d] 966)
      valu     instructions_per_wave_step (SQ_INSTS_VALU / SQ_WAVES / timesteps a launch routes), cycles_per_instruction
               (4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: the counter is in quad-cycles), busy_frac (VALU-active cycles per
               SIMD over the busy cycles of a shader engine: 4 x SQ_ACTIVE_INST_VALU / nSIMD  /  SQ_BUSY_CYCLES / nSE)
    {} when the profiler is not available, when this process is itself a profiled child or one rank of several, or on any
    failure -- never a figure from a file."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if args.no_traffic or os.environ.get("TRMC_BENCH_CHILD") or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return {}
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return {}                                       # this process is being profiled itself
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {}
    vals, out = {}, {}
    passes = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_CYCLES"))
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for i, counters in enumerate(passes):
                d = os.path.join(td, f"p{i}")
                cmd = [exe, "--pmc", *counters, "--kernel-trace", "-d", d, "-o", "c", "--", sys.executable,
                       os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--headline-only", "--no-cpu-baseline",
                       "--no-full-ts", "--no-diffusive", "--no-parity-mode", "--no-traffic", "--no-parity-full",
                       "--nsteps", str(args.nsteps), "--qts", str(args.qts), "--precision", str(args.precision)]
                if args.nseg:
                    cmd += ["--nseg", str(args.nseg)]
                if args.nnet:
                    cmd += ["--nnet", str(args.nnet)]
                if args.no_retune:
                    cmd += ["--no-retune"]
                env = dict(os.environ, TRMC_BENCH_CHILD="1", TMPDIR="/tmp")
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180, check=True)
                dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
                con = sqlite3.connect(dbs[0])
                disp = con.execute(
                    "select d.id, d.event_id, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                    "on d.kernel_id = s.id where s.kernel_name like ? order by d.start", (f"%{pattern}%",)).fetchall()
                if stream:      # (a stream of windows ends with `lag_max` launches that only bring the last days to their end: the
                    #  last FULL day is the launches before those)
                    disp = disp[:len(disp) - int(stream["lag_max"])]
                last = disp[-int(launches_per_window):]
                ev = [x[1] for x in last]
                q = ",".join("?" * len(ev))
                for name, total, ninst in con.execute(
                        f"select p.name, sum(e.value), count(*) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                        f"where e.event_id in ({q}) group by p.name", ev):
                    vals[name] = total / len(last)                    # per launch, summed over the hardware instances
                    vals[name + "#inst"] = ninst / len(last)          # hardware instances that report the counter
                vals["avg_us_" + str(i)] = sum(x[2] for x in last) / len(last) / 1e3
                # ... and every kernel of the last window, summed
                alld = con.execute(
                    "select d.event_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                    "on d.kernel_id = s.id order by d.start").fetchall()
                first = max(j for j, x in enumerate(alld) if "k_prep_qlat" in x[1])   # (a window starts with its forcing transpose)
                wev = [x[0] for x in alld[first:]]
                if stream:      # (the last day pushed: the tile launches of both kinds that its push queued, nothing of the flush behind them)
                    wev = [x[0] for x in alld[first:] if "k_mc_tile" in x[1] or "k_mc_ctile" in x[1]][:int(stream["launches_per_day"])]
                for lo in range(0, len(wev), 500):
                    part = wev[lo:lo + 500]
                    q = ",".join("?" * len(part))
                    for name, total in con.execute(
                            f"select p.name, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                            f"where e.event_id in ({q}) group by p.name", part):
                        vals["window:" + name] = vals.get("window:" + name, 0.0) + total
                vals["window:dispatches"] = len(wev)
                con.close()
        out["traffic"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        out["window_traffic"] = (2.0 * vals["window:FETCH_SIZE"] + vals["window:WRITE_SIZE"]) * 1024.0
        out["window_valu_instructions"] = vals["window:SQ_INSTS_VALU"]
        out["window_dispatches"] = vals["window:dispatches"]
        steps_per_launch = float(args.nsteps) / float(launches_per_window) if "k_mc_tile" in pattern else 1.0
        waves, insts, active = vals["SQ_WAVES"], vals["SQ_INSTS_VALU"], vals["SQ_ACTIVE_INST_VALU"]
        n_simd = 256 * 4
        n_se = max(vals.get("SQ_BUSY_CYCLES#inst", 32.0), 1.0)
        out["valu"] = {
            "instructions_per_wave_step": insts / waves / steps_per_launch,
            "instructions_per_launch": insts, "wavefronts_per_launch": waves,
            # (4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: NOT an issue cost -- the counter equals the instruction count for a
            # kernel of nothing but v_fma_f32, which issues every 2.9 cycles; per-class issue costs: tools/valu_rates.hip)
            "cycles_per_instruction": 4.0 * active / insts,
            "busy_frac": (4.0 * active / n_simd) / (vals["SQ_BUSY_CYCLES"] / n_se),
            # SQ_BUSY_CYCLES under-counts the cycles of a launch by about a tenth on this part (the device runs this load at
            # 2.27 GHz by rocm-smi, SQ_BUSY_CYCLES / duration says 2.0): the same ratio over SQ_CYCLES is the honest one
            "busy_frac_of_all_cycles": (4.0 * active / n_simd) / (vals["SQ_CYCLES"] / max(vals.get("SQ_CYCLES#inst", 32.0), 1.0)),
            "clock_ghz_sq_cycles": vals["SQ_CYCLES"] / max(vals.get("SQ_CYCLES#inst", 32.0), 1.0) / (vals["avg_us_2"] * 1e3),
            "mean_resident_wavefronts_per_simd": 4.0 * vals["SQ_WAVE_CYCLES"] / n_simd
                                                 / (vals["SQ_CYCLES"] / max(vals.get("SQ_CYCLES#inst", 32.0), 1.0)),
            "launch_us_under_counters": vals["avg_us_2"],
            "how": "SQ_INSTS_VALU / SQ_WAVES / steps per launch; 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU; (4 x SQ_ACTIVE_INST_VALU / "
                   f"{n_simd} SIMDs) / (SQ_BUSY_CYCLES / {int(n_se)} instances); last window's launches of {pattern}",
        }
        return out
    except Exception as e:
        out.setdefault("traffic", None)
        out["error"] = repr(e)
        return out


if __name__ == "__main__":
    main()
