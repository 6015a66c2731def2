#!/usr/bin/env python3
"""Time every rank's share of an N-GPU CONUS run on ONE GPU, one rank after the other.

The cut-edge hydrographs a rank would receive from its peers are taken from a complete single-GPU route
(so the trunk sees its true inflows); the collective is a device-to-device copy of what the peers would have sent (a stand-in for troute_amd.comm.Comm).  Reports, per rank, the wall
time of route_on_device -- the job time of the real N-GPU run is about the maximum (plus RCCL latency).

    python tools/sim_ranks.py --world 8 [--chunks 4] [--full-ts]
    python tools/sim_ranks.py --world 8 --retune --sequence 6     # the per-day PERIOD of every rank under the sequence
                                                                  # pipeline (troute_amd.sequence.DaySequence): distinct days,
                                                                  # forcing staged, state carried on in HBM, products fetched
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=None)
    ap.add_argument("--full-ts", action="store_true")
    ap.add_argument("--nseg", type=int, default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ranks", type=str, default=None, help="comma list (default: all)")
    ap.add_argument("--retune", action="store_true", help="rebuild every router with the cost hint of a tuning window")
    ap.add_argument("--sequence", type=int, default=0, help="D > 1: time D consecutive distinct days per rank through "
                    "troute_amd.sequence.DaySequence (the first is a warm-up) and report the period per day")
    ap.add_argument("--rebalance", type=int, nargs="?", const=1, default=0, help="time all ranks, repartition by their "
                    "measured pace (sharding.partition rank_speed, what bench.py does after its tuning window), time them "
                    "again; N: that many feedback steps (bench.py takes up to two more on the hinted plan)")
    ap.add_argument("--stream", type=int, default=0, help="D > 1: every rank under troute_amd.sequence.RouteStream over D distinct "
                    "days (ONE stream of tile launches per rank, the cut-edge hydrographs exchanged once a day): ms per day")
    ap.add_argument("--cycles", type=int, default=3, help="--stream: the D days are routed so many times over in one stream (a longer steady state)")
    ap.add_argument("--wide-k", type=int, default=0)
    ap.add_argument("--wide-min-rows", type=int, default=0)
    ap.add_argument("--cluster-rows", type=int, default=0)
    ap.add_argument("--wide-levels", type=int, default=0)
    ap.add_argument("--velocity-on-demand", type=int, default=0)
    a = ap.parse_args()
    from troute_amd import comm as X
    from troute_amd import sharding, synthetic
    from troute_amd.distributed import ShardedRouter

    kw = {"nseg": a.nseg, "nnet": max(3, a.nseg // 185)} if a.nseg else {}
    net = synthetic.generate(cache_dir=os.environ.get("TRMC_CACHE", "/tmp/trmc_cache"), **kw)
    to, params, qlat = net["to"], net["params"], net["qlat"]
    nseg = to.shape[0]
    q0 = np.zeros((nseg, 3), np.float32)
    nsteps, qts, short = 288, 12, not a.full_ts
    dev = 0

    part = sharding.partition(to, a.world)
    # true hydrographs of the cut rows from a whole-network route
    eng = os.environ.get("PROBE_ENGINE", "auto")
    single = ShardedRouter(to, params, assume_short_ts=short, engine=eng)
    single.upload(nsteps, qlat, q0)
    if a.retune:
        single.collect_cost(True)
    single.route_resident(qts, short)
    hint = None
    if a.retune:
        hint = single.iteration_hint()
        single.close()
        single = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=short, engine=eng)
        single.upload(nsteps, qlat, q0)
        single.route_resident(qts, short)
    t0 = time.perf_counter()
    single.route_resident(qts, short)
    t_single = time.perf_counter() - t0
    if hint is not None:                      # measured costs: the partition is packed by them (as bench.py does)
        part = sharding.partition(to, a.world, row_cost=hint)
    cut_rows = part["cut_rows"]
    cut_q = single.plan0.gather_flow_rows(cut_rows) if cut_rows.size else np.zeros((0, nsteps), np.float32)
    ref_rows = single.my_out0_global
    ref_hyd = single.outlet_hydrographs()
    single_engine = single.plan0.engine
    single.close()
    print(f"single GPU ({single_engine} engine): {t_single * 1e3:.2f} ms   cut rows {cut_rows.size}")

    if a.stream > 1:
        return stream_mode(a, net, to, params, qlat, q0, part, hint, nsteps, qts, dev, X, synthetic, ShardedRouter)
    if a.sequence > 1:
        return sequence_mode(a, net, to, params, qlat, q0, part, hint, nsteps, qts, eng, dev, X, synthetic, ShardedRouter, t_single)
    passes = [part]          # (the partition the cut-edge hydrographs above belong to keeps its cut rows: same trunks)
    times_of = {}
    for ipass in range(1 + a.rebalance):
      if ipass >= 1:
        cost = hint.astype(np.float64) if hint is not None else np.ones(nseg)
        own = part["owner"][part["piece"]]
        loads = np.bincount(own, weights=cost, minlength=a.world)
        speed = sharding.rank_speeds(loads, [times_of[k] for k in range(a.world)])
        print("measured pace of the ranks:", np.round(speed, 3))
        part = sharding.partition(to, a.world, row_cost=hint, rank_speed=speed, previous=part)
      worst = 0.0
      for rank in ([int(k) for k in a.ranks.split(',')] if a.ranks else range(a.world)):
          r = ShardedRouter(to, params, rank=rank, world=a.world, device=0, partition=part, cost_hint=hint,
                            assume_short_ts=short, engine=eng)
          default_chunks = 4 if getattr(r.plan0, "engine", "levels") == "flow" else 24     # route_on_device's defaults
          nchunks_eff = (a.chunks if a.chunks else default_chunks) if short else (a.chunks if a.chunks else 1)

          class SimComm:
              """the peers of this rank, played back: an all-gather of a time chunk of cut-edge hydrographs hands out what a
              whole-network route says the other ranks would have sent (device-to-device copies on the caller's stream);
              the outlet gather returns this rank's own block only"""
              rank, world, backend = 0, a.world, "sim"

              def __init__(self, router):
                  self.r = router
                  self.call = 0
                  self.blocks = None

              def prepare(self, K, C):
                  mc = max(self.r._max_cut, 1)
                  peers = np.zeros((a.world, mc, nsteps), np.float32)
                  if cut_rows.size:
                      idx = np.zeros(cut_rows.size, dtype=np.int64)
                      for k in range(a.world):
                          m = self.r.cut_owner == k
                          idx[m] = np.arange(int(m.sum()))
                      peers[self.r.cut_owner.astype(np.int64), idx] = cut_q
                  self.blocks = [X.DeviceBuffer.from_array(dev, np.ascontiguousarray(peers[:, :, c * K:min(nsteps, (c + 1) * K)]))
                                 for c in range(C)]

              def all_gather(self, send_ptr, recv_ptr, nbytes, stream=0):
                  c = self.call
                  self.call += 1
                  if self.r._max_cut > 0 and c < len(self.blocks):
                      X.device_copy(dev, recv_ptr, self.blocks[c].ptr, self.blocks[c].nbytes, stream)
                  X.device_copy(dev, recv_ptr + rank * nbytes, send_ptr, nbytes, stream)

              def barrier(self):
                  pass

          sim = SimComm(r)
          r.enable_device_exchange(sim, dev)
          r.upload(nsteps, qlat, q0)
          r.upload_trunk()
          if short:
              K = max(1, -(-nsteps // max(1, int(nchunks_eff))))
              sim.prepare(K, -(-nsteps // K))
          else:
              bounds = np.round(np.linspace(0, nsteps, nchunks_eff + 1)).astype(np.int64)
              sim.blocks = []
              mc = max(r._max_cut, 1)
              peers = np.zeros((a.world, mc, nsteps), np.float32)
              if cut_rows.size:
                  idx = np.zeros(cut_rows.size, dtype=np.int64)
                  for k in range(a.world):
                      m = r.cut_owner == k
                      idx[m] = np.arange(int(m.sum()))
                  peers[r.cut_owner.astype(np.int64), idx] = cut_q
              sim.blocks = [X.DeviceBuffer.from_array(dev, np.ascontiguousarray(peers[:, :, int(bounds[c]):int(bounds[c + 1])]))
                            for c in range(nchunks_eff)]
          acc = {}
          times = []
          for _ in range(a.reps + 1):
              sim.call = 0
              X.device_synchronize(dev)
              t0 = time.perf_counter()
              rows, hyd = r.route_on_device(qts, short, nchunks_eff)
              X.device_synchronize(dev)
              times.append(time.perf_counter() - t0)
          t = min(times[1:])
          worst = max(worst, t)
          times_of[rank] = t
          st = r.last_stats
          # check my own outlets against the single-GPU run
          mine = np.concatenate([r.my_out0_global, r.my_out1_global])
          h = hyd.numpy()
          sel = np.searchsorted(rows, mine)
          ok = np.array_equal(h[sel].view(np.uint32), ref_hyd[np.searchsorted(ref_rows, mine)].view(np.uint32))
          print(f"rank {rank} ({r.plan0.engine}, {st['phase0'].get('wide_levels', 0)} wide levels): {t * 1e3:7.2f} ms  phase0 {r.rows0.size} rows main {st['phase0']['ms_main']:.2f} ms"
                + (f"  trunk {r.rows1.size} rows" + (f" main {st['phase1']['ms_main']:.2f} ms" if "phase1" in st else " (skewed)") if r.plan1 is not None else "")
                + f"  outlets bit-identical: {ok}")
          if acc:
              print("   host ms (all reps):", {k: round(v * 1e3, 2) for k, v in acc.items()})
          r.close()
      print(f"max over ranks {worst * 1e3:.2f} ms -> speed-up vs single {t_single / worst:.2f}x at world {a.world}")


def sequence_mode(a, net, to, params, qlat, q0, part, hint, nsteps, qts, eng, dev, X, synthetic, ShardedRouter, t_single):
    """D consecutive distinct days: first on ONE router (the truth: every day's cut-edge and outlet hydrographs, and the
    single-GPU period under the same pipeline), then every rank of the partition by itself through DaySequence, its peers
    played back day by day."""
    from troute_amd.sequence import DaySequence, pinned_like
    nseg, D = to.shape[0], a.sequence
    days, prev = [], qlat
    for i in range(D):
        prev = synthetic.forcing(nseg, qlat.shape[1], synthetic.DEFAULT_SEED + 1 + i, previous=prev)
        days.append(prev)
    cut_rows = part["cut_rows"]
    single = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True, engine=eng)
    cut_q, ref_hyd = [], []
    state = q0
    for d in days:                                       # the truth, day by day
        single.upload(nsteps, d, state)
        state = None
        single.route_resident(qts, True)
        cut_q.append(single.plan0.gather_flow_rows(cut_rows) if cut_rows.size else np.zeros((0, nsteps), np.float32))
        ref_hyd.append(single.outlet_hydrographs())
    ref_rows = single.my_out0_global
    period_single = None
    if single.plan0.engine == "levels":
        with DaySequence(single, nsteps, qts) as ds:
            ring = [pinned_like(d) for d in days]
            ds.run(ring[:2], q0, 2, 0)
            out = ds.run(ring, q0, D - 1, 1)
            period_single = out["el"] / (D - 1)
    single.close()
    print(f"single GPU under the sequence pipeline: {period_single * 1e3 if period_single else float('nan'):.2f} ms per day "
          f"(one window alone: {t_single * 1e3:.2f} ms)")
    from troute_amd import sharding
    periods = {}
    for ipass in range(1 + a.rebalance):
      if ipass >= 1:          # the partition fed back with every rank's measured PERIOD (what bench.py does with its tuning day)
        cost = hint.astype(np.float64) if hint is not None else np.ones(nseg)
        own = part["owner"][part["piece"]]
        loads = np.bincount(own, weights=cost, minlength=a.world)
        speed = sharding.rank_speeds(loads, [periods[k] for k in range(a.world)])
        print("measured pace of the ranks:", np.round(speed, 3))
        part = sharding.partition(to, a.world, row_cost=hint, rank_speed=speed, previous=part)
      worst = 0.0
      for rank in ([int(k) for k in a.ranks.split(',')] if a.ranks else range(a.world)):
          r = ShardedRouter(to, params, rank=rank, world=a.world, device=0, partition=part, cost_hint=hint,
                            assume_short_ts=True, engine=eng)

          class SimComm:
              """the peers of this rank, played back day by day (see main())"""
              rank, world, backend = 0, a.world, "sim"

              def __init__(self, router):
                  self.r, self.call, self.blocks, self.per_day = router, 0, None, 1

              def prepare(self, K, C):
                  mc = max(self.r._max_cut, 1)
                  idx = np.zeros(cut_rows.size, dtype=np.int64)
                  for k in range(a.world):
                      m = self.r.cut_owner == k
                      idx[m] = np.arange(int(m.sum()))
                  self.blocks = []
                  for dq in cut_q:
                      peers = np.zeros((a.world, mc, nsteps), np.float32)
                      if cut_rows.size:
                          peers[self.r.cut_owner.astype(np.int64), idx] = dq
                      self.blocks.append([X.DeviceBuffer.from_array(dev, np.ascontiguousarray(peers[:, :, c * K:min(nsteps, (c + 1) * K)]))
                                          for c in range(C)])
                  self.per_day = (C if self.r._max_cut > 0 else 0) + 1

              def all_gather(self, send_ptr, recv_ptr, nbytes, stream=0):
                  day, c = divmod(self.call, self.per_day)
                  self.call += 1
                  blocks = self.blocks[day % len(self.blocks)]
                  if self.r._max_cut > 0 and c < len(blocks):
                      X.device_copy(dev, recv_ptr, blocks[c].ptr, blocks[c].nbytes, stream)
                  X.device_copy(dev, recv_ptr + rank * nbytes, send_ptr, nbytes, stream)

              def barrier(self):
                  pass

              def all_reduce_max_host(self, x):
                  return x

          sim = SimComm(r)
          r.enable_device_exchange(sim, dev)
          r.nsteps = nsteps
          ds = DaySequence(r, nsteps, qts, nchunks=a.chunks, hydrographs_on_every_rank=True)
          K, C, _ = r._chunking(a.chunks)
          sim.prepare(K, C)
          local = ds.prepare_days(days)
          got = {}
          mine = np.concatenate([r.my_out0_global, r.my_out1_global])
          best = None
          for rep in range(max(1, a.reps)):
              sim.call = 0
              got.clear()
              out = ds.run(local, q0, D - 1, 1, prepared=True,
                           on_day=lambda w, h, s: got.__setitem__(w, None if h is None else np.array(h[np.searchsorted(r._out_rows, mine)], copy=True)))
              per = out["el"] / (D - 1)
              best = per if best is None else min(best, per)
          ok = all(np.array_equal(got[w].view(np.uint32), ref_hyd[w][np.searchsorted(ref_rows, mine)].view(np.uint32)) for w in range(D))
          worst = max(worst, best)
          periods[rank] = best
          st = r.last_stats["phase0"]
          print(f"rank {rank} ({r._state_plans[0].engine}, {st.get('wide_levels', 0)} wide levels): period {best * 1e3:7.2f} ms per day  "
                f"(device window {np.mean(out['ms_main']):.2f} ms)  rows {r.sequence_rows().size}  outlets of every day bit-identical: {ok}")
          ds.close()
          r.close()
    base = period_single if period_single else t_single
    print(f"slowest rank {worst * 1e3:.2f} ms per day -> speed-up vs single GPU under the same pipeline {base / worst:.2f}x at world {a.world}")


def stream_mode(a, net, to, params, qlat, q0, part, hint, nsteps, qts, dev, X, synthetic, ShardedRouter):
    """D consecutive distinct days: the truth day by day on one router; the single-GPU stream; then every rank of the partition
    by itself under RouteStream, its peers' cut-edge hydrographs played back day by day."""
    from troute_amd.sequence import RouteStream, pinned_like
    from troute_amd import sharding
    from troute_amd.plan import topology_clusters
    from troute_amd.distributed import restrict_csr
    nseg, D = to.shape[0], a.stream
    days, prev = [], qlat
    for i in range(D):
        prev = synthetic.forcing(nseg, qlat.shape[1], synthetic.DEFAULT_SEED + 1 + i, previous=prev)
        days.append(prev)
    cut_rows = part["cut_rows"]
    single = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True)
    cut_q, ref_hyd = [], []
    state = q0
    for d in days:                                       # the truth, day by day
        single.upload(nsteps, d, state)
        state = None
        single.route_resident(qts, True)
        cut_q.append(single.plan0.gather_flow_rows(cut_rows) if cut_rows.size else np.zeros((0, nsteps), np.float32))
        ref_hyd.append(single.outlet_hydrographs())
    ref_rows = single.my_out0_global
    single.close()
    opts = {k: v for k, v in (("wide_k", a.wide_k), ("wide_min_rows", a.wide_min_rows), ("cluster_rows", a.cluster_rows),
                              ("wide_levels", a.wide_levels), ("velocity_on_demand", a.velocity_on_demand)) if v}
    ring = [pinned_like(d) for d in days]

    def timed(router, comm_days=None):
        """(ms per day over the whole stream, median ms between deliveries, outlets of every day, info)"""
        best = None
        for rep in range(max(1, a.reps)):
            if comm_days is not None:
                comm_days.call = 0
            got, marks = {}, []
            X.device_synchronize(dev)
            t0 = time.perf_counter()
            with RouteStream(router, nsteps, qts, hydrographs_on_every_rank=True) as rs:
                local = rs.prepare_days(days)
                X.device_synchronize(dev)
                t0 = time.perf_counter()
                for item in rs.route(local * a.cycles, q0, prepared=True):
                    marks.append(time.perf_counter())
                    if item[0] < D:
                        got[item[0]] = (np.array(item[1], copy=True), rs.outlet_rows)
                info = rs.plan.stream_info()
            X.device_synchronize(dev)
            el = (time.perf_counter() - t0) / (D * a.cycles)
            # the steady state: the deliveries in the middle of the stream (the first ones wait for the pipeline to fill, the
            # last ones come in a burst after the flush)
            n = len(marks)
            lo, hi = n // 4, max(n // 4 + 1, n - 1 - max(3, n // 4))
            steady = (marks[hi] - marks[lo]) / (hi - lo) if hi > lo else el
            if best is None or steady < best[1]:
                best = (el, steady, got, info)
        return best
    one = ShardedRouter(to, params, cost_hint=hint, stream=True, options=opts)
    el1, st1, got1, info1 = timed(one)
    ok1 = all(np.array_equal(got1[w][0].view(np.uint32), ref_hyd[w][np.searchsorted(ref_rows, got1[w][1])].view(np.uint32)) for w in range(D))
    one.close()
    print(f"single GPU as a stream: {st1 * 1e3:.2f} ms per day between deliveries ({el1 * 1e3:.2f} over all {D} days with fill and drain)  "
          f"{info1['wide_levels']} slices + {info1['cluster_levels']} cluster levels, lag {info1['lag_max']} tiles, {info1['slots']} slots  "
          f"outlets of every day bit-identical: {ok1}", flush=True)
    periods = {}
    for ipass in range(1 + a.rebalance):
      if ipass >= 1:
        cost = hint.astype(np.float64) if hint is not None else np.ones(nseg)
        own = part["owner"][part["piece"]]
        loads = np.bincount(own, weights=cost, minlength=a.world)
        speed = sharding.rank_speeds(loads, [periods[k] for k in range(a.world)])
        print("measured pace of the ranks:", np.round(speed, 3))
        part = sharding.partition(to, a.world, row_cost=hint, rank_speed=speed, previous=part)
        if not np.array_equal(part["cut_rows"], cut_rows):
            print("(the repartition moved the cuts: stopping the feedback here)")
            break
      worst = 0.0
      # the largest lag among ALL ranks' cut rows (every rank of a real job learns it by an all-reduce): from the plans' own
      # topology routine on every rank's sub-basins, host only
      up_ptr, up_idx = synthetic.upstream_csr(to)
      piece, phase, owner = part["piece"], part["phase"], part["owner"]
      pmax = 0
      for k in range(a.world):
          rows0 = np.flatnonzero((phase[piece] == 0) & (owner[piece] == k))
          g2l = np.full(nseg, -1, dtype=np.int64)
          g2l[rows0] = np.arange(rows0.shape[0])
          lp, li = restrict_csr(up_ptr, up_idx, rows0, g2l)
          mine = cut_rows[owner[piece][cut_rows] == k] if cut_rows.size else cut_rows
          if mine.size:
              _, lag, _, _, _, _ = topology_clusters(lp, li, cost_hint=None if hint is None else hint[rows0],
                                                     wide_min_rows=opts.get("wide_min_rows", 1024), wide_max_levels=32, cluster_rows=128)
              pmax = max(pmax, int(lag[g2l[mine]].max()))
      for rank in ([int(k) for k in a.ranks.split(',')] if a.ranks else range(a.world)):
          r = ShardedRouter(to, params, rank=rank, world=a.world, device=0, partition=part, cost_hint=hint, stream=True, options=opts)

          class SimComm:
              """the peers of this rank, played back: the all-gather of day e's cut-edge hydrographs hands out what the whole-network
              route says the other ranks would have sent"""
              rank, world, backend = 0, a.world, "sim"

              def __init__(self, router):
                  self.r, self.call = router, 0
                  mc = max(router._max_cut, 1) if hasattr(router, "_max_cut") else 1
                  self.blocks = None

              def prepare(self):
                  mc = max(self.r._max_cut, 1)
                  idx = np.zeros(cut_rows.size, dtype=np.int64)
                  for k in range(a.world):
                      m = self.r.cut_owner == k
                      idx[m] = np.arange(int(m.sum()))
                  self.blocks = []
                  for dq in cut_q:
                      peers = np.zeros((a.world, mc, nsteps), np.float32)
                      if cut_rows.size:
                          peers[self.r.cut_owner.astype(np.int64), idx] = dq
                      self.blocks.append(X.DeviceBuffer.from_array(dev, peers))

              def all_gather(self, send_ptr, recv_ptr, nbytes, stream=0):
                  day = self.call
                  self.call += 1
                  blk = self.blocks[day % len(self.blocks)]
                  X.device_copy(dev, recv_ptr, blk.ptr, blk.nbytes, stream)
                  X.device_copy(dev, recv_ptr + rank * nbytes, send_ptr, nbytes, stream)

              def all_gather_rows_host(self, arr):
                  return [arr]

              def all_reduce_max_host(self, x):
                  return np.maximum(x, pmax) if x.shape == (1,) and self.r.my_cut_local is not None and x[0] <= 200 and self._first else x

              def barrier(self):
                  pass

          sim = SimComm(r)
          sim._first = True
          r.enable_device_exchange(sim, dev)
          sim.prepare()
          # (the first all-reduce of a stream is the cut rows' lag, the second the plans' largest lag: only the first is global here)
          calls = {"n": 0}

          def armax(x, sim=sim, calls=calls):
              calls["n"] += 1
              return np.maximum(x, pmax) if calls["n"] % 2 == 1 else x
          sim.all_reduce_max_host = armax
          el, steady, got, info = timed(r, sim)
          mine = r._outS_global
          ok = all(np.array_equal(got[w][0][np.searchsorted(got[w][1], mine)].view(np.uint32),
                                  ref_hyd[w][np.searchsorted(ref_rows, mine)].view(np.uint32)) for w in range(D))
          worst = max(worst, steady)
          periods[rank] = steady
          print(f"rank {rank}: {steady * 1e3:7.2f} ms per day between deliveries ({el * 1e3:.2f} over all {D} days)  rows {r._rowsS.size}  "
                f"{info['wide_levels']} slices + {info['cluster_levels']} cluster levels, lag {info['lag_max']} tiles, {info['slots']} slots, "
                f"{info['launches']} launches  outlets of every day bit-identical: {ok}", flush=True)
          r.close()
      print(f"slowest rank {worst * 1e3:.2f} ms per day -> {st1 / worst:.2f}x the single-GPU stream ({st1 * 1e3:.2f} ms) at world {a.world}", flush=True)


if __name__ == "__main__":
    main()
