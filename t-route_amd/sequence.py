"""A SEQUENCE of routing windows ("days") on one network -- what an operational cycle does with the reference: every call of
``compute_nhd_routing_v02`` gets that window's ``qlat_values`` (mc_reach.pyx:173,:723) and starts from the state the window
before left (``AbstractNetwork.new_q0``, AbstractNetwork.py:177-191).  Here the windows follow each other on the device
without the host in between:

  * every day's forcing travels host -> device on a copy stream while the day before is being routed (page-locked memory,
    ``trmc_stage_forcing``);
  * the state is handed from day to day in HBM (``trmc_plan_chain_from`` between a plan and its clone on one GPU; the
    resident state of the merged plan on a rank of a multi-GPU job);
  * every day's products -- the outlet hydrographs and the final state, SURVEY 8d's throughput mode -- are copied to
    page-locked host arrays beside the next day's kernels (``trmc_fetch_begin``) and handed to the caller a day later.

One GPU (``world == 1``): the days take turns on the router's plan and a clone of it (ONE copy of topology and parameter
columns in HBM, two sets of window buffers), both in sequence mode (trmc_plan_options.sequence_mode), so that day w + 1's
leading levels run while day w's narrow levels are finishing.  The order in which a day's pieces are queued is what the
shared hardware queues need ("A sequence of days"): the window; BEHIND its last launch the gathers of its
products and their copy to the host; behind its set-up the forcing of the plan's NEXT day, two days ahead.

Several ranks (``ShardedRouter`` with a communicator): the same protocol on every rank's merged plan (its sub-basins plus
the time-skewed trunk it owns): the day's forcing of the rank's rows staged from page-locked memory, the cut-edge
hydrographs exchanged chunk by chunk (``ShardedRouter.route_on_device``), the outlet block gathered, the rank's final
state and the block fetched beside the next day.

Reference analogue of the loop: nwm_routing/__main__.py:1112-1226 (``nwm_route`` per run set, ``new_q0`` between them).
"""
import time

import numpy as np

from . import _lib


def pinned_like(array):
    """A page-locked copy of a forcing array (what ``stage_forcing`` wants to copy from without a wait)."""
    out = _lib.result_empty(array.shape, array.dtype, always_pinned=True)
    out[...] = array
    return out


class DaySequence:
    """``DaySequence(router, nsteps, qts_subdivisions)`` then ``run(days, state0, steps, warmup)``.

    router : a ``ShardedRouter`` (one rank of a job of any size; ``enable_device_exchange`` done when world > 1)
    output_stride : None, or n -- every n-th step of every row's (q, v, d) with each day's products
    """

    def __init__(self, router, nsteps, qts_subdivisions, assume_short_ts=True, nchunks=None, hydrographs_on_every_rank=False,
                 output_stride=None, timeline=False):
        if not assume_short_ts:
            raise ValueError("a pipelined sequence of windows needs assume_short_ts (a day's leading levels run ahead of the "
                             "day before's narrow ones only there); route general-mode windows one by one")
        self.r = router
        self.nsteps, self.qts, self.nchunks = int(nsteps), int(qts_subdivisions), nchunks
        self.world = router.world
        self._hyd_everywhere = bool(hydrographs_on_every_rank)   # (default: the gathered outlet block goes to rank 0's host only)
        # every output_stride-th step of (q, v, d) of every row as a further product of each day -- what the reference's
        # writers take at stream_output_internal_frequency (nwm_routing/output.py:209-216) -- decimated on the device and
        # copied beside the next day (trmc_fetch_begin_fvd); on_day then gets a fourth argument
        self.output_stride = None if output_stride is None else int(output_stride)
        self._clone = None
        self._local_days = None
        self.timeline = [] if timeline else None            # (diagnosis: host time, in ms, at which each call of a day returned)
        if self.world == 1:
            p = router.plan0
            if p.engine != "levels":
                raise ValueError("the one-GPU sequence pipeline runs on the level engine (a plan created for assume_short_ts "
                                 "with a million rows or more, or engine='levels')")
            self._clone = p.clone()
            self.plans = [p, self._clone]
            for pl in self.plans:
                pl.set_sequence_mode(True)
            self.set_output_stride(self.output_stride)
            self._rs = [pl.rowset(router.my_out0_local) for pl in self.plans]

    def set_output_stride(self, stride):
        """(Between runs.)  None / 0: the days' products are the outlet hydrographs and the final state; n: also every n-th step
        of every row's (q, v, d) -- the plans are told, so that their windows write the kept steps aside as they go
        (trmc_plan_set_output_stride)."""
        stride = int(stride) if stride else None
        if self.world == 1 and (stride or self.output_stride):
            for pl in self.plans:
                pl.set_output_stride(stride or 0)
        self.output_stride = stride

    def close(self):
        if self._clone is not None:
            self._clone.close()
            self._clone = None
            self.r.plan0.set_sequence_mode(False)
            if self.output_stride:
                self.r.plan0.set_output_stride(0)
                self.output_stride = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------------------------------
    def prepare_days(self, days):
        """Multi-rank jobs: this rank's rows of every day in page-locked memory (where a caller would have read them from the
        forcing file to).  One GPU: the arrays as they are (they should be page-locked: ``pinned_like``)."""
        if self.world == 1:
            return list(days)
        rows = self.r.sequence_rows()
        return [pinned_like(np.ascontiguousarray(d[rows])) for d in days]

    def run(self, days, state0, steps, warmup=0, on_day=None, prepared=False):
        """Route ``warmup + steps`` consecutive days; day w takes ``days[w % len(days)]`` (global rows; a ring that is used in
        turn -- the state evolves on, no two windows are the same work) and starts from day w - 1's final state (day 0 from
        ``state0`` [nseg, 3], or from the router's resident state if None).  The clock covers the ``steps`` days after the
        warm-up ones, everything a day needs inside it.  ``on_day(w, hydrographs, final_state)`` -- with ``output_stride``
        ``on_day(w, hydrographs, final_state, fvd)``, fvd [rows, nsteps // output_stride, 3] (one GPU: all rows in the
        caller's order; a rank: a list with the block of its ``sequence_rows()``) -- is called as each day's products arrive
        on the host (arrays of a ring of three: copy what is kept).

        Returns {"el": wall seconds of the timed days, "ms_main": [per-day device ms], "day_ms": [host-observed periods],
        "hyd", "final": the last day's products, "days_routed", "last_plan"}."""
        days = list(days) if prepared else self.prepare_days(days)
        if self.world == 1:
            return self._run_one_gpu(days, state0, steps, warmup, on_day)
        return self._run_rank(days, state0, steps, warmup, on_day)

    # ---- one GPU: a plan and its clone ---------------------------------------------------------------------------------
    def _run_one_gpu(self, days, state0, steps, warmup, on_day):
        from . import comm as X
        plans, rs, nsteps, qts = self.plans, self._rs, self.nsteps, self.qts
        nd, total = len(days), warmup + steps
        dev = plans[0].info()["device"]

        tl, tl0 = self.timeline, time.perf_counter()
        rings = [p.set_stamps(16) for p in plans] if tl is not None else None   # (diagnosis: the device's own clock per window)

        def mark(what, w):
            if tl is not None:
                tl.append((round((time.perf_counter() - tl0) * 1e3, 2), what, w))

        def queue(p):
            p.route_begin(nsteps, qts, True)
            p.route_advance(nsteps)

        def after_window(i, w):
            """behind day w's set-up on plan i the forcing of the plan's next day (w + 2), behind its last launch its products to
            the host.  In this order: copies and the events behind them share ONE in-order hardware queue (the low-priority
            one), and the forcing queued behind a day's products would wait for their copy to end -- with the decimated result
            among them (14 ms) it then reached the device 5 ms after the day it is for should have begun."""
            if w + 2 < total:
                plans[i].stage_forcing(nsteps, days[(w + 2) % nd])
                mark("stage_forcing", w + 2)
            plans[i].fetch_begin(rs[i], True, self.output_stride)
            mark("fetch_begin", w)
        plans[0].upload_forcing(nsteps, days[0], state0)        # the first day the ordinary way (synchronous)
        if total > 1:
            plans[1].stage_forcing(nsteps, days[1 % nd])
        ms_main, ends, got = [], [], (None, None, None)

        def deliver(d):
            nonlocal got
            got = plans[d % 2].fetch_wait()                     # day d's products are on the host
            mark("fetch_wait", d)
            if on_day is not None:
                on_day(d, *got)
        # Products are handed over a day after their window -- or TWO days after it when every row's decimated series is
        # among them: that block (0.8 GB of a CONUS day) takes most of the next window to cross PCIe, and a host that waited
        # for it before queueing the day after would let the device run out of queued work once a day.  (A plan has one fetch
        # in flight: day w - 2's is waited for right before day w's begins, on the same plan.)
        lag = 2 if self.output_stride else 1
        t0 = time.perf_counter() if warmup == 0 else None       # (no warm-up day: the clock starts with day 0's window)
        queue(plans[0])
        after_window(0, 0)
        for w in range(1, total + 1):
            cur, prev = plans[w % 2], plans[(w - 1) % 2]
            if w < total:
                cur.chain_from(prev)                            # day w starts where day w - 1 ends: handed over in HBM
                mark("chained", w)
                queue(cur)
                mark("queued", w)
                if lag == 2 and w >= 2:
                    deliver(w - 2)
                after_window(w % 2, w)
            st = prev.route_end()                               # day w - 1 is through
            mark("route_end", w - 1)
            ends.append(time.perf_counter())
            if w - 1 >= warmup:
                ms_main.append(st["ms_main"])
            if lag == 1:
                deliver(w - 1)
            if w == warmup and warmup > 0:                      # the clock starts when the last warm-up day is through
                t0 = time.perf_counter()
        if lag == 2:
            for d in range(max(0, total - 2), total):
                deliver(d)
        X.device_synchronize(dev)
        el = time.perf_counter() - t0
        day_ms = [round((b - a) * 1e3, 2) for a, b in zip(ends[:-1], ends[1:])][max(0, warmup - 1):]
        if rings is not None:       # the last days on the device's clock (ms from the first of them): tiles begin / end, tail begins / ends
            first = max(0, total - 8)
            rows = [rings[d % 2][(d // 2) % 16].astype(np.int64) for d in range(first, total)]
            base = int(min(r[0] if r[0] else r[2] for r in rows))
            self.device_days = [(first + i, *[round((int(v) - base) / 1e5, 2) if v else None for v in r]) for i, r in enumerate(rows)]
            for p in plans:
                p.set_stamps(0)
        return {"el": el, "ms_main": ms_main, "hyd": got[0], "final": got[1], "fvd": got[2] if len(got) > 2 else None,
                "days_routed": total, "last_plan": plans[(total - 1) % 2], "day_ms": day_ms}

    # ---- one rank of a multi-GPU job: the merged plan, day after day ----------------------------------------------------------
    def _run_rank(self, days, state0, steps, warmup, on_day):
        r = self.r
        X, dev, comm = r._X, r._dev, r._comm
        nsteps, qts = self.nsteps, self.qts
        nd, total = len(days), warmup + steps
        r.begin_sequence(nsteps, days[0], state0, qts, self.nchunks)   # day 0's forcing and state, synchronously
        ms_main, ends, got = [], [], (None, None)

        def sync():
            X.device_synchronize(dev)
            comm.barrier()
        sync()
        t0 = time.perf_counter() if warmup == 0 else None
        for w in range(total):
            # day w: every hand-off in HBM (route_on_device's body); day w + 1's forcing is staged as soon as day w is queued
            # to its end -- it travels beside the window, and day w + 1 continues from the state day w leaves
            nxt = days[(w + 1) % nd] if w + 1 < total else None
            rows, hyd = r.route_staged(qts, self.nchunks, next_qlat=nxt)
            ends.append(time.perf_counter())
            if w >= warmup:
                ms_main.append(r.last_stats["phase0"]["ms_main"])
            if w >= 1:
                got = r.fetch_wait()                                # day w - 1's products (copied beside day w)
                if on_day is not None:
                    on_day(w - 1, *got)
            r.fetch_begin(hyd, want_hyd=(r.rank == 0 or self._hyd_everywhere), output_stride=self.output_stride)
            if w + 1 == warmup:                                     # the clock starts when the last warm-up day is through
                sync()
                t0 = time.perf_counter()
        got = r.fetch_wait()
        if on_day is not None:
            on_day(total - 1, *got)
        sync()
        el = time.perf_counter() - t0
        el = float(comm.all_reduce_max_host(np.array([el], dtype=np.float64))[0])
        day_ms = [round((b - a) * 1e3, 2) for a, b in zip(ends[:-1], ends[1:])][max(0, warmup - 1):]
        return {"el": el, "ms_main": ms_main, "hyd": got[0], "final": got[1], "fvd": got[2] if len(got) > 2 else None,
                "days_routed": total, "last_plan": r._state_plans[0], "day_ms": day_ms}
