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


class RouteStream:
    """The product's run-set loop (the reference: ``for run in run_sets: nwm_route(...) -> compute_nhd_routing_v02(...); new_q0;
    assemble_forcings(next)``, nwm_routing/__main__.py:195-333) as an iterator over days::

        router = ShardedRouter(to, params, ..., stream=True)
        with RouteStream(router, nsteps, qts_subdivisions) as rs:
            for day, hydrographs, final_state in rs.route(forcings, state0):
                ...

    ``forcings``: any iterable of [nseg, nq] arrays (global rows; page-locked ones -- ``pinned_like`` -- are copied from where
    they are, others through a page-locked ring); it is consumed as far ahead as the device can take days and may yield late.
    Every day's products come back in order: the outlet hydrographs [noutlets, nsteps] (rows ``outlet_rows``; on rank 0 of a
    multi-rank job, or everywhere with ``hydrographs_on_every_rank``), the final state ([nseg, 3]; a rank of a multi-rank job:
    of its ``rows``), with ``output_stride=n`` also every n-th step of (q, v, d) of those rows.  The arrays belong to a ring:
    copy what is kept beyond the next few days.

    Underneath is ONE stream of tile launches (include/trmc.h, trmc_stream_*): the tile index runs on over the days, every
    launch routes every row through the next K steps of its own place in the stream -- a row deep in the network works on an
    earlier tile than the headwaters, possibly of an earlier day -- so a day costs nsteps / K launches and nothing else, and a
    day's products are complete ``lag`` launches after its first row finished it.  ``latency="low"`` flushes after every day
    (each day's products right after its push, at the price of the partly filled launches a single window has).

    Several ranks: every rank streams its sub-basins and small networks; the trunk of the dominant basin rides in its owner's
    stream a day or two behind, and the cut-edge hydrographs are exchanged ONCE PER DAY (an all-gather on the exchange
    stream, ``comm``) when every rank's cut rows are through that day -- SURVEY 8e's "upstream piece finishes, sends the
    tailwater hydrograph", with the order DAG resolved by the trunk's lag instead of by waiting.
    """

    def __init__(self, router, nsteps, qts_subdivisions, output_stride=None, full_output=False, slots=0, latency="throughput",
                 hydrographs_on_every_rank=False, comm=None, exchange=None):
        """comm / exchange (several ranks): the communicator -- default: the one given to ``router.enable_device_exchange`` -- and
        how the daily cut-edge hydrographs travel: "device" (gather kernel -> all-gather on the exchange stream -> boundary rows,
        everything in HBM: RCCL over xGMI) or "host" (through ``comm.all_gather_rows_host``: any communicator with that method,
        ``all_reduce_max_host`` and ``barrier`` -- a transport without device collectives; the days of slack the trunk's lag gives
        cover the round trip).  Default: "device" when the router has a device exchange, else "host"."""
        if latency not in ("throughput", "low"):
            raise ValueError("latency must be 'throughput' or 'low'")
        self.r = router
        self.comm = comm if comm is not None else getattr(router, "_comm", None)
        self.exchange = exchange or ("device" if getattr(router, "_X", None) is not None else "host")
        if self.exchange not in ("device", "host"):
            raise ValueError("exchange must be 'device' or 'host'")
        if router.world > 1 and self.comm is None:
            raise ValueError("several ranks need a communicator (router.enable_device_exchange(comm), or comm=...)")
        self.nsteps, self.qts = int(nsteps), int(qts_subdivisions)
        self.output_stride = int(output_stride) if output_stride else 0
        self.full_output = bool(full_output)
        self.slots, self.latency = int(slots), latency
        self.world, self.rank = router.world, router.rank
        self._hyd_everywhere = bool(hydrographs_on_every_rank)
        self.plan = None
        self.info = None
        self.last_info = None
        self._dc = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.plan is not None and self.info is not None:
            try:
                self.plan.stream_end()
            except Exception:
                pass
        self.info = None

    # ------------------------------------------------------------------------------------------------------------------
    def _setup(self):
        """the plan, the trunk's lag (every rank must agree on it: the largest lag among ALL ranks' cut rows decides how many
        days after a day its cut-edge hydrographs can be exchanged), the row sets"""
        r = self.r
        K = None
        if self.world == 1:
            P = r.stream_plan(0)
            self._dc = 0
        else:
            comm = self.comm
            # (the cut rows are rows of plan0 -- the sub-basins -- whose order does not depend on the trunk: their lag is known
            # before the merged plan is built, so that plan is built once, with the lag the trunk needs)
            lag0, _, _ = r.plan0.lags()
            mine = int(lag0[r.my_cut_local].max()) if r.my_cut_local.size else 0
            pmax = int(comm.all_reduce_max_host(np.array([mine], dtype=np.float64))[0])
            tpd = self.nsteps // self._tile_steps(r.plan0)
            dc = -(-(pmax + 1) // tpd)                      # days after which every rank's cut rows are through a day
            P = r.stream_plan((dc + 1) * tpd)               # ... and the tiles the trunk must run behind for that
            self._dc = dc
        self.plan = P
        self.rows = r._rowsS
        return P

    def _tile_steps(self, P):
        return int(getattr(P, "tile_steps", 16))

    @property
    def outlet_rows(self):
        """global rows of the hydrographs handed out (ascending)"""
        return self._out_rows

    def prepare_days(self, days):
        """This rank's rows of every day in page-locked memory (where a rank of a real job reads them from the forcing files to);
        pass ``prepared=True`` to ``route`` with such arrays.  One GPU: page-locked copies of the arrays."""
        if self.plan is None:
            self._setup()
        rows = self.rows
        local = rows.shape[0] != self.r.nseg or self.world > 1
        return [pinned_like(np.ascontiguousarray(d[rows] if local else d, dtype=self.plan.dtype)) for d in days]

    def run(self, days, state0, steps, warmup=0, on_day=None, prepared=False):
        """Measurement form (bench.py, tools/sim_ranks.py): route ``warmup + steps`` consecutive days -- day w takes
        ``days[w % len(days)]`` -- and time the ``steps`` days pushed after the warm-up ones: the device is drained (and the ranks
        meet at a barrier) before the first of them is pushed and after the last of them has been, so the clock covers exactly
        ``steps`` days of launches, each carrying every row (the stream is full: ``warmup`` is raised to the days the rows' lag
        spans), the forcing of those days on its way in and the products of as many earlier days on their way out.  What brings
        the last days to their end afterwards (the flush) is outside the clock, as the work of the warm-up days' last rows is
        inside it.  Returns {"el": seconds, "days_routed", "warmup", "hyd", "final", "fvd": the last day's products, "info"}."""
        from . import comm as X
        if self.plan is None:
            self._setup()
        days = list(days) if prepared else self.prepare_days(days)
        P, multi = self.plan, self.world > 1
        lag, _, _ = P.lags()
        tpd = self.nsteps // self._tile_steps(P)
        fill = -(-int(lag.max(initial=0)) // tpd) + 1 + (self._dc + 1 if multi else 0) + (1 if (self.output_stride or self.full_output) else 0)
        if multi:
            fill = int(self.comm.all_reduce_max_host(np.array([fill], dtype=np.float64))[0])
        wu = max(int(warmup), fill)
        total, nd = wu + int(steps), len(days)
        dev = P.info()["device"]
        marks = {}
        comm = self.comm

        def meet():
            if dev >= 0:
                X.device_synchronize(dev)
            if multi:
                self.comm.barrier()

        def feed():
            for w in range(total):
                if w == wu:
                    meet()
                    marks["t0"] = time.perf_counter()
                yield days[w % nd]
            meet()
            marks["t1"] = time.perf_counter()
        last, push_ms = None, []
        for item in self.route(feed(), state0, prepared=True):
            if on_day is not None:
                on_day(*item)
            if item[0] >= wu and self.info is not None:
                try:                                        # (device time of the day's own launches on the slices' stream)
                    push_ms.append(P.stream_day_ms(item[0]))
                except (RuntimeError, ValueError):
                    pass
            last = item
        el = marks["t1"] - marks["t0"]
        if multi:
            el = float(self.comm.all_reduce_max_host(np.array([el], dtype=np.float64))[0])
        return {"el": el, "days_routed": total, "warmup": wu, "hyd": last[1], "final": last[2], "fvd": last[3] if len(last) > 3 else None,
                "info": self.last_info, "push_ms": push_ms}

    def route(self, forcings, state0=None, prepared=False):
        """generator of (day, hydrographs, final_state[, fvd]) -- see the class.  prepared: the arrays hold this rank's rows only
        (``prepare_days``)."""
        r, nsteps, qts = self.r, self.nsteps, self.qts
        P = self.plan if self.plan is not None else self._setup()
        world, multi = self.world, self.world > 1
        rows = self.rows
        local = multi or rows.shape[0] != r.nseg
        it = iter(forcings)
        try:
            first = next(it)
        except StopIteration:
            return
        nq = first.shape[1]
        dtype = P.dtype
        nrows = rows.shape[0]

        def take(q):
            return q[rows] if (local and not prepared) else q
        # the state and the shape of the forcing
        q0 = None if state0 is None else np.ascontiguousarray(state0[rows] if (local and state0.shape[0] == r.nseg) else state0, dtype=dtype)
        P.upload_forcing(nsteps, np.ascontiguousarray(take(first), dtype=dtype), q0)
        slots = self.slots
        want_fvd = bool(self.output_stride or self.full_output)
        lmax_mine = int(P.lags()[0].max(initial=0))
        tpd0 = nsteps // self._tile_steps(P)
        if want_fvd and not multi:
            # a day's (q, v, d) block takes most of a day to cross PCIe (0.79 GB of a CONUS day with hourly output: 14.5 ms):
            # it is handed over a day later than the other products would be, so that the host never waits for a copy while the
            # device runs out of queued days -- and the ring holds a slot more for it
            slots = max(slots, 3 + -(-(lmax_mine + 1) // tpd0))
        if multi:
            # every rank hands a day over in the same iteration -- when the rank whose rows run furthest behind (the trunk's owner)
            # has it: a rank's ring must hold its days that long
            lmax_all = int(self.comm.all_reduce_max_host(np.array([lmax_mine], dtype=np.float64))[0])
            slots = max(slots, (-(-lmax_all // tpd0) if lmax_all else 0) + 3 + (1 if want_fvd else 0))
        P.stream_begin(nsteps, qts, slots=slots, full_output=self.full_output and not self.output_stride,
                       output_stride=self.output_stride)
        self.info = info = P.stream_info()
        D, tpd, lmax = info["slots"], info["tiles_per_day"], info["lag_max"]
        keep = nsteps // self.output_stride if self.output_stride else nsteps
        nout = r._outS_global.shape[0]
        hyds = [_lib.result_empty((max(nout, 1), nsteps), dtype, always_pinned=True) for _ in range(D)]
        fins = [_lib.result_empty((nrows, 3), dtype, always_pinned=True) for _ in range(D)]
        fvds = [_lib.result_empty((nrows, keep, 3), dtype, always_pinned=True) if want_fvd else None for _ in range(D)]
        stage = [_lib.result_empty((nrows, nq), dtype, always_pinned=True) for _ in range(3)]
        # when a day's products are waited for: a day after they were queued (the host then never waits for launches it has
        # just queued -- the device always holds a day of work); latency="low": right after the day's own push (flushed)
        low = self.latency == "low"
        behind = 0 if low else (-(-lmax // tpd) if lmax else 0) + 1 + (1 if want_fvd else 0)
        if multi:
            comm = self.comm
            on_device = self.exchange == "device"
            ncut = int(r.cut_rows.shape[0])
            if on_device:
                X, dev = r._X, r._dev
                e = np.dtype(dtype).itemsize
                mc = max(r._max_cut, 1)
                send = [X.DeviceBuffer(dev, mc * nsteps * e) for _ in range(2)]
                recv = [X.DeviceBuffer(dev, world * mc * nsteps * e) for _ in range(2)]
                sc = r._sc
            behind = (-(-lmax_all // tpd) if lmax_all else 0) + 1 + (1 if want_fvd else 0)   # (every rank hands a day over in the same iteration)
            order = None
        self._out_rows = np.sort(r._outS_global) if not multi else None
        sel_own = np.argsort(r._outS_global, kind="stable")

        def exchange(day):
            """the cut-edge hydrographs of `day` (through on every rank): gathered, all-gathered, into the trunk's boundary rows"""
            if not multi or ncut == 0 or day < 0:
                return
            if not on_device:                           # through the host: any communicator with all_gather_rows_host
                parts = comm.all_gather_rows_host(P.stream_gather_host(day, r._rsS_cut))
                if r.plan1 is not None:
                    cut_q = np.zeros((ncut, nsteps), dtype=dtype)
                    for k, blk in enumerate(parts):
                        cut_q[r.cut_owner == k] = blk
                    P.stream_boundary_host(day, cut_q[r.b_cut_index])
                return
            k = day % 2
            P.stream_gather(day, r._rsS_cut, send[k].ptr, stream=sc)
            comm.all_gather(send[k].ptr, recv[k].ptr, mc * nsteps * e, sc)
            if r.plan1 is not None:
                P.stream_boundary(day, recv[k].ptr, nsteps, index_ptr=r._d_b_index.ptr, stream=sc)

        def deliver(day):
            nonlocal order
            P.stream_wait(day)
            k = day % D
            hyd = hyds[k][:nout]
            if multi:
                parts = comm.all_gather_rows_host(np.ascontiguousarray(hyd))
                if order is None:
                    rows_all = np.concatenate(comm.all_gather_rows_host(np.ascontiguousarray(r._outS_global.astype(np.int64)[:, None])))[:, 0]
                    order = np.argsort(rows_all, kind="stable")
                    self._out_rows = rows_all[order]
                hyd = np.concatenate(parts, 0)[order] if (self.rank == 0 or self._hyd_everywhere) else None
                fin = [fins[k]]
                fvd = [fvds[k]] if want_fvd else None
            else:
                hyd = hyd[sel_own]
                fin, fvd = fins[k], fvds[k]
            return (day, hyd, fin, fvd) if want_fvd else (day, hyd, fin)

        d, nxt, pending = 0, first, 0
        delivered = 0
        while nxt is not None:
            q = take(nxt)
            if not (_lib.is_pinned(q) and q.dtype == dtype and q.flags.c_contiguous):
                buf = stage[d % 3]
                buf[...] = q
                q = buf
            P.stream_push(q, rowset=r._rsS_out, hyd=hyds[d % D], q0=fins[d % D], fvd=fvds[d % D])
            if low:
                if multi:
                    raise ValueError("latency='low' is a one-GPU option")
                P.stream_flush()
            exchange(d - self._dc)
            while delivered <= d - behind:
                yield deliver(delivered)
                delivered += 1
            d += 1
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
        total = d
        # the days still under way: no new day starts, the stream advances a day's launches at a time (the exchange of the last
        # days' cut-edge hydrographs between them), until every day has been queued to its end
        if multi:
            if low:
                raise ValueError("latency='low' is a one-GPU option")
            for k in range(self._dc):
                P.stream_advance(tpd)
                exchange(total + k - self._dc)
        P.stream_flush()
        while delivered < total:
            yield deliver(delivered)
            delivered += 1
        self.last_info = P.stream_info()
        P.stream_end()
        self.info = None
