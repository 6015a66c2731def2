"""One-process-per-GPU routing of a partitioned network (see sharding.py).

Phase 0: every rank routes the sub-basins / small networks it owns (one plan).
Exchange: outlet hydrographs of the cut sub-basins are all-gathered (troute_amd.comm.Comm:
RCCL over xGMI bound through the C ABI, or its shared-memory transport) -- the data the
reference hands from one sub-network order to the next as
``flowveldepth_interorder`` (compute.py:882-897, consumed mc_reach.pyx:458-469).
Phase 1: the rank that owns a trunk routes it with those hydrographs as
prescribed boundary rows.  Finally the network-outlet hydrographs are gathered.

The compute backend is injected (``plan_factory``): the product passes
``troute_amd.plan.RoutingPlan`` (HIP); the CPU test-suite passes an
oracle-backed stand-in with the same interface so the partition / exchange /
gather logic is exercised with world_size 2 on gloo without a GPU.
"""
import os as _os

import numpy as np

from . import sharding


def restrict_csr(up_ptr, up_idx, rows, g2l):
    """Upstream CSR of the sub-table `rows` (ascending global rows) in local indices."""
    cnt = up_ptr[rows + 1] - up_ptr[rows]
    lp = np.zeros(rows.shape[0] + 1, dtype=np.int64)
    lp[1:] = np.cumsum(cnt)
    if lp[-1] == 0:
        return lp, np.zeros(0, dtype=np.int64)
    rep = np.repeat(up_ptr[rows] - lp[:-1], cnt)
    gi = up_idx[np.arange(lp[-1]) + rep]
    li = g2l[gi]
    if (li < 0).any():
        raise ValueError("sub-table is not closed under upstream links")
    return lp, li


class _EngineOf:
    """stands in for a plan that does not exist yet where only its engine is asked for"""
    def __init__(self, engine):
        self.engine = engine


class ShardedRouter:
    def __init__(self, to, params, rank=0, world=1, device=0, plan_factory=None, precision=32,
                 partition=None, cost_hint=None, assume_short_ts=None, engine="auto", options=None, stream=False):
        """cost_hint: optional uint8 [nseg] (global rows), the ``iteration_hint()`` of a router of the same network
        after a window -- every plan then groups its rows by that cost (RoutingPlan ``cost_hint``; results unchanged,
        the kernels' wavefronts become uniform in cost).  assume_short_ts / engine: passed to every RoutingPlan (the
        timestep mode the router will be used with, if known; "auto" | "levels" | "flow"); options: RoutingPlan's (a dict of
        trmc_plan_options fields, e.g. {"arithmetic": "tolerance"})."""
        # stream=True: the router's windows follow each other as a STREAM (troute_amd.sequence.RouteStream, trmc_stream_*): its
        # plans are short-timestep plans of the level engine in cluster order, with as many levels in slices as are worth a
        # slice (in a stream they cost no launches)
        self._stream = bool(stream)
        if stream:
            options = {"cluster_rows": 128, "wide_min_rows": 1024, "wide_levels": 32, **(options or {})}
            assume_short_ts, engine = True, "levels"
        if plan_factory is None:
            from .plan import RoutingPlan  # the HIP engine; no fallback
            from . import _lib
            # one hardware queue per stream priority for this process's HIP runtime, if it is not up yet (why: DESIGN.md
            # section 7b; _lib.single_hw_queue_per_priority).  Done here, when a router is built -- importing this module
            # changes nothing for other HIP users of the process.
            _lib.single_hw_queue_per_priority("troute_amd.distributed.ShardedRouter")

            def plan_factory(lp, li, par, boundary, prec, dev, short=None, extra_options=None, **kw):
                # short: the merged plan of the short-timestep device path is built for that mode whatever the caller said
                opt = options if not extra_options else {**(options or {}), **extra_options}
                return RoutingPlan(lp, li, par, boundary, prec, dev,
                                   assume_short_ts=assume_short_ts if short is None else short, engine=engine, options=opt, **kw)
        from .synthetic import upstream_csr
        self._hint = None if cost_hint is None else np.ascontiguousarray(cost_hint, dtype=np.uint8)
        if self._hint is not None:
            if self._hint.shape != (to.shape[0],):
                raise ValueError("cost_hint shape mismatch")
            base_factory = plan_factory

            def plan_factory(lp, li, par, boundary, prec, dev, rows=None, **kw):   # noqa: F811 - the hinted factory
                return base_factory(lp, li, par, boundary, prec, dev, cost_hint=self._hint[rows], **kw)
        else:
            base_factory = plan_factory

            def plan_factory(lp, li, par, boundary, prec, dev, rows=None, **kw):   # noqa: F811
                return base_factory(lp, li, par, boundary, prec, dev, **kw)
        self.rank, self.world = rank, world
        self._engine_arg, self._short_arg = engine, assume_short_ts
        self.nseg = to.shape[0]
        self.dtype = np.float32 if precision == 32 else np.float64
        part = sharding.partition(to, world) if partition is None else partition
        self.part = part
        piece, phase, owner = part["piece"], part["phase"], part["owner"]
        up_ptr, up_idx = upstream_csr(to)
        row_phase = phase[piece]
        row_owner = owner[piece]

        # ---- phase 0: my sub-basins and small networks, one plan -----------------------------
        self.rows0 = np.flatnonzero((row_phase == 0) & (row_owner == rank))
        g2l = np.full(self.nseg, -1, dtype=np.int64)
        g2l[self.rows0] = np.arange(self.rows0.shape[0])
        lp, li = restrict_csr(up_ptr, up_idx, self.rows0, g2l)
        self.plan0 = plan_factory(lp, li, params[self.rows0], None, precision, device, rows=self.rows0)
        self._mk = {"factory": plan_factory, "precision": precision, "device": device, "params": params,
                    "csr0": (lp, li)}
        self.planM = None     # sub-basins + time-skewed trunk in one plan (short-timestep device path)
        self._collect = False
        # cut rows: every rank knows the global list (ascending); mine are a subset
        self.cut_rows = part["cut_rows"]
        self.cut_owner = row_owner[self.cut_rows] if self.cut_rows.size else np.zeros(0, np.int32)
        self.my_cut_local = g2l[self.cut_rows[self.cut_owner == rank]] if self.cut_rows.size else np.zeros(0, np.int64)
        outlets = np.flatnonzero(to < 0)
        self.outlets = outlets
        o0 = outlets[(row_phase[outlets] == 0) & (row_owner[outlets] == rank)]
        self.my_out0_global, self.my_out0_local = o0, g2l[o0]

        # ---- phase 1: trunks I own, cut rows as boundary rows ----------------------------------
        trunk_rows = np.flatnonzero((row_phase == 1) & (row_owner == rank))
        self.plan1 = None
        self.rows1 = np.zeros(0, dtype=np.int64)
        self.my_out1_global = np.zeros(0, dtype=np.int64)
        if trunk_rows.size:
            feeds_mine = np.isin(part["cut_into"], trunk_rows)
            b_rows = self.cut_rows[feeds_mine]
            rows1 = np.union1d(trunk_rows, b_rows)
            g2l1 = np.full(self.nseg, -1, dtype=np.int64)
            g2l1[rows1] = np.arange(rows1.shape[0])
            boundary = np.isin(rows1, b_rows)
            # boundary rows keep an empty upstream list inside the trunk table
            up_ptr1 = up_ptr.copy()
            cnt = up_ptr[rows1 + 1] - up_ptr[rows1]
            cnt[boundary] = 0
            lp1 = np.zeros(rows1.shape[0] + 1, dtype=np.int64)
            lp1[1:] = np.cumsum(cnt)
            rep = np.repeat(up_ptr[rows1] - lp1[:-1], cnt)
            li1 = g2l1[up_idx[np.arange(lp1[-1]) + rep]] if lp1[-1] else np.zeros(0, np.int64)
            if (li1 < 0).any():
                raise ValueError("trunk table is not closed under upstream links")
            del up_ptr1
            self.rows1 = rows1
            self.boundary1 = boundary
            # position of each boundary row (ascending local row) in the global cut list
            self.b_cut_index = np.searchsorted(self.cut_rows, rows1[boundary])
            self.plan1 = plan_factory(lp1, li1, params[rows1], boundary.astype(np.uint8), precision, device, rows=rows1)
            self._mk["csr1"] = (lp1, li1)
            o1 = outlets[np.isin(outlets, trunk_rows)]
            self.my_out1_global, self.my_out1_local = o1, g2l1[o1]

    def collect_cost(self, enable=True):
        """Have the plans sum every row's iteration class over the windows routed from now on (a finer hint than the
        last step alone: a row that is dry for the first hours of a window and wet afterwards sorts between the
        always-dry and the always-wet ones)."""
        self._collect = bool(enable)
        for plan in (self.plan0, self.plan1, self.planM):
            if plan is not None and hasattr(plan, "collect_cost"):
                plan.collect_cost(enable)

    def iteration_hint(self):
        """uint8 [nseg]: the secant iterations (clamped at 3) this rank's rows needed on the last step of the last
        window, 0 for rows of other ranks -- what ``ShardedRouter(..., cost_hint=...)`` takes.  Iteration classes
        persist (a row repeats its count from one step to the next 99.3 % of the time), so the hint of one window
        orders the next."""
        hint = np.zeros(self.nseg, dtype=np.uint8)

        def take(plan, rows):
            try:
                if self._collect:
                    # the whole window: sum of min(iterations, 3) over its steps, in sixteenths of a step-mean
                    cost, n = plan.download_cost()
                    it = np.minimum((cost.astype(np.int64) * 16 + n - 1) // max(n, 1), 255)
                else:
                    it = np.minimum(plan.download_iterations(), 3)
            except RuntimeError:          # this plan has not routed a window yet
                return False
            hint[rows] = np.maximum(hint[rows], it.astype(np.uint8))
            return True
        if self.planM is not None:
            take(self.planM, self._rowsM)
        if getattr(self, "_planS", None) is not None and self._planS is not self.plan0:
            take(self._planS, self._rowsS)
        take(self.plan0, self.rows0)
        if self.plan1 is not None:
            take(self.plan1, self.rows1)
        return hint

    # ---- device-resident exchange (HIP buffers, streams and events through the C ABI; RCCL = troute_amd.comm.Comm) ----
    def enable_device_exchange(self, comm, device=None):
        """Precompute the index maps for route_on_device(): every rank knows the whole partition, so the
        position of every cut row / outlet row inside the all-gathered buffers is known up front.
        comm: a troute_amd.comm.Comm of this job's ranks (its all_gather is ncclAllGather over xGMI, or the
        shared-memory transport when ranks share a device)."""
        from . import comm as X
        self._X, self._comm = X, comm
        self._dev = self._mk["device"] if device is None else int(device)
        world = self.world
        part = self.part
        piece, phase, owner = part["piece"], part["phase"], part["owner"]
        row_phase, row_owner = phase[piece], owner[piece]
        self._esz = np.dtype(self.dtype).itemsize
        # cut rows: slot (owner, index within the owner's ascending list)
        ncut = self.cut_rows.shape[0]
        self._max_cut = 0
        self._d_b_index = None
        if ncut:
            counts = np.bincount(self.cut_owner, minlength=world)
            self._max_cut = int(counts.max())
            idx_in_owner = np.zeros(ncut, dtype=np.int64)
            for r in range(world):
                m = self.cut_owner == r
                idx_in_owner[m] = np.arange(int(m.sum()))
            flat = self.cut_owner.astype(np.int64) * self._max_cut + idx_in_owner
            if self.plan1 is not None:
                self._n_b_index = int(self.b_cut_index.shape[0])
                self._d_b_index = X.DeviceBuffer.from_array(self._dev, flat[self.b_cut_index].astype(np.int64))
        # outlets: per rank, phase-0 outlets (ascending) followed by trunk outlets (ascending)
        outlets = self.outlets
        per_rank = []
        for r in range(world):
            o0 = outlets[(row_phase[outlets] == 0) & (row_owner[outlets] == r)]
            o1 = outlets[(row_phase[outlets] == 1) & (row_owner[outlets] == r)]
            per_rank.append(np.concatenate([o0, o1]))
        self._max_out = max(1, max(len(x) for x in per_rank))
        rows = np.concatenate(per_rank)
        slot = np.concatenate([r * self._max_out + np.arange(len(x)) for r, x in enumerate(per_rank)])
        order = np.argsort(rows, kind="stable")
        self._out_rows = rows[order]
        self._d_out_index = X.DeviceBuffer.from_array(self._dev, slot[order].astype(np.int64))
        # exchange stream: the collectives are ordered on it.  With one hardware queue per stream priority (set at import) it
        # shares a queue with the plan stream of its priority, in order of submission: on the LEVEL engine the ordinary queue
        # holds the wide tiles, all queued at the window's first call -- a chunk's exchange queued behind them ran after the
        # last tile and held the trunk's owner back until then (N = 2: its tail started at 5.7 ms of 12.2) -- so there the
        # exchange goes to the low-priority queue, beside the result transpose; the dataflow engine has nothing in the ordinary
        # queue and keeps it there.
        levels = getattr(self.plan0, "engine", "levels") == "levels"
        prio = _os.environ.get("TRMC_EXCHANGE_PRIORITY")
        self._sc = X.stream_create(self._dev, int(prio) if prio not in (None, "") else (-1 if levels else None))
        self._events = []

    def _event(self):
        e = self._X.event_create(self._dev)
        self._events.append(e)
        return e

    def upload_trunk(self):
        """Stage the trunk's forcing once (its boundary hydrographs arrive per route via the exchange)."""
        if self.plan1 is not None:
            self.plan1.upload_forcing(self.nsteps, self._qlat[self.rows1], self._q0_of(self.rows1), None)

    def _window_buffers(self, widths, nsteps):
        """Send / receive blocks of every time chunk, the outlet blocks and one event per hand-off, made once per window
        shape (zero-filled device memory from the library)."""
        key = (tuple(int(w) for w in widths), int(nsteps))
        if getattr(self, "_xbuf_key", None) == key:
            return self._xbuf
        X, dev, world, e = self._X, self._dev, self.world, self._esz
        x = {"send": [], "recv": [], "ev_sent": [], "ev_filled": []}
        for w in key[0]:
            if self._max_cut:
                x["send"].append(X.DeviceBuffer(dev, self._max_cut * w * e))
                x["recv"].append(X.DeviceBuffer(dev, world * self._max_cut * w * e))
                x["ev_sent"].append(self._event())
                x["ev_filled"].append(self._event())
        x["send_o"] = X.DeviceBuffer(dev, self._max_out * nsteps * e)
        x["recv_o"] = X.DeviceBuffer(dev, world * self._max_out * nsteps * e)
        # (two: the block of window k may still be on its way to the host while window k + 1 fills the other one)
        x["hyd"] = [X.DeviceBuffer(dev, max(1, self._out_rows.shape[0]) * nsteps * e) for _ in range(2)]
        x["flip"] = 0
        x["ev_out"] = self._event()
        x["ev_out1"] = self._event()
        self._xbuf_key, self._xbuf = key, x
        return x

    def route_on_device(self, qts_subdivisions, assume_short_ts, nchunks=None):
        """One routing window with every hand-off in HBM; returns (outlet_rows, hydrographs as a comm.DeviceArray).
        The collectives go through the communicator given to enable_device_exchange().

        assume_short_ts: the trunk rides in the launches of this rank's sub-basins, `2 * chunk` steps behind
        them (`_route_skewed`); otherwise sub-basins first, trunk after the exchange (`_route_phased`)."""
        if assume_short_ts:
            return self._route_skewed(qts_subdivisions, nchunks)
        if not getattr(self, "_plan0_staged", True):
            raise RuntimeError("this router continued from the state of its merged (short-timestep) plan; upload() the "
                               "forcing with an explicit q0 before routing in the general mode")
        return self._route_phased(qts_subdivisions, assume_short_ts, nchunks)

    # ---- short-timestep path: one plan, trunk time-skewed -----------------------------------------------
    def _merged_plan(self, lag, upload=True):
        """plan0's table followed by the trunk table (trunk rows + boundary copies of the cut rows that feed
        it); trunk rows carry `lag`.  Built once per lag, forcing staged once per upload() (upload=False: the caller stages
        the forcing itself -- a sequence of days, troute_amd.sequence)."""
        if self.plan1 is None:                      # nothing to merge: plan0 itself, forcing staged by upload()
            if not hasattr(self, "_rsM_cut"):
                self._rsM_cut = self.plan0.rowset(self.my_cut_local)
                self._rsM_out0 = self.plan0.rowset(self.my_out0_local)
                self._rsM_out1 = None
            return self.plan0
        if self.planM is not None and self._planM_lag == lag and (not upload or self._planM_upload == self._upload_gen):
            return self.planM
        mk = self._mk
        n0 = self.rows0.shape[0]
        if self.planM is None or self._planM_lag != lag:
            if self.planM is not None:
                self.planM.close()
            lp0, li0 = mk["csr0"]
            lp1, li1 = mk["csr1"]
            up_ptr = np.concatenate([lp0, lp0[-1] + lp1[1:]])
            up_idx = np.concatenate([li0, li1 + n0])
            rows = np.concatenate([self.rows0, self.rows1])
            # (1 = boundary copy of a cut row; 2 = trunk row: routed, lagged, kept out of the leading levels of a level-engine
            # plan so that those can still run ahead of the window -- trmc.h, trmc_plan_create)
            boundary = np.concatenate([np.zeros(n0, np.uint8), np.where(self.boundary1, 1, 2).astype(np.uint8)])
            lagv = np.concatenate([np.zeros(n0, np.int32), np.where(self.boundary1, 0, lag).astype(np.int32)])
            self._rowsM = rows
            self.planM = mk["factory"](up_ptr, up_idx, mk["params"][rows], boundary, mk["precision"], mk["device"], rows=rows,
                                       short=True)
            self.planM.set_lag(lagv)
            if self._collect:
                self.planM.collect_cost(True)
            self._planM_lag = lag
            self._rsM_cut = self.planM.rowset(self.my_cut_local)
            self._rsM_out0 = self.planM.rowset(self.my_out0_local)
            self._rsM_out1 = self.planM.rowset(n0 + self.my_out1_local) if self.my_out1_global.size else None
        if upload:
            self.planM.upload_forcing(self.nsteps, self._qlat[self._rowsM], self._q0_of(self._rowsM), None)
            self._planM_upload = self._upload_gen
        return self.planM

    # ---- a STREAM of days on this rank (troute_amd.sequence.RouteStream) ---------------------------------------------------
    def stream_plan(self, late_lag=0):
        """The plan a stream of windows runs on (router made with ``stream=True``): plan0 itself where this rank owns no trunk,
        else plan0's table followed by the trunk table (trunk rows + boundary copies of the cut rows that feed it) in ONE
        cluster-ordered plan whose trunk rows run at least ``late_lag`` tiles behind the headwaters -- their inflows are
        exchanged once a day (no lag table, no chunks: every row of a stream has its own tile lag).  Also sets the row sets
        ``_rsS_cut`` (my cut rows), ``_rsS_out`` (my outlets: sub-basins', then the trunk's)."""
        if not self._stream:
            raise ValueError("ShardedRouter(..., stream=True) builds the plans a stream of windows runs on")
        if getattr(self, "_planS", None) is not None and self._planS_lag == late_lag:
            return self._planS
        self._close_stream_plan()
        mk = self._mk
        n0 = self.rows0.shape[0]
        if self.plan1 is None:
            P = self.plan0
            self._rowsS = self.rows0
            out_local = self.my_out0_local
        else:
            lp0, li0 = mk["csr0"]
            lp1, li1 = mk["csr1"]
            up_ptr = np.concatenate([lp0, lp0[-1] + lp1[1:]])
            up_idx = np.concatenate([li0, li1 + n0])
            rows = np.concatenate([self.rows0, self.rows1])
            boundary = np.concatenate([np.zeros(n0, np.uint8), np.where(self.boundary1, 1, 2).astype(np.uint8)])
            self._rowsS = rows
            P = mk["factory"](up_ptr, up_idx, mk["params"][rows], boundary, mk["precision"], mk["device"], rows=rows,
                              short=True, extra_options={"cluster_late_lag": int(late_lag)})
            out_local = np.concatenate([self.my_out0_local, n0 + self.my_out1_local]) if self.my_out1_global.size else self.my_out0_local
        self._planS, self._planS_lag = P, late_lag
        self._rsS_cut = P.rowset(self.my_cut_local)
        self._rsS_out = P.rowset(out_local)
        self._outS_global = np.concatenate([self.my_out0_global, self.my_out1_global])
        if self._collect:
            P.collect_cost(True)
        return P

    def _close_stream_plan(self):
        P = getattr(self, "_planS", None)
        if P is not None and P is not self.plan0:
            P.close()
        self._planS = None

    # ---- a sequence of days on this rank (troute_amd.sequence.DaySequence): the forcing staged by the caller, day by day ----
    def sequence_rows(self):
        """global rows of the plan a short-timestep window of this rank runs on, in its row order: the sub-basins, then (a
        trunk owner) the trunk table -- trunk rows and the boundary copies of the cut rows that feed it"""
        return self.rows0 if self.plan1 is None else np.concatenate([self.rows0, self.rows1])

    def _chunking(self, nchunks):
        """(steps per chunk, chunks, the trunk's lag) of a short-timestep window: decided ONCE per router (see _route_skewed)"""
        if nchunks is None:
            if getattr(self, "_nchunks_default", None) is None:
                self._nchunks_default = self._default_chunks(self.planM if self.planM is not None
                                                             else _EngineOf(self._merged_engine()))
            nchunks = self._nchunks_default
        K = max(1, -(-self.nsteps // max(1, int(nchunks))))
        return K, -(-self.nsteps // K), (2 * K if self.plan1 is not None else 0)

    def begin_sequence(self, nsteps, local_qlat, state0, qts_subdivisions=None, nchunks=None, local=None):
        """Day 0 of a sequence: this rank's rows of the forcing (``sequence_rows()`` order) and the state -- None to continue
        from what the rank's last window left in HBM, else [nseg, 3] of every row (``local=False``) or this rank's rows in
        ``sequence_rows()`` order (``local=True``) -- staged synchronously.  ``local=None`` tells the two apart by the row count
        and refuses the one case in which that is ambiguous (a rank whose table, boundary copies included, has exactly nseg
        rows).  (``qts_subdivisions`` is not needed here; kept for the callers that pass it.)"""
        self.nsteps = nsteps
        P = self._merged_plan(self._chunking(nchunks)[2], upload=False)
        rows = self.sequence_rows()
        if state0 is not None and local is None:
            if rows.shape[0] == self.nseg and self.world > 1:
                raise ValueError("state0 has as many rows as the network AND as this rank's table: say local=True / False")
            local = state0.shape[0] == rows.shape[0] and rows.shape[0] != self.nseg
        if state0 is None:
            q0 = None
        elif local:                                     # (already this rank's rows, in that order)
            if state0.shape[0] != rows.shape[0]:
                raise ValueError("local state0 must have one row per row of sequence_rows()")
            q0 = np.ascontiguousarray(state0)
        else:
            if state0.shape[0] != self.nseg:
                raise ValueError("global state0 must have one row per row of the network")
            q0 = np.ascontiguousarray(state0[rows])
        P.upload_forcing(nsteps, local_qlat, q0, None)
        self._plan0_staged = self.plan1 is None

    def stage_next(self, nsteps, local_qlat):
        """The NEXT day's forcing of this rank's rows (page-locked memory) on its way to the device beside whatever runs; the
        day starts from the state the last window leaves (trmc_stage_forcing)."""
        P = self.planM if self.plan1 is not None else self.plan0
        P.stage_forcing(nsteps, local_qlat)

    def route_staged(self, qts_subdivisions, nchunks=None, next_qlat=None):
        """One short-timestep window on the forcing staged by begin_sequence() / stage_next(): route_on_device()'s body.
        next_qlat: the NEXT day's forcing of this rank's rows (page-locked) -- staged as soon as this window is queued to its
        end, so that it travels beside the window instead of between two (the next window then continues from this one's
        final state: trmc_stage_forcing on a busy plan)."""
        hook = None if next_qlat is None else (lambda: self.stage_next(self.nsteps, next_qlat))
        return self._route_skewed(qts_subdivisions, nchunks, staged=True, before_end=hook)

    def _merged_engine(self):
        """The engine trmc_plan_create_ex's TRMC_ENGINE_AUTO gives the merged short-timestep plan (rows0 + trunk), without
        building it: the rule of csrc/trmc.hip (fp64 -> levels; TRMC_ENGINE overrides; fp32 meant for assume_short_ts with a
        million routed rows or more -> levels; else dataflow)."""
        if self.plan1 is None:
            return getattr(self.plan0, "engine", "levels")
        if self._mk["precision"] != 32:
            return "levels"
        env = _os.environ.get("TRMC_ENGINE")
        if env in ("levels", "flow"):
            return env
        routed = int(self.rows0.shape[0]) + int((~np.asarray(self.boundary1, dtype=bool)).sum())
        return "levels" if routed >= 1_000_000 else "flow"

    def _default_chunks(self, plan):
        """time chunks of the hand-off pipeline: a launch per chunk.  The level engine launches per timestep anyway and
        wants the trunk's skew (two chunks) short; the dataflow engine runs a chunk as one persistent launch and wants
        few of them (349 k-row ranks: 5.7 ms with 24 chunks, 4.5 with 8, 4.3 with 4 of 72 steps -- every launch ends in
        a drain; with 2 or 3 the trunk's owner, who trails by two chunks, is the slowest rank again).
        The number of chunks is the number of all-gathers of a window: EVERY rank must arrive at the same one.  It is
        therefore decided from what every rank knows -- the whole partition, hence the engine every rank's window runs on
        (the rule of trmc_plan_create_opt applied to each rank's row count) -- not from this rank's own engine: a partition
        rebalanced by measured pace can leave one rank of two under the million rows at which the engines change, and
        ranks that disagreed on the count waited for each other for ever (seen once in three bench launches at N = 2)."""
        engines = set(self._engines_of_all_ranks())
        if engines == {"flow"}:
            return 4
        if engines == {"levels"}:
            return 24
        return 8

    def _engines_of_all_ranks(self):
        """the engine of every rank's short-timestep window (sub-basins + the trunk it owns), from the partition alone"""
        part = self.part
        piece, phase, owner = part["piece"], part["phase"], part["owner"]
        routed = np.bincount(owner[piece], minlength=self.world)          # sub-basin rows and trunk rows of every rank
        owns_trunk = np.bincount(owner[phase == 1], minlength=self.world) > 0   # (its merged plan is built for assume_short_ts)
        env = _os.environ.get("TRMC_ENGINE")
        out = []
        for r in range(self.world):
            if self._mk["precision"] != 32:
                out.append("levels")
            elif self._engine_arg in ("levels", "flow"):
                out.append(self._engine_arg)
            elif env in ("levels", "flow"):
                out.append(env)
            else:
                out.append("levels" if ((self._short_arg or owns_trunk[r]) and int(routed[r]) >= 1_000_000) else "flow")
        return out

    def _route_skewed(self, qts_subdivisions, nchunks, staged=False, before_end=None):
        """assume_short_ts: a row at step t reads its upstream rows at step t-1 only.  The window is cut into
        chunks of K steps.  After this rank's sub-basins have been queued through chunk c, the chunk's
        cut-edge hydrographs are gathered (plan stream), all-gathered and written into the trunk's boundary
        rows (exchange stream).  The trunk rows sit in the SAME launches as the sub-basins, 2K steps behind:
        when a launch needs chunk c's boundary values, their exchange was queued a whole chunk of launches
        earlier, so the plan stream's wait on it never stalls, and the trunk costs no launches of its own
        except the 2K that drain it at the end.  The host never waits inside the window."""
        X, dev, comm, e = self._X, self._dev, self._comm, self._esz
        nsteps = self.nsteps
        # (the default number of chunks is decided ONCE per router, from the engine the MERGED plan runs on -- not from plan0's
        # on the first window and the merged plan's afterwards: the two can differ (plan0 under a million rows: dataflow;
        # rows0 + trunk at or above it: levels), the chunk count would change between the first and the second window, with
        # it the trunk's lag, and a merged plan rebuilt for the new lag has no resident state to continue from)
        K, C, lag = self._chunking(nchunks)
        P = self._merged_plan(lag, upload=not staged)
        x = self._window_buffers([min(nsteps, (c + 1) * K) - c * K for c in range(C)], nsteps)
        sc = self._sc
        P.route_begin(nsteps, qts_subdivisions, True)
        filled = [False] * C
        last = nsteps + lag
        c = 0
        sP = P.stream()
        while True:
            d_end = min((c + 1) * K, last)
            # the stream this chunk's launch goes to (the dataflow engine may alternate between two, so that consecutive
            # chunks overlap: trmc_plan_stream); the gather below is queued behind it
            sP = P.stream()
            if lag and c >= 2 and filled[min(c - 2, C - 1)]:
                X.stream_wait_event(dev, sP, x["ev_filled"][min(c - 2, C - 1)])   # boundary values of chunk c-2: queued a chunk ago
            P.route_advance(d_end)
            if c < C and self._max_cut:
                tb, te = c * K, min(nsteps, (c + 1) * K)
                w = te - tb
                P.gather_flow_range(self._rsM_cut, tb, te, x["send"][c].ptr, w)
                X.event_record(dev, x["ev_sent"][c], sP)
                X.stream_wait_event(dev, sc, x["ev_sent"][c])
                comm.all_gather(x["send"][c].ptr, x["recv"][c].ptr, self._max_cut * w * e, sc)
                if self.plan1 is not None:
                    # straight from the all-gathered block into the boundary rows: the fill kernel gathers by index
                    P.set_boundary_flow_range(tb, te, x["recv"][c].ptr, w, stream=sc, index_ptr=self._d_b_index.ptr)
                    X.event_record(dev, x["ev_filled"][c], sc)
                    filled[c] = True
            if d_end >= last:
                break
            c += 1
        # network outlets: phase-0 outlets then trunk outlets in this rank's slot of the final all-gather
        send_o, recv_o = x["send_o"], x["recv_o"]
        n0 = self.my_out0_local.shape[0]
        P.gather_flow_range(self._rsM_out0, 0, nsteps, send_o.ptr, nsteps)
        if self._rsM_out1 is not None:
            P.gather_flow_range(self._rsM_out1, 0, nsteps, send_o.ptr + n0 * nsteps * e, nsteps)
        X.event_record(dev, x["ev_out"], sP)
        X.stream_wait_event(dev, sc, x["ev_out"])
        comm.all_gather(send_o.ptr, recv_o.ptr, self._max_out * nsteps * e, sc)
        x["flip"] ^= 1
        hyd = x["hyd"][x["flip"]]
        X.gather_rows(dev, recv_o.ptr, self._d_out_index.ptr, self._out_rows.shape[0], nsteps * e, hyd.ptr, sc)
        if before_end is not None:        # (the window is queued to its end: what the caller wants in flight beside it)
            before_end()
        self.last_stats = {"phase0": P.route_end()}
        X.stream_synchronize(dev, sc)
        self._state_plans = [P]
        return self._out_rows, X.DeviceArray(hyd, (self._out_rows.shape[0], nsteps), self.dtype, sc)

    # ---- general path: sub-basins, exchange, trunk (optionally pipelined in time chunks) ---------------------
    def _route_phased(self, qts_subdivisions, assume_short_ts, nchunks=None):
        """Sub-basins on plan0's stream, trunk on plan1's, the exchange between them on a third stream; with
        nchunks > 1 the three are pipelined in time (the level wavefront of the general mode restarts per
        chunk, so the default keeps the window whole)."""
        X, dev, comm, e = self._X, self._dev, self._comm, self._esz
        nsteps = self.nsteps
        if nchunks is None:
            nchunks = 1
        nchunks = max(1, min(int(nchunks), nsteps))
        bounds = np.round(np.linspace(0, nsteps, nchunks + 1)).astype(np.int64)
        x = self._window_buffers(np.diff(bounds), nsteps)
        if not hasattr(self, "_rs_cut"):
            self._rs_cut = self.plan0.rowset(self.my_cut_local)
            self._rs_out0 = self.plan0.rowset(self.my_out0_local)
            self._rs_out1 = self.plan1.rowset(self.my_out1_local) if self.plan1 is not None else None
        sc = self._sc
        self.plan0.route_begin(nsteps, qts_subdivisions, assume_short_ts)
        if self.plan1 is not None:
            self.plan1.route_begin(nsteps, qts_subdivisions, assume_short_ts)
        s0 = self.plan0.stream()
        s1 = self.plan1.stream() if self.plan1 is not None else 0
        for c in range(nchunks):
            tb, te = int(bounds[c]), int(bounds[c + 1])
            w = te - tb
            self.plan0.route_advance(te)
            if self._max_cut:
                self.plan0.gather_flow_range(self._rs_cut, tb, te, x["send"][c].ptr, w)
                X.event_record(dev, x["ev_sent"][c], s0)
                X.stream_wait_event(dev, sc, x["ev_sent"][c])
                comm.all_gather(x["send"][c].ptr, x["recv"][c].ptr, self._max_cut * w * e, sc)
                if self.plan1 is not None:
                    X.event_record(dev, x["ev_filled"][c], sc)
                    X.stream_wait_event(dev, s1, x["ev_filled"][c])
                    self.plan1.set_boundary_flow_range(tb, te, x["recv"][c].ptr, w, index_ptr=self._d_b_index.ptr)
            if self.plan1 is not None:
                if not self._max_cut:
                    self.plan1.set_boundary_flow_range(tb, te, 0, w)
                self.plan1.route_advance(te)
        # network outlets: phase-0 outlets then trunk outlets in this rank's slot of the final all-gather
        send_o, recv_o = x["send_o"], x["recv_o"]
        n0 = self.my_out0_local.shape[0]
        self.plan0.gather_flow_range(self._rs_out0, 0, nsteps, send_o.ptr, nsteps)
        X.event_record(dev, x["ev_out"], s0)
        X.stream_wait_event(dev, sc, x["ev_out"])
        if self.plan1 is not None:
            if self.my_out1_global.size:
                self.plan1.gather_flow_range(self._rs_out1, 0, nsteps, send_o.ptr + n0 * nsteps * e, nsteps)
            X.event_record(dev, x["ev_out1"], s1)
            X.stream_wait_event(dev, sc, x["ev_out1"])
        comm.all_gather(send_o.ptr, recv_o.ptr, self._max_out * nsteps * e, sc)
        x["flip"] ^= 1
        hyd = x["hyd"][x["flip"]]
        X.gather_rows(dev, recv_o.ptr, self._d_out_index.ptr, self._out_rows.shape[0], nsteps * e, hyd.ptr, sc)
        stats = {"phase0": self.plan0.route_end()}
        if self.plan1 is not None:
            stats["phase1"] = self.plan1.route_end()
        X.stream_synchronize(dev, sc)
        self.last_stats = stats
        self._state_plans = [self.plan0] + ([self.plan1] if self.plan1 is not None else [])
        return self._out_rows, X.DeviceArray(hyd, (self._out_rows.shape[0], nsteps), self.dtype, sc)

    def close(self):
        self._close_stream_plan()
        if self.planM is not None:
            self.planM.close()
        self.plan0.close()
        if self.plan1 is not None:
            self.plan1.close()
        if getattr(self, "_X", None) is not None:          # exchange stream, events and buffers of route_on_device()
            for ev in getattr(self, "_events", []):
                self._X.event_destroy(self._dev, ev)
            self._events = []
            if getattr(self, "_sc", 0):
                self._X.stream_destroy(self._dev, self._sc)
                self._sc = 0
            if getattr(self, "_cs", 0):
                self._X.stream_destroy(self._dev, self._cs)
                self._cs = 0
            for b in list(getattr(self, "_xbuf", {}).values()):
                for bb in (b if isinstance(b, list) else [b]):
                    if hasattr(bb, "free"):
                        bb.free()
            self._xbuf, self._xbuf_key = {}, None
            for name in ("_d_b_index", "_d_out_index"):
                if getattr(self, name, None) is not None:
                    getattr(self, name).free()

    def _q0_of(self, rows):
        return None if self._q0 is None else self._q0[rows]

    def upload(self, nsteps, qlat, q0):
        """Stage this rank's slice of the forcing (global arrays in, local slices uploaded).  q0 = None: every plan
        continues from the state its last window left in HBM (the reference's new_q0 warm start between windows,
        AbstractNetwork.py:177-191, without the host round trip)."""
        self.nsteps = nsteps
        self._qlat, self._q0 = qlat, q0
        # every upload() re-stages the merged plan too, whether or not the caller reuses its arrays (a forcing buffer
        # refilled in place is a new window all the same)
        self._upload_gen = getattr(self, "_upload_gen", 0) + 1
        if q0 is None and self.planM is not None:
            # the windows so far ran on the merged plan (sub-basins + time-skewed trunk, the short-timestep device path):
            # that is where the resident state lives; _merged_plan() stages the new forcing there
            self._plan0_staged = False
            return
        self.plan0.upload_forcing(nsteps, qlat[self.rows0], self._q0_of(self.rows0))
        self._plan0_staged = True

    def route_resident(self, qts_subdivisions, assume_short_ts):
        """Single-rank form that leaves the outlet hydrographs in HBM (throughput mode): returns the
        outlet rows; ``outlet_hydrographs()`` copies the block to the host when it is wanted."""
        if self.world != 1:
            raise ValueError("route_resident is the one-GPU path; use route_on_device with a process group")
        self.last_stats = {"phase0": self.plan0.route_device(self.nsteps, qts_subdivisions, assume_short_ts)}
        self.plan0.gather_flow_rows_resident(self.my_out0_local)
        return self.my_out0_global

    def outlet_hydrographs(self):
        return self.plan0.download_gathered()

    # ---- what a throughput-mode caller consumes of a window (outlet hydrographs + final state), copied to the
    # host on a copy stream BESIDE the next window: fetch_begin() after a window, fetch_wait() after the next one is queued
    def route_and_fetch(self, qts_subdivisions, assume_short_ts):
        """Single-rank form: route the window, start the asynchronous copy of its outlet hydrographs and final state
        (page-locked arrays), and hand back the products of the PREVIOUS call (None, None the first time)."""
        if self.world != 1:
            raise ValueError("route_and_fetch is the one-GPU path")
        self.last_stats = {"phase0": self.plan0.route_device(self.nsteps, qts_subdivisions, assume_short_ts)}
        prev = self.fetch_wait()          # (its copy ran beside the window just routed)
        if not hasattr(self, "_rs_fetch"):
            self._rs_fetch = self.plan0.rowset(self.my_out0_local)
        self.plan0.fetch_begin(self._rs_fetch, True)
        self._fetching = ("single",)
        return prev

    def fetch_begin(self, hyd_dev, want_hyd=True, output_stride=None):
        """Multi-rank form, after route_on_device(): this rank's final state (every plan that holds one) and, if wanted,
        the gathered outlet block start their way to the host; with output_stride = n also every n-th step of (q, v, d) of
        this rank's rows (fetch_wait() then returns a third item: one block per plan that holds state)."""
        X = self._X
        if not hasattr(self, "_cs"):
            self._cs = X.stream_create(self._dev)
        for plan in self._state_plans:
            plan.fetch_begin(None, True, output_stride)
        hyd = None
        if want_hyd:                       # (a ring of three page-locked blocks, made once: no allocation in a steady pipeline)
            ring = getattr(self, "_hyd_ring", None)
            if ring is None or ring[0][0].shape != hyd_dev.shape:
                from . import _lib
                ring = [[_lib.result_empty(hyd_dev.shape, hyd_dev.dtype, always_pinned=True) for _ in range(3)], 0]
                self._hyd_ring = ring
            hyd = hyd_dev.download_async(self._cs, out=ring[0][ring[1] % 3])
            ring[1] += 1
        self._fetching = ("dist", hyd, output_stride is not None)

    def fetch_wait(self):
        f, self._fetching = getattr(self, "_fetching", None), None
        if f is None:
            return None, None
        if f[0] == "single":
            return self.plan0.fetch_wait()
        got = [plan.fetch_wait() for plan in self._state_plans]
        self._X.stream_synchronize(self._dev, self._cs)
        if f[2]:
            return f[1], [g[1] for g in got], [g[2] for g in got]
        return f[1], [g[1] for g in got]

    def route(self, qts_subdivisions, assume_short_ts, all_gather=None):
        """One routing window.  ``all_gather(array) -> list of arrays (one per rank)``.
        Returns (outlet_rows, outlet_hydrographs[nout, nsteps]) for the whole job."""
        nsteps = self.nsteps
        st0 = self.plan0.route_device(nsteps, qts_subdivisions, assume_short_ts)
        stats = {"phase0": st0}
        cut_q = None
        if self.cut_rows.size:
            mine = (self.plan0.gather_flow_rows(self.my_cut_local) if self.my_cut_local.size
                    else np.zeros((0, nsteps), dtype=self.dtype))
            parts = [mine] if all_gather is None else all_gather(mine)
            cut_q = np.zeros((self.cut_rows.shape[0], nsteps), dtype=self.dtype)
            for r, blk in enumerate(parts):
                cut_q[self.cut_owner == r] = blk
        out1 = np.zeros((0, nsteps), dtype=self.dtype)
        if self.plan1 is not None:
            bf = np.zeros((int(self.boundary1.sum()), nsteps, 3), dtype=self.dtype)
            bf[:, :, 0] = cut_q[self.b_cut_index]
            self.plan1.upload_forcing(nsteps, self._qlat[self.rows1], self._q0_of(self.rows1), bf)
            stats["phase1"] = self.plan1.route_device(nsteps, qts_subdivisions, assume_short_ts)
            if self.my_out1_global.size:
                out1 = self.plan1.gather_flow_rows(self.my_out1_local)
        out0 = (self.plan0.gather_flow_rows(self.my_out0_local) if self.my_out0_global.size
                else np.zeros((0, nsteps), dtype=self.dtype))
        rows = np.concatenate([self.my_out0_global, self.my_out1_global])
        hyd = np.concatenate([out0, out1], 0)
        if all_gather is not None:
            rows = np.concatenate(all_gather(rows))
            hyd = np.concatenate(all_gather(hyd), 0)
        order = np.argsort(rows, kind="stable")
        self.last_stats = stats
        return rows[order], hyd[order]
