"""One-process-per-GPU routing of a partitioned network (see sharding.py).

Phase 0: every rank routes the sub-basins / small networks it owns (one plan).
Exchange: outlet hydrographs of the cut sub-basins are all-gathered (RCCL over
xGMI when the process group is NCCL; gloo in the CPU tests) -- the data the
reference hands from one sub-network order to the next as
``flowveldepth_interorder`` (compute.py:882-897, consumed mc_reach.pyx:458-469).
Phase 1: the rank that owns a trunk routes it with those hydrographs as
prescribed boundary rows.  Finally the network-outlet hydrographs are gathered.

The compute backend is injected (``plan_factory``): the product passes
``troute_amd.plan.RoutingPlan`` (HIP); the CPU test-suite passes an
oracle-backed stand-in with the same interface so the partition / exchange /
gather logic is exercised with world_size 2 on gloo without a GPU.
"""
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


class ShardedRouter:
    def __init__(self, to, params, rank=0, world=1, device=0, plan_factory=None, precision=32,
                 partition=None):
        if plan_factory is None:
            from .plan import RoutingPlan as plan_factory  # the HIP engine; no fallback
        from .synthetic import upstream_csr
        self.rank, self.world = rank, world
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
        self.plan0 = plan_factory(lp, li, params[self.rows0], None, precision, device)
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
            self.plan1 = plan_factory(lp1, li1, params[rows1], boundary.astype(np.uint8), precision, device)
            o1 = outlets[np.isin(outlets, trunk_rows)]
            self.my_out1_global, self.my_out1_local = o1, g2l1[o1]

    # ---- device-resident exchange (torch tensors; NCCL = RCCL over xGMI on the GPU box) --------------
    def enable_device_exchange(self, torch, device):
        """Precompute the index maps for route_on_device(): every rank knows the whole partition, so the
        position of every cut row / outlet row inside the all-gathered buffers is known up front."""
        self._torch, self._tdev = torch, device
        world = self.world
        part = self.part
        piece, phase, owner = part["piece"], part["phase"], part["owner"]
        row_phase, row_owner = phase[piece], owner[piece]
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        self._tdt = tdt
        # cut rows: slot (owner, index within the owner's ascending list)
        ncut = self.cut_rows.shape[0]
        self._max_cut = 0
        if ncut:
            counts = np.bincount(self.cut_owner, minlength=world)
            self._max_cut = int(counts.max())
            idx_in_owner = np.zeros(ncut, dtype=np.int64)
            for r in range(world):
                m = self.cut_owner == r
                idx_in_owner[m] = np.arange(int(m.sum()))
            flat = self.cut_owner.astype(np.int64) * self._max_cut + idx_in_owner
            if self.plan1 is not None:
                self._t_b_index = torch.from_numpy(flat[self.b_cut_index]).to(device)
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
        self._t_out_index = torch.from_numpy(slot[order].astype(np.int64)).to(device)

    def upload_trunk(self):
        """Stage the trunk's forcing once (its boundary hydrographs arrive per route via the exchange)."""
        if self.plan1 is not None:
            self.plan1.upload_forcing(self.nsteps, self._qlat[self.rows1], self._q0[self.rows1], None)

    def route_on_device(self, qts_subdivisions, assume_short_ts, all_gather_tensor):
        """As route(), but every hand-off stays in HBM.  ``all_gather_tensor(t) -> [world, *t.shape]``
        (torch.distributed.all_gather_into_tensor).  Returns (outlet_rows, hydrographs tensor on device)."""
        torch, dev, nsteps = self._torch, self._tdev, self.nsteps
        stats = {"phase0": self.plan0.route_device(nsteps, qts_subdivisions, assume_short_ts)}
        if self._max_cut:
            send = torch.zeros((self._max_cut, nsteps), dtype=self._tdt, device=dev)
            if self.my_cut_local.size:
                torch.cuda.current_stream().synchronize()
                self.plan0.gather_flow_rows(self.my_cut_local, device_ptr=send.data_ptr())
            recv = all_gather_tensor(send)
            if self.plan1 is not None:
                bq = recv.reshape(-1, nsteps).index_select(0, self._t_b_index).contiguous()
                torch.cuda.current_stream().synchronize()
                self.plan1.set_boundary_flow_device(nsteps, bq.data_ptr())
        if self.plan1 is not None:
            stats["phase1"] = self.plan1.route_device(nsteps, qts_subdivisions, assume_short_ts)
        send_o = torch.zeros((self._max_out, nsteps), dtype=self._tdt, device=dev)
        torch.cuda.current_stream().synchronize()
        n0 = self.my_out0_local.shape[0]
        if n0:
            self.plan0.gather_flow_rows(self.my_out0_local, device_ptr=send_o.data_ptr())
        if self.plan1 is not None and self.my_out1_global.size:
            self.plan1.gather_flow_rows(self.my_out1_local, device_ptr=send_o[n0:].data_ptr())
        recv_o = all_gather_tensor(send_o)
        hyd = recv_o.reshape(-1, nsteps).index_select(0, self._t_out_index)
        self.last_stats = stats
        return self._out_rows, hyd

    def close(self):
        self.plan0.close()
        if self.plan1 is not None:
            self.plan1.close()

    def upload(self, nsteps, qlat, q0):
        """Stage this rank's slice of the forcing (global arrays in, local slices uploaded)."""
        self.nsteps = nsteps
        self._qlat, self._q0 = qlat, q0
        self.plan0.upload_forcing(nsteps, qlat[self.rows0], q0[self.rows0])

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
            self.plan1.upload_forcing(nsteps, self._qlat[self.rows1], self._q0[self.rows1], bf)
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
