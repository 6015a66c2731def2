"""TEST INFRASTRUCTURE: an oracle-backed stand-in with RoutingPlan's interface, so the multi-rank
partition / exchange / gather logic (troute_amd/distributed.py) can be exercised on CPU with gloo.
Never used by the product."""
import numpy as np

from oracle import oracle as O
from troute_amd.plan import topology_levels


class OraclePlan:
    def __init__(self, up_ptr, up_idx, params, boundary=None, precision=32, device=0):
        self.up_ptr = np.asarray(up_ptr, np.int64)
        self.up_idx = np.asarray(up_idx, np.int64)
        self.params = np.asarray(params, np.float32)
        self.nseg = self.params.shape[0]
        self.boundary = np.zeros(self.nseg, bool) if boundary is None else (np.asarray(boundary).astype(np.uint8) == 1)   # (2: routed, see trmc.h)
        self.level, _, self.nlevels = topology_levels(self.up_ptr, self.up_idx, self.boundary.astype(np.uint8))
        self.dtype = np.float32

    def close(self):
        pass

    def info(self):
        return {"nseg": self.nseg, "nseg_routed": int((~self.boundary).sum()), "nlevels": self.nlevels,
                "precision": 32, "device": -1}

    def upload_forcing(self, nsteps, qlat, q0, boundary_fvd=None):
        self.nsteps, self.qlat, self.q0, self.bf = nsteps, np.asarray(qlat, np.float32), np.asarray(q0, np.float32), boundary_fvd

    def route_device(self, nsteps, qts, short_ts):
        init = np.zeros((self.nseg, nsteps + 1, 3), np.float32)
        if self.boundary.any():
            init[self.boundary, 1:, :] = self.bf
        self.fvd = O.network_by_segment(nsteps, qts, self.up_ptr, self.up_idx, self.level, self.params, self.q0,
                                        self.qlat, short_ts, routed=~self.boundary,
                                        prefilled=self.boundary.astype(np.uint8), fvd_init=init, det=True)
        return {"segment_steps": int((~self.boundary).sum()) * nsteps, "ms_main": 0.0, "ms_total": 0.0,
                "main_launches": 0}

    def gather_flow_rows(self, rows):
        return np.ascontiguousarray(self.fvd[np.asarray(rows, np.int64), 1:, 0])

    def download_fvd(self):
        return np.ascontiguousarray(self.fvd[:, 1:, :])


class OracleStreamPlan(OraclePlan):
    """... and of a plan in cluster order that runs a STREAM of windows (include/trmc.h trmc_stream_*), for troute_amd.sequence.
    RouteStream's multi-rank protocol on the CPU.  The ORDER is the product's (trmc_topology_clusters: host code of the library --
    the lags that decide when a day's cut-edge hydrographs may be exchanged are the real ones); the SCHEDULE is checked the way the
    library checks it (a row set may be gathered, a day waited for, only when the launches queued so far have brought the rows that
    far); the ARITHMETIC is the oracle's, day by day.  Rows below a boundary row whose flows have not arrived yet are NaN: a
    protocol that reads them too early fails loudly."""
    wide_min_rows, cluster_rows, tile_steps = 64, 128, 8

    def __init__(self, up_ptr, up_idx, params, boundary=None, precision=32, device=0, short=None, extra_options=None, cost_hint=None, **kw):
        super().__init__(up_ptr, up_idx, params, boundary, precision, device)
        from troute_amd.plan import topology_clusters
        b = None if boundary is None else np.asarray(boundary, np.uint8)
        late = int((extra_options or {}).get("cluster_late_lag", 0))
        import ctypes as C
        from troute_amd import _lib
        # (trmc_topology_clusters has no late-lag argument: the plan's own order is asked for through a host-only plan build would
        # need a device -- so the late lag is applied here the way topology.cpp does: rows fed by boundary rows or marked 2 start
        # a cluster level of their own at `late`, and everything below them follows)
        pos, lag, blk, W, Cn, nb = topology_clusters(self.up_ptr, self.up_idx, b, None, wide_min_rows=self.wide_min_rows,
                                                     wide_max_levels=16, cluster_rows=self.cluster_rows)
        lag = lag.copy()
        if late > 0 and b is not None:
            order = np.argsort(self.level, kind="stable")
            latest = np.zeros(self.nseg, bool)
            for r in order:
                if self.level[r] < 0:
                    continue
                ups = self.up_idx[self.up_ptr[r]:self.up_ptr[r + 1]]
                if b[r] == 2 or (ups.size and ((b[ups] == 1).any() or latest[ups].any())):
                    latest[r] = True
            lag[latest] = np.maximum(lag[latest], late) + (lag[latest] - lag[latest].min() if latest.any() else 0)
        self._lag, self._W, self._C = lag.astype(np.int32), W, int(lag.max(initial=0)) - W + 1
        self._rowsets, self.nboundary = [], int(self.boundary.sum())
        self._s = None

    def lags(self):
        return self._lag, self._W, self._C

    def rowset(self, rows):
        self._rowsets.append(np.asarray(rows, np.int64))
        return len(self._rowsets) - 1

    def collect_cost(self, enable=True):
        pass

    def upload_forcing(self, nsteps, qlat, q0, boundary_fvd=None):
        if q0 is None:
            q0 = self._final
        super().upload_forcing(nsteps, qlat, q0, boundary_fvd)

    # ---- the stream ---------------------------------------------------------------------------------------------------
    def stream_begin(self, nsteps, qts, slots=0, full_output=False, output_stride=0):
        assert nsteps % self.tile_steps == 0
        tpd = nsteps // self.tile_steps
        lmax = int(self._lag.max(initial=0))
        self._s = {"nsteps": nsteps, "qts": qts, "tpd": tpd, "lmax": lmax, "slots": max(slots, 2 + -(-(lmax + 1) // tpd)), "days": [],
                   "g": -1, "state0": np.asarray(self.q0, np.float32), "clean": 0, "launches": 0,
                   "stride": int(output_stride or 0), "full": bool(full_output)}

    def stream_info(self):
        s = self._s
        done = sum(1 for d in range(len(s["days"])) if s["g"] >= (d + 1) * s["tpd"] - 1 + s["lmax"])
        return {"slots": s["slots"], "tiles_per_day": s["tpd"], "lag_max": s["lmax"], "wide_levels": self._W, "cluster_levels": self._C,
                "days_pushed": len(s["days"]), "days_complete": done, "launches": s["launches"]}

    def stream_push(self, qlat, boundary_q_ptr=None, rowset=None, hyd=None, q0=None, fvd=None):
        s = self._s
        assert s["g"] <= len(s["days"]) * s["tpd"] - 1 or self.stream_info()["days_complete"] == len(s["days"]), "advanced, not flushed"
        assert fvd is None or s["stride"] or s["full"], "no (q, v, d) block in a stream begun without full_output / output_stride"
        s["days"].append({"qlat": np.array(qlat, np.float32), "bq": None, "rowset": rowset, "hyd": hyd, "q0": q0, "fvd": None,
                          "fvd_out": fvd})
        s["g"] = len(s["days"]) * s["tpd"] - 1
        s["launches"] += s["tpd"]
        return len(s["days"]) - 1

    def _compute_through(self, day):
        """route days clean..day with the oracle (boundary rows whose flows have not been set: NaN)"""
        s = self._s
        for d in range(s["clean"], day + 1):
            rec = s["days"][d]
            state = s["state0"] if d == 0 else s["days"][d - 1]["final"]
            init = np.zeros((self.nseg, s["nsteps"] + 1, 3), np.float32)
            if self.boundary.any():
                init[self.boundary, 1:, 0] = np.nan if rec["bq"] is None else rec["bq"]
            rec["fvd"] = O.network_by_segment(s["nsteps"], s["qts"], self.up_ptr, self.up_idx, self.level, self.params, state, rec["qlat"],
                                              True, routed=~self.boundary, prefilled=self.boundary.astype(np.uint8), fvd_init=init, det=True)
            f = rec["fvd"][:, -1, :]
            rec["final"] = np.stack([f[:, 0], f[:, 0], f[:, 2]], 1)
        # (a day is final once its boundary flows are there -- and those of every day before it)
        while s["clean"] <= day and (not self.boundary.any() or s["days"][s["clean"]]["bq"] is not None):
            s["clean"] += 1

    def stream_gather_host(self, day, rowset):
        s, rows = self._s, self._rowsets[rowset]
        need = (day + 1) * s["tpd"] - 1 + (int(self._lag[rows].max()) if rows.size else 0)
        assert s["g"] >= need, f"rows gathered {need - s['g']} launches before they are through day {day}"
        self._compute_through(day)
        out = np.ascontiguousarray(s["days"][day]["fvd"][rows, 1:, 0])
        assert np.isfinite(out).all(), "a row set gathered below boundary rows whose flows have not arrived"
        return out

    def stream_boundary_host(self, day, flows):
        s = self._s
        # the rows that read boundary rows must not have begun the day: their first tile of it is queued lag tiles after day * tpd
        fed = np.zeros(self.nseg, bool)
        for r in np.flatnonzero(~self.boundary):
            ups = self.up_idx[self.up_ptr[r]:self.up_ptr[r + 1]]
            fed[r] = ups.size > 0 and self.boundary[ups].any()
        first = day * s["tpd"] + int(self._lag[fed].min()) if fed.any() else 1 << 60
        assert s["g"] < first, f"boundary flows of day {day} arrive {s['g'] - first + 1} launches after the rows that read them began the day"
        assert all(s["days"][d]["bq"] is not None for d in range(day)), "boundary days out of order"
        s["days"][day]["bq"] = np.array(flows, np.float32)
        s["clean"] = min(s["clean"], day)

    def stream_advance(self, ntiles):
        s = self._s
        last = len(s["days"]) * s["tpd"] - 1 + s["lmax"]
        new = min(last, max(s["g"], len(s["days"]) * s["tpd"] - 1) + ntiles)
        s["launches"] += max(0, new - s["g"])
        s["g"] = max(s["g"], new)

    def stream_flush(self):
        self.stream_advance(1 << 30)

    def stream_wait(self, day):
        s = self._s
        assert s["g"] >= (day + 1) * s["tpd"] - 1 + s["lmax"], "RuntimeError: not queued to its end"
        self._compute_through(day)
        rec = s["days"][day]
        assert np.isfinite(rec["fvd"][~self.boundary]).all(), f"day {day} handed over with rows that never got their inflows"
        if rec["hyd"] is not None and rec["rowset"] is not None:
            rows = self._rowsets[rec["rowset"]]
            rec["hyd"][:rows.shape[0]] = rec["fvd"][rows, 1:, 0]
        if rec["q0"] is not None:
            rec["q0"][...] = rec["final"]
        if rec["fvd_out"] is not None:                      # every stride-th step (or every step) of every row's (q, v, d)
            n = s["stride"] or 1
            rec["fvd_out"][...] = rec["fvd"][:, 1:, :][:, n - 1::n, :]

    def stream_day_ms(self, day):
        return 0.0

    def stream_end(self):
        self.stream_flush()
        s = self._s
        if s["days"]:
            self._compute_through(len(s["days"]) - 1)
            self._final = s["days"][-1]["final"]
