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
