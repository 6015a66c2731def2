"""Device-resident, time-pipelined multi-rank path (ShardedRouter.route_on_device) on ONE GPU: two ranks run as two
threads of this process and exchange through an in-process stand-in for
torch.distributed.all_gather_into_tensor.  Checks that the HBM-to-HBM hand-off (sub-basin outlets ->
trunk boundary rows, final outlet gather) reproduces the single-rank result bit for bit."""
import threading

import numpy as np
import pytest

from troute_amd import synthetic
from troute_amd.distributed import ShardedRouter

pytestmark = pytest.mark.gpu


class ThreadAllGather:
    def __init__(self, world):
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)

    def for_rank(self, rank):
        def all_gather_into(out, t):
            """torch.distributed.all_gather_into_tensor for threads of one process: ordered against the
            caller's current stream like the real collective (waits for it, leaves the result on it)"""
            import torch
            torch.cuda.current_stream().synchronize()
            self.slots[rank] = t
            self.barrier.wait()
            for r in range(self.world):
                out[r].copy_(self.slots[r])
            torch.cuda.current_stream().synchronize()
            self.barrier.wait()
        return all_gather_into


@pytest.mark.parametrize("short,nchunks,retune", [(True, None, False), (True, 1, False), (True, 5, False),
                                                  (False, None, False), (False, 3, False), (True, None, True),
                                                  (False, None, True)])
def test_two_ranks_device_exchange_equals_single_rank(short, nchunks, retune):
    _two_ranks(short, nchunks, retune)


def test_two_ranks_with_chunk_overlap_on_two_streams(monkeypatch):
    """TRMC_FLOW_OVERLAP=1 (opt-in): consecutive time chunks of a resident window alternate between two compute streams,
    flow and depth handed over through granules -- the same bits."""
    monkeypatch.setenv("TRMC_FLOW_OVERLAP", "1")
    _two_ranks(True, 5, False)
    _two_ranks(True, None, True)


def _two_ranks(short, nchunks, retune):
    import torch
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    q0 = np.zeros((nseg, 3), np.float32)
    nsteps, qts = 24, 12

    single = ShardedRouter(net["to"], net["params"])
    single.upload(nsteps, net["qlat"], q0)
    rows1, hyd1 = single.route(qts, short)
    single.close()

    world = 2
    ag = ThreadAllGather(world)
    results = [None] * world
    errors = []

    def run(rank):
        try:
            r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0)
            r.enable_device_exchange(torch, torch.device("cuda", 0))
            r.upload(nsteps, net["qlat"], q0)
            r.upload_trunk()
            if retune:                                  # every rank rebuilds its plans from its own tuning window
                r.collect_cost(True)
                r.route_on_device(qts, short, ag.for_rank(rank), nchunks)
                hint = r.iteration_hint()
                assert hint.max() > 0
                r.close()
                r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0, cost_hint=hint)
                r.enable_device_exchange(torch, torch.device("cuda", 0))
                r.upload(nsteps, net["qlat"], q0)
                r.upload_trunk()
            for _ in range(2):                          # twice: the staged buffers must be reusable
                rows, hyd = r.route_on_device(qts, short, ag.for_rank(rank), nchunks)
            results[rank] = (rows, hyd.cpu().numpy(), r.cut_rows.shape[0], r.plan1 is not None)
            r.close()
        except Exception as e:                          # pragma: no cover
            errors.append(e)
            ag.barrier.abort()

    ts = [threading.Thread(target=run, args=(k,)) for k in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert results[0][2] > 0 and (results[0][3] or results[1][3])
    for rows, hyd, _, _ in results:
        assert np.array_equal(rows, rows1)
        assert np.array_equal(hyd.view(np.uint32), hyd1.view(np.uint32))
