"""Device-resident, time-pipelined multi-rank path (ShardedRouter.route_on_device) on ONE GPU: two ranks run as two
threads of this process, each with its own communicator handle (troute_amd.comm.Comm over the C ABI; the shared-memory
transport, because RCCL refuses two ranks on one device -- the calls, buffers, streams and events are the ones the RCCL
transport gets).  Checks that the HBM-to-HBM hand-off (sub-basin outlets -> trunk boundary rows, final outlet gather)
reproduces the single-rank result bit for bit."""
import os
import threading

import numpy as np
import pytest

from troute_amd import synthetic
from troute_amd.comm import Comm
from troute_amd.distributed import ShardedRouter

pytestmark = pytest.mark.gpu
_serial = [0]


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
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    q0 = np.zeros((nseg, 3), np.float32)
    nsteps, qts = 24, 12

    single = ShardedRouter(net["to"], net["params"])
    single.upload(nsteps, net["qlat"], q0)
    rows1, hyd1 = single.route(qts, short)
    single.close()

    world = 2
    _serial[0] += 1
    key = f"sim{os.getpid()}_{_serial[0]}"
    results = [None] * world
    errors = []

    def run(rank):
        try:
            comm = Comm(rank, world, device=0, backend="shm", key=key)
            r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0)
            r.enable_device_exchange(comm)
            r.upload(nsteps, net["qlat"], q0)
            r.upload_trunk()
            if retune:                                  # every rank rebuilds its plans from its own tuning window
                r.collect_cost(True)
                r.route_on_device(qts, short, nchunks)
                hint = comm.all_reduce_max_host(r.iteration_hint())   # every rank measured its own rows
                assert hint.max() > 0
                r.close()
                r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0, cost_hint=hint)
                r.enable_device_exchange(comm)
                r.upload(nsteps, net["qlat"], q0)
                r.upload_trunk()
            for _ in range(2):                          # twice: the staged buffers must be reusable
                rows, hyd = r.route_on_device(qts, short, nchunks)
            results[rank] = (rows, hyd.numpy(), r.cut_rows.shape[0], r.plan1 is not None)
            r.close()
            comm.close()
        except Exception as e:                          # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=run, args=(k,)) for k in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert results[0][2] > 0 and (results[0][3] or results[1][3])
    for rows, hyd, _, _ in results:
        assert np.array_equal(rows, rows1)
        assert np.array_equal(hyd.view(np.uint32), hyd1.view(np.uint32))


def _probe_worker(rank, world, key, tmp):
    import numpy as np
    from troute_amd import comm as X
    c = X.Comm(rank, world, device=0, backend="probe", key=key)
    send = X.DeviceBuffer.from_array(0, np.full(64, rank + 1, np.uint8))
    recv = X.DeviceBuffer(0, 64 * world)
    st = X.stream_create(0)
    c.all_gather(send.ptr, recv.ptr, 64, st)
    got = recv.download((world, 64), np.uint8, st)
    assert (got == (np.arange(world, dtype=np.uint8) + 1)[:, None]).all()
    c.barrier()
    open(os.path.join(tmp, f"backend_{rank}"), "w").write(c.backend)
    c.close()


def test_communicator_probes_rccl_and_falls_back_when_it_cannot_start(tmp_path):
    """backend="probe" (what "auto" does when every rank has a device): RCCL is tried in a child process per rank under a
    time-out, the verdicts are agreed, and the job's communicator is RCCL only if it started everywhere.  Two ranks that
    SHARE device 0 -- RCCL refuses that -- must agree on the shared-memory transport and still gather correctly."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    key = f"p{os.getpid()}"
    ps = [ctx.Process(target=_probe_worker, args=(r, 2, key, str(tmp_path))) for r in range(2)]
    [p.start() for p in ps]
    [p.join(300) for p in ps]
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    backends = [open(tmp_path / f"backend_{r}").read() for r in range(2)]
    assert backends[0] == backends[1] and backends[0] in ("shm", "rccl"), backends
