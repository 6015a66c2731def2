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


@pytest.mark.parametrize("nchunks,retune,wide_k", [(None, False, 4), (6, False, 8), (3, True, 4), (1, False, 4)])
def test_two_ranks_on_the_level_engine_with_wide_tiles_beside_a_lagged_trunk(monkeypatch, nchunks, retune, wide_k):
    """The trunk's owner on the level engine: its lagged rows (and the rows the cut-edge boundary rows feed) stay in the
    tail, so the leading levels of its merged plan are still routed several timesteps per launch ahead of the window
    (k_mc_tile) while the trunk rides `lag` launches behind in the tail's LAG form -- the same bits as one rank, whatever
    the chunking of the hand-off."""
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
    monkeypatch.setenv("TRMC_WIDE_K", str(wide_k))
    used = _two_ranks(True, nchunks, retune, engine="levels", nsteps=36)
    assert any(w > 0 for w in used), used        # at least one rank (the trunk's owner among them) took the wide path


def _two_ranks(short, nchunks, retune, engine=None, nsteps=24):
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    q0 = np.zeros((nseg, 3), np.float32)
    qts = 12
    kw = {} if engine is None else {"engine": engine, "assume_short_ts": short}

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
            r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0, **kw)
            r.enable_device_exchange(comm)
            r.upload(nsteps, net["qlat"], q0)
            r.upload_trunk()
            if retune:                                  # every rank rebuilds its plans from its own tuning window
                r.collect_cost(True)
                r.route_on_device(qts, short, nchunks)
                hint = comm.all_reduce_max_host(r.iteration_hint())   # every rank measured its own rows
                assert hint.max() > 0
                r.close()
                r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0, cost_hint=hint, **kw)
                r.enable_device_exchange(comm)
                r.upload(nsteps, net["qlat"], q0)
                r.upload_trunk()
            for _ in range(2):                          # twice: the staged buffers must be reusable
                rows, hyd = r.route_on_device(qts, short, nchunks)
            results[rank] = (rows, hyd.numpy(), r.cut_rows.shape[0], r.plan1 is not None,
                             (r.plan1 is not None, r.last_stats["phase0"].get("wide_levels", 0)))
            r.close()
            comm.close()
        except Exception as e:                          # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=run, args=(k,)) for k in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert results[0][2] > 0 and (results[0][3] or results[1][3])
    for rows, hyd, _, _, _ in results:
        assert np.array_equal(rows, rows1)
        assert np.array_equal(hyd.view(np.uint32), hyd1.view(np.uint32))
    owners = [res[4][1] for res in results if res[4][0]]
    return owners if owners else [res[4][1] for res in results]


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
