"""N > 1 path on CPU: world_size-2 gloo processes run the sharded router with an oracle-backed plan.

Checks the partition (independent networks + dominant basin cut at tributary mouths), the
phase-0 -> trunk hydrograph hand-off and the final outlet gather: the sharded job must reproduce
the single-process result bit for bit."""
import os
import sys

import numpy as np
import pytest

import helpers as H
from troute_amd import sharding, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_conus(seed=5, nseg=6000, nnet=40):
    return synthetic.generate(nseg=nseg, nnet=nnet, seed=seed, nq=3)


def test_partition_properties():
    net = small_conus()
    to = net["to"]
    for nparts in (1, 2, 4, 8):
        part = sharding.partition(to, nparts)
        piece, phase, owner = part["piece"], part["phase"], part["owner"]
        assert piece.min() >= 0 and piece.max() == phase.shape[0] - 1
        sizes = np.bincount(piece, minlength=phase.shape[0])
        assert (sizes > 0).all()
        # flow never enters a phase-0 piece from another piece; trunks are fed only through cut rows
        has = to >= 0
        src = np.flatnonzero(has)
        dst = to[src]
        cross = piece[src] != piece[dst]
        assert (phase[piece[dst[cross]]] == 1).all() and (phase[piece[src[cross]]] == 0).all()
        assert sorted(src[cross].tolist()) == sorted(part["cut_rows"].tolist())
        if nparts == 1:
            assert part["cut_rows"].size == 0
        else:
            # sub-basin rows are balanced AFTER what a trunk costs its owner has been put on that worker's scale
            load = np.bincount(owner[phase == 0], weights=sizes[phase == 0], minlength=nparts) + part["owner_bias"]
            assert load.max() <= 1.25 * load.mean() + 50
            assert (part["owner_bias"] > 0).sum() == len(np.unique(owner[phase == 1]))


def test_outlet_and_subtree_helpers():
    to = np.array([1, 2, -1, 2, -1, 4], np.int64)
    assert sharding.outlet_of(to).tolist() == [2, 2, 2, 2, 4, 4]
    assert sharding.subtree_sizes(to).tolist() == [1, 2, 4, 1, 2, 1]
    part, load = sharding.lpt_assign([5, 4, 3, 3, 1], 2)
    assert sorted(load.tolist()) == [8, 8]


def _worker(rank, world, port, tmp, short):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_plan import OraclePlan
    from troute_amd.distributed import ShardedRouter
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = small_conus()

    def all_gather_np(arr):
        arr = np.ascontiguousarray(arr)
        n = torch.tensor([arr.shape[0]], dtype=torch.int64)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        ns = [int(x) for x in ns]
        buf = torch.zeros((max(ns),) + arr.shape[1:], dtype=torch.from_numpy(arr[:0]).dtype)
        if arr.shape[0]:
            buf[:arr.shape[0]] = torch.from_numpy(arr)
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf)
        return [o[:k].numpy() for o, k in zip(outs, ns)]

    router = ShardedRouter(net["to"], net["params"], rank=rank, world=world, plan_factory=OraclePlan)
    q0 = np.zeros((net["to"].shape[0], 3), np.float32)
    router.upload(24, net["qlat"], q0)
    rows, hyd = router.route(12, short, all_gather_np)
    np.savez(os.path.join(tmp, f"out_{rank}.npz"), rows=rows, hyd=hyd, ncut=router.cut_rows.shape[0],
             has_trunk=router.plan1 is not None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("short", [True, False])
def test_world2_gloo_equals_single_process(tmp_path, short):
    import torch.multiprocessing as mp
    from oracle_plan import OraclePlan
    from troute_amd.distributed import ShardedRouter
    net = small_conus()
    q0 = np.zeros((net["to"].shape[0], 3), np.float32)
    single = ShardedRouter(net["to"], net["params"], plan_factory=OraclePlan)
    single.upload(24, net["qlat"], q0)
    rows1, hyd1 = single.route(12, short)
    assert rows1.shape[0] == 40                                     # one outlet per independent network
    port = 29500 + (os.getpid() % 2000) + (1 if short else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path), short), nprocs=2, join=True)
    outs = [np.load(tmp_path / f"out_{r}.npz") for r in range(2)]
    assert int(outs[0]["ncut"]) > 0 and (bool(outs[0]["has_trunk"]) or bool(outs[1]["has_trunk"]))
    for o in outs:                                                  # every rank ends with the full gather
        assert np.array_equal(o["rows"], rows1)
        assert np.array_equal(o["hyd"].view(np.uint32), hyd1.view(np.uint32))


def _stream_worker(rank, world, port, tmp, stride=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_plan import OracleStreamPlan
    from troute_amd.distributed import ShardedRouter
    from troute_amd.sequence import RouteStream
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooComm:
        """the three host collectives RouteStream(exchange="host") asks of a communicator, over torch.distributed / gloo"""
        def all_gather_rows_host(self, arr):
            arr = np.ascontiguousarray(arr)
            n = torch.tensor([arr.shape[0]], dtype=torch.int64)
            ns = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(ns, n)
            ns = [int(x) for x in ns]
            buf = torch.zeros((max(max(ns), 1),) + arr.shape[1:], dtype=torch.from_numpy(arr[:0]).dtype)
            if arr.shape[0]:
                buf[:arr.shape[0]] = torch.from_numpy(arr)
            outs = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(outs, buf)
            return [o[:k].numpy() for o, k in zip(outs, ns)]

        def all_reduce_max_host(self, arr):
            t = torch.from_numpy(np.array(arr, copy=True))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.numpy()

        def barrier(self):
            dist.barrier()
    net = small_conus()
    nseg = net["to"].shape[0]
    rng = np.random.default_rng(9)
    days = [rng.uniform(0, 0.5, (nseg, 2)).astype(np.float32) for _ in range(3)]
    ndays = 7
    q0 = np.zeros((nseg, 3), np.float32)
    router = ShardedRouter(net["to"], net["params"], rank=rank, world=world, plan_factory=OracleStreamPlan, stream=True)
    got = {}
    order, pushed = [], [0]

    def feed():
        for w in range(ndays):
            pushed[0] = w + 1
            yield days[w % 3]
    with RouteStream(router, 16, 8, comm=GlooComm(), output_stride=stride) as rs:
        assert rs.exchange == "host"
        for item in rs.route(feed(), q0):
            day, hyd, fin = item[:3]
            order.append((day, pushed[0]))
            got[day] = (None if hyd is None else np.array(hyd, copy=True), np.array(fin[0], copy=True),
                        np.array(item[3][0], copy=True) if stride else np.zeros((0, 0, 3), np.float32))
        rows_out, srows, dc = np.array(rs.outlet_rows, copy=True), np.array(rs.rows, copy=True), rs._dc
        lag = router._planS_lag
        slots = rs.last_info["slots"]
    np.savez(os.path.join(tmp, f"stream_{rank}.npz"), rows=rows_out, srows=srows, ndays=len(got), dc=dc, lag=lag,
             has_trunk=router.plan1 is not None, ncut=router.cut_rows.shape[0], slots=slots, order=np.array(order),
             boundary=(np.concatenate([np.zeros(router.rows0.shape[0], bool), router.boundary1]) if router.plan1 is not None
                       else np.zeros(srows.shape[0], bool)),
             **{f"blk{w}": got[w][2] for w in got},
             **{f"hyd{w}": (got[w][0] if got[w][0] is not None else np.zeros((0, 16), np.float32)) for w in got},
             **{f"fin{w}": got[w][1] for w in got})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stride", [None, 8])
def test_world2_gloo_stream_of_days_equals_single_process(tmp_path, stride):
    """troute_amd.sequence.RouteStream on two ranks (gloo, host exchange) with an oracle-backed stand-in for the plan: the
    protocol -- the trunk's lag agreed by an all-reduce over every rank's cut rows, the cut-edge hydrographs of day e exchanged
    when every rank's cut rows are through it, the drain after the last day -- against seven days routed one by one on one
    process.  The stand-in checks the schedule the way the library does and poisons rows whose inflows arrive too late."""
    import torch.multiprocessing as mp
    from oracle_plan import OraclePlan
    from troute_amd.distributed import ShardedRouter
    net = small_conus()
    nseg = net["to"].shape[0]
    rng = np.random.default_rng(9)
    days = [rng.uniform(0, 0.5, (nseg, 2)).astype(np.float32) for _ in range(3)]
    ndays = 7
    single = ShardedRouter(net["to"], net["params"], plan_factory=OraclePlan)
    state = np.zeros((nseg, 3), np.float32)
    want_h, want_s, want_f = [], [], []
    for w in range(ndays):
        single.upload(16, days[w % 3], state)
        rows1, hyd = single.route(8, True)
        f = single.plan0.download_fvd()
        state = np.stack([f[:, -1, 0], f[:, -1, 0], f[:, -1, 2]], 1)
        want_h.append(hyd)
        want_s.append(state)
        want_f.append(f)
    port = 31500 + (os.getpid() % 2000) + (7 if stride else 0)
    mp.spawn(_stream_worker, args=(2, port, str(tmp_path), stride), nprocs=2, join=True)
    outs = [np.load(tmp_path / f"stream_{r}.npz") for r in range(2)]
    assert int(outs[0]["ncut"]) > 0 and (bool(outs[0]["has_trunk"]) or bool(outs[1]["has_trunk"]))
    assert int(outs[0]["dc"]) == int(outs[1]["dc"]) >= 1 and int(outs[0]["lag"]) == int(outs[1]["lag"])
    assert np.array_equal(outs[0]["rows"], rows1)
    for r, o in enumerate(outs):
        assert int(o["ndays"]) == ndays
        srows = o["srows"]
        for w in range(ndays):
            if r == 0:                                               # rank 0 holds every day's outlet hydrographs of the whole network
                assert np.array_equal(o[f"hyd{w}"].view(np.uint32), want_h[w].view(np.uint32)), w
            assert np.array_equal(o[f"fin{w}"][:, 0].view(np.uint32), want_s[w][srows][:, 0].view(np.uint32)), (r, w)
            if stride:       # every stride-th step of every ROUTED row of the rank (the boundary copies of cut rows carry flows only)
                routed = ~o["boundary"]
                assert np.array_equal(o[f"blk{w}"][routed].view(np.uint32), want_f[w][srows[routed]][:, stride - 1::stride, :].view(np.uint32)), (r, w)
    # a stream that hands blocks over holds a slot more and delivers a day later, the same on every rank
    assert int(outs[0]["slots"]) == int(outs[1]["slots"])
    first = {int(o["order"][0][1]) for o in outs}
    assert len(first) == 1


def test_stream_with_blocks_delivers_a_day_later_and_holds_a_slot_more():
    """RouteStream on one process (oracle-backed stand-in for the plan): with an output stride every row's kept steps are among
    each day's products -- a block that takes most of a day to reach the host -- so such a stream hands a day over one day later
    than a products-only one and its ring holds one slot more; the products are the same days' either way."""
    from oracle_plan import OracleStreamPlan
    from troute_amd.distributed import ShardedRouter
    from troute_amd.sequence import RouteStream
    net = small_conus()
    nseg = net["to"].shape[0]
    rng = np.random.default_rng(4)
    days = [rng.uniform(0, 0.5, (nseg, 8)).astype(np.float32) for _ in range(9)]
    q0 = np.zeros((nseg, 3), np.float32)
    seen = {}
    for stride in (None, 4):
        router = ShardedRouter(net["to"], net["params"], plan_factory=OracleStreamPlan, stream=True)
        pushed, order, got = [0], [], {}

        def feed():
            for d in days:
                pushed[0] += 1
                yield d
        with RouteStream(router, 64, 8, output_stride=stride) as rs:
            for item in rs.route(feed(), q0):
                order.append((item[0], pushed[0]))
                got[item[0]] = [np.array(x, copy=True) for x in item[1:]]
            slots = rs.last_info["slots"]
        router.close()
        assert [d for d, _ in order] == list(range(len(days)))
        seen[stride] = (slots, order[0][1], got)
    assert seen[4][0] == seen[None][0] + 1                  # a slot more
    assert seen[4][1] == seen[None][1] + 1                  # ... and the first day handed over one push later
    for d in range(len(days)):
        assert np.array_equal(seen[4][2][d][0].view(np.uint32), seen[None][2][d][0].view(np.uint32))      # hydrographs
        assert np.array_equal(seen[4][2][d][1].view(np.uint32), seen[None][2][d][1].view(np.uint32))      # final states
        blk = seen[4][2][d][2]
        assert blk.shape == (nseg, 16, 3)
        assert np.array_equal(blk[:, -1, [0, 2]].view(np.uint32), seen[None][2][d][1][:, [0, 2]].view(np.uint32))


def _shm_worker(rank, world, key, tmp, short):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_plan import OraclePlan
    from troute_amd.comm import Comm
    from troute_amd.distributed import ShardedRouter
    comm = Comm(rank, world, device=-1, backend="shm", key=key)
    net = small_conus()
    router = ShardedRouter(net["to"], net["params"], rank=rank, world=world, plan_factory=OraclePlan)
    q0 = np.zeros((net["to"].shape[0], 3), np.float32)
    router.upload(24, net["qlat"], q0)
    rows, hyd = router.route(12, short, comm.all_gather_rows_host)
    t = comm.all_reduce_max_host(np.array([float(rank + 1)]))
    np.savez(os.path.join(tmp, f"out_{rank}.npz"), rows=rows, hyd=hyd, ncut=router.cut_rows.shape[0],
             has_trunk=router.plan1 is not None, tmax=t)
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("short", [True, False])
def test_world2_own_communicator_equals_single_process(tmp_path, short):
    """The same job with the product's OWN communicator (troute_amd.comm.Comm over the C ABI, shared-memory transport,
    no device: host-pointer collectives) instead of gloo: two processes, hand-off of the cut-edge hydrographs and the
    final outlet gather through trmc_comm_all_gather_host."""
    import multiprocessing as mp
    from oracle_plan import OraclePlan
    from troute_amd.distributed import ShardedRouter
    net = small_conus()
    q0 = np.zeros((net["to"].shape[0], 3), np.float32)
    single = ShardedRouter(net["to"], net["params"], plan_factory=OraclePlan)
    single.upload(24, net["qlat"], q0)
    rows1, hyd1 = single.route(12, short)
    ctx = mp.get_context("spawn")
    key = f"t{os.getpid()}_{int(short)}"
    ps = [ctx.Process(target=_shm_worker, args=(r, 2, key, str(tmp_path), short)) for r in range(2)]
    [p.start() for p in ps]
    [p.join(120) for p in ps]
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    outs = [np.load(tmp_path / f"out_{r}.npz") for r in range(2)]
    assert int(outs[0]["ncut"]) > 0 and (bool(outs[0]["has_trunk"]) or bool(outs[1]["has_trunk"]))
    for o in outs:
        assert np.array_equal(o["rows"], rows1)
        assert np.array_equal(o["hyd"].view(np.uint32), hyd1.view(np.uint32))
        assert float(o["tmax"][0]) == 2.0


def test_communicator_chunks_blocks_larger_than_its_segment():
    """all_gather_host of blocks larger than the shared segment goes through it a slice at a time (two threads of this
    process as the two ranks)."""
    import threading
    from troute_amd.comm import Comm
    key = f"c{os.getpid()}"
    rng = np.random.default_rng(3)
    data = [rng.integers(0, 255, 300_000, dtype=np.uint8) for _ in range(2)]
    got, errs = [None, None], []

    def run(r):
        try:
            c = Comm(r, 2, device=-1, backend="shm", key=key, shm_bytes=64 * 1024)
            got[r] = c.all_gather_host(data[r])
            parts = c.all_gather_rows_host(data[r][: 1000 * (r + 1)].reshape(-1, 10))
            assert [p.shape for p in parts] == [(100, 10), (200, 10)]
            c.close()
        except Exception as e:      # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert not errs, errs
    for r in range(2):
        assert np.array_equal(got[r][0], data[0]) and np.array_equal(got[r][1], data[1])


def test_a_segment_left_by_a_dead_job_is_never_used():
    """A crashed or killed job leaves its shared-memory segment behind: `attached` >= world, barrier counters mid-count.  A
    new launch under the SAME key must not trust it: rank 0 removes and re-creates the name exclusively, the other ranks
    only proceed on a segment whose live rank 0 has answered their token -- also when they come first and find the dead
    one.  Same for the RCCL id, which now travels over such a communicator instead of a file."""
    import mmap
    import threading
    import time
    from troute_amd import comm as X
    from troute_amd.comm import Comm
    key = f"stale{os.getpid()}"
    path = f"/dev/shm/trmc_{key}"
    world = 2

    def plant_dead_segment():
        with open(path, "wb") as f:                                   # header page + data area of a dead two-rank job
            hdr = np.zeros(1024, np.uint32)
            hdr[0], hdr[1], hdr[2], hdr[3] = 1, 7, 2, world           # arrived = 1 (mid-barrier), generation, attached = world
            hdr[4] = 0x74726d63                                       # even its magic is intact
            f.write(hdr.tobytes() + bytes(64 * 1024))
    for late_rank0 in (False, True):
        plant_dead_segment()
        got, errs = [None] * world, []

        def run(r):
            try:
                if (r == 0) == late_rank0:
                    time.sleep(0.5)                                   # the other rank meets the dead segment first
                c = Comm(r, world, device=-1, backend="shm", key=key, shm_bytes=64 * 1024)
                for it in range(20):                                  # barriers inside: counters must start from zero
                    got[r] = c.all_gather_host(np.full(1000, 10 * it + r, np.int32))
                    assert (got[r][0] == 10 * it).all() and (got[r][1] == 10 * it + 1).all()
                c.close()
            except Exception as e:      # pragma: no cover
                errs.append(e)
        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join(60) for t in ts]
        assert not errs, errs
        assert not os.path.exists(path)                               # the last rank out removed the name
    # the id of an RCCL communicator reaches the other ranks over the same kind of channel, never through a stale file
    ids = [None] * world
    blob = bytes(range(128))

    def run_id(r):
        ids[r] = X.exchange_id(r, world, key, lambda: blob)
    ts = [threading.Thread(target=run_id, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert ids == [blob, blob]


def test_partition_by_measured_cost_balances_cost_not_rows():
    """sharding.partition(row_cost=...): pieces are packed by the cost their rows were measured to need."""
    from troute_amd import sharding, synthetic
    net = synthetic.generate(nseg=60000, nnet=300, seed=5)
    to = net["to"]
    rng = np.random.default_rng(0)
    sub = sharding.subtree_sizes(to)
    cost = np.where(sub > 20, 40, 16).astype(np.uint8)           # wet downstream rows cost 2.5 x the dry headwaters
    cost[rng.random(to.shape[0]) < 0.01] = 112
    for nparts in (2, 4, 8):
        plain = sharding.partition(to, nparts)
        byc = sharding.partition(to, nparts, row_cost=cost)
        assert np.array_equal(plain["cut_rows"], byc["cut_rows"]) and np.array_equal(plain["piece"], byc["piece"])
        def loads(p):
            own = p["owner"][p["piece"]]
            return np.bincount(own, weights=cost.astype(np.float64), minlength=nparts)
        lp, lc = loads(plain), loads(byc)
        # workers that own no trunk carry equal COST (a trunk's owner is spared some on purpose, see partition)
        free = np.setdiff1d(np.arange(nparts), byc["owner"][byc["phase"] == 1])
        if free.size > 1:
            assert lc[free].max() / lc[free].min() < 1.05
            assert lc[free].max() / lc[free].min() <= lp[free].max() / lp[free].min() + 0.02
        if free.size:
            assert (lc[np.setdiff1d(np.arange(nparts), free)] <= lc[free].max()).all()


def test_one_hardware_queue_per_stream_priority_is_set_at_construction_not_at_import():
    """GPU_MAX_HW_QUEUES=1 (with several queues per priority some stream-to-queue assignments put a plan's
    launches in a slow mode) is set when a communicator or a sharded router is BUILT and the HIP runtime is not up yet --
    importing troute_amd.distributed has no side effect on other HIP users of the process; a caller's own setting wins; and
    when the runtime has already been initialised the library says so instead of changing the variable to no effect."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)

    def run(code):
        out = subprocess.run([sys.executable, "-W", "always", "-c", code], capture_output=True, text=True, env=env, cwd=root)
        assert out.returncode == 0, out.stderr
        return out.stdout.strip(), out.stderr
    pre = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import troute_amd.distributed, troute_amd.comm; from troute_amd import _lib; "
    assert run(pre + "print(os.environ.get('GPU_MAX_HW_QUEUES'))")[0] == "None"                       # import alone: nothing
    assert run(pre + "_lib.single_hw_queue_per_priority('t'); print(os.environ['GPU_MAX_HW_QUEUES'])")[0] == "1"
    out, err = run(pre + "_lib.mark_hip_started(); _lib.single_hw_queue_per_priority('t'); print(os.environ.get('GPU_MAX_HW_QUEUES'))")
    assert out == "None" and "RuntimeWarning" in err and "GPU_MAX_HW_QUEUES" in err
    assert run("import os; os.environ['GPU_MAX_HW_QUEUES'] = '4'; from troute_amd import _lib; _lib.single_hw_queue_per_priority('t'); "
               "print(os.environ['GPU_MAX_HW_QUEUES'])")[0] == "4"
    # the constructors call it: a router built on the CPU stand-in (plan_factory given) leaves the variable alone, the
    # product path (no factory: the HIP plan) sets it before the first plan is made
    src = open(os.path.join(root, "t-route_amd", "distributed.py")).read() + open(os.path.join(root, "t-route_amd", "comm.py")).read()
    assert src.count("single_hw_queue_per_priority(") == 2 and "environ.setdefault(\"GPU_MAX_HW_QUEUES\"" not in src


@pytest.mark.parametrize("seed,nseg,nnet,nparts", [(5, 60000, 300, 4), (23, 90000, 120, 8), (7, 40000, 500, 2)])
def test_partition_by_measured_rank_pace(seed, nseg, nnet, nparts):
    """sharding.partition(rank_speed=..., previous=...): what a trunk costs its owner is MEASURED -- the pace (cost carried /
    time taken) every rank kept on a window routed with an earlier partition -- not assumed: pieces go where they are
    finished first, so the predicted finish times (cost / pace) come out equal; trunks keep the owners the paces were
    measured with.  Several networks and rank counts: nothing here is fitted to one of them."""
    net = synthetic.generate(nseg=nseg, nnet=nnet, seed=seed)
    to = net["to"]
    rng = np.random.default_rng(seed)
    cost = rng.choice(np.array([16, 24, 40, 112], np.uint8), size=to.shape[0], p=[0.45, 0.35, 0.19, 0.01])
    first = sharding.partition(to, nparts, row_cost=cost)
    own = first["owner"][first["piece"]]
    loads = np.bincount(own, weights=cost.astype(np.float64), minlength=nparts)
    assert loads.max() / loads.mean() < 1.03                       # without paces: equal cost everywhere, trunks at their cost
    owners = np.unique(first["owner"][first["phase"] == 1])
    # a measured window in which the trunks' owners were 25 % slower than their cost says, another rank 10 % faster
    pace = np.ones(nparts)
    pace[owners] = 0.75
    fast = np.setdiff1d(np.arange(nparts), owners)
    if fast.size:
        pace[fast[0]] = 1.10
    times = loads / pace
    speed = sharding.rank_speeds(loads, times)
    assert speed.mean() == pytest.approx(1.0) and np.allclose(speed / speed[0], pace / pace[0])
    second = sharding.partition(to, nparts, row_cost=cost, rank_speed=speed, previous=first)
    assert np.array_equal(second["cut_rows"], first["cut_rows"]) and np.array_equal(second["piece"], first["piece"])
    p1 = first["phase"] == 1
    assert np.array_equal(second["owner"][p1], first["owner"][p1])                      # trunks stay where they were timed
    own2 = second["owner"][second["piece"]]
    loads2 = np.bincount(own2, weights=cost.astype(np.float64), minlength=nparts)
    finish = loads2 / pace
    assert finish.max() / finish.min() < 1.05, finish
    if fast.size:
        assert (loads2[owners] < loads[owners]).all()                                   # the slow ranks were relieved


def test_every_rank_arrives_at_the_same_number_of_chunks(monkeypatch):
    """The chunks of a window are its all-gathers: every rank must count the same.  The count follows the engines of ALL
    ranks' windows, derived from the partition every rank holds -- not from a rank's own engine: with a partition that leaves
    one rank under the row count at which the engines change, ranks that decided for themselves would wait for each other
    for ever (seen at N = 2 after a rebalancing by measured pace)."""
    import troute_amd.distributed as D
    from oracle_plan import OraclePlan
    from troute_amd.distributed import ShardedRouter
    net = small_conus()
    to = net["to"]
    world = 2
    part = sharding.partition(to, world, rank_speed=np.array([0.55, 1.45]))      # a lopsided partition
    sizes = np.bincount(part["owner"][part["piece"]], minlength=world)
    assert sizes.min() < 0.9 * sizes.max()
    # the threshold at which trmc_plan_create_opt changes engines, scaled down to this network: between the two ranks' sizes
    threshold = int((sizes.min() + sizes.max()) // 2)
    src = D.ShardedRouter._engines_of_all_ranks
    monkeypatch.setattr(D.ShardedRouter, "_engines_of_all_ranks",
                        lambda self: ["levels" if n >= threshold else "flow"
                                      for n in np.bincount(self.part["owner"][self.part["piece"]], minlength=self.world)])
    counts = []
    for rank in range(world):
        r = ShardedRouter(to, net["params"], rank=rank, world=world, plan_factory=OraclePlan, partition=part, assume_short_ts=True)
        r.nsteps = 288
        counts.append((r._default_chunks(None), r._chunking(None)))
    assert counts[0] == counts[1] and counts[0][0] == 8            # mixed engines: one agreed count
    monkeypatch.setattr(D.ShardedRouter, "_engines_of_all_ranks", src)
    r = ShardedRouter(to, net["params"], rank=0, world=world, plan_factory=OraclePlan, partition=part, assume_short_ts=True)
    assert r._engines_of_all_ranks() == ["flow", "flow"] and r._default_chunks(None) == 4   # (both far under a million rows)
    monkeypatch.setenv("TRMC_ENGINE", "levels")
    assert r._engines_of_all_ranks() == ["levels", "levels"] and r._default_chunks(None) == 24
