"""A STREAM of windows (include/trmc.h trmc_stream_*, troute_amd.sequence.RouteStream): the reference's run-set loop
(nwm_routing/__main__.py:195-333: compute_nhd_routing_v02 per run set, new_q0 between them) as one sequence of tile launches
whose tile index runs on over the days -- against the same days routed one by one by plain ``upload`` / ``route`` calls on a plan
WITHOUT the cluster order (one launch per timestep for the narrow levels: another code path), the state taken through the host,
bit for bit; against the oracle; against the reference-Fortran golden of LowerColorado; and as two ranks of a job (two threads, one
device, the shared-memory transport) whose cut-edge hydrographs are exchanged once a day."""
import os
import threading

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O
from troute_amd import _lib, synthetic
from troute_amd.comm import Comm
from troute_amd.distributed import ShardedRouter
from troute_amd.plan import RoutingPlan, csr_from_lists
from troute_amd.sequence import RouteStream, pinned_like

pytestmark = pytest.mark.gpu
_serial = [0]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def days_of(net, n, seed=3):
    rng = np.random.default_rng(seed)
    return [rng.uniform(0, 0.6, net["qlat"].shape).astype(np.float32) for _ in range(n)]


def reference_days(net, days, q0, nsteps, qts, stride=None, full=False, options=None):
    """every day on ONE plain router (no cluster order), the state through the host"""
    r = ShardedRouter(net["to"], net["params"], assume_short_ts=True, options=options)
    hyds, states, fvds, state = [], [], [], q0
    for d in days:
        r.upload(nsteps, d, state)
        rows, hyd = r.route(qts, True)
        state = r.plan0.download_final_state()
        hyds.append(hyd)
        states.append(state)
        if stride or full:
            f = r.plan0.download_fvd()
            fvds.append(f[:, stride - 1::stride].copy() if stride else f.copy())
    r.close()
    return rows, hyds, states, fvds


@pytest.mark.parametrize("variant", ["slices+clusters", "clusters-only", "stride", "low-latency", "k4-small-clusters",
                                     "velocity-on-demand", "velocity-on-demand+stride", "velocity-on-demand+full",
                                     "tolerance", "tolerance+velocity-on-demand", "tolerance+velocity-on-demand+stride"])
def test_a_stream_of_days_on_one_gpu_equals_the_days_routed_one_by_one(variant):
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    nsteps, qts, ndays = 48, 16, 7
    q0 = np.random.default_rng(1).uniform(0, 1, (nseg, 3)).astype(np.float32)
    days = days_of(net, 4)
    seq_days = [days[w % 4] for w in range(ndays)]
    stride = 12 if variant in ("stride", "velocity-on-demand+stride", "tolerance+velocity-on-demand+stride") else None
    full = variant in ("slices+clusters", "k4-small-clusters", "velocity-on-demand+full")
    # (TRMC_ARITH_TOLERANCE: another arithmetic, the same for every kernel of a plan -- the stream's tiles against the one-step
    # launches of a plain plan in that arithmetic, bit for bit as well)
    tol = {"arithmetic": "tolerance"} if variant.startswith("tolerance") else None
    rows, want_h, want_s, want_f = reference_days(net, seq_days, q0, nsteps, qts, stride, full, options=tol)
    opts = {"wide_min_rows": 64, "wide_k": 8}
    if variant == "clusters-only":
        opts = {"wide_min_rows": -1, "wide_k": 16}
    if variant == "k4-small-clusters":
        opts = {"wide_min_rows": 500, "wide_k": 4, "cluster_rows": 24}
    if tol:
        opts = dict(opts, **tol)
    if "velocity-on-demand" in variant:
        # (trmc_plan_options.velocity_on_demand: a step's velocity is formed where it is handed on only -- the kept steps of the
        # stride, every step of a full result, nowhere for hydrographs and states; every product keeps its bits)
        opts = dict(opts, velocity_on_demand=1)
    r = ShardedRouter(net["to"], net["params"], stream=True, options=opts)
    lag, W, C = r.plan0.lags()
    assert C > 0 and (W > 0) == (variant != "clusters-only") and lag.max() == W + C - 1

    def late(it):                                       # a generator that yields late: nothing may depend on its pace
        import time
        for k, x in enumerate(it):
            if k in (2, 5):
                time.sleep(0.05)
            yield x
    got = {}
    with RouteStream(r, nsteps, qts, output_stride=stride, full_output=full, latency="low" if variant == "low-latency" else "throughput") as rs:
        # (two of the days from page-locked arrays, the others through the stream's own staging ring)
        feed = [pinned_like(d) if w % 2 else d for w, d in enumerate(seq_days)]
        for item in rs.route(late(feed), q0):
            got[item[0]] = tuple(None if x is None else np.array(x, copy=True) for x in item[1:])
        info = rs.plan.stream_info() if rs.info else None
        assert np.array_equal(rs.outlet_rows, rows)
    assert sorted(got) == list(range(ndays))
    for w in range(ndays):
        assert np.array_equal(bits(got[w][0]), bits(want_h[w])), w
        assert np.array_equal(bits(got[w][1]), bits(want_s[w])), w
        if stride or full:
            assert got[w][2].shape == want_f[w].shape and np.array_equal(bits(got[w][2]), bits(want_f[w])), w
    # the router is an ordinary one afterwards: a single window on the cluster-ordered plan (cluster tiles inside ONE window),
    # continuing from the state the stream left, equals the next day of the reference
    rows2, hyd2 = None, None
    r.upload(nsteps, seq_days[0], None)
    rows2, hyd2 = r.route(qts, True)
    ref = ShardedRouter(net["to"], net["params"], assume_short_ts=True, options=tol)
    ref.upload(nsteps, seq_days[0], want_s[-1])
    _, hyd3 = ref.route(qts, True)
    ref.close()
    assert np.array_equal(bits(hyd2), bits(hyd3))
    st = r.plan0.stats()
    assert st["main_launches"] < 60                     # (48 steps: no launch per timestep on this plan)
    r.close()


def test_a_second_stream_with_more_slots_on_the_same_plan():
    """The bench's order of things: a products-only stream, then -- on the same plan -- a stream that also hands every row's
    decimated (q, v, d) block over and therefore holds a slot more (the block leaves a day later, RouteStream): the per-slot
    buffers of the second are sized for ITS ring (the hydrograph slots once were not: a write past the allocation)."""
    net = synthetic.generate(nseg=20000, nnet=60, seed=12, nq=3)
    nseg = net["to"].shape[0]
    nsteps, qts, ndays = 48, 16, 9
    q0 = np.random.default_rng(2).uniform(0, 1, (nseg, 3)).astype(np.float32)
    days = days_of(net, 3, seed=8)
    seq_days = [days[w % 3] for w in range(ndays)]
    rows, want_h, want_s, want_f = reference_days(net, seq_days, q0, nsteps, qts, 12, False)
    r = ShardedRouter(net["to"], net["params"], stream=True, options={"wide_min_rows": 64, "wide_k": 8})
    slots = []
    for stride in (None, 12, None):
        got = {}
        with RouteStream(r, nsteps, qts, output_stride=stride) as rs:
            for item in rs.route(iter(seq_days), q0):
                got[item[0]] = tuple(None if x is None else np.array(x, copy=True) for x in item[1:])
            slots.append(rs.last_info["slots"])
        assert sorted(got) == list(range(ndays))
        for w in range(ndays):
            assert np.array_equal(bits(got[w][0]), bits(want_h[w])), (stride, w)
            assert np.array_equal(bits(got[w][1]), bits(want_s[w])), (stride, w)
            if stride:
                assert np.array_equal(bits(got[w][2]), bits(want_f[w])), w
    assert slots[1] == slots[0] + 1 and slots[2] == slots[0], slots
    r.close()


def test_stream_api_errors_and_bookkeeping():
    net = synthetic.generate(nseg=3000, nnet=9, seed=5, nq=3)
    nseg = net["to"].shape[0]
    up_ptr, up_idx = synthetic.upstream_csr(net["to"])
    q0 = np.zeros((nseg, 3), np.float32)
    day = pinned_like(np.full((nseg, 3), 0.1, np.float32))
    with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=True, engine="levels") as plain:
        plain.upload_forcing(32, day, q0)
        with pytest.raises(ValueError, match="cluster order"):
            plain.stream_begin(32, 16)
    with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=True, engine="levels", options={"cluster_rows": 128, "wide_k": 8}) as p:
        with pytest.raises(RuntimeError, match="trmc_upload_forcing"):
            p.stream_begin(32, 16)
        p.upload_forcing(32, day, q0)
        with pytest.raises(ValueError, match="multiple"):
            p.stream_begin(30, 16)
        with pytest.raises(RuntimeError, match="no stream"):
            p.stream_push(day)
        p.stream_begin(32, 16)
        info = p.stream_info()
        assert info["tiles_per_day"] == 4 and info["slots"] >= 2 + (info["lag_max"] + 1 + 3) // 4 and info["days_pushed"] == 0
        with pytest.raises(RuntimeError, match="in progress"):
            p.upload_forcing(32, day, q0) or p.stream_begin(32, 16)
        fin = _lib.result_empty((nseg, 3), np.float32, always_pinned=True)
        assert p.stream_push(day, q0=fin) == 0
        if info["lag_max"] > 0:
            with pytest.raises(RuntimeError, match="not been queued to its end"):
                p.stream_wait(0)
        with pytest.raises(ValueError, match="no such day"):
            p.stream_wait(3)
        with pytest.raises(ValueError, match="forcing columns"):
            p.stream_push(pinned_like(np.zeros((nseg, 2), np.float32)))
        p.stream_flush()
        p.stream_wait(0)
        assert p.stream_info()["days_complete"] == 1 and p.stream_info()["launches"] > 0
        first = fin.copy()
        p.stream_push(day, q0=fin)                      # a further day after a flush: continues where the first ended
        p.stream_end()
        # the same two days as two single windows
        p.upload_forcing(32, day, q0)
        p.route_device(32, 16, True)
        assert np.array_equal(bits(p.download_final_state()), bits(first))
        p.upload_forcing(32, day, None)
        p.route_device(32, 16, True)
        assert np.array_equal(bits(p.download_final_state()), bits(fin))


@pytest.mark.parametrize("precision", [32, 64])
def test_stream_against_the_oracle_day_by_day(precision):
    """random forest, three days with distinct forcing: every (q, v, d) of every row against the CPU restatement of the reference
    loop (oracle.network_by_segment, mc_reach.pyx:492-750), the state handed from day to day as new_q0 does"""
    rng = np.random.default_rng(77)
    nseg, nsteps, qts = 5000, 24, 8
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    from test_gpu_parity import synth_inputs
    params, qlat, q0 = synth_inputs(rng, nseg, 3)
    days = [qlat, (qlat * 0.5).astype(np.float32), (qlat * 1.7).astype(np.float32)]
    dt = np.float32 if precision == 32 else np.float64
    with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=True, engine="levels", precision=precision,
                     options={"cluster_rows": 64, "wide_min_rows": 200, "wide_k": 4}) as p:
        lvl, _ = p.levels()
        p.upload_forcing(nsteps, days[0].astype(dt), q0.astype(dt))
        p.stream_begin(nsteps, qts, full_output=True)
        D = p.stream_info()["slots"]
        outs = [_lib.result_empty((nseg, nsteps, 3), dt, always_pinned=True) for _ in range(D)]
        for d, q in enumerate(days):
            p.stream_push(pinned_like(q.astype(dt)), fvd=outs[d % D])
        p.stream_flush()
        state = q0
        for d, q in enumerate(days):
            p.stream_wait(d)
            want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params.astype(dt), state, q, True, det=precision == 32)[:, 1:, :]
            assert want.dtype == dt and np.array_equal(bits(outs[d % D]), bits(want)), d
            state = np.stack([want[:, -1, 0], want[:, -1, 0], want[:, -1, 2]], 1)
        p.stream_end()


def test_stream_lowercolorado_bit_identical_to_reference_golden():
    """the reference Fortran kernel driven through the restated loop (tests/golden/make_fixtures.py): 12 time slices of every
    segment and 100 probe hydrographs of the LowerColorado day -- routed as day 0 of a stream (7 cluster levels of 128 rows for
    its 649 levels of segments)"""
    lc = H.LowerColorado()
    up_ptr, up_idx = lc.csr()
    g = lc.golden()
    with RoutingPlan(up_ptr, up_idx, lc.params9, assume_short_ts=True, engine="levels", options={"cluster_rows": 128}) as p:
        lag, W, C = p.lags()
        assert W == 0 and C == 7
        p.upload_forcing(lc.nts, lc.qlat, lc.q0)
        p.stream_begin(lc.nts, lc.qts, full_output=True)
        out = _lib.result_empty((lc.nseg, lc.nts, 3), np.float32, always_pinned=True)
        p.stream_push(pinned_like(lc.qlat), fvd=out)
        p.stream_flush()
        p.stream_wait(0)
        info = p.stream_info()
        assert info["launches"] == lc.nts // 16 + C - 1            # 24 launches for the day instead of 288
        p.stream_end()
    assert np.array_equal(bits(out[:, g["tsel"] - 1, :]), bits(g["shortts_f32_tsel"]))
    assert np.array_equal(bits(out[g["probes"]]), bits(g["shortts_f32_probes"][:, 1:, :]))


@pytest.mark.parametrize("stride", [None, 8])
def test_a_stream_on_two_ranks_equals_the_days_routed_one_by_one(stride):
    """Two ranks (threads) on one device over the shared-memory transport: every rank streams its sub-basins, the trunk rides in
    its owner's stream behind them, the cut-edge hydrographs are all-gathered once a day; rank 0 gets every day's outlet
    hydrographs of the WHOLE network, every rank the final state of its rows -- and, with an output stride, every n-th step of its
    rows' (q, v, d) a day later (the ring then holds a slot more on every rank)."""
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    nsteps, qts, ndays = 32, 16, 7
    q0 = np.random.default_rng(2).uniform(0, 1, (nseg, 3)).astype(np.float32)
    days = days_of(net, 3, seed=5)
    seq_days = [days[w % 3] for w in range(ndays)]
    rows1, want_h, want_s, want_f = reference_days(net, seq_days, q0, nsteps, qts, stride)
    world = 2
    _serial[0] += 1
    key = f"stream{os.getpid()}_{_serial[0]}"
    results, errors = [None] * world, []

    def run(rank):
        try:
            comm = Comm(rank, world, device=0, backend="shm", key=key)
            r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0, stream=True, options={"wide_min_rows": 64, "wide_k": 8})
            r.enable_device_exchange(comm)
            got = {}
            with RouteStream(r, nsteps, qts, output_stride=stride) as rs:
                for item in rs.route(seq_days, q0):
                    day, hyd, fin = item[:3]
                    got[day] = (None if hyd is None else np.array(hyd, copy=True), np.array(fin[0], copy=True),
                                np.array(item[3][0], copy=True) if stride else None)
                out_rows = np.array(rs.outlet_rows, copy=True)
                srows = np.array(rs.rows, copy=True)
            routed = np.ones(srows.shape[0], bool)          # (not the boundary copies of the cut rows: flow only)
            if r.plan1 is not None:
                routed[r.rows0.shape[0]:] = ~r.boundary1
            results[rank] = (out_rows, got, srows, r.plan1 is not None, routed, r._planS_lag)
            r.close()
            comm.close()
        except Exception as e:                          # pragma: no cover
            import traceback
            traceback.print_exc()
            errors.append(e)
    ts = [threading.Thread(target=run, args=(k,)) for k in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert results[0][3] or results[1][3]               # one of them owns a trunk
    rows, got0 = results[0][0], results[0][1]
    assert np.array_equal(rows, rows1) and sorted(got0) == list(range(ndays))
    for w in range(ndays):
        assert np.array_equal(bits(got0[w][0]), bits(want_h[w])), w
        assert results[1][1][w][0] is None
    for rank in range(world):
        _, got, srows, _, routed, _ = results[rank]
        for w in range(ndays):
            state = got[w][1]
            assert state.shape == (srows.shape[0], 3)
            assert np.array_equal(bits(state[routed][:, [0, 2]]), bits(want_s[w][srows[routed]][:, [0, 2]])), (rank, w)
            assert np.array_equal(bits(state[:, 0]), bits(want_s[w][srows][:, 0])), (rank, w)
            if stride:
                blk = got[w][2]
                assert blk.shape == (srows.shape[0], nsteps // stride, 3)
                assert np.array_equal(bits(blk[routed]), bits(want_f[w][srows[routed]])), (rank, w)
