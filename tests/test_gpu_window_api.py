"""The routing window in parts (trmc_route_begin / _advance / _end), row sets, ranged gather / boundary feed and
time-skewed rows (trmc_plan_set_lag) on one GPU: equal to the one-call route bit for bit, call-order errors
reported as RuntimeError / ValueError instead of undefined behaviour."""
import numpy as np
import pytest

import helpers as H
from helpers import flow_engine
from troute_amd import _lib
from troute_amd.comm import DeviceBuffer
from troute_amd.plan import RoutingPlan, csr_from_lists

pytestmark = pytest.mark.gpu


def small_forest(seed=2, nseg=3000):
    rng = np.random.default_rng(seed)
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    p = np.stack([np.full(nseg, 300.0), rng.uniform(300, 3000, nseg), rng.uniform(1, 9, nseg), np.zeros(nseg), np.zeros(nseg),
                  np.full(nseg, 0.06), np.full(nseg, 0.12), rng.uniform(0.2, 1.5, nseg), rng.uniform(1e-3, 2e-2, nseg)], 1)
    p[:, 3] = p[:, 2] * 5 / 3
    p[:, 4] = 3 * p[:, 3]
    qlat = rng.uniform(0, 0.4, (nseg, 4)).astype(np.float32)
    q0 = rng.uniform(0, 1, (nseg, 3)).astype(np.float32)
    return to, ups, up_ptr, up_idx, p.astype(np.float32), qlat, q0


@pytest.mark.parametrize("short", [True, False])
def test_window_in_parts_equals_one_call(short):
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest()
    nsteps, qts = 40, 12
    with RoutingPlan(up_ptr, up_idx, p) as plan:
        want = plan.route(nsteps, qts, short, qlat, q0)
        plan.upload_forcing(nsteps, qlat, q0)
        plan.route_begin(nsteps, qts, short)
        rows = np.array([5, 17, 2999, 0], np.int64)
        rs = plan.rowset(rows)
        buf = DeviceBuffer(0, 4 * nsteps * 4)               # [4][nsteps] float32 in HBM (the library's own allocator)
        done = 0
        for t_end in (1, 1, 7, 23, 40):                      # uneven chunks, one empty
            plan.route_advance(t_end)
            plan.gather_flow_range(rs, done, t_end, buf.ptr + done * 4, nsteps)
            done = t_end
        st = plan.route_end()
        assert st["nsteps"] == nsteps and st["main_launches"] >= (4 if flow_engine() else nsteps)
        got = plan.download_fvd()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(buf.download((4, nsteps), np.float32).view(np.uint32), want[rows, :, 0].view(np.uint32))
        buf.free()


@pytest.mark.parametrize("nseg,nsteps,qts,K,K2,wide_rows,mid_rows,mid_levels", [
    (6000, 48, 12, 8, 4, 64, 8, 12), (6000, 50, 7, 16, 3, 128, 16, 32), (6000, 37, 5, 5, 5, 64, 4, 3), (6000, 9, 4, 2, 1, 32, 8, 32),
    (900, 33, 12, 8, 2, 16, 2, 12), (20000, 64, 16, 16, 4, 256, 32, 6), (150, 16, 4, 4, 2, 4, 1, 32)])
def test_two_tiers_of_tiles_equal_the_one_step_launches(nseg, nsteps, qts, K, K2, wide_rows, mid_rows, mid_levels):
    """The level engine's second tier (trmc_plan_options.mid_*): the levels right below the wide ones routed K2 steps per
    launch under a skew of their own, on the plan's stream between the tail's launches -- against the one-step launches of the
    same network (which the other tests pin to the oracle and the reference goldens): window lengths that are no multiple of
    K or K2, forcing columns that change inside a tile, K2 = 1 .. K, a tail or none, a window in parts, several windows on one
    plan, a warm start, cost collection, the asynchronous fetch and a state handed to a clone."""
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(seed=nseg + K, nseg=nseg)
    qlat = np.ascontiguousarray(np.tile(qlat, (1, (nsteps + qts - 1) // qts // qlat.shape[1] + 1)))
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels", options={"wide_min_rows": -1}) as ref:
        want = ref.route(nsteps, qts, True, qlat, q0)
        assert ref.stats()["wide_levels"] == 0
        ref.upload_forcing(nsteps, qlat, None)
        ref.route_device(nsteps, qts, True)
        want2, state2 = ref.download_fvd(), ref.download_final_state()
    opts = {"wide_min_rows": wide_rows, "wide_k": K, "mid_min_rows": mid_rows, "mid_k": K2, "mid_levels": mid_levels}
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels", options=opts) as plan:
        for _ in range(2):
            got = plan.route(nsteps, qts, True, qlat, q0)
            st = plan.stats()
            assert st["wide_levels"] >= 1 and st["wide_k"] == K and 1 <= st["mid_levels"] <= mid_levels and st["mid_k"] == min(K, K2)
            assert st["mid_launches"] == -(-nsteps // st["mid_k"]) + st["mid_levels"] - 1
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        iters = plan.download_iterations()
        plan.upload_forcing(nsteps, qlat, None)                 # the next window from the resident state
        plan.route_device(nsteps, qts, True)
        assert np.array_equal(plan.download_fvd().view(np.uint32), want2.view(np.uint32))
        assert np.array_equal(plan.download_final_state().view(np.uint32), state2.view(np.uint32))
        plan.upload_forcing(nsteps, qlat, q0)                   # a window that arrives in parts
        plan.route_begin(nsteps, qts, True)
        plan.route_advance(nsteps // 3)
        plan.route_advance(nsteps // 3)                         # (nothing new)
        plan.route_advance(nsteps - 1)
        plan.route_advance(nsteps)
        plan.route_end()
        assert np.array_equal(plan.download_fvd().view(np.uint32), want.view(np.uint32))
        assert np.array_equal(plan.download_iterations(), iters)
        plan.collect_cost(True)
        assert np.array_equal(plan.route(nsteps, qts, True, qlat, q0).view(np.uint32), want.view(np.uint32))
        cost, n = plan.download_cost()
        assert n == nsteps and cost.max() > 0
        plan.collect_cost(False)
        # the state handed to a clone on the device, both plans in sequence mode; products fetched with the window
        clone = plan.clone()
        for pl in (plan, clone):
            pl.set_sequence_mode(True)
        outlets = np.flatnonzero(to < 0)
        rs = clone.rowset(outlets)
        pinned = _lib.result_empty(qlat.shape, np.float32, always_pinned=True)
        pinned[...] = qlat
        plan.upload_forcing(nsteps, qlat, q0)
        plan.route_begin(nsteps, qts, True)
        plan.route_advance(nsteps)
        clone.stage_forcing(nsteps, pinned)
        clone.chain_from(plan)
        clone.route_begin(nsteps, qts, True)
        clone.route_advance(nsteps)
        clone.fetch_begin(rs, True)
        plan.route_end()
        clone.route_end()
        hyd, state = clone.fetch_wait()
        assert np.array_equal(state.view(np.uint32), state2.view(np.uint32))
        assert np.array_equal(hyd.view(np.uint32), np.ascontiguousarray(want2[outlets, :, 0]).view(np.uint32))
        clone.close()


def test_call_order_errors():
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=200)
    with RoutingPlan(up_ptr, up_idx, p) as plan:
        with pytest.raises(RuntimeError, match="trmc_upload_forcing must precede"):
            plan.route_begin(12, 12, True)
        plan.upload_forcing(12, qlat, q0)
        with pytest.raises(RuntimeError, match="trmc_route_begin must precede"):
            plan.route_advance(3)
        with pytest.raises(RuntimeError, match="no routing window"):
            plan.route_end()
        plan.route_begin(12, 12, True)
        with pytest.raises(RuntimeError, match="already in progress"):
            plan.route_begin(12, 12, True)
        with pytest.raises(RuntimeError, match="in progress"):
            plan.upload_forcing(12, qlat, q0)
        with pytest.raises(ValueError, match="t_end outside"):
            plan.route_advance(13)
        plan.route_advance(5)
        with pytest.raises(ValueError, match="t_end outside"):
            plan.route_advance(4)
        rs = plan.rowset(np.array([0], np.int64))
        with pytest.raises(RuntimeError, match="not all routed"):
            plan.gather_flow_range(rs, 0, 6, 1, 6)
        with pytest.raises(ValueError, match="unknown row set"):
            plan.gather_flow_range(99, 0, 5, 1, 5)
        with pytest.raises(RuntimeError, match="window abandoned"):
            plan.route_end()                                  # not every step queued: abandoned, plan reusable
        with pytest.raises(RuntimeError, match="nothing routed"):
            plan.download_fvd()
        plan.route_device(12, 12, True)                       # and it is
        assert np.isfinite(plan.download_fvd()).all()
        with pytest.raises(ValueError, match="row out of range"):
            plan.rowset(np.array([200], np.int64))


def test_time_skewed_rows_equal_two_phase_routing():
    """A chain 0..9 feeding (through a cut) a chain 10..19: routing both in one plan with the lower chain lagged
    and its boundary values supplied chunk by chunk equals routing them one after the other."""
    n = 10
    rng = np.random.default_rng(7)
    p = np.stack([np.full(2 * n, 300.0), rng.uniform(300, 3000, 2 * n), rng.uniform(1, 9, 2 * n), np.zeros(2 * n),
                  np.zeros(2 * n), np.full(2 * n, 0.06), np.full(2 * n, 0.12), rng.uniform(0.2, 1.5, 2 * n),
                  rng.uniform(1e-3, 2e-2, 2 * n)], 1)
    p[:, 3] = p[:, 2] * 5 / 3
    p[:, 4] = 3 * p[:, 3]
    p = p.astype(np.float32)
    qlat = rng.uniform(0, 0.4, (2 * n, 3)).astype(np.float32)
    q0 = rng.uniform(0, 1, (2 * n, 3)).astype(np.float32)
    nsteps, qts, K = 30, 12, 4
    # reference: the whole chain in one ordinary plan
    ups = [[]] + [[i - 1] for i in range(1, 2 * n)]
    with RoutingPlan(*csr_from_lists(ups), p) as plan:
        want = plan.route(nsteps, qts, True, qlat, q0)
    # merged table: rows 0..9 upper chain, row 10 = boundary copy of row 9, rows 11..20 = lower chain (lag 2K)
    ups_m = [[]] + [[i - 1] for i in range(1, n)] + [[]] + [[n]] + [[i - 1] for i in range(n + 2, 2 * n + 1)]
    sel = list(range(n)) + [n - 1] + list(range(n, 2 * n))
    boundary = np.zeros(2 * n + 1, np.uint8)
    boundary[n] = 1
    lag = np.zeros(2 * n + 1, np.int32)
    lag[n + 1:] = 2 * K
    with RoutingPlan(*csr_from_lists(ups_m), p[sel], boundary) as plan:
        with pytest.raises(ValueError, match="0 or one common value"):
            plan.set_lag(np.arange(2 * n + 1, dtype=np.int32))
        bad = lag.copy()
        bad[:n] = 2 * K
        bad[n + 1:] = 0
        with pytest.raises(ValueError, match="lagged row feeds|boundary rows may only feed"):
            plan.set_lag(bad)
        plan.set_lag(lag)
        plan.upload_forcing(nsteps, qlat[sel], q0[sel], None)
        with pytest.raises(ValueError, match="assume_short_ts only"):
            plan.route_begin(nsteps, qts, False)
        plan.route_begin(nsteps, qts, True)
        rs = plan.rowset(np.array([n - 1], np.int64))
        last = nsteps + 2 * K
        C_ = -(-nsteps // K)
        bufs = []
        c = 0
        with pytest.raises(RuntimeError, match="boundary hydrographs are staged through"):
            plan.route_advance(last)
        while True:
            d_end = min((c + 1) * K, last)
            plan.route_advance(d_end)
            if c < C_:
                tb, te = c * K, min(nsteps, (c + 1) * K)
                b = DeviceBuffer(0, (te - tb) * 4)
                plan.gather_flow_range(rs, tb, te, b.ptr, te - tb)
                plan.set_boundary_flow_range(tb, te, b.ptr, te - tb)
                bufs.append(b)
            if d_end >= last:
                break
            c += 1
        st = plan.route_end()
        assert st["main_launches"] == (c + 1 if flow_engine() else last)   # one launch per advance / per timestep
        got = plan.download_fvd()
    assert np.array_equal(got[:n].view(np.uint32), want[:n].view(np.uint32))
    assert np.array_equal(got[n + 1:].view(np.uint32), want[n:].view(np.uint32))


def test_packed_forcing_argument_errors():
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=50)
    lib = _lib.lib()
    with RoutingPlan(up_ptr, up_idx, p) as plan:
        raw = np.zeros((2, 60), np.int32)
        pk = np.array([1e-5, 0, np.nan, np.nan, np.nan, np.nan])
        feat = np.arange(50, dtype=np.int64)
        feat[3] = 60                                          # outside the feature axis
        rc = lib.trmc_upload_forcing_packed(plan._h, 12, 2, 60, _lib.ptr(raw), None, _lib.ptr(pk), None, _lib.ptr(feat),
                                            _lib.ptr(q0), None)
        assert rc == _lib.TRMC_EINVAL and b"feature axis" in lib.trmc_last_error()
        rc = lib.trmc_upload_forcing_packed(plan._h, 12, 2, 60, _lib.ptr(raw), _lib.ptr(raw), _lib.ptr(pk), None,
                                            _lib.ptr(np.arange(50, dtype=np.int64)), _lib.ptr(q0), None)
        assert rc == _lib.TRMC_EINVAL and b"pack_b" in lib.trmc_last_error()


def test_large_results_come_in_pooled_page_locked_arrays(monkeypatch):
    """download_fvd() hands out an array backed by page-locked memory when the result is large (trmc_host_alloc); the
    buffer returns to a pool when the array is dropped and is the same memory the next window's result arrives in;
    TRMC_PINNED_RESULTS=0 gives ordinary arrays; the values are the same either way."""
    import gc
    from troute_amd import _lib, synthetic
    from troute_amd.distributed import ShardedRouter
    net = synthetic.generate(nseg=60000, nnet=150, seed=5, nq=3)
    n = net["to"].shape[0]
    r = ShardedRouter(net["to"], net["params"])
    r.upload(24, net["qlat"], np.zeros((n, 3), np.float32))
    r.route_resident(12, True)
    assert n * 24 * 3 * 4 >= _lib._PINNED_MIN
    _lib.pinned_pool_clear()
    a = r.plan0.download_fvd()
    addr = a.ctypes.data
    assert not a.flags["OWNDATA"] and a.flags["WRITEABLE"] and a.shape == (n, 24, 3)
    keep = a.copy()
    view = a[5:9]                     # a view keeps the buffer alive
    del a
    gc.collect()
    assert not _lib._pinned_free.get(keep.nbytes)
    del view
    gc.collect()
    assert _lib._pinned_free.get(keep.nbytes) == [addr]
    b = r.plan0.download_fvd()        # the pooled buffer again
    assert b.ctypes.data == addr and np.array_equal(b.view(np.uint32), keep.view(np.uint32))
    monkeypatch.setenv("TRMC_PINNED_RESULTS", "0")
    c = r.plan0.download_fvd()
    assert c.flags["OWNDATA"] and np.array_equal(c.view(np.uint32), keep.view(np.uint32))
    del b
    gc.collect()
    _lib.pinned_pool_clear()
    assert not _lib._pinned_free.get(keep.nbytes)
    r.close()


def test_asynchronous_fetch_equals_the_synchronous_downloads():
    """trmc_fetch_begin / trmc_fetch_wait (outlet hydrographs of a row set + final state, copied on a copy stream beside the
    NEXT window) hand over exactly what gather_flow_rows / download_final_state return, window after window, and the
    arrays a wait returned stay intact while the next windows run (a ring of three)."""
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=5000)
    nsteps, qts = 48, 12
    rows = np.array([0, 7, 4999, 2500, 31], np.int64)
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as plan:
        rs = plan.rowset(rows)
        plan.upload_forcing(nsteps, qlat, q0)
        kept = []
        for k in range(4):
            plan.route_device(nsteps, qts, True)
            want_h, want_s = plan.gather_flow_rows(rows), plan.download_final_state()
            prev = plan.fetch_wait()                       # (nothing in flight the first time)
            assert (prev[0] is None) == (k == 0)
            plan.fetch_begin(rs, True)
            kept.append((want_h.copy(), want_s.copy()))
            plan.upload_forcing(nsteps, qlat * np.float32(1.0 + 0.1 * (k + 1)), None)   # next window: warm start, other forcing
            if k > 0:
                assert np.array_equal(prev[0].view(np.uint32), kept[k - 1][0].view(np.uint32))
                assert np.array_equal(prev[1].view(np.uint32), kept[k - 1][1].view(np.uint32))
        last = plan.fetch_wait()
        assert np.array_equal(last[0].view(np.uint32), kept[-1][0].view(np.uint32))
        assert np.array_equal(last[1].view(np.uint32), kept[-1][1].view(np.uint32))
        plan.route_device(nsteps, qts, True)
        with pytest.raises(RuntimeError, match="fetch is in flight"):
            plan.fetch_begin(rs, True)
            plan.fetch_begin(rs, True)
        plan.fetch_wait()


@pytest.mark.parametrize("engine", [None, "levels"])
def test_asynchronous_fetch_of_the_decimated_result_equals_slicing(monkeypatch, engine):
    """trmc_fetch_begin_fvd: every n-th step of (q, v, d) of every row joins the products of a window -- decimated on the copy
    stream, copied beside the next window, which must not overwrite the result before it has been read (the next window's
    set-up waits for the decimation).  Both engines, strides that do and do not divide the window, stride 1 (the whole array
    straight from the result buffer), and the plain fetch in between."""
    if engine == "levels":
        monkeypatch.setenv("TRMC_ENGINE", "levels")
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
        monkeypatch.setenv("TRMC_WIDE_K", "8")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=6000)
    nsteps, qts = 48, 12
    rows = np.array([3, 5999, 17], np.int64)
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as plan:
        rs = plan.rowset(rows)
        plan.upload_forcing(nsteps, qlat, q0)
        want = []
        for k, stride in enumerate([12, 5, 1, None, 12, 100]):
            plan.route_device(nsteps, qts, True)
            full = plan.download_fvd()
            want.append((stride, plan.gather_flow_rows(rows), plan.download_final_state(),
                         None if stride is None else full[:, stride - 1::stride][:, :nsteps // stride].copy()))
            prev = plan.fetch_wait()
            plan.fetch_begin(rs, True, stride)
            # the next window is queued while the copy is (possibly) still running: other forcing, warm start
            plan.upload_forcing(nsteps, qlat * np.float32(1.0 + 0.2 * (k + 1)), None)
            if k > 0:
                s0, h0, f0, d0 = want[k - 1]
                assert len(prev) == (2 if s0 is None else 3)
                assert np.array_equal(prev[0].view(np.uint32), h0.view(np.uint32))
                assert np.array_equal(prev[1].view(np.uint32), f0.view(np.uint32))
                if s0 is not None:
                    assert prev[2].shape == d0.shape and np.array_equal(prev[2].view(np.uint32), d0.view(np.uint32)), s0
        plan.route_device(nsteps, qts, True)                 # a window AFTER the last fetch was begun, before it is waited for
        last = plan.fetch_wait()
        assert last[2].shape == (6000, 0, 3)
        with pytest.raises(ValueError, match="output_stride"):
            plan.fetch_begin(rs, True, 0)


@pytest.mark.parametrize("mid", [False, True])
def test_windows_that_decimate_as_they_go_hand_over_the_same_block(monkeypatch, mid):
    """trmc_plan_set_output_stride: the tiles write the kept steps of their rows aside as they route them, the other rows are
    gathered from the time-major planes at the fetch -- the block must equal the slices of the full result, for strides that
    do and do not divide the window or the tiles' K, with a second tier of tiles, after a change of the stride, and for a
    fetch that asks for another stride than the plan was told (decimated from the result as before)."""
    monkeypatch.setenv("TRMC_ENGINE", "levels")
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    monkeypatch.setenv("TRMC_MID_MIN_ROWS", "8" if mid else "0")
    monkeypatch.setenv("TRMC_MID_K", "2")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=6000)
    nsteps, qts = 48, 12
    rows = np.array([3, 5999, 17], np.int64)
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as plan:
        rs = plan.rowset(rows)
        plan.upload_forcing(nsteps, qlat, q0)
        want = []
        for k, (told, asked) in enumerate([(12, 12), (5, 5), (3, 3), (1, 1), (12, 4), (100, 100), (7, 7), (0, 6)]):
            plan.set_output_stride(told)
            plan.route_device(nsteps, qts, True)
            if k == 0:
                st = plan.stats()
                assert st["wide_levels"] > 0 and (st["mid_levels"] > 0) == mid
            full = plan.download_fvd()
            want.append(full[:, asked - 1::asked][:, :nsteps // asked].copy())
            prev = plan.fetch_wait()
            plan.fetch_begin(rs, True, asked)
            plan.upload_forcing(nsteps, qlat * np.float32(1.0 + 0.2 * (k + 1)), None)   # (the next window is queued beside the copy)
            if k > 0:
                assert prev[2].shape == want[k - 1].shape and np.array_equal(prev[2].view(np.uint32), want[k - 1].view(np.uint32)), k - 1
        last = plan.fetch_wait()
        assert np.array_equal(last[2].view(np.uint32), want[-1].view(np.uint32))
        with pytest.raises(ValueError, match="stride"):
            plan.set_output_stride(-1)


@pytest.mark.parametrize("hinted", [False, True])
def test_hot_rows_are_routed_by_blocks_of_their_own_and_change_nothing(monkeypatch, hinted):
    """trmc_plan_options.hot_rows: rows that end a tile in three or more secant iterations (or over bank) are taken out of their
    blocks for the next tile.  A forcing that floods part of the network makes sure there are some -- more than the list holds
    (the rest stay where they are) -- over several windows, and the result must be the one without the option, bit for bit; so
    must the cost every row collected (trmc_plan_collect_cost): a row routed twice in a launch would have counted twice."""
    monkeypatch.setenv("TRMC_ENGINE", "levels")
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
    monkeypatch.setenv("TRMC_WIDE_K", "4")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=6000)
    rng = np.random.default_rng(5)
    qlat = qlat.copy()
    qlat[rng.uniform(0, 1, qlat.shape[0]) < 0.08] *= np.float32(400.0)          # (over bank, many iterations)
    nsteps, qts = 48, 12
    out = {}
    for hot in (0, 1):
        monkeypatch.setenv("TRMC_HOT_ROWS", str(hot))
        hint = (rng.integers(0, 4, to.shape[0]).astype(np.uint8) if hinted else None)
        with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, cost_hint=hint) as plan:
            res = []
            plan.collect_cost(True)
            plan.upload_forcing(nsteps, qlat, q0)
            for k in range(3):
                plan.route_device(nsteps, qts, True)
                res.append(plan.download_fvd())
                res.append(plan.download_cost()[0])
                plan.upload_forcing(nsteps, qlat * np.float32(1.0 + 0.5 * (k + 1)), None)
            assert plan.stats()["wide_levels"] > 0
            n = plan.hot_rows()
            assert (n > 200) == bool(hot), n
            out[hot] = res
    # (... and every row is routed ONCE per tile, by its block or by the list's: the cost sums -- one addition per routing of a
    # row -- are the same numbers either way)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a.view(np.uint32 if a.dtype == np.float32 else a.dtype), b.view(np.uint32 if b.dtype == np.float32 else b.dtype))


def test_device_clock_stamps_of_consecutive_windows(monkeypatch):
    """trmc_plan_set_stamps: every window leaves the device clock at four points (tiles begin / end, tail begins / ends) in a
    page-locked ring -- in order inside a window, and the windows one after the other."""
    monkeypatch.setenv("TRMC_ENGINE", "levels")
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(nseg=6000)
    nsteps, qts = 48, 12
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as plan:
        ring = plan.set_stamps(4)
        plan.upload_forcing(nsteps, qlat, q0)
        for _ in range(3):
            plan.route_device(nsteps, qts, True)
        assert plan.stats()["wide_levels"] > 0
        r = ring[:3].astype(np.int64)
        assert (r > 0).all() and (ring[3] == 0).all()
        assert (r[:, 0] < r[:, 1]).all() and (r[:, 2] < r[:, 3]).all() and (r[:, 0] <= r[:, 2]).all()
        assert (r[1:, 0] > r[:-1, 3]).all()                       # (one plan: a window begins after the one before has ended)
        plan.set_stamps(0)
        before = ring.copy()
        plan.route_device(nsteps, qts, True)
        assert np.array_equal(ring, before)


def test_rccl_communicator_single_rank_and_device_plumbing():
    """The RCCL transport of the package's communicator (librccl.so by dlopen: ncclGetUniqueId / ncclCommInitRank /
    ncclAllGather through include/trmc.h) at world size 1 -- what one GPU can run of it -- plus the device buffers, streams,
    events and the indexed row gather of the hand-off."""
    from troute_amd import comm as X
    c = X.Comm(0, 1, device=0, backend="rccl", key=f"t{np.random.default_rng().integers(1 << 30)}")
    assert c.backend == "rccl"
    src = np.arange(6 * 40, dtype=np.float32).reshape(6, 40)
    a = X.DeviceBuffer.from_array(0, src)
    b = X.DeviceBuffer(0, src.nbytes)
    st = X.stream_create(0)
    ev = X.event_create(0)
    c.all_gather(a.ptr, b.ptr, src.nbytes, st)
    X.event_record(0, ev, st)
    st2 = X.stream_create(0)
    X.stream_wait_event(0, st2, ev)
    idx = X.DeviceBuffer.from_array(0, np.array([5, 0, 3], np.int64))
    d = X.DeviceBuffer(0, 3 * 40 * 4)
    X.gather_rows(0, b.ptr, idx.ptr, 3, 40 * 4, d.ptr, st2)
    got = d.download((3, 40), np.float32, st2)
    assert np.array_equal(got, src[[5, 0, 3]])
    assert np.array_equal(c.all_gather_host(np.array([1.5, 2.5]))[0], [1.5, 2.5])
    assert c.all_reduce_max_host(np.array([3, 1], np.uint8)).tolist() == [3, 1]
    c.barrier()
    for x in (a, b, idx, d):
        x.free()
    X.event_destroy(0, ev)
    X.stream_destroy(0, st)
    X.stream_destroy(0, st2)
    c.close()


def test_two_plans_taking_turns_on_the_device_give_each_its_own_window(monkeypatch):
    """TRMC_SETUP_ASIDE=1 (sequence mode, trmc_plan_options.sequence_mode): a window's set-up goes to the tile stream and its end is queued with its
    last launch, so that two plans of the level engine can take turns on one device -- one's window queued while the other's
    is running, in the shared hardware queues -- without one waiting for the other's tail.  Each plan must come out with
    exactly what it routes alone, whatever the interleaving."""
    monkeypatch.setenv("TRMC_SETUP_ASIDE", "1")
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(seed=5, nseg=6000)
    rng = np.random.default_rng(9)
    qlat_b = rng.uniform(0, 0.7, qlat.shape).astype(np.float32)        # the other member's forcing
    nsteps, qts = 48, 12
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels") as ref:
        want_a = ref.route(nsteps, qts, True, qlat, q0)
        want_b = ref.route(nsteps, qts, True, qlat_b, q0)
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels") as a, \
            RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels") as b:
        a.upload_forcing(nsteps, qlat, q0)
        b.upload_forcing(nsteps, qlat_b, q0)

        def queue(pl):
            pl.route_begin(nsteps, qts, True)
            pl.route_advance(nsteps)
        for pl in (a, b):                       # a first window each: the tile stream exists from the second on
            queue(pl)
            assert pl.route_end()["wide_levels"] > 0
        queue(a)
        for _ in range(3):
            queue(b)
            a.route_end()
            got_a = a.download_fvd()
            queue(a)
            b.route_end()
            got_b = b.download_fvd()
            assert np.array_equal(got_a.view(np.uint32), want_a.view(np.uint32))
            assert np.array_equal(got_b.view(np.uint32), want_b.view(np.uint32))
        a.route_end()
        assert np.array_equal(a.download_fvd().view(np.uint32), want_a.view(np.uint32))


@pytest.mark.parametrize("aside", ["1", None])
def test_a_sequence_of_windows_on_two_plans_taking_turns_equals_the_sequence_on_one(monkeypatch, aside):
    """trmc_plan_chain_from: consecutive windows of ONE sequence alternate between two plans of the same network, each
    starting from the state the other's window leaves -- handed over on the device, the wide levels' rows behind the other
    plan's last tile, the rest behind its tail -- while the other plan's window may still be running.  The same bits as the
    sequence routed window after window on one plan (trmc_upload_forcing with q0 = NULL).  The API does not require
    TRMC_SETUP_ASIDE: without it the receiver's set-up and tail run on its own stream, which must wait for BOTH copies."""
    if aside:
        monkeypatch.setenv("TRMC_SETUP_ASIDE", aside)
    else:
        monkeypatch.delenv("TRMC_SETUP_ASIDE", raising=False)
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(seed=7, nseg=6000)
    nsteps, qts, nwin = 48, 12, 6
    want = []
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels") as ref:
        ref.upload_forcing(nsteps, qlat, q0)
        for w in range(nwin):
            ref.route_device(nsteps, qts, True)
            want.append(ref.download_fvd().copy())
            ref.upload_forcing(nsteps, qlat, None)          # the next window continues from this one's final state
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels") as a, \
            RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True, engine="levels") as b:
        a.upload_forcing(nsteps, qlat, q0)
        b.upload_forcing(nsteps, qlat, q0)                  # (its initial state is replaced by the hand-over)

        def queue(pl):
            pl.route_begin(nsteps, qts, True)
            pl.route_advance(nsteps)
        plans = [a, b]
        queue(a)                                            # window 0 on a
        for w in range(1, nwin):
            cur, prev = plans[w % 2], plans[(w - 1) % 2]
            cur.chain_from(prev)                            # window w starts where window w - 1 ends ...
            queue(cur)                                      # ... and is queued while that one may still be running
            prev.route_end()
            got = prev.download_fvd()
            assert np.array_equal(got.view(np.uint32), want[w - 1].view(np.uint32)), w - 1
        last = plans[(nwin - 1) % 2]
        last.route_end()
        assert np.array_equal(last.download_fvd().view(np.uint32), want[nwin - 1].view(np.uint32))
        with pytest.raises(ValueError, match="two different plans"):
            a.chain_from(a)


@pytest.mark.parametrize("engine,ahead", [("levels", False), ("levels", True), ("flow", False)])
def test_a_sequence_of_distinct_days_on_a_plan_and_its_clone_with_staged_forcing(monkeypatch, engine, ahead):
    """The sequence bench.py times: consecutive windows with DIFFERENT forcing take turns on a plan and its clone
    (trmc_plan_clone: the original's topology and parameter columns in HBM, its own window buffers), every day's forcing
    staged from page-locked host memory on the copy stream without a wait (trmc_stage_forcing), the state handed on in HBM
    (trmc_plan_chain_from), outlet hydrographs and final state fetched asynchronously -- against the same days routed one
    after the other on one plan with synchronous uploads (warm start q0 = None: AbstractNetwork.py:177-191)."""
    monkeypatch.setenv("TRMC_SETUP_ASIDE", "1")
    monkeypatch.setenv("TRMC_ENGINE", engine)
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    to, ups, up_ptr, up_idx, p, qlat, q0 = small_forest(seed=11, nseg=6000)
    nsteps, qts, ndays = 48, 12, 7
    rng = np.random.default_rng(3)
    days = []
    for _ in range(ndays):                                   # page-locked, distinct
        d = _lib.result_empty(qlat.shape, np.float32, always_pinned=True)
        d[...] = rng.uniform(0, 0.6, qlat.shape).astype(np.float32)
        days.append(d)
    outlets = np.flatnonzero(to < 0)
    want_h, want_s = [], []
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as ref:
        for w in range(ndays):
            ref.upload_forcing(nsteps, days[w], q0 if w == 0 else None)
            ref.route_device(nsteps, qts, True)
            want_h.append(ref.gather_flow_rows(outlets))
            want_s.append(ref.download_final_state())
    with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as a:
        b = a.clone()
        if engine == "flow":
            with pytest.raises(ValueError, match="level engine"):
                b.chain_from(a)
            b.close()
            return
        plans = [a, b]
        rs = [pl.rowset(outlets) for pl in plans]
        a.upload_forcing(nsteps, days[0], q0)               # day 0 the ordinary way: the sequence needs a first state
        a.route_begin(nsteps, qts, True)
        with pytest.raises(RuntimeError, match="queued to its end"):
            a.stage_forcing(nsteps, days[2])                # (a window in progress must be queued to its end first)
        a.route_advance(nsteps)
        if ahead:                                           # ahead: a day's forcing is staged TWO days before, on the plan
            b.stage_forcing(nsteps, days[1])                # that will route it, behind the set-up of the day it is routing
            a.fetch_begin(rs[0], True)
            a.stage_forcing(nsteps, days[2])
        got = []
        keep = lambda hs: (hs[0].copy(), hs[1].copy())      # noqa: E731  (the arrays belong to a ring of three sets per plan)
        for w in range(1, ndays):
            cur, prev = plans[w % 2], plans[(w - 1) % 2]
            if not ahead:
                cur.stage_forcing(nsteps, days[w])          # (no wait: beside the window `prev` is routing)
            cur.chain_from(prev)
            cur.route_begin(nsteps, qts, True)
            cur.route_advance(nsteps)
            if ahead:                                       # ... and the day's fetch is queued WITH its window: the gathers
                cur.fetch_begin(rs[w % 2], True)            # right behind its last launch, not behind the other plan's tail
                if w + 2 < ndays:
                    cur.stage_forcing(nsteps, days[w + 2])
            prev.route_end()
            if ahead:
                got.append(keep(prev.fetch_wait()))
            else:
                if w >= 2:
                    got.append(keep(plans[w % 2].fetch_wait()))  # (the fetch of day w - 2 ran on this plan's copy stream)
                prev.fetch_begin(rs[(w - 1) % 2], True)
        last = plans[(ndays - 1) % 2]
        last.route_end()
        if ahead:
            got.append(keep(last.fetch_wait()))
        else:
            got.append(keep(plans[(ndays - 2) % 2].fetch_wait()))
            last.fetch_begin(rs[(ndays - 1) % 2], True)
            got.append(keep(last.fetch_wait()))
        assert len(got) == ndays
        for w, (h, st) in enumerate(got):
            assert np.array_equal(h.view(np.uint32), want_h[w].view(np.uint32)), w
            assert np.array_equal(st.view(np.uint32), want_s[w].view(np.uint32)), w
        # a staged forcing without a hand-over continues from the plan's own last window
        last.stage_forcing(nsteps, days[0])
        last.route_device(nsteps, qts, True)
        with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as ref:
            ref.upload_forcing(nsteps, days[0], want_s[-1])
            ref.route_device(nsteps, qts, True)
            assert np.array_equal(last.download_final_state().view(np.uint32), ref.download_final_state().view(np.uint32))
        # a forcing staged twice before its window (a corrected file), and a synchronous upload that replaces a staged one:
        # the state gathered by the first staging stands, and the older copy cannot land on top of the newer forcing
        want_next = ref_state = None
        with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as ref:
            ref.upload_forcing(nsteps, days[3], last.download_final_state())
            ref.route_device(nsteps, qts, True)
            want_next = ref.download_final_state()
        last.stage_forcing(nsteps, days[2])                  # the wrong day
        last.stage_forcing(nsteps, days[3])                  # corrected: routed_nsteps is -1 by now, the state is in in_q0
        last.route_device(nsteps, qts, True)
        assert np.array_equal(last.download_final_state().view(np.uint32), want_next.view(np.uint32))
        with RoutingPlan(up_ptr, up_idx, p, assume_short_ts=True) as ref:
            ref.upload_forcing(nsteps, days[4], want_next)
            ref.route_device(nsteps, qts, True)
            ref_state = ref.download_final_state()
        last.stage_forcing(nsteps, days[5])                  # the wrong day again, asynchronously ...
        last.upload_forcing(nsteps, days[4], None)           # ... replaced by a synchronous upload that continues the state
        last.route_device(nsteps, qts, True)
        assert np.array_equal(last.download_final_state().view(np.uint32), ref_state.view(np.uint32))
        # ... and a plan that has routed nothing must be chained to
        c = a.clone()
        c.stage_forcing(nsteps, days[1])
        with pytest.raises(RuntimeError, match="routed nothing"):
            c.route_begin(nsteps, qts, True)
        c.close()
        a.close()                                            # (the original first: the shared memory goes with the last user)
        b.stage_forcing(nsteps, days[2])
        b.route_device(nsteps, qts, True)
        b.close()
