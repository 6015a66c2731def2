"""troute_amd.sequence.DaySequence -- the pipeline bench.py times, as the package's API -- on one GPU and as two ranks of a
job (two threads, one device, the shared-memory transport): consecutive windows with DISTINCT forcing, every day's forcing
staged from page-locked memory, the state carried on in HBM, the products of every day fetched beside the next one --
against the same days routed one after the other by plain ``upload`` / ``route`` calls with the state taken through the
host (``AbstractNetwork.new_q0``'s round trip, AbstractNetwork.py:177-191), bit for bit."""
import os
import threading

import numpy as np
import pytest

from troute_amd import synthetic
from troute_amd.comm import Comm
from troute_amd.distributed import ShardedRouter
from troute_amd.sequence import DaySequence, pinned_like

pytestmark = pytest.mark.gpu
_serial = [0]


def days_of(net, n, seed=3):
    rng = np.random.default_rng(seed)
    return [rng.uniform(0, 0.6, net["qlat"].shape).astype(np.float32) for _ in range(n)]


def reference_days(net, days, q0, nsteps, qts, stride=None):
    """every day on ONE plain router, the state through the host: (outlet rows, [hydrographs per day], [final state per day]
    [, every stride-th step of every row per day])"""
    r = ShardedRouter(net["to"], net["params"], assume_short_ts=True)
    hyds, states, fvds, state = [], [], [], q0
    for d in days:
        r.upload(nsteps, d, state)
        rows, hyd = r.route(qts, True)
        state = r.plan0.download_final_state()
        hyds.append(hyd)
        states.append(state)
        if stride:
            fvds.append(r.plan0.download_fvd()[:, stride - 1::stride].copy())
    r.close()
    return (rows, hyds, states, fvds) if stride else (rows, hyds, states)


@pytest.mark.parametrize("mid", [False, True])
def test_a_sequence_of_days_on_one_gpu_equals_the_days_routed_one_by_one(monkeypatch, mid):
    monkeypatch.setenv("TRMC_ENGINE", "levels")
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    monkeypatch.setenv("TRMC_MID_MIN_ROWS", "8" if mid else "0")
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    nsteps, qts, ndays = 36, 12, 6
    q0 = np.random.default_rng(1).uniform(0, 1, (nseg, 3)).astype(np.float32)
    days = days_of(net, 4)                                   # a ring of four distinct days, six windows
    seq_days = [days[w % 4] for w in range(ndays)]
    rows, want_h, want_s = reference_days(net, seq_days, q0, nsteps, qts)
    r = ShardedRouter(net["to"], net["params"], assume_short_ts=True)
    got = {}
    with DaySequence(r, nsteps, qts) as seq:
        out = seq.run([pinned_like(d) for d in days], q0, ndays - 1, 1,
                      on_day=lambda w, h, s: got.__setitem__(w, (h.copy(), s.copy())))
        assert out["days_routed"] == ndays and len(out["ms_main"]) == ndays - 1 and out["el"] > 0
        st = out["last_plan"].stats()
        assert st["wide_levels"] > 0 and (st["mid_levels"] > 0) == mid
        # a second run on the same object continues from an explicit state again
        out2 = seq.run([pinned_like(d) for d in days], q0, 2, 0)
        assert np.array_equal(out2["final"].view(np.uint32), want_s[1].view(np.uint32))
    assert sorted(got) == list(range(ndays))
    assert np.array_equal(r.my_out0_global, rows)
    for w in range(ndays):
        assert np.array_equal(got[w][0].view(np.uint32), want_h[w].view(np.uint32)), w
        assert np.array_equal(got[w][1].view(np.uint32), want_s[w].view(np.uint32)), w
    # the router is an ordinary one again afterwards
    r.upload(nsteps, seq_days[0], q0)
    rows2, hyd2 = r.route(qts, True)
    assert np.array_equal(hyd2.view(np.uint32), want_h[0].view(np.uint32))
    r.close()
    with pytest.raises(ValueError, match="assume_short_ts"):
        DaySequence(r, nsteps, qts, assume_short_ts=False)


@pytest.mark.parametrize("engine", [None, "levels"])
def test_a_sequence_of_days_on_two_ranks_equals_the_days_routed_one_by_one(monkeypatch, engine):
    """Every rank stages its rows of each day, carries its state on in HBM (the merged plan's resident state, the lagged
    trunk included) and fetches its final state; rank 0 also the all-gathered outlet block."""
    if engine == "levels":
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
        monkeypatch.setenv("TRMC_WIDE_K", "4")
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    nsteps, qts, ndays = 36, 12, 5
    q0 = np.random.default_rng(2).uniform(0, 1, (nseg, 3)).astype(np.float32)
    days = days_of(net, 3, seed=5)
    seq_days = [days[w % 3] for w in range(ndays)]
    stride = 12 if engine == "levels" else None          # (one of the two variants also asks for the decimated result)
    rows1, want_h, want_s, *want_f = reference_days(net, seq_days, q0, nsteps, qts, stride)
    kw = {} if engine is None else {"engine": engine, "assume_short_ts": True}
    world = 2
    _serial[0] += 1
    key = f"seq{os.getpid()}_{_serial[0]}"
    results, errors = [None] * world, []

    def run(rank):
        try:
            comm = Comm(rank, world, device=0, backend="shm", key=key)
            r = ShardedRouter(net["to"], net["params"], rank=rank, world=world, device=0, **kw)
            r.enable_device_exchange(comm)
            got = {}
            seq = DaySequence(r, nsteps, qts, output_stride=stride)
            out = seq.run(days, q0, ndays - 1, 1, on_day=lambda w, h, s, *f: got.__setitem__(
                w, (None if h is None else np.array(h, copy=True), [np.array(x, copy=True) for x in s],
                    [np.array(x, copy=True) for x in f[0]] if f else None)))
            routed = np.ones(r.sequence_rows().shape[0], bool)          # (not the boundary copies of the cut rows: flow only)
            if r.plan1 is not None:
                routed[r.rows0.shape[0]:] = ~r.boundary1
            results[rank] = (np.array(r._out_rows, copy=True), got, r.sequence_rows(), out["days_routed"], r.plan1 is not None, routed)
            seq.close()
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
    assert results[0][4] or results[1][4]               # one of them owns a trunk
    rows, got0 = results[0][0], results[0][1]
    assert np.array_equal(rows, rows1) and sorted(got0) == list(range(ndays))
    for w in range(ndays):
        assert np.array_equal(got0[w][0].view(np.uint32), want_h[w].view(np.uint32)), w
    for rank in range(world):
        _, got, srows, n, _, routed = results[rank]
        assert n == ndays
        for w in range(ndays):
            state = got[w][1][0]                         # the merged plan's final state: this rank's rows, boundary copies too
            assert state.shape == (srows.shape[0], 3)
            assert np.array_equal(state[routed][:, [0, 2]].view(np.uint32), want_s[w][srows[routed]][:, [0, 2]].view(np.uint32)), (rank, w)
            assert np.array_equal(state[:, 0].view(np.uint32), want_s[w][srows][:, 0].view(np.uint32)), (rank, w)
            if stride:
                fvd = got[w][2][0]
                assert fvd.shape == (srows.shape[0], nsteps // stride, 3)
                assert np.array_equal(fvd[routed].view(np.uint32), want_f[0][w][srows[routed]].view(np.uint32)), (rank, w)


def test_a_sequence_with_decimated_output_hands_every_day_the_slices_of_its_result(monkeypatch):
    """DaySequence(output_stride=n): every n-th step of every row's (q, v, d) arrives with each day's products -- copied
    beside the NEXT day on the plan's clone -- and equals the slices of the day routed by itself."""
    monkeypatch.setenv("TRMC_ENGINE", "levels")
    monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
    monkeypatch.setenv("TRMC_WIDE_K", "8")
    net = synthetic.generate(nseg=20000, nnet=60, seed=11, nq=3)
    nseg = net["to"].shape[0]
    nsteps, qts, ndays, stride = 36, 12, 5, 12
    q0 = np.random.default_rng(1).uniform(0, 1, (nseg, 3)).astype(np.float32)
    days = days_of(net, 3)
    r = ShardedRouter(net["to"], net["params"], assume_short_ts=True)
    want, state = [], q0
    for w in range(ndays):
        r.upload(nsteps, days[w % 3], state)
        r.route(qts, True)
        want.append(r.plan0.download_fvd()[:, stride - 1::stride].copy())
        state = r.plan0.download_final_state()
    got = {}
    with DaySequence(r, nsteps, qts, output_stride=stride) as seq:
        out = seq.run([pinned_like(d) for d in days], q0, ndays - 1, 1,
                      on_day=lambda w, h, s, f: got.__setitem__(w, f.copy()))
        assert out["fvd"].shape == (nseg, nsteps // stride, 3)
    assert sorted(got) == list(range(ndays))
    for w in range(ndays):
        assert np.array_equal(got[w].view(np.uint32), want[w].view(np.uint32)), w
    r.close()
