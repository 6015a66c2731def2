"""Parity of the HIP path against the oracle and the reference goldens -- needs an MI355X.

All calls go through the C ABI (ctypes -> libtrmc.so).  The fp32 path uses the bit-reproducible
powf of det_pow.h (glibc 2.35's algorithm, equal to this image's libm powf on all 2^32 inputs at
the kernel's two exponents), so:

  fp32 GPU  ==  oracle (det instantiation)  ==  oracle (libm)  ==  reference Fortran, bit for bit.

Stated tolerances
  fp32 vs oracle and vs the reference-Fortran goldens : bit-identical (NaN patterns included) --
       segment steps, LowerColorado 11 248 x 288 in BOTH timestep modes, all ragged cases
  fp64 vs the reference built with -fdefault-real-8 (BASELINE configs[1]) : bit-identical as well -- the fp64 path
       takes its power from det_pow64.h (glibc 2.35's pow restated; pinned statistically, tests/test_diffusive.py)
  full-ts from a cold start is chaotic in the reference itself (test_oracle_pinning.py), which is
  why bit-exactness -- not a tolerance -- is the parity statement.
"""
import os

import numpy as np
import pytest

import helpers as H
from helpers import flow_engine
from oracle import oracle as O
from troute_amd import _lib
from troute_amd.plan import RoutingPlan, csr_from_lists, segments, topology_levels
from troute_amd.routing.fast_reach.mc_reach import compute_network_structured, mc_only_args
from troute_amd.routing.fast_reach.reach import compute_reach_kernel

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def assert_bit_identical(got, want, what=""):
    gb, wb = bits(got), bits(want)
    if not np.array_equal(gb, wb):
        bad = np.argwhere(gb != wb)
        raise AssertionError(f"{what}: {bad.shape[0]} of {gb.size} values differ; first at {bad[0].tolist()}: "
                             f"{got[tuple(bad[0])]!r} vs {want[tuple(bad[0])]!r}")


def test_gpu_present_and_native_library_loaded():
    assert _lib.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libtrmc.so" in maps


# ---- one segment, one timestep (reference: c_muskingcungenwm / compute_reach_kernel) ----------------
def test_segment_step_fp32_bit_identical_to_det_oracle():
    x = H.load_kernel_vectors()["inputs_f64"].astype(np.float32)
    assert_bit_identical(segments(x), O.segments(x, det=True), "12k kernel vectors")


def test_segment_step_fp32_bit_identical_to_reference_fortran():
    """Golden = the reference Fortran (canonical Qj_0) compiled in the dev container."""
    kv = H.load_kernel_vectors()
    got = segments(kv["inputs_f64"].astype(np.float32))
    assert_bit_identical(got, kv["ref_qj0_f32"], "kernel vectors vs reference Fortran")


def test_segment_step_fp64_bit_identical_to_reference_fortran():
    """Golden = the reference Fortran promoted to double (-fdefault-real-8, oracle/_ref/libmc_ref_qj0_f64.so)."""
    kv = H.load_kernel_vectors()
    x = kv["inputs_f64"].astype(np.float32).astype(np.float64)
    got = segments(x)
    assert_bit_identical(got, kv["ref_qj0_f64"], "fp64 kernel vectors vs reference Fortran (-fdefault-real-8)")


def test_reference_signature_entry_point():
    """trmc_muskingcungenwm: the reference's own C binding of one segment-step (c_muskingcungenwm,
    pyMCsingleSegStime_NoLoop.f90:8-21 / .h:1-21 -- 21 float pointers, no return value), one step per call.  The first
    200 golden vectors of the reference Fortran plus its low-flow KAT, all six outputs, bit for bit; a NULL argument gives
    NaNs and a message, like a routine that cannot signal."""
    import ctypes as C
    from troute_amd import _lib
    lib = _lib.lib()
    kv = H.load_kernel_vectors()
    x = kv["inputs_f64"].astype(np.float32)[:200]
    want = kv["ref_qj0_f32"][:200]
    got = np.zeros((x.shape[0], 6), np.float32)
    for i, row in enumerate(x):
        a = [C.c_float(float(v)) for v in row] + [C.c_float(123.0) for _ in range(6)]   # (outputs start as garbage)
        lib.trmc_muskingcungenwm(*[C.byref(v) for v in a])
        got[i] = [v.value for v in a[15:]]
    assert_bit_identical(got, want, "trmc_muskingcungenwm vs reference Fortran")
    a = [C.c_float(1.0) for _ in range(21)]
    ptrs = [C.byref(v) for v in a]
    ptrs[3] = None
    lib.trmc_muskingcungenwm(*ptrs)
    assert all(np.isnan(v.value) for v in a[15:]) and b"NULL" in lib.trmc_last_error()


def test_compute_reach_kernel_dict_entry():
    """The reference's dict-returning test entry (reach.pyx:66-103) on its low-flow KAT."""
    r = compute_reach_kernel(60.0, 0.04598825, 0.04598825, 0.21487340, 40.0, 1800.0, 112.0, 448.0,
                             623.5999755859375, 0.02800000086426735, 0.03136000037193298,
                             1.399999976158142, 0.0017999999690800905, 0.0704801953, 0.0100334705)
    assert set(r) == {"qdc", "velc", "depthc", "cn", "ck", "X"}
    exp = (0.7570106983184814, 0.12373604625463486, 0.02334451675415039)   # mc_sseg_stime_NOLOOP_demo.py:229-231
    assert np.allclose([r["qdc"], r["velc"], r["depthc"]], exp, rtol=2e-7)


# ---- LowerColorado_TX MC-only through the drop-in callable --------------------------------------------
@pytest.fixture(scope="module")
def lc():
    return H.LowerColorado()


def route_lc(lc, short, precision=32, **kw):
    args = mc_only_args(lc.nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols, lc.data_values,
                        lc.q0, lc.qlat, assume_short_ts=short)
    r = compute_network_structured(*args, precision=precision, return_stats=True, **kw)
    return r, r[1].reshape(lc.nseg, lc.nts, 3)


@pytest.mark.parametrize("short", [True, False])
def test_lowercolorado_fp32_bit_identical_to_det_oracle(lc, short):
    r, fvd = route_lc(lc, short)
    reaches, ups = lc.row_lists()
    want = O.network(lc.nts, lc.qts, reaches, ups, lc.params9, lc.q0, lc.qlat, short, det=True)[:, 1:, :]
    assert_bit_identical(fvd, want, f"LowerColorado short={short}")
    st = r[-1]
    assert st["nlevels"] == 649 and st["nseg_routed"] == 11248
    # level engine: one launch per timestep / per wavefront diagonal; dataflow engine: one launch per window
    assert st["main_launches"] == (1 if flow_engine() else 288 if short else 649 + 288 - 1)


def test_lowercolorado_return_tuple_shape(lc):
    r, _ = route_lc(lc, True)
    assert len(r) == 11                                   # 10-tuple + stats
    ids, fvd = r[0], r[1]
    assert ids.dtype == np.intp and np.array_equal(ids, lc.ids)
    assert fvd.shape == (lc.nseg, lc.nts * 3) and fvd.dtype == np.float32
    assert r[2] == 0 and len(r[3]) == 3 and len(r[4]) == 5 and len(r[5]) == 5
    assert r[6].shape == (lc.nseg, lc.nts) and len(r[7]) == 3 and r[8].shape == (0, lc.nts + 1) and len(r[9]) == 4


@pytest.mark.parametrize("engine", ["flow", "levels", "levels-wide", "levels-mid"])
@pytest.mark.parametrize("short", [True, False])
def test_lowercolorado_fp32_bit_identical_to_reference_golden(lc, short, engine, monkeypatch):
    """Golden = reference Fortran kernel symbol driven through the restated loop (make_fixtures.py):
    12 time slices x every segment and 100 probe segments x every step, both timestep modes, on BOTH engines
    (the dataflow engine k_mc_flow and the level engine k_mc_step), the level engine also with its wide levels several
    steps per launch (k_mc_tile), and with a second tier of levels below those (short-timestep mode)."""
    monkeypatch.setenv("TRMC_ENGINE", engine.split("-")[0])
    if engine.endswith("-wide"):
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
        monkeypatch.setenv("TRMC_WIDE_K", "5")
    if engine.endswith("-mid"):
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
        monkeypatch.setenv("TRMC_WIDE_K", "8")
        monkeypatch.setenv("TRMC_MID_MIN_ROWS", "8")
        monkeypatch.setenv("TRMC_MID_K", "2")
    _, fvd = route_lc(lc, short)
    g = lc.golden()
    tag = "shortts" if short else "fullts"
    assert_bit_identical(fvd[:, g["tsel"] - 1, :], g[f"{tag}_f32_tsel"], f"{tag} time slices")
    assert_bit_identical(fvd[g["probes"]], g[f"{tag}_f32_probes"][:, 1:, :], f"{tag} probes")


def test_lowercolorado_fp64_bit_identical_to_reference_golden(lc):
    """BASELINE configs[1]: LowerColorado MC-only in fp64 against the reference Fortran promoted to double, driven
    through the restated loop (make_fixtures.py) -- final step of every segment and 100 probe hydrographs, bit for bit."""
    _, fvd = route_lc(lc, True, precision=64)
    g = lc.golden()
    assert fvd.dtype == np.float64
    assert_bit_identical(fvd[:, -1, :], g["shortts_f64_final"], "fp64 final step")
    assert_bit_identical(fvd[g["probes"]], g["shortts_f64_probes"][:, 1:, :], "fp64 probes")


def test_final_state_and_outlet_gather_agree_with_full_result(lc):
    from troute_amd.routing.fast_reach.mc_reach import _flatten_network
    up_ptr, up_idx, _ = _flatten_network([(r, 0) for r in lc.reaches], lc.rconn, lc.ids)
    with RoutingPlan(up_ptr, up_idx, lc.params9) as plan:
        fvd = plan.route(lc.nts, lc.qts, True, lc.qlat, lc.q0)
        fs = plan.download_final_state()
        rows = np.array([0, 17, lc.nseg - 1, 5000], np.int64)
        hyd = plan.gather_flow_rows(rows)
    # new_q0 = fvd[:, [-3,-3,-1]] (AbstractNetwork.py:182-190)
    assert_bit_identical(fs, fvd[:, -1, :][:, [0, 0, 2]], "final state")
    assert_bit_identical(hyd, fvd[rows, :, 0], "gathered hydrographs")


def test_upstream_results_composition_equals_whole_network(lc):
    """Ordered sub-network execution (compute.py:553-907): route the part above a cut segment first,
    hand its tailwater hydrograph over as upstream_results (mc_reach.pyx:451-469), route the rest --
    bit-identical to routing the whole network at once, and the hand-off row is masked from the output."""
    row = {int(s): i for i, s in enumerate(lc.ids)}
    # pick the last segment of a mid-network reach with a sizeable sub-tree
    up_of = lc.rconn
    def subtree(seg):
        out, stack = [], [seg]
        while stack:
            s = stack.pop()
            out.append(s)
            stack.extend(up_of.get(s, []))
        return out
    cut = None
    for r in lc.reaches[::-1]:
        n = len(subtree(r[-1]))
        if 800 < n < 4000:
            cut = r[-1]
            break
    assert cut is not None
    upper = set(subtree(cut))
    for short in (True, False):
        _, whole = route_lc(lc, short)
        # upper part on its own
        r_up = [r for r in lc.reaches if r[0] in upper]
        ids_up = np.array(sorted(upper), np.int64)
        sel = np.array([row[s] for s in ids_up])
        a = mc_only_args(lc.nts, lc.dt, lc.qts, r_up, lc.rconn, ids_up, lc.data_cols, lc.data_values[sel],
                         lc.q0[sel], lc.qlat[sel], assume_short_ts=short)
        ru = compute_network_structured(*a)
        fu = ru[1].reshape(len(ids_up), lc.nts, 3)
        assert_bit_identical(fu, whole[sel], "upper part alone")
        # lower part, fed by the cut segment's hydrograph
        ids_lo = np.array(sorted((set(lc.ids.tolist()) - upper) | {cut}), np.int64)
        sel = np.array([row[s] for s in ids_lo])
        r_lo = [r for r in lc.reaches if r[0] not in upper]
        pos = int(np.searchsorted(ids_lo, cut))
        ur = {cut: {"position_index": pos, "results": fu[int(np.searchsorted(ids_up, cut))].reshape(-1)}}
        a = mc_only_args(lc.nts, lc.dt, lc.qts, r_lo, lc.rconn, ids_lo, lc.data_cols, lc.data_values[sel],
                         lc.q0[sel], lc.qlat[sel], upstream_results=ur, assume_short_ts=short)
        rl = compute_network_structured(*a)
        assert rl[0].shape[0] == len(ids_lo) - 1 and cut not in rl[0]
        keep = np.array([row[s] for s in rl[0]])
        assert_bit_identical(rl[1].reshape(-1, lc.nts, 3), whole[keep], f"lower part short={short}")


# ---- ragged / edge shapes on random forests ---------------------------------------------------------------
def synth_inputs(rng, n, nq, dt_uniform=True):
    p = np.stack([np.full(n, 300.0) if dt_uniform else rng.choice([60.0, 300.0, 600.0], n),
                  rng.uniform(200, 4000, n), rng.uniform(0.5, 20, n), np.zeros(n), np.zeros(n),
                  rng.choice([0.04, 0.05, 0.06], n), np.zeros(n), rng.uniform(0.1, 2.0, n),
                  np.exp(rng.uniform(np.log(1e-4), np.log(0.1), n))], 1)
    p[:, 3] = p[:, 2] * 5 / 3
    p[:, 4] = p[:, 3] * 3
    p[:, 6] = 2 * p[:, 5]
    ql = (np.exp(rng.normal(np.log(5e-3), 2.0, (n, nq))) * (rng.random((n, nq)) > 0.1)).astype(np.float32)
    q0 = np.stack([rng.uniform(0, 2, n), rng.uniform(0, 2, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    q0[rng.random(n) < 0.3] = 0
    return p.astype(np.float32), ql, q0


def run_both(ups, params, qlat, q0, nsteps, qts, short):
    up_ptr, up_idx = csr_from_lists(ups)
    lvl, _, _ = topology_levels(up_ptr, up_idx)
    with RoutingPlan(up_ptr, up_idx, params) as plan:
        got = plan.route(nsteps, qts, short, qlat, q0)
    want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, short, det=True)[:, 1:, :]
    return got, want


# TRMC_ENGINE: the dataflow engine (k_mc_flow*), the level engine one step per launch (k_mc_step), and the level engine
# with its wide levels routed several steps per launch under a level skew (k_mc_tile; at its default thresholds only
# networks of CONUS width take that path -- here every level of 32 rows or more does, five steps per launch)
ENGINES = ["flow", "levels", "levels-wide", "levels-mid"]


def set_engine(monkeypatch, engine):
    """levels: one launch per timestep (k_mc_step); levels-wide: the leading levels several steps per launch beside it
    (k_mc_tile); levels-mid: two tiers of them -- levels of at least 64 rows eight steps per launch, the levels of at least 8
    rows right below them four steps per launch under their own skew, the rest one step per launch."""
    monkeypatch.setenv("TRMC_ENGINE", engine.split("-")[0])
    if engine.endswith("-mid"):
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "64")
        monkeypatch.setenv("TRMC_WIDE_K", "8")
        monkeypatch.setenv("TRMC_MID_MIN_ROWS", "8")
        monkeypatch.setenv("TRMC_MID_K", "4")
        monkeypatch.setenv("TRMC_MID_LEVELS", "20")
        monkeypatch.setenv("TRMC_HOT_ROWS", "1")
    elif engine.endswith("-wide"):
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "32")
        monkeypatch.setenv("TRMC_WIDE_K", "8")        # (a multiple of 4: the 16-byte result stores where nsteps allows them)
        monkeypatch.setenv("TRMC_TILE_PERM", "512")   # (rows re-dealt to a tile's threads by class, also on plans without a hint)
        monkeypatch.setenv("TRMC_HOT_ROWS", "1")      # (rows of three or more iterations in blocks of their own, hinted plans too)
    else:
        monkeypatch.setenv("TRMC_WIDE_MIN_ROWS", "0")


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("nseg", [1, 2, 63, 64, 65, 1000, 6000])
@pytest.mark.parametrize("short", [True, False])
def test_random_forests_bit_identical(nseg, short, engine, monkeypatch):
    set_engine(monkeypatch, engine)
    rng = np.random.default_rng(1000 + nseg)
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    nsteps, qts = 30, 4
    params, qlat, q0 = synth_inputs(rng, nseg, 8)
    got, want = run_both(ups, params, qlat, q0, nsteps, qts, short)
    assert_bit_identical(got, want, f"forest n={nseg} short={short}")


@pytest.mark.parametrize("place", ["lean", "staged"])
def test_flow_engine_block_placement_changes_nothing(place, monkeypatch):
    """Lean launches of the dataflow engine deal their blocks to the compute units by cost (per-unit queues, row groups
    matched with SIMDs; trmc.hip lean_pick_block / flow_place_blocks); the staged form hands its blocks out by ticket
    (trmc_plan_options.flow_lean < 0).  The same bits either way -- also with far fewer blocks than units (every workgroup
    but a few takes from a queue not its own) and with a cost hint that makes the queues uneven."""
    set_engine(monkeypatch, "flow")
    monkeypatch.setenv("TRMC_FLOW_LEAN", "1" if place == "lean" else "0")
    for nseg in (700, 40000):
        rng = np.random.default_rng(77 + nseg)
        to = H.random_network(rng, nseg)
        _, _, ups = H.reaches_from_to(to)
        nsteps, qts = 24, 4
        params, qlat, q0 = synth_inputs(rng, nseg, 6)
        up_ptr, up_idx = csr_from_lists(ups)
        lvl, _, _ = topology_levels(up_ptr, up_idx)
        hint = rng.integers(0, 8, nseg).astype(np.uint8) * np.uint8(16)
        with RoutingPlan(up_ptr, up_idx, params, cost_hint=hint, assume_short_ts=True, engine="flow") as plan:
            assert plan.engine == "flow"
            got = plan.route(nsteps, qts, True, qlat, q0)
            again = plan.route(nsteps, qts, True, qlat, q0)
        want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, True, det=True)[:, 1:, :]
        assert_bit_identical(got, want, f"placement {place} n={nseg}")
        assert_bit_identical(again, want, f"placement {place} n={nseg}, second window")


@pytest.mark.parametrize("nsteps,qts", [(1, 1), (65, 1), (130, 12), (64, 64)])
def test_timestep_and_forcing_shapes(nsteps, qts):
    """Tile edges of the result transpose (64-step tiles), qts = 1 (one forcing column per step), one
    single step."""
    rng = np.random.default_rng(nsteps * 7 + qts)
    nseg = 777
    _, _, ups = H.reaches_from_to(H.random_network(rng, nseg))
    nq = (nsteps - 1) // qts + 1
    params, qlat, q0 = synth_inputs(rng, nseg, nq)
    for short in (True, False):
        got, want = run_both(ups, params, qlat, q0, nsteps, qts, short)
        assert_bit_identical(got, want, f"nsteps={nsteps} qts={qts} short={short}")


def test_deep_chain_and_wide_junction():
    """A single 1500-segment reach (1500 levels, width 1) and a junction with 6 tributaries."""
    rng = np.random.default_rng(9)
    n = 1500
    ups = [[]] + [[i - 1] for i in range(1, n)]
    params, qlat, q0 = synth_inputs(rng, n, 3)
    for short in (True, False):
        got, want = run_both(ups, params, qlat, q0, 20, 12, short)
        assert_bit_identical(got, want, "deep chain")
    ups = [[] for _ in range(6)] + [[3, 0, 5, 1, 4, 2]] + [[6]]          # summation order is the given order
    params, qlat, q0 = synth_inputs(rng, 8, 3)
    for short in (True, False):
        got, want = run_both(ups, params, qlat, q0, 20, 12, short)
        assert_bit_identical(got, want, "wide junction")


@pytest.mark.parametrize("stem_min", [None, "64"])
def test_general_mode_on_a_basin_with_a_long_stem_starts_the_stem_first_and_keeps_the_bits(monkeypatch, stem_min):
    """A dataflow plan built for the general mode lays a basin with a long stem out stem-last, side tributaries from the top
    of the stem down, and starts the stem's blocks first (topology.hpp, stem_min_rows; k_mc_flow<false>'s ticket map) --
    the same bits as the oracle, window after window; also with small stems (several basins started early) and a
    short-timestep window on the same plan."""
    from troute_amd.plan import topology_blocks_general
    set_engine(monkeypatch, "flow")
    if stem_min:
        monkeypatch.setenv("TRMC_STEM_MIN_ROWS", stem_min)
    rng = np.random.default_rng(21)
    ups = [[] if i == 0 else [i - 1] for i in range(1300)]             # a stem of 1 300 rows ...
    for i in range(1, 1300):
        if rng.random() < 0.5:                                         # ... with side tributaries: small random trees
            base, m = len(ups), int(rng.integers(1, 40))
            for j in range(m):
                ups.append([])
                if j:
                    ups[base + int(rng.integers(0, j))].append(base + j)
            ups[i].append(base)
    for _ in range(40):                                                # other basins: chains of 1 to 200 rows
        base, m = len(ups), int(rng.integers(1, 200))
        for j in range(m):
            ups.append([] if j == 0 else [base + j - 1])
    n = len(ups)
    perm = rng.permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    ups_p = [[] for _ in range(n)]
    for r in range(n):
        ups_p[inv[r]] = [int(inv[u]) for u in ups[r]]
    up_ptr, up_idx = csr_from_lists(ups_p)
    early = topology_blocks_general(up_ptr, up_idx, None, int(stem_min or 1024))[4]
    assert early.size >= (6 if stem_min is None else 7)               # the long stem's blocks (and, with 64, other basins')
    lvl, _, _ = topology_levels(up_ptr, up_idx)
    params, qlat, q0 = synth_inputs(rng, n, 3)
    nsteps, qts = 30, 12
    with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=False, engine="flow") as plan:
        assert plan.engine == "flow"
        got = plan.route(nsteps, qts, False, qlat, q0)
        again = plan.route(nsteps, qts, False, qlat, q0)
        short = plan.route(nsteps, qts, True, qlat, q0)
    want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, False, det=True)[:, 1:, :]
    assert_bit_identical(got, want, "long stem, general mode")
    assert_bit_identical(again, want, "long stem, general mode, second window")
    want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, True, det=True)[:, 1:, :]
    assert_bit_identical(short, want, "long stem, a short-timestep window on the plan laid out for the general mode")


def test_non_uniform_dt_column():
    rng = np.random.default_rng(11)
    nseg = 500
    _, _, ups = H.reaches_from_to(H.random_network(rng, nseg))
    params, qlat, q0 = synth_inputs(rng, nseg, 4, dt_uniform=False)
    for short in (True, False):
        got, want = run_both(ups, params, qlat, q0, 24, 6, short)
        assert_bit_identical(got, want, "per-segment dt")


def test_zero_forcing_zero_state_stays_zero():
    rng = np.random.default_rng(12)
    nseg = 300
    _, _, ups = H.reaches_from_to(H.random_network(rng, nseg))
    params, qlat, q0 = synth_inputs(rng, nseg, 2)
    got, _ = run_both(ups, params, np.zeros_like(qlat), np.zeros_like(q0), 12, 12, False)
    assert (got == 0).all()


@pytest.mark.parametrize("short", [True, False])
def test_random_forest_fp64_bit_identical_to_oracle(short):
    rng = np.random.default_rng(13)
    nseg = 2000
    _, _, ups = H.reaches_from_to(H.random_network(rng, nseg))
    params, qlat, q0 = synth_inputs(rng, nseg, 4)
    up_ptr, up_idx = csr_from_lists(ups)
    lvl, _, _ = topology_levels(up_ptr, up_idx)
    with RoutingPlan(up_ptr, up_idx, params, precision=64) as plan:
        got = plan.route(24, 6, short, qlat, q0)
    want = O.network_by_segment(24, 6, up_ptr, up_idx, lvl, params.astype(np.float64), q0, qlat, short)[:, 1:, :]
    assert_bit_identical(got, np.ascontiguousarray(want), f"fp64 forest short={short}")


# ---- the reference's error behaviour ---------------------------------------------------------------------
def test_errors_match_reference(lc):
    base = lambda **k: mc_only_args(lc.nts, lc.dt, lc.qts, lc.reaches, lc.rconn, lc.ids, lc.data_cols,  # noqa: E731
                                    lc.data_values, lc.q0, lc.qlat, **k)
    a = base()
    a[9] = lc.qlat[:-1]
    with pytest.raises(ValueError, match="Number of rows in Qlat is incorrect"):
        compute_network_structured(*a)
    a = base()
    a[9] = lc.qlat[:, :10]
    with pytest.raises(ValueError, match="Number of columns"):
        compute_network_structured(*a)
    a = base()
    a[7] = lc.data_values[:, :5]
    with pytest.raises(ValueError, match="data_values shape mismatch"):
        compute_network_structured(*a)
    a = base()
    a[3] = [(r, 0) for r in lc.reaches[:-1]] + [(lc.reaches[-1] + [123456789], 0)]
    with pytest.raises(ValueError, match="not found"):
        compute_network_structured(*a)
    a = base()
    single = next(r for r in lc.reaches if len(r) == 1)
    a[3] = [(r, 1 if r is single else 0) for r in lc.reaches]      # a "reservoir" that is in no lake table
    with pytest.raises(ValueError, match="not found"):
        compute_network_structured(*a)


# ---- BASELINE's full size: synthetic CONUS, 2 729 077 segments x 288 steps ------------------------------
@pytest.fixture(scope="module")
def conus():
    from troute_amd import synthetic
    net = synthetic.generate(cache_dir="/tmp/trmc_cache")
    up_ptr, up_idx = synthetic.upstream_csr(net["to"])
    return net, up_ptr, up_idx


def conus_sample_rows(to, rng, n_mid=80, n_small=200, lo=50, hi=20000):
    """rows of a random sub-collection of whole independent networks"""
    from troute_amd import sharding
    outlet = sharding.outlet_of(to)
    _, lab = np.unique(outlet, return_inverse=True)
    sizes = np.bincount(lab)
    cand = np.flatnonzero((sizes >= lo) & (sizes <= hi))
    pick = np.concatenate([rng.choice(cand, n_mid, replace=False), rng.choice(np.flatnonzero(sizes < lo), n_small, replace=False)])
    return np.flatnonzero(np.isin(lab, pick))


@pytest.mark.parametrize("short,plan_mode,engine", [(True, None, "flow"), (False, None, "flow"), (True, True, "levels"),
                                                    (True, True, "levels-mid"), (False, False, "flow")])
def test_conus_full_size_samples_bit_identical_to_oracle(conus, short, plan_mode, engine, monkeypatch):
    """Size-independent property at full size: independent networks do not interact, so any
    sub-collection of them routed ALONE by the oracle must equal -- bit for bit -- what the GPU
    produced for them inside the 2.7 M-segment run (outlet and interior hydrographs, final state).
    plan_mode: the timestep mode the plan is told it is for (RoutingPlan assume_short_ts) -- None: unknown, the dataflow
    engine; True at this size: TRMC_ENGINE_AUTO picks the LEVEL engine, k_mc_step at 2.7 M rows, the kernel bench.py times."""
    from troute_amd.distributed import restrict_csr
    net, up_ptr, up_idx = conus
    to = net["to"]
    nseg = to.shape[0]
    nsteps, qts = 288, 12
    q0 = np.zeros((nseg, 3), np.float32)
    rows = conus_sample_rows(to, np.random.default_rng(77))
    assert 10000 < rows.size < 400000
    # "levels": the default of a short-timestep plan at this size (k_mc_tile + k_mc_step + k_emit); "levels-mid": the
    # same plan with a second tier of tiles below the wide levels (levels of at least 16 384 rows, four steps per launch)
    if engine == "levels-mid":
        monkeypatch.setenv("TRMC_MID_MIN_ROWS", "16384")
    with RoutingPlan(up_ptr, up_idx, net["params"], assume_short_ts=plan_mode) as plan:
        assert plan.engine == engine.split("-")[0]
        plan.upload_forcing(nsteps, net["qlat"], q0)
        st = plan.route_device(nsteps, qts, short)
        hyd = plan.gather_flow_rows(rows)
        final = plan.download_final_state()
    assert st["segment_steps"] == nseg * nsteps
    if engine.startswith("levels"):
        assert st["wide_levels"] > 0 and (st["mid_levels"] > 0) == (engine == "levels-mid")
    g2l = np.full(nseg, -1, np.int64)
    g2l[rows] = np.arange(rows.size)
    lp, li = restrict_csr(up_ptr, up_idx, rows, g2l)
    lvl, _, _ = topology_levels(lp, li)
    want = O.network_by_segment(nsteps, qts, lp, li, lvl, net["params"][rows], q0[rows], net["qlat"][rows], short,
                                det=True)
    assert_bit_identical(hyd, want[:, 1:, 0], f"CONUS sample hydrographs short={short}")
    assert_bit_identical(final[rows], want[:, -1, :][:, [0, 0, 2]], "CONUS sample final state")
    assert np.isfinite(final).all() and (final[:, 0] >= 0).all()


def test_conus_bench_sequence_day_n_plus_1_bit_identical_to_oracle(conus):
    """The configuration bench.py TIMES, replayed step for step at full size and checked against the oracle where the
    bench is timed: day N-1 from a cold start on the plan built from the topology alone -> day N warm, with cost
    collection -> the plan rebuilt with that hint (assume_short_ts plan at 2.7 M rows: the level engine, k_mc_step, rows
    of a level grouped by cost) -> spun up again through days N-1 and N -> day N+1, warm.  A sub-collection of whole
    networks routed alone by the oracle through the same three days must equal what the GPU holds for them on day N+1:
    hydrographs of every sampled row, final state (reference loop semantics: mc_reach.pyx:492-505,:719-750; warm start
    between windows AbstractNetwork.py:177-191)."""
    from troute_amd import synthetic
    from troute_amd.distributed import ShardedRouter, restrict_csr
    net, up_ptr, up_idx = conus
    to, params = net["to"], net["params"]
    nseg = to.shape[0]
    nsteps, qts = 288, 12
    qlat_s = net["qlat"]
    qlat_a = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 1, previous=qlat_s)
    qlat_b = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2, previous=qlat_a)
    q0 = np.zeros((nseg, 3), np.float32)
    rows = conus_sample_rows(to, np.random.default_rng(78), n_mid=80, n_small=120, hi=10000)
    assert 3000 < rows.size < 120000

    def make(hint):
        r = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True)
        assert r.plan0.engine == "levels"
        return r
    r = make(None)
    r.upload(nsteps, qlat_s, q0)
    r.route_resident(qts, True)                      # day N-1, cold
    r.upload(nsteps, qlat_a, None)                   # day N, warm, cost collection on
    r.collect_cost(True)
    r.route_resident(qts, True)
    hint = r.iteration_hint()
    r.close()
    r = make(hint)
    r.upload(nsteps, qlat_s, q0)
    r.route_resident(qts, True)
    r.upload(nsteps, qlat_a, None)
    r.route_resident(qts, True)
    r.upload(nsteps, qlat_b, None)                   # day N+1: the timed window
    for _ in range(2):                               # (the bench routes it several times: warm-up, then the timed steps)
        r.route_resident(qts, True)
    hyd = r.plan0.gather_flow_rows(rows)
    final = r.plan0.download_final_state()
    r.close()

    g2l = np.full(nseg, -1, np.int64)
    g2l[rows] = np.arange(rows.size)
    lp, li = restrict_csr(up_ptr, up_idx, rows, g2l)
    lvl, _, _ = topology_levels(lp, li)
    state = q0[rows]
    for ql in (qlat_s, qlat_a, qlat_b):
        want = O.network_by_segment(nsteps, qts, lp, li, lvl, params[rows], state, ql[rows], True, det=True)
        state = np.ascontiguousarray(want[:, -1, :][:, [0, 0, 2]])
    assert_bit_identical(hyd, want[:, 1:, 0], "day N+1 hydrographs on the tuned plan")
    assert_bit_identical(final[rows], state, "day N+1 final state on the tuned plan")
    assert (hyd > 0).mean() > 0.5


def test_conus_every_segment_bit_identical_to_reference(conus):
    """EVERY one of the 2 729 077 segments of the timed configuration -- the dominant 1.35 M-segment basin, the only owner
    of the deep levels the tail kernel routes, included -- against the reference Fortran at full size: the three bench
    days (N-1 cold, N, N+1 warm; bench.py's sequence on the plan tuned on day N) routed on the CPU by the reference
    kernel with the canonical Qj_0 (oracle/_ref/libmc_ref_qj0_f32.so built from the reference's sources; the pinned
    restatement where that build is absent) over the reference's own decomposition into ordered sub-networks
    (oracle.reference_windows; loop semantics mc_reach.pyx:492-505,:719-750, warm start AbstractNetwork.py:177-191).
    Compared bit for bit on day N+1: the flow of every row at every one of the 288 steps, the velocity and depth series
    of every row (exact position-weighted checksums of their bit patterns), the final state of every row -- and the
    14 713 outlet hydrographs the two-rank device-exchange job gathers for the same three days."""
    import threading
    from troute_amd import synthetic
    from troute_amd.comm import Comm
    from troute_amd.distributed import ShardedRouter
    net, up_ptr, up_idx = conus
    to, params = net["to"], net["params"]
    nseg = to.shape[0]
    nsteps, qts = 288, 12
    qlat_s = net["qlat"]
    qlat_a = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 1, previous=qlat_s)
    qlat_b = synthetic.forcing(nseg, qlat_s.shape[1], synthetic.DEFAULT_SEED + 2, previous=qlat_a)
    q0 = np.zeros((nseg, 3), np.float32)

    ref = O.reference_windows(to, params, (qlat_s, qlat_a, qlat_b), q0, nsteps, qts, True)
    assert ref["q"].shape == (nseg, nsteps + 1)

    r = ShardedRouter(to, params, assume_short_ts=True)
    assert r.plan0.engine == "levels"
    r.upload(nsteps, qlat_s, q0)
    r.route_resident(qts, True)                      # day N-1, cold
    r.upload(nsteps, qlat_a, None)                   # day N, warm, cost collection on
    r.collect_cost(True)
    r.route_resident(qts, True)
    hint = r.iteration_hint()
    r.close()
    r = ShardedRouter(to, params, cost_hint=hint, assume_short_ts=True)
    r.upload(nsteps, qlat_s, q0)
    r.route_resident(qts, True)
    r.upload(nsteps, qlat_a, None)
    r.route_resident(qts, True)
    r.upload(nsteps, qlat_b, None)                   # day N+1: the timed window
    for _ in range(2):
        r.route_resident(qts, True)
    assert r.last_stats["phase0"]["segment_steps"] == nseg * nsteps
    fvd = r.plan0.download_fvd().reshape(nseg, nsteps, 3)
    final = r.plan0.download_final_state()
    r.close()
    for lo in range(0, nseg, 200000):                # (in slices: the comparison's temporaries stay small)
        assert_bit_identical(np.ascontiguousarray(fvd[lo:lo + 200000, :, 0]), ref["q"][lo:lo + 200000, 1:],
                             f"flow of every row, rows {lo}..")
    assert np.array_equal(O.series_checksum(fvd[:, :, 1]), ref["chk_v"]), "velocity series of some row differs"
    assert np.array_equal(O.series_checksum(fvd[:, :, 2]), ref["chk_d"]), "depth series of some row differs"
    assert_bit_identical(final, ref["state"], "final state of every row")
    del fvd

    # the two-rank job (two threads, one device, shared-memory transport): its product is the all-gathered outlet block
    world, key = 2, f"full{os.getpid()}"
    results, errors = [None] * world, []

    def run(rank):
        try:
            comm = Comm(rank, world, device=0, backend="shm", key=key)
            rr = ShardedRouter(to, params, rank=rank, world=world, device=0, assume_short_ts=True, cost_hint=hint)
            rr.enable_device_exchange(comm)
            rr.upload(nsteps, qlat_s, q0)
            rr.upload_trunk()
            rr.route_on_device(qts, True, None)
            rr.upload(nsteps, qlat_a, None)
            rr.route_on_device(qts, True, None)
            rr.upload(nsteps, qlat_b, None)
            rows, hyd = rr.route_on_device(qts, True, None)
            results[rank] = (rows, hyd.numpy())
            rr.close()
            comm.close()
        except Exception as e:                          # pragma: no cover
            errors.append(e)
    ts = [threading.Thread(target=run, args=(k,)) for k in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    outlets = np.flatnonzero(to < 0)
    for rows, hyd in results:
        assert np.array_equal(np.sort(rows), outlets)
        assert_bit_identical(hyd, ref["q"][rows, 1:], "outlet hydrographs gathered by the two-rank job")


def test_conus_general_mode_every_segment_bit_identical_to_reference(conus):
    """The GENERAL mode (assume_short_ts=False, the reference's configuration default, compute_parameters.py:47) at full
    size: every one of the 2 729 077 segments through a 288-step day from the cold start on the dataflow engine -- on the
    plan built for that mode, i.e. with the dominant basin's 2 700-row stem laid out last, its side tributaries from the top
    down, and its blocks started first (topology.hpp, stem_min_rows) -- against the reference Fortran on the CPU over the
    reference's ordered sub-networks (loop semantics without the short-timestep assumption: mc_reach.pyx:499-505): the flow
    of every row at every step, the velocity and depth series (exact checksums), the final state."""
    from troute_amd.plan import topology_blocks_general
    net, up_ptr, up_idx = conus
    to, params, qlat = net["to"], net["params"], net["qlat"]
    nseg = to.shape[0]
    nsteps, qts = 288, 12
    q0 = np.zeros((nseg, 3), np.float32)
    assert topology_blocks_general(up_ptr, up_idx, None, 1024)[4].size >= 8      # the long stems' blocks, started first
    ref = O.reference_windows(to, params, (qlat,), q0, nsteps, qts, False)
    with RoutingPlan(up_ptr, up_idx, params, assume_short_ts=False) as plan:
        assert plan.engine == "flow"
        plan.upload_forcing(nsteps, qlat, q0)
        plan.route_device(nsteps, qts, False)
        fvd = plan.download_fvd().reshape(nseg, nsteps, 3)
        final = plan.download_final_state()
    for lo in range(0, nseg, 200000):
        assert_bit_identical(np.ascontiguousarray(fvd[lo:lo + 200000, :, 0]), ref["q"][lo:lo + 200000, 1:],
                             f"general mode: flow of every row, rows {lo}..")
    assert np.array_equal(O.series_checksum(fvd[:, :, 1]), ref["chk_v"]), "general mode: velocity series of some row differs"
    assert np.array_equal(O.series_checksum(fvd[:, :, 2]), ref["chk_d"]), "general mode: depth series of some row differs"
    assert_bit_identical(final, ref["state"], "general mode: final state of every row")


def test_conus_row_relabelling_invariance(conus):
    """Permuting the caller's row labels must permute the result and change nothing else (bitwise):
    exercises the level flattening, the gather maps and the result transpose at full width."""
    net, up_ptr, up_idx = conus
    to = net["to"]
    nseg = to.shape[0]
    nsteps, qts = 24, 12
    q0 = np.zeros((nseg, 3), np.float32)
    with RoutingPlan(up_ptr, up_idx, net["params"]) as plan:
        plan.upload_forcing(nsteps, net["qlat"], q0)
        plan.route_device(nsteps, qts, False)
        a = plan.download_final_state()
    rng = np.random.default_rng(5)
    perm = rng.permutation(nseg)                     # new row of old row r = perm[r]
    inv = np.empty(nseg, np.int64)
    inv[perm] = np.arange(nseg)
    from troute_amd.distributed import restrict_csr
    p2, i2 = restrict_csr(up_ptr, up_idx, inv, perm)  # same upstream lists (same summation order), new labels
    with RoutingPlan(p2, i2, net["params"][inv]) as plan:
        plan.upload_forcing(nsteps, net["qlat"][inv], q0)
        plan.route_device(nsteps, qts, False)
        b = plan.download_final_state()
    assert_bit_identical(a, b[perm], "relabelled CONUS")


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("short", [True, False])
def test_cost_hinted_plan_order_changes_nothing(short, engine, monkeypatch):
    """A plan built with a cost hint (rows of a level -- or of a block, on the dataflow engine -- grouped by the secant
    iterations they needed in an earlier window) visits the rows in another order and must produce the same bits, which
    are the ORACLE's: random hints, the plan's own iteration counts as hint; both engines."""
    set_engine(monkeypatch, engine)
    rng = np.random.default_rng(4242)
    nseg = 90000
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    params, qlat, q0 = synth_inputs(rng, nseg, 4)
    nsteps, qts = 24, 6
    lvl, _, _ = topology_levels(up_ptr, up_idx)
    want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, short, det=True)[:, 1:, :]
    with RoutingPlan(up_ptr, up_idx, params) as plan:
        base = plan.route(nsteps, qts, short, qlat, q0)
        own = plan.download_iterations()
    assert_bit_identical(base, want, f"unhinted plan vs oracle short={short} engine={engine}")
    assert own.max() >= 2 and (own == 0).any()
    for hint in (rng.integers(0, 4, nseg).astype(np.uint8), np.minimum(own, 3)):
        # (assume_short_ts given: the dataflow engine then sorts its block order by cost tiers as well)
        with RoutingPlan(up_ptr, up_idx, params, cost_hint=hint, assume_short_ts=short) as plan:
            got = plan.route(nsteps, qts, short, qlat, q0)
            assert np.array_equal(plan.download_iterations(), own)
        assert_bit_identical(got, base, f"hinted plan short={short}")
    with pytest.raises(ValueError):
        RoutingPlan(up_ptr, up_idx, params, cost_hint=np.zeros(3, np.uint8))
    # cost collection: per row the sum over the window of min(iterations, 3) (+ 4 per over-bank step); same flows with it on
    with RoutingPlan(up_ptr, up_idx, params) as plan:
        with pytest.raises(RuntimeError):
            plan.download_cost()
        plan.collect_cost(True)
        got = plan.route(nsteps, qts, short, qlat, q0)
        cost, n = plan.download_cost()
        last = plan.download_iterations()
        plan.collect_cost(False)
        with pytest.raises(RuntimeError):
            plan.download_cost()
    assert_bit_identical(got, base, "cost collection on")
    assert n == nsteps and cost.dtype == np.uint16 and cost.max() <= 7 * nsteps
    assert (cost >= np.minimum(last, 3)).all() and (cost[last >= 2] >= 2).all() and (cost == 0).any()


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("short", [True, False])
def test_extreme_parameters_and_depths_around_the_fast_division_guard(short, engine, monkeypatch):
    """The hydraulic point drops the scaling / fix-up steps of its divisions when a plan-wide parameter check and a
    per-call depth test hold (DevMathF::fast_ok, trmc.hip); the oracle always divides plainly.  Forests with parameters
    log-uniform over the whole admitted range [2**-14, 2**17] (twcc / ncc sometimes 0) and initial depths from 1e-12 to
    1e4 -- on both sides of the depth test -- must agree bit for bit; one parameter outside the range switches the
    plan to plain divisions, same results.  Both engines."""
    set_engine(monkeypatch, engine)
    rng = np.random.default_rng(77)
    nseg = 70000
    to = H.random_network(rng, nseg)
    _, _, ups = H.reaches_from_to(to)
    up_ptr, up_idx = csr_from_lists(ups)
    lvl, _, _ = topology_levels(up_ptr, up_idx)

    def logu(lo, hi, n):
        return np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    lo, hi = 2.0 ** -14, 2.0 ** 17
    bw = logu(lo * 1.01, 300.0, nseg)
    tw = bw * rng.uniform(1.0, 3.0, nseg)
    twcc = np.where(rng.random(nseg) < 0.2, 0.0, tw * rng.uniform(1.0, 4.0, nseg))
    n = logu(lo * 1.01, 0.5, nseg)
    ncc = np.where(rng.random(nseg) < 0.1, 0.0, n * rng.uniform(1.0, 3.0, nseg))
    cs = logu(0.02, 50.0, nseg)
    params = np.stack([np.full(nseg, 300.0), logu(10.0, 5e4, nseg), bw, tw, twcc, n, ncc, cs, logu(1e-5, 1.0, nseg)], 1)
    params = params.astype(np.float32)
    qlat = logu(1e-9, 5.0, (nseg, 3)).astype(np.float32) * (rng.random((nseg, 3)) > 0.2)
    q0 = np.stack([logu(1e-9, 50.0, nseg), logu(1e-9, 50.0, nseg), logu(1e-12, 1e4, nseg)], 1).astype(np.float32)
    nsteps, qts = 12, 4
    want = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params, q0, qlat, short, det=True)[:, 1:, :]
    fin = np.isfinite(want).all(axis=(1, 2))
    assert fin.mean() > 0.99
    with RoutingPlan(up_ptr, up_idx, params) as plan:
        got = plan.route(nsteps, qts, short, qlat, q0)
    assert_bit_identical(got[fin], want[fin], f"extreme parameters short={short}")
    assert (np.isfinite(got).all(axis=(1, 2)) == fin).all()
    # one row outside the admitted range: the whole plan divides plainly
    params2 = params.copy()
    params2[0, 2] = np.float32(2.0 ** 18)
    params2[0, 3] = np.float32(2.0 ** 19)
    want2 = O.network_by_segment(nsteps, qts, up_ptr, up_idx, lvl, params2, q0, qlat, short, det=True)[:, 1:, :]
    fin2 = np.isfinite(want2).all(axis=(1, 2))
    with RoutingPlan(up_ptr, up_idx, params2) as plan:
        got2 = plan.route(nsteps, qts, short, qlat, q0)
    assert_bit_identical(got2[fin2], want2[fin2], "plain divisions")


def test_conus_cost_hint_from_a_tuning_window(conus):
    """The bench's sequence at full size: route, take the iteration hint, rebuild the router with it, route again --
    same outlet hydrographs, bit for bit."""
    from troute_amd.distributed import ShardedRouter
    net, _, _ = conus
    to, params, qlat = net["to"], net["params"], net["qlat"]
    nseg = to.shape[0]
    q0 = np.zeros((nseg, 3), np.float32)
    nsteps, qts = 48, 12
    r = ShardedRouter(to, params)
    r.upload(nsteps, qlat, q0)
    r.collect_cost(True)
    rows = r.route_resident(qts, True)
    a = r.outlet_hydrographs()
    hint = r.iteration_hint()
    r.close()
    # sixteenths of the mean of min(iterations, 3) + 4 * (over-bank step): at most 16 * 7
    assert hint.shape == (nseg,) and 32 <= hint.max() <= 112 and len(np.unique(hint)) > 8
    r = ShardedRouter(to, params, cost_hint=hint)
    r.upload(nsteps, qlat, q0)
    rows2 = r.route_resident(qts, True)
    b = r.outlet_hydrographs()
    r.close()
    assert np.array_equal(rows, rows2)
    assert_bit_identical(a, b, "CONUS outlets, hinted plan")


def test_resident_warm_start_between_windows(lc):
    """Long runs are chunked into windows warm-started from new_q0 = fvd[:, [-3,-3,-1]]
    (AbstractNetwork.py:177-191).  Keeping that state in HBM (q0 = None) is bit-identical to
    downloading it and uploading it again, and two 144-step windows reproduce one 288-step window
    when the forcing columns line up."""
    from troute_amd.routing.fast_reach.mc_reach import _flatten_network
    up_ptr, up_idx, _ = _flatten_network([(r, 0) for r in lc.reaches], lc.rconn, lc.ids)
    for short in (True, False):
        with RoutingPlan(up_ptr, up_idx, lc.params9) as plan:
            whole = plan.route(288, 12, short, lc.qlat, lc.q0)
            a1 = plan.route(144, 12, short, lc.qlat[:, :12], lc.q0)
            state = plan.download_final_state()
            plan.upload_forcing(144, lc.qlat[:, 12:24], None)          # resident warm start
            plan.route_device(144, 12, short)
            a2 = plan.download_fvd()
            b2 = plan.route(144, 12, short, lc.qlat[:, 12:24], state)   # host round trip
        assert_bit_identical(a2, b2, "resident vs re-uploaded warm start")
        assert_bit_identical(np.concatenate([a1, a2], 1), whole, f"two windows vs one (short={short})")


def test_short_exact_forms_equal_the_operations_they_stand_for():
    """The fp32 step takes short forms of sqrt, division and max where its range proofs hold (csrc/trmc.hip, DevMathF:
    sqrt_r -- v_sqrt_f32 and its two residual corrections without the scaling and the special-value pass-through;
    k_of -- the refinement of a reciprocal without v_div_scale / v_div_fmas / v_div_fixup; max_num -- v_max).  On the
    device itself: sqrt over EVERY float of the admitted range [2**-60, 2**63] and 2**32 pseudo-random (dx, celerity, dt)
    triples over the whole admitted exponent ranges, ends included, must not differ from sqrtf, `/` and the select in one
    bit."""
    import ctypes as C
    lib = _lib.lib()
    checked, bad = C.c_int64(0), C.c_int64(-1)
    _lib.check(lib.trmc_selfcheck_fast_arith(0, 0, 0, 0, C.byref(checked), C.byref(bad)))
    assert checked.value == (123 << 23) + 1 and bad.value == 0, (checked.value, bad.value)
    for seed in (1, 0x9e3779b97f4a7c15):
        _lib.check(lib.trmc_selfcheck_fast_arith(0, 1, 1 << 32, seed, C.byref(checked), C.byref(bad)))
        assert checked.value == 1 << 32 and bad.value == 0, (seed, checked.value, bad.value)
