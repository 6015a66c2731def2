"""Diffusive-wave mainstem solver.

Parity chain:  reference Fortran diffnw built from its own sources (oracle/_ref/libdiff_ref.so; goldens in
tests/golden/diffusive_*.npz, inputs marshalled by the reference's own diffusive_input_data_v02)
  == host instantiation of t-route_amd/csrc/diffusive_core.hpp (oracle/libdw_oracle.so), BITWISE
  == GPU (trdw_diffnw through troute_amd.routing.fast_reach.diffusive.compute_diffusive), BITWISE,
the last link resting on det_pow64.h == libm pow (checked here on random arguments)."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARG_ORDER = ["timestep_ar_g", "nts_ql_g", "nts_ub_g", "nts_db_g", "ntss_ev_g", "nts_qtrib_g", "nts_da_g", "mxncomp_g",
             "nrch_g", "z_ar_g", "bo_ar_g", "traps_ar_g", "tw_ar_g", "twcc_ar_g", "mann_ar_g", "manncc_ar_g", "so_ar_g",
             "dx_ar_g", "iniq", "frnw_col", "frnw_g", "qlat_g", "ubcd_g", "dbcd_g", "qtrib_g", "paradim", "para_ar_g",
             "mxnbathy_g", "x_bathy_g", "z_bathy_g", "mann_bathy_g", "size_bathy_g", "usgs_da_g", "usgs_da_reach_g",
             "rdx_ar_g", "cwnrow_g", "cwncol_g", "crosswalk_g", "z_thalweg_g"]
INT_SCALARS = {"nts_ql_g", "nts_ub_g", "nts_db_g", "ntss_ev_g", "nts_qtrib_g", "nts_da_g", "mxncomp_g", "nrch_g", "frnw_col",
               "paradim", "mxnbathy_g", "cwnrow_g", "cwncol_g"}
INT_ARRAYS = {"frnw_g", "size_bathy_g", "usgs_da_reach_g"}
SMALL = ("chain1", "y3", "comb", "y3_nat", "comb_nat")       # *_nat: natural (bathymetry) cross sections


def load_small(name):
    z = np.load(os.path.join(H.GOLDEN, "diffusive_small.npz"))
    d = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(name + "__")}
    return {k[3:]: v for k, v in d.items() if k.startswith("in_")}, (d["out_q"], d["out_elv"], d["out_depth"])


def load_crosswalk(name):
    """refactored-hydrofabric cases: results mapped back through a crosswalk (tests/golden/make_diffusive_fixtures.py
    --crosswalk; outputs of the reference Fortran)"""
    z = np.load(os.path.join(H.GOLDEN, "diffusive_crosswalk.npz"))
    d = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(name + "__")}
    return {k[3:]: v for k, v in d.items() if k.startswith("in_")}, (d["out_q"], d["out_elv"], d["out_depth"])


def load_lowercolorado(nsteps=None, fixture="diffusive_lowercolorado.npz"):
    z = np.load(os.path.join(H.GOLDEN, fixture))
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    want = (z["out_q"], z["out_elv"], z["out_depth"])
    if nsteps is not None:
        # A shorter window of the same run (the 72-step golden took the reference Fortran 64 s; the suite stays short).
        # Records 0 .. nsteps-1 are those of the long run bit for bit; the LAST record of a window is reached by a
        # sub-step truncated at tfin (calculateDT, diffusive.f90:986), whose length differs in the last bit from the
        # save-interval formula the long run used there -- it is compared to 1e-13 instead.
        ins["timestep_ar_g"] = ins["timestep_ar_g"].copy()
        ins["timestep_ar_g"][2] = 300.0 * nsteps / 3600.0
        ins["ntss_ev_g"] = np.array(nsteps + 1)
        want = tuple(w[:nsteps + 1] for w in want)
    return ins, want


def check_window(got, want):
    for g, w in zip(got, want):
        assert same_bits(g[:-1], w[:-1])
        assert np.allclose(g[-1], w[-1], rtol=1e-13, atol=1e-13)


def call_c(lib, sym, ins):
    keep, args = [], []
    for k in ARG_ORDER:
        v = ins[k]
        if k in INT_SCALARS:
            c = C.c_int(int(v))
            keep.append(c)
            args.append(C.byref(c))
        else:
            a = np.asfortranarray(v, dtype=np.int32 if k in INT_ARRAYS else np.float64)
            if a.size == 0:
                a = np.zeros(1, dtype=a.dtype)
            keep.append(a)
            args.append(a.ctypes.data_as(C.c_void_p))
    shape = (int(ins["ntss_ev_g"]), int(ins["mxncomp_g"]), int(ins["nrch_g"]))
    outs = [np.zeros(shape, dtype=np.float64, order="F") for _ in range(3)]
    args += [o.ctypes.data_as(C.c_void_p) for o in outs]
    rc = getattr(lib, sym)(*args)
    return rc, outs


def host_oracle():
    from oracle import oracle as O
    O.build()
    return C.CDLL(os.path.join(ROOT, "oracle", "libdw_oracle.so"))


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


@pytest.mark.parametrize("name", SMALL)
def test_host_restatement_equals_reference_fortran_bitwise_small(name):
    ins, want = load_small(name)
    rc, got = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0
    for g, w in zip(got, want):
        assert np.abs(w).max() > 0.1 and same_bits(g, w)


def test_host_restatement_equals_reference_fortran_bitwise_lowercolorado():
    """The coastal subset of the shipped hybrid configuration: 230 reaches (115 diffusive), 24 of its 72 steps
    (960 sub-steps) -- every recorded flow, elevation and depth to the last bit."""
    ins, want = load_lowercolorado(24)
    rc, got = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0
    assert int((ins["frnw_g"] == 555).sum()) == 115
    check_window(got, want)
    assert want[0].max() > 0.3 and want[2].max() > 0.1


def test_host_restatement_equals_reference_fortran_bitwise_natural_sections_and_coastal_depth():
    """The same domain with a station table per mainstem segment (natural cross sections, 9-17 stations) and a
    prescribed water depth at the tailwater (boundary option 1) -- the v4 hybrid configuration's switches; the whole
    36-step run of the reference Fortran, to the last bit."""
    ins, want = load_lowercolorado(None, "diffusive_lowercolorado_nat.npz")
    assert int(ins["mxnbathy_g"]) == 17 and ins["para_ar_g"][10] == 1.0 and ins["size_bathy_g"].max() == 17
    rc, got = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0
    for g, w in zip(got, want):
        assert same_bits(g, w)
    assert want[2].max() > 2.0


def lowercolorado_diffusive_network(fixture="diffusive_lowercolorado.npz"):
    """The diffusive domain of the shipped hybrid configuration, built with THIS package's graph utilities the way
    AbstractRouting.py:255-310 builds it; raw tables from the fixtures."""
    import pandas as pd
    from functools import partial
    from troute_amd import nhd_network as nn
    z = np.load(os.path.join(H.GOLDEN, fixture))
    lc = H.LowerColorado()
    tw = int(z["tw"])
    mainstem, trib = [int(x) for x in z["mainstem"]], [int(x) for x in z["trib"]]
    conn_all = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    connections = {k: conn_all[k] for k in (mainstem + trib)}
    connections[tw] = []
    rconn = nn.reverse_network(connections)
    net = nn.reachable_network(rconn)
    reaches = nn.dfs_decomposition(net[tw], partial(nn.split_at_waterbodies_and_junctions, set(trib), net[tw]))
    cols = {c: lc.params9[:, i] for i, c in enumerate(("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"))}
    param_df = pd.DataFrame(cols, index=lc.ids)
    param_df["alt"] = z["alt"]
    dn = {"connections": connections, "rconn": rconn, "reaches": reaches, "mainstem_segs": mainstem,
          "tributary_segments": trib, "param_df": param_df.loc[mainstem + trib]}
    qlat_df = pd.DataFrame(lc.qlat, index=lc.ids)
    q0 = pd.DataFrame(lc.q0, index=lc.ids, columns=["qu0", "qd0", "h0"])
    return z, lc, tw, dn, qlat_df, q0


def test_input_marshalling_equals_reference_dictionary():
    """troute_amd.routing.diffusive_utils_v02.diffusive_input_data_v02 against the dictionary the reference's own
    function produced for the same network (every array of the c_diffnw call, bit for bit), and unpack_output against
    the reference's unpacking of the reference outputs."""
    import pandas as pd
    from troute_amd.routing import diffusive_utils_v02 as DU
    z, lc, tw, dn, qlat_df, q0 = lowercolorado_diffusive_network()
    nsteps = int(z["in_ntss_ev_g"]) - 1
    junction_inflows = pd.DataFrame(z["junction_inflows"], index=dn["tributary_segments"])
    ins = DU.diffusive_input_data_v02(
        tw, dn["connections"], dn["rconn"], dn["reaches"], dn["mainstem_segs"], dn["tributary_segments"], None,
        dn["param_df"], qlat_df, q0, junction_inflows, lc.qts, pd.Timestamp("2021-08-23 13:00"), nsteps, lc.dt,
        pd.DataFrame(), pd.DataFrame(), pd.DataFrame(), None, None, pd.DataFrame(), pd.DataFrame())
    for k in ARG_ORDER:
        want = z["in_" + k]
        got = np.asarray(ins[k])
        if k in INT_SCALARS:
            assert int(got) == int(want), k
        else:
            assert got.shape == want.shape or got.size == want.size == 0, k
            assert np.array_equal(got.astype(want.dtype), want), k
    ids, dat = DU.unpack_output(ins["pynw"], ins["ordered_reaches"], z["out_q"], z["out_depth"])
    assert np.array_equal(ids, z["unpacked_ids"])
    assert dat.dtype == np.float32 and np.array_equal(dat, z["unpacked_dat"], equal_nan=True)
    with pytest.raises(NotImplementedError, match="refactored"):
        DU.diffusive_input_data_v02(tw, dn["connections"], dn["rconn"], dn["reaches"], dn["mainstem_segs"],
                                    dn["tributary_segments"], None, dn["param_df"], qlat_df, q0, junction_inflows, lc.qts,
                                    None, nsteps, lc.dt, pd.DataFrame(), pd.DataFrame(), pd.DataFrame(), {"links": [1]}, None,
                                    pd.DataFrame(), pd.DataFrame())
    # a refactored domain that brings the solver's crosswalk arguments: forwarded into the dictionary (compute.py:1785-1812
    # hands the domain over per tailwater; the reference's own marshalling leaves these arguments empty, :1033-1038)
    cw = np.arange(3 * 18, dtype=np.float64).reshape(3, 18)
    rdom = {"rlinks": dn["mainstem_segs"], "refac_tw": tw, "rdx_ar_g": np.ones((4, 2)), "crosswalk_g": cw,
            "z_thalweg_g": np.zeros((4, 2))}
    ins2 = DU.diffusive_input_data_v02(
        tw, dn["connections"], dn["rconn"], dn["reaches"], dn["mainstem_segs"], dn["tributary_segments"], None,
        dn["param_df"], qlat_df, q0, junction_inflows, lc.qts, pd.Timestamp("2021-08-23 13:00"), nsteps, lc.dt,
        pd.DataFrame(), pd.DataFrame(), pd.DataFrame(), rdom, dn["reaches"], pd.DataFrame(), pd.DataFrame())
    assert ins2["cwnrow_g"] == 3 and ins2["cwncol_g"] == 18 and np.array_equal(ins2["crosswalk_g"], cw)
    assert ins2["rdx_ar_g"].shape == (4, 2) and np.array_equal(ins2["z_ar_g"], ins["z_ar_g"])


def gage_table_case():
    """The LowerColorado domain with the gage table of tests/golden/diffusive_da.npz handed to THIS package's marshalling."""
    import pandas as pd
    from troute_amd.routing import diffusive_utils_v02 as DU
    z, lc, tw, dn, qlat_df, q0 = lowercolorado_diffusive_network()
    g = np.load(os.path.join(H.GOLDEN, "diffusive_da.npz"))
    usgs_df = pd.DataFrame(g["usgs_values"], index=g["usgs_index"], columns=pd.to_datetime(g["usgs_times"]))
    nsteps = int(g["in_nts_da_g"]) - 1
    junction_inflows = pd.DataFrame(g["junction_inflows"], index=dn["tributary_segments"])
    ins = DU.diffusive_input_data_v02(
        tw, dn["connections"], dn["rconn"], dn["reaches"], dn["mainstem_segs"], dn["tributary_segments"], None,
        dn["param_df"], qlat_df, q0, junction_inflows, lc.qts, pd.Timestamp("2021-08-23 13:00"), nsteps, lc.dt,
        pd.DataFrame(), pd.DataFrame(), usgs_df, None, None, pd.DataFrame(), pd.DataFrame())
    return ins, g, usgs_df


def test_gage_table_is_marshalled_like_the_reference_and_changes_nothing_in_the_solver():
    """``usgs_df`` reaches the marshalling whenever the DA dictionary has the key (compute.py:1798-1803; the reference's
    dictionary ALWAYS has it, DataAssimilation.py:86): fp_da_map's three arguments equal the reference's (a reach with two
    gaged nodes, a NaN record, stamps missing at the window's end, an id outside the domain), and the solver -- the
    reference Fortran in the fixture, the host restatement here -- gives the bits it gives without them
    (diffusive.f90:1282-1303: the branch is commented out)."""
    ins, g, usgs_df = gage_table_case()
    assert int(ins["nts_da_g"]) == int(g["in_nts_da_g"])
    assert ins["usgs_da_g"].dtype == np.float64 and same_bits(ins["usgs_da_g"], g["in_usgs_da_g"])
    assert np.array_equal(ins["usgs_da_reach_g"], g["in_usgs_da_reach_g"]) and ins["usgs_da_reach_g"].dtype == np.int32
    assert (ins["usgs_da_reach_g"] != 0).sum() >= 2 and (ins["usgs_da_g"] == -4444.0).any() and (ins["usgs_da_g"] > 0).any()
    rc, got = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0
    for gg, name in zip(got, ("out_q", "out_elv", "out_depth")):
        assert same_bits(gg, g[name]), name
    plain = dict(ins)
    plain["usgs_da_g"] = np.full_like(ins["usgs_da_g"], -4444.0)
    plain["usgs_da_reach_g"] = np.zeros_like(ins["usgs_da_reach_g"])
    rc, got0 = call_c(host_oracle(), "dw_oracle_diffnw", plain)
    assert rc == 0 and all(same_bits(a, b) for a, b in zip(got, got0))


@pytest.mark.gpu
def test_gpu_with_gage_table_equals_reference_fortran_bitwise():
    from troute_amd.routing.fast_reach import diffusive as D
    ins, g, _ = gage_table_case()
    got = D.compute_diffusive(ins)
    for gg, name in zip(got, ("out_q", "out_elv", "out_depth")):
        assert same_bits(gg, g[name]), name


def natural_and_coastal_tables(z):
    import pandas as pd
    topo = pd.DataFrame({"xid_d": z["topo_xid_d"], "z": z["topo_z"], "n": z["topo_n"]},
                        index=pd.Index(z["topo_index"], name="comid"))
    coast = pd.DataFrame(z["coast_values"], index=z["coast_index"], columns=pd.to_datetime(z["coast_times"]))
    return topo, coast


def test_input_marshalling_natural_sections_and_coastal_depth_equal_reference_dictionary():
    """fp_naturalxsec_map and fp_coastal_boundary_input_map of the mirror against the dictionary the reference made
    from the same station table and depth series: a non-positive depth replaced, a gap bridged, option 1 chosen."""
    import pandas as pd
    from troute_amd.routing import diffusive_utils_v02 as DU
    z, lc, tw, dn, qlat_df, q0 = lowercolorado_diffusive_network("diffusive_lowercolorado_nat.npz")
    topo, coast = natural_and_coastal_tables(z)
    assert np.isnan(coast.values).sum() == 1 and (coast.values <= 0).sum() == 1
    nsteps = int(z["in_ntss_ev_g"]) - 1
    junction_inflows = pd.DataFrame(z["junction_inflows"], index=dn["tributary_segments"])
    t0 = pd.Timestamp("2021-08-23 13:00")
    a = (tw, dn["connections"], dn["rconn"], dn["reaches"], dn["mainstem_segs"], dn["tributary_segments"], None,
         dn["param_df"], qlat_df, q0, junction_inflows, lc.qts, t0, nsteps, lc.dt, pd.DataFrame())
    ins = DU.diffusive_input_data_v02(*a, topo, pd.DataFrame(), None, None, coast, pd.DataFrame())
    for k in ARG_ORDER:
        want = z["in_" + k]
        got = np.asarray(ins[k])
        if k in INT_SCALARS:
            assert int(got) == int(want), k
        else:
            assert got.shape == want.shape or got.size == want.size == 0, k
            assert np.array_equal(got.astype(want.dtype), want), k
    # a row that stays incomplete anywhere in the table sends the domain back to the normal-depth boundary (:642-645)
    holes = coast.copy()
    holes.iloc[1, :] = np.nan
    ins = DU.diffusive_input_data_v02(*a, topo, pd.DataFrame(), None, None, holes, pd.DataFrame())
    assert ins["para_ar_g"][10] == 2.0 and not ins["dbcd_g"].any() and len(ins["dbcd_g"]) == len(z["in_dbcd_g"])
    # the newer column vocabulary of the station table (:486-489)
    newer = topo.rename(columns={"xid_d": "relative_dist", "z": "Z", "n": "roughness"}).assign(cs_id=0)
    ins2 = DU.diffusive_input_data_v02(*a, newer, pd.DataFrame(), None, None, coast, pd.DataFrame())
    assert np.array_equal(ins2["z_bathy_g"], z["in_z_bathy_g"]) and np.array_equal(ins2["mann_bathy_g"], z["in_mann_bathy_g"])


def test_det_pow64_equals_libm_pow():
    """The restated glibc pow (det_pow64.h) against this machine's libm at the solver's exponents."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <math.h>
#include "det_pow64.h"
static unsigned long long s = 88172645463325252ull;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
int main(void)
{
    const float ef[6] = {0.3f, 0.4f, 0.6f, 2.f / 3.f, 5.f / 3.f, 1.5f};   /* 1.5: natural-section Manning weights */
    long bad = 0;
    volatile double probe = 6.1817510939031299e-05;      /* libm's result here is not the correctly rounded one */
    double a = pow(probe, (double)0.4f), c = det_pow64(probe, (double)0.4f);
    bad += memcmp(&a, &c, 8) != 0;
    const double edge[] = {0.0, 1.0, 2.2250738585072014e-308, 4.9406564584124654e-324, 1e-310, 1.7976931348623157e308, 0.5, 2.0};
    for (int e = 0; e < 6; ++e) {
        const double y = (double)ef[e];
        for (unsigned k = 0; k < sizeof edge / sizeof edge[0]; ++k) {
            volatile double x = edge[k];
            a = pow(x, y); c = det_pow64(x, y);
            bad += memcmp(&a, &c, 8) != 0;
        }
        for (long i = 0; i < 600000; ++i) {
            volatile double x = exp((-40.0 + 60.0 * rnd()) * 0.6931471805599453);
            a = pow(x, y); c = det_pow64(x, y);
            bad += memcmp(&a, &c, 8) != 0;
        }
    }
    printf("%ld\n", bad);
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "t-route_amd", "csrc"), "-o",
                               os.path.join(td, "t"), os.path.join(td, "t.c"), "-lm"])
        out = subprocess.run([os.path.join(td, "t")], capture_output=True, text=True, check=True).stdout
    assert int(out.strip()) == 0


def test_python_mirror_refuses_what_is_not_covered_and_has_no_cpu_fallback():
    from troute_amd import _lib
    from troute_amd.routing.fast_reach import diffusive as D
    ins, _ = load_small("chain1")
    if _lib.device_count() == 0:
        with pytest.raises(RuntimeError, match="no HIP device"):
            D.compute_diffusive(ins)
        return
    bad = dict(load_small("y3_nat")[0])
    bad["size_bathy_g"] = np.where(bad["size_bathy_g"] > 0, 1, 0).astype(bad["size_bathy_g"].dtype)
    with pytest.raises(ValueError, match="fewer than two bathymetry stations"):
        D.compute_diffusive(bad)
    bad = dict(ins)
    bad["cwnrow_g"] = np.array(2)                      # crosswalk rows announced, none supplied
    with pytest.raises(ValueError, match="crosswalk"):
        D.compute_diffusive(bad)
    bad = dict(load_crosswalk("y3_cw")[0])
    bad["crosswalk_g"] = bad["crosswalk_g"].copy()
    bad["crosswalk_g"][3, 3] = 99                      # an original link outside the arrays
    with pytest.raises(ValueError, match="out of range"):
        D.compute_diffusive(bad)
    bad = dict(ins)
    bad["frnw_g"] = np.where(ins["frnw_g"] == 555, -555, ins["frnw_g"])
    with pytest.raises(ValueError, match="no mainstem"):
        D.compute_diffusive(bad)


@pytest.mark.parametrize("name", ("y3_cw", "y3_nat_cw"))
def test_host_restatement_crosswalk_equals_reference_fortran_bitwise(name):
    """diffnw :849-920: results of a refactored hydrofabric mapped back to the original links -- every branch of the
    mapping (links covered at once, in two and in three parts, a segment over two links, the 0.99 threshold)."""
    ins, want = load_crosswalk(name)
    rc, got = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0
    for g, w in zip(got, want):
        assert same_bits(np.asarray(g), w)
    assert (want[0] != 0).sum() == 11 * want[0].shape[0]      # 11 mapped node cells per recording instant, the rest zeroed


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("y3_cw", "y3_nat_cw"))
def test_gpu_crosswalk_equals_reference_fortran_bitwise(name):
    from troute_amd.routing.fast_reach import diffusive as D
    ins, want = load_crosswalk(name)
    got = D.compute_diffusive(ins)
    for g, w in zip(got, want):
        assert same_bits(g, w)
    many = D.compute_diffusive_batch([ins, load_small("y3")[0], ins])    # mapped and unmapped domains in one launch
    assert all(same_bits(g, w) for g, w in zip(many[2], want)) and same_bits(many[1][0], load_small("y3")[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL)
def test_gpu_equals_reference_fortran_bitwise_small(name):
    from troute_amd.routing.fast_reach import diffusive as D
    ins, want = load_small(name)
    got = D.compute_diffusive(ins)
    for g, w in zip(got, want):
        assert g.shape == w.shape and g.flags["C_CONTIGUOUS"] and same_bits(g, w)


@pytest.mark.gpu
def test_gpu_equals_reference_fortran_bitwise_lowercolorado():
    from troute_amd.routing.fast_reach import diffusive as D
    ins, want = load_lowercolorado(24)
    got = D.compute_diffusive(ins)
    check_window(got, want)
    rc, host = call_c(host_oracle(), "dw_oracle_diffnw", ins)     # and the whole window, last record included,
    for g, h in zip(got, host):                                       # equals the host instantiation bit for bit
        assert rc == 0 and same_bits(g, h)
    tables_ms, solve_ms = D.last_timing()
    assert tables_ms > 0 and solve_ms > 0


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"TRDW_SOLVER": "serial"}, {"TRDW_CHAIN_GLOBAL": "1"}, {"TRDW_WINDOW_ROWS": "5"}],
                         ids=["one-wavefront kernel", "chain state in global memory", "5-row windows"])
def test_gpu_solver_variants_give_the_same_bits(env, monkeypatch):
    """The parallel time loop's other forms -- the serial kernel it replaced, the chain state in global memory (domains
    too long for LDS), narrow windows (look-ups that leave the window take the full-column search) -- on the
    LowerColorado golden and a small natural-section case: bit for bit the reference Fortran."""
    from troute_amd.routing.fast_reach import diffusive as D
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ins, want = load_lowercolorado(6)
    check_window(D.compute_diffusive(ins), want)
    ins, want = load_small("comb_nat")
    for g, w in zip(D.compute_diffusive(ins), want):
        assert same_bits(g, w)


@pytest.mark.gpu
def test_gpu_equals_reference_fortran_bitwise_natural_sections_and_coastal_depth():
    from troute_amd.routing.fast_reach import diffusive as D
    ins, want = load_lowercolorado(12, "diffusive_lowercolorado_nat.npz")
    got = D.compute_diffusive(ins)
    check_window(got, want)
    assert want[2].max() > 1.5


@pytest.mark.gpu
def test_gpu_hybrid_coupling_tributary_flows():
    """The hybrid configuration end to end on the device: the tributary hydrographs the diffusive solver is fed
    (qtrib_g of the golden, made from the reference-equivalent MC flows) are, bit for bit, what the GPU MC engine
    produces for the tributary segments -- so MC on the GPU followed by trdw_diffnw reproduces the reference's
    MC + diffnw chain (compute.py:1766-1781 hands over results[...][x, ::3])."""
    from troute_amd.plan import RoutingPlan
    z = np.load(os.path.join(H.GOLDEN, "diffusive_lowercolorado.npz"))
    trib = z["trib"]
    qtrib = z["in_qtrib_g"]                       # [nts_qtrib, nrch]; row 0 = initial condition, rows 1.. = MC flows
    lc = H.LowerColorado()
    nsteps = qtrib.shape[0] - 1
    up_ptr, up_idx = lc.csr()
    with RoutingPlan(up_ptr, up_idx, lc.params9) as plan:
        fvd = plan.route(nsteps, lc.qts, True, lc.qlat, lc.q0)
    row = {int(s): i for i, s in enumerate(lc.ids)}
    flows = fvd[[row[int(s)] for s in trib], :, 0].astype(np.float64)          # junction_inflows as float64
    cols = {qtrib[1:, j].tobytes() for j in range(qtrib.shape[1])}
    assert len(trib) == 115 and all(f.tobytes() in cols for f in flows) and np.abs(flows).max() > 0.1


@pytest.mark.gpu
def test_gpu_hybrid_driver_mc_then_diffusive():
    """compute_nhd_routing_v02 (MC, GPU) -> compute_diffusive_routing (marshalling mirror + GPU solver) on the
    LowerColorado hybrid domain, 12 steps: the mainstem rows equal the reference's unpacking of the reference Fortran's
    outputs (flows bit for bit; the last record to 1e-13, see load_lowercolorado)."""
    import pandas as pd
    from troute_amd import nhd_network as nn
    from troute_amd.routing.compute import compute_diffusive_routing, compute_nhd_routing_v02
    z, lc, tw, dn, qlat_df, q0 = lowercolorado_diffusive_network()
    nts = 12
    conn = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    ind, reaches_bytw, rconn = nn.organize_independent_networks(conn)
    param_df = dn["param_df"].reindex(lc.ids)
    cols = {c: lc.params9[:, i] for i, c in enumerate(("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"))}
    full = pd.DataFrame(cols, index=lc.ids).drop(columns="dt")
    full["alt"] = z["alt"]
    e = pd.DataFrame()
    t0 = pd.Timestamp("2021-08-23 13:00")
    results = compute_nhd_routing_v02(
        conn, rconn, {}, reaches_bytw, "V02-structured", "by-network", 10000, 4, t0, lc.dt, nts, lc.qts, ind, full, q0,
        qlat_df, e, e, e, e, e, e, e, e, e, e, e, {}, True, False, e, {}, e, False, [{}, {}])
    # nwm_route's own unpacking of the return value (nwm_routing/__main__.py:1256-1257)
    subnetwork_list = results[1]
    results = results[0]
    assert subnetwork_list == [{}, {}]
    rd = compute_diffusive_routing(results, {tw: dn}, 1, t0, lc.dt, nts, q0, qlat_df, lc.qts, e, e, {}, e, e, None, None, e, e)
    assert len(rd) == 1 and len(rd[0]) == 10
    ids, dat = rd[0][0], rd[0][1]
    keep = ~np.isin(z["unpacked_ids"], dn["tributary_segments"])
    assert np.array_equal(ids, z["unpacked_ids"][keep])
    want = z["unpacked_dat"][keep][:, 3:3 * (nts + 1)]
    assert dat.shape == want.shape == (len(dn["mainstem_segs"]), 3 * nts)
    assert np.array_equal(dat[:, :-3], want[:, :-3], equal_nan=True)
    assert np.allclose(dat[:, -3:], want[:, -3:], rtol=1e-6, equal_nan=True)
    assert rd[0][6].shape == (len(ids), nts)
    del param_df


@pytest.mark.gpu
def test_gpu_batch_of_domains_in_one_launch():
    """trdw_diffnw_batch: different domains (sizes, reach layouts) as the blocks of one launch, each bit-identical to
    its reference golden -- and to itself when it appears several times in the batch."""
    from troute_amd.routing.fast_reach import diffusive as D
    names = ["comb", "chain1", "y3_nat", "y3", "comb", "comb_nat", "y3"]
    cases = [load_small(n) for n in names]
    outs = D.compute_diffusive_batch([c[0] for c in cases])
    assert len(outs) == len(names)
    for (ins, want), got in zip(cases, outs):
        for g, w in zip(got, want):
            assert same_bits(g, w)
    assert D.compute_diffusive_batch([]) == []
    bad = dict(cases[1][0])
    bad["cwnrow_g"] = np.array(1)
    with pytest.raises(ValueError):
        D.compute_diffusive_batch([cases[0][0], bad])


@pytest.mark.gpu
def test_gpu_given_depth_boundary_equals_host_restatement():
    """Downstream boundary option 1 (a prescribed depth series, the coastal coupling) has no reference golden here:
    the GPU is held to the host instantiation, which is itself pinned to the reference on option 2."""
    from troute_amd.routing.fast_reach import diffusive as D
    ins, _ = load_small("comb")
    ins = dict(ins)
    ins["para_ar_g"] = ins["para_ar_g"].copy()
    ins["para_ar_g"][10] = 1.0
    nts_db = 9
    ins["nts_db_g"] = np.array(nts_db)
    ins["dbcd_g"] = 0.6 + 0.3 * np.sin(np.arange(nts_db) / 2.0)
    ins["timestep_ar_g"] = ins["timestep_ar_g"].copy()
    ins["timestep_ar_g"][6] = 1350.0
    rc, want = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0
    got = D.compute_diffusive(ins)
    for g, w in zip(got, want):
        assert same_bits(g, w)
    assert np.isfinite(got[2]).all() and got[2].max() > 0.3


def long_mainstem(nmain=220, seed=4):
    """A synthetic domain much longer than anything that fits a compute unit's LDS: `nmain` mainstem reaches in series
    (6-8 nodes each, about 1 500 nodes) with a tributary hydrograph at every fifth junction and a mainstem side branch of
    three reaches joining half-way down -- the layout fp_network_map produces (reaches upstream first)."""
    rng = np.random.default_rng(seed)
    layout = []                      # dicts n, up (1-based reach ids), ds, main
    # side branch (mainstem): reaches 1..3
    for k in range(3):
        layout.append(dict(n=int(rng.integers(4, 7)), up=[k] if k else [], ds=None, main=True))
    join = nmain // 2
    first_main = len(layout) + 1
    trib_of = {}
    for k in range(nmain):
        up = [len(layout)] if k else []
        if k == join:
            up = up + [3]
        if k and k % 5 == 0:
            trib_of[k] = None
        layout.append(dict(n=int(rng.integers(6, 9)), up=up, ds=None, main=True))
    # tributaries (two nodes), appended where the reference lists them: anywhere before their junction is fine for frnw
    for k in sorted(trib_of):
        layout.insert(0, dict(n=2, up=[], ds=None, main=False))
        for r in layout[1:]:
            r["up"] = [u + 1 for u in r["up"]]
        first_main += 1
    ntrib = len(trib_of)
    for idx, k in enumerate(sorted(trib_of)):
        layout[first_main - 1 + k]["up"].append(ntrib - idx)
    nrch = len(layout)
    for j, r in enumerate(layout):
        for u in r["up"]:
            layout[u - 1]["ds"] = j + 1
    layout[-1]["ds"] = -99
    mx = max(r["n"] for r in layout)
    nsteps, dt = 4, 300.0
    tfin = dt * nsteps / 3600.0
    ts = np.zeros(10)
    ts[[0, 1, 2, 3, 4, 5, 7, 8, 9]] = [dt, 0.0, tfin, dt, 3600.0, dt, dt, dt, 10.0]
    para = np.array([0.95, 0.5, 10.0, 10000.0, -15.0, -10.0, 1.0, 0.02831, 0.0001, 1.0, 2.0])
    frnw = np.zeros((nrch, 20), np.int32)
    geo = {k: np.zeros((mx, nrch)) for k in ("z", "bo", "traps", "tw", "twcc", "mann", "manncc", "so", "dx")}
    iniq = np.zeros((mx, nrch))
    zbot = {}
    for j in reversed(range(nrch)):
        r = layout[j]
        n = r["n"]
        frnw[j, 0], frnw[j, 1], frnw[j, 2] = n, r["ds"], len(r["up"])
        frnw[j, 3:3 + len(r["up"])] = r["up"]
        frnw[j, 3 + len(r["up"])] = 555 if r["main"] else -555
        dx = rng.uniform(300.0, 1200.0, n)
        so = rng.uniform(2e-4, 1.5e-3, n)
        z = np.zeros(n)
        z[n - 1] = 2.0 if r["ds"] < 0 else zbot[r["ds"] - 1]
        for i in range(n - 2, -1, -1):
            z[i] = z[i + 1] + so[i] * dx[i]
        zbot[j] = z[0]
        bw = rng.uniform(10.0, 40.0)
        geo["z"][:n, j], geo["dx"][:n, j], geo["so"][:n, j] = z, dx, so
        geo["bo"][:n, j] = bw
        geo["traps"][:n, j] = rng.uniform(1.0, 3.0)
        geo["tw"][:n, j] = bw * 2.0
        geo["twcc"][:n, j] = bw * 6.0
        geo["mann"][:n, j] = 0.035
        geo["manncc"][:n, j] = 0.07
        iniq[:n, j] = rng.uniform(2.0, 6.0)
    nts_ql = max(1, int(np.ceil(tfin)))
    qlat = rng.uniform(0.0, 2e-4, (nts_ql, mx, nrch))
    nts_qtrib = nsteps + 1
    qtrib = np.zeros((nts_qtrib, nrch))
    for j, r in enumerate(layout):
        if not r["main"]:
            qtrib[:, j] = 2.0 + np.sin(np.arange(nts_qtrib) / 3.0 + j) ** 2
    return {"timestep_ar_g": ts, "nts_ql_g": nts_ql, "nts_ub_g": nsteps, "nts_db_g": 1, "ntss_ev_g": nsteps + 1,
            "nts_qtrib_g": nts_qtrib, "nts_da_g": 1, "mxncomp_g": mx, "nrch_g": nrch,
            "z_ar_g": geo["z"], "bo_ar_g": geo["bo"], "traps_ar_g": geo["traps"], "tw_ar_g": geo["tw"],
            "twcc_ar_g": geo["twcc"], "mann_ar_g": geo["mann"], "manncc_ar_g": geo["manncc"], "so_ar_g": geo["so"],
            "dx_ar_g": geo["dx"], "iniq": iniq, "frnw_col": 20, "frnw_g": frnw, "qlat_g": qlat,
            "ubcd_g": np.zeros((nsteps, nrch)), "dbcd_g": np.zeros(1), "qtrib_g": qtrib, "paradim": 11, "para_ar_g": para,
            "mxnbathy_g": 0, "x_bathy_g": np.zeros((0, mx, nrch)), "z_bathy_g": np.zeros((0, mx, nrch)),
            "mann_bathy_g": np.zeros((0, mx, nrch)), "size_bathy_g": np.zeros((mx, nrch), np.int32),
            "usgs_da_g": np.full((1, nrch), -4444.0), "usgs_da_reach_g": np.zeros(nrch, np.int32),
            "rdx_ar_g": np.zeros((0, 0)), "cwnrow_g": 0, "cwncol_g": 0, "crosswalk_g": np.zeros((0, 0)),
            "z_thalweg_g": np.zeros((0, 0))}


@pytest.mark.gpu
def test_gpu_long_mainstem_with_a_side_branch_equals_host_restatement():
    """~1 500 mainstem nodes in 223 reaches (the chain state does not fit LDS: records and windows in global memory, the
    per-node phases in three passes of the 512 threads), a mainstem junction of two mainstem reaches, tributaries at 43
    junctions: the parallel time loop against the host instantiation (itself pinned to the reference Fortran on the
    goldens), bit for bit; the one-wavefront kernel gives the same."""
    from troute_amd.routing.fast_reach import diffusive as D
    ins = long_mainstem()
    nodes = int(ins["frnw_g"][ins["frnw_g"][np.arange(ins["nrch_g"]), 3 + ins["frnw_g"][:, 2]] == 555, 0].sum())
    assert nodes > 1400
    rc, want = call_c(host_oracle(), "dw_oracle_diffnw", ins)
    assert rc == 0 and np.isfinite(want[0]).all() and np.abs(want[0]).max() > 1.0
    got = D.compute_diffusive(ins)
    for g, w in zip(got, want):
        assert same_bits(g, w)
