#!/usr/bin/env python3
"""Golden vectors for the diffusive-wave solver, made in the dev container only.

Inputs are marshalled by the REFERENCE's own diffusive_input_data_v02
(src/troute-routing/troute/routing/diffusive_utils_v02.py:659-1155, imported where it lies, with the
reference's nhd_network) for the LowerColorado_TX coastal subset of the shipped hybrid configuration
(test/LowerColorado_TX/test_AnA.yaml:45-54, domain/coastal_domain_subset.yaml; synthetic cross sections,
use_natl_xsections: False) following AbstractRouting.py:255-310 and compute.py:1740-1850; tributary inflows are
the Muskingum-Cunge flows of the tributary segments (oracle network loop == reference Fortran, bitwise).
Outputs come from the reference Fortran built from its own sources (oracle/_ref/libdiff_ref.so, c_diffnw).

Writes tests/golden/diffusive_lowercolorado.npz (every input array of the c_diffnw call + q/elv/depth outputs)
and tests/golden/diffusive_small.npz (hand-made small mainstems, same call).

--da: only tests/golden/diffusive_da.npz -- the same domain with a gage table handed to the reference's marshalling
(``usgs_df`` as nwm_route passes it on, nwm_routing/__main__.py:1292-1311 -> compute.py:1798-1803 -> fp_da_map :512-574):
the table, the three DA arguments the reference made of it, and the reference Fortran's outputs WITH those arguments,
which the script asserts are the bits of its outputs WITHOUT them (diffusive.f90:1282-1303: the branch is commented out).
"""
import ctypes as C
import importlib.util
import os
import sys
import types
from functools import partial

import numpy as np
import pandas as pd
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

import helpers as H  # noqa: E402
from make_fixtures import h5var, import_ref_nhd_network  # noqa: E402
from oracle import oracle as O  # noqa: E402

ARG_ORDER = ["timestep_ar_g", "nts_ql_g", "nts_ub_g", "nts_db_g", "ntss_ev_g", "nts_qtrib_g", "nts_da_g", "mxncomp_g",
             "nrch_g", "z_ar_g", "bo_ar_g", "traps_ar_g", "tw_ar_g", "twcc_ar_g", "mann_ar_g", "manncc_ar_g", "so_ar_g",
             "dx_ar_g", "iniq", "frnw_col", "frnw_g", "qlat_g", "ubcd_g", "dbcd_g", "qtrib_g", "paradim", "para_ar_g",
             "mxnbathy_g", "x_bathy_g", "z_bathy_g", "mann_bathy_g", "size_bathy_g", "usgs_da_g", "usgs_da_reach_g",
             "rdx_ar_g", "cwnrow_g", "cwncol_g", "crosswalk_g", "z_thalweg_g"]
INT_SCALARS = {"nts_ql_g", "nts_ub_g", "nts_db_g", "ntss_ev_g", "nts_qtrib_g", "nts_da_g", "mxncomp_g", "nrch_g", "frnw_col",
               "paradim", "mxnbathy_g", "cwnrow_g", "cwncol_g"}
INT_ARRAYS = {"frnw_g", "size_bathy_g", "usgs_da_reach_g"}


def call_reference(d):
    """c_diffnw of the reference build on a diff_inputs dict (pydiffusive.f90:8-55); Fortran-ordered arrays."""
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdiff_ref.so"))
    keep, args = [], []
    for k in ARG_ORDER:
        v = d[k]
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
    shape = (int(d["ntss_ev_g"]), int(d["mxncomp_g"]), int(d["nrch_g"]))
    outs = [np.zeros(shape, dtype=np.float64, order="F") for _ in range(3)]
    args += [o.ctypes.data_as(C.c_void_p) for o in outs]
    lib.c_diffnw(*args)
    return outs


def import_ref_diffusive_utils(nn):
    for name in ("troute",):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["troute.nhd_network"] = nn
    sys.modules["troute"].nhd_network = nn
    spec = importlib.util.spec_from_file_location(
        "ref_diffusive_utils", f"{REF}/src/troute-routing/troute/routing/diffusive_utils_v02.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pack(d, outs):
    z = {}
    for k in ARG_ORDER:
        v = d[k]
        z["in_" + k] = (np.array(int(v), dtype=np.int64) if k in INT_SCALARS
                        else np.asarray(v, dtype=np.int32 if k in INT_ARRAYS else np.float64))
    z["out_q"], z["out_elv"], z["out_depth"] = outs
    return z


def synthetic_topobathy(mainstem, param_df, seed=5):
    """A station table in the layout of the topobathy file (index = link id; xid_d, z, n per station): a parabolic
    channel with bumpy overbanks around each segment's RouteLink altitude, 9-17 stations, rough overbanks."""
    rng = np.random.default_rng(seed)
    idx, xs, zs, ns = [], [], [], []
    for seg in mainstem:
        m = int(rng.integers(9, 18))
        half = float(param_df.loc[seg, "tw"]) * rng.uniform(1.5, 3.0)
        x = np.sort(np.concatenate([[0.0], rng.uniform(-half, half, m - 1)]))
        prof = 6.0 * (x / half) ** 2 + 0.4 * np.abs(np.sin(x / 7.0)) * (np.abs(x) > 0.4 * half)
        prof[np.argmin(np.abs(x))] = 0.0
        idx += [seg] * m
        xs.append(x - x[0])
        zs.append(float(param_df.loc[seg, "alt"]) + prof)
        ns.append(np.where(np.abs(x) > 0.5 * half, rng.uniform(0.06, 0.2, m), rng.uniform(0.025, 0.05, m)))
    return pd.DataFrame({"xid_d": np.concatenate(xs), "z": np.concatenate(zs), "n": np.concatenate(ns)},
                        index=pd.Index(idx, name="comid"))


def synthetic_coastal_depths(tw, t0, hours):
    """Hourly water depths at two boundary nodes: a tide with a non-positive record and a missing record at `tw`."""
    cols = pd.date_range(t0, periods=hours, freq="h")
    k = np.arange(hours)
    a = 1.8 + 0.9 * np.sin(k / 2.3)
    a[1] = -0.2
    b = 2.5 + 0.3 * np.cos(k / 3.1)
    df = pd.DataFrame([a, b], index=[tw, tw + 1], columns=cols)
    if hours > 3:
        df.iloc[0, 2] = np.nan
    return df


def synthetic_gage_table(mainstem, trib, t0, nsteps, dt):
    """Observations "interpolated at dt" as nwm_route holds them: rows = gaged segments (two in one mainstem reach, one
    tributary, one id outside the domain), columns = stamps; a NaN record, and the last stamps missing altogether."""
    cols = pd.date_range(t0, periods=nsteps - 4, freq=pd.Timedelta(seconds=dt))
    ids = [mainstem[1], mainstem[2], mainstem[len(mainstem) // 2], trib[0], 999999999]
    k = np.arange(len(cols))
    vals = np.stack([3.0 + i + np.sin(k / (5.0 + i)) for i in range(len(ids))]).astype(np.float32)
    vals[1, 3] = np.nan
    return pd.DataFrame(vals, index=ids, columns=cols)


def lowercolorado(nn, du, nsteps, natural=False, coastal=False, gages=False):
    lc = H.LowerColorado()
    d = f"{REF}/test/LowerColorado_TX"
    dom = yaml.safe_load(open(f"{d}/domain/coastal_domain_subset.yaml"))
    (tw, spec), = dom.items()
    link = h5var(f"{d}/domain/RouteLink.nc", "link", np.int32).astype(np.int64)
    alt = pd.Series(h5var(f"{d}/domain/RouteLink.nc", "alt", np.float32), index=link)
    conn_all = {int(s): ([int(t)] if t != 0 else []) for s, t in zip(lc.ids, lc.to)}
    rconn_all = nn.reverse_network(conn_all)
    mainstem = [s for s in spec["links"] if s not in set(spec["upstream_boundary_link_mainstem"])]   # AbstractRouting.py:262-270
    ms = set(mainstem)
    trib = [u for s in mainstem for u in rconn_all[s] if u not in ms]                                   # :279-286
    connections = {k: conn_all[k] for k in (mainstem + trib)}
    connections[tw] = []
    rconn = nn.reverse_network(connections)
    net = nn.reachable_network(rconn)                                                                  # organize_independent_networks
    reaches = nn.dfs_decomposition(net[tw], partial(nn.split_at_waterbodies_and_junctions, set(trib), net[tw]))
    cols = {c: lc.params9[:, i] for i, c in enumerate(("dt", "dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"))}
    param_df = pd.DataFrame(cols, index=lc.ids)
    param_df["alt"] = alt.reindex(lc.ids).values
    param_df = param_df.loc[mainstem + trib]
    # tributary inflows: MC flows of the tributary segments (== reference Fortran, bitwise)
    rl, ul = lc.row_lists()
    fvd = O.network(nsteps, lc.qts, rl, ul, lc.params9, lc.q0, lc.qlat, True, det=True)
    row = {int(s): i for i, s in enumerate(lc.ids)}
    junction_inflows = pd.DataFrame(fvd[[row[s] for s in trib], 1:, 0].astype(np.float32), index=trib)
    qlat_df = pd.DataFrame(lc.qlat, index=lc.ids)
    q0 = pd.DataFrame(lc.q0, index=lc.ids, columns=["qu0", "qd0", "h0"])
    t0 = pd.Timestamp("2021-08-23 13:00")
    topo = synthetic_topobathy(mainstem, param_df) if natural else pd.DataFrame()
    coast = synthetic_coastal_depths(tw, t0, max(2, int(np.ceil(nsteps * lc.dt / 3600.0)) + 1)) if coastal else pd.DataFrame()
    usgs = synthetic_gage_table(mainstem, trib, t0, nsteps, lc.dt) if gages else pd.DataFrame()
    ins = du.diffusive_input_data_v02(
        tw, connections, rconn, reaches, mainstem, trib, None, param_df, qlat_df, q0, junction_inflows, lc.qts,
        t0, nsteps, lc.dt, pd.DataFrame(), topo, usgs, None, None, coast, pd.DataFrame())
    if gages:
        ins["_usgs_df"] = usgs
    extra = {"mainstem": np.array(mainstem), "trib": np.array(trib), "tw": np.array(tw),
             "alt": param_df_alt(alt, lc), "junction_inflows": junction_inflows.values}
    if natural:
        extra.update({"topo_index": topo.index.values, "topo_xid_d": topo["xid_d"].values, "topo_z": topo["z"].values,
                      "topo_n": topo["n"].values})
    if coastal:
        extra.update({"coast_index": coast.index.values, "coast_values": coast.values,
                      "coast_times": np.array([str(c) for c in coast.columns])})
    return ins, extra


def param_df_alt(alt, lc):
    return alt.reindex(lc.ids).values.astype(np.float32)


def small_cases():
    """Hand-made mainstems in the layout fp_network_map produces (diffusive_utils_v02.py:55-165): reaches listed
    upstream first; frnw_g = [ncomp, downstream reach (1-based, -99 at the tailwater), n upstream, upstream reach
    ids..., 555 (mainstem) / -555 (tributary)]."""
    cases = []
    rng = np.random.default_rng(12)
    for name, layout in (("chain1", [dict(n=6, up=[], ds=-99, main=True)]),
                         ("y3", [dict(n=2, up=[], ds=3, main=False), dict(n=2, up=[], ds=3, main=False),
                                 dict(n=5, up=[1, 2], ds=4, main=True), dict(n=7, up=[3], ds=-99, main=True)]),
                         ("comb", [dict(n=2, up=[], ds=4, main=False), dict(n=2, up=[], ds=5, main=False),
                                   dict(n=2, up=[], ds=4, main=False),
                                   dict(n=4, up=[1, 3], ds=5, main=True), dict(n=9, up=[4, 2], ds=-99, main=True)])):
        nrch = len(layout)
        mx = max(r["n"] for r in layout)
        nsteps, dt = 36, 300.0
        tfin = dt * nsteps / 3600.0
        ts = np.zeros(10)
        ts[[0, 1, 2, 3, 4, 5, 7, 8, 9]] = [dt, 0.0, tfin, dt, 3600.0, dt, dt, dt, 10.0]
        para = np.array([0.95, 0.5, 10.0, 10000.0, -15.0, -10.0, 1.0, 0.02831, 0.0001, 1.0, 2.0])
        frnw = np.zeros((nrch, 20), np.int32)
        geo = {k: np.zeros((mx, nrch)) for k in ("z", "bo", "traps", "tw", "twcc", "mann", "manncc", "so", "dx")}
        iniq = np.zeros((mx, nrch))
        zdown = 2.0
        for j in reversed(range(nrch)):
            r = layout[j]
            frnw[j, 0], frnw[j, 1], frnw[j, 2] = r["n"], r["ds"], len(r["up"])
            frnw[j, 3:3 + len(r["up"])] = r["up"]
            frnw[j, 3 + len(r["up"])] = 555 if r["main"] else -555
        # elevations: march upstream from the tailwater so that joined reaches share the junction elevation
        zbot = {}
        for j in reversed(range(nrch)):
            r = layout[j]
            n = r["n"]
            dx = rng.uniform(300.0, 1500.0, n)
            so = rng.uniform(2e-4, 2e-3, n)
            z_end = zdown if r["ds"] < 0 else zbot[r["ds"] - 1]
            z = np.zeros(n)
            z[n - 1] = z_end
            for i in range(n - 2, -1, -1):
                z[i] = z[i + 1] + so[i] * dx[i]
            zbot[j] = z[0]
            bw = rng.uniform(8.0, 40.0)
            geo["z"][:n, j], geo["dx"][:n, j], geo["so"][:n, j] = z, dx, so
            geo["bo"][:n, j] = bw
            geo["traps"][:n, j] = rng.uniform(1.0, 3.0)
            geo["tw"][:n, j] = bw * 2.0
            geo["twcc"][:n, j] = bw * 6.0
            geo["mann"][:n, j] = 0.035
            geo["manncc"][:n, j] = 0.07
            iniq[:n, j] = rng.uniform(2.0, 6.0)
        nts_ql = int(np.ceil(tfin * 3600.0 / 3600.0))
        qlat = rng.uniform(0.0, 2e-4, (nts_ql, mx, nrch))
        nts_qtrib = nsteps + 1
        tt = np.arange(nts_qtrib)
        qtrib = np.zeros((nts_qtrib, nrch))
        for j, r in enumerate(layout):
            if not r["main"]:
                qtrib[:, j] = 3.0 + 2.5 * np.sin(tt / 9.0 + j) ** 2 + rng.uniform(0, 0.2, nts_qtrib)
        d = {"timestep_ar_g": ts, "nts_ql_g": nts_ql, "nts_ub_g": nsteps, "nts_db_g": 1, "ntss_ev_g": nsteps + 1,
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
        cases.append((name, d))
        if name in ("y3", "comb"):
            # the same mainstem with natural cross sections (use_natl_xsections: True): 7-13 surveyed stations per node,
            # x negative left of the stream line, Manning's n per station (some above the 0.15 cap), bed = the node elevation
            dn = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in d.items()}
            mxb = 13
            xb, zb, mb = (np.zeros((mxb, mx, nrch)) for _ in range(3))
            sb = np.zeros((mx, nrch), np.int32)
            for j, r in enumerate(layout):
                for k in range(r["n"]):
                    ns = int(rng.integers(7, mxb + 1))
                    half = rng.uniform(15.0, 60.0)
                    x = np.sort(rng.uniform(-half, half, ns))
                    x[0], x[-1] = -half, half
                    prof = 0.002 * x ** 2 + 0.6 * np.abs(np.sin(x / 9.0)) * (np.abs(x) > 6.0)   # a channel and bumpy overbanks
                    prof[np.argmin(np.abs(x))] = 0.0
                    sb[k, j] = ns
                    xb[:ns, k, j] = x
                    zb[:ns, k, j] = geo["z"][k, j] + prof
                    mb[:ns, k, j] = np.where(np.abs(x) > 8.0, rng.uniform(0.06, 0.2, ns), rng.uniform(0.025, 0.045, ns))
            dn.update({"mxnbathy_g": mxb, "x_bathy_g": xb, "z_bathy_g": zb, "mann_bathy_g": mb, "size_bathy_g": sb})
            cases.append((name + "_nat", dn))
    return cases


def crosswalk_cases():
    """The small mainstems routed as a REFACTORED hydrofabric whose results are mapped back to an original one
    (diffnw :849-920; the reference's own Python never builds these arguments any more -- diffusive_utils_v02.py:1033-1038
    keeps empty placeholders -- so the crosswalk is hand-made).  Rows: refactored segment (ri, rj), number of original links,
    then (oi, oj, fraction of the link's length) per link.  Together they take every branch of the mapping: a link covered
    at once, in two parts, in three parts (the middle part writes nothing), a segment over two links, the 0.99 threshold."""
    base = {name: d for name, d in small_cases()}
    rows = [[1, 3, 1, 1, 3, 1.0], [2, 3, 1, 2, 3, 0.4], [3, 3, 1, 2, 3, 0.6], [4, 3, 2, 3, 3, 1.0, 4, 3, 0.5],
            [1, 4, 2, 4, 3, 0.5, 1, 4, 1.0], [2, 4, 1, 2, 4, 0.3], [3, 4, 1, 2, 4, 0.3], [4, 4, 1, 2, 4, 0.4],
            [5, 4, 1, 3, 4, 0.995], [6, 4, 1, 5, 4, 1.0]]
    cases = []
    for name in ("y3", "y3_nat"):
        d = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base[name].items()}
        rng = np.random.default_rng(77)
        cw = np.zeros((len(rows), 9))
        for r, row in enumerate(rows):
            cw[r, :len(row)] = row
        mx, nrch = int(d["mxncomp_g"]), int(d["nrch_g"])
        d.update({"rdx_ar_g": d["dx_ar_g"] * rng.uniform(0.8, 1.25, (mx, nrch)), "cwnrow_g": cw.shape[0], "cwncol_g": cw.shape[1],
                  "crosswalk_g": cw, "z_thalweg_g": d["z_ar_g"] - rng.uniform(0.0, 0.4, (mx, nrch))})
        cases.append((name + "_cw", d))
    return cases


if __name__ == "__main__":
    O.build()
    if "--crosswalk" in sys.argv:       # only the crosswalk fixture (the others are unchanged)
        cwz = {}
        for name, d in crosswalk_cases():
            outs = call_reference(d)
            assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 1.0 and (outs[1] != 0).any(), name
            for k, v in pack(d, outs).items():
                cwz[f"{name}__{k}"] = v
            print(name, "mapped cells", int((outs[0] != 0).sum()), "q max", outs[0].max())
        np.savez_compressed(os.path.join(HERE, "diffusive_crosswalk.npz"), **cwz)
        sys.exit(0)
    nn = import_ref_nhd_network()
    du = import_ref_diffusive_utils(nn)
    if "--da" in sys.argv:              # only the gage-table fixture (the others are unchanged)
        ins, segs = lowercolorado(nn, du, nsteps=36, gages=True)
        usgs = ins.pop("_usgs_df")
        plain, _ = lowercolorado(nn, du, nsteps=36)
        assert (ins["usgs_da_reach_g"] != 0).sum() >= 2 and (ins["usgs_da_g"] > 0).any() and not plain["usgs_da_reach_g"].any()
        for k in ARG_ORDER:
            if k not in ("usgs_da_g", "usgs_da_reach_g"):
                assert np.array_equal(np.asarray(ins[k]), np.asarray(plain[k])), k
        outs, outs_plain = call_reference(ins), call_reference(plain)
        for a, b in zip(outs, outs_plain):
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))     # the solver computes nothing from the gage arrays
        z = {"in_nts_da_g": np.array(int(ins["nts_da_g"])), "in_usgs_da_g": np.asarray(ins["usgs_da_g"], np.float64),
             "in_usgs_da_reach_g": np.asarray(ins["usgs_da_reach_g"], np.int32), "usgs_index": usgs.index.values,
             "usgs_values": usgs.values, "usgs_times": np.array([str(c) for c in usgs.columns]),
             "out_q": outs[0], "out_elv": outs[1], "out_depth": outs[2], "junction_inflows": segs["junction_inflows"]}
        np.savez_compressed(os.path.join(HERE, "diffusive_da.npz"), **z)
        print("gage table", usgs.shape, "reaches with a gage", np.flatnonzero(ins["usgs_da_reach_g"]) + 1, "nts_da", ins["nts_da_g"],
              "size", os.path.getsize(os.path.join(HERE, "diffusive_da.npz")))
        sys.exit(0)
    small = {}
    for name, d in small_cases():
        outs = call_reference(d)
        assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 1.0, name
        for k, v in pack(d, outs).items():
            small[f"{name}__{k}"] = v
        print(name, "q range", outs[0][outs[0] != 0].min(), outs[0].max(), "depth max", outs[2].max())
    np.savez_compressed(os.path.join(HERE, "diffusive_small.npz"), **small)
    ins, segs = lowercolorado(nn, du, nsteps=72)
    outs = call_reference(ins)
    print("LowerColorado coastal subset: nrch", ins["nrch_g"], "mxncomp", ins["mxncomp_g"], "mainstem reaches",
          int((np.asarray(ins["frnw_g"]) == 555).sum()), "q max", outs[0].max(), "depth max", outs[2].max())
    z = pack(ins, outs)
    z.update(segs)
    # the reference's unpacking of these outputs (diffusive_utils_v02.py:1156-1212, called with out_depth as in
    # compute.py:1852-1857)
    ids, dat = du.unpack_output(ins["pynw"], ins["ordered_reaches"], outs[0], outs[2])
    z["unpacked_ids"], z["unpacked_dat"] = ids, dat
    np.savez_compressed(os.path.join(HERE, "diffusive_lowercolorado.npz"), **z)
    # the same domain with natural cross sections and a coastal depth boundary (use_natl_xsections / coastal_boundary_domain
    # of the v4 hybrid configuration; the station table and the depth series are synthetic, no such file ships with the
    # reference's test data), 36 steps = 3 h
    ins, segs = lowercolorado(nn, du, nsteps=36, natural=True, coastal=True)
    outs = call_reference(ins)
    print("LowerColorado natural + coastal: mxnbathy", ins["mxnbathy_g"], "dsbd option", ins["para_ar_g"][10], "dbcd", ins["dbcd_g"],
          "q max", outs[0].max(), "depth max", outs[2].max())
    assert ins["mxnbathy_g"] > 0 and ins["para_ar_g"][10] == 1 and np.isfinite(outs[0]).all()
    z = pack(ins, outs)
    z.update(segs)
    np.savez_compressed(os.path.join(HERE, "diffusive_lowercolorado_nat.npz"), **z)
    print({k: os.path.getsize(os.path.join(HERE, k)) for k in ("diffusive_small.npz", "diffusive_lowercolorado.npz",
                                                                "diffusive_lowercolorado_nat.npz")})
