#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the development container (needs /root/reference, h5dump and
amdflang); the fixtures it writes are data -- inputs and expected outputs --
and are what the tests and the GPU box use.  Nothing here is imported by the
product.

Sources of truth (paths relative to /root/reference):
  * kernel vectors: reference Fortran builds in oracle/_ref (oracle/Makefile),
    inputs from the reference's own generator
    src/kernel/muskingum/test_suite_parameters.py:57-92 (imported, seed 16 as
    in test_MC_kernel.py:8) and the KATs of
    src/kernel/muskingum/mc_sseg_stime_NOLOOP_demo.py:173-248,:311-328
  * LowerColorado_TX domain: test/LowerColorado_TX/domain/RouteLink.nc and
    test/LowerColorado_TX/channel_forcing/*.CHRTOUT_DOMAIN1 (HDF5-backed
    NetCDF4, read with h5dump), column mapping NHDNetwork.py:137-156, qlat =
    qBucket + qSfcLatRunoff (nhd_io.py:397-434)
  * reach decomposition: the reference's own graph code
    src/troute-network/troute/nhd_network.py (imported with two stub modules
    for its absent third-party imports toolz / deprecated) driven exactly as
    organize_independent_networks does (nhd_network_utilities_v02.py:133-200)
  * toy network: data literals of src/troute-network/troute/test_nhd_network.py
  * network goldens: the reference Fortran kernel symbol (Qj_0 = 0 build)
    called per segment by the restated network loop (oracle.network(ref_name=))

Usage:  python tests/golden/make_fixtures.py
"""
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import types
from functools import partial

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

H5DUMP = "/opt/conda/bin/h5dump"


def h5var(path, var, dtype):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tf:
        subprocess.check_call([H5DUMP, "-d", "/" + var, "-b", "LE", "-o", tf.name, path],
                              stdout=subprocess.DEVNULL)
        return np.fromfile(tf.name, dtype=dtype)


def import_ref_nhd_network():
    toolz = types.ModuleType("toolz")
    toolz.pluck = lambda ind, seqs: (s[ind] for s in seqs)
    dep = types.ModuleType("deprecated")
    dep.deprecated = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault("toolz", toolz)
    sys.modules.setdefault("deprecated", dep)
    spec = importlib.util.spec_from_file_location(
        "ref_nhd_network", f"{REF}/src/troute-network/troute/nhd_network.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_organize(nn, connections):
    """organize_independent_networks with no break segments
    (nhd_network_utilities_v02.py:155-200)."""
    rconn = nn.reverse_network(connections)
    independent_networks = nn.reachable_network(rconn)
    reaches_bytw = {}
    for tw, net in independent_networks.items():
        reaches_bytw[tw] = nn.dfs_decomposition(net, partial(nn.split_at_junction, net))
    return independent_networks, reaches_bytw, rconn


# ----------------------------------------------------------------------------
def kernel_vectors():
    rng = np.random.default_rng(20250117)
    rows = []
    tags = []

    def add(tag, arr):
        arr = np.atleast_2d(np.asarray(arr, dtype=np.float64))
        rows.append(arr)
        tags.extend([tag] * arr.shape[0])

    # KATs (mc_sseg_stime_NOLOOP_demo.py:173-248 fp32 inputs; :311-328 compound)
    add("kat_lowflow_f32", [60.0, 0.04598825, 0.04598825, 0.21487340, 40.0, 1800.0, 112.0, 448.0,
                            623.5999755859375, 0.02800000086426735, 0.03136000037193298,
                            1.399999976158142, 0.0017999999690800905, 0.0704801953, 0.0100334705])
    add("kat_lowflow_f64in", [60.0, 0.04598825885217007, 0.04598825885217007, 0.21487345087737053,
                              40.0, 1800.0, 112.0, 448.0, 623.5999755859375, 0.02800000086426735,
                              0.03136000037193298, 1.399999976158142, 0.0017999999690800905,
                              0.07048020184743511, 0.010033471026476835])
    add("kat_compound", [60.0, 45009, 50098, 50014, 40.0, 1800.0, 112.0, 248.0, 623.5999755859375,
                         0.02800000086426735, 0.03136000037193298, 0.42, 0.007999999690800905,
                         0.0, 30])
    # test_MC_kernel.py docstring vector (:14-30)
    add("kat_test_docstring", [300.0, 4509, 5098, 5017, 40.0, 1800.0, 112.0, 448.0,
                               623.5999755859375, 0.02800000086426735, 0.03136000037193298,
                               1.399999976158142, 0.0017999999690800905, 0.0, 30])

    # the reference's own random generator, seed 16, 5000 vectors
    # (mc_sseg_stime_NOLOOP_demo.py:337-340)
    spec = importlib.util.spec_from_file_location(
        "ref_tsp", f"{REF}/src/kernel/muskingum/test_suite_parameters.py")
    tsp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tsp)
    gen = np.array(list(tsp.generate_conus_MC_parameters(5000, 16)), dtype=np.float64)
    # generator tuple order: dx bw tw twcc n ncc cs s0 qlat qup quc qdp depthp dt
    dx, bw, tw, twcc, n, ncc, cs, s0, qlat, qup, quc, qdp, depthp, dt = gen.T
    z = np.zeros_like(dx)
    add("refgen_mapped", np.stack([dt, qup, quc, qdp, qlat, dx, bw, tw, twcc, n, ncc, cs, s0, z,
                                   depthp], 1))
    # ... and exactly as the reference test splats it into compare_methods
    # (dt,qup,quc,qdp,qlat,dx,bw,tw,twcc,n,ncc,cs,s0,depthp <- tuple positions 0..13)
    g = gen
    add("refgen_as_splatted", np.stack([g[:, 0], g[:, 1], g[:, 2], g[:, 3], g[:, 4], g[:, 5],
                                        g[:, 6], g[:, 7], g[:, 8], g[:, 9], g[:, 10], g[:, 11],
                                        g[:, 12], z, g[:, 13]], 1)[:1000])

    # CONUS-like realistic states (SURVEY 8d distributions)
    m = 6000
    bw_ = np.clip(rng.lognormal(np.log(2.77), 0.9, m), 0.135, 230)
    tw_ = bw_ * 5 / 3
    nn_ = np.where(rng.random(m) < 0.95, 0.06, np.where(rng.random(m) < 0.5, 0.05, 0.04))
    add("realistic", np.stack([
        np.full(m, 300.0),
        rng.lognormal(-3, 2.5, m) * (rng.random(m) > 0.2),
        rng.lognormal(-3, 2.5, m) * (rng.random(m) > 0.2),
        rng.lognormal(-3, 2.5, m) * (rng.random(m) > 0.15),
        rng.lognormal(np.log(2.3e-4), 2.3, m) * (rng.random(m) > 0.11),
        np.clip(rng.lognormal(np.log(1549), 1.0, m), 1, 95714),
        bw_, tw_, 3 * tw_, nn_, 2 * nn_,
        np.clip(rng.normal(0.586, 0.195, m), 0.085, 2.25),
        np.clip(rng.lognormal(np.log(0.006), 1.5, m), 1e-5, 4.6),
        np.zeros(m),
        rng.lognormal(-2, 1.5, m) * (rng.random(m) > 0.1)], 1))

    # edge cases the reference code branches on
    base = np.array([300.0, 1.0, 1.2, 0.9, 0.01, 1500.0, 3.0, 5.0, 15.0, 0.06, 0.12, 0.6, 0.005,
                     0.0, 0.3])
    e = []

    def edge(**kw):
        r = base.copy()
        for k, v in kw.items():
            r[O.IN_COLS.index(k)] = v
        e.append(r)

    edge(qup=0, quc=0, qdp=0, ql=0)                 # no-flow branch (f90:171-178)
    edge(qup=0, quc=0, qdp=0, ql=0, depthp=0)
    edge(qup=0, quc=0.5, qdp=0, ql=0)               # only quc > 0 (differs from WRF-Hydro)
    edge(cs=0.0)                                    # z = 1 (f90:49-50)
    edge(bw=6.0, tw=5.0)                            # bw > tw (f90:55-56)
    edge(bw=5.0, tw=5.0)                            # bw == tw (f90:57-58)
    edge(twcc=0.0, depthp=5.0, qup=400, quc=400, qdp=400)   # NWM 3.0 exception (f90:400-403)
    edge(ncc=0.0, depthp=5.0, qup=400, quc=400, qdp=400)    # over-bank guard (f90:248)
    edge(depthp=5.0, qup=400, quc=420, qdp=390)     # compound channel active
    edge(depthp=-1.0)                               # negative depthp clamps (f90:69)
    edge(ql=-0.5)                                   # channel loss
    edge(ql=-50.0)                                  # loss > water (f90:150-151, :316-318)
    edge(ql=-2.0, qup=0.5, quc=0.2, qdp=3.0)
    edge(dx=1.0)                                    # Km = dt branch
    edge(dx=95714.0)
    edge(s0=1e-5)
    edge(s0=4.6)
    edge(qup=1e-6, quc=1e-6, qdp=1e-6, ql=0, depthp=1e-4)   # h < mindepth early exit
    edge(qup=70000, quc=70000, qdp=70000, depthp=0.01)       # far-off initial bracket
    edge(qup=5e-3, quc=5e-3, qdp=60000, depthp=100.0)
    edge(dt=5.0)
    edge(dt=3600.0)
    add("edge", np.array(e))

    X = np.concatenate(rows, 0)
    X32 = X.astype(np.float32)
    out = {
        "inputs_f64": X,
        "tags": np.array(tags),
        # canonical (Qj_0 = 0) reference Fortran, fp32 and fp64
        "ref_qj0_f32": O.ref_segments(X32, "libmc_ref_qj0_f32.so"),
        "ref_qj0_f64": O.ref_segments(X32.astype(np.float64), "libmc_ref_qj0_f64.so"),
        # unmodified WRF-Hydro original (qdc velc depthc)
        "wrf_f32": O.wrf_segments(X32),
        # as-shipped T-Route kernel (uninitialised Qj_0: call-history dependent; statistical only)
        "ref_asshipped_f32": O.ref_segments(X32, "libmc_ref_f32.so"),
    }
    # sanity: the restatement equals the canonical reference bit-for-bit
    a = O.segments(X32)
    assert np.array_equal(a.view(np.uint32), out["ref_qj0_f32"].view(np.uint32))
    a64 = O.segments(X32.astype(np.float64))
    assert np.array_equal(a64.view(np.uint64), out["ref_qj0_f64"].view(np.uint64))
    np.savez_compressed(os.path.join(HERE, "kernel_vectors.npz"), **out)
    print("kernel_vectors:", X.shape)


# ----------------------------------------------------------------------------
def toy_network(nn):
    src = open(f"{REF}/src/troute-network/troute/test_nhd_network.py").read()
    lit = src.split("import pandas as pd")[0]          # data literals only
    ns = {}
    exec(lit, ns)
    rows = ns["network_clean"]
    conn = ns["expected_connections"]
    ind, reaches_bytw, rconn = ref_organize(nn, conn)
    assert {k: v for k, v in rconn.items()} == ns["expected_rconn"]
    fx = {
        "rows_key_dx_downstream_waterbody": rows,
        "terminal_code": ns["test_terminal_code"],
        "expected_connections": {str(k): v for k, v in conn.items()},
        "expected_rconn": {str(k): v for k, v in ns["expected_rconn"].items()},
        "independent_networks": {str(tw): {str(k): v for k, v in net.items()}
                                 for tw, net in ind.items()},
        "reaches_bytw": {str(tw): r for tw, r in reaches_bytw.items()},
        "headwaters": sorted(nn.headwaters(conn)),
        "tailwaters": sorted(nn.tailwaters(conn)),
    }
    json.dump(fx, open(os.path.join(HERE, "toy_network.json"), "w"), indent=0)
    print("toy network: networks", list(reaches_bytw))


# ----------------------------------------------------------------------------
def lowercolorado(nn):
    d = f"{REF}/test/LowerColorado_TX"
    rl = f"{d}/domain/RouteLink.nc"
    link = h5var(rl, "link", np.int32).astype(np.int64)
    to = h5var(rl, "to", np.int32).astype(np.int64)
    cols = {"dx": "Length", "bw": "BtmWdth", "tw": "TopWdth", "twcc": "TopWdthCC", "n": "n",
            "ncc": "nCC", "cs": "ChSlp", "s0": "So"}
    par = {k: h5var(rl, v, np.float32) for k, v in cols.items()}
    mask = np.loadtxt(f"{d}/domain/coastal_subset.txt", dtype=np.int64)
    keep = np.isin(link, mask)
    order = np.argsort(link[keep], kind="stable")      # set_index("key").sort_index()
    ids = link[keep][order]
    to = to[keep][order]
    par = {k: v[keep][order] for k, v in par.items()}
    # terminal codes: 0 plus any downstream id not in the index (NHDNetwork.py:215-224)
    to_in = np.isin(to, ids)
    connections = {}
    for s, t, ok in zip(ids.tolist(), to.tolist(), to_in.tolist()):
        connections[s] = [t] if (ok and t != 0) else []
    ind, reaches_bytw, rconn = ref_organize(nn, connections)

    # forcing: 25 hourly columns (max_col = 1 + nts // qts_subdivisions, NHDNetwork.py:400)
    files = sorted(os.listdir(f"{d}/channel_forcing"))[:25]
    fid = h5var(f"{d}/channel_forcing/{files[0]}", "feature_id", np.int64)
    pos = {int(v): i for i, v in enumerate(fid)}
    take = np.array([pos[int(s)] for s in ids])
    sf = np.float32(1e-5)
    ql = []
    for f in files:
        p = f"{d}/channel_forcing/{f}"
        qb = h5var(p, "qBucket", np.int32)
        qs = h5var(p, "qSfcLatRunoff", np.int32)
        fill = -999900000
        b = np.where(qb == fill, 0.0, qb * np.float64(sf))     # netCDF4 scaling -> float64
        s_ = np.where(qs == fill, 0.0, qs * np.float64(sf))
        ql.append((b + s_)[take])
    qlat = np.stack(ql, 1).astype(np.float32)

    # reaches as flat arrays, reference list order, per tailwater
    tws, reach_ptr, reach_ids, reach_tw = [], [0], [], []
    for tw, rl_ in reaches_bytw.items():
        for r in rl_:
            reach_ids.extend(r)
            reach_ptr.append(len(reach_ids))
            reach_tw.append(tw)
        tws.append(tw)
    up_ptr = [0]
    up_ids = []
    for s in ids.tolist():
        up_ids.extend(rconn.get(s, []))
        up_ptr.append(len(up_ids))

    np.savez_compressed(
        os.path.join(HERE, "lowercolorado_domain.npz"),
        ids=ids, to=np.where(to_in, to, 0),
        params=np.stack([par[k] for k in ("dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0")], 1),
        param_cols=np.array(["dx", "bw", "tw", "twcc", "n", "ncc", "cs", "s0"]),
        qlat=qlat, forcing_files=np.array(files),
        ref_tailwaters=np.array(tws, dtype=np.int64),
        ref_reach_ptr=np.array(reach_ptr, dtype=np.int64),
        ref_reach_ids=np.array(reach_ids, dtype=np.int64),
        ref_reach_tw=np.array(reach_tw, dtype=np.int64),
        ref_rconn_ptr=np.array(up_ptr, dtype=np.int64),
        ref_rconn_ids=np.array(up_ids, dtype=np.int64),
    )
    print("lowercolorado: nseg", len(ids), "networks", len(tws), "reaches", len(reach_ptr) - 1)

    # ---- waterbodies: LAKEPARM + NHDWaterbodyComID, reference graph collapse --------
    wb = h5var(rl, "NHDWaterbodyComID", np.int32).astype(np.int64)[keep][order]
    lp = f"{d}/domain/LAKEPARM.nc"
    lake_id = h5var(lp, "lake_id", np.int32).astype(np.int64)
    lcols = ["LkArea", "LkMxE", "OrificeA", "OrificeC", "OrificeE", "WeirC", "WeirE", "WeirL", "ifd"]
    ltab = np.stack([h5var(lp, c, np.float64 if h5var(lp, c, np.uint8).size == 8 * lake_id.size else np.float32)
                     for c in lcols], 1).astype(np.float64)
    wbody_map = {int(s_): int(w) for s_, w in zip(ids, wb) if w != -9999}     # extract_waterbody_connections
    present = np.array(sorted(set(wbody_map.values())), dtype=np.int64)
    lsel = np.array([int(np.flatnonzero(lake_id == w)[0]) for w in present])   # first row per lake id
    conn_wb, link_lake = nn.replace_waterbodies_connections(connections, wbody_map)
    rconn_wb = nn.reverse_network(conn_wb)
    ind_wb = nn.reachable_network(rconn_wb)
    wbset = set(present.tolist())
    wb_nodes, wb_to, wr_ptr, wr_ids, wr_tw = [], [], [0], [], []
    for n_, dst in conn_wb.items():
        wb_nodes.append(n_)
        wb_to.append(dst[0] if dst else 0)
        assert len(dst) <= 1
    for tw, net in ind_wb.items():
        rl_ = nn.dfs_decomposition(net, partial(nn.split_at_waterbodies_and_junctions, wbset, net))
        for r_ in rl_:
            wr_ids.extend(r_)
            wr_ptr.append(len(wr_ids))
            wr_tw.append(tw)
    np.savez_compressed(
        os.path.join(HERE, "lowercolorado_waterbodies.npz"),
        seg_ids=ids, wb_of_seg=wb, lake_ids=present, lake_cols=np.array(lcols), lake_table=ltab[lsel],
        ref_conn_nodes=np.array(wb_nodes, dtype=np.int64), ref_conn_to=np.array(wb_to, dtype=np.int64),
        ref_link_lake_keys=np.array(list(link_lake.keys()), dtype=np.int64),
        ref_link_lake_vals=np.array(list(link_lake.values()), dtype=np.int64),
        ref_reach_ptr=np.array(wr_ptr, dtype=np.int64), ref_reach_ids=np.array(wr_ids, dtype=np.int64),
        ref_reach_tw=np.array(wr_tw, dtype=np.int64))
    print("waterbodies:", len(present), "lakes,", int((wb != -9999).sum()), "segments inside,",
          len(wb_nodes), "nodes after collapse,", len(wr_ptr) - 1, "reaches")

    # ---- network goldens -----------------------------------------------------------------
    nseg = len(ids)
    row = {int(s): i for i, s in enumerate(ids)}
    reaches, ups = [], []
    for tw, rl_ in reaches_bytw.items():
        for r in rl_:
            reaches.append(np.array([row[s] for s in r], dtype=np.int64))
            ups.append(np.array([row[s] for s in ind[tw].get(r[0], [])], dtype=np.int64))
    dt = 300.0
    params9 = np.concatenate([np.full((nseg, 1), dt, np.float32),
                              np.stack([par[k] for k in ("dx", "bw", "tw", "twcc", "n", "ncc",
                                                         "cs", "s0")], 1)], 1)
    q0 = np.zeros((nseg, 3), np.float32)               # cold start (restart line commented out)
    nts, qts = 288, 12
    tsel = np.array([1, 2, 3, 6, 12, 24, 48, 96, 144, 192, 240, 288])
    rng = np.random.default_rng(7)
    probes = np.sort(rng.choice(nseg, 100, replace=False))
    gold = {"tsel": tsel, "probes": probes, "nts": nts, "qts_subdivisions": qts, "dt": dt}
    for short in (True, False):
        tag = "shortts" if short else "fullts"
        ref = O.network(nts, qts, reaches, ups, params9, q0, qlat, short,
                        ref_name="libmc_ref_qj0_f32.so")
        mine, iters = O.network(nts, qts, reaches, ups, params9, q0, qlat, short,
                                return_iters=True)
        assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32)), tag
        gold[f"{tag}_f32_tsel"] = ref[:, tsel, :]
        gold[f"{tag}_f32_probes"] = ref[probes]
        gold[f"{tag}_f32_mean_iters"] = iters / (nseg * nts)
        ref64 = O.network(nts, qts, reaches, ups, params9.astype(np.float64), q0, qlat, short,
                          ref_name="libmc_ref_qj0_f64.so")
        mine64 = O.network(nts, qts, reaches, ups, params9.astype(np.float64), q0, qlat, short)
        assert np.array_equal(ref64.view(np.uint64), mine64.view(np.uint64)), tag
        gold[f"{tag}_f64_final"] = ref64[:, nts, :]
        gold[f"{tag}_f64_probes"] = ref64[probes]
        shipped = O.network(nts, qts, reaches, ups, params9, q0, qlat, short,
                            ref_name="libmc_ref_f32.so")
        gold[f"{tag}_asshipped_f32_final"] = shipped[:, nts, :]
        print(tag, "mean secant iters/seg-step", iters / (nseg * nts),
              "max q", float(ref[:, :, 0].max()))
    np.savez_compressed(os.path.join(HERE, "lowercolorado_golden.npz"), **gold)


# ----------------------------------------------------------------------------
def nudging_vectors():
    """simple_da golden vectors from the reference's own Cython source
    (src/troute-routing/troute/routing/fast_reach/simple_da.pyx), compiled where it lies; a 10-line
    probe .pyx (written here, into a temp dir) exposes the cdef function to Python."""
    import shutil
    td = tempfile.mkdtemp()
    try:
        open(os.path.join(td, "da_probe.pyx"), "w").write(
            "# cython: language_level=3\n"
            "from troute.routing.fast_reach.simple_da cimport simple_da, simple_da_with_decay\n"
            "def call_simple_da(float t, float rp, float dc, float gm, float target, float model, float lt, float lv):\n"
            "    cdef (float, float, float, float) r = simple_da(t, rp, dc, gm, target, model, lt, lv, 0)\n"
            "    return (r[0], r[1], r[2], r[3])\n"
            "def call_decay(float lo, float m, float minutes, float decay):\n"
            "    return simple_da_with_decay(lo, m, minutes, decay)\n")
        open(os.path.join(td, "setup.py"), "w").write(
            "from setuptools import setup, Extension\nfrom Cython.Build import cythonize\n"
            f"R='{REF}/src/troute-routing'\n"
            "exts=[Extension('troute.routing.fast_reach.simple_da',[R+'/troute/routing/fast_reach/simple_da.pyx']),"
            "Extension('da_probe',['da_probe.pyx'])]\n"
            "setup(ext_modules=cythonize(exts, include_path=[R], build_dir='build_c'))\n")
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--build-lib", "out", "--build-temp", "tmp"],
                              cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sys.path.insert(0, os.path.join(td, "out"))
        import da_probe
        rng = np.random.default_rng(42)
        n = 20000
        nan = np.float32(np.nan)
        t = rng.integers(1, 400, n).astype(np.float32)
        gm = rng.choice([0, 12, 100, 288, 289, 400], n).astype(np.float32)
        target = np.where(rng.random(n) < 0.35, nan, rng.lognormal(0, 2, n)).astype(np.float32)
        model = rng.lognormal(0, 2, n).astype(np.float32)
        lt = np.where(rng.random(n) < 0.2, nan, rng.uniform(-86400, 86400, n)).astype(np.float32)
        lv = np.where(np.isnan(lt) | (rng.random(n) < 0.1), nan, rng.lognormal(0, 2, n)).astype(np.float32)
        dc = rng.choice([1.0, 60.0, 120.0, 720.0], n).astype(np.float32)
        rp = rng.choice([60.0, 300.0, 3600.0], n).astype(np.float32)
        x = np.stack([t, rp, dc, gm, target, model, lt, lv], 1)
        y = np.array([da_probe.call_simple_da(*map(float, r)) for r in x], dtype=np.float32)
        mine = np.array([O.simple_da(*map(float, r)) for r in x], dtype=np.float32)
        assert np.array_equal(y.view(np.uint32), mine.view(np.uint32)), "oracle simple_da != reference"
        kat = np.float32(da_probe.call_decay(9.5, 12.0, 60.0, 120.0))       # test_compute.py:33-42
        assert abs(float(kat) - 10.483673095703125) < 2.3e-6 * 10.5
        np.savez_compressed(os.path.join(HERE, "simple_da_vectors.npz"), inputs=x, outputs=y, kat_decay=kat)
        print("simple_da vectors:", x.shape, "branches:",
              int(((t <= gm) & ~np.isnan(target)).sum()), int((np.isnan(target) & np.isnan(lv)).sum()))
    finally:
        shutil.rmtree(td, ignore_errors=True)


def file_format_vectors():
    """Expected contents of the data files copied from the reference's test tree into
    tests/golden/lowercolorado_files/ (two CHRTOUT forcing files, the HYDRO_RST restart), read with h5dump --
    independent of troute_amd.h5 -- plus RouteLink's link ids in file order (the restart's crosswalk)."""
    import shutil
    d = f"{REF}/test/LowerColorado_TX"
    out = os.path.join(HERE, "lowercolorado_files")
    os.makedirs(out, exist_ok=True)
    files = sorted(os.listdir(f"{d}/channel_forcing"))[:2]
    for f in files:
        shutil.copyfile(f"{d}/channel_forcing/{f}", os.path.join(out, f))
    rst = "HYDRO_RST.2021-08-23_12:00_DOMAIN1"
    shutil.copyfile(f"{d}/restart/{rst}", os.path.join(out, rst.replace(":", "_")))
    exp = {"link": h5var(f"{d}/domain/RouteLink.nc", "link", np.int32)}
    for v in ("qlink1", "qlink2", "hlink"):
        exp[v] = h5var(f"{d}/restart/{rst}", v, np.float32)
    for k, f in enumerate(files):
        exp[f"feature_id_{k}"] = h5var(f"{d}/channel_forcing/{f}", "feature_id", np.int64)
        exp[f"qBucket_{k}"] = h5var(f"{d}/channel_forcing/{f}", "qBucket", np.int32)
        exp[f"qSfcLatRunoff_{k}"] = h5var(f"{d}/channel_forcing/{f}", "qSfcLatRunoff", np.int32)
        exp[f"streamflow_{k}"] = h5var(f"{d}/channel_forcing/{f}", "streamflow", np.int32)
    np.savez_compressed(os.path.join(HERE, "lowercolorado_files_expected.npz"), **exp)
    print("lowercolorado_files_expected.npz:", {k: v.shape for k, v in exp.items()})


def import_ref_compute(nn):
    """The reference's routing/compute.py, for its pure-pandas helpers.  The modules it imports that cannot exist
    here (compiled Cython extensions, the diffusive utilities) are replaced by empty stand-ins: none of them is
    on the path of _prep_da_dataframes / _prep_da_positions_byreach."""
    for name in ("troute", "troute.routing", "troute.routing.fast_reach"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["troute.nhd_network"] = nn
    sys.modules["troute"].nhd_network = nn
    mc = types.ModuleType("troute.routing.fast_reach.mc_reach")
    mc.compute_network_structured = lambda *a, **k: None
    sys.modules["troute.routing.fast_reach.mc_reach"] = mc
    for name in ("troute.routing.diffusive_utils_v02", "troute.routing.fast_reach.diffusive"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["troute.routing"].diffusive_utils_v02 = sys.modules["troute.routing.diffusive_utils_v02"]
    sys.modules["troute.routing.fast_reach"].diffusive = sys.modules["troute.routing.fast_reach.diffusive"]
    if "joblib" not in sys.modules:
        try:
            import joblib  # noqa: F401
        except Exception:
            jb = types.ModuleType("joblib")
            jb.delayed = jb.Parallel = None
            sys.modules["joblib"] = jb
    spec = importlib.util.spec_from_file_location(
        "ref_compute", os.path.join(REF, "src/troute-routing/troute/routing/compute.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def da_prep_vectors(nn):
    """_prep_da_dataframes / _prep_da_positions_byreach (compute.py:49-140) on seeded tables, the four
    (usgs_df, lastobs_df) presence cases -> tests/golden/da_prep_vectors.json."""
    import pandas as pd
    rc = import_ref_compute(nn)
    rng = np.random.default_rng(21)
    seg_ids = np.sort(rng.choice(np.arange(1000, 9000), 400, replace=False))
    idx = pd.Index(seg_ids)
    # reaches: consecutive runs of the sorted ids, 1-4 long
    reaches, i = [], 0
    while i < len(seg_ids):
        k = int(rng.integers(1, 5))
        reaches.append([int(x) for x in seg_ids[i:i + k]])
        i += k
    cases = []
    for case in range(8):
        off = rng.choice(np.arange(9000, 9500), 6, replace=False)           # gages outside the table
        g_usgs = np.concatenate([rng.choice(seg_ids, 30, replace=False), off[:3]])
        g_last = np.concatenate([rng.permutation(g_usgs)[:25], rng.choice(seg_ids, 6, replace=False), off[3:]])
        g_last = np.array(list(dict.fromkeys(g_last.tolist())))
        rng.shuffle(g_usgs)
        usgs = pd.DataFrame(rng.lognormal(0, 1, (len(g_usgs), 5)).astype("float32"), index=g_usgs)
        last = pd.DataFrame({"time_since_lastobs": -rng.uniform(0, 7200, len(g_last)).astype("float32"),
                             "lastobs_discharge": rng.lognormal(0, 1, len(g_last)).astype("float32")}, index=g_last)
        which = case % 4
        if which == 0:   # both present: the reference indexes usgs_df by the lastobs gages (compute.py:93-99),
            last = last.loc[[g for g in last.index if g in set(usgs.index)]]   # so they must be a subset
        u = usgs if which in (0, 2) else pd.DataFrame()
        l = last if which in (0, 1) else pd.DataFrame()
        excl = [int(x) for x in rng.choice(seg_ids, 20, replace=False)] if case >= 4 else None
        us, ls, byseg = rc._prep_da_dataframes(u, l, idx, excl)
        byreach, bygage = rc._prep_da_positions_byreach(reaches, ls.index)
        cases.append({
            "usgs_index": [int(x) for x in u.index], "usgs_values": np.asarray(u.values, dtype="float64").tolist(),
            "lastobs_index": [int(x) for x in l.index],
            "lastobs_cols": [str(c) for c in l.columns], "lastobs_values": np.asarray(l.values, dtype="float64").tolist(),
            "exclude": excl,
            "out_usgs_index": [int(x) for x in us.index], "out_usgs_shape": list(us.shape),
            "out_usgs_values": np.asarray(us.values, dtype="float64").tolist(),
            "out_lastobs_index": [int(x) for x in ls.index], "out_lastobs_cols": [str(c) for c in ls.columns],
            "out_byseg": [int(x) for x in byseg], "out_byreach": [int(x) for x in byreach],
            "out_bygage": [int(x) for x in bygage],
        })
    with open(os.path.join(HERE, "da_prep_vectors.json"), "w") as f:
        json.dump({"seg_ids": [int(x) for x in seg_ids], "reaches": reaches, "cases": cases}, f)
    print("da_prep_vectors.json:", len(cases), "cases")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "da_prep":
        da_prep_vectors(import_ref_nhd_network())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "files":
        file_format_vectors()
        sys.exit(0)
    O.build()
    nudging_vectors()
    nn = import_ref_nhd_network()
    kernel_vectors()
    toy_network(nn)
    lowercolorado(nn)
    da_prep_vectors(nn)
    file_format_vectors()
