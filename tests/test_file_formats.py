"""On-disk formats either side of the path: the HDF5/NetCDF-4 reader against h5dump's view of
the reference's own data files (tests/golden/lowercolorado_files/, copied from test/LowerColorado_TX), the
nhd_io mirrors, and the device-side ingest of packed CHRTOUT forcing (bit-exact against the unpacking rule)."""
import datetime
import os

import numpy as np
import pandas as pd
import pytest

import helpers as H
from troute_amd import h5, nhd_io

FILES = os.path.join(H.GOLDEN, "lowercolorado_files")
CHRT = [os.path.join(FILES, "202108231300.CHRTOUT_DOMAIN1"), os.path.join(FILES, "202108231400.CHRTOUT_DOMAIN1")]
RST = os.path.join(FILES, "HYDRO_RST.2021-08-23_12_00_DOMAIN1")
EXP = np.load(os.path.join(H.GOLDEN, "lowercolorado_files_expected.npz"))


def test_hdf5_reader_equals_h5dump():
    for k, path in enumerate(CHRT):
        with h5.File(path) as f:
            assert {"feature_id", "qBucket", "qSfcLatRunoff", "q_lateral", "streamflow"} <= set(f.names())
            for v in ("feature_id", "qBucket", "qSfcLatRunoff", "streamflow"):
                got = f.read(v)
                assert got.dtype == EXP[f"{v}_{k}"].dtype and np.array_equal(got, EXP[f"{v}_{k}"])
            pk = f.packing("qBucket")
            assert pk["scale"].dtype == np.float32 and pk["scale"] == np.float32(1e-5)
            assert pk["offset"] == np.float32(0) and list(pk["fills"]) == [-999900000, -999900000]
            assert (pk["vmin"], pk["vmax"]) == (0, 2000000000)
            assert f.attr(None, "model_output_valid_time") == (b"2021-08-23_13:00:00", b"2021-08-23_14:00:00")[k]
    with h5.File(RST) as f:
        for v in ("qlink1", "qlink2", "hlink"):
            assert np.array_equal(f.read(v).view(np.uint32), EXP[v].view(np.uint32))
    with pytest.raises(OSError):
        h5.File(os.path.join(FILES, "absent.nc"))
    with h5.File(RST) as f, pytest.raises(KeyError):
        f.read("nope")


def test_unpack_rule_masks_fills_and_range():
    raw = np.array([5, -999900000, -3, 2000000001, 0, 2000000000], dtype=np.int32)
    pk = {"scale": np.float32(1e-5), "offset": np.float32(0.0), "fills": [np.int32(-999900000)], "vmin": np.int32(0),
          "vmax": np.int32(2000000000)}
    v, m = h5.unpack(raw, pk, 0.0)
    assert v.dtype == np.float64 and m.tolist() == [False, True, True, True, False, False]
    assert v[0] == 5 * np.float64(np.float32(1e-5)) and v[1] == v[2] == v[3] == 0.0 and v[5] == 2000000000 * np.float64(np.float32(1e-5))


def test_get_ql_from_chrtout_equals_domain_fixture():
    """qBucket + qSfcLatRunoff, joined on feature_id, is the forcing the routing fixtures were made with."""
    lc = H.LowerColorado()
    for k, path in enumerate(CHRT):
        dat = nhd_io.get_ql_from_chrtout(path)
        assert dat.dtype == np.float64 and dat.shape == (11248,)
        fid = EXP[f"feature_id_{k}"]
        pos = {int(v): i for i, v in enumerate(fid)}
        take = np.array([pos[int(s)] for s in lc.ids])
        assert np.array_equal(dat[take].astype(np.float32).view(np.uint32), lc.qlat[:, k].view(np.uint32))
    df = nhd_io.get_ql_from_wrf_hydro_mf(CHRT)
    assert df.shape == (11248, 2) and list(df.columns) == [pd.Timestamp("2021-08-23 13:00"), pd.Timestamp("2021-08-23 14:00")]
    # q_lateral (the files carry it) is the sum of the two fluxes to its own packing resolution (0.1 m3/s)
    assert np.nanmax(np.abs(df.iloc[:, 0].values - nhd_io.get_ql_from_chrtout(CHRT[0]))) <= 0.051
    rn = nhd_io.read_netcdf(CHRT[0])
    assert len(rn) == 11248 and {"feature_id", "streamflow", "qBucket"} <= set(rn.columns)
    assert np.array_equal(rn["feature_id"].values, EXP["feature_id_0"])
    ok = EXP["streamflow_0"] != -999900
    assert np.array_equal(rn["streamflow"].values[ok], EXP["streamflow_0"][ok] * np.float64(np.float32(0.01)))


def test_restart_readers_and_writers(tmp_path):
    xwalk = tmp_path / "RouteLink_links.nc"
    with h5.File(xwalk, "w") as f:
        f.write("link", EXP["link"])
    q0 = nhd_io.get_channel_restart_from_wrf_hydro(RST, xwalk, "link")
    assert list(q0.columns) == ["qu0", "qd0", "h0"] and q0.index.name == "link" and len(q0) == 11141
    assert np.array_equal(q0["qu0"].values, EXP["qlink1"]) and np.array_equal(q0["h0"].values, EXP["hlink"])
    assert np.array_equal(q0.index.values, EXP["link"][:11141])
    t0 = datetime.datetime(2021, 8, 23, 13, 0)
    wb = pd.DataFrame({"qd0": [1.0], "h0": [2.0], "LkArea": [3.0]}, index=[77])
    ch_path, wb_path = nhd_io.write_lite_restart(q0, wb, t0, {"lite_restart_output_directory": str(tmp_path)})
    assert ch_path.name == "channel_restart_202108231300" and wb_path.name == "waterbody_restart_202108231300"
    back, t_back = nhd_io.read_lite_restart(ch_path)
    assert t_back == t0 and back.equals(q0)
    wback, _ = nhd_io.read_lite_restart(wb_path)
    assert list(wback.columns) == ["qd0", "h0"]
    assert nhd_io.write_lite_restart(q0, wb, t0, {}) is None


def test_write_flowveldepth_netcdf_layout(tmp_path):
    """Variables, attributes and fill values of the reference's writer (nhd_io.py:2089-2235), in a NetCDF-4 container:
    the dimensions feature_id, time and type_strlen (:2100-2104) exist as HDF5 dimension scales and are attached to
    every variable -- the structure the netCDF-4 library writes, read off the reference's own NetCDF files below."""
    n, nt = 7, 5
    rng = np.random.default_rng(0)
    idx = pd.MultiIndex.from_arrays([np.arange(100, 100 + n), ["ch"] * (n - 1) + ["wb"]], names=["featureID", "Type"])
    fr = [pd.DataFrame(rng.random((n, nt)).astype(np.float32), index=idx) for _ in range(4)]
    t0 = datetime.datetime(2021, 8, 23, 13, 0)
    path = nhd_io.write_flowveldepth_netcdf(tmp_path, "fvd.nc", fr[0], fr[1], fr[2], fr[3], np.arange(nt) * 300.0, t0)
    # the reference writer's variables and their attributes, name for name
    want_attrs = {"time": {"long_name", "standard_name", "units", "missing_value", "_FillValue"},
                  "feature_id": {"long_name"}, "type": {"long_name"}}
    for v in ("flow", "velocity", "depth", "nudge"):
        want_attrs[v] = {"long_name", "units", "missing_value", "_FillValue"}
    with h5.File(path) as f:
        assert set(f.names()) == set(want_attrs) | {"type_strlen"}
        for v, names in want_attrs.items():
            for a in names:
                assert f.has_attr(v, a), (v, a)
        assert f.shape("flow") == (n, nt) and f.read("flow").dtype == np.float32
        for name, frame in zip(("flow", "velocity", "depth", "nudge"), fr):
            assert np.array_equal(f.read(name), frame.values)
        assert np.array_equal(f.read("feature_id"), np.arange(100, 100 + n))
        assert f.attr("flow", "units") == b"m3 s-1" and f.attr("time", "units") == b"seconds since 2021-08-23 13:00:00"
        assert f.attr("flow", "long_name") == b"Flow" and f.attr("nudge", "long_name") == b"Streamflow Nudge Value"
        assert f.attr("velocity", "units") == b"m/s" and f.attr("depth", "units") == b"m"
        assert f.attr("flow", "_FillValue")[0] == np.float32(-9999.0) and f.attr("flow", "missing_value")[0] == np.float32(-9999.0)
        assert f.attr("time", "_FillValue")[0] == -9999.0 and f.attr("time", "standard_name") == b"time"
        assert f.attr(None, "TITLE") == b"OUTPUT FROM T-ROUTE" and f.attr(None, "file_reference_time") == b"2021-08-23_13:00:00"
        assert f.has_attr(None, "code_version")
        assert b"".join(f.read("type")[-1].tolist()) == b"wb" and f.shape("type") == (n, 2)
        # NetCDF-4 structure: dimensions are dimension scales, variables reference them
        for dim in ("feature_id", "time", "type_strlen"):
            assert f.attr(dim, "CLASS") == b"DIMENSION_SCALE" and f.has_attr(dim, "NAME") and f.has_attr(dim, "_Netcdf4Dimid")
        assert f.attr("type_strlen", "NAME").startswith(b"This is a netCDF dimension but not a netCDF variable.")
        for v in ("flow", "velocity", "depth", "nudge", "type"):
            assert f.has_attr(v, "DIMENSION_LIST")
        for dim in ("feature_id", "time"):
            assert f.has_attr(dim, "REFERENCE_LIST")
    # the same structural attributes on a file the netCDF-4 library itself wrote (the reference's CHRTOUT)
    with h5.File(CHRT[0]) as f:
        assert f.attr("feature_id", "CLASS") == b"DIMENSION_SCALE" and f.has_attr("feature_id", "NAME")
        assert f.has_attr("feature_id", "_Netcdf4Dimid") and f.has_attr("feature_id", "REFERENCE_LIST")
        assert f.has_attr("streamflow", "DIMENSION_LIST")


def test_write_flowveldepth_from_a_device_decimated_frame_writes_the_same_files(tmp_path):
    """``write_flowveldepth`` (the reference's stream-output writer, nhd_io.py:2348-2462): hourly files from 5-minute steps,
    once from the full frame and once from the frame ``compute_nhd_routing_v02(..., output_stride=12)`` returns (every
    twelfth step, decimated on the device): the same bytes, the reference's step selection (:2379-2382) and file names."""
    import filecmp
    rng = np.random.default_rng(0)
    n, nts, dt, stride = 9, 48, 300, 12
    full = rng.random((n, nts * 3)).astype(np.float32)
    ids = np.arange(100, 100 + n)
    dec = full.reshape(n, nts, 3)[:, stride - 1::stride, :].reshape(n, -1)
    nudge = np.zeros((3, nts + 1), np.float32)
    nudge[:, 1:] = rng.random((3, nts))
    gids = ids[[1, 4, 7]]
    t0 = datetime.datetime(2021, 8, 23, 13, 0)
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    for timediff, names in ((-1, ["troute_output_202108231300.nc"]),
                            (2, ["troute_output_202108231300.nc", "troute_output_202108231500.nc"])):
        fa = nhd_io.write_flowveldepth(a, None, pd.DataFrame(full, index=ids), nudge, gids, t0, dt, timediff, ".nc", 60)
        fb = nhd_io.write_flowveldepth(b, None, pd.DataFrame(dec, index=ids), nudge, gids, t0, dt, timediff, ".nc", 60,
                                       output_stride=stride)
        assert [os.path.basename(x) for x in fa] == names
        assert all(filecmp.cmp(x, y, shallow=False) for x, y in zip(fa, fb))
    with h5.File(fa[0]) as f:
        assert f.shape("flow") == (n, 2) and np.array_equal(f.read("time"), [3600.0, 7200.0])
        assert np.array_equal(f.read("flow"), full[:, 0::3][:, [11, 23]])
        assert np.array_equal(f.read("depth"), full[:, 2::3][:, [11, 23]])
        nd = f.read("nudge")
        assert np.array_equal(nd[1], nudge[0, 1:][[11, 23]]) and (nd[0] == -9999.0).all()
    with pytest.raises(ValueError, match="must divide"):
        nhd_io.write_flowveldepth(b, None, pd.DataFrame(dec, index=ids), nudge, gids, t0, dt, -1, ".nc", 60, output_stride=5)


def test_decoding_of_planted_fill_and_out_of_range_cells_against_h5dump(tmp_path):
    """A packed variable written with cells planted at _FillValue, at missing_value, below valid_min and above valid_max:
    h5dump's independent view of the bytes and attributes on disk, decoded by the CF rule the forcing ingest implements
    (masked -> 0, else raw * scale_factor + add_offset in double), equals the reader + decoder of this package.  (The
    rule itself is netCDF4-python's documented default; that library is not in the image, so this pins everything BUT
    the library's own behaviour -- DESIGN.md states it.)"""
    import subprocess
    raw = np.array([7, -999900000, -888800000, -1, 2000000001, 0, 2000000000, 123456], dtype=np.int32)
    path = os.path.join(tmp_path, "packed.nc")
    with h5.File(path, "w") as f:
        f.write("feature_id", np.arange(raw.shape[0], dtype=np.int64), {}, dims=["feature_id"])
        f.write("qBucket", raw, {"scale_factor": np.float32(1e-5), "add_offset": np.float32(0.0),
                                 "_FillValue": np.int32(-999900000), "missing_value": np.int32(-888800000),
                                 "valid_range": np.array([0, 2000000000], np.int32)}, dims=["feature_id"])
    dump = os.path.join(tmp_path, "q.bin")
    h5dump = "/opt/conda/bin/h5dump"
    if not os.path.exists(h5dump):
        pytest.skip("h5dump not available")
    subprocess.run([h5dump, "-d", "/qBucket", "-b", "LE", "-o", dump, path], check=True, stdout=subprocess.DEVNULL)
    on_disk = np.fromfile(dump, dtype="<i4")
    assert np.array_equal(on_disk, raw)
    txt = subprocess.run([h5dump, "-A", "-d", "/qBucket", path], check=True, capture_output=True, text=True).stdout
    assert "-999900000" in txt and "-888800000" in txt and "2000000000" in txt and "1e-05" in txt
    masked = (on_disk == -999900000) | (on_disk == -888800000) | (on_disk < 0) | (on_disk > 2000000000)
    want = np.where(masked, 0.0, on_disk.astype(np.float64) * np.float64(np.float32(1e-5)) + 0.0)
    with h5.File(path) as f:
        pk = f.packing("qBucket")
        got, m = h5.unpack(f.read("qBucket"), pk, 0.0)
    assert m.tolist() == masked.tolist() and np.array_equal(got, want)
    assert sorted(int(x) for x in pk["fills"]) == [-999900000, -888800000] and (pk["vmin"], pk["vmax"]) == (0, 2000000000)


def test_chrtout_packed_host_rule():
    pk = nhd_io.chrtout_packed(CHRT)
    assert pk["raw_a"].shape == (2, 11248) and pk["raw_b"].shape == (2, 11248)
    assert np.array_equal(pk["raw_a"][1], EXP["qBucket_1"]) and np.array_equal(pk["raw_b"][0], EXP["qSfcLatRunoff_0"])
    assert pk["pack_a"].tolist() == [float(np.float32(1e-5)), 0.0, -999900000.0, -999900000.0, 0.0, 2000000000.0]
    host = nhd_io.unpack_packed(pk)
    for k in range(2):
        assert np.array_equal(host[k], nhd_io.get_ql_from_chrtout(CHRT[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [32, 64])
def test_gpu_ingest_of_packed_forcing_is_bit_exact_and_routes_identically(precision):
    from troute_amd.plan import RoutingPlan
    lc = H.LowerColorado()
    pk = nhd_io.chrtout_packed(CHRT)
    rng = np.random.default_rng(5)
    # make the rule's branches fire: fill values, out-of-range entries, ids that are not in the files
    pk["raw_a"] = pk["raw_a"].copy()
    pk["raw_b"] = pk["raw_b"].copy()
    pk["raw_a"][0, rng.choice(11248, 200, replace=False)] = -999900000
    pk["raw_b"][1, rng.choice(11248, 200, replace=False)] = -7
    pk["raw_a"][1, rng.choice(11248, 50, replace=False)] = 2000000001
    ids = lc.ids.copy()
    host = nhd_io.unpack_packed(pk)                                       # [2, nfeat] float64
    pos = {int(v): i for i, v in enumerate(pk["feature_id"])}
    take = np.array([pos[int(s)] for s in ids])
    qlat = host[:, take].T.astype(np.float32)                             # the reference's cast
    missing = rng.choice(lc.nseg, 25, replace=False)
    ids_m = ids.copy()
    ids_m[missing] = 9_000_000_000 + np.arange(25)                        # rows the files do not know: inflow 0
    qlat_m = qlat.copy()
    qlat_m[missing] = 0.0
    up_ptr, up_idx = lc.csr()
    nsteps, qts = 24, 12
    dt = np.float32 if precision == 32 else np.float64
    with RoutingPlan(up_ptr, up_idx, lc.params9, precision=precision) as plan:
        want = plan.route(nsteps, qts, True, qlat_m.astype(dt), lc.q0.astype(dt))
        feat_of_row = plan.upload_forcing_packed(nsteps, pk, ids_m, lc.q0.astype(dt))
        assert (feat_of_row[missing] == -1).all() and (np.delete(feat_of_row, missing) >= 0).all()
        plan.route_device(nsteps, qts, True)
        got = plan.download_fvd()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        # a second window through the ordinary upload still works (the transposed staging is rebuilt)
        again = plan.route(nsteps, qts, True, qlat_m.astype(dt), lc.q0.astype(dt))
        assert np.array_equal(again.view(np.uint8), want.view(np.uint8))
