"""On-disk formats either side of the routing path without netCDF4 / xarray.

Same names and return shapes as the reference's readers/writers in
``src/troute-network/troute/nhd_io.py`` for the files the Muskingum-Cunge path touches:

  read_netcdf                          :25-44    a NetCDF-4 file's 1-D variables as a DataFrame
  get_ql_from_chrtout                  :397-434  lateral inflow of one CHRTOUT file (qBucket + qSfcLatRunoff)
  get_ql_from_wrf_hydro_mf             :437-510  ... of a list of files, [feature_id x time]
  get_channel_restart_from_wrf_hydro   :1368-1430 (qu0, qd0, h0) from a HYDRO_RST file
  read_lite_restart / write_lite_restart :1433-1504 pandas pickles of the state frames
  write_flowveldepth_netcdf            :2089-2235 flow / velocity / depth / nudge [feature_id, time]
  write_flowveldepth                   :2348-2462 the stream-output files of a run from its flowveldepth frame (no mask file)

plus ``chrtout_packed`` -- the raw packed columns and packing facts of a list of CHRTOUT files, which
``RoutingPlan.upload_forcing_packed`` hands to the device so that decoding, the join on feature id and
the forcing layout happen in one kernel (include/trmc.h trmc_upload_forcing_packed).

NetCDF-4 files are HDF5 files; ``troute_amd.h5`` reads them through libhdf5.  Unpacking follows
netCDF4-python's default read (what the reference relies on): entries equal to _FillValue /
missing_value or outside valid_range are masked, the rest are ``raw * scale_factor + add_offset`` with
the attributes in their stored type (int32 x float32 scalar -> float64).  netCDF4 / xarray themselves
are not in this image, so the rule is pinned to its documentation, not to the library.
"""
import os
import pathlib

import numpy as np

from . import h5


def _main_length(f, names):
    """length of the dimension most 1-D variables share (ds.to_dataframe() of a single-dimension file)"""
    lens = {}
    for n in names:
        try:
            shp = f.shape(n)
        except KeyError:
            continue
        if len(shp) == 1:
            lens[shp[0]] = lens.get(shp[0], 0) + 1
    return max(lens, key=lens.get) if lens else 0


def read_netcdf(geo_file_path):
    """The 1-D numeric variables along the file's main dimension as a DataFrame (reference: xarray's
    ``ds.to_dataframe()``, nhd_io.py:25-44); packed variables are unpacked, masked entries become NaN."""
    import pandas as pd
    with h5.File(geo_file_path) as f:
        names = f.names()
        n = _main_length(f, names)
        cols = {}
        for name in names:
            try:
                if f.shape(name) != (n,):
                    continue
                raw = f.read(name)
            except (TypeError, KeyError):
                continue
            pk = f.packing(name)
            if pk["scale"] is not None or pk["offset"] is not None or (pk["fills"] and raw.dtype.kind == "f"):
                vals, mask = h5.unpack(raw, pk)
                vals = np.array(vals, dtype=np.float64 if vals.dtype.kind != "f" else vals.dtype)
                vals[mask] = np.nan
                cols[name] = vals
            else:
                cols[name] = raw
    return pd.DataFrame(cols)


def _unpacked(f, name, fill_value=0.0):
    vals, _ = h5.unpack(f.read(name), f.packing(name), fill_value)
    return vals


def get_ql_from_chrtout(f, qlateral_varname="q_lateral", qbucket_varname="qBucket",
                        runoff_varname="qSfcLatRunoff"):
    """Lateral inflow of ONE CHRTOUT file, file order: qBucket + qSfcLatRunoff when both exist, else
    q_lateral; masked entries count as 0 (nhd_io.py:397-434)."""
    with h5.File(f) as ds:
        if qbucket_varname in ds and runoff_varname in ds:
            return _unpacked(ds, qbucket_varname) + _unpacked(ds, runoff_varname)
        return _unpacked(ds, qlateral_varname)


def chrtout_packed(qlat_files, index_col="feature_id", value_col="q_lateral", gw_col="qBucket",
                   runoff_col="qSfcLatRunoff"):
    """Raw columns of a list of CHRTOUT files for the device-side ingest.

    Returns dict(feature_id int64[nfeat], raw_a int32[nfiles, nfeat], raw_b int32[nfiles, nfeat] | None,
    pack_a float64[6], pack_b float64[6] | None, times [nfiles] bytes) with pack = (scale_factor, add_offset,
    _FillValue, missing_value, valid_min, valid_max), NaN where the file defines none.  Every file must
    carry the same feature axis (the reference stacks them along time without re-indexing as well)."""
    feat, a, b, pa, pb, times = None, [], [], None, None, []

    def spec(ds, name):
        pk = ds.packing(name)
        fills = list(pk["fills"]) + [np.nan, np.nan]
        return np.array([1.0 if pk["scale"] is None else np.float64(pk["scale"]),
                         0.0 if pk["offset"] is None else np.float64(pk["offset"]),
                         fills[0], fills[1],
                         np.nan if pk["vmin"] is None else pk["vmin"],
                         np.nan if pk["vmax"] is None else pk["vmax"]], dtype=np.float64)

    for path in qlat_files:
        with h5.File(path) as ds:
            fid = ds.read(index_col).astype(np.int64).ravel()
            if feat is None:
                feat = fid
            elif not np.array_equal(fid, feat):
                raise ValueError(f"{path}: feature axis differs from the first file's")
            both = gw_col in ds and runoff_col in ds
            va, vb = (gw_col, runoff_col) if both else (value_col, None)
            ra = ds.read(va)
            if ra.dtype != np.int32:
                raise TypeError(f"{path}:{va} is {ra.dtype}, expected packed int32")
            a.append(ra.ravel())
            spa = spec(ds, va)
            if pa is None:
                pa = spa
            elif not np.array_equal(spa, pa, equal_nan=True):
                raise ValueError(f"{path}: packing of {va} differs from the first file's")
            if vb is not None:
                rb = ds.read(vb)
                if rb.dtype != np.int32:
                    raise TypeError(f"{path}:{vb} is {rb.dtype}, expected packed int32")
                b.append(rb.ravel())
                spb = spec(ds, vb)
                if pb is None:
                    pb = spb
                elif not np.array_equal(spb, pb, equal_nan=True):
                    raise ValueError(f"{path}: packing of {vb} differs from the first file's")
            times.append(ds.attr(None, "model_output_valid_time", b""))
    if b and len(b) != len(a):
        raise ValueError("some files carry qBucket/qSfcLatRunoff and some do not")
    return {"feature_id": feat, "raw_a": np.ascontiguousarray(np.stack(a)),
            "raw_b": np.ascontiguousarray(np.stack(b)) if b else None,
            "pack_a": pa, "pack_b": pb if b else None, "times": times}


def unpack_packed(packed):
    """Host evaluation of the ingest rule on chrtout_packed() output: float64 [nfiles, nfeat] (what the
    device kernel computes before its cast to float32)."""
    def one(raw, pk):
        mask = np.zeros(raw.shape, dtype=bool)
        for k in (2, 3):
            if not np.isnan(pk[k]):
                mask |= raw == np.int32(pk[k])
        if not np.isnan(pk[4]):
            mask |= raw < np.int32(pk[4])
        if not np.isnan(pk[5]):
            mask |= raw > np.int32(pk[5])
        return np.where(mask, 0.0, raw.astype(np.float64) * pk[0] + pk[1])
    v = one(packed["raw_a"], packed["pack_a"])
    if packed["raw_b"] is not None:
        v = v + one(packed["raw_b"], packed["pack_b"])
    return v


def get_ql_from_wrf_hydro_mf(qlat_files, index_col="feature_id", value_col="q_lateral", gw_col="qBucket",
                             runoff_col="qSfcLatRunoff"):
    """[feature_id x time] DataFrame of lateral inflow from a list of CHRTOUT files (nhd_io.py:437-510;
    value_col when the files carry it, else gw_col + runoff_col), columns = the files' valid times."""
    import pandas as pd
    cols, feat, times = [], None, []
    for path in qlat_files:
        with h5.File(path) as ds:
            fid = ds.read(index_col).astype(np.int64).ravel()
            feat = fid if feat is None else feat
            if value_col in ds:
                v, m = h5.unpack(ds.read(value_col), ds.packing(value_col))
                v = np.array(v, dtype=np.float64)
                v[m] = np.nan                                   # xarray decodes masked entries as NaN
            else:
                va, ma = h5.unpack(ds.read(gw_col), ds.packing(gw_col))
                vb, mb = h5.unpack(ds.read(runoff_col), ds.packing(runoff_col))
                v = np.array(va, dtype=np.float64) + np.array(vb, dtype=np.float64)
                v[ma | mb] = np.nan
            cols.append(v)
            t = ds.attr(None, "model_output_valid_time", b"").decode()
            times.append(pd.to_datetime(t, format="%Y-%m-%d_%H:%M:%S") if t else len(times))
    return pd.DataFrame(np.stack(cols, 1), index=feat, columns=times)


def get_channel_restart_from_wrf_hydro(channel_initial_states_file, crosswalk_file, channel_ID_column,
                                       us_flow_column="qlink1", ds_flow_column="qlink2", depth_column="hlink",
                                       default_us_flow_column="qu0", default_ds_flow_column="qd0",
                                       default_depth_column="h0"):
    """(qu0, qd0, h0) of a WRF-Hydro HYDRO_RST file, indexed by the ids of `crosswalk_file` (RouteLink) in
    file order -- the restart file lists channels in RouteLink order (nhd_io.py:1368-1430)."""
    import pandas as pd
    with h5.File(crosswalk_file) as x:
        ids = x.read(channel_ID_column).ravel()
    with h5.File(channel_initial_states_file) as q:
        qu = q.read(us_flow_column).ravel()
        qd = q.read(ds_flow_column).ravel()
        h = q.read(depth_column).ravel() if depth_column in q else np.zeros_like(qu)
    n = min(len(ids), len(qu))
    df = pd.DataFrame({default_us_flow_column: qu[:n], default_ds_flow_column: qd[:n], default_depth_column: h[:n]},
                      index=pd.Index(ids[:n], name=channel_ID_column))
    return df


def read_lite_restart(file):
    """(state DataFrame, t0) from a lite restart pickle (nhd_io.py:1433-1455)."""
    import pandas as pd
    df = pd.read_pickle(pathlib.Path(file))
    t0 = df["time"].iloc[0].to_pydatetime()
    return df.drop(columns="time"), t0


def write_lite_restart(q0, waterbodies_df, t0, restart_parameters):
    """Channel (and waterbody) state frames as pickles named by t0 (nhd_io.py:1458-1504)."""
    out_dir = restart_parameters.get("lite_restart_output_directory", None)
    if not out_dir:
        return None
    out = pathlib.Path(out_dir)
    stamp = t0.strftime("%Y%m%d%H%M")
    ch = q0.copy()
    ch["time"] = t0
    ch_path = out / ("channel_restart_" + stamp)
    ch.to_pickle(ch_path)
    wb_path = None
    if waterbodies_df is not None and not waterbodies_df.empty:
        wb = waterbodies_df.loc[:, ["qd0", "h0"]].copy()
        wb["time"] = t0
        wb_path = out / ("waterbody_restart_" + stamp)
        wb.to_pickle(wb_path)
    return ch_path, wb_path


def write_flowveldepth_netcdf(stream_output_directory, file_name, flow, velocity, depth, nudge_df, timestamps, t0):
    """flow / velocity / depth / nudge [feature_id, time] float32 with the reference's variable names,
    attributes and fill values (nhd_io.py:2089-2235), written as HDF5 (NetCDF-4's container).  `flow` is
    indexed by (featureID, Type) or by feature id alone."""
    idx = flow.index
    if getattr(idx, "nlevels", 1) > 1:
        fid = np.asarray(idx.get_level_values("featureID"), dtype=np.int64)
        typ = [str(x) for x in idx.get_level_values("Type")]
    else:
        fid = np.asarray(idx, dtype=np.int64)
        typ = ["ch"] * len(fid)
    path = os.path.join(os.fspath(stream_output_directory), file_name)
    with h5.File(path, "w") as f:
        # dimensions feature_id, time, type_strlen (nhd_io.py:2100-2104) as HDF5 dimension scales: feature_id and time
        # are coordinate variables, type_strlen a bare dimension; every variable has them attached
        f.write("feature_id", fid, {"long_name": "Segment ID"}, dims=["feature_id"])
        f.write("time", np.asarray(timestamps, dtype=np.float64),
                {"long_name": "valid output time", "standard_name": "time",
                 "units": f"seconds since {t0.strftime('%Y-%m-%d %H:%M:%S')}", "missing_value": np.float64(-9999.0),
                 "_FillValue": np.float64(-9999.0)}, dims=["time"])
        width = max([len(t) for t in typ] + [1])
        f.dimension("type_strlen", width)
        f.write_chars("type", [t.encode().ljust(width, b"\0") for t in typ], width, {"long_name": "Type"},
                      dims=["feature_id", "type_strlen"])
        for name, frame, long_name, units in (("flow", flow, "Flow", "m3 s-1"), ("velocity", velocity, "Velocity", "m/s"),
                                              ("depth", depth, "Depth", "m"),
                                              ("nudge", nudge_df, "Streamflow Nudge Value", "m3 s-1")):
            f.write(name, np.asarray(frame, dtype=np.float32),
                    {"long_name": long_name, "units": units, "missing_value": np.float32(-9999.0),
                     "_FillValue": np.float32(-9999.0)}, dims=["feature_id", "time"])
        f.set_attr("TITLE", "OUTPUT FROM T-ROUTE")
        f.set_attr("file_reference_time", t0.strftime("%Y-%m-%d_%H:%M:%S"))
        f.set_attr("code_version", "")
    return path


def write_flowveldepth(stream_output_directory, stream_output_mask, flowveldepth, nudge, usgs_positions_id, t0, dt,
                       stream_output_timediff, stream_output_type, stream_output_internal_frequency=5, cpu_pool=1,
                       poi_crosswalk=None, nexus_dict=None, *, output_stride=1):
    """The reference's stream-output writer (nhd_io.py:2348-2462) for a run without a mask file: flow, velocity, depth and
    nudge at ``stream_output_internal_frequency`` minutes -- the steps ts, 2 ts, ... with ts = frequency // (dt // 60) -- in
    one file (``stream_output_timediff`` = -1) or one per ``stream_output_timediff`` hours, named
    ``troute_output_<YYYYmmddHHMM><type>``; every feature is labelled "wb" as the reference does (updated_flowveldepth,
    :2270-2274).  NetCDF only.

    ``output_stride`` (keyword-only): the frame already holds only every n-th timestep -- what
    ``compute_nhd_routing_v02(..., output_stride=n)`` returns, decimated on the device -- so the writer takes every
    (ts // n)-th of ITS columns; n must divide ts.  The files are the same bytes either way."""
    import datetime

    import pandas as pd
    if stream_output_mask is not None:
        raise NotImplementedError("stream-output mask files (HYFeatures nexus / POI selection) are outside the NHD Muskingum-Cunge path")
    if stream_output_type != ".nc":
        raise NotImplementedError("stream output as NetCDF only ('.nc')")
    n = int(output_stride)
    ts = int(stream_output_internal_frequency // (dt // 60))
    if n < 1 or ts % n:
        raise ValueError(f"output_stride {n} must divide the {ts} timesteps between two output times")
    n_cols = flowveldepth.shape[1] // 3
    ind = list(range(ts // n - 1, n_cols, ts // n))                  # (:2379-2381 on the decimated frame)
    timestamps_sec = [(i + 1) * n * dt for i in ind]
    idx = pd.MultiIndex.from_arrays([np.asarray(flowveldepth.index), ["wb"] * flowveldepth.shape[0]], names=["featureID", "Type"])
    values = np.asarray(flowveldepth)
    flow = pd.DataFrame(values[:, 0::3][:, ind], index=idx)
    velocity = pd.DataFrame(values[:, 1::3][:, ind], index=idx)
    depth = pd.DataFrame(values[:, 2::3][:, ind], index=idx)
    nudge = np.asarray(nudge)
    if nudge.shape[0] and np.all(nudge[:, 0] == 0):                   # (:2388-2391: the column of the initial time)
        nudge = nudge[:, 1:]
    full_ind = list(range(ts - 1, n_cols * n, ts))                    # the nudge array always has every timestep
    gage_rows = {int(g): k for k, g in enumerate(np.asarray(usgs_positions_id).tolist())}
    nudge_v = np.full((flowveldepth.shape[0], len(ind)), -9999.0, dtype=np.float64)
    for r, fid in enumerate(np.asarray(flowveldepth.index).tolist()):
        k = gage_rows.get(int(fid))
        if k is not None:
            nudge_v[r] = nudge[k][full_ind]
    nudge_df = pd.DataFrame(nudge_v, index=idx)
    written, file_time = [], t0
    if stream_output_timediff > 0:
        per_file = stream_output_timediff * 60 // stream_output_internal_frequency
        num_files = max(1, int(n_cols * n * dt // (stream_output_timediff * 60 * 60)))
        for k in range(num_files):
            sl = slice(k * per_file, (k + 1) * per_file)
            name = "troute_output_" + file_time.strftime("%Y%m%d%H%M") + stream_output_type
            written.append(write_flowveldepth_netcdf(stream_output_directory, name, flow.iloc[:, sl], velocity.iloc[:, sl],
                                                     depth.iloc[:, sl], nudge_df.iloc[:, sl], timestamps_sec[sl], t0))
            file_time = file_time + datetime.timedelta(hours=stream_output_timediff)
    elif stream_output_timediff == -1:
        name = "troute_output_" + file_time.strftime("%Y%m%d%H%M") + stream_output_type
        written.append(write_flowveldepth_netcdf(stream_output_directory, name, flow, velocity, depth, nudge_df, timestamps_sec, t0))
    return written
