#!/usr/bin/env python3
"""Fixture of the reference's SHIPPED LowerColorado configuration (test/LowerColorado_TX/test_AnA.yaml: streamflow
nudging at the real USGS gages :71-93, level-pool waterbodies :17, hybrid diffusive domain :46) -- data only, read out of
the reference's own files with troute_amd.h5 in the dev container (the files do not travel to the GPU box):

  domain/RouteLink.nc                 gages[link]   -> the segments that carry a gage (reaches are split there,
                                                      nhd_network.py:319-338)
  usgs_TimeSlice/*.usgsTimeSlice.ncdf 15-min observations + quality per station, 2021-08-23 00:00 .. 23:45
  lastobs/nudgingLastObs.2021-08-23_12:00:00.nc    last valid observation and its time per station

Written: tests/golden/lowercolorado_gages.npz
  gage_ids [ng] int64 (ascending), station [ng] S15,
  usgs [ng, nobs] float32: observations on the routing grid (dt = 300 s, column k = t0 + k * dt, t0 = 2021-08-23 13:00,
      the restart time of the shipped configuration), NaN where there is none.  Prepared in outline like the reference's
      get_obs_from_timeslices (nhd_io.py:1000-1200: quality screen, 15-min stamps onto the dt grid, linear interpolation
      across gaps of at most 59 minutes) -- the DataFrame preparation itself is outside the hot path (SURVEY section 2,
      component 13); what the test pins is the routing WITH these real gage positions and observation records.
  lastobs_discharge [ng] float32, time_since_lastobs [ng] float32 (seconds, <= 0; NaN where the file has no station)

    python tests/golden/make_shipped_config_fixture.py
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from troute_amd import h5  # noqa: E402

REF = "/root/reference/test/LowerColorado_TX"
T0 = np.datetime64("2021-08-23T13:00:00")
DT = 300


def chars(a):
    """[n, width] array of 1-byte strings -> n stripped python strings"""
    a = np.asarray(a)
    return np.array([b"".join(r.tolist()).decode("ascii", "ignore").strip("\x00 ") for r in a], dtype=object)


def stamp(a):
    s = b"".join(np.asarray(a).tolist()).decode("ascii").strip("\x00 ").replace("_", "T")
    return np.datetime64(s)


with h5.File(os.path.join(REF, "domain", "RouteLink.nc")) as f:
    link = f.read("link").astype(np.int64)
    gages = chars(f.read("gages"))
has = np.array([g != "" for g in gages])
seg_of_station = {g: int(s) for g, s in zip(gages[has], link[has])}
print(len(seg_of_station), "gaged segments in RouteLink")

obs = {}                                        # station -> {time: value}
for path in sorted(glob.glob(os.path.join(REF, "usgs_TimeSlice", "*.usgsTimeSlice.ncdf"))):
    with h5.File(path) as f:
        st = chars(f.read("stationId"))
        q = f.read("discharge").astype(np.float32)
        ql = f.read("discharge_quality").astype(np.int64)
        tm = f.read("time")
    for s, v, k, t in zip(st, q, ql, tm):
        if s in seg_of_station and k >= 100 and v > 0:             # quality screen: qc_threshold 1.0 (x 100 in the file)
            obs.setdefault(s, {})[stamp(t)] = float(v)
stations = sorted(obs, key=lambda s: seg_of_station[s])
print(len(stations), "of them have observations on 2021-08-23")

nobs = 12 * 24 + 1
grid = T0 + np.arange(nobs) * np.timedelta64(DT, "s")
usgs = np.full((len(stations), nobs), np.nan, np.float32)
for i, s in enumerate(stations):
    ts = np.array(sorted(obs[s]))
    vs = np.array([obs[s][t] for t in ts], np.float64)
    x = (ts - T0) / np.timedelta64(1, "s")
    gx = (grid - T0) / np.timedelta64(1, "s")
    for k, g in enumerate(gx):
        j = np.searchsorted(x, g)
        if j < len(x) and x[j] == g:
            usgs[i, k] = vs[j]
        elif 0 < j < len(x) and x[j] - x[j - 1] <= 59 * 60:        # interpolation across short gaps only
            w = (g - x[j - 1]) / (x[j] - x[j - 1])
            usgs[i, k] = np.float32(vs[j - 1] * (1 - w) + vs[j] * w)

with h5.File(os.path.join(REF, "lastobs", "nudgingLastObs.2021-08-23_12:00:00.nc")) as f:
    lst = chars(f.read("stationId"))
    ld = f.read("discharge").astype(np.float32)
    lt = f.read("time")
    lq = f.read("discharge_quality").astype(np.int64)
lv = np.full(len(stations), np.nan, np.float32)
ltime = np.full(len(stations), np.nan, np.float32)
for s, d, t, k in zip(lst, ld, lt, lq):
    if s not in stations:
        continue
    good = np.flatnonzero((k >= 100) & (d > 0))
    if good.size:
        j = good[-1]
        i = stations.index(s)
        lv[i] = d[j]
        ltime[i] = np.float32((stamp(t[j]) - T0) / np.timedelta64(1, "s"))
out = os.path.join(ROOT, "tests", "golden", "lowercolorado_gages.npz")
np.savez_compressed(out, gage_ids=np.array([seg_of_station[s] for s in stations], np.int64),
                    station=np.array(stations, dtype="S15"), usgs=usgs, lastobs_discharge=lv, time_since_lastobs=ltime,
                    t0=str(T0), dt=DT)
print("wrote", out, usgs.shape, "obs columns with data:", int((~np.isnan(usgs)).any(0).sum()),
      "lastobs stations:", int((~np.isnan(lv)).sum()))
