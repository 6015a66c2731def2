"""Minimal HDF5 access over ctypes -- enough to read the NetCDF-4 files either side of the routing path
(RouteLink, CHRTOUT forcing, HYDRO_RST restart) and to write result files, without netCDF4 / xarray / h5py.

NetCDF-4 files are HDF5 files (magic ``\\x89HDF``): variables are datasets, variable attributes
(``scale_factor``, ``add_offset``, ``_FillValue``, ``missing_value``, ``valid_range``) are dataset attributes.
The reference reads them through netCDF4 / xarray (src/troute-network/troute/nhd_io.py); this module talks
to the system's libhdf5 (1.10) directly.  Only numeric datasets / attributes of native integer and float
types and fixed-length string attributes are supported -- what those files contain.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = ("/opt/conda/lib/libhdf5.so", "libhdf5.so", "libhdf5_serial.so")
_LIB = None

hid_t = C.c_int64
hsize_t = C.c_uint64

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_SGN_NONE = 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0


def lib():
    """libhdf5 (once).  Raises RuntimeError when the library cannot be found."""
    global _LIB
    if _LIB is not None:
        return _LIB
    names = list(_CANDIDATES)
    found = ctypes.util.find_library("hdf5")
    if found:
        names.append(found)
    err = None
    for n in ([os.environ["TRMC_HDF5_LIB"]] if os.environ.get("TRMC_HDF5_LIB") else []) + names:
        try:
            h = C.CDLL(n)
        except OSError as e:
            err = e
            continue
        sig = {
            "H5open": (C.c_int, []),
            "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
            "H5Fclose": (C.c_int, [hid_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dclose": (C.c_int, [hid_t]),
            "H5Dget_space": (hid_t, [hid_t]),
            "H5Dget_type": (hid_t, [hid_t]),
            "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (C.c_int, [hid_t]),
            "H5Tget_class": (C.c_int, [hid_t]),
            "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tis_variable_str": (C.c_int, [hid_t]),
            "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tcopy": (hid_t, [hid_t]),
            "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]),
            "H5Tclose": (C.c_int, [hid_t]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Aget_type": (hid_t, [hid_t]),
            "H5Aget_space": (hid_t, [hid_t]),
            "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]),
            "H5Aclose": (C.c_int, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Gget_num_objs": (C.c_int, [hid_t, C.POINTER(hsize_t)]),
            "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
            "H5Gget_objtype_by_idx": (C.c_int, [hid_t, hsize_t]),
            "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        h.H5open()
        h.H5Eset_auto2(0, None, None)          # errors are reported through return codes, not on stderr
        _LIB = h
        return h
    raise RuntimeError(f"libhdf5 not found ({err}); set TRMC_HDF5_LIB to its path")


_HL = None


def hl():
    """libhdf5_hl (dimension scales: how NetCDF-4 stores its dimensions), loaded beside libhdf5."""
    global _HL
    if _HL is not None:
        return _HL
    lib()
    err = None
    for n in ("/opt/conda/lib/libhdf5_hl.so", "libhdf5_hl.so", "libhdf5_serial_hl.so", ctypes.util.find_library("hdf5_hl")):
        if not n:
            continue
        try:
            h = C.CDLL(n)
        except OSError as e:
            err = e
            continue
        h.H5DSset_scale.restype, h.H5DSset_scale.argtypes = C.c_int, [hid_t, C.c_char_p]
        h.H5DSattach_scale.restype, h.H5DSattach_scale.argtypes = C.c_int, [hid_t, hid_t, C.c_uint]
        _HL = h
        return h
    raise RuntimeError(f"libhdf5_hl not found ({err})")


def _native(name):
    return hid_t.in_dll(lib(), name).value


_NP_TO_H5 = {
    np.dtype("int8"): "H5T_NATIVE_INT8_g", np.dtype("uint8"): "H5T_NATIVE_UINT8_g",
    np.dtype("int16"): "H5T_NATIVE_INT16_g", np.dtype("uint16"): "H5T_NATIVE_UINT16_g",
    np.dtype("int32"): "H5T_NATIVE_INT32_g", np.dtype("uint32"): "H5T_NATIVE_UINT32_g",
    np.dtype("int64"): "H5T_NATIVE_INT64_g", np.dtype("uint64"): "H5T_NATIVE_UINT64_g",
    np.dtype("float32"): "H5T_NATIVE_FLOAT_g", np.dtype("float64"): "H5T_NATIVE_DOUBLE_g",
}


def _np_dtype_of(tid):
    h = lib()
    cls, size = h.H5Tget_class(tid), h.H5Tget_size(tid)
    if cls == H5T_INTEGER:
        return np.dtype(("u" if h.H5Tget_sign(tid) == H5T_SGN_NONE else "i") + str(size))
    if cls == H5T_FLOAT:
        return np.dtype("f" + str(size))
    if cls == H5T_STRING:
        return np.dtype("S" + str(size))
    raise TypeError(f"unsupported HDF5 type class {cls}")


def _dims(sid):
    h = lib()
    nd = h.H5Sget_simple_extent_ndims(sid)
    if nd <= 0:
        return ()
    d = (hsize_t * nd)()
    h.H5Sget_simple_extent_dims(sid, d, None)
    return tuple(int(x) for x in d)


class File:
    """One HDF5 / NetCDF-4 file.  ``with File(path) as f: f.read("qBucket"); f.attrs("qBucket")``."""

    def __init__(self, path, mode="r"):
        h = lib()
        self.path = os.fspath(path)
        if mode == "r":
            self._id = h.H5Fopen(self.path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self._id = h.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self._id < 0:
            raise OSError(f"cannot open {self.path!r} as HDF5 ({'read' if mode == 'r' else 'write'})")
        self._dims = {}            # NetCDF-4 dimensions created in this file: name -> size

    def close(self):
        if getattr(self, "_id", -1) >= 0:
            lib().H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reading -----------------------------------------------------------------------------------
    def names(self):
        """Datasets of the root group (NetCDF variables and dimension scales)."""
        h = lib()
        n = hsize_t(0)
        h.H5Gget_num_objs(self._id, C.byref(n))
        out = []
        for i in range(n.value):
            if h.H5Gget_objtype_by_idx(self._id, i) != 1:      # H5G_DATASET
                continue
            ln = h.H5Gget_objname_by_idx(self._id, i, None, 0)
            buf = C.create_string_buffer(ln + 1)
            h.H5Gget_objname_by_idx(self._id, i, buf, ln + 1)
            out.append(buf.value.decode())
        return out

    def __contains__(self, name):
        return lib().H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def _open(self, name):
        d = lib().H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(f"{name!r} not in {self.path}")
        return d

    def shape(self, name):
        h = lib()
        d = self._open(name)
        s = h.H5Dget_space(d)
        try:
            return _dims(s)
        finally:
            h.H5Sclose(s)
            h.H5Dclose(d)

    def read(self, name):
        """The raw (packed) values of a numeric dataset as a numpy array of its stored type."""
        h = lib()
        d = self._open(name)
        s, t = h.H5Dget_space(d), h.H5Dget_type(d)
        try:
            dt = _np_dtype_of(t)
            if dt.kind == "S":
                # fixed-length strings (netCDF char arrays such as RouteLink's gages [link][15] or a TimeSlice's
                # stationId: HDF5 strings of size 1 per element) come back as bytes, read with their own type
                if h.H5Tis_variable_str(t) > 0:
                    raise TypeError(f"{name!r} is a variable-length string dataset")
                out = np.zeros(_dims(s), dtype=dt)
                if out.size and h.H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                    raise OSError(f"H5Dread({name!r}) failed")
                return out
            out = np.empty(_dims(s), dtype=dt)
            if out.size and h.H5Dread(d, _native(_NP_TO_H5[dt]), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                                      out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"H5Dread({name!r}) failed")
            return out
        finally:
            h.H5Tclose(t)
            h.H5Sclose(s)
            h.H5Dclose(d)

    def has_attr(self, name, attr):
        """Does dataset `name` (the file when None) carry attribute `attr`?  (Also for attribute types ``attr`` cannot
        return, such as the object-reference lists of dimension scales.)"""
        h = lib()
        obj = self._id if name is None else self._open(name)
        try:
            return h.H5Aexists(obj, attr.encode()) > 0
        finally:
            if name is not None:
                h.H5Dclose(obj)

    def attr(self, name, attr, default=None):
        """One attribute of dataset `name` (or of the file when name is None): numpy array / bytes."""
        h = lib()
        obj = self._id if name is None else self._open(name)
        try:
            if h.H5Aexists(obj, attr.encode()) <= 0:
                return default
            a = h.H5Aopen(obj, attr.encode(), H5P_DEFAULT)
            t, s = h.H5Aget_type(a), h.H5Aget_space(a)
            try:
                try:
                    dt = _np_dtype_of(t)
                except TypeError:
                    return default
                shape = _dims(s)
                if dt.kind == "S":
                    mt = h.H5Tcopy(t)
                    buf = C.create_string_buffer(dt.itemsize * max(1, int(np.prod(shape)) if shape else 1) + 1)
                    ok = h.H5Aread(a, mt, buf)
                    h.H5Tclose(mt)
                    return buf.value if ok >= 0 else default
                out = np.empty(shape if shape else (1,), dtype=dt)
                if h.H5Aread(a, _native(_NP_TO_H5[dt]), out.ctypes.data_as(C.c_void_p)) < 0:
                    return default
                return out
            finally:
                h.H5Sclose(s)
                h.H5Tclose(t)
                h.H5Aclose(a)
        finally:
            if name is not None:
                h.H5Dclose(obj)

    def packing(self, name):
        """CF packing facts of a variable: dict(scale, offset, fills, vmin, vmax) -- scale/offset as the
        stored numpy scalars (their TYPE decides the unpacked precision), None when absent."""
        sc, off = self.attr(name, "scale_factor"), self.attr(name, "add_offset")
        fills = [v[0] for v in (self.attr(name, "_FillValue"), self.attr(name, "missing_value")) if v is not None and v.size]
        vr = self.attr(name, "valid_range")
        vmin = self.attr(name, "valid_min")
        vmax = self.attr(name, "valid_max")
        lo = vr[0] if vr is not None and vr.size == 2 else (vmin[0] if vmin is not None and vmin.size else None)
        hi = vr[1] if vr is not None and vr.size == 2 else (vmax[0] if vmax is not None and vmax.size else None)
        return {"scale": None if sc is None else sc[0], "offset": None if off is None else off[0],
                "fills": fills, "vmin": lo, "vmax": hi}

    # ---- writing -------------------------------------------------------------------------------------
    def dimension(self, name, size):
        """A NetCDF-4 dimension that is not a variable: an HDF5 dimension scale of `size` elements whose NAME says so
        (the convention the netCDF-4 library writes and expects, e.g. /string15 in the reference's lastobs files)."""
        h = lib()
        dims = (hsize_t * 1)(size)
        sp = h.H5Screate_simple(1, dims, None)
        t = _native("H5T_NATIVE_FLOAT_g")
        d = h.H5Dcreate2(self._id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        h.H5Sclose(sp)
        if d < 0:
            raise OSError(f"cannot create dimension {name!r}")
        try:
            if hl().H5DSset_scale(d, ("This is a netCDF dimension but not a netCDF variable.%10d" % size).encode()) < 0:
                raise OSError(f"H5DSset_scale({name!r}) failed")
            _write_attr(d, "_Netcdf4Dimid", np.int32(len(self._dims)))
        finally:
            h.H5Dclose(d)
        self._dims[name] = size

    def write(self, name, array, attrs=None, dims=None):
        """Create dataset `name` from a numeric numpy array; attrs: {name: number | numpy array | str}.
        dims: names of the NetCDF-4 dimensions of its axes.  A dimension of the dataset's own name makes it a coordinate
        variable (the dataset becomes the dimension scale); other names must exist (``dimension`` or an earlier
        coordinate variable) and are attached as scales (DIMENSION_LIST / REFERENCE_LIST), so that netCDF-4 readers see
        named, shared dimensions instead of anonymous ones."""
        h = lib()
        a = np.ascontiguousarray(array)
        if a.dtype not in _NP_TO_H5:
            raise TypeError(f"unsupported dtype {a.dtype}")
        extent = (hsize_t * max(1, a.ndim))(*(a.shape if a.ndim else (1,)))
        s = h.H5Screate_simple(max(1, a.ndim), extent, None)
        t = _native(_NP_TO_H5[a.dtype])
        d = h.H5Dcreate2(self._id, name.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            h.H5Sclose(s)
            raise OSError(f"cannot create dataset {name!r}")
        try:
            if a.size and h.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"H5Dwrite({name!r}) failed")
            for k, v in (attrs or {}).items():
                _write_attr(d, k, v)
            if dims is not None:
                if len(dims) != a.ndim:
                    raise ValueError(f"{name!r}: {a.ndim} axes but {len(dims)} dimension names")
                if len(dims) == 1 and dims[0] == name:          # coordinate variable: it IS the dimension
                    if hl().H5DSset_scale(d, name.encode()) < 0:
                        raise OSError(f"H5DSset_scale({name!r}) failed")
                    _write_attr(d, "_Netcdf4Dimid", np.int32(len(self._dims)))
                    self._dims[name] = a.shape[0]
                else:
                    for ax, dn in enumerate(dims):
                        if dn not in self._dims or self._dims[dn] != a.shape[ax]:
                            raise ValueError(f"{name!r}: axis {ax} does not match dimension {dn!r}")
                        sc = h.H5Dopen2(self._id, dn.encode(), H5P_DEFAULT)
                        rc = hl().H5DSattach_scale(d, sc, ax)
                        h.H5Dclose(sc)
                        if rc < 0:
                            raise OSError(f"H5DSattach_scale({name!r}, {dn!r}) failed")
        finally:
            h.H5Sclose(s)
            h.H5Dclose(d)

    def write_chars(self, name, rows, width, attrs=None, dims=None):
        """A NetCDF char array [len(rows)][width] (HDF5 strings of size 1, null-terminated padding: what the netCDF-4
        library writes for NC_CHAR, e.g. RouteLink's gages) from byte strings of exactly `width` bytes."""
        h = lib()
        raw = np.frombuffer(b"".join(rows), dtype="S1").reshape(len(rows), width) if rows else np.zeros((0, width), "S1")
        t = h.H5Tcopy(_native("H5T_C_S1_g"))
        h.H5Tset_size(t, 1)
        shape = (hsize_t * 2)(len(rows), width)
        sp = h.H5Screate_simple(2, shape, None)
        d = h.H5Dcreate2(self._id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            h.H5Sclose(sp)
            h.H5Tclose(t)
            raise OSError(f"cannot create dataset {name!r}")
        try:
            buf = np.ascontiguousarray(raw)
            if buf.size and h.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"H5Dwrite({name!r}) failed")
            for k, v in (attrs or {}).items():
                _write_attr(d, k, v)
            for ax, dn in enumerate(dims or []):
                if dn not in self._dims or self._dims[dn] != raw.shape[ax]:
                    raise ValueError(f"{name!r}: axis {ax} does not match dimension {dn!r}")
                sc = h.H5Dopen2(self._id, dn.encode(), H5P_DEFAULT)
                rc = hl().H5DSattach_scale(d, sc, ax)
                h.H5Dclose(sc)
                if rc < 0:
                    raise OSError(f"H5DSattach_scale({name!r}, {dn!r}) failed")
        finally:
            h.H5Sclose(sp)
            h.H5Tclose(t)
            h.H5Dclose(d)

    def set_attr(self, key, value):
        _write_attr(self._id, key, value)


def _write_attr(obj, key, value):
    h = lib()
    if isinstance(value, (str, bytes)):
        raw = value.encode() if isinstance(value, str) else value
        t = h.H5Tcopy(_native("H5T_C_S1_g"))
        h.H5Tset_size(t, max(1, len(raw)))
        s = h.H5Screate(0)                                   # H5S_SCALAR
        a = h.H5Acreate2(obj, key.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT)
        buf = C.create_string_buffer(raw, max(1, len(raw)))
        h.H5Awrite(a, t, buf)
        h.H5Aclose(a)
        h.H5Sclose(s)
        h.H5Tclose(t)
        return
    v = np.atleast_1d(np.asarray(value))
    if v.dtype not in _NP_TO_H5:
        v = v.astype(np.float64)
    dims = (hsize_t * 1)(v.shape[0])
    s = h.H5Screate_simple(1, dims, None)
    t = _native(_NP_TO_H5[v.dtype])
    a = h.H5Acreate2(obj, key.encode(), t, s, H5P_DEFAULT, H5P_DEFAULT)
    h.H5Awrite(a, t, np.ascontiguousarray(v).ctypes.data_as(C.c_void_p))
    h.H5Aclose(a)
    h.H5Sclose(s)


def unpack(raw, packing, fill_value=None):
    """netCDF4-python's default read of a packed variable, in numpy: values equal to _FillValue /
    missing_value or outside valid_range/valid_min/valid_max are masked; the rest become
    ``raw * scale_factor + add_offset`` with the attributes as numpy scalars of their stored type (an int32
    array times a float32 scalar is float64).  Masked entries are replaced by `fill_value` when given
    (``.filled(fill_value)``), else returned through the mask.  Returns (values, mask)."""
    raw = np.asarray(raw)
    mask = np.zeros(raw.shape, dtype=bool)
    for f in packing["fills"]:
        mask |= raw == f
    if packing["vmin"] is not None:
        mask |= raw < packing["vmin"]
    if packing["vmax"] is not None:
        mask |= raw > packing["vmax"]
    vals = raw
    if packing["scale"] is not None or packing["offset"] is not None:
        sc = packing["scale"] if packing["scale"] is not None else np.float64(1.0)
        off = packing["offset"] if packing["offset"] is not None else np.float64(0.0)
        vals = raw * sc + off
    if fill_value is not None:
        vals = np.where(mask, np.asarray(fill_value, dtype=vals.dtype if vals.dtype.kind == "f" else np.float64), vals)
    return vals, mask
