"""Thin ctypes binding to the HDF5 C library (libhdf5 >= 1.10), the writer behind recorder.py when h5py is not importable.

The reference's RecordHDF5Wrapper goes through h5py (gym_lowcostrobot/envs/wrappers/record_hdf5.py:52-61: one
`file.create_dataset(name, data=array)` per dataset, intermediate groups created implicitly).  h5py is only a wrapper of this C
library, so calling the same library directly produces the same files: contiguous datasets of the native little-endian types
under the groups `observations/` and `observations/images/`.

Only what the recorder needs is bound: create / open a file, create the intermediate groups, write a whole array as one dataset,
read a whole dataset back.  `available()` is False when no libhdf5 >= 1.10 can be loaded (search order: $LCR_LIBHDF5, the dynamic
linker's `hdf5` / `hdf5_serial`, the usual distribution and conda locations).
"""
import ctypes
import ctypes.util
import glob
import os
import sys

import numpy as np

_hid = ctypes.c_int64       # hid_t is 64 bits since HDF5 1.10
_hsize = ctypes.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
_NATIVE = {"uint8": "H5T_NATIVE_UINT8_g", "int8": "H5T_NATIVE_INT8_g", "uint16": "H5T_NATIVE_UINT16_g", "int16": "H5T_NATIVE_INT16_g",
           "uint32": "H5T_NATIVE_UINT32_g", "int32": "H5T_NATIVE_INT32_g", "uint64": "H5T_NATIVE_UINT64_g", "int64": "H5T_NATIVE_INT64_g",
           "float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g"}
_lib = None
_probed = False


def _candidates():
    if os.environ.get("LCR_LIBHDF5"):
        yield os.environ["LCR_LIBHDF5"]
    for n in ("hdf5", "hdf5_serial"):
        p = ctypes.util.find_library(n)
        if p:
            yield p
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*",
                os.path.join(sys.prefix, "lib", "libhdf5.so*"), "/opt/conda/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def _bind(lib):
    c, P = ctypes, ctypes.POINTER
    sig = {
        "H5open": (c.c_int, []),
        "H5Fcreate": (_hid, [c.c_char_p, c.c_uint, _hid, _hid]), "H5Fopen": (_hid, [c.c_char_p, c.c_uint, _hid]), "H5Fclose": (c.c_int, [_hid]),
        "H5Gcreate2": (_hid, [_hid, c.c_char_p, _hid, _hid, _hid]), "H5Gclose": (c.c_int, [_hid]),
        "H5Lexists": (c.c_int, [_hid, c.c_char_p, _hid]),
        "H5Screate_simple": (_hid, [c.c_int, P(_hsize), P(_hsize)]), "H5Sclose": (c.c_int, [_hid]),
        "H5Sget_simple_extent_ndims": (c.c_int, [_hid]), "H5Sget_simple_extent_dims": (c.c_int, [_hid, P(_hsize), P(_hsize)]),
        "H5Dcreate2": (_hid, [_hid, c.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dopen2": (_hid, [_hid, c.c_char_p, _hid]),
        "H5Dwrite": (c.c_int, [_hid, _hid, _hid, _hid, _hid, c.c_void_p]), "H5Dread": (c.c_int, [_hid, _hid, _hid, _hid, _hid, c.c_void_p]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]), "H5Dclose": (c.c_int, [_hid]),
        "H5Tget_class": (c.c_int, [_hid]), "H5Tget_size": (c.c_size_t, [_hid]), "H5Tget_sign": (c.c_int, [_hid]), "H5Tclose": (c.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args


def _load():
    global _lib, _probed
    if _probed:
        return _lib
    _probed = True
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
            maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
            if lib.H5open() < 0 or lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel)) < 0:
                continue
            if (maj.value, mnr.value) < (1, 10):      # 32-bit hid_t before 1.10: not bound
                continue
            _bind(lib)
            lib._lcr_version = f"{maj.value}.{mnr.value}.{rel.value}"
            lib._lcr_path = path
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


def available():
    return _load() is not None


def version():
    lib = _load()
    return None if lib is None else f"libhdf5 {lib._lcr_version} ({lib._lcr_path})"


def _ok(v, what):
    if v < 0:
        raise OSError(f"libhdf5: {what} failed")
    return v


def _native(lib, dtype):
    name = _NATIVE.get(np.dtype(dtype).name)
    if name is None:
        raise TypeError(f"no HDF5 native type bound for numpy dtype {dtype}")
    return _hid.in_dll(lib, name).value


def write_file(path, datasets):
    """datasets: {"group/sub/name": ndarray}; every array becomes one contiguous dataset, groups are created on the way."""
    lib = _load()
    if lib is None:
        raise OSError("no libhdf5 >= 1.10 found")
    f = _ok(lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"H5Fcreate({path})")
    try:
        for name, arr in datasets.items():
            arr = np.asarray(arr)
            if arr.dtype == np.bool_:
                arr = arr.astype(np.uint8)
            arr = np.ascontiguousarray(arr)
            parts = name.strip("/").split("/")
            for i in range(1, len(parts)):
                g = "/".join(parts[:i]).encode()
                if _ok(lib.H5Lexists(f, g, H5P_DEFAULT), "H5Lexists") == 0:
                    _ok(lib.H5Gclose(_ok(lib.H5Gcreate2(f, g, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Gcreate2({g!r})")), "H5Gclose")
            dims = (_hsize * max(arr.ndim, 1))(*arr.shape)
            space = _ok(lib.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
            t = _native(lib, arr.dtype)
            d = _ok(lib.H5Dcreate2(f, "/".join(parts).encode(), t, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({name})")
            try:
                if arr.size:
                    _ok(lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)), f"H5Dwrite({name})")
            finally:
                lib.H5Dclose(d)
                lib.H5Sclose(space)
    finally:
        _ok(lib.H5Fclose(f), "H5Fclose")
    return path


def read_file(path, names):
    """{name: ndarray} for those of `names` that exist in the file (whole datasets, integer and IEEE float types)."""
    lib = _load()
    if lib is None:
        raise OSError("no libhdf5 >= 1.10 found")
    f = _ok(lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({path})")
    out = {}
    try:
        for name in names:
            parts = name.strip("/").split("/")
            if any(lib.H5Lexists(f, "/".join(parts[:i]).encode(), H5P_DEFAULT) <= 0 for i in range(1, len(parts) + 1)):
                continue
            d = _ok(lib.H5Dopen2(f, "/".join(parts).encode(), H5P_DEFAULT), f"H5Dopen2({name})")
            space, t = lib.H5Dget_space(d), lib.H5Dget_type(d)
            try:
                nd = _ok(lib.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims")
                dims = (_hsize * max(nd, 1))()
                _ok(lib.H5Sget_simple_extent_dims(space, dims, None), "H5Sget_simple_extent_dims")
                cls, size, sign = lib.H5Tget_class(t), lib.H5Tget_size(t), lib.H5Tget_sign(t)
                if cls == 0:
                    dt = np.dtype(f"{'i' if sign == 1 else 'u'}{size}")
                elif cls == 1:
                    dt = np.dtype(f"f{size}")
                else:
                    raise TypeError(f"{name}: HDF5 type class {cls} not bound")
                arr = np.empty(tuple(dims[:nd]), dt)
                if arr.size:
                    _ok(lib.H5Dread(d, _native(lib, dt), H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)), f"H5Dread({name})")
                out[name] = arr
            finally:
                lib.H5Tclose(t)
                lib.H5Sclose(space)
                lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)
    return out
