"""The handful of h5py calls qpnet_amd.loaders makes, on the HDF5 C library itself (ctypes over libhdf5).

The reference reads and writes its acoustic features and scaler statistics as HDF5 datasets through h5py
(src/utils/utils.py:43-128: `/world`, `/world/mean`, `/world/scale`).  h5py is a wrapper of libhdf5; where h5py is not
installed but the library is (this image: /opt/conda/lib/libhdf5.so, 1.10.6), this module offers the same subset --
File(name, mode) as a context manager, `path in f`, `f[path][()]`, `f[path].shape`, `del f[path]`,
`f.create_dataset(path, data=array)` with intermediate groups -- and the files are real HDF5: `h5dump` / `h5ls` / h5py read
what it writes, and it reads what `h5import` or h5py wrote (tests/test_formats_cpu.py).  Numeric datasets only (the
reference stores float arrays); contiguous layout, native byte order, as h5py's create_dataset(data=...) does.

QPN_LIBHDF5=<path> names the library; otherwise the loader's search path and a few usual places are tried.  libhdf5 >= 1.10 (64-bit hid_t) is
required and checked.  The C library is not thread-safe unless built so: a File object and everything read through it stay on the thread that
opened it (the loaders open, read and close inside one call of the prefetch thread)."""
import ctypes as C
import ctypes.util
import os

import numpy as np

_hid = C.c_int64
_LIB = None
_SEARCH = ("libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
           "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so", "/usr/local/lib/libhdf5.so")

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    names = []
    if os.environ.get("QPN_LIBHDF5"):
        names.append(os.environ["QPN_LIBHDF5"])
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        names.append(found)
    names += list(_SEARCH)
    err = None
    for n in names:
        try:
            lib = C.CDLL(n)
            break
        except OSError as e:
            err = e
    else:
        raise ImportError("no HDF5 library found (tried %s): %s" % (", ".join(names), err))
    # hid_t (and the H5T_NATIVE_*_g / H5P_* id globals read below) are 64-bit integers since HDF5 1.10; in 1.8.x they are 32-bit ints, and
    # reading 8 bytes of a 4-byte global or passing 64-bit ids would give garbage ids -- failed or silently wrong reads.  Refuse older libraries.
    maj, mnr, rel = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    lib.H5get_libversion.restype = C.c_int
    lib.H5get_libversion.argtypes = [C.POINTER(C.c_uint)] * 3
    if lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel)) < 0 or (maj.value, mnr.value) < (1, 10):
        raise ImportError("HDF5 %d.%d.%d at %s is older than 1.10 (32-bit hid_t): qpnet_amd._hdf5 needs libhdf5 >= 1.10 or h5py"
                          % (maj.value, mnr.value, rel.value, n))
    sig = {
        "H5open": (C.c_int, []), "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]), "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]), "H5Fflush": (C.c_int, [_hid, C.c_int]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]), "H5Ldelete": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Oopen": (_hid, [_hid, C.c_char_p, _hid]), "H5Oclose": (C.c_int, [_hid]), "H5Iget_type": (C.c_int, [_hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Dclose": (C.c_int, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]), "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Screate": (_hid, [C.c_int]), "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), "H5Sclose": (C.c_int, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]), "H5Tget_sign": (C.c_int, [_hid]), "H5Tclose": (C.c_int, [_hid]),
        "H5Pcreate": (_hid, [_hid]), "H5Pset_create_intermediate_group": (C.c_int, [_hid, C.c_uint]), "H5Pclose": (C.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise ImportError("H5open failed")
    lib.H5Eset_auto2(0, None, None)                 # errors are reported through return codes here, not printed by the library
    _LIB = lib
    return lib


def _native(lib, dtype):
    names = {"float64": "H5T_NATIVE_DOUBLE_g", "float32": "H5T_NATIVE_FLOAT_g", "int64": "H5T_NATIVE_INT64_g", "int32": "H5T_NATIVE_INT32_g",
             "int16": "H5T_NATIVE_INT16_g", "int8": "H5T_NATIVE_INT8_g", "uint64": "H5T_NATIVE_UINT64_g", "uint32": "H5T_NATIVE_UINT32_g",
             "uint16": "H5T_NATIVE_UINT16_g", "uint8": "H5T_NATIVE_UINT8_g"}
    key = np.dtype(dtype).name
    if key not in names:
        raise TypeError("HDF5 datasets of dtype %s are not supported here (numeric arrays only)" % key)
    return _hid.in_dll(lib, names[key]).value


class _Dataset:
    def __init__(self, f, path):
        lib = f._lib
        d = lib.H5Dopen2(f._id, path.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError("Unable to open object (%s is not a dataset)" % path)
        try:
            sp = lib.H5Dget_space(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            lib.H5Sclose(sp)
            self.shape = tuple(int(dims[i]) for i in range(nd))
            t = lib.H5Dget_type(d)
            cls, size, sign = lib.H5Tget_class(t), int(lib.H5Tget_size(t)), lib.H5Tget_sign(t)
            lib.H5Tclose(t)
            if cls == H5T_FLOAT and size in (4, 8):
                self.dtype = np.dtype("float%d" % (8 * size))
            elif cls == H5T_INTEGER and size in (1, 2, 4, 8):
                self.dtype = np.dtype("%sint%d" % ("" if sign else "u", 8 * size))
            else:
                raise TypeError("dataset %s: HDF5 type class %d of %d bytes is not supported here" % (path, cls, size))
        finally:
            lib.H5Dclose(d)
        self._f, self._path = f, path

    def __getitem__(self, key):
        lib = self._f._lib
        out = np.empty(self.shape, dtype=self.dtype)
        d = lib.H5Dopen2(self._f._id, self._path.encode(), H5P_DEFAULT)
        try:
            if out.size and lib.H5Dread(d, _native(lib, self.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("Can't read data (%s)" % self._path)
        finally:
            lib.H5Dclose(d)
        if isinstance(key, tuple) and key == ():
            return out if out.ndim else out[()]
        return out[key]


class File:
    """h5py.File's modes 'r', 'r+', 'a', 'w' and the calls listed in the module docstring."""

    def __init__(self, name, mode="r"):
        lib = _lib()
        self._lib, self.name, self.mode = lib, name, mode
        b = os.fsencode(name)
        if mode == "r":
            fid = lib.H5Fopen(b, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "r+":
            fid = lib.H5Fopen(b, H5F_ACC_RDWR, H5P_DEFAULT)
        elif mode == "a":
            fid = lib.H5Fopen(b, H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(name) else lib.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "w":
            fid = lib.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("Invalid mode; must be one of r, r+, w, a")
        if fid < 0:
            raise OSError("Unable to open file (%s, mode %s)" % (name, mode))
        self._id = fid

    @staticmethod
    def _parts(path):
        return [s for s in path.split("/") if s]

    def __contains__(self, path):
        # H5Lexists needs every intermediate link to exist: walk the path
        if self._id is None:
            return False
        cur = ""
        for s in self._parts(path):
            cur += "/" + s
            if self._lib.H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def __getitem__(self, path):
        if path not in self:
            raise KeyError("Unable to open object (object '%s' doesn't exist)" % path)
        return _Dataset(self, "/" + "/".join(self._parts(path)))

    def __delitem__(self, path):
        if path not in self or self._lib.H5Ldelete(self._id, ("/" + "/".join(self._parts(path))).encode(), H5P_DEFAULT) < 0:
            raise KeyError("Couldn't delete link (%s)" % path)

    def create_dataset(self, path, data=None):
        lib = self._lib
        a = np.asarray(data)
        if a.ndim and not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)
        if a.dtype == np.bool_ or a.dtype.kind not in "fiu":
            raise TypeError("HDF5 datasets of dtype %s are not supported here (numeric arrays only)" % a.dtype)
        mem = _native(lib, a.dtype)
        name = ("/" + "/".join(self._parts(path))).encode()
        if path in self:
            raise ValueError("Unable to create dataset (name already exists)")
        if a.ndim == 0:
            sp = lib.H5Screate(0)                        # H5S_SCALAR
        else:
            dims = (C.c_uint64 * a.ndim)(*a.shape)
            sp = lib.H5Screate_simple(a.ndim, dims, None)
        lcpl = lib.H5Pcreate(_hid.in_dll(lib, "H5P_CLS_LINK_CREATE_ID_g").value)
        lib.H5Pset_create_intermediate_group(lcpl, 1)    # h5py creates the groups of "/world/mean" on the way as well
        d = lib.H5Dcreate2(self._id, name, mem, sp, lcpl, H5P_DEFAULT, H5P_DEFAULT)
        try:
            if d < 0:
                raise OSError("Unable to create dataset (%s)" % path)
            if a.size and lib.H5Dwrite(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("Can't write data (%s)" % path)
        finally:
            if d >= 0:
                lib.H5Dclose(d)
            lib.H5Pclose(lcpl)
            lib.H5Sclose(sp)
        return _Dataset(self, name.decode())

    def flush(self):
        if self._id is not None:
            self._lib.H5Fflush(self._id, 1)              # H5F_SCOPE_GLOBAL

    def close(self):
        if self._id is not None:
            self._lib.H5Fclose(self._id)
            self._id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
