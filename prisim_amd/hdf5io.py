"""Minimal HDF5 writer / reader over the HDF5 C library through ctypes (no h5py in this image).

Only what PRISim's on-disk schema needs (InterferometerArray.save, interferometry.py:8717-8830), written the way h5py
writes it so that PRISim's own readers (``h5py.File(init_file)``) see the same objects:
  * python str            -> scalar dataset, variable-length UTF-8 string
  * python float / int    -> scalar dataset, float64 / int64
  * numpy float / int     -> simple dataspace of the array's shape, native type
  * numpy complex128/64   -> compound {'r', 'i'} of float64 / float32 (h5py's complex convention)
  * numpy 'S' / 'U' array -> fixed-length strings
  * structured array      -> compound with the same field names (string and numeric members)
  * attrs                 -> string or numeric scalar attributes
The library is looked up in PRISIM_HDF5_LIB, the loader path, and the usual conda / system locations; a missing library
raises HDF5Unavailable (output formats are not on the compute path).
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as NP

hid_t = C.c_int64
hsize_t = C.c_uint64
H5P_DEFAULT = 0
H5S_ALL = 0
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 2, 4
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND = 0, 1, 3, 6
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5T_STR_NULLPAD = 1


class HDF5Unavailable(RuntimeError):
    """libhdf5 could not be loaded."""


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = []
    if os.environ.get('PRISIM_HDF5_LIB'):
        cands.append(os.environ['PRISIM_HDF5_LIB'])
    found = ctypes.util.find_library('hdf5')
    if found:
        cands.append(found)
    for pat in ('/opt/conda/lib/libhdf5.so*', '/usr/lib/x86_64-linux-gnu/libhdf5*.so*', '/usr/lib64/libhdf5.so*', '/usr/local/lib/libhdf5.so*'):
        cands += sorted(glob.glob(pat))
    err = None
    for path in cands:
        try:
            lib = C.CDLL(path)
            lib.H5open()
            _declare(lib)
            _lib = lib
            return lib
        except (OSError, AttributeError) as exc:
            err = exc
    raise HDF5Unavailable('the HDF5 C library was not found (set PRISIM_HDF5_LIB): {0!r}'.format(err))


def _declare(lib):
    p, i, sz, cs = C.c_void_p, C.c_int, C.c_size_t, C.c_char_p
    sig = {
        'H5Fcreate': (hid_t, [cs, C.c_uint, hid_t, hid_t]), 'H5Fopen': (hid_t, [cs, C.c_uint, hid_t]), 'H5Fclose': (i, [hid_t]),
        'H5Gcreate2': (hid_t, [hid_t, cs, hid_t, hid_t, hid_t]), 'H5Gclose': (i, [hid_t]),
        'H5Screate_simple': (hid_t, [i, C.POINTER(hsize_t), C.POINTER(hsize_t)]), 'H5Screate': (hid_t, [i]), 'H5Sclose': (i, [hid_t]),
        'H5Sget_simple_extent_ndims': (i, [hid_t]), 'H5Sget_simple_extent_dims': (i, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        'H5Dcreate2': (hid_t, [hid_t, cs, hid_t, hid_t, hid_t, hid_t, hid_t]), 'H5Dopen2': (hid_t, [hid_t, cs, hid_t]),
        'H5Dwrite': (i, [hid_t, hid_t, hid_t, hid_t, hid_t, p]), 'H5Dread': (i, [hid_t, hid_t, hid_t, hid_t, hid_t, p]),
        'H5Dget_space': (hid_t, [hid_t]), 'H5Dget_type': (hid_t, [hid_t]), 'H5Dclose': (i, [hid_t]),
        'H5Dvlen_reclaim': (i, [hid_t, hid_t, hid_t, p]),
        'H5Acreate2': (hid_t, [hid_t, cs, hid_t, hid_t, hid_t, hid_t]), 'H5Awrite': (i, [hid_t, hid_t, p]), 'H5Aclose': (i, [hid_t]),
        'H5Aopen': (hid_t, [hid_t, cs, hid_t]), 'H5Aread': (i, [hid_t, hid_t, p]), 'H5Aget_type': (hid_t, [hid_t]),
        'H5Tcopy': (hid_t, [hid_t]), 'H5Tset_size': (i, [hid_t, sz]), 'H5Tset_cset': (i, [hid_t, i]), 'H5Tset_strpad': (i, [hid_t, i]),
        'H5Tcreate': (hid_t, [i, sz]), 'H5Tinsert': (i, [hid_t, cs, sz, hid_t]), 'H5Tclose': (i, [hid_t]),
        'H5Tget_class': (i, [hid_t]), 'H5Tget_size': (sz, [hid_t]), 'H5Tis_variable_str': (i, [hid_t]),
        'H5Tget_nmembers': (i, [hid_t]), 'H5Tget_member_name': (p, [hid_t, C.c_uint]), 'H5Tget_member_type': (hid_t, [hid_t, C.c_uint]),
        'H5Tget_member_offset': (sz, [hid_t, C.c_uint]), 'H5Tget_sign': (i, [hid_t]), 'H5free_memory': (i, [p]),
        'H5Lexists': (i, [hid_t, cs, hid_t]), 'H5Eset_auto2': (i, [hid_t, p, p]),
        'H5Gopen2': (hid_t, [hid_t, cs, hid_t]), 'H5Gget_info': (i, [hid_t, p]),
        'H5Lget_name_by_idx': (C.c_ssize_t, [hid_t, cs, i, i, hsize_t, p, sz, hid_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)        # errors are reported through return codes, not printed


def _native(lib, name):
    return hid_t.in_dll(lib, name).value


_NUMERIC = {'f8': 'H5T_NATIVE_DOUBLE_g', 'f4': 'H5T_NATIVE_FLOAT_g', 'i8': 'H5T_NATIVE_INT64_g', 'i4': 'H5T_NATIVE_INT32_g',
            'i2': 'H5T_NATIVE_INT16_g', 'i1': 'H5T_NATIVE_INT8_g', 'u8': 'H5T_NATIVE_UINT64_g', 'u4': 'H5T_NATIVE_UINT32_g',
            'u2': 'H5T_NATIVE_UINT16_g', 'u1': 'H5T_NATIVE_UINT8_g', 'b1': 'H5T_NATIVE_INT8_g'}


def _check(rc, what):
    if rc < 0:
        raise IOError('HDF5 call failed: ' + what)
    return rc


class File(object):
    """h5py-flavoured writer: ``f.create_group('a/b')``, ``f.write('a/b/name', value, attrs={'units': 'Jy'})``."""

    def __init__(self, filename, mode='w'):
        self._lib = _load()
        self._owned_types = []
        flags = {'w': H5F_ACC_TRUNC, 'w-': H5F_ACC_EXCL}
        if mode in flags:
            self._fid = self._lib.H5Fcreate(filename.encode(), flags[mode], H5P_DEFAULT, H5P_DEFAULT)
        elif mode == 'r':
            self._fid = self._lib.H5Fopen(filename.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'w', 'w-' or 'r'")
        if self._fid < 0:
            raise IOError('cannot open {0!r} in mode {1!r}'.format(filename, mode))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._fid is not None and self._fid >= 0:
            self._lib.H5Fclose(self._fid)
        self._fid = None

    # ---- writing ----------------------------------------------------------------------------
    def create_group(self, path):
        lib = self._lib
        parts = [q for q in path.split('/') if q]
        for k in range(1, len(parts) + 1):
            sub = '/'.join(parts[:k]).encode()
            if lib.H5Lexists(self._fid, sub, H5P_DEFAULT) > 0:
                continue
            gid = _check(lib.H5Gcreate2(self._fid, sub, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2 ' + path)
            lib.H5Gclose(gid)

    def _string_type(self, size):
        lib = self._lib
        tid = lib.H5Tcopy(_native(lib, 'H5T_C_S1_g'))
        lib.H5Tset_size(tid, size)
        if size == H5T_VARIABLE:
            lib.H5Tset_cset(tid, H5T_CSET_UTF8)
        else:
            lib.H5Tset_strpad(tid, H5T_STR_NULLPAD)
        return tid

    def _type_for(self, dt):
        """(hdf5 type id, owned?) for a numpy dtype."""
        lib = self._lib
        if dt.kind == 'c':
            base = 'f8' if dt.itemsize == 16 else 'f4'
            tid = lib.H5Tcreate(H5T_COMPOUND, dt.itemsize)
            lib.H5Tinsert(tid, b'r', 0, _native(lib, _NUMERIC[base]))
            lib.H5Tinsert(tid, b'i', dt.itemsize // 2, _native(lib, _NUMERIC[base]))
            return tid, True
        if dt.kind == 'S':
            return self._string_type(max(dt.itemsize, 1)), True
        if dt.names:
            tid = lib.H5Tcreate(H5T_COMPOUND, dt.itemsize)
            for name in dt.names:
                sub, off = dt.fields[name][0], dt.fields[name][1]
                mt, owned = self._type_for(sub)
                lib.H5Tinsert(tid, name.encode(), off, mt)
                if owned:
                    lib.H5Tclose(mt)
            return tid, True
        key = dt.kind + str(dt.itemsize)
        if key not in _NUMERIC:
            raise TypeError('dtype {0} cannot be written'.format(dt))
        return _native(lib, _NUMERIC[key]), False

    def _prepare(self, value):
        """-> (array or bytes buffer, hdf5 type, owned, shape or None for scalar, keepalive)."""
        if isinstance(value, bytes):
            value = value.decode()
        if isinstance(value, str):
            buf = C.c_char_p(value.encode('utf-8'))
            return C.byref(buf), self._string_type(H5T_VARIABLE), True, None, buf
        arr = NP.asarray(value)
        if arr.dtype.kind == 'U':
            arr = NP.char.encode(arr, 'utf-8')
        if arr.dtype.kind == 'O':
            raise TypeError('object arrays cannot be written')
        if arr.dtype.kind == 'b':
            arr = arr.astype(NP.int8)
        shape = arr.shape if arr.ndim else None              # 0-d -> scalar dataspace (ascontiguousarray would make it 1-d)
        arr = NP.ascontiguousarray(arr)
        tid, owned = self._type_for(arr.dtype)
        return arr.ctypes.data_as(C.c_void_p), tid, owned, shape, arr

    def _space(self, shape):
        lib = self._lib
        if shape is None:
            return lib.H5Screate(H5S_SCALAR)
        dims = (hsize_t * len(shape))(*shape)
        return lib.H5Screate_simple(len(shape), dims, None)

    def write(self, path, value, attrs=None):
        lib = self._lib
        parent = '/'.join(path.split('/')[:-1])
        if parent:
            self.create_group(parent)
        buf, tid, owned, shape, keep = self._prepare(value)
        sid = self._space(shape)
        did = lib.H5Dcreate2(self._fid, path.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if did < 0:
            raise IOError('cannot create dataset ' + path)
        try:
            if shape is None or all(n > 0 for n in shape):
                _check(lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), 'H5Dwrite ' + path)
            for name, aval in (attrs or {}).items():
                abuf, atid, aowned, ashape, akeep = self._prepare(aval)
                asid = self._space(ashape)
                aid = _check(lib.H5Acreate2(did, name.encode(), atid, asid, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2 ' + name)
                _check(lib.H5Awrite(aid, atid, abuf), 'H5Awrite ' + name)
                lib.H5Aclose(aid)
                lib.H5Sclose(asid)
                if aowned:
                    lib.H5Tclose(atid)
        finally:
            lib.H5Dclose(did)
            lib.H5Sclose(sid)
            if owned:
                lib.H5Tclose(tid)

    # ---- reading (round-trip checks) -------------------------------------------------------------
    def _numpy_dtype(self, tid):
        lib = self._lib
        cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
        if cls == H5T_FLOAT:
            return NP.dtype('f%d' % size)
        if cls == H5T_INTEGER:
            return NP.dtype(('i%d' if lib.H5Tget_sign(tid) else 'u%d') % size)
        if cls == H5T_STRING:
            return NP.dtype('S%d' % size)
        if cls == H5T_COMPOUND:
            names, formats, offsets = [], [], []
            for k in range(lib.H5Tget_nmembers(tid)):
                raw = lib.H5Tget_member_name(tid, k)
                names.append(C.cast(raw, C.c_char_p).value.decode())
                lib.H5free_memory(raw)
                mt = lib.H5Tget_member_type(tid, k)
                formats.append(self._numpy_dtype(mt))
                lib.H5Tclose(mt)
                offsets.append(lib.H5Tget_member_offset(tid, k))
            dt = NP.dtype({'names': names, 'formats': formats, 'offsets': offsets, 'itemsize': size})
            if names == ['r', 'i'] and formats[0] == formats[1] and formats[0].kind == 'f':
                return NP.dtype('c%d' % size)
            return dt
        raise TypeError('HDF5 type class {0} is not supported by this reader'.format(cls))

    def read(self, path):
        lib = self._lib
        did = lib.H5Dopen2(self._fid, path.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError(path)
        tid, sid = lib.H5Dget_type(did), lib.H5Dget_space(did)
        try:
            nd = lib.H5Sget_simple_extent_ndims(sid)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sid, dims, None)
            shape = tuple(int(dims[k]) for k in range(nd))
            if lib.H5Tget_class(tid) == H5T_STRING and lib.H5Tis_variable_str(tid) > 0:
                n = int(NP.prod(shape)) if shape else 1
                ptrs = (C.c_char_p * n)()
                _check(lib.H5Dread(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs), 'H5Dread ' + path)
                out = [ptrs[k].decode('utf-8') if ptrs[k] is not None else '' for k in range(n)]
                lib.H5Dvlen_reclaim(tid, sid, H5P_DEFAULT, ptrs)
                return out[0] if not shape else NP.asarray(out, dtype=object).reshape(shape)
            dt = self._numpy_dtype(tid)
            out = NP.empty(shape, dtype=dt)
            if out.size:
                _check(lib.H5Dread(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), 'H5Dread ' + path)
            return out if shape else out[()]
        finally:
            lib.H5Tclose(tid)
            lib.H5Sclose(sid)
            lib.H5Dclose(did)

    def read_attr(self, path, name):
        lib = self._lib
        did = lib.H5Dopen2(self._fid, path.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError(path)
        aid = lib.H5Aopen(did, name.encode(), H5P_DEFAULT)
        if aid < 0:
            lib.H5Dclose(did)
            raise KeyError(name)
        tid = lib.H5Aget_type(aid)
        try:
            if lib.H5Tget_class(tid) == H5T_STRING and lib.H5Tis_variable_str(tid) > 0:
                ptr = C.c_char_p()
                _check(lib.H5Aread(aid, tid, C.byref(ptr)), 'H5Aread ' + name)
                val = ptr.value.decode('utf-8')
                lib.H5free_memory(ptr)
                return val
            out = NP.empty((), dtype=self._numpy_dtype(tid))
            _check(lib.H5Aread(aid, tid, out.ctypes.data_as(C.c_void_p)), 'H5Aread ' + name)
            return out[()]
        finally:
            lib.H5Tclose(tid)
            lib.H5Aclose(aid)
            lib.H5Dclose(did)

    def exists(self, path):
        return self._lib.H5Lexists(self._fid, path.encode(), H5P_DEFAULT) > 0

    def list(self, group):
        """Names of the links of a group, in name order (h5py: list(f[group]))."""
        lib = self._lib
        gid = lib.H5Gopen2(self._fid, group.encode(), H5P_DEFAULT)
        if gid < 0:
            raise KeyError(group)

        class _GInfo(C.Structure):                 # H5G_info_t
            _fields_ = [('storage_type', C.c_int), ('nlinks', hsize_t), ('max_corder', C.c_int64), ('mounted', C.c_int)]
        try:
            info = _GInfo()
            _check(lib.H5Gget_info(gid, C.byref(info)), 'H5Gget_info ' + group)
            names = []
            for k in range(int(info.nlinks)):
                n = lib.H5Lget_name_by_idx(gid, b'.', 0, 0, k, None, 0, H5P_DEFAULT)         # H5_INDEX_NAME, H5_ITER_INC
                buf = C.create_string_buffer(int(n) + 1)
                lib.H5Lget_name_by_idx(gid, b'.', 0, 0, k, buf, int(n) + 1, H5P_DEFAULT)
                names.append(buf.value.decode('utf-8'))
            return names
        finally:
            lib.H5Gclose(gid)
