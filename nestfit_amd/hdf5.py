"""Real HDF5 files for the result store without h5py: a ctypes binding of the
slice of the HDF5 C library (libhdf5 1.10 / 1.12 / 1.14: the API names they share; 1.8 with its 32-bit hid_t is refused) the store
needs -- groups, contiguous datasets, attributes, external links -- so that `<name>.store/table.hdf`
and `chunk<i>.hdf` are what the reference writes through h5py (nestfit/main.py:233-377,
docs/store_spec.rst:45-110): the table's `/pix/<i_lon>/<i_lat>` are external links into the chunk
files, strings are variable-length UTF-8, booleans the FALSE/TRUE enumeration h5py uses.

The library is looked up at run time (`NFA_LIBHDF5`, the loader's search path, then the usual
install locations); `available()` says whether it was found.  Nothing here touches the GPU.
"""
import ctypes as C
import ctypes.util
import os
from pathlib import Path

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

H5P_DEFAULT = 0
H5S_ALL = 0
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5S_SCALAR = 0
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_SGN_NONE = 0
H5L_TYPE_HARD, H5L_TYPE_SOFT, H5L_TYPE_EXTERNAL = 0, 1, 64
H5I_GROUP, H5I_DATASET = 2, 5
H5_INDEX_NAME, H5_ITER_INC = 0, 0

_CANDIDATES = (
    'libhdf5.so', 'libhdf5_serial.so',
    '/opt/conda/lib/libhdf5.so', '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so',
    '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so', '/usr/lib64/libhdf5.so', '/usr/local/lib/libhdf5.so',
)

_lib = None
_lib_err = None


class Hdf5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get('NFA_LIBHDF5')
    if env:
        yield env
    found = ctypes.util.find_library('hdf5') or ctypes.util.find_library('hdf5_serial')
    if found:
        yield found
    for c in _CANDIDATES:
        yield c
        # versioned names beside an unversioned one that is missing (runtime-only installs)
        p = Path(c)
        if p.is_absolute() and p.parent.is_dir():
            for q in sorted(p.parent.glob(p.name + '.*')):
                yield str(q)


_SIGS = {
    # name: (restype, argtypes)
    'H5open': (herr_t, []),
    'H5get_libversion': (herr_t, [C.POINTER(C.c_uint)] * 3),
    'H5Eset_auto2': (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    'H5Fcreate': (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
    'H5Fopen': (hid_t, [C.c_char_p, C.c_uint, hid_t]),
    'H5Fclose': (herr_t, [hid_t]),
    'H5Gcreate2': (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
    'H5Gclose': (herr_t, [hid_t]),
    'H5Oopen': (hid_t, [hid_t, C.c_char_p, hid_t]),
    'H5Oclose': (herr_t, [hid_t]),
    'H5Iget_type': (C.c_int, [hid_t]),
    'H5Screate': (hid_t, [C.c_int]),
    'H5Screate_simple': (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
    'H5Sget_simple_extent_ndims': (C.c_int, [hid_t]),
    'H5Sget_simple_extent_dims': (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
    'H5Sclose': (herr_t, [hid_t]),
    'H5Dcreate2': (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
    'H5Dwrite': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
    'H5Dread': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
    'H5Dget_space': (hid_t, [hid_t]),
    'H5Dget_type': (hid_t, [hid_t]),
    'H5Dvlen_reclaim': (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
    'H5Dclose': (herr_t, [hid_t]),
    'H5Acreate2': (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
    'H5Awrite': (herr_t, [hid_t, hid_t, C.c_void_p]),
    'H5Aread': (herr_t, [hid_t, hid_t, C.c_void_p]),
    'H5Aopen': (hid_t, [hid_t, C.c_char_p, hid_t]),
    'H5Aget_space': (hid_t, [hid_t]),
    'H5Aget_type': (hid_t, [hid_t]),
    'H5Aclose': (herr_t, [hid_t]),
    'H5Tcopy': (hid_t, [hid_t]),
    'H5Tset_size': (herr_t, [hid_t, C.c_size_t]),
    'H5Tset_cset': (herr_t, [hid_t, C.c_int]),
    'H5Tget_class': (C.c_int, [hid_t]),
    'H5Tget_size': (C.c_size_t, [hid_t]),
    'H5Tget_sign': (C.c_int, [hid_t]),
    'H5Tis_variable_str': (C.c_int, [hid_t]),
    'H5Tenum_create': (hid_t, [hid_t]),
    'H5Tenum_insert': (herr_t, [hid_t, C.c_char_p, C.c_void_p]),
    'H5Tclose': (herr_t, [hid_t]),
    'H5Lcreate_external': (herr_t, [C.c_char_p, C.c_char_p, hid_t, C.c_char_p, hid_t, hid_t]),
    'H5Lget_val': (herr_t, [hid_t, C.c_char_p, C.c_void_p, C.c_size_t, hid_t]),
    'H5Lunpack_elink_val': (herr_t, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint),
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
}
# iteration callbacks: (group or object, name, info record, user data); only the link type -- the
# first int of H5L_info_t in every library version -- is read from the records
_LINK_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.POINTER(C.c_int), C.c_void_p)
_ATTR_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)


def _bind(lib):
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # 1.12 renamed the iteration entry points of the 1.10 records to *1; both spell the same call here
    lib.nf_literate = getattr(lib, 'H5Literate1', None) or lib.H5Literate
    lib.nf_literate.restype = herr_t
    lib.nf_literate.argtypes = [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), _LINK_CB, C.c_void_p]
    lib.nf_aiterate = lib.H5Aiterate2
    lib.nf_aiterate.restype = herr_t
    lib.nf_aiterate.argtypes = [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), _ATTR_CB, C.c_void_p]
    if lib.H5open() < 0:
        raise OSError('H5open failed')
    # hid_t is a 64-bit integer from 1.10 on and a 32-bit int before: with the 64-bit binding used here a 1.8
    # library would hand back garbage for every handle and type constant -- refuse it instead
    major, minor, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(major), C.byref(minor), C.byref(rel))
    if (major.value, minor.value) < (1, 10):
        raise OSError(f'libhdf5 {major.value}.{minor.value}.{rel.value} is older than 1.10 (32-bit hid_t)')
    lib.H5Eset_auto2(0, None, None)                    # errors come back as return codes, not as stderr text
    g = lambda n: hid_t.in_dll(lib, n).value           # noqa: E731  (type handles are valid after H5open)
    lib.T = {
        np.dtype('f8'): g('H5T_NATIVE_DOUBLE_g'), np.dtype('f4'): g('H5T_NATIVE_FLOAT_g'),
        np.dtype('i1'): g('H5T_NATIVE_INT8_g'), np.dtype('i2'): g('H5T_NATIVE_INT16_g'),
        np.dtype('i4'): g('H5T_NATIVE_INT32_g'), np.dtype('i8'): g('H5T_NATIVE_INT64_g'),
        np.dtype('u1'): g('H5T_NATIVE_UINT8_g'), np.dtype('u2'): g('H5T_NATIVE_UINT16_g'),
        np.dtype('u4'): g('H5T_NATIVE_UINT32_g'), np.dtype('u8'): g('H5T_NATIVE_UINT64_g'),
    }
    lib.T_C_S1 = g('H5T_C_S1_g')
    return lib


def _load():
    global _lib, _lib_err
    if _lib is not None or _lib_err is not None:
        return _lib
    tried = []
    # a chunk file has one writer and is read only after it was closed (main.py:516-526): advisory file locks add
    # nothing, and flock() on a network or overlay scratch directory is where such a job hangs; the user's own
    # setting wins (the library reads the variable when it starts up)
    os.environ.setdefault('HDF5_USE_FILE_LOCKING', 'FALSE')
    for cand in _candidates():
        try:
            _lib = _bind(C.CDLL(cand))
            return _lib
        except (OSError, AttributeError) as e:
            tried.append(f'{cand}: {e}')
    _lib_err = 'no usable libhdf5 (' + '; '.join(tried[:4]) + ' ...)'
    return None


def available():
    """True when a libhdf5 with the entry points used here can be loaded."""
    return _load() is not None


# ---------------------------------------------------------------------------------------------
#  Native helper (csrc/nfa_h5.cpp): the attributes / datasets of one object per call instead of one HDF5 call
#  per step per attribute from the interpreter (~18 us an attribute: seconds per thousand fitted pixels).
#  Optional: without the helper library the pure-ctypes paths below do the same work.
# ---------------------------------------------------------------------------------------------
_helper = None
_helper_tried = False
_NATIVE_ORDER = ('i1', 'i2', 'i4', 'i8', 'u1', 'u2', 'u4', 'u8', 'f4', 'f8')


def _fast():
    global _helper, _helper_tried
    if _helper_tried:
        return _helper
    _helper_tried = True
    lib = _load()
    path = Path(__file__).resolve().parent / 'lib' / 'libnestfit_amd_h5.so'
    if lib is None or not path.exists() or os.environ.get('NFA_HDF5_HELPER', '1') == '0':
        return None
    try:
        h = C.CDLL(str(path))
        h.nfa_h5_init.restype, h.nfa_h5_init.argtypes = C.c_int, [C.c_char_p]
        h.nfa_h5_write_items.restype = C.c_int
        h.nfa_h5_write_items.argtypes = [hid_t, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(hid_t), C.POINTER(C.c_int),
                                         C.POINTER(hsize_t), C.POINTER(C.c_void_p)]
        h.nfa_h5_read_attrs.restype = C.c_int
        h.nfa_h5_read_attrs.argtypes = [hid_t, hid_t, hid_t, C.POINTER(hid_t), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        h.nfa_h5_free.restype, h.nfa_h5_free.argtypes = None, [C.c_void_p]
        if h.nfa_h5_init(os.fsencode(lib._name)) != 0:
            return None
        h.native = (hid_t * len(_NATIVE_ORDER))(*[lib.T[np.dtype(c)] for c in _NATIVE_ORDER])
        _helper = h
    except (OSError, AttributeError):
        _helper = None
    return _helper


def _write_items_fast(h, loc, items, is_dataset, what):
    """items: [(name, type id, shape, buffer, keep-alive)] of one object, written in one native call."""
    n = len(items)
    if n == 0:
        return
    names = (C.c_char_p * n)(*[k.encode('utf-8') for k, *_ in items])
    types = (hid_t * n)(*[it[1] for it in items])
    ndims = (C.c_int * n)(*[len(it[2]) for it in items])
    dims = (hsize_t * (8 * n))()
    data = (C.c_void_p * n)()
    for i, (_k, _t, shape, buf, _keep) in enumerate(items):
        if len(shape) > 8:
            raise TypeError('more than eight axes')
        for j, d in enumerate(shape):
            dims[8 * i + j] = d
        # (the address without a ctypes object per item: there are some fifty items per fitted pixel)
        data[i] = buf.__array_interface__['data'][0] if isinstance(buf, np.ndarray) else C.addressof(buf)
    rc = h.nfa_h5_write_items(loc, int(is_dataset), n, names, types, ndims, dims, data)
    if rc != 0:
        raise Hdf5Error(f'HDF5: writing {what} {items[rc - 1][0] if rc > 0 else ""} failed')


_KIND_DTYPE = {0: 'i', 1: 'u', 2: 'f'}


def _read_attrs_fast(h, ty, loc, attrs):
    """All attributes of `loc` into the dict `attrs`, one native call and one packed buffer."""
    import struct
    out, nbytes = C.c_void_p(), C.c_uint64()
    if h.nfa_h5_read_attrs(loc, ty.vstr, ty.boolean, h.native, C.byref(out), C.byref(nbytes)) != 0:
        raise Hdf5Error('HDF5: reading attributes failed')
    try:
        blob = C.string_at(out, nbytes.value)
    finally:
        h.nfa_h5_free(out)
    (n,), pos = struct.unpack_from('<I', blob, 0), 4
    for _ in range(n):
        (nl,) = struct.unpack_from('<H', blob, pos)
        pos += 2
        name = blob[pos:pos + nl].decode('utf-8')
        pos += nl
        kind, itemsize, nd, _pad = struct.unpack_from('<4B', blob, pos)
        pos += 4
        shape = struct.unpack_from(f'<{nd}Q', blob, pos) if nd else ()
        pos += 8 * nd
        (nb,) = struct.unpack_from('<Q', blob, pos)
        pos += 8
        payload = blob[pos:pos + nb]
        pos += nb
        if kind == 255:
            raise Hdf5Error(f'attribute {name}: an HDF5 type the store does not use')
        if kind == 3:
            strings = [b.decode('utf-8') for b in payload.split(b'\0')[:-1]] if nb else []
            attrs[name] = strings[0] if not shape else (strings if len(shape) == 1 else
                                                        np.array(strings, dtype=object).reshape(shape))
        elif kind == 4:
            a = np.frombuffer(payload, dtype=np.int8).astype(bool).reshape(shape)
            attrs[name] = bool(a) if not shape else a
        else:
            a = np.frombuffer(payload, dtype=np.dtype(_KIND_DTYPE[kind] + str(itemsize))).reshape(shape)
            attrs[name] = a.item() if not shape else a.copy()


def library_version():
    lib = _need()
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _need():
    lib = _load()
    if lib is None:
        raise Hdf5Error(_lib_err)
    return lib


def _ok(code, what):
    if code < 0:
        raise Hdf5Error(f'HDF5: {what} failed')
    return code


# ---------------------------------------------------------------------------------------------
#  Python value  <->  (type, dataspace, buffer)
# ---------------------------------------------------------------------------------------------
class _Types:
    """The derived types of one file session: variable-length UTF-8 strings and h5py's boolean."""

    def __init__(self, lib):
        self.lib = lib
        self.vstr = _ok(lib.H5Tcopy(lib.T_C_S1), 'H5Tcopy')
        _ok(lib.H5Tset_size(self.vstr, H5T_VARIABLE), 'H5Tset_size')
        _ok(lib.H5Tset_cset(self.vstr, H5T_CSET_UTF8), 'H5Tset_cset')
        self.boolean = _ok(lib.H5Tenum_create(lib.T[np.dtype('i1')]), 'H5Tenum_create')
        for name, val in ((b'FALSE', 0), (b'TRUE', 1)):
            v = C.c_int8(val)
            _ok(lib.H5Tenum_insert(self.boolean, name, C.byref(v)), 'H5Tenum_insert')

    def close(self):
        self.lib.H5Tclose(self.vstr)
        self.lib.H5Tclose(self.boolean)


def _space_of(lib, shape):
    if len(shape) == 0:
        return _ok(lib.H5Screate(H5S_SCALAR), 'H5Screate')
    dims = (hsize_t * len(shape))(*shape)
    return _ok(lib.H5Screate_simple(len(shape), dims, None), 'H5Screate_simple')


def _encode(ty, value):
    """(type id, shape, ctypes buffer or ndarray, keep-alive) for a Python / numpy value."""
    lib = ty.lib
    if isinstance(value, (bytes, str)):
        value = np.array(value, dtype=object)
    arr = value if isinstance(value, np.ndarray) else np.asarray(value)
    if arr.dtype.kind in 'US' or (arr.dtype.kind == 'O' and all(isinstance(x, (str, bytes)) for x in arr.ravel())):
        raw = [x if isinstance(x, bytes) else str(x).encode('utf-8') for x in arr.ravel()]
        buf = (C.c_char_p * max(len(raw), 1))(*raw)
        return ty.vstr, arr.shape, buf, raw
    if arr.dtype.kind == 'b':
        a = np.ascontiguousarray(arr, dtype=np.int8)
        return ty.boolean, arr.shape, a, a
    if arr.dtype.kind == 'O':
        raise TypeError(f'no HDF5 representation for {value!r}')
    if arr.dtype.kind == 'f' and arr.dtype.itemsize == 2:
        arr = arr.astype(np.float32)
    dt = arr.dtype.newbyteorder('=')
    if dt not in lib.T:
        raise TypeError(f'no HDF5 representation for dtype {arr.dtype}')
    a = np.ascontiguousarray(arr, dtype=dt)
    return lib.T[dt], arr.shape, a, a


def _ptr(buf):
    return buf.ctypes.data_as(C.c_void_p) if isinstance(buf, np.ndarray) else C.cast(buf, C.c_void_p)


def _decode(ty, type_id, space_id, read, scalar_as_python):
    """The value behind `read(memtype, pointer)` for an object of file type `type_id` over `space_id`."""
    lib = ty.lib
    nd = lib.H5Sget_simple_extent_ndims(space_id)
    _ok(nd, 'H5Sget_simple_extent_ndims')
    shape = ()
    if nd > 0:
        dims = (hsize_t * nd)()
        lib.H5Sget_simple_extent_dims(space_id, dims, None)
        shape = tuple(int(d) for d in dims)
    n = int(np.prod(shape)) if shape else 1
    cls = lib.H5Tget_class(type_id)
    size = lib.H5Tget_size(type_id)
    if cls == H5T_STRING:
        if lib.H5Tis_variable_str(type_id) > 0:
            buf = (C.c_char_p * max(n, 1))()
            if n:
                _ok(read(ty.vstr, C.cast(buf, C.c_void_p)), 'read')
            out = [(b or b'').decode('utf-8') for b in buf[:n]]
            if n:
                lib.H5Dvlen_reclaim(ty.vstr, space_id, H5P_DEFAULT, C.cast(buf, C.c_void_p))
        else:                                           # fixed-length strings, as some writers store them
            mem = _ok(lib.H5Tcopy(type_id), 'H5Tcopy')
            raw = np.zeros(max(n, 1), dtype=f'S{size}')
            if n:
                _ok(read(mem, raw.ctypes.data_as(C.c_void_p)), 'read')
            lib.H5Tclose(mem)
            out = [r.decode('utf-8') for r in raw[:n]]
        if not shape:
            return out[0]
        return out if len(shape) == 1 else np.array(out, dtype=object).reshape(shape)
    if cls == H5T_ENUM and size == 1:                   # h5py's boolean
        a = np.zeros(max(n, 1), dtype=np.int8)
        if n:
            _ok(read(ty.boolean, a.ctypes.data_as(C.c_void_p)), 'read')
        a = a[:n].astype(bool).reshape(shape)
        return bool(a) if (not shape and scalar_as_python) else a
    if cls == H5T_FLOAT:
        dt = np.dtype(f'f{size}')
    elif cls == H5T_INTEGER:
        dt = np.dtype(('u' if lib.H5Tget_sign(type_id) == H5T_SGN_NONE else 'i') + str(size))
    else:
        raise Hdf5Error(f'HDF5 type class {cls} is not one the store uses')
    a = np.zeros(max(n, 1), dtype=dt)
    if n:
        _ok(read(lib.T[dt], a.ctypes.data_as(C.c_void_p)), 'read')
    a = a[:n].reshape(shape)
    if not shape and scalar_as_python:
        return a.item()
    return a


# ---------------------------------------------------------------------------------------------
#  A tree of groups  ->  file
# ---------------------------------------------------------------------------------------------
def write_tree(path, root, external=None):
    """Write the group tree under `root` (objects with `.attrs`, `._datasets`, `._children`: the store's
    `Group`) as the HDF5 file `path`.  `external(child)` says where a child that belongs to another file
    lives -- `(file name, object path)` -- and is stored as an external link; it returns None for the
    children that are written here."""
    lib = _need()
    path = Path(path)
    fid = lib.H5Fcreate(os.fsencode(str(path)), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    _ok(fid, f'creating {path}')
    ty = _Types(lib)
    try:
        _write_group(lib, ty, fid, root, external)
    finally:
        ty.close()
        _ok(lib.H5Fclose(fid), f'closing {path}')


def _write_attr(lib, ty, loc, name, value):
    if value is None:
        return                                          # h5py has no None either; the key is simply absent
    tid, shape, buf, _keep = _encode(ty, value)
    sid = _space_of(lib, shape)
    aid = lib.H5Acreate2(loc, name.encode('utf-8'), tid, sid, H5P_DEFAULT, H5P_DEFAULT)
    _ok(aid, f'creating attribute {name}')
    try:
        if int(np.prod(shape)) if shape else 1:
            _ok(lib.H5Awrite(aid, tid, _ptr(buf)), f'writing attribute {name}')
    finally:
        lib.H5Aclose(aid)
        lib.H5Sclose(sid)


def _write_group(lib, ty, gid, node, external):
    fast = _fast()
    if fast is not None:
        _write_items_fast(fast, gid, [(k,) + _encode(ty, v) for k, v in node.attrs.items() if v is not None], False, 'attribute')
        _write_items_fast(fast, gid, [(k,) + _encode(ty, d) for k, d in node._datasets.items()], True, 'dataset')
    for k, v in (node.attrs.items() if fast is None else ()):
        _write_attr(lib, ty, gid, k, v)
    for k, d in (node._datasets.items() if fast is None else ()):
        tid, shape, buf, _keep = _encode(ty, d)
        sid = _space_of(lib, shape)
        did = lib.H5Dcreate2(gid, k.encode('utf-8'), tid, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        _ok(did, f'creating dataset {k}')
        try:
            if int(np.prod(shape)) if shape else 1:
                _ok(lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, _ptr(buf)), f'writing dataset {k}')
        finally:
            lib.H5Dclose(did)
            lib.H5Sclose(sid)
    for k, child in node._children.items():
        where = external(child) if external else None
        if where is not None:
            fname, obj = where
            _ok(lib.H5Lcreate_external(os.fsencode(fname), obj.encode('utf-8'), gid, k.encode('utf-8'),
                                       H5P_DEFAULT, H5P_DEFAULT), f'linking {k}')
            continue
        cid = lib.H5Gcreate2(gid, k.encode('utf-8'), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        _ok(cid, f'creating group {k}')
        try:
            _write_group(lib, ty, cid, child, external)
        finally:
            lib.H5Gclose(cid)


# ---------------------------------------------------------------------------------------------
#  File  ->  a tree of groups
# ---------------------------------------------------------------------------------------------
def read_tree(path, root, on_external=None):
    """Fill the group tree `root` (the store's `Group`: `attrs`, `create_dataset`, `require_group`) from the
    HDF5 file `path`.  An external link is handed to `on_external(parent group, link name, file name,
    object path)`; without a handler it is skipped."""
    lib = _need()
    path = Path(path)
    fid = lib.H5Fopen(os.fsencode(str(path)), H5F_ACC_RDONLY, H5P_DEFAULT)
    _ok(fid, f'opening {path}')
    ty = _Types(lib)
    try:
        _read_group(lib, ty, fid, root, on_external)
    finally:
        ty.close()
        lib.H5Fclose(fid)


def _names(lib, gid, iterate, cb_type, pick):
    out = []

    def cb(_loc, name, info, _data):
        out.append(pick(name.decode('utf-8'), info))
        return 0
    idx = hsize_t(0)
    _ok(iterate(gid, H5_INDEX_NAME, H5_ITER_INC, C.byref(idx), cb_type(cb), None), 'iterating')
    return out


def _read_group(lib, ty, gid, node, on_external):
    fast = _fast()
    if fast is not None:
        _read_attrs_fast(fast, ty, gid, node.attrs)
    for name in (_names(lib, gid, lib.nf_aiterate, _ATTR_CB, lambda n, _i: n) if fast is None else ()):
        aid = _ok(lib.H5Aopen(gid, name.encode('utf-8'), H5P_DEFAULT), f'opening attribute {name}')
        tid, sid = lib.H5Aget_type(aid), lib.H5Aget_space(aid)
        try:
            node.attrs[name] = _decode(ty, tid, sid, lambda mem, p: lib.H5Aread(aid, mem, p), True)
        finally:
            lib.H5Tclose(tid)
            lib.H5Sclose(sid)
            lib.H5Aclose(aid)
    for name, kind in _names(lib, gid, lib.nf_literate, _LINK_CB, lambda n, info: (n, info[0])):
        bname = name.encode('utf-8')
        if kind == H5L_TYPE_EXTERNAL:
            if on_external is None:
                continue
            buf = C.create_string_buffer(4096)
            _ok(lib.H5Lget_val(gid, bname, buf, len(buf), H5P_DEFAULT), f'reading link {name}')
            flags, fname, obj = C.c_uint(), C.c_char_p(), C.c_char_p()
            _ok(lib.H5Lunpack_elink_val(buf, len(buf), C.byref(flags), C.byref(fname), C.byref(obj)),
                f'unpacking link {name}')
            on_external(node, name, os.fsdecode(fname.value), obj.value.decode('utf-8'))
            continue
        if kind != H5L_TYPE_HARD:
            continue
        oid = _ok(lib.H5Oopen(gid, bname, H5P_DEFAULT), f'opening {name}')
        try:
            what = lib.H5Iget_type(oid)
            if what == H5I_GROUP:
                _read_group(lib, ty, oid, node.require_group(name), on_external)
            elif what == H5I_DATASET:
                tid, sid = lib.H5Dget_type(oid), lib.H5Dget_space(oid)
                try:
                    data = _decode(ty, tid, sid,
                                   lambda mem, p: lib.H5Dread(oid, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, p), False)
                finally:
                    lib.H5Tclose(tid)
                    lib.H5Sclose(sid)
                node.create_dataset(name, data=data)
        finally:
            lib.H5Oclose(oid)
