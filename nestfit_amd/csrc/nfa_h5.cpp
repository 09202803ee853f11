// libnestfit_amd_h5.so -- host helper of nestfit_amd/hdf5.py (the result store as real HDF5, SURVEY.md 8f-3;
// reference layout nestfit/main.py:233-377, docs/store_spec.rst:45-110).  A fitted map is tens of thousands of
// small groups with some twenty attributes each (core.pyx:648-676); one interpreter-level call into the HDF5 C
// library per attribute per step (create / write / close, or open / type / space / read / close) costs ~18 us an
// attribute -- seconds per thousand pixels.  Here the attributes (or datasets) of ONE object travel in ONE call:
// Python hands over flat arrays, the loop over the HDF5 calls runs natively.  The HDF5 library itself is the one
// hdf5.py found: opened again by path with dlopen (same instance, same identifiers).  No HIP in here.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

typedef int64_t  hid_t;
typedef int      herr_t;
typedef uint64_t hsize_t;

namespace {
struct Api {
    void *lib = nullptr;
    hid_t  (*H5Screate)(int) = nullptr;
    hid_t  (*H5Screate_simple)(int, const hsize_t *, const hsize_t *) = nullptr;
    herr_t (*H5Sclose)(hid_t) = nullptr;
    int    (*H5Sget_simple_extent_ndims)(hid_t) = nullptr;
    int    (*H5Sget_simple_extent_dims)(hid_t, hsize_t *, hsize_t *) = nullptr;
    hid_t  (*H5Acreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*H5Awrite)(hid_t, hid_t, const void *) = nullptr;
    herr_t (*H5Aread)(hid_t, hid_t, void *) = nullptr;
    herr_t (*H5Aclose)(hid_t) = nullptr;
    hid_t  (*H5Aopen)(hid_t, const char *, hid_t) = nullptr;
    hid_t  (*H5Aget_type)(hid_t) = nullptr;
    hid_t  (*H5Aget_space)(hid_t) = nullptr;
    herr_t (*H5Aiterate2)(hid_t, int, int, hsize_t *, herr_t (*)(hid_t, const char *, const void *, void *), void *) = nullptr;
    hid_t  (*H5Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *) = nullptr;
    herr_t (*H5Dclose)(hid_t) = nullptr;
    herr_t (*H5Dvlen_reclaim)(hid_t, hid_t, hid_t, void *) = nullptr;
    int    (*H5Tget_class)(hid_t) = nullptr;
    size_t (*H5Tget_size)(hid_t) = nullptr;
    int    (*H5Tget_sign)(hid_t) = nullptr;
    int    (*H5Tis_variable_str)(hid_t) = nullptr;
    hid_t  (*H5Tcopy)(hid_t) = nullptr;
    herr_t (*H5Tclose)(hid_t) = nullptr;
} H;

template <typename F> bool bind(F &fn, const char *name) {
    fn = (F)dlsym(H.lib, name);
    return fn != nullptr;
}
}  // namespace

extern "C" {

// 0 = ready.  `libpath`: the HDF5 library nestfit_amd/hdf5.py loaded (dlopen returns that same instance).
int nfa_h5_init(const char *libpath) {
    if (H.lib) return 0;
    H.lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!H.lib) return 1;
    bool ok = bind(H.H5Screate, "H5Screate") & bind(H.H5Screate_simple, "H5Screate_simple") & bind(H.H5Sclose, "H5Sclose") &
              bind(H.H5Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims") &
              bind(H.H5Sget_simple_extent_dims, "H5Sget_simple_extent_dims") & bind(H.H5Acreate2, "H5Acreate2") &
              bind(H.H5Awrite, "H5Awrite") & bind(H.H5Aread, "H5Aread") & bind(H.H5Aclose, "H5Aclose") &
              bind(H.H5Aopen, "H5Aopen") & bind(H.H5Aget_type, "H5Aget_type") & bind(H.H5Aget_space, "H5Aget_space") &
              bind(H.H5Aiterate2, "H5Aiterate2") & bind(H.H5Dcreate2, "H5Dcreate2") & bind(H.H5Dwrite, "H5Dwrite") &
              bind(H.H5Dclose, "H5Dclose") & bind(H.H5Dvlen_reclaim, "H5Dvlen_reclaim") & bind(H.H5Tget_class, "H5Tget_class") &
              bind(H.H5Tget_size, "H5Tget_size") & bind(H.H5Tget_sign, "H5Tget_sign") &
              bind(H.H5Tis_variable_str, "H5Tis_variable_str") & bind(H.H5Tcopy, "H5Tcopy") & bind(H.H5Tclose, "H5Tclose");
    if (!ok) { H.lib = nullptr; return 2; }
    return 0;
}

// n attributes (is_dataset == 0) or datasets (1) of the object `loc`: name, HDF5 type identifier, rank and
// extents (dims: 8 per item), data (NULL or an empty extent: nothing is written).  Returns 0, or 1 + the index
// of the item that failed.
int nfa_h5_write_items(hid_t loc, int is_dataset, int n, const char *const *names, const hid_t *types, const int *ndims,
                       const hsize_t *dims, const void *const *data) {
    if (!H.lib) return -1;
    for (int i = 0; i < n; ++i) {
        const hsize_t *d = dims + 8 * (size_t)i;
        hsize_t count = 1;
        for (int k = 0; k < ndims[i]; ++k) count *= d[k];
        const hid_t sid = ndims[i] == 0 ? H.H5Screate(0) : H.H5Screate_simple(ndims[i], d, nullptr);
        if (sid < 0) return 1 + i;
        herr_t rc = 0;
        if (is_dataset) {
            const hid_t did = H.H5Dcreate2(loc, names[i], types[i], sid, 0, 0, 0);
            if (did < 0) { H.H5Sclose(sid); return 1 + i; }
            if (count && data[i]) rc = H.H5Dwrite(did, types[i], 0, 0, 0, data[i]);
            H.H5Dclose(did);
        } else {
            const hid_t aid = H.H5Acreate2(loc, names[i], types[i], sid, 0, 0);
            if (aid < 0) { H.H5Sclose(sid); return 1 + i; }
            if (count && data[i]) rc = H.H5Awrite(aid, types[i], data[i]);
            H.H5Aclose(aid);
        }
        H.H5Sclose(sid);
        if (rc < 0) return 1 + i;
    }
    return 0;
}

// All attributes of `loc` as one packed buffer (malloc'd, *out; the caller frees it with nfa_h5_free):
//   u32 n | n x { u16 name_len, name, u8 kind, u8 itemsize, u8 ndim, u8 pad, u64 dims[ndim], u64 nbytes, payload }
// kind: 0 signed int, 1 unsigned int, 2 float, 3 strings (payload: the strings, NUL-terminated, back to back),
// 4 boolean (h5py's FALSE/TRUE enumeration over one byte), 255 a type the store does not use (no payload).
// `vstr` / `boolean`: the memory types hdf5.py made for variable-length strings and booleans; `native[k]` the
// native type identifiers in the order i1 i2 i4 i8 u1 u2 u4 u8 f4 f8.
struct Walk {
    std::vector<uint8_t> buf;
    uint32_t n = 0;
    hid_t vstr, boolean;
    const hid_t *native;
    bool failed = false;
};

static void put(std::vector<uint8_t> &b, const void *p, size_t n) { b.insert(b.end(), (const uint8_t *)p, (const uint8_t *)p + n); }

static herr_t walk_attr(hid_t loc, const char *name, const void *, void *ud) {
    Walk &w = *(Walk *)ud;
    const hid_t aid = H.H5Aopen(loc, name, 0);
    if (aid < 0) { w.failed = true; return -1; }
    const hid_t tid = H.H5Aget_type(aid), sid = H.H5Aget_space(aid);
    const int nd = H.H5Sget_simple_extent_ndims(sid);
    hsize_t dims[8] = {0};
    if (nd > 0 && nd <= 8) H.H5Sget_simple_extent_dims(sid, dims, nullptr);
    uint64_t count = 1;
    for (int k = 0; k < nd && k < 8; ++k) count *= dims[k];
    const int cls = H.H5Tget_class(tid);
    const size_t size = H.H5Tget_size(tid);
    uint8_t kind = 255, itemsize = (uint8_t)(size < 256 ? size : 0);
    std::vector<uint8_t> payload;
    bool ok = nd >= 0 && nd <= 8;
    if (ok && cls == 0 && (size == 1 || size == 2 || size == 4 || size == 8)) {             // integer
        const bool uns = H.H5Tget_sign(tid) == 0;
        kind = uns ? 1 : 0;
        const int slot = (uns ? 4 : 0) + (size == 1 ? 0 : size == 2 ? 1 : size == 4 ? 2 : 3);
        payload.resize(count * size);
        if (count) ok = H.H5Aread(aid, w.native[slot], payload.data()) >= 0;
    } else if (ok && cls == 1 && (size == 4 || size == 8)) {                                  // float
        kind = 2;
        payload.resize(count * size);
        if (count) ok = H.H5Aread(aid, w.native[size == 4 ? 8 : 9], payload.data()) >= 0;
    } else if (ok && cls == 8 && size == 1) {                                                 // h5py's boolean
        kind = 4;
        payload.resize(count);
        if (count) ok = H.H5Aread(aid, w.boolean, payload.data()) >= 0;
    } else if (ok && cls == 3) {                                                              // strings
        kind = 3;
        if (H.H5Tis_variable_str(tid) > 0) {
            std::vector<char *> ptrs(count ? count : 1, nullptr);
            if (count) ok = H.H5Aread(aid, w.vstr, ptrs.data()) >= 0;
            if (ok) {
                for (uint64_t k = 0; k < count; ++k) {
                    const char *s = ptrs[k] ? ptrs[k] : "";
                    put(payload, s, strlen(s) + 1);
                }
                if (count) H.H5Dvlen_reclaim(w.vstr, sid, 0, ptrs.data());
            }
        } else {
            std::vector<char> raw(count * size + 1, 0);
            const hid_t mem = H.H5Tcopy(tid);
            if (count) ok = H.H5Aread(aid, mem, raw.data()) >= 0;
            H.H5Tclose(mem);
            for (uint64_t k = 0; ok && k < count; ++k) {
                const size_t len = strnlen(raw.data() + k * size, size);
                put(payload, raw.data() + k * size, len);
                payload.push_back(0);
            }
        }
        itemsize = 0;
    }
    H.H5Tclose(tid); H.H5Sclose(sid); H.H5Aclose(aid);
    if (!ok) { w.failed = true; return -1; }
    const uint16_t nl = (uint16_t)strlen(name);
    put(w.buf, &nl, 2); put(w.buf, name, nl);
    const uint8_t head[4] = {kind, itemsize, (uint8_t)nd, 0};
    put(w.buf, head, 4);
    for (int k = 0; k < nd; ++k) { const uint64_t d = dims[k]; put(w.buf, &d, 8); }
    const uint64_t nb = kind == 255 ? 0 : payload.size();
    put(w.buf, &nb, 8);
    if (nb) put(w.buf, payload.data(), nb);
    w.n += 1;
    return 0;
}

int nfa_h5_read_attrs(hid_t loc, hid_t vstr, hid_t boolean, const hid_t *native, uint8_t **out, uint64_t *nbytes) {
    if (!H.lib || !out || !nbytes) return -1;
    Walk w;
    w.vstr = vstr; w.boolean = boolean; w.native = native;
    w.buf.resize(4);
    hsize_t idx = 0;
    const herr_t rc = H.H5Aiterate2(loc, 0 /* by name */, 0 /* increasing */, &idx, walk_attr, &w);
    if (rc < 0 || w.failed) return 1;
    memcpy(w.buf.data(), &w.n, 4);
    uint8_t *p = (uint8_t *)malloc(w.buf.size());
    if (!p) return 2;
    memcpy(p, w.buf.data(), w.buf.size());
    *out = p; *nbytes = w.buf.size();
    return 0;
}

void nfa_h5_free(void *p) { free(p); }

}  // extern "C"
