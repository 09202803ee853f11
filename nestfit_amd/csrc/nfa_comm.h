// nfa_comm.h -- the one exchange step of the sharded cube fit: RCCL over xGMI, one process per GPU.
//
// The reference forks one process per longitude stripe and exchanges nothing while it samples
// (nestfit/main.py:516-523, 565-571); its ranks meet only through the chunk files
// (docs/store_spec.rst:12-32).  The GPU build keeps that: no data-path collective.  What the ranks
// do exchange is an end-of-run all-gather of fixed-size per-pixel records, a max / sum over ranks of
// a few doubles (timing, evaluation counts) and a barrier -- a few hundred KB, latency bound.
// librccl.so is opened at run time (dlopen), so the engine has no link-time dependency on it and a
// single-GPU user never loads it.  Bootstrap: rank 0 creates the ncclUniqueId, the host language
// carries its 128 bytes to the other ranks (a file, the launcher's store, MPI: the caller's choice;
// nestfit_amd/comm.py uses a file next to MASTER_PORT).
#pragma once
#include <dlfcn.h>

struct nfa_nccl_id { char internal[128]; };        // ncclUniqueId (rccl.h:43)
typedef void *nfa_nccl_comm;
struct NcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(nfa_nccl_id *) = nullptr;
    int (*CommInitRank)(nfa_nccl_comm *, int, nfa_nccl_id, int) = nullptr;
    int (*CommDestroy)(nfa_nccl_comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nfa_nccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nfa_nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static NcclApi g_nccl;
enum { NFA_NCCL_SUM = 0, NFA_NCCL_MAX = 2, NFA_NCCL_MIN = 3, NFA_NCCL_FLOAT64 = 8 };   // rccl.h:448-467

static int nccl_load() {
    if (g_nccl.lib) return NFA_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return fail(NFA_ERR_DEVICE, std::string("cannot open librccl.so: ") + dlerror());
    NcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce)
        return fail(NFA_ERR_DEVICE, "librccl.so lacks an expected entry point");
    g_nccl = a;
    return NFA_OK;
}
#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        int e_ = (expr);                                                                     \
        if (e_ != 0)                                                                         \
            return fail(NFA_ERR_DEVICE, std::string(#expr) + ": " +                          \
                        (g_nccl.GetErrorString ? g_nccl.GetErrorString(e_) : "RCCL error")); \
    } while (0)

struct nfa_comm {
    nfa_nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    double *d_buf = nullptr;           // staging: [send | recv]
    int64_t cap = 0;                   // doubles in each half
};

static int comm_reserve(nfa_comm *c, int64_t send, int64_t recv) {
    const int64_t need = std::max(send, recv);
    if (need <= c->cap) return NFA_OK;
    (void)hipFree(c->d_buf); c->d_buf = nullptr; c->cap = 0;
    const int64_t cap = std::max<int64_t>(need, 1024);
    HIP_TRY(hipMalloc(&c->d_buf, sizeof(double) * 2 * cap));
    c->cap = cap;
    return NFA_OK;
}

extern "C" {

int nfa_comm_unique_id(unsigned char *id128) {
    if (!id128) return fail(NFA_ERR_ARG, "null argument");
    int rc = engine_init(); if (rc) return rc;
    rc = nccl_load(); if (rc) return rc;
    nfa_nccl_id id;
    NCCL_TRY(g_nccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return NFA_OK;
}

int nfa_comm_create(nfa_comm **out, const unsigned char *id128, int rank, int world) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(NFA_ERR_ARG, "bad communicator arguments");
    int rc = engine_init(); if (rc) return rc;          // the rank's device: nfa_set_device before this call
    rc = nccl_load(); if (rc) return rc;
    nfa_comm *c = new nfa_comm();
    c->rank = rank; c->world = world;
    nfa_nccl_id id;
    memcpy(id.internal, id128, 128);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(NFA_ERR_DEVICE, "hipStreamCreate failed"); }
    int e = g_nccl.CommInitRank(&c->comm, world, id, rank);
    if (e != 0) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(NFA_ERR_DEVICE, std::string("ncclCommInitRank: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(e) : "RCCL error"));
    }
    *out = c;
    return NFA_OK;
}

int nfa_comm_destroy(nfa_comm *c) {
    if (!c) return NFA_OK;
    if (c->comm) (void)g_nccl.CommDestroy(c->comm);
    (void)hipFree(c->d_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return NFA_OK;
}

int nfa_comm_rank(const nfa_comm *c) { return c ? c->rank : 0; }
int nfa_comm_world(const nfa_comm *c) { return c ? c->world : 1; }

int nfa_comm_allgather(nfa_comm *c, const double *send, int64_t count, double *recv) {
    if (!c || !send || !recv || count < 0) return fail(NFA_ERR_ARG, "bad all-gather arguments");
    if (count == 0) return NFA_OK;
    int rc = engine_init(); if (rc) return rc;
    rc = comm_reserve(c, count, count * c->world); if (rc) return rc;
    double *d_send = c->d_buf, *d_recv = c->d_buf + c->cap;
    HIP_TRY(hipMemcpyAsync(d_send, send, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(g_nccl.AllGather(d_send, d_recv, (size_t)count, NFA_NCCL_FLOAT64, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(recv, d_recv, sizeof(double) * count * c->world, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NFA_OK;
}

int nfa_comm_allreduce(nfa_comm *c, double *values, int64_t count, int op) {
    if (!c || !values || count < 0 || (op != NFA_NCCL_SUM && op != NFA_NCCL_MAX && op != NFA_NCCL_MIN))
        return fail(NFA_ERR_ARG, "bad all-reduce arguments");
    if (count == 0) return NFA_OK;
    int rc = engine_init(); if (rc) return rc;
    rc = comm_reserve(c, count, count); if (rc) return rc;
    double *d_send = c->d_buf, *d_recv = c->d_buf + c->cap;
    HIP_TRY(hipMemcpyAsync(d_send, values, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(g_nccl.AllReduce(d_send, d_recv, (size_t)count, NFA_NCCL_FLOAT64, op, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(values, d_recv, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NFA_OK;
}

int nfa_comm_barrier(nfa_comm *c) {
    double one = 1.0;
    return nfa_comm_allreduce(c, &one, 1, NFA_NCCL_SUM);
}

}  // extern "C"
