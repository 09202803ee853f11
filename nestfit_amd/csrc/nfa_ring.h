// nfa_ring.h -- cross-process transport for the callback broker, host code only.
//
// The reference fits a cube with one process per stripe (nestfit/main.py:516-523), each with its own
// MultiNest instance (Fortran, global state: one per process) that hands LogLike one point at a time
// (nestfit/core/cmultinest.pxd:27-28).  Processes that each own a runner time-slice the GPU (two of
// them overlap, no more).  Here the sampler processes do not touch the GPU at all: each one owns a slot
// of a POSIX shared-memory ring, writes its point there and sleeps on the slot; ONE process serves the
// ring, gathers the posted slots into a batch for the engine and writes the results back.
//
//   /dev/shm/<name>:  Header | Slot 0 | Slot 1 | ...      (slot stride: 64-byte multiple)
//   slot state: FREE -> (client) POSTED -> (a server's poll) CLAIMED -> (its complete) DONE -> (client) FREE
//
// Waiting is a short spin, then a futex on the word that changes (the slot's state for a client, the
// header's post counter for the server): no busy process per sampler.  This file is compiled twice:
// into the engine library (which adds nfa_ring_serve, the loop around nfa_runner_loglike_batch) and,
// alone, into libnestfit_amd_ring.so -- the only library a sampler process loads (no HIP in it).
#pragma once
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <linux/futex.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <signal.h>
#include <sys/syscall.h>
#include <unistd.h>

#define NFA_RING_MAGIC   0x4e46524eu       // "NFRN"
#define NFA_RING_VERSION 4u
#define NFA_RING_MAXBATCH 1024             // points one serving round hands the engine at most
#define NFA_RING_MAXPOINTS 64              // points a client may post in one call (nfa_ring_loglike_many)
#define NFA_RING_GRACE_MS 5000             // a client gives up after this long without any serving loop on the ring

enum { RING_FREE = 0, RING_POSTED = 1, RING_DONE = 2, RING_CLAIMED = 3 };

struct RingHeader {
    uint32_t magic, version;
    int32_t  n_slots, ndim;
    uint64_t slot_stride, total_bytes;
    int32_t  server_pid, max_points;       // the creator (a ring whose creator is gone is stale); points per slot
    std::atomic<uint32_t> stop;            // set by nfa_ring_stop: everybody leaves
    std::atomic<uint32_t> posts;           // bumped by every post: the word the server sleeps on
    std::atomic<uint32_t> n_attached;      // clients holding a slot
    std::atomic<uint32_t> n_servers;       // serving loops at work (each with a runner of its own): they share the clients
    std::atomic<uint32_t> servers_asleep;  // serving loops inside a futex wait on `posts`: only then does a post pay for a wake call
    std::atomic<uint64_t> n_batches, n_evals, max_batch_seen;
    std::atomic<int64_t>  last_serve_us;   // heartbeat (CLOCK_MONOTONIC): refreshed by every nfa_ring_poll / nfa_ring_complete --
                                           // a server built on poll / complete alone (no nfa_ring_serve) is seen through it
    uint8_t  pad[56];
};

struct RingSlot {
    std::atomic<uint32_t> state;
    std::atomic<uint32_t> owner;           // 0 = nobody, else the pid of the client holding the slot
    std::atomic<uint32_t> asleep;          // the client is inside a futex wait on `state` (a spinning one needs no wake call)
    std::atomic<uint32_t> gen;             // bumped whenever the slot changes hands: a result claimed under an older
                                           // generation (its client died, the slot was inherited) is dropped, not delivered
    int32_t  pix, rc, n_points, pad;
    double   data[1];                      // lnl[max_points], then cube[max_points][ndim]
};

struct nfa_ring {
    RingHeader *hdr = nullptr;
    uint8_t    *base = nullptr;
    size_t      bytes = 0;
    std::string name;
    bool        creator = false;
    int         slot = -1;                 // client: the slot it holds
    uint32_t   *claim_gen = nullptr;       // server: generation of every slot at the time this handle claimed it
    void       *dev_base = nullptr;        // engine library: the mapping as the device addresses it (nfa_ring_serve_device)
};

#ifndef NFA_RING_STANDALONE
// inside the engine library fail() and the error codes come from nfa_engine.hip
#else
#define NFA_OK 0
#define NFA_ERR_ARG 1
#define NFA_ERR_STATE 3
static thread_local char g_ring_err[256];
static int fail(int code, const char *msg) { snprintf(g_ring_err, sizeof g_ring_err, "%s", msg); return code; }
extern "C" const char *nfa_ring_last_error(void) { return g_ring_err; }
#endif

static inline RingSlot *ring_slot(const nfa_ring *r, int k) {
    return (RingSlot *)(r->base + sizeof(RingHeader) + (size_t)k * r->hdr->slot_stride);
}
static inline double *slot_lnl(RingSlot *s) { return s->data; }
static inline double *slot_cube(const RingHeader *h, RingSlot *s) { return s->data + h->max_points; }
// a process that no longer runs: no such pid, or a zombie (it has exited, its parent has not collected it yet --
// kill(pid, 0) still succeeds for those; /proc/<pid>/stat says "Z" behind the command name)
static inline bool ring_pid_gone(int32_t pid) {
    if (pid <= 0) return false;
    if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) return true;
    char path[64], buf[512];
    snprintf(path, sizeof path, "/proc/%d/stat", (int)pid);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    const ssize_t n = read(fd, buf, sizeof buf - 1);
    close(fd);
    if (n <= 0) return false;
    buf[n] = 0;
    const char *q = strrchr(buf, ')');                        // the command name may hold anything, also ')'
    return q && q[1] == ' ' && (q[2] == 'Z' || q[2] == 'X');
}

static inline long ring_futex(std::atomic<uint32_t> *word, int op, uint32_t val, const timespec *ts) {
    return syscall(SYS_futex, (uint32_t *)word, op, val, ts, nullptr, 0);      // shared (not _PRIVATE): across processes
}

static inline int64_t ring_now_us() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (int64_t)t.tv_sec * 1000000 + t.tv_nsec / 1000;
}

// how long a client waits without any sign of a serving loop (NFA_RING_GRACE_MS in the environment overrides the default)
static inline int64_t ring_grace_us() {
    static const int64_t us = [] {
        const char *e = getenv("NFA_RING_GRACE_MS");
        const long v = e ? atol(e) : 0;
        return (int64_t)(v > 0 ? v : NFA_RING_GRACE_MS) * 1000;
    }();
    return us;
}

static inline void ring_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

static std::string ring_shm_name(const char *name) {
    std::string s = name && name[0] == '/' ? name : std::string("/") + (name ? name : "");
    return s;
}

extern "C" {

// Server side: create the ring `name` (a POSIX shared-memory object) with n_slots slots, each for up to max_points
// points of ndim doubles (nfa_ring_create: one point per slot, MultiNest's one LogLike per call).
int nfa_ring_create_multi(nfa_ring **out, const char *name, int n_slots, int ndim, int max_points) {
    if (!out || !name || !name[0]) return fail(NFA_ERR_ARG, "null argument");
    if (n_slots < 1 || n_slots > 4096 || ndim < 1 || ndim > 4096 || max_points < 1 || max_points > NFA_RING_MAXPOINTS)
        return fail(NFA_ERR_ARG, "bad ring shape");
    const std::string shm = ring_shm_name(name);
    const size_t stride = (offsetof(RingSlot, data) + sizeof(double) * (size_t)max_points * (size_t)(ndim + 1) + 63) / 64 * 64;
    const size_t bytes = sizeof(RingHeader) + stride * (size_t)n_slots;
    {   // a ring of that name whose server is alive is somebody's: refuse; one left by a crashed server goes
        const int old = shm_open(shm.c_str(), O_RDWR, 0600);
        if (old >= 0) {
            RingHeader h0;
            const bool whole = pread(old, &h0, sizeof h0, 0) == (ssize_t)sizeof h0;
            close(old);
            if (whole && h0.magic == NFA_RING_MAGIC && h0.server_pid > 0 && !ring_pid_gone(h0.server_pid) && !h0.stop.load())
                return fail(NFA_ERR_STATE, "a ring of that name is being served");
            shm_unlink(shm.c_str());
        }
    }
    const int fd = shm_open(shm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(NFA_ERR_STATE, "shm_open failed");
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(shm.c_str()); return fail(NFA_ERR_STATE, "ftruncate failed"); }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(shm.c_str()); return fail(NFA_ERR_STATE, "mmap failed"); }
    memset(p, 0, bytes);
    nfa_ring *r = new nfa_ring();
    r->base = (uint8_t *)p; r->hdr = (RingHeader *)p; r->bytes = bytes; r->name = shm; r->creator = true;
    r->claim_gen = new uint32_t[n_slots]();
    r->hdr->version = NFA_RING_VERSION; r->hdr->n_slots = n_slots; r->hdr->ndim = ndim; r->hdr->max_points = max_points;
    r->hdr->slot_stride = stride; r->hdr->total_bytes = bytes; r->hdr->server_pid = (int32_t)getpid();
    std::atomic_thread_fence(std::memory_order_release);
    r->hdr->magic = NFA_RING_MAGIC;                            // last: an attaching client waits for it
    *out = r;
    return NFA_OK;
}
int nfa_ring_create(nfa_ring **out, const char *name, int n_slots, int ndim) {
    return nfa_ring_create_multi(out, name, n_slots, ndim, 1);
}

// Client side: map the ring `name` and take a free slot (wait_ms: how long to wait for the ring to appear).
int nfa_ring_attach(nfa_ring **out, const char *name, int wait_ms) {
    if (!out || !name || !name[0]) return fail(NFA_ERR_ARG, "null argument");
    const std::string shm = ring_shm_name(name);
    const int64_t t_end = ring_now_us() + (int64_t)wait_ms * 1000;
    struct stat st;
    void *p = nullptr;
    RingHeader *h = nullptr;
    // until the deadline: no object yet, an object still being set up, or one left behind by a server that
    // died (its successor unlinks and recreates it) all mean "look again in a millisecond"
    for (const char *why = "no such ring";; usleep(1000)) {
        if (p) { munmap(p, (size_t)st.st_size); p = nullptr; }
        if (ring_now_us() >= t_end && why) return fail(NFA_ERR_STATE, why);
        const int fd = shm_open(shm.c_str(), O_RDWR, 0600);
        if (fd < 0) { why = "no such ring"; continue; }
        if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(RingHeader)) { close(fd); why = "ring never initialised"; continue; }
        p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { p = nullptr; return fail(NFA_ERR_STATE, "mmap failed"); }
        h = (RingHeader *)p;
        if (*(volatile uint32_t *)&h->magic != NFA_RING_MAGIC) { why = "ring never initialised"; continue; }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (ring_pid_gone(h->server_pid)) { why = "stale ring (its server is gone)"; continue; }
        break;
    }
    if (h->version != NFA_RING_VERSION || h->total_bytes != (uint64_t)st.st_size) {
        munmap(p, (size_t)st.st_size);
        return fail(NFA_ERR_STATE, "ring layout mismatch");
    }
    nfa_ring *r = new nfa_ring();
    r->base = (uint8_t *)p; r->hdr = h; r->bytes = (size_t)st.st_size; r->name = shm;
    const uint32_t me = (uint32_t)getpid();
    bool inherited = false;                        // the slot of a client that died without closing: still counted
    for (int k = 0; k < h->n_slots && r->slot < 0; ++k) {
        uint32_t nobody = 0;
        if (ring_slot(r, k)->owner.compare_exchange_strong(nobody, me, std::memory_order_acq_rel)) r->slot = k;
    }
    for (int k = 0; k < h->n_slots && r->slot < 0; ++k) {
        uint32_t owner = ring_slot(r, k)->owner.load(std::memory_order_acquire);
        if (owner != 0 && owner != me && ring_pid_gone((int32_t)owner) &&
            ring_slot(r, k)->owner.compare_exchange_strong(owner, me, std::memory_order_acq_rel)) {
            r->slot = k;
            inherited = true;
        }
    }
    if (r->slot < 0) { munmap(p, r->bytes); delete r; return fail(NFA_ERR_STATE, "no free slot in the ring"); }
    // a new generation first: whatever a server still holds for the previous owner of this slot is dropped on delivery
    ring_slot(r, r->slot)->gen.fetch_add(1, std::memory_order_acq_rel);
    ring_slot(r, r->slot)->state.store(RING_FREE, std::memory_order_release);
    if (!inherited) h->n_attached.fetch_add(1, std::memory_order_acq_rel);
    ring_futex(&h->posts, FUTEX_WAKE, 1, nullptr);             // a waiting server re-evaluates its target
    *out = r;
    return NFA_OK;
}

int nfa_ring_ndim(const nfa_ring *r) { return r && r->hdr ? r->hdr->ndim : -1; }
int nfa_ring_max_points(const nfa_ring *r) { return r && r->hdr ? r->hdr->max_points : -1; }
int nfa_ring_slot(const nfa_ring *r) { return r ? r->slot : -1; }

// Client: give the slot back and unmap.  Server (creator): unmap and remove the shared-memory object.
int nfa_ring_close(nfa_ring *r) {
    if (!r) return NFA_OK;
    if (r->slot >= 0) {
        ring_slot(r, r->slot)->owner.store(0, std::memory_order_release);
        r->hdr->n_attached.fetch_sub(1, std::memory_order_acq_rel);
        ring_futex(&r->hdr->posts, FUTEX_WAKE, 1, nullptr);
    }
#ifndef NFA_RING_STANDALONE
    if (r->dev_base) (void)hipHostUnregister(r->base);
#endif
    munmap(r->base, r->bytes);
    if (r->creator) shm_unlink(r->name.c_str());
    delete[] r->claim_gen;
    delete r;
    return NFA_OK;
}

// Everybody leaves: blocked clients return NFA_ERR_STATE, a serving loop returns.
int nfa_ring_stop(nfa_ring *r) {
    if (!r) return fail(NFA_ERR_ARG, "null argument");
    r->hdr->stop.store(1, std::memory_order_release);
    r->hdr->posts.fetch_add(1, std::memory_order_acq_rel);
    ring_futex(&r->hdr->posts, FUTEX_WAKE, INT32_MAX, nullptr);
    for (int k = 0; k < r->hdr->n_slots; ++k) ring_futex(&ring_slot(r, k)->state, FUTEX_WAKE, INT32_MAX, nullptr);
    return NFA_OK;
}

// Client: blocking LogLike through the ring for k points at once (k <= the ring's max_points; all against pixel
// `pix`, < 0 = the runner's single pixel).  `cubes` (k x ndim doubles, unit cube) is overwritten with the physical
// parameters like AmmoniaRunner.c_loglikelihood (ammonia.pyx:423-432); lnew[k].  MultiNest asks for one point per
// call (k = 1: nfa_ring_loglike); a sampler that draws its next proposals independently of each other -- points
// uniform in the current bounding ellipsoid are -- may post several and use them in order.
// Returns NFA_ERR_STATE when the ring was stopped, when the process that created the ring is gone, or when no
// serving loop has shown on the ring for NFA_RING_GRACE_MS -- neither a thread inside nfa_ring_serve nor the heartbeat
// that nfa_ring_poll / nfa_ring_complete leave --: a client never waits for a dead server.  A request a server has
// claimed is never timed out while the ring's creator lives (an evaluator may take as long as it likes over a batch).
int nfa_ring_loglike_many(nfa_ring *r, int32_t pix, double *cubes, double *lnew, int k) {
    if (!r || !cubes || !lnew || r->slot < 0) return fail(NFA_ERR_ARG, "not an attached ring client");
    RingHeader *h = r->hdr;
    if (k < 1 || k > h->max_points) return fail(NFA_ERR_ARG, "more points than the ring's slots hold (nfa_ring_create_multi)");
    RingSlot *s = ring_slot(r, r->slot);
    if (h->stop.load(std::memory_order_acquire)) return fail(NFA_ERR_STATE, "ring stopped");
    const size_t nd = (size_t)h->ndim;
    memcpy(slot_cube(h, s), cubes, sizeof(double) * nd * (size_t)k);
    s->pix = pix;
    s->n_points = k;
    s->state.store(RING_POSTED, std::memory_order_release);
    h->posts.fetch_add(1);                                     // (sequentially consistent with the servers' flag)
    if (h->servers_asleep.load() != 0) ring_futex(&h->posts, FUTEX_WAKE, INT32_MAX, nullptr);     // every sleeping server looks
    // a launch takes tens of microseconds: spin first -- unless there are more sampler processes than cores,
    // where a spinning process only keeps another one from posting
    static const long n_cpu = sysconf(_SC_NPROCESSORS_ONLN);
    const int spin_limit = (long)h->n_attached.load(std::memory_order_relaxed) + 2 <= n_cpu ? 4000 : 50;
    int64_t unserved_since = -1;                               // first wake-up that found no serving loop
    for (int spin = 0;; ++spin) {
        const uint32_t cur = s->state.load(std::memory_order_acquire);
        if (cur == RING_DONE) break;
        if (h->stop.load(std::memory_order_acquire)) return fail(NFA_ERR_STATE, "ring stopped");
        if (spin < spin_limit) { ring_pause(); continue; }
        const timespec ts = {0, 2000000};                      // then sleep on the slot (2 ms: re-check `stop` and the server)
        s->asleep.store(1);                                    // before the kernel re-reads `state`: the server either
        ring_futex(&s->state, FUTEX_WAIT, cur, &ts);           // sees the flag or has changed `state` already
        s->asleep.store(0);
        // is anybody going to answer?  the creator's death ends the ring; no serving loop for a while does too
        const char *dead = nullptr;
        if (ring_pid_gone(h->server_pid)) dead = "the ring's server process is gone";
        else if (h->n_servers.load(std::memory_order_acquire) == 0 && s->state.load(std::memory_order_acquire) != RING_CLAIMED) {
            // nobody inside nfa_ring_serve: a server built on nfa_ring_poll / nfa_ring_complete shows through the
            // heartbeat those calls leave (it may take as long as it likes over a batch it has CLAIMED: a claimed
            // request is never taken back while the ring's creator lives)
            // (a ring nobody has served yet belongs to a server that is still starting -- its process lives, that was
            // checked above --: twelve grace periods for that)
            const int64_t now = ring_now_us();
            const int64_t beat = h->last_serve_us.load(std::memory_order_acquire);
            const int64_t grace = ring_grace_us() * (beat == 0 ? 12 : 1);
            if (beat != 0 && now - beat <= grace) unserved_since = -1;
            else if (unserved_since < 0) unserved_since = now;
            else if (now - unserved_since > grace) dead = "no serving loop on the ring";
        } else unserved_since = -1;
        if (dead && s->state.load(std::memory_order_acquire) != RING_DONE) {
            uint32_t posted = RING_POSTED;                     // take the request back unless a server holds it
            s->state.compare_exchange_strong(posted, RING_FREE, std::memory_order_acq_rel);
            s->gen.fetch_add(1, std::memory_order_acq_rel);    // a late delivery for it is dropped
            s->state.store(RING_FREE, std::memory_order_release);
            return fail(NFA_ERR_STATE, dead);
        }
    }
    const int rc = s->rc;
    if (rc == NFA_OK) memcpy(cubes, slot_cube(h, s), sizeof(double) * nd * (size_t)k);
    for (int i = 0; i < k; ++i) lnew[i] = rc == NFA_OK ? slot_lnl(s)[i] : NAN;
    s->state.store(RING_FREE, std::memory_order_release);
    return rc == NFA_OK ? NFA_OK : fail(rc, "the served batch failed");
}
int nfa_ring_loglike(nfa_ring *r, int32_t pix, double *cube, double *lnew) {
    return nfa_ring_loglike_many(r, pix, cube, lnew, 1);
}

// MultiNest's LogLike signature (cmultinest.pxd:27-28); context = nfa_ring_client*.  No error channel: NaN.
#ifdef NFA_RING_STANDALONE
typedef struct { nfa_ring *ring; int32_t pix; } nfa_ring_client;       // the engine build takes it from nestfit_amd.h
#endif
void nfa_ring_callback(double *Cube, int *ndim, int *npars, double *lnew, void *ctx) {
    (void)npars;
    nfa_ring_client *c = (nfa_ring_client *)ctx;
    if (!c || !c->ring || !Cube || !lnew || !ndim || *ndim != c->ring->hdr->ndim) {
        if (lnew) *lnew = NAN;
        return;
    }
    if (nfa_ring_loglike(c->ring, c->pix, Cube, lnew) != NFA_OK) *lnew = NAN;
}

// Server: gather posted slots (each one claimed with a compare-and-swap: several serving loops, each with a
// runner of its own, may poll one ring).  Returns as soon as this server holds its share of the attached clients'
// requests (all of them for a lone server; at most max_batch POINTS), or -- once it holds at least one -- after
// max_wait_us; with nothing posted it sleeps until a post, `stop`, or idle_ms have passed.
// Row k of the *n gathered: slots[k] / pix[k] / U[k * ndim ...]; the points of one request are neighbouring rows
// with the same slot.  (*n == 0: nothing arrived in idle_ms, or the ring was stopped: *stopped says which.)
int nfa_ring_poll(nfa_ring *r, int max_batch, int64_t max_wait_us, int idle_ms, int32_t *slots, int32_t *pix,
                  double *U, int *n, int *stopped) {
    if (!r || !slots || !pix || !U || !n) return fail(NFA_ERR_ARG, "null argument");
    RingHeader *h = r->hdr;
    if (!r->claim_gen) r->claim_gen = new uint32_t[h->n_slots]();      // a second serving handle (attached, not created)
    if (max_batch < 1) max_batch = 1;
    const int64_t t_idle = ring_now_us() + (int64_t)idle_ms * 1000;
    int64_t t_first = -1;
    *n = 0;
    if (stopped) *stopped = 0;
    int rows = 0, claimed = 0;                                 // points and requests this call has claimed so far
    for (int spin = 0;; ++spin) {
        if (h->stop.load(std::memory_order_acquire)) { if (stopped) *stopped = 1; return NFA_OK; }
        if ((spin & 1023) == 0) h->last_serve_us.store(ring_now_us(), std::memory_order_release);     // heartbeat
        const uint32_t posts = h->posts.load(std::memory_order_acquire);
        for (int k = 0; k < h->n_slots && rows < max_batch; ++k) {
            uint32_t posted = RING_POSTED;
            RingSlot *s = ring_slot(r, k);
            if (s->state.load(std::memory_order_acquire) != RING_POSTED) continue;
            const int np = s->n_points;
            if (np < 1 || np > h->max_points || rows + np > max_batch) continue;      // (the next round takes it)
            const uint32_t gen = s->gen.load(std::memory_order_acquire);
            if (!s->state.compare_exchange_strong(posted, RING_CLAIMED, std::memory_order_acq_rel)) continue;
            r->claim_gen[k] = gen;
            for (int i = 0; i < np; ++i) slots[rows + i] = k;
            rows += np;
            claimed += 1;
        }
        const int64_t now = ring_now_us();
        if (rows > 0) {
            if (t_first < 0) t_first = now;
            const int attached = (int)h->n_attached.load(std::memory_order_acquire);
            const int servers = (int)h->n_servers.load(std::memory_order_acquire);
            const int share = servers > 1 ? (attached + servers - 1) / servers : attached;
            if ((share > 0 && claimed >= share) || rows + 1 > max_batch || now - t_first >= max_wait_us) {
                for (int k = 0; k < rows;) {
                    RingSlot *s = ring_slot(r, slots[k]);
                    const int np = s->n_points;
                    memcpy(U + (size_t)k * h->ndim, slot_cube(h, s), sizeof(double) * (size_t)h->ndim * (size_t)np);
                    for (int i = 0; i < np; ++i) pix[k + i] = s->pix;
                    k += np;
                }
                *n = rows;
                return NFA_OK;
            }
            ring_pause();                                      // company is microseconds away: spin
            continue;
        }
        if (now >= t_idle) return NFA_OK;
        if (spin < 2000) { ring_pause(); continue; }
        const timespec ts = {0, 1000000};
        h->servers_asleep.fetch_add(1);
        ring_futex(&h->posts, FUTEX_WAIT, posts, &ts);         // sleeps only if nothing was posted since the scan
        h->servers_asleep.fetch_sub(1);
    }
}

// Server: results of rows nfa_ring_poll handed out (whole requests: the neighbouring rows of a slot; rc != 0: those
// requests failed, their clients get NaN and the code).  A request whose slot changed hands since it was claimed (its
// client died and another process inherited the slot) is dropped.
int nfa_ring_complete(nfa_ring *r, int n, const int32_t *slots, const double *U, const double *lnL, int rc) {
    if (!r || n < 0 || (n > 0 && (!slots || !U || !lnL))) return fail(NFA_ERR_ARG, "null argument");
    RingHeader *h = r->hdr;
    h->last_serve_us.store(ring_now_us(), std::memory_order_release);          // heartbeat
    for (int k = 0; k < n;) {
        const int sl = slots[k];
        int np = 1;
        while (k + np < n && slots[k + np] == sl) ++np;
        if (sl < 0 || sl >= h->n_slots) return fail(NFA_ERR_ARG, "slot index out of range");
        RingSlot *s = ring_slot(r, sl);
        const bool mine = r->claim_gen && s->gen.load(std::memory_order_acquire) == r->claim_gen[sl] &&
                          s->state.load(std::memory_order_acquire) == RING_CLAIMED && np == s->n_points;
        if (mine) {
            if (rc == NFA_OK) memcpy(slot_cube(h, s), U + (size_t)k * h->ndim, sizeof(double) * (size_t)h->ndim * (size_t)np);
            for (int i = 0; i < np; ++i) slot_lnl(s)[i] = rc == NFA_OK ? lnL[k + i] : NAN;
            s->rc = rc;
            uint32_t held = RING_CLAIMED;
            if (s->state.compare_exchange_strong(held, RING_DONE) && s->asleep.load() != 0)
                ring_futex(&s->state, FUTEX_WAKE, 1, nullptr);
        }
        k += np;
    }
    if (n > 0) {
        h->n_batches.fetch_add(1, std::memory_order_relaxed);
        h->n_evals.fetch_add((uint64_t)n, std::memory_order_relaxed);
        if ((uint64_t)n > h->max_batch_seen.load(std::memory_order_relaxed)) h->max_batch_seen.store((uint64_t)n, std::memory_order_relaxed);
    }
    return NFA_OK;
}

// out[0] batches served, out[1] evaluations served, out[2] largest batch, out[3] clients attached
int nfa_ring_stats(nfa_ring *r, int64_t *out) {
    if (!r || !out) return fail(NFA_ERR_ARG, "null argument");
    out[0] = (int64_t)r->hdr->n_batches.load(); out[1] = (int64_t)r->hdr->n_evals.load();
    out[2] = (int64_t)r->hdr->max_batch_seen.load(); out[3] = (int64_t)r->hdr->n_attached.load();
    return NFA_OK;
}

#ifndef NFA_RING_STANDALONE
// Server loop of the engine: poll -> nfa_runner_loglike_batch (up to 128 points: one point-kernel launch; more: the
// batch kernels reading the unit cubes from, and writing theta and lnL to, buffers the device addresses itself)
// -> complete, until the ring is stopped, max_batches (> 0) have been served or nothing has arrived for
// idle_ms.  The runner must not be used by anyone else meanwhile.
// A request with a pixel index outside the runner's cube fails alone (its client gets NFA_ERR_ARG); the loop goes on.
// A device error ends the ring: the batch's clients get the code, everybody else NFA_ERR_STATE (nfa_ring_stop).
int nfa_ring_serve(nfa_ring *r, nfa_runner *run, int64_t max_wait_us, int64_t max_batches, int idle_ms) {
    if (!r || !run) return fail(NFA_ERR_ARG, "null argument");
    if (run->ndim != r->hdr->ndim) return fail(NFA_ERR_ARG, "ring and runner disagree on ndim");
    const int ndim = run->ndim;
    const int cap = NFA_RING_MAXBATCH;
    std::vector<int32_t> slots(cap), good_slots(cap);
    // unit cubes, theta, lnL and pixel indices in memory the device addresses itself: no staging copies for batches
    // beyond the point kernel's 128 (nfa_runner_loglike_batch recognises such buffers)
    void *h_buf = nullptr;
    const size_t n_dbl = (size_t)cap * (ndim + 1);
    if (nfa_host_alloc(&h_buf, sizeof(double) * n_dbl + sizeof(int32_t) * (size_t)cap) != NFA_OK) return NFA_ERR_DEVICE;
    double *U = (double *)h_buf, *lnL = U + (size_t)cap * ndim;
    int32_t *pix = (int32_t *)(lnL + cap);
    const int64_t n_pix = run->ss->n_pix;
    r->hdr->n_servers.fetch_add(1, std::memory_order_acq_rel);
    int rc_out = NFA_OK;
    for (int64_t served = 0; max_batches <= 0 || served < max_batches;) {
        int n = 0, stopped = 0;
        int rc = nfa_ring_poll(r, cap, max_wait_us, idle_ms, slots.data(), pix, U, &n, &stopped);
        if (rc != NFA_OK) { rc_out = rc; break; }
        if (stopped || n == 0) break;
        // requests that name a pixel the runner does not have fail alone; the rest close ranks
        int m = 0;
        bool any_pix = false;
        for (int k = 0; k < n;) {
            int np = 1;
            while (k + np < n && slots[k + np] == slots[k]) ++np;
            if ((int64_t)pix[k] >= n_pix) {
                nfa_ring_complete(r, np, slots.data() + k, U + (size_t)k * ndim, lnL + k, NFA_ERR_ARG);
            } else {
                any_pix |= pix[k] >= 0;
                if (m != k) {
                    memmove(U + (size_t)m * ndim, U + (size_t)k * ndim, sizeof(double) * (size_t)ndim * (size_t)np);
                    for (int i = 0; i < np; ++i) pix[m + i] = pix[k + i];
                }
                for (int i = 0; i < np; ++i) { good_slots[m + i] = slots[k]; if (pix[m + i] < 0) pix[m + i] = 0; }
                m += np;
            }
            k += np;
        }
        if (m > 0) {
            rc = nfa_runner_loglike_batch(run, any_pix ? pix : nullptr, U, lnL, m);
            nfa_ring_complete(r, m, good_slots.data(), U, lnL, rc);
            if (rc == NFA_ERR_DEVICE) { rc_out = rc; nfa_ring_stop(r); break; }     // nobody will be served any more
        }
        ++served;
    }
    r->hdr->n_servers.fetch_sub(1, std::memory_order_acq_rel);
    (void)nfa_host_free(h_buf);
    return rc_out;
}
#endif

}  // extern "C"
