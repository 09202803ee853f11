// nfa_ring_serve.h -- the ring served by a resident kernel (engine library only).
//
// nfa_ring_serve (nfa_ring.h) gathers the posted points on the host and launches one point_kernel per round: the
// round trip of a serial sampler's LogLike (nestfit/core/core.pyx:622-624; one call per point, cmultinest.pxd:27-28)
// is then a launch, its dispatch and two trips over the bus -- 60 us for a round of 14 points, 217-266 k
// evaluations/s from 14 sampler processes where the host's 16 cores deliver 378 k (profiles/r03/ring.txt).
// nfa_ring_serve_device keeps a kernel RESIDENT instead: the ring's shared-memory mapping is registered with the
// runtime, the workgroups of ring_serve_kernel poll the slots' state words themselves (system-scope loads over the
// bus), claim a posted point with a compare-and-swap, run the whole path -- set-up stage, likelihood waves, sum: the
// point kernel's own device functions, so a point gives the same bits on either route -- and write theta, lnL and
// the DONE state straight into the slot the client spins on.  No launch, no host thread in the round trip.
//
// A kernel that never ends is a hung GPU for whoever comes next, so this one always ends: every instance lives at
// most `lifetime_ms` (wall_clock64 against its start), ends at once when the ring's stop word is set, and is launched
// again by the host loop as long as there is anything to serve.  The host loop meanwhile does what a kernel cannot:
// the heartbeat the clients' liveness test looks at, futex wake-ups for clients that went to sleep on their slot,
// the ring's statistics, and the end of serving after `idle_ms` without a request.
#pragma once

struct RingServeArgs {
    unsigned char *base;                 // the ring's mapping as the device addresses it
    unsigned long long slot0, stride;    // byte offset of slot 0, bytes per slot
    unsigned long long stop_off;         // byte offset of the header's stop word
    int n_slots, ndim, max_points, n_blocks;
    long n_pix;                          // pixels of the runner's cube (a request beyond fails alone)
    int has_pix;                         // the runner is a cube runner (pixel indices mean something)
    unsigned long long lifetime_ticks;   // wall_clock64 ticks (100 MHz) this instance may live
    unsigned long long *counters;        // device: [0] points served, [1] requests refused (bad pixel)
    int ctl_double;                      // index of the workgroup's control words inside its dynamic LDS
    int pause;                           // s_sleep(8) units (~0.2 us each) the workgroup sleeps after an empty turn
};

// field offsets inside a RingSlot (nfa_ring.h): state, owner, asleep, gen, pix, rc, n_points, pad, data[]
#define RS_STATE 0
#define RS_GEN 12
#define RS_PIX 16
#define RS_RC 20
#define RS_NPTS 24
#define RS_DATA 32

__device__ __forceinline__ unsigned rs_load_u32(const unsigned char *p) {
    return __hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double rs_load_f64(const unsigned char *p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __longlong_as_double((long long)v);
}
__device__ __forceinline__ void rs_store_f64(unsigned char *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One workgroup of POINT_THREADS threads serves the slots k = blockIdx.x, blockIdx.x + gridDim.x, ...
template <int MODE, int NCOMP>
__global__ void __launch_bounds__(POINT_THREADS) ring_serve_kernel(const PriorProg *__restrict__ ppp, SpecDev S, RingServeArgs A,
                                                                   int *__restrict__ d_pix, double *__restrict__ U_all,
                                                                   double *__restrict__ D_all, double *__restrict__ part_all,
                                                                   LnlGeom G, const double *__restrict__ g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int SMODE = MODE == 0 ? 0 : 1;
    int n_shared;
    const double *sm = stage_exp_tables<SMODE>(smem, g_tabs, &n_shared);          // once per instance, not per point
    // ... and so are the prior program and its tables (round 5); the likelihood's line tables go behind them
    const int n_lines = n_shared + setup_stage_priors(ppp, S, smem, n_shared);
    volatile int *ctl = (volatile int *)(smem + A.ctl_double);                     // [0] command, [1] slot
    const int tid = threadIdx.x, ndim = A.ndim;
    const long wg = blockIdx.x;
    double *U = U_all + wg * ndim, *D = D_all + wg * drec_size(S.ncomp, S.n_spec), *part = part_all + wg * S.n_spec;
    const unsigned long long t0 = wall_clock64();
    int next = (int)wg;                                                           // round robin over this workgroup's slots
    unsigned turn = 0;
    for (;;) {
        // One turn: wave 0 looks at ONE slot -- lane 0 reads its state word, a trip over the bus; a posted point then
        // costs the claim (below) and one more trip for everything the request holds: lane j reads coordinate j, the last two
        // lanes the pixel and the point count, all in flight together.  Then the whole workgroup meets at the barrier and,
        // with nothing posted, sleeps a microsecond.
        // (A polling wave that spins by itself while the others wait at the barrier looks cheaper and is not: 14
        // workgroups served 164 k points/s that way against 390 k with a barrier per turn -- flags left in the
        // experiment's place, profiles/r04/ring.txt.)  The stop word and the clock are looked at every 16th turn.
        if (tid < 64) {
            int cmd = 0, found = -1;
            if ((turn & 15u) == 0u) {
                unsigned stop = 0u;
                if (tid == 0) stop = rs_load_u32(A.base + A.stop_off);
                stop = __builtin_amdgcn_readfirstlane(stop);
                if (stop != 0u || wall_clock64() - t0 > A.lifetime_ticks) cmd = 2;
            }
            if (cmd == 0) {
                const int k = next;
                next += (int)gridDim.x;
                if (next >= A.n_slots) next = (int)wg;
                unsigned char *s = A.base + A.slot0 + (unsigned long long)k * A.stride;
                unsigned state = 0;
                if (tid == 0) state = rs_load_u32(s + RS_STATE);            // (one lane: 64 lanes at system scope are 64 trips)
                state = __builtin_amdgcn_readfirstlane(state);
                // A posted point is claimed by compare-and-swap, like every other server of a ring claims (nfa_ring_serve's
                // threads, a second resident kernel, nfa_ring_poll: nothing makes this workgroup the only server of its slots,
                // and the client may take its request back -- POSTED -> FREE -- at any moment): one more trip over the bus, on
                // a hit only.  The request is read behind the successful claim (acquire).
                if (state == (unsigned)RING_POSTED) {
                    unsigned got = 0u;
                    if (tid == 0) {
                        unsigned expect = (unsigned)RING_POSTED;
                        got = __hip_atomic_compare_exchange_strong((unsigned *)(s + RS_STATE), &expect, (unsigned)RING_CLAIMED, __ATOMIC_ACQUIRE,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 1u : 0u;
                    }
                    state = __builtin_amdgcn_readfirstlane(got) ? (unsigned)RING_POSTED : (unsigned)RING_FREE;
                }
                if (state == (unsigned)RING_POSTED) {
                    const unsigned char *s_cube = s + RS_DATA + 8ull * A.max_points;
                    double v = 0.0;
                    unsigned w = 0;
                    if (tid < ndim) v = rs_load_f64(s_cube + 8ull * tid);
                    if (tid == 62) w = rs_load_u32(s + RS_PIX);
                    if (tid == 63) w = rs_load_u32(s + RS_NPTS);
                    if (tid < ndim) U[tid] = v;
                    if (tid == 62) ctl[2] = (int)w;
                    if (tid == 63) ctl[3] = (int)w;
                    cmd = 1; found = k;
                }
            }
            if (tid == 0) { ctl[0] = cmd; ctl[1] = found; }
        }
        turn += 1;
        __syncthreads();
        const int cmd = ctl[0], k = ctl[1];
        const int pixv = ctl[2], npts = ctl[3];
        __syncthreads();                                        // (wave 0 rewrites the words in its next turn)
        if (cmd == 2) break;
        if (cmd == 0) { for (int z = 0; z < A.pause; ++z) __builtin_amdgcn_s_sleep(8); continue; }
        unsigned char *s = A.base + A.slot0 + (unsigned long long)k * A.stride;
        unsigned char *s_lnl = s + RS_DATA, *s_cube = s + RS_DATA + 8ull * A.max_points;
        const bool bad = npts != 1 || (long)pixv >= A.n_pix;      // (requests of several points belong to nfa_ring_serve)
        int my_pix = pixv < 0 ? 0 : pixv;
        if (!bad) {
            if (tid == 0 && A.has_pix) d_pix[wg] = my_pix;
            __syncthreads();                                    // (the unit cube wave 0 wrote to U: the barrier's workgroup-scope release)
            setup_body<SMODE, MODE == 2, 1, true>(ppp, S, U, D, 1, 1, g_tabs, 0, smem, sm, n_shared, 0u);
            __threadfence();                                    // theta in U, the derived record in D: at L2, stale lines of the last point gone
            __syncthreads();
            __builtin_amdgcn_s_dcache_inv();
            const int *pix = A.has_pix ? d_pix + wg : nullptr;
            for (int blk = 0; blk < A.n_blocks; ++blk) {
                if (blk) __syncthreads();
                // (the line tables go BEHIND the staged exponential table in every mode: the point kernel lets the fast
                // mode's waves overwrite it, but here the next point's set-up stage wants it again)
                lnl_body<MODE, false, false, NCOMP>(S, pix, D, part, nullptr, 1, G, g_tabs, smem, sm, n_lines, (unsigned)blk);
            }
            __threadfence();
            __syncthreads();
            if (tid < ndim) rs_store_f64(s_cube + 8ull * tid, U[tid]);
        }
        __syncthreads();
        if (tid == 0) {
            double lnl = NAN;
            if (!bad) lnl = lnl_of_item(part, S.noise, A.has_pix ? (long)my_pix : 0, 0, S.n_spec);
            rs_store_f64(s_lnl, lnl);
            __hip_atomic_store((unsigned *)(s + RS_RC), bad ? (unsigned)NFA_ERR_ARG : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            // delivered only while the slot is still the claimer's: a slot that changed hands meanwhile (its client died,
            // another process inherited it) is FREE or POSTED again, never CLAIMED -- this workgroup alone claims it
            unsigned held = RING_CLAIMED;
            __hip_atomic_compare_exchange_strong((unsigned *)(s + RS_STATE), &held, (unsigned)RING_DONE, __ATOMIC_ACQ_REL,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            atomicAdd(A.counters + (bad ? 1 : 0), 1ull);
        }
        __syncthreads();
    }
}

template <int MODE, int NCOMP>
static int launch_ring_serve_t(nfa_runner *r, const SpecDev &S, const RingServeArgs &A, const LnlGeom &G, size_t lds, int n_wg) {
    auto kern = ring_serve_kernel<MODE, NCOMP>;
    { int rc = ensure_dynamic_lds((const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(POINT_THREADS), lds, r->lanes[0], (const PriorProg *)r->pr->d_prog, S, A,
                       r->d_pix, r->d_U, r->d_D[0], r->d_part[0], G, (const double *)g_eng.d_tabs);
    HIP_TRY(hipGetLastError());
    return NFA_OK;
}
template <int MODE>
static int launch_ring_serve_n(nfa_runner *r, const SpecDev &S, const RingServeArgs &A, const LnlGeom &G, size_t lds, int n_wg) {
    switch (r->ncomp) {
    case 1: return launch_ring_serve_t<MODE, 1>(r, S, A, G, lds, n_wg);
    case 2: return launch_ring_serve_t<MODE, 2>(r, S, A, G, lds, n_wg);
    case 3: return launch_ring_serve_t<MODE, 3>(r, S, A, G, lds, n_wg);
    default: return launch_ring_serve_t<MODE, 0>(r, S, A, G, lds, n_wg);
    }
}

extern "C" {

// Serve the ring from a resident kernel until the ring is stopped or nothing has been served for idle_ms.  One point per
// slot (nfa_ring_create; rings made with nfa_ring_create_multi are served by nfa_ring_serve).  lifetime_ms: how long one
// kernel instance lives before the host launches the next (1..1000; 0 = 20).  The runner must not be used by anyone
// else meanwhile.  Results are bitwise those of nfa_runner_loglike_batch.
int nfa_ring_serve_device(nfa_ring *ring, nfa_runner *run, int lifetime_ms, int idle_ms) {
    if (!ring || !run) return fail(NFA_ERR_ARG, "null argument");
    if (!run->pr) return fail(NFA_ERR_STATE, "runner has no priors (predict-only)");
    RingHeader *h = ring->hdr;
    if (run->ndim != h->ndim) return fail(NFA_ERR_ARG, "ring and runner disagree on ndim");
    if (h->max_points != 1) return fail(NFA_ERR_ARG, "the resident kernel serves one point per slot: use nfa_ring_serve for this ring");
    if (run->ndim > NFA_POINT_MAXDIM || lnl_wide(run)) return fail(NFA_ERR_ARG, "this runner's points go through the batch kernels: use nfa_ring_serve");
    if (lifetime_ms <= 0) lifetime_ms = 20;
    if (lifetime_ms > 1000) lifetime_ms = 1000;
    RUNNER_LOCK(run);
    { int rc = sync_all_lanes(run); if (rc) return rc; }
    const int mode = run->exp_mode >= 0 ? run->exp_mode : g_eng.exp_mode;
    const SpecDev S = runner_specdev(run);
    LnlGeom G;
    G.ablate = 0;
    G.nhf_max = run->ss->nhf_max;
    G.inv_nspec = S.n_spec == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)S.n_spec) + 1u;
    G.inv_nhf = G.nhf_max == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)G.nhf_max) + 1u;
    G.split = resolve_split(run, S, 1);
    if (G.split > POINT_WAVES) return fail(NFA_ERR_ARG, "spectra too short for the point kernel's split");
    G.wave_doubles = lnl_wave_doubles(run);
    const int upw = POINT_WAVES / G.split;
    const int n_shared = mode == 0 ? (SM_END_TABLE - SM_EXP2) : 0;
    const size_t n_staged = mode == 0 ? (SM_END_TABLE - SM_EXP2) : NFA_EXP2_N;
    // [exponential tables][theta, partition records][prior program + tables: staged once][line tables of the likelihood waves]
    size_t lds = setup_lds_bytes(run, 1, true) + sizeof(double) * (n_staged - NFA_EXP2_N)
                 + sizeof(double) * (((size_t)G.wave_doubles + (G.split > 1 ? LNL_PARTS * 64 : 0)) * upw);
    if (mode == 0) lds = std::max(lds, sizeof(double) * (size_t)(n_shared + SM_TABLE_TAIL));
    lds = (lds + 15) & ~(size_t)15;
    const int ctl_double = (int)(lds / sizeof(double));
    lds += 16;                                                   // the workgroup's control words
    if (lds > 160 * 1024) return fail(NFA_ERR_ARG, "too many parameters for the resident kernel");
    const int n_wg = std::max(1, std::min(h->n_slots, 64));
    { int rc = runner_reserve(run, n_wg, false); if (rc) return rc; }
    { int rc = reserve_lane(run, 0, n_wg); if (rc) return rc; }
    // the ring's mapping as the device sees it
    if (!ring->dev_base) {
        HIP_TRY(hipHostRegister(ring->base, ring->bytes, hipHostRegisterMapped | hipHostRegisterPortable));
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, ring->base, 0) != hipSuccess) { (void)hipHostUnregister(ring->base); (void)hipGetLastError(); return fail(NFA_ERR_DEVICE, "the ring's memory cannot be mapped for the device"); }
        ring->dev_base = dev;
    }
    unsigned long long *d_cnt = nullptr;
    HIP_TRY(hipMalloc((void **)&d_cnt, 2 * sizeof(unsigned long long)));
    if (hipMemset(d_cnt, 0, 2 * sizeof(unsigned long long)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(d_cnt);
        return fail(NFA_ERR_DEVICE, "the resident kernel's counters could not be cleared");
    }
    RingServeArgs A;
    A.base = (unsigned char *)ring->dev_base;
    A.slot0 = sizeof(RingHeader); A.stride = h->slot_stride;
    A.stop_off = (unsigned long long)((unsigned char *)&h->stop - (unsigned char *)h);
    A.n_slots = h->n_slots; A.ndim = h->ndim; A.max_points = h->max_points;
    A.n_blocks = (S.n_spec + upw - 1) / upw;
    A.n_pix = run->ss->n_pix; A.has_pix = run->ss->n_pix > 1 ? 1 : 0;
    A.lifetime_ticks = (unsigned long long)lifetime_ms * 100000ull;          // wall_clock64: 100 MHz
    A.counters = d_cnt;
    A.ctl_double = ctl_double;
    A.pause = getenv("NFA_RING_PAUSE") ? atoi(getenv("NFA_RING_PAUSE")) : 4;     // ~1 us
    h->n_servers.fetch_add(1, std::memory_order_acq_rel);
    int rc_out = NFA_OK;
    unsigned long long served_before = 0, h_cnt[2] = {0, 0};
    int64_t t_last_served = ring_now_us();
    hipStream_t st = run->lanes[0];
    for (;;) {
        if (h->stop.load(std::memory_order_acquire)) break;
        int rc;
        switch (mode) {
        case 0: rc = launch_ring_serve_n<0>(run, S, A, G, lds, n_wg); break;
        default: rc = launch_ring_serve_n<2>(run, S, A, G, lds, n_wg); break;
        }
        if (rc) { rc_out = rc; break; }
        // while the instance lives: heartbeat, wake-ups for clients asleep on a finished slot
        hipError_t q;
        while ((q = hipStreamQuery(st)) == hipErrorNotReady) {
            h->last_serve_us.store(ring_now_us(), std::memory_order_release);
            for (int k = 0; k < h->n_slots; ++k) {
                RingSlot *s = ring_slot(ring, k);
                if (s->asleep.load(std::memory_order_acquire) != 0 && s->state.load(std::memory_order_acquire) == RING_DONE)
                    ring_futex(&s->state, FUTEX_WAKE, 1, nullptr);
            }
            timespec ts = {0, 50000};
            nanosleep(&ts, nullptr);
        }
        if (q != hipSuccess) { (void)hipGetLastError(); rc_out = fail(NFA_ERR_DEVICE, "the resident serving kernel failed"); nfa_ring_stop(ring); break; }
        // (a device error from here on leaves through the common exit below: the ring must not go on advertising a
        // server -- clients only give up on a ring with n_servers == 0 -- and is stopped for everybody)
        if (hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError(); rc_out = fail(NFA_ERR_DEVICE, "the resident kernel's counters could not be read"); nfa_ring_stop(ring); break;
        }
        const unsigned long long served = h_cnt[0] + h_cnt[1];
        const int64_t now = ring_now_us();
        if (served != served_before) {
            h->n_batches.fetch_add(served - served_before, std::memory_order_relaxed);
            h->n_evals.fetch_add(served - served_before, std::memory_order_relaxed);
            if (h->max_batch_seen.load(std::memory_order_relaxed) < 1) h->max_batch_seen.store(1, std::memory_order_relaxed);
            served_before = served;
            t_last_served = now;
        } else if (now - t_last_served > (int64_t)idle_ms * 1000) {
            break;
        }
    }
    h->n_servers.fetch_sub(1, std::memory_order_acq_rel);
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_cnt);
    return rc_out;
}

}  // extern "C"
