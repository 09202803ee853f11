// nfa_sampler.h -- device-resident batched nested sampler.
//
// Stand-in for the serial MultiNest run per pixel of the reference (run_multinest,
// nestfit/core/core.pyx:727-823; pixel loop nestfit/main.py:452-469) when libmultinest is not
// available, laid out for the GPU: every pixel is an independent nested-sampling run, all runs
// advance in lock-step rounds and their whole state (live points, bounding ellipsoids, evidence
// accumulators, dead points) stays in HBM.  One round =
//     ns_propose_kernel   Kr proposals per active pixel from a counter-based random stream (the host
//                         twin in nestfit_amd/sampler.py draws the very same numbers): uniform in the
//                         pixel's bounding ellipsoid (or in the unit cube while that is the smaller
//                         bound), or -- for pixels that have switched to constrained random walks --
//                         one differential-evolution Metropolis step of each of its 64 walkers.
//                         Proposals inside the prior's support are compacted into the rows the
//                         likelihood will see.
//     set-up + lnl kernels the engine's likelihood batch over the compacted rows of all pixels
//     ns_update_kernel    one wave per pixel: scan the candidates in order, every one above the
//                         pixel's current threshold replaces the worst live point (Skilling's
//                         bookkeeping: dead point, ln w, running lnZ), stop test, ellipsoid refit;
//                         walkers accept / reject their step, and at the end of a cycle their end
//                         points are the candidates
// The host counts rounds, reads back how many rows a round has, compacts the list of still-active
// pixels every few rounds and keeps three groups of pixels in flight on three stream lanes.
// Only the unit-cube slots the likelihood depends on are sampled (free_mask).
#pragma once

#define NS_MAXD      60          // 6 parameters x MAXCOMP
#define NS_TAG_LIVE  (1ull << 62)
#define NS_B_RADIUS  255ull
#define NS_B_START   250ull      // stream index of a walker's starting live point
#define NS_WALK_TARGET 0.5     // acceptance the walk scale is tuned to
#define NS_W         64          // walkers per lane slice of the update wave
#define NS_WMAX      256         // walkers per pixel at most
// Walkers of a pixel with n live points: a cycle's walkers are harvested against a threshold that rises with every
// replacement, so many more than a third of n mostly harvest each other's leftovers (of k walkers n ln(1 + k / n) pass);
// 64 of them are a small batch once the pixels are few -- 128 from 384 live points, 256 from 768.
__host__ __device__ inline int ns_walkers_for(int n) { return n >= 768 ? 256 : n >= 384 ? 128 : 64; }

__host__ __device__ inline uint64_t ns_mix(uint64_t x) {             // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// uniform in (0, 1), a pure function of (seed, pixel, a, b)
__host__ __device__ inline double ns_uniform(uint64_t seed, uint64_t p, uint64_t a, uint64_t b) {
    const uint64_t h = ns_mix(ns_mix(ns_mix(ns_mix(seed) + p) + a) + b);
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// the same in two steps: what (seed, pixel, a) contribute is formed once per proposal (three of the four mixing rounds)
__host__ __device__ inline uint64_t ns_stream(uint64_t seed, uint64_t p, uint64_t a) { return ns_mix(ns_mix(ns_mix(seed) + p) + a); }
__host__ __device__ inline double ns_uniform_of(uint64_t stream, uint64_t b) {
    const uint64_t h = ns_mix(stream + b);
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// Several bounding ellipsoids per pixel (MultiNest's `mmodal` bound in its simplest form): up to NS_ME of them where at
// most NS_ME_MAXD dimensions are sampled.  A cluster of live points is cut in two across its principal axis at its
// centre; the cut is kept when the two halves' ellipsoids together have less than NS_ME_GAIN of the parent's volume.
#define NS_ME 4
#ifndef NS_ME_MAXD
#define NS_ME_MAXD 6
#endif
#define NS_ME_GAIN 0.7
static_assert(NS_ME_MAXD <= 6, "ns_refit_kernel dispatches the cluster fits for up to six sampled dimensions");
#define NS_B_ELL 253ull            // random-stream slots of a proposal: which ellipsoid, and the 1 / (number that hold it) test
#define NS_B_KEEP 254ull
// Free rejections of a one-ellipsoid bound: a proposal outside the bounding box of the live points -- in the unit cube's
// axes, in the ellipsoid's own (Cholesky) frame, or in one of NS_FRAMES fixed rotations of that frame -- is dropped before
// its likelihood is evaluated.  Every box holds the live region, so what passes is uniform over the intersection.  A face
// lies beyond the extreme live point by c max(0.1 s, extreme - mean - 1.5 s), s = the spread along the face's direction:
// small where the marginal ends abruptly (a flat direction), large where it thins out (the projection of a round body).
// scripts/proto_intersection.py measured what each family of bounds cuts off the true region and what it saves; the
// numpy twin's _fit_boxes / _box_veto hold the same arithmetic.
#define NS_FRAMES 32               // rotated frames a caller gets who asks for boxes without naming a number
#define NS_FRAMES_MAX 64
#define NS_MARGIN_C 2.5            // (round 4: 1.75, sampler.py precision='speed')
#define NS_MARGIN_A 1.5
#define NS_MARGIN_FLOOR 0.1
#define NS_RATIO_MAX 32            // proposals drawn per round: at most this multiple of the evaluations aimed for
#define NS_FRAME_SEED 0x5EEDF00Dull
// A volume-preserving shear in front of the one-ellipsoid bound (the twin's _fit_shear): every sampled coordinate minus a
// quadratic function of the earlier ones -- the curved tex / ntot ridges of faint pixels come out straight, and an
// ellipsoid around straight things is small
#define NS_SHEAR_RIDGE 1e-6        // on the Gram matrix's diagonal, times the live points
#define NS_SHEAR_ENLARGE 3.0       // safety factor on the enclosing volume of the sheared ellipsoid (round 4: 2.5, sampler.py precision='speed')
#define NS_SHEAR_PIVOT 1e-9        // a Cholesky pivot below this fraction of its diagonal entry: the monomial is dropped
#define NS_SHEAR_MMAX 64           // monomials at most
#define NS_REFIT_THREADS 512       // of the workgroup that fits a one-ellipsoid bound
#define NS_PAIRS_ENLARGE 2.0       // safety factor on the area of a pair ellipse (round 4: 1.75, sampler.py precision='speed')
#define NS_KP_START 256            // a pixel's share of proposals in its first rejection round
#define NS_K_TARGET 16             // replacements per pixel and rejection round the per-pixel share of proposals aims at
#define NS_WALK_LOWD 6             // up to this many sampled dimensions ...
#define NS_WALK_FACTOR_LOWD 64     // ... the switch to walks waits for an acceptance below 1 / (64 n_steps)
#define NS_WALK_FACTOR 2           // above: 1 / (2 n_steps)
struct NsDev {
    int     P, N, D, K;                 // pixels, live points, SAMPLED dimensions, candidates per round (at least)
    int     DT;                         // length of a theta row (all unit-cube slots of the runner)
    const int *fmap;                    // [D] slot of every sampled dimension; the others stay at u = 0.5
    long    cap;                        // dead-point slots per pixel
    double  tol, ln_shrink, ln_efr, ln_enlarge, log_zero;
    long    maxiter;
    int     upd;
    uint64_t seed;
    const int *pixmap;                  // sampler pixel -> cube pixel
    double *Ulive, *Tlive, *Llive;      // [P][N][D], [P][N][DT], [P][N]
    double *centre, *axes;              // [P][NS_ME][D], [P][NS_ME][D][D] (lower triangular, scaled)
    double *elnv;                       // [P][NS_ME] ln volume of each ellipsoid
    int    *nell;                       // [P] ellipsoids in use
    int     multi;                      // 1: the bound may be split (D <= NS_ME_MAXD and the live points fit in LDS)
    int     max_ell;                    // ... into this many ellipsoids at most (2..NS_ME)
    long   *n_iter, *n_evals;           // [P]
    long   *cand_base;                  // [P] candidates drawn so far (index into the pixel's random stream)
    double *lnZ;                        // [P] running evidence of the dead points
    int    *active, *since_fit;         // [P]
    int    *refit_due;                  // [P] set by the update wave, cleared by the refit wave
    int    *use_cube;                   // [P] 1: the ellipsoid is larger than the unit cube, draw from the cube
    double  ln_vball;                   // ln volume of the unit D-ball
    double *deadT, *deadL, *deadlnw;    // [P][cap][DT], [P][cap], [P][cap]
    double *candU, *candT, *candL;      // proposals [rows][D]; compact: theta [rows][DT], lnL [rows]
    int    *candpix, *valid;            // [rows]: pixel of a compact row; validity of a proposal
    // the update wave sums a row's per-spectrum chi^2 parts itself (lnl_of_item's arithmetic): no summing launch per round
    const double *part, *noise;         // [rows][nspec] of the round's batch (null: candL holds lnL); [pixels][nspec]
    int     nspec;
    int    *slot;                       // [rows] compact row of a valid proposal
    int    *count;                      // number of compact rows filled in this round
    // A one-thread launch behind the proposing one writes that number and the round's sequence number into host memory
    // the device can address and zeroes the counter for the next round (ns_publish_kernel)
    int    *host_rows;                  // mapped host memory: rows of the round
    unsigned long long *host_seq;       // mapped host memory: sequence number of the round, written last
    const int *actlist;                 // [n_act] active pixels of this round
    // constrained random walks (pixels whose rejection sampling has become too inefficient)
    int     method, n_steps;            // 0 rejection only, 1 automatic switch, 2 walks from the start
    int    *walk, *wstep, *wW;          // [P] mode flag, step inside the current cycle, walkers in it
    double *wscale, *wLthr;             // [P] proposal scale, threshold frozen at the cycle start
    long   *wacc_sum, *wtot_sum;        // [P] accepted / evaluated steps of the current cycle
    double *wU, *wT, *wL;               // walker states [P][w_stride][D], [P][w_stride][DT], [P][w_stride]
    int    *wnacc;                      // [P][w_stride] accepted steps of each walker in the cycle
    int     w_stride;                   // walker slots per pixel (a multiple of 64)
    int     w_fixed;                    // > 0: this many walkers whatever the live points (A/B knob), else ns_walkers_for
    double *lnvol;                      // [P] ln volume of the bounding ellipsoid (last refit)
    int     stage_live;                 // the refit stages the centred live points in LDS
    int     refit_every;                // rejection-mode pixels refit in rounds that are multiples of this
    int     walk_factor;                // to walks below an acceptance of 1 / (walk_factor n_steps), back above 8 times that
    // Pixels of one lock-step group may run with different numbers of live points (the cube driver gives every pixel
    // nlive + int(5 SNR), main.py:445-447): N is then the stride of the live arrays and the largest count, and these
    // hold each pixel's own count, dead-point slots and refit interval (null: N, cap, upd for everybody)
    const int  *nlive;                  // [P]
    const long *capp;                   // [P]
    const int  *updp;                   // [P]
    // free rejections by boxes (one-ellipsoid bounds whose live points are staged in LDS)
    int     boxes, n_frames;            // on / off; rotated frames beside the unit cube's axes and the ellipsoid's own
    double  margin_c;
    const double *frames;               // [n_frames][D][D]: column b of frame k = direction of its coordinate b
    double *ubox;                       // [P][D][2] box in the unit cube's axes
    double *fbox;                       // [P][n_frames + 1][D][2] boxes around zz = A^-1 (u - c): frame 0 = the Cholesky frame itself
    // what a pixel's rejection rounds did since the last decision point (every n_steps rounds): candidates scanned and
    // accepted, proposals drawn and evaluated; ln of the last window's evaluated / drawn when the pixel turned to walks
    long   *rj_scan, *rj_acc, *rj_raw, *rj_val;   // [P]
    double *ln_pass;                    // [P]
    // the shear (one-ellipsoid bounds of 10 or 15 sampled dimensions, component of dimension j = j % (D / 5))
    int     shear, sh_M;                // on / off; monomials
    double  ln_enlarge_shear;
    const int *sh_mono;                 // [sh_M][2] factors of monomial m (-1: the factor 1)
    const int *sh_start;                // [D] monomials before coordinate j's own = the features z_j is regressed on
    double *sh_mu, *sh_sg;              // [P][D] z = (u - mu) / sg
    double *sh_beta;                    // [P][D][sh_M] w_j = z_j - phi(z_<j) . beta_j
    // pair ellipses (with the shear and the boxes): every pair of sheared coordinates has the bounding ellipse of the live
    // points' projection as one more free veto
    int     pairs;
    double  pairs_enlarge;              // safety factor on an ellipse's area
    double *pair_tab;                   // [P][D (D - 1) / 2][5] c_i, c_j, 1 / L00, L10, 1 / L11
    // proposals per pixel: a pixel whose rejection rounds accept far more than k_target candidates halves its share of the
    // next round, one that accepts far fewer doubles it (up to the round's Kr): the round's longest update workgroup is the
    // pixel with the most replacements, and a bound that has seen 100 of them in a round is stale
    int    *Kp;                         // [P] 0 = the round's Kr
    int     k_target;                   // 0 = everybody gets Kr
    long   *dbg;                        // NFA_NS_TIMING=1: stage times of the update workgroup of the first listed pixel (100 MHz ticks)
};
#define NS_TICK(slot) do { if (timing) { const long t_ = (long)wall_clock64(); S.dbg[slot] += t_ - t_last; t_last = t_; } } while (0)

// The monomials of the shear (host; the twin's _shear_monomials): [1], then per coordinate j its own z_j, z_j^2 and
// z_k z_j for the earlier coordinates k of the same velocity component (k % nc == j % nc).
static void ns_shear_monomials(int D, int nc, std::vector<int> &mono, std::vector<int> &start) {
    mono.assign({-1, -1});
    start.clear();
    for (int j = 0; j < D; ++j) {
        start.push_back((int)mono.size() / 2);
        mono.push_back(j); mono.push_back(-1);
        mono.push_back(j); mono.push_back(j);
        for (int k = 0; k < j; ++k)
            if (k % nc == j % nc) { mono.push_back(k); mono.push_back(j); }
    }
    mono.resize((size_t)(start.back() + 1) * 2);      // the last coordinate is nobody's feature
}

// The fixed frames (host; the twin's _frames): entries 2 u - 1 from the counter-based stream, columns orthonormalised one
// after the other (modified Gram-Schmidt).
static void ns_make_frames(int D, int K, std::vector<double> &Q) {
    Q.assign((size_t)K * D * D, 0.0);
    std::vector<double> v((size_t)D);
    for (int k = 0; k < K; ++k) {
        double *q = Q.data() + (size_t)k * D * D;
        for (int b = 0; b < D; ++b) {
            for (int a = 0; a < D; ++a) v[a] = 2.0 * ns_uniform(NS_FRAME_SEED, (uint64_t)(k + 1), (uint64_t)a, (uint64_t)b) - 1.0;
            for (int c = 0; c < b; ++c) {
                double dot = 0.0;
                for (int a = 0; a < D; ++a) dot += q[a * D + c] * v[a];
                for (int a = 0; a < D; ++a) v[a] -= dot * q[a * D + c];
            }
            double n2 = 0.0;
            for (int a = 0; a < D; ++a) n2 += v[a] * v[a];
            const double inv = 1.0 / sqrt(n2);
            for (int a = 0; a < D; ++a) q[a * D + b] = v[a] / sqrt(n2);
            (void)inv;
        }
    }
}
__device__ __forceinline__ int  ns_n(const NsDev &S, int p)   { return S.nlive ? S.nlive[p] : S.N; }
__device__ __forceinline__ long ns_cap(const NsDev &S, int p) { return S.capp ? S.capp[p] : S.cap; }
__device__ __forceinline__ int  ns_upd(const NsDev &S, int p) { return S.updp ? S.updp[p] : S.upd; }
__device__ __forceinline__ int  ns_kp(const NsDev &S, int p, int Kr) { const int k = S.k_target > 0 ? S.Kp[p] : 0; return k > 0 ? min(k, Kr) : Kr; }
__device__ __forceinline__ int  ns_wmax(const NsDev &S, int p) { return min(S.w_stride, S.w_fixed > 0 ? S.w_fixed : ns_walkers_for(ns_n(S, p))); }

// ---- live points -------------------------------------------------------------------------
__global__ void ns_init_live_kernel(NsDev S, int *__restrict__ livepix) {
    const long pi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= (long)S.P * S.N) return;
    const long p = pi / S.N;
    const long i = pi - p * S.N;
    double *u = S.Ulive + pi * S.D, *t = S.Tlive + pi * S.DT;
    for (int j = 0; j < S.DT; ++j) t[j] = 0.5;          // slots the likelihood does not depend on
    for (int j = 0; j < S.D; ++j) {
        const double v = ns_uniform(S.seed, (uint64_t)p, NS_TAG_LIVE + (uint64_t)i, (uint64_t)j);
        u[j] = v;
        t[S.fmap[j]] = v;
    }
    livepix[pi] = S.pixmap[p];
}

__global__ void ns_sanitize_kernel(double *__restrict__ L, long n, double log_zero) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n) { const double v = L[gid]; L[gid] = isfinite(v) ? v : log_zero; }
}

// Is the proposal x (zz = its coordinates in the ellipsoid's frame, A^-1 (x - c)) inside every box of pixel p?  The tests
// in order of their price: the unit cube's axes, the ellipsoid's frame, then the rotated frames, each a D x D product
// that the first failure cuts short.
// (in two parts: the axis boxes cost 4 D comparisons, the rotated frames D x D products each -- where a workgroup can, it
// packs the survivors of the first part, and of the unit cube's test, before it pays for the second)
template <int DD>
__device__ __forceinline__ bool ns_in_axis_boxes(const NsDev &S, int p, const double *x, const double *zz) {
    const int D = DD > 0 ? DD : S.D;
    // (the bound is read through the constant address space: written by the refit launch, not by this one, and the same
    // for the whole workgroup -- scalar loads; as ordinary global loads the compiler sent every one of the 110 reads per
    // frame through the vector memory pipe, and a wave of proposals took 90 us)
    const k_dbl_p ub = (k_dbl_p)(S.ubox + (long)p * D * 2);
    bool ok = true;
    for (int j = 0; j < D; ++j) ok = ok && (x[j] >= ub[2 * j]) && (x[j] <= ub[2 * j + 1]);
    if (!ok) return false;
    const k_dbl_p fb = (k_dbl_p)(S.fbox + (long)p * (S.n_frames + 1) * D * 2);
    for (int j = 0; j < D; ++j) ok = ok && (zz[j] >= fb[2 * j]) && (zz[j] <= fb[2 * j + 1]);
    return ok;
}
// inside every pair ellipse?  x = the proposal in the sheared coordinates
template <int DD>
__device__ __forceinline__ bool ns_in_pairs(const NsDev &S, int p, const double *x) {
    const k_dbl_p pt = (k_dbl_p)(S.pair_tab + (long)p * (DD * (DD - 1) / 2) * 5);
    bool ok = true;
    int e = 0;
#pragma unroll
    for (int j = 1; j < DD; ++j)
#pragma unroll
        for (int i = 0; i < j; ++i) {
            const double y0 = (x[i] - pt[5 * e]) * pt[5 * e + 2];         // (the table holds 1 / L00, L10, 1 / L11: no divisions here)
            const double y1 = ((x[j] - pt[5 * e + 1]) - pt[5 * e + 3] * y0) * pt[5 * e + 4];
            ok = ok && (y0 * y0 + y1 * y1 <= 1.0);
            ++e;
        }
    return ok;
}
template <int DD>
__device__ __forceinline__ bool ns_in_frames(const NsDev &S, int p, const double *zz) {
    const int D = DD > 0 ? DD : S.D;
    bool ok = true;
    k_dbl_p fb = (k_dbl_p)(S.fbox + (long)p * (S.n_frames + 1) * D * 2);
    for (int k = 1; k <= S.n_frames && ok; ++k) {
        const k_dbl_p Q = (k_dbl_p)(S.frames + (long)(k - 1) * D * D);
        fb += D * 2;
        for (int j = 0; j < D; ++j) {
            double w = 0.0;
            for (int a = 0; a < D; ++a) w += zz[a] * Q[a * D + j];
            ok = ok && (w >= fb[2 * j]) && (w <= fb[2 * j + 1]);
        }
    }
    return ok;
}
template <int DD>
__device__ __forceinline__ bool ns_in_boxes(const NsDev &S, int p, const double *x, const double *zz) {
    return ns_in_axis_boxes<DD>(S, p, x, zz) && ns_in_frames<DD>(S, p, zz);
}

// The shear of pixel p on a point in registers.  DD sampled dimensions, NC = DD / 5 components: the monomial list is known at
// compile time (same order as ns_shear_monomials), the loops unroll, the coefficients come through the scalar cache.
// inverse: z holds w on entry, the standardised unit-cube point on return (coordinate by coordinate, each from the ones before)
template <int DD, bool INVERSE>
__device__ __forceinline__ void ns_shear_apply(const k_dbl_p beta, int M, double *z) {
    constexpr int NC = DD / 5;
    double acc[DD];
#pragma unroll
    for (int j = 1; j < DD; ++j) {
        const k_dbl_p b = beta + j * M;
        double a = b[0];
        int m = 1;
#pragma unroll
        for (int t = 0; t < j; ++t) {
            a += b[m++] * z[t];
            a += b[m++] * (z[t] * z[t]);
#pragma unroll
            for (int k = 0; k < t; ++k)
                if (k % NC == t % NC) a += b[m++] * (z[k] * z[t]);
        }
        if (INVERSE) z[j] += a;
        else acc[j] = a;
    }
    if (!INVERSE) {
#pragma unroll
        for (int j = 1; j < DD; ++j) z[j] -= acc[j];
    }
}

// ---- candidates --------------------------------------------------------------------------
// Kr = candidates per pixel in this round (>= K: grows when few pixels are left, so the tail of
// slow pixels does not cost one launch per handful of candidates)
// DD > 0: the number of sampled dimensions at compile time (loops unroll, the proposal's coordinates live in registers)
#define NS_PROPOSE_THREADS 128
#ifndef NFA_PROPOSE_ATTR
#define NFA_PROPOSE_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#ifndef NFA_UPD_ATTR
#define NFA_UPD_ATTR            // (held to 128 registers -- four workgroups per CU, every pixel of a launch resident -- it spills 296 B and the two-component run takes 3.70 s against 3.42: profiles/r05/ab_sampler_registers.txt)
#endif
// in_range: the thread has a proposal of its own (k < Kr); the others of a pixel's last workgroup go along to the barriers
template <int DD>
__device__ __forceinline__ void ns_propose_one(const NsDev &S, int q, int k, int n_act, int Kr, bool in_range) {
    constexpr int DM = DD > 0 ? DD : NS_MAXD;
    long gid = (long)q * Kr + k;
    const int p = __builtin_amdgcn_readfirstlane(S.actlist[q]);      // a workgroup serves ONE pixel: its bound comes through scalar loads
    const int D = DD > 0 ? DD : S.D;
    if (!S.active[p]) {                 // finished since the last compaction of the pixel list
        if (in_range) S.valid[gid] = 0;
        return;
    }
    const uint64_t a = (uint64_t)S.cand_base[p] + (uint64_t)k;
    const uint64_t strm = ns_stream(S.seed, (uint64_t)p, a);
    // the proposal is kept in registers and written once it is known to be worth a likelihood: a proposal the bound
    // vetoes is never looked at again, and with no store ahead of them the loads of the bound (uniform over the
    // workgroup) go through the scalar cache -- as per-lane loads, 3200 of them per proposal, they were the kernel
    double x[DM], zq[DM];
    const bool walking = S.walk[p] != 0;
    if (!walking) {                     // the pixel's own share of the round's proposals
        const int kp = ns_kp(S, p, Kr);
        if ((int)(blockIdx.x * blockDim.x) >= kp) return;          // (the whole workgroup: nothing to do)
        in_range = in_range && k < kp;
    }
    bool ok = in_range;
    // the rotated frames' tests on packed survivors (compile-time dimensions: the queue's LDS is sized by them)
    const bool queued = DD > 0 && !walking && S.boxes && S.n_frames > 0;
    // (ONE branch per mode: stores of the walkers' branch ahead of the rejection branch on a common path -- a join
    // between two ifs -- would make the compiler read the bound with per-lane loads instead of scalar ones)
    if (walking) {
        // Metropolis step of walker k inside {L > threshold}; a cycle starts from a random live point
        if (!in_range) return;
        const int step = S.wstep[p];
        const int W = step == 0 ? min(ns_wmax(S, p), Kr) : S.wW[p];
        if (k >= W) { S.valid[gid] = 0; return; }
        double *wu = S.wU + ((long)p * S.w_stride + k) * D;
        if (step == 0) {
            const int Np = ns_n(S, p);
            const int idx = min(Np - 1, (int)(ns_uniform_of(strm, NS_B_START) * Np));
            const double *lu = S.Ulive + ((long)p * S.N + idx) * D, *lt = S.Tlive + ((long)p * S.N + idx) * S.DT;
            double *wt = S.wT + ((long)p * S.w_stride + k) * S.DT;
            for (int j = 0; j < D; ++j) wu[j] = lu[j];
            for (int j = 0; j < S.DT; ++j) wt[j] = lt[j];
            S.wL[(long)p * S.w_stride + k] = S.Llive[(long)p * S.N + idx];
            S.wnacc[(long)p * S.w_stride + k] = 0;
        }
        const double *origin = wu;
        const double wscale = S.wscale[p];
        // differential-evolution move (ter Braak 2006): the step is a scaled difference of two random
        // live points, so its shape follows the constraint region whatever that looks like (measured
        // against ellipsoid-shaped steps and an alternation of both on 576 two-component pixels: half
        // the lnZ bias at equal length, scripts/sampler_bias_check.py)
        const int N = ns_n(S, p);
        int ia = min(N - 1, (int)(ns_uniform_of(strm, 251ull) * N));
        int ib = min(N - 2, (int)(ns_uniform_of(strm, 252ull) * (N - 1)));
        if (ib >= ia) ib += 1;
        const double gam = wscale * 2.38 / sqrt(2.0 * D);
        const double *ua = S.Ulive + ((long)p * S.N + ia) * D, *ub = S.Ulive + ((long)p * S.N + ib) * D;
        for (int j = 0; j < D; ++j) {
            const double v = origin[j] + gam * (ua[j] - ub[j]);
            ok = ok && (v >= 0.0) && (v < 1.0);
            x[j] = v;
        }
    } else if (S.use_cube[p]) {    // early on the bounding ellipsoid is no better than the prior itself
        for (int j = 0; j < D; ++j) x[j] = ns_uniform_of(strm, (uint64_t)j);
        if (S.boxes) {                         // the boxes hold the live region whatever the proposal was drawn from
            const k_dbl_p c = (k_dbl_p)(S.centre + (long)p * NS_ME * D), A = (k_dbl_p)(S.axes + (long)p * NS_ME * D * D);
            double *zz = zq, w[DM];
            for (int j = 0; j < D; ++j) w[j] = x[j];
            if constexpr (DD == 10 || DD == 15) {
                if (S.shear) {                 // the boxes live in the sheared frame
                    const k_dbl_p mu = (k_dbl_p)(S.sh_mu + (long)p * D), sg = (k_dbl_p)(S.sh_sg + (long)p * D);
                    for (int j = 0; j < D; ++j) w[j] = (x[j] - mu[j]) / sg[j];
                    ns_shear_apply<DD, false>((k_dbl_p)(S.sh_beta + (long)p * D * S.sh_M), S.sh_M, w);
                }
            }
            for (int j = 0; j < D; ++j) {       // zz = A^-1 (w - c)
                double v = w[j] - c[j];
                for (int i = 0; i < j; ++i) v -= A[j * D + i] * zz[i];
                zz[j] = v / A[j * D + j];
            }
            ok = ok && ns_in_axis_boxes<DD>(S, p, w, zz);
            if constexpr (DD == 10 || DD == 15) { if (S.pairs) ok = ok && ns_in_pairs<DD>(S, p, w); }
            if (!queued) ok = ok && ns_in_frames<DD>(S, p, zz);
        }
    } else {
        double z[DM];
        double n2 = 0.0;
        for (int m = 0; m < D; m += 2) {    // Box-Muller pairs
            const double u1 = ns_uniform_of(strm, (uint64_t)m);
            const double u2 = ns_uniform_of(strm, (uint64_t)(m + 1));
            const double r = sqrt(-2.0 * log(u1));
            // (sine and cosine of 2 pi u2 in one call whose argument is in half turns: no reduction of a large angle, half
            // the instructions of cos(ang) and sin(ang); against the twin's numpy the values differ in the last bit at most)
            double sn, cs;
            sincospi(2.0 * u2, &sn, &cs);
            z[m] = r * cs;
            n2 += z[m] * z[m];
            if (m + 1 < D) { z[m + 1] = r * sn; n2 += z[m + 1] * z[m + 1]; }
        }
        const double ur = ns_uniform_of(strm, NS_B_RADIUS);
        const double f = exp(log(ur) / D) / sqrt(n2);               // uniform in the unit ball
        // several ellipsoids: one is drawn by volume, and a point that lies in q of them is kept with probability 1 / q
        // (uniform over the union)
        const int ne = S.multi ? S.nell[p] : 1;
        int ke = 0;
        if (ne > 1) {
            const double usel = ns_uniform_of(strm, NS_B_ELL), lv = S.lnvol[p];
            double acc = 0.0;
            ke = ne - 1;
            for (int k = 0; k < ne - 1; ++k) {
                acc += exp(S.elnv[(long)p * NS_ME + k] - lv);
                if (usel < acc) { ke = k; break; }
            }
        }
        const k_dbl_p c = (k_dbl_p)(S.centre + ((long)p * NS_ME + ke) * D), A = (k_dbl_p)(S.axes + ((long)p * NS_ME + ke) * D * D);
        for (int i = 0; i < D; ++i) z[i] *= f;                // the point of the unit ball
        bool sheared = false;
        if constexpr (DD == 10 || DD == 15) sheared = S.shear != 0;
        for (int j = 0; j < D; ++j) {
            double v = c[j];
            for (int i = 0; i <= j; ++i) v += A[j * D + i] * z[i];
            ok = ok && (sheared || ((v >= 0.0) && (v < 1.0)));   // outside the unit cube = outside the prior
            x[j] = v;
        }
        if (ok && S.boxes) ok = ns_in_axis_boxes<DD>(S, p, x, z);   // (one ellipsoid: the ball point IS A^-1 (x - c))
        if constexpr (DD == 10 || DD == 15) { if (ok && S.pairs) ok = ns_in_pairs<DD>(S, p, x); }     // (x: still the sheared point)
        if (queued) { for (int j = 0; j < D; ++j) zq[j] = z[j]; }
        else if (ok && S.boxes) ok = ns_in_frames<DD>(S, p, z);
        if constexpr (DD == 10 || DD == 15) {
            if (sheared && ok) {                // the ellipsoid's point is w: back through the shear to the unit cube
                const k_dbl_p mu = (k_dbl_p)(S.sh_mu + (long)p * D), sg = (k_dbl_p)(S.sh_sg + (long)p * D);
                ns_shear_apply<DD, true>((k_dbl_p)(S.sh_beta + (long)p * D * S.sh_M), S.sh_M, x);
                for (int j = 0; j < D; ++j) {
                    const double v = mu[j] + sg[j] * x[j];
                    ok = ok && (v >= 0.0) && (v < 1.0);
                    x[j] = v;
                }
            }
        }
        if (ok && ne > 1) {
            int q = 1;
            for (int k = 0; k < ne; ++k) {
                if (k == ke) continue;
                const double *ck = S.centre + ((long)p * NS_ME + k) * D, *Ak = S.axes + ((long)p * NS_ME + k) * D * D;
                double y[DM], s2 = 0.0;
                for (int j = 0; j < D; ++j) {
                    double v = x[j] - ck[j];
                    for (int i = 0; i < j; ++i) v -= Ak[j * D + i] * y[i];
                    y[j] = v / Ak[j * D + j];
                    s2 += y[j] * y[j];
                }
                q += s2 <= 1.0 ? 1 : 0;
            }
            if (q > 1) ok = ns_uniform_of(strm, NS_B_KEEP) * q < 1.0;
        }
    }
    if constexpr (DD > 0) {
        if (queued) {
            // Survivors of the cheap tests (the axis boxes, the unit cube) queue up in LDS; the first `n` threads of the
            // workgroup take one each through the rotated frames -- D x D products per frame that the whole wave pays while
            // any of its lanes is alive: packed, a quarter of the waves do
            __shared__ double qbuf[2 * DM * NS_PROPOSE_THREADS];
            __shared__ int qk[NS_PROPOSE_THREADS];
            __shared__ int qfail[NS_PROPOSE_THREADS];
            __shared__ int qn;
            const int tid = (int)threadIdx.x;
            if (tid == 0) qn = 0;
            qfail[tid] = 0;
            __syncthreads();
            if (ok) {
                const int e = atomicAdd(&qn, 1);
                for (int j = 0; j < DM; ++j) { qbuf[j * NS_PROPOSE_THREADS + e] = zq[j]; qbuf[(DM + j) * NS_PROPOSE_THREADS + e] = x[j]; }
                qk[e] = k;
            } else if (in_range) {
                S.valid[gid] = 0;
            }
            __syncthreads();
            const int n_q = qn;
            // The rotated frames, lanes = (frame, coordinate) pairs: a thread keeps the frame columns of its pairs in registers
            // -- loaded once per workgroup -- and every queued proposal's ball point is one broadcast read per coordinate for
            // all of them.  (With lanes = proposals the 100 entries of every frame came through the scalar cache for every
            // wave: 480 cache lines per wave, more than the cache holds for the pixels of a CU, and the launch spent its
            // time waiting for them -- 66 k cycles per wave of proposals.)
            constexpr int R = DM <= 10 ? 3 : 2;                // pairs per thread and pass
            const int n_pair = S.n_frames * DM;
            for (int e0 = 0; e0 < n_pair && n_q > 0; e0 += R * NS_PROPOSE_THREADS) {
                double qc[R][DM], lo[R], hi[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = e0 + r * NS_PROPOSE_THREADS + tid;
                    const bool on = e < n_pair;
                    const int kf = on ? e / DM : 0, j = on ? e - kf * DM : 0;
                    const double *Q = S.frames + (long)kf * DM * DM;
#pragma unroll
                    for (int a2 = 0; a2 < DM; ++a2) qc[r][a2] = Q[a2 * DM + j];
                    lo[r] = on ? S.fbox[((long)p * (S.n_frames + 1) + 1 + kf) * DM * 2 + 2 * j] : -INFINITY;
                    hi[r] = on ? S.fbox[((long)p * (S.n_frames + 1) + 1 + kf) * DM * 2 + 2 * j + 1] : INFINITY;
                }
                for (int i = 0; i < n_q; ++i) {
                    double zi[DM];
#pragma unroll
                    for (int j = 0; j < DM; ++j) zi[j] = qbuf[j * NS_PROPOSE_THREADS + i];
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        double w = 0.0;
#pragma unroll
                        for (int a2 = 0; a2 < DM; ++a2) w += zi[a2] * qc[r][a2];
                        bad = bad || !((w >= lo[r]) && (w <= hi[r]));
                    }
                    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (tid & 63) == 0) qfail[i] = 1;
                }
            }
            __syncthreads();
            ok = tid < n_q;
            if (ok) {
                for (int j = 0; j < DM; ++j) x[j] = qbuf[(DM + j) * NS_PROPOSE_THREADS + tid];
                gid = (long)q * Kr + qk[tid];
                ok = qfail[tid] == 0;
                S.valid[gid] = ok ? 1 : 0;
            }
        } else if (in_range) {
            S.valid[gid] = ok ? 1 : 0;
        }
    } else if (in_range) {
        S.valid[gid] = ok ? 1 : 0;
    }
    double *cu = S.candU + gid * D;
    if (ok) for (int j = 0; j < D; ++j) cu[j] = x[j];
    // only candidates inside the prior go to the likelihood: compact rows (order is irrelevant, every proposal
    // remembers its row).  One atomic per wave, not per proposal: tens of thousands of them on ONE counter took the
    // proposing launch 40 us by themselves.
    const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
    if (!ok) return;
    const int lane_id = (int)(threadIdx.x & 63);
    const int leader = __builtin_ctzll(okm);
    int base = 0;
    if (lane_id == leader) base = atomicAdd(S.count, __builtin_popcountll(okm));
    base = __builtin_amdgcn_readlane(base, leader);
    const int row = base + __builtin_popcountll(okm & ((1ull << lane_id) - 1ull));
    S.slot[gid] = row;
    S.candpix[row] = S.pixmap[p];
    double *ct = S.candT + (long)row * S.DT;
    for (int j = 0; j < S.DT; ++j) ct[j] = 0.5;
    for (int j = 0; j < D; ++j) ct[S.fmap[j]] = x[j];
}

// grid: x = chunks of a pixel's Kr proposals, y = the active pixels of this part (z: beyond 65535 of them)
template <int DD>
__global__ void __launch_bounds__(NS_PROPOSE_THREADS) ns_propose_kernel(NsDev S, int n_act, int Kr) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int q = (int)(blockIdx.y + blockIdx.z * 65535u);
    if (q < n_act) ns_propose_one<DD>(S, q, k, n_act, Kr, k < Kr);
}

// the same held to 128 vector registers: four waves per SIMD where the ten-dimensional form asked for 129 (and three)
template <int DD>
__global__ void __launch_bounds__(NS_PROPOSE_THREADS) NFA_PROPOSE_ATTR ns_propose_kernel_v128(NsDev S, int n_act, int Kr) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int q = (int)(blockIdx.y + blockIdx.z * 65535u);
    if (q < n_act) ns_propose_one<DD>(S, q, k, n_act, Kr, k < Kr);
}

// One thread behind a proposing launch: the number of compact rows and the round's sequence number go into host
// memory the device can address, the counter back to zero for the next round.  (In place of a memset before the
// launch, a four-byte copy and an event behind it.  The same done by the proposing launch's last workgroup -- a
// fence and a tick per workgroup -- made the rounds slower: an agent-scope fence writes the L2 back.)
__global__ void ns_publish_kernel(NsDev S, unsigned long long seq) {
    const int rows = *S.count;
    *S.count = 0;
    __hip_atomic_store(S.host_rows, rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(S.host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- wave helpers (one 64-lane wave per pixel) -------------------------------------------------
// DPP butterflies inside the rows of 16 lanes, the four rows combined through readlane: a
// replacement costs one arg-min over the live points, and there are ~15 k of them per pixel.
__device__ __forceinline__ double ns_wave_sum(double v) { return wave_sum(v); }
template <int CTRL>
__device__ __forceinline__ void ns_max_step(double &v) { v = fmax(v, dpp_move<CTRL>(v)); }
__device__ __forceinline__ double ns_wave_max(double v) {
    ns_max_step<0xB1>(v); ns_max_step<0x4E>(v); ns_max_step<0x141>(v); ns_max_step<0x140>(v);
    return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}
// minimum with the lowest index among equals (numpy.argmin)
__device__ __forceinline__ void ns_min_pick(double &v, int &ix, double ov, int oi) {
    if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
}
template <int CTRL>
__device__ __forceinline__ void ns_argmin_step(double &v, int &ix) {
    const double ov = dpp_move<CTRL>(v);
    const int oi = __builtin_amdgcn_update_dpp(0, ix, CTRL, 0xf, 0xf, false);
    ns_min_pick(v, ix, ov, oi);
}
__device__ __forceinline__ void ns_wave_argmin(double &v, int &ix) {
    ns_argmin_step<0xB1>(v, ix); ns_argmin_step<0x4E>(v, ix); ns_argmin_step<0x141>(v, ix); ns_argmin_step<0x140>(v, ix);
    double m = readlane_d(v, 0);
    int mi = __builtin_amdgcn_readlane(ix, 0);
    ns_min_pick(m, mi, readlane_d(v, 16), __builtin_amdgcn_readlane(ix, 16));
    ns_min_pick(m, mi, readlane_d(v, 32), __builtin_amdgcn_readlane(ix, 32));
    ns_min_pick(m, mi, readlane_d(v, 48), __builtin_amdgcn_readlane(ix, 48));
    v = m; ix = mi;
}
__device__ __forceinline__ double ns_logaddexp(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const double m = fmax(a, b);
    return m + log1p(exp(-fabs(a - b)));
}

// Bounding ellipsoid of the live points of pixel p (same arithmetic as _fit_ellipsoids in
// nestfit_amd/sampler.py): centre = mean, A = chol(cov) * sqrt(max Mahalanobis^2) * growth, the
// growth bringing the volume up to X / efr where the bounding ellipsoid is smaller than that.
// sA: D*D doubles of LDS, sc: D doubles.
// sum / maximum over the workgroup's NT threads (a multiple of 64), every thread gets the result; sred: 8 doubles of LDS
__device__ __forceinline__ double ns_block_sum(double v, double *sred, int tid, int NT) {
    v = ns_wave_sum(v);
    if (NT == 64) return v;
    __syncthreads();
    if ((tid & 63) == 0) sred[tid >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < NT / 64; ++k) t += sred[k];
    return t;
}
__device__ __forceinline__ double ns_block_max(double v, double *sred, int tid, int NT) {
    v = ns_wave_max(v);
    if (NT == 64) return v;
    __syncthreads();
    if ((tid & 63) == 0) sred[tid >> 6] = v;
    __syncthreads();
    double t = sred[0];
    for (int k = 1; k < NT / 64; ++k) t = fmax(t, sred[k]);
    return t;
}

// The shear of pixel p from its live points, staged in sd[N][D] (LDS); on return sd holds the sheared points w, the
// coefficients are in global memory and the result is ln |du / dw| = sum ln sg.  sh = LDS scratch: [M*M][D*M][2 D].
// One Gram matrix of all monomials and ONE Cholesky factorisation serve every coordinate: the factor of a leading block is
// the leading block of the factor, and row start[j] of the factor is the forward substitution of coordinate j's normal
// equations (its right-hand side is the Gram column of the monomial z_j itself).
#define NS_RTICK(slot) do { if (S.dbg && blockIdx.x == 0 && tid == 0) { const long t_ = (long)wall_clock64(); S.dbg[slot] += t_ - t_r; t_r = t_; } } while (0)
__device__ double ns_shear_fit(const NsDev &S, int p, int N, double *sd, double *sh, int tid, int NT, double *sred, long &t_r) {
    const int D = S.D, M = S.sh_M;
    double *sG = sh, *sB = sG + M * M, *smu = sB + D * M, *ssg = smu + D, *sone = ssg + D;
    if (tid == 0) sone[0] = 1.0;
    const int *mono = S.sh_mono, *start = S.sh_start;
    for (int j = tid; j < D; j += NT) {
        double acc = 0.0;
        for (int i = 0; i < N; ++i) acc += sd[i * D + j];
        const double mu = acc / N;
        double q = 0.0;
        for (int i = 0; i < N; ++i) { const double d = sd[i * D + j] - mu; q += d * d; }
        smu[j] = mu;
        ssg[j] = fmax(sqrt(q / (N - 1)), 1e-300);
    }
    __syncthreads();
    for (int e = tid; e < N * D; e += NT) { const int j = e % D; sd[e] = (sd[e] - smu[j]) / ssg[j]; }
    __syncthreads();
    NS_RTICK(48);                                   // mean, spread, standardise
    // Gram matrix, lower triangle: lanes = entries, each walking all points
    const int n_ent = M * (M + 1) / 2;
    for (int e = tid; e < n_ent; e += NT) {
        int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (r * (r + 1) / 2 > e) --r;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        const int c = e - r * (r + 1) / 2;
        // the entry's (up to) four factors as (address of the first point's value, stride): a missing factor reads the
        // constant 1 with stride 0 -- no branches inside the walk over the points, four points in flight
        const int f[4] = {mono[2 * r], mono[2 * r + 1], mono[2 * c], mono[2 * c + 1]};
        const double *fp[4];
        int fs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { fp[t] = f[t] < 0 ? sone : sd + f[t]; fs[t] = f[t] < 0 ? 0 : D; }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int i = 0;
        for (; i + 4 <= N; i += 4) {
            a0 += (fp[0][(i + 0) * fs[0]] * fp[1][(i + 0) * fs[1]]) * (fp[2][(i + 0) * fs[2]] * fp[3][(i + 0) * fs[3]]);
            a1 += (fp[0][(i + 1) * fs[0]] * fp[1][(i + 1) * fs[1]]) * (fp[2][(i + 1) * fs[2]] * fp[3][(i + 1) * fs[3]]);
            a2 += (fp[0][(i + 2) * fs[0]] * fp[1][(i + 2) * fs[1]]) * (fp[2][(i + 2) * fs[2]] * fp[3][(i + 2) * fs[3]]);
            a3 += (fp[0][(i + 3) * fs[0]] * fp[1][(i + 3) * fs[1]]) * (fp[2][(i + 3) * fs[2]] * fp[3][(i + 3) * fs[3]]);
        }
        for (; i < N; ++i) a0 += (fp[0][i * fs[0]] * fp[1][i * fs[1]]) * (fp[2][i * fs[2]] * fp[3][i * fs[3]]);
        sG[r * M + c] = ((a0 + a1) + (a2 + a3)) + (r == c ? NS_SHEAR_RIDGE * N : 0.0);
    }
    __syncthreads();
    NS_RTICK(49);                                   // Gram matrix
    // Cholesky in place, column by column: lane 0 the diagonal, lanes = rows below it.  A monomial whose pivot has
    // drowned in rounding (live points squeezed onto a line at ln X ~ -50: the ridge keeps the matrix positive definite
    // on paper only) is dropped -- pivot = its own norm, nothing below it: its coefficient comes out as zero and the
    // rest are those of the regression without it -- instead of dividing by a rounding error
    for (int j = 0; j < M; ++j) {
        if (tid == 0) {
            const double g = sG[j * M + j];
            double d = g;
            for (int k = 0; k < j; ++k) d -= sG[j * M + k] * sG[j * M + k];
            const bool keep = d > NS_SHEAR_PIVOT * g;
            sG[j * M + j] = keep ? sqrt(d) : -sqrt(fmax(g, 1e-300));       // (the sign: the flag for the rows below)
        }
        __syncthreads();
        const double dj = sG[j * M + j];
        for (int i = j + 1 + tid; i < M; i += NT) {
            double v = sG[i * M + j];
            for (int k = 0; k < j; ++k) v -= sG[i * M + k] * sG[j * M + k];
            sG[i * M + j] = dj > 0.0 ? v / dj : 0.0;
        }
        __syncthreads();
        if (tid == 0 && dj < 0.0) sG[j * M + j] = -dj;
        // (row j's own off-diagonal entries stay: they are the forward substitution of a coordinate whose monomial this is)
    }
    NS_RTICK(50);                                   // Cholesky
    // coefficients: lanes = coordinates, back substitution with the transposed leading block
    for (int j = tid; j < D; j += NT) {
        const int pj = j == 0 ? 0 : start[j];
        double *b = sB + j * M;
        for (int r = 0; r < M; ++r) b[r] = 0.0;
        for (int r = pj - 1; r >= 0; --r) {
            double v = sG[pj * M + r];
            for (int k = r + 1; k < pj; ++k) v -= sG[k * M + r] * b[k];
            b[r] = v / sG[r * M + r];
        }
    }
    __syncthreads();
    NS_RTICK(51);                                   // back substitutions
    for (int e = tid; e < D * M; e += NT) S.sh_beta[(long)p * D * M + e] = sB[e];
    for (int j = tid; j < D; j += NT) { S.sh_mu[(long)p * D + j] = smu[j]; S.sh_sg[(long)p * D + j] = ssg[j]; }
    // w in place: lanes = points, from the last coordinate down (a coordinate's features are the earlier z)
    for (int i = tid; i < N; i += NT) {
        double *z = sd + i * D;
        for (int j = D - 1; j >= 1; --j) {
            const int pj = start[j];
            const double *b = sB + j * M;
            double acc = b[0];
            for (int m = 1; m < pj; ++m) {
                const int a0 = mono[2 * m], a1 = mono[2 * m + 1];
                acc += b[m] * (a1 < 0 ? z[a0] : z[a0] * z[a1]);
            }
            z[j] -= acc;
        }
    }
    double lj = 0.0;
    for (int j = 0; j < D; ++j) lj += log(ssg[j]);
    __syncthreads();
    NS_RTICK(52);                                   // w in place
    return lj;
}

// Extent of the live points (rows of sd, DD coordinates each, in LDS) along column j of the frame Q: the column in
// registers, four points in flight (a lone chain of broadcast read -> DD multiply-adds per point was most of a refit)
template <int DD>
__device__ __forceinline__ void ns_frame_extent(const double *sd, int N, const double *Q, int j, double &lo, double &hi) {
    double qc[DD];
#pragma unroll
    for (int a = 0; a < DD; ++a) qc[a] = Q[a * DD + j];
    int i = 0;
    for (; i + 4 <= N; i += 4) {
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll
        for (int a = 0; a < DD; ++a) {          // (each sum in the order a = 0, 1, ...: the same bits as one point at a time)
            w0 += sd[(i + 0) * DD + a] * qc[a];
            w1 += sd[(i + 1) * DD + a] * qc[a];
            w2 += sd[(i + 2) * DD + a] * qc[a];
            w3 += sd[(i + 3) * DD + a] * qc[a];
        }
        lo = fmin(fmin(lo, w0), fmin(w1, fmin(w2, w3)));
        hi = fmax(fmax(hi, w0), fmax(w1, fmax(w2, w3)));
    }
    for (; i < N; ++i) {
        double w = 0.0;
#pragma unroll
        for (int a = 0; a < DD; ++a) w += sd[i * DD + a] * qc[a];
        lo = fmin(lo, w); hi = fmax(hi, w);
    }
}

__device__ void ns_refit(const NsDev &S, int p, long n_iter, double *sA, double *sc, double *sd, double *sh, int tid, int NT, double *sred) {
    const int N = ns_n(S, p), D = S.D;
    const double *U = S.Ulive + (long)p * S.N * D;
    double tr = 0.0, ln_jac = 0.0, ln_enl = S.ln_enlarge;
    long t_r = S.dbg && blockIdx.x == 0 && tid == 0 ? (long)wall_clock64() : 0;
    if (sd) {
        // the live points fit in LDS (N * D doubles): staged once with coalesced loads, then lanes =
        // dimensions for the mean and lanes = entries of the covariance matrix, each walking all
        // points in LDS -- the D(D+1)/2 wave reductions over global memory of the fallback below cost
        // ~200 us per refit for D = 12
        for (int e = tid; e < N * D; e += NT) sd[e] = U[e];
        __syncthreads();
        if (S.shear) { ln_jac = ns_shear_fit(S, p, N, sd, sh, tid, NT, sred, t_r); ln_enl = S.ln_enlarge_shear; }   // sd: w from here on
        for (int j = tid; j < D; j += NT) {
            double acc = 0.0;
            for (int i = 0; i < N; ++i) acc += sd[i * D + j];
            sc[j] = acc / N;
        }
        __syncthreads();
        for (int e = tid; e < N * D; e += NT) sd[e] -= sc[e % D];
        __syncthreads();
        const int n_ent = D * (D + 1) / 2;
        for (int e = tid; e < n_ent; e += NT) {
            int a = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);          // row of the lower triangle
            while (a * (a + 1) / 2 > e) --a;
            while ((a + 1) * (a + 2) / 2 <= e) ++a;
            const int b = e - a * (a + 1) / 2;
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;         // (four points in flight)
            int i = 0;
            for (; i + 4 <= N; i += 4) {
                c0 += sd[(i + 0) * D + a] * sd[(i + 0) * D + b];
                c1 += sd[(i + 1) * D + a] * sd[(i + 1) * D + b];
                c2 += sd[(i + 2) * D + a] * sd[(i + 2) * D + b];
                c3 += sd[(i + 3) * D + a] * sd[(i + 3) * D + b];
            }
            for (; i < N; ++i) c0 += sd[i * D + a] * sd[i * D + b];
            sA[a * D + b] = ((c0 + c1) + (c2 + c3)) / (N - 1);
        }
        __syncthreads();
        for (int a = 0; a < D; ++a) tr += sA[a * D + a];
    } else {
        for (int j = 0; j < D; ++j) {
            double acc = 0.0;
            for (int i = tid; i < N; i += NT) acc += U[(long)i * D + j];
            acc = ns_block_sum(acc, sred, tid, NT);
            if (tid == 0) sc[j] = acc / N;
        }
        __syncthreads();
        for (int a = 0; a < D; ++a)
            for (int b = 0; b <= a; ++b) {
                double acc = 0.0;
                for (int i = tid; i < N; i += NT) acc += (U[(long)i * D + a] - sc[a]) * (U[(long)i * D + b] - sc[b]);
                acc = ns_block_sum(acc, sred, tid, NT) / (N - 1);
                if (tid == 0) sA[a * D + b] = acc;
                if (a == b) tr += acc;
            }
        __syncthreads();
    }
    NS_RTICK(53);                                   // (stage,) mean, covariance
    if (S.pairs && sd && sh) {
        // The pair ellipses (the twin's _fit_pairs): for every pair i < j the ellipse around the live points' projection
        // onto (w_i, w_j) -- its covariance is three entries of the matrix just formed.  Threads = (pair, slice of the
        // points) for the largest Mahalanobis distance, combined through an integer maximum on the (non-negative) bits.
        const int n_pr = D * (D - 1) / 2;
        double *sp = sh;                            // [n_pr][4]: L00, L10, L11, r2 (the shear's scratch is free again)
        for (int e = tid; e < n_pr; e += NT) {
            int j = (int)((sqrt(8.0 * e + 1.0) + 1.0) * 0.5);          // pairs in the order j = 1.., i = 0..j-1
            while (j * (j - 1) / 2 > e) --j;
            while ((j + 1) * j / 2 <= e) ++j;
            const int i = e - j * (j - 1) / 2;
            const double l00 = sqrt(fmax(sA[i * D + i], 1e-300));
            const double l10 = sA[j * D + i] / l00;
            const double l11 = sqrt(fmax(sA[j * D + j] - l10 * l10, 1e-300));
            sp[4 * e] = l00; sp[4 * e + 1] = l10; sp[4 * e + 2] = l11; sp[4 * e + 3] = 0.0;
        }
        __syncthreads();
        const int n_sl = max(1, NT / n_pr);
        if (tid < n_pr * n_sl) {
            const int e = tid % n_pr, sl = tid / n_pr;
            int j = (int)((sqrt(8.0 * e + 1.0) + 1.0) * 0.5);
            while (j * (j - 1) / 2 > e) --j;
            while ((j + 1) * j / 2 <= e) ++j;
            const int i = e - j * (j - 1) / 2;
            const double l00 = sp[4 * e], l10 = sp[4 * e + 1], l11 = sp[4 * e + 2];
            double r2 = 0.0;
            for (int n = sl; n < N; n += n_sl) {
                const double y0 = sd[n * D + i] / l00;
                const double y1 = (sd[n * D + j] - l10 * y0) / l11;
                r2 = fmax(r2, y0 * y0 + y1 * y1);
            }
            atomicMax((unsigned long long *)&sp[4 * e + 3], (unsigned long long)__double_as_longlong(r2));
        }
        __syncthreads();
        for (int e = tid; e < n_pr; e += NT) {
            int j = (int)((sqrt(8.0 * e + 1.0) + 1.0) * 0.5);
            while (j * (j - 1) / 2 > e) --j;
            while ((j + 1) * j / 2 <= e) ++j;
            const int i = e - j * (j - 1) / 2;
            const double sc2 = sqrt(sp[4 * e + 3] * S.pairs_enlarge);
            double *pt = S.pair_tab + ((long)p * n_pr + e) * 5;
            pt[0] = sc[i]; pt[1] = sc[j]; pt[2] = 1.0 / (sp[4 * e] * sc2); pt[3] = sp[4 * e + 1] * sc2; pt[4] = 1.0 / (sp[4 * e + 2] * sc2);
        }
        __syncthreads();
    }
    if (tid == 0) {                    // Cholesky, lower triangle in place
        const double eps = 1e-12 * fmax(tr, 1e-30);
        for (int a = 0; a < D; ++a) sA[a * D + a] += eps;
        for (int j = 0; j < D; ++j) {
            double d = sA[j * D + j];
            for (int k = 0; k < j; ++k) d -= sA[j * D + k] * sA[j * D + k];
            d = sqrt(fmax(d, 1e-300));
            sA[j * D + j] = d;
            for (int i = j + 1; i < D; ++i) {
                double v = sA[i * D + j];
                for (int k = 0; k < j; ++k) v -= sA[i * D + k] * sA[j * D + k];
                sA[i * D + j] = v / d;
            }
        }
    }
    __syncthreads();
    const bool boxes = S.boxes && sd;
    if (boxes) {
        // the box in the unit cube's own axes (the centred points are still in LDS): lanes = dimensions
        for (int j = tid; j < D; j += NT) {
            double lo = INFINITY, hi = -INFINITY, acc = 0.0;
            for (int i = 0; i < N; ++i) {
                const double d = sd[i * D + j];
                lo = fmin(lo, d); hi = fmax(hi, d); acc += d * d;
            }
            const double sg = sqrt(acc / (N - 1));
            double *ub = S.ubox + ((long)p * D + j) * 2;
            ub[0] = sc[j] + lo - S.margin_c * fmax(NS_MARGIN_FLOOR * sg, -lo - NS_MARGIN_A * sg);
            ub[1] = sc[j] + hi + S.margin_c * fmax(NS_MARGIN_FLOOR * sg, hi - NS_MARGIN_A * sg);
        }
        __syncthreads();
    }
    NS_RTICK(54);                                   // Cholesky of the covariance, the box in the w axes
    double r2 = 0.0, ssq = 0.0;
    for (int i = tid; i < N; i += NT) {           // y = L^-1 (u_i - c), forward substitution
        double s2 = 0.0;
        if (boxes) {                               // in place: the boxes below want every point's y
            for (int a = 0; a < D; ++a) {
                double v = sd[i * D + a];
                for (int k = 0; k < a; ++k) v -= sA[a * D + k] * sd[i * D + k];
                v /= sA[a * D + a];
                sd[i * D + a] = v;
                s2 += v * v;
            }
        } else {
            double y[NS_MAXD];
            for (int a = 0; a < D; ++a) {
                double v = sd ? sd[i * D + a] : U[(long)i * D + a] - sc[a];
                for (int k = 0; k < a; ++k) v -= sA[a * D + k] * y[k];
                y[a] = v / sA[a * D + a];
                s2 += y[a] * y[a];
            }
        }
        r2 = fmax(r2, s2);
        ssq += s2;
    }
    r2 = ns_block_max(r2, sred, tid, NT);
    NS_RTICK(55);                                   // y of every point
    // the covariance ellipsoid scaled to enclose every live point, then MultiNest's rule: enlarged
    // until its volume is at least the expected prior volume over the target efficiency, X / efr
    double lnv = S.ln_vball + 0.5 * D * log(r2) + ln_enl;   // safety factor on the enclosing volume
    for (int a = 0; a < D; ++a) lnv += log(sA[a * D + a]);
    // (with the shear the ellipsoid is in w units: the prior volume to hold is X / |du / dw| there)
    const double ln_x = -(double)n_iter / N - ln_jac;
    const double grow = fmax((ln_x - S.ln_efr) - lnv, 0.0);
    const double scale = sqrt(r2) * exp((grow + ln_enl) / D);
    const double lnvol_u = (lnv + grow) + ln_jac;
    if (tid == 0) { S.use_cube[p] = lnvol_u >= 0.0 ? 1 : 0; S.lnvol[p] = lnvol_u; }   // >= cube: use the cube
    double *A = S.axes + (long)p * NS_ME * D * D, *c = S.centre + (long)p * NS_ME * D;     // ellipsoid 0 of the pixel
    if (tid == 0) { S.nell[p] = 1; S.elnv[(long)p * NS_ME] = lnvol_u; }
    for (int e = tid; e < D * D; e += NT) {
        const int a = e / D, b = e - a * D;
        A[e] = b <= a ? sA[e] * scale : 0.0;
    }
    for (int j = tid; j < D; j += NT) c[j] = sc[j];
    if (boxes) {
        // zz = A^-1 (u - c) = y / scale: the live points in the frame a proposal is drawn in (its unit-ball point);
        // their boxes in that frame and in the fixed rotations of it.  lanes = points, one (frame, coordinate) at a time
        const double inv = 1.0 / scale;
        ssq = ns_block_sum(ssq, sred, tid, NT);
        const double sz = sqrt(ssq * inv * inv / ((double)(N - 1) * D));     // the spread of zz, the same in every direction
        const double mfloor = NS_MARGIN_FLOOR * sz, moff = NS_MARGIN_A * sz;
        // (zz in place first: one multiplication per coordinate instead of one per use)
        for (int e = tid; e < N * D; e += NT) sd[e] *= inv;
        __syncthreads();
        // lanes = (frame, coordinate) pairs, each walking all points: a lane keeps its column of the frame in registers
        // (up to NS_QCOL entries at a time) and its own minimum and maximum -- no wave reductions, and every point's
        // coordinates are one broadcast read for all lanes.  (The first version had lanes = points: 660 wave reductions
        // and a scalar load of the frame's entry inside the innermost loop made a refit 2 ms.)
        constexpr int NS_QCOL = 16;
        const int n_pair = (S.n_frames + 1) * D;
        for (int e0 = 0; e0 < n_pair; e0 += NT) {
            const int e = e0 + tid;
            const bool on = e < n_pair;
            const int k = on ? e / D : 0, j = on ? e - k * D : 0;
            double lo = INFINITY, hi = -INFINITY;
            if (k == 0) {
                for (int i = 0; i < N; ++i) { const double w = sd[i * D + j]; lo = fmin(lo, w); hi = fmax(hi, w); }
            } else if (D == 10) {
                ns_frame_extent<10>(sd, N, S.frames + (long)(k - 1) * D * D, j, lo, hi);
            } else if (D == 15) {
                ns_frame_extent<15>(sd, N, S.frames + (long)(k - 1) * D * D, j, lo, hi);
            } else if (D <= NS_QCOL) {
                double qc[NS_QCOL];
                const double *Q = S.frames + (long)(k - 1) * D * D;
#pragma unroll
                for (int a = 0; a < NS_QCOL; ++a) qc[a] = a < D ? Q[a * D + j] : 0.0;
                for (int i = 0; i < N; ++i) {
                    double w = 0.0;
#pragma unroll
                    for (int a = 0; a < NS_QCOL; ++a) if (a < D) w += sd[i * D + a] * qc[a];
                    lo = fmin(lo, w); hi = fmax(hi, w);
                }
            } else {
                const double *Q = S.frames + (long)(k - 1) * D * D;
                for (int i = 0; i < N; ++i) {
                    double w = 0.0;
                    for (int a = 0; a < D; ++a) w += sd[i * D + a] * Q[a * D + j];
                    lo = fmin(lo, w); hi = fmax(hi, w);
                }
            }
            if (on) {
                double *fb = S.fbox + ((long)p * n_pair + e) * 2;
                fb[0] = lo - S.margin_c * fmax(mfloor, -lo - moff);
                fb[1] = hi + S.margin_c * fmax(mfloor, hi - moff);
            }
        }
    }
    __syncthreads();
    NS_RTICK(56);                                   // the frames' boxes
    if (S.dbg && blockIdx.x == 0 && tid == 0) S.dbg[57] += 1;
}

// ---- several ellipsoids ----------------------------------------------------------------------
// One fit slot in LDS: [c: D][L: D*D, lower][cov: D*D, lower][r2, lnv, n, final]
__host__ __device__ inline int ns_me_slot(int D) { return D + 2 * D * D + 4; }
// Mean, covariance, Cholesky factor, largest Mahalanobis distance and ln volume (safety factor included) of the live
// points whose label is k (and, with side >= 0, whose side bit is `side`); su = the pixel's live points in LDS.
template <int DD>
__device__ void ns_me_fit(const NsDev &S, const double *su, const int *lab, int N, int k, int side, double *f, int lane) {
    constexpr int D = DD;                // (a compile-time dimension: every loop below unrolls)
    double *fc = f, *fL = f + D, *fC = f + D + D * D, *fs = f + D + 2 * D * D;
    auto member = [&](int i) { const int l = lab[i]; return (l & 7) == k && (side < 0 || ((l >> 3) & 1) == side); };
    // One pass over the lane's points for the count and the sums of the coordinates, one for all entries of the
    // covariance: a lane's partial sums are formed point by point in the order entry-by-entry loops would form them, but
    // a pass is a handful of trips to LDS where those are one trip per (entry, point) -- the fit is a lone wave's chain
    // of LDS latencies, not arithmetic.
    constexpr int ME = DD * (DD + 1) / 2;
    double cnt = 0.0, sm[DD];
#pragma unroll
    for (int j = 0; j < DD; ++j) sm[j] = 0.0;
    for (int i = lane; i < N; i += 64) {
        const bool in = member(i);
        cnt += in ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < DD; ++j) sm[j] += in ? su[i * D + j] : 0.0;
    }
    cnt = ns_wave_sum(cnt);
#pragma unroll
    for (int j = 0; j < DD; ++j) {
        const double acc = ns_wave_sum(sm[j]);
        if (lane == 0) fc[j] = acc / cnt;
    }
    wave_lds_sync();
    double cv[ME], mean[DD];
#pragma unroll
    for (int e = 0; e < ME; ++e) cv[e] = 0.0;
#pragma unroll
    for (int j = 0; j < DD; ++j) mean[j] = fc[j];
    for (int i = lane; i < N; i += 64) {
        const bool in = member(i);
        double dx[DD];
#pragma unroll
        for (int j = 0; j < DD; ++j) dx[j] = su[i * D + j] - mean[j];
#pragma unroll
        for (int a2 = 0; a2 < DD; ++a2)
#pragma unroll
            for (int b2 = 0; b2 <= a2; ++b2) cv[a2 * (a2 + 1) / 2 + b2] += in ? dx[a2] * dx[b2] : 0.0;
    }
    double tr = 0.0;
#pragma unroll
    for (int a2 = 0; a2 < DD; ++a2)
#pragma unroll
        for (int b2 = 0; b2 <= a2; ++b2) {
            const double acc = ns_wave_sum(cv[a2 * (a2 + 1) / 2 + b2]) / (cnt - 1.0);
            if (lane == 0) { fC[a2 * D + b2] = acc; fL[a2 * D + b2] = acc; }
            if (a2 == b2) tr += acc;
        }
    wave_lds_sync();
    if (lane == 0) {                    // Cholesky, lower triangle in place (as ns_refit)
        const double eps = 1e-12 * fmax(tr, 1e-30);
        for (int a = 0; a < D; ++a) fL[a * D + a] += eps;
        for (int j = 0; j < D; ++j) {
            double d = fL[j * D + j];
            for (int q = 0; q < j; ++q) d -= fL[j * D + q] * fL[j * D + q];
            d = sqrt(fmax(d, 1e-300));
            fL[j * D + j] = d;
            for (int i = j + 1; i < D; ++i) {
                double v = fL[i * D + j];
                for (int q = 0; q < j; ++q) v -= fL[i * D + q] * fL[j * D + q];
                fL[i * D + j] = v / d;
            }
        }
    }
    wave_lds_sync();
    double r2 = 0.0;
    for (int i = lane; i < N; i += 64) {
        if (!member(i)) continue;
        double y[DD];
        double s2 = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double v = su[i * D + a] - fc[a];
#pragma unroll
            for (int q = 0; q < a; ++q) v -= fL[a * D + q] * y[q];
            y[a] = v / fL[a * D + a];
            s2 += y[a] * y[a];
        }
        r2 = fmax(r2, s2);
    }
    r2 = ns_wave_max(r2);
    double lnv = S.ln_vball + 0.5 * D * log(r2) + S.ln_enlarge;
    for (int a = 0; a < D; ++a) lnv += log(fL[a * D + a]);
    if (lane == 0) { fs[0] = r2; fs[1] = lnv; fs[2] = cnt; fs[3] = 0.0; }
    wave_lds_sync();
}

// The bound of pixel p as up to NS_ME ellipsoids (same decisions as _fit_multi in nestfit_amd/sampler.py): the cluster
// with the largest ellipsoid is cut across its principal axis at its centre; the cut stays when the halves' ellipsoids
// together have less than NS_ME_GAIN of its volume, else the cluster is final; a cluster below 4 (D + 2) points is not
// cut, a half below 2 (D + 2) not accepted.  Then MultiNest's rule on the sum of the volumes (X / efr at least).
// su: N * D doubles, lab: N ints, wf: (NS_ME + 2) fit slots -- all LDS.
template <int DD>
__device__ void ns_refit_multi(const NsDev &S, int p, long n_iter, double *su, int *lab, double *wf, int lane) {
    constexpr int D = DD;
    const int N = ns_n(S, p), FS = ns_me_slot(D), minp = 2 * (D + 2);
    const double *U = S.Ulive + (long)p * S.N * D;
    for (int e = lane; e < N * D; e += 64) su[e] = U[e];
    for (int i = lane; i < N; i += 64) lab[i] = 0;
    wave_lds_sync();
    ns_me_fit<DD>(S, su, lab, N, 0, -1, wf, lane);
    int ncl = 1;
    while (ncl < S.max_ell) {
        int best = -1;
        for (int k = 0; k < ncl; ++k) {
            const double *fs = wf + k * FS + D + 2 * D * D;
            if (fs[3] == 0.0 && fs[2] >= 2.0 * minp && (best < 0 || fs[1] > wf[best * FS + D + 2 * D * D + 1])) best = k;
        }
        if (best < 0) break;
        double *fb = wf + best * FS;
        // principal axis of the cluster's covariance: twenty steps of the power iteration from (1, ..., 1)
        double v[DD], cm[DD][DD];       // (the covariance out of LDS once: 400 dependent reads otherwise)
#pragma unroll
        for (int a = 0; a < DD; ++a)
#pragma unroll
            for (int b = 0; b < DD; ++b) cm[a][b] = fb[D + D * D + (a >= b ? a * D + b : b * D + a)];
#pragma unroll
        for (int a = 0; a < DD; ++a) v[a] = 1.0;
        for (int it = 0; it < 20; ++it) {
            double w[DD], n2 = 0.0;
#pragma unroll
            for (int a = 0; a < DD; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int b = 0; b < DD; ++b) acc += cm[a][b] * v[b];
                w[a] = acc;
                n2 += acc * acc;
            }
            const double inv = 1.0 / sqrt(n2);
#pragma unroll
            for (int a = 0; a < DD; ++a) v[a] = w[a] * inv;
        }
        for (int i = lane; i < N; i += 64) {
            if ((lab[i] & 7) != best) continue;
            double proj = 0.0;
            for (int a = 0; a < D; ++a) proj += (su[i * D + a] - fb[a]) * v[a];
            lab[i] = best | ((proj >= 0.0 ? 1 : 0) << 3);
        }
        wave_lds_sync();
        double *fA = wf + NS_ME * FS, *fB = wf + (NS_ME + 1) * FS;
        ns_me_fit<DD>(S, su, lab, N, best, 0, fA, lane);
        ns_me_fit<DD>(S, su, lab, N, best, 1, fB, lane);
        const double nA = fA[D + 2 * D * D + 2], nB = fB[D + 2 * D * D + 2];
        const double lvA = fA[D + 2 * D * D + 1], lvB = fB[D + 2 * D * D + 1], lvP = fb[D + 2 * D * D + 1];
        const bool keep = nA >= minp && nB >= minp && ns_logaddexp(lvA, lvB) < lvP + log(NS_ME_GAIN);
        wave_lds_sync();
        if (keep) {
            for (int i = lane; i < N; i += 64) {
                const int l = lab[i];
                if ((l & 7) == best) lab[i] = ((l >> 3) & 1) ? ncl : best;
            }
            double *fn = wf + ncl * FS;
            for (int e = lane; e < FS; e += 64) { fb[e] = fA[e]; fn[e] = fB[e]; }
            ncl += 1;
        } else if (lane == 0) {
            fb[D + 2 * D * D + 3] = 1.0;
        }
        wave_lds_sync();
    }
    double tot = -INFINITY;
    for (int k = 0; k < ncl; ++k) tot = ns_logaddexp(tot, wf[k * FS + D + 2 * D * D + 1]);
    const double ln_x = -(double)n_iter / N;
    const double grow = fmax((ln_x - S.ln_efr) - tot, 0.0);
    for (int k = 0; k < ncl; ++k) {
        const double *f = wf + k * FS;
        const double scale = sqrt(f[D + 2 * D * D]) * exp((grow + S.ln_enlarge) / D);
        double *A = S.axes + ((long)p * NS_ME + k) * D * D, *c = S.centre + ((long)p * NS_ME + k) * D;
        for (int e = lane; e < D * D; e += 64) {
            const int a = e / D, b = e - a * D;
            A[e] = b <= a ? f[D + e] * scale : 0.0;
        }
        for (int j = lane; j < D; j += 64) c[j] = f[j];
        if (lane == 0) S.elnv[(long)p * NS_ME + k] = f[D + 2 * D * D + 1] + grow;
    }
    if (lane == 0) { S.nell[p] = ncl; S.lnvol[p] = tot + grow; S.use_cube[p] = (tot + grow) >= 0.0 ? 1 : 0; }
    wave_lds_sync();
}

// ---- one workgroup per pixel: accept / replace / evidence / stop / refit -------------------
// Wave 0 owns the pixel's state and does everything sequential (replacements in the candidates' order, walks); the other
// waves help where a rejection round is wide: a segment of NS_UPD_SEG proposals has its flags, rows and likelihoods
// fetched by all threads at once, and only the candidates above the threshold the segment started with -- the only ones
// that can replace anything, the threshold never falls -- are handed to wave 0, in order.  (One wave walking 16 k flags
// in chunks of 512, three dependent global loads per chunk, was ~0.6-1.1 ms per round of the boxes' runs.)
// q indexes actlist
#define NS_UPD_THREADS 256
#define NS_UPD_SEG (4 * NS_UPD_THREADS)
__host__ __device__ inline size_t ns_upd_lds(int N) {   // doubles: live lnL | survivors' lnL | their k, row, rank (ints) | counts | control
    return (size_t)((N + 1) & ~1) + NS_UPD_SEG + (3 * NS_UPD_SEG) / 2 + 16 + 2;
}
__global__ void __launch_bounds__(NS_UPD_THREADS) NFA_UPD_ATTR ns_update_kernel(NsDev S, int n_act, int Kr, long round) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (q >= n_act) return;
    const int p = S.actlist[q];
    const int N = ns_n(S, p), NS = S.N, D = S.D, K = Kr;      // live points of this pixel; stride of the live arrays
    const long cap = ns_cap(S, p);
    const double ln_shrink = S.nlive ? log1p(-exp(-1.0 / N)) : S.ln_shrink;
    double *sL = smem;                              // live log-likelihoods of the pixel
    double *svL = sL + ((S.N + 1) & ~1);            // survivors of a segment: lnL ...
    int *svK = (int *)(svL + NS_UPD_SEG), *svRow = svK + NS_UPD_SEG, *svRank = svRow + NS_UPD_SEG;   // ... proposal, compact row, rank among the valid
    int *sCnt = svRank + NS_UPD_SEG;                // [2][4][4] valid / surviving proposals per (quarter of the segment, wave)
    double *sCtl = (double *)(sCnt + 32);           // threshold and done flag after wave 0's pass
    if (!S.active[p]) return;
    const bool timing = S.dbg != nullptr && q == 0 && tid == 0;
    long t_last = timing ? (long)wall_clock64() : 0;
    const long t_wg = S.dbg != nullptr && tid == 0 ? (long)wall_clock64() : 0;      // every pixel's workgroup: sum and maximum of its time
    double *Ll = S.Llive + (long)p * NS;
    for (int i = tid; i < N; i += NS_UPD_THREADS) sL[i] = Ll[i];
    __syncthreads();
    const bool was_walking = S.walk[p] != 0;
    int k_used = Kr;                                // what the pixel's random stream advanced by in this round
    if (was_walking && wave != 0) return;           // (a walk round is wave 0's alone: no barrier below on that path)
    auto worst_point = [&](double &lmin, int &w) {
        double mn = INFINITY;
        int ix = 0x7fffffff;
        for (int i = lane; i < N; i += 64) {
            const double v = sL[i];
            if (v < mn) { mn = v; ix = i; }         // ascending i: first occurrence per lane
        }
        ns_wave_argmin(mn, ix);
        lmin = mn; w = ix;
    };
    double Lmin, Lmax;
    int w;
    worst_point(Lmin, w);
    {
        double mx = -INFINITY;
        for (int i = lane; i < N; i += 64) mx = fmax(mx, sL[i]);
        Lmax = ns_wave_max(mx);
    }
    long it = S.n_iter[p], evals = S.n_evals[p];
    double lnZ = S.lnZ[p];
    int since = S.since_fit[p];
    bool done = false;
    // a candidate above the threshold replaces the worst live point, which dies with prior mass
    // X_it - X_(it+1); returns nothing, updates the wave-uniform state above
    // (the rows travel in registers, lane j = entry j -- D, DT <= NS_MAXD <= 64: the candidate's are loaded by the caller
    // ahead of time, the dying point's when it became the worst.  As three load-then-store copies through pointers a
    // replacement was three memory latencies long, and a round lasts as long as its busiest pixel's replacements.)
    double tw = lane < S.DT ? S.Tlive[((long)p * NS + w) * S.DT + lane] : 0.0;       // theta of the worst live point
    NS_TICK(0);                                     // prologue
    if (timing) S.dbg[8] += 1;
    auto replace = [&](double cu, double ct, double Lk) {
        const double lnw = -(double)it / N + ln_shrink;
        lnZ = ns_logaddexp(lnZ, lnw + Lmin);
        if (it < cap) {
            if (lane < S.DT) S.deadT[((long)p * S.cap + it) * S.DT + lane] = tw;
            if (lane == 0) { S.deadL[(long)p * S.cap + it] = Lmin; S.deadlnw[(long)p * S.cap + it] = lnw; }
        }
        if (lane < D) S.Ulive[((long)p * NS + w) * D + lane] = cu;
        if (lane < S.DT) S.Tlive[((long)p * NS + w) * S.DT + lane] = ct;
        if (lane == 0) { sL[w] = Lk; Ll[w] = Lk; }
        wave_lds_sync();
        it += 1; since += 1;
        Lmax = fmax(Lmax, Lk);                      // the point that left was the minimum
        const int w_old = w;
        worst_point(Lmin, w);
        tw = w == w_old ? ct : (lane < S.DT ? S.Tlive[((long)p * NS + w) * S.DT + lane] : 0.0);
        const double remain = Lmax - (double)it / N;
        done = (ns_logaddexp(lnZ, remain) - lnZ < S.tol) || it >= S.maxiter || it >= cap;
    };
    if (was_walking) {
        // ---- one Metropolis step of every walker (lane = walker), cycle end every n_steps rounds
        const int step = S.wstep[p];
        const int W = step == 0 ? min(ns_wmax(S, p), K) : S.wW[p];
        const double Lthr = step == 0 ? Lmin : S.wLthr[p];
        int tot = 0, acc = 0;
        for (int kw = lane; kw < W; kw += 64) {         // walkers lane, lane + 64, ...
            const long g = (long)q * K + kw;
            if (S.valid[g]) {
                tot += 1;
                const long row = S.slot[g];
                double Lk = S.part ? lnl_of_item(S.part, S.noise, (long)S.candpix[row], row, S.nspec) : S.candL[row];
                if (!isfinite(Lk)) Lk = S.log_zero;
                if (Lk > Lthr) {
                    acc += 1;
                    double *wu = S.wU + ((long)p * S.w_stride + kw) * D, *wt = S.wT + ((long)p * S.w_stride + kw) * S.DT;
                    for (int j = 0; j < D; ++j) wu[j] = S.candU[g * D + j];
                    for (int j = 0; j < S.DT; ++j) wt[j] = S.candT[row * S.DT + j];
                    S.wL[(long)p * S.w_stride + kw] = Lk;
                    S.wnacc[(long)p * S.w_stride + kw] += 1;
                }
            }
        }
        const long tot_w = (long)ns_wave_sum((double)tot), acc_w = (long)ns_wave_sum((double)acc);
        evals += tot_w;
        long acc_sum = S.wacc_sum[p] + acc_w, tot_sum = S.wtot_sum[p] + tot_w;
        double scale = S.wscale[p];
        int next_step = step + 1;
        if (next_step >= S.n_steps) {
            __threadfence();                        // walker states written by other lanes of this wave
            for (int k = 0; k < W && !done; ++k) {
                const int moved = __hip_atomic_load(&S.wnacc[(long)p * S.w_stride + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!moved) continue;               // never left its starting live point: not a new sample
                const double Lk = __hip_atomic_load(&S.wL[(long)p * S.w_stride + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(Lk > Lmin)) continue;
                replace(lane < D ? S.wU[((long)p * S.w_stride + k) * D + lane] : 0.0, lane < S.DT ? S.wT[((long)p * S.w_stride + k) * S.DT + lane] : 0.0, Lk);
            }
            // acceptance near one half (as dynesty's rwalk tunes it)
            if (tot_sum > 0) scale = fmin(1.0, scale * exp(((double)acc_sum / (double)tot_sum - NS_WALK_TARGET) / (0.5 * sqrt((double)D))));
            acc_sum = 0; tot_sum = 0; next_step = 0;
            // back to rejection sampling once the bound promises clearly more than a walk delivers:
            // expected acceptance X / min(V_ellipsoid, 1) > 4 / n_steps
            // (with boxes the bound is the ellipsoid's share that passes them: ln_pass, measured before the pixel left)
            if (lane == 0 && S.method == 1 && (-(double)it / N - fmin(S.lnvol[p] + S.ln_pass[p], 0.0)) > log(8.0 / ((double)S.walk_factor * S.n_steps))) S.walk[p] = 0;
        }
        if (lane == 0) {
            if (step == 0) { S.wLthr[p] = Lthr; S.wW[p] = W; }
            S.wstep[p] = next_step; S.wacc_sum[p] = acc_sum; S.wtot_sum[p] = tot_sum; S.wscale[p] = scale;
        }
    } else {
        // ---- rejection sampling: every candidate of the round is used: within a round the bound only
        // goes stale by the factor exp(-replacements / N) in volume, far cheaper than throwing evaluated
        // points away.  The proposals are walked 64 at a time; only the valid ones cost anything.
        long scanned = 0, accepted = 0, n_valid = 0;
        const int K_scan = ns_kp(S, p, K);          // the proposals this pixel drew (the flags' stride stays K)
        double Lseg = Lmin;                         // the threshold the segment starts with (wave 0's Lmin, shared below)
        if (tid == 0) { sCtl[0] = Lmin; sCtl[1] = 0.0; }
        __syncthreads();
        Lseg = sCtl[0];
        bool stop = false;                          // every thread's copy of `done`
        // NS_UPD_U x 256 proposals have their flags, rows and likelihoods in flight together (three dependent loads each:
        // fetched segment by segment they were the round); then segment by segment -- 4 x 256 proposals -- through the
        // threshold, in order
        constexpr int NS_UPD_U = 8;
        for (int kb = 0; kb < K_scan && !stop; kb += NS_UPD_U * NS_UPD_THREADS) {
            unsigned vbits = 0;
            int rows[NS_UPD_U];
            double Ls[NS_UPD_U];
#pragma unroll
            for (int u = 0; u < NS_UPD_U; ++u) {    // proposal kb + u * 256 + tid: coalesced
                const int kk = kb + u * NS_UPD_THREADS + tid;
                const bool v = kk < K_scan && S.valid[(long)q * K + kk] != 0;
                vbits |= v ? (1u << u) : 0u;
            }
#pragma unroll
            for (int u = 0; u < NS_UPD_U; ++u) rows[u] = (vbits >> u) & 1u ? S.slot[(long)q * K + kb + u * NS_UPD_THREADS + tid] : 0;
#pragma unroll
            for (int u = 0; u < NS_UPD_U; ++u) {
                const bool mine = (vbits >> u) & 1u;
                const double L = !mine ? 0.0 : S.part ? lnl_of_item(S.part, S.noise, (long)S.candpix[rows[u]], (long)rows[u], S.nspec) : S.candL[rows[u]];
                Ls[u] = isfinite(L) ? L : S.log_zero;
            }
            if (timing) { S.dbg[10] += (long)(Ls[0] != 12345.678); S.dbg[9] += 1; }      // (the loads have landed)
            NS_TICK(1);                             // flags, rows, likelihoods of NS_UPD_U x 256 proposals
#pragma unroll
            for (int sub = 0; sub < NS_UPD_U / 4; ++sub) {
                if (stop || kb + sub * NS_UPD_SEG >= K_scan) break;     // (uniform over the workgroup)
                unsigned long long mv[4], ms[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool mine = (vbits >> (4 * sub + u)) & 1u;
                    mv[u] = __ballot(mine);
                    ms[u] = __ballot(mine && Ls[4 * sub + u] > Lseg);
                    if (lane == 0) { sCnt[u * 4 + wave] = __builtin_popcountll(mv[u]); sCnt[16 + u * 4 + wave] = __builtin_popcountll(ms[u]); }
                }
                __syncthreads();
                int tot_v = 0, tot_s = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int base_v = 0, base_s = 0;     // valid / surviving proposals ahead of this (quarter, wave) in the segment
                    for (int e = 0; e < 16; ++e) {
                        const int cv = sCnt[e], cs = sCnt[16 + e];
                        if (e < u * 4 + wave) { base_v += cv; base_s += cs; }
                        if (u == 0) { tot_v += cv; tot_s += cs; }
                    }
                    if ((ms[u] >> lane) & 1ull) {
                        const unsigned long long lt = (1ull << lane) - 1ull;
                        const int e = base_s + __builtin_popcountll(ms[u] & lt);
                        svL[e] = Ls[4 * sub + u];
                        svK[e] = kb + (4 * sub + u) * NS_UPD_THREADS + tid;
                        svRow[e] = rows[4 * sub + u];
                        svRank[e] = base_v + __builtin_popcountll(mv[u] & lt) + 1;
                    }
                }
                if (tot_s == 0) {                   // nobody above the threshold: the segment only counts
                    if (wave == 0) { scanned += tot_v; evals += tot_v; n_valid += tot_v; }
                    __syncthreads();                // (sCnt is rewritten by the next segment)
                    NS_TICK(2);
                    continue;
                }
                __syncthreads();
                NS_TICK(2);                         // counts, compaction
                if (wave == 0) {
                    long seg_scanned = tot_v;
                    // (the next survivor's rows are on their way while this one is dealt with)
                    double cu_n = lane < D ? S.candU[((long)q * K + svK[0]) * D + lane] : 0.0;
                    double ct_n = lane < S.DT ? S.candT[(long)svRow[0] * S.DT + lane] : 0.0;
                    for (int i = 0; i < tot_s && !done; ++i) {
                        const double Lk = svL[i], cu = cu_n, ct = ct_n;
                        if (i + 1 < tot_s) {
                            cu_n = lane < D ? S.candU[((long)q * K + svK[i + 1]) * D + lane] : 0.0;
                            ct_n = lane < S.DT ? S.candT[(long)svRow[i + 1] * S.DT + lane] : 0.0;
                        }
                        if (!(Lk > Lmin)) continue;
                        accepted += 1;
                        replace(cu, ct, Lk);
                        if (timing) S.dbg[11] += 1;
                        if (done) seg_scanned = svRank[i];      // the candidates behind the last one are never looked at
                    }
                    scanned += seg_scanned; evals += seg_scanned; n_valid += tot_v;
                    if (lane == 0) { sCtl[0] = Lmin; sCtl[1] = done ? 1.0 : 0.0; }
                }
                __syncthreads();
                NS_TICK(3);                         // wave 0's pass over the survivors
                if (timing) S.dbg[12] += tot_s;
                Lseg = sCtl[0];
                stop = sCtl[1] != 0.0;
            }
        }
        if (wave != 0) return;
        // Walk cycles of all pixels are kept in phase (they start at rounds that are multiples of
        // n_steps): the expensive cycle end then falls into the same launch for everybody instead of
        // making every launch wait for somebody's.
        // The decision looks at all rejection rounds since the last decision point: one round of a few hundred
        // candidates is noise (a pixel that a single unlucky round sent to the walks stayed there for thousands of rounds).
        if (lane == 0) {
            long ws = S.rj_scan[p] + scanned, wa = S.rj_acc[p] + accepted, wr = S.rj_raw[p] + K_scan, wv = S.rj_val[p] + n_valid;
            if (S.k_target > 0) {
                int kn = K_scan;
                if (accepted > 2 * S.k_target) kn = max(K_scan / 2, S.K);
                else if (2 * accepted < S.k_target) kn = min(K_scan * 2, 1 << 20);
                S.Kp[p] = kn;
            }
            k_used = K_scan;
            if ((round + 1) % S.n_steps == 0) {
                // (a window that let fewer than 64 of at least 4096 proposals through has no bound worth the name: walk --
                // and be back when a walk cycle's refit has made a new one)
                if (!done && (S.method == 2 || (S.method == 1 && (ws >= 64 ? S.walk_factor * wa * S.n_steps < ws : wr >= 4096)))) {
                    S.walk[p] = 1; S.wstep[p] = 0; S.wscale[p] = 1.0; S.wacc_sum[p] = 0; S.wtot_sum[p] = 0;
                    S.ln_pass[p] = S.boxes ? log((double)(wv > 1 ? wv : 1) / (double)(wr > 1 ? wr : 1)) : 0.0;
                }
                ws = 0; wa = 0; wr = 0; wv = 0;
            }
            S.rj_scan[p] = ws; S.rj_acc[p] = wa; S.rj_raw[p] = wr; S.rj_val[p] = wv;
        }
    }
    if (lane == 0) {
        S.n_iter[p] = it; S.n_evals[p] = evals; S.lnZ[p] = lnZ; S.cand_base[p] += k_used;
        if (done) S.active[p] = 0;
    }
    // A refit costs ~100 us and pixels are in lock-step: rejection-mode pixels refit only in every
    // fourth round, so that three launches out of four do not wait for anybody's refit (walking
    // pixels refit at their common cycle end).
    // (the fit itself is ns_refit_kernel's, launched behind this one in the rounds where a pixel can be due: it needs
    // four times the registers of everything above, and a round's update should not carry them)
    const bool due = !done && since >= ns_upd(S, p) && (was_walking || (round + 1) % S.refit_every == 0);
    if (lane == 0) { S.since_fit[p] = since; S.refit_due[p] = due ? 1 : 0; }
    NS_TICK(4);                                     // the tail
    if (S.dbg != nullptr && tid == 0) {
        const long dt = (long)wall_clock64() - t_wg;
        atomicAdd((unsigned long long *)&S.dbg[13], (unsigned long long)dt);
        atomicAdd((unsigned long long *)&S.dbg[14], 1ull);
        atomicMax((unsigned long long *)&S.dbg[15], (unsigned long long)dt);
        if (was_walking) { atomicAdd((unsigned long long *)&S.dbg[5], (unsigned long long)dt); atomicAdd((unsigned long long *)&S.dbg[6], 1ull); }
        int bin = 0;
        while ((1l << bin) * 100 < dt && bin < 15) ++bin;                   // 1, 2, 4 ... us
        atomicAdd((unsigned long long *)&S.dbg[16 + bin], 1ull);
        atomicAdd((unsigned long long *)&S.dbg[32 + bin], (unsigned long long)dt);
        const long nrep = it - S.n_iter[p] + 0;                             // (n_iter was written above: so this is 0; see dbg[48+])
        (void)nrep;
    }
}

// The bounds of the pixels that are due (ns_update_kernel marks them; `direct`: all pixels, before the first round).
// One workgroup per pixel: one wave where the bound may be several ellipsoids, NS_REFIT_THREADS threads for the one-ellipsoid
// fit with its shear and boxes (a lone wave's chain of LDS latencies was 2.2 ms per refit).
// LDS: [8: reductions][D*D][D][live points: N*D][fit slots and labels | the shear's scratch].
__global__ void __launch_bounds__(NS_REFIT_THREADS) ns_refit_kernel(NsDev S, int n_act, int direct) {
    extern __shared__ __attribute__((aligned(16))) double smem_all[];
    double *smem = smem_all + 8;
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= n_act) return;
    const int p = direct ? q : S.actlist[q];
    if (!direct && (!S.active[p] || !S.refit_due[p])) return;
    const int NS = S.N, D = S.D;
    double *sA = smem;                              // D*D
    double *sc = sA + D * D;                        // D
    double *sd = S.stage_live ? sc + ((D + 1) & ~1) : nullptr;   // N*D live points
    double *wf = sd ? sd + (long)NS * D : nullptr;  // several ellipsoids: the fit slots, then the points' labels
    int *lab = (int *)(wf + (NS_ME + 2) * ns_me_slot(D));
    const long it = direct ? 0 : S.n_iter[p];
    if (S.multi) {
        switch (D) {                                // (D <= NS_ME_MAXD here)
        case 1: ns_refit_multi<1>(S, p, it, sd, lab, wf, lane); break;
        case 2: ns_refit_multi<2>(S, p, it, sd, lab, wf, lane); break;
        case 3: ns_refit_multi<3>(S, p, it, sd, lab, wf, lane); break;
        case 4: ns_refit_multi<4>(S, p, it, sd, lab, wf, lane); break;
        case 5: ns_refit_multi<5>(S, p, it, sd, lab, wf, lane); break;
        default: ns_refit_multi<6>(S, p, it, sd, lab, wf, lane); break;
        }
    } else {
        ns_refit(S, p, it, sA, sc, sd, wf, (int)threadIdx.x, (int)blockDim.x, smem_all);     // (wf: the shear's scratch where the bound is one ellipsoid)
    }
    if (lane == 0) { S.since_fit[p] = 0; S.refit_due[p] = 0; }
}

// ---- results ---------------------------------------------------------------------------------
// dead points of all pixels, packed: rows [off[p], off[p + 1]) of the outputs = the first off[p+1] - off[p] dead
// points of pixel p (one copy off the device instead of three per pixel)
__global__ void ns_pack_dead_kernel(NsDev S, const long *__restrict__ off, double *__restrict__ outT,
                                    double *__restrict__ outL, double *__restrict__ outW) {
    const int DT = S.DT;
    for (long p = blockIdx.y; p < S.P; p += gridDim.y) {       // (the grid's second dimension ends at 65535)
        const long n = off[p + 1] - off[p];
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n * DT; e += (long)gridDim.x * blockDim.x)
            outT[off[p] * DT + e] = S.deadT[p * S.cap * DT + e];
        for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long)gridDim.x * blockDim.x) {
            outL[off[p] + r] = S.deadL[p * S.cap + r];
            outW[off[p] + r] = S.deadlnw[p * S.cap + r];
        }
    }
}

// The posterior samples of every pixel in the layout of the result (the reference's post_equal_weights-like table,
// core.pyx:627-687): rows [off[p], off[p + 1]) of out = pixel p's dead points (the first off[p+1] - off[p] - nlive_p of
// them) followed by its live points; a row = theta[DT], -2 lnL, ln(prior mass x likelihood) -- lnw + lnL for a dead point,
// lnL + live_off[p] for a live one (live_off[p] = -n_iter / nlive - ln nlive, from the host: the host turns the last
// column into weights).  One copy off the device and no copy on the host: a pixel's table is a view into `out`.
__global__ void ns_pack_post_kernel(NsDev S, const long *__restrict__ off, const double *__restrict__ live_off, double *__restrict__ out) {
    const int DT = S.DT, W = DT + 2;
    for (long p = blockIdx.y; p < S.P; p += gridDim.y) {
        const long n = off[p + 1] - off[p], nl = ns_n(S, (int)p), nd = n - nl;
        double *o = out + off[p] * W;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n * W; e += (long)gridDim.x * blockDim.x) {
            const long r = e / W;
            const int c = (int)(e - r * W);
            double v;
            if (r < nd) {
                const double L = S.deadL[p * S.cap + r];
                v = c < DT ? S.deadT[(p * S.cap + r) * DT + c] : c == DT ? -2.0 * L : S.deadlnw[p * S.cap + r] + L;
            } else {
                const long i = r - nd;
                const double L = S.Llive[p * S.N + i];
                v = c < DT ? S.Tlive[(p * S.N + i) * DT + c] : c == DT ? -2.0 * L : L + live_off[p];
            }
            o[e] = v;
        }
    }
}

// What a result needs from a pixel's table, formed where the table is: the last column turned into weights in place
// (exp(ln(prior mass x likelihood) - lnZ)), and per pixel stats[p] = [lnZ, lnZ of the dead points alone, information H,
// largest lnL, largest lnL of the live points, sum of the weights] + weighted mean[DT] + weighted raw second moment[DT] +
// theta of the largest likelihood[DT] + theta of the largest weight[DT].  One workgroup per pixel.
#define NS_FIN_THREADS 256
__device__ __forceinline__ double ns_fin_sum(double v, double *sred) {
    v = ns_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < NS_FIN_THREADS / 64; ++k) t += sred[k];
    return t;
}
__device__ __forceinline__ double ns_fin_max(double v, double *sred) {
    v = ns_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = sred[0];
    for (int k = 1; k < NS_FIN_THREADS / 64; ++k) t = fmax(t, sred[k]);
    return t;
}
__global__ void __launch_bounds__(NS_FIN_THREADS) ns_finish_kernel(NsDev S, const long *__restrict__ off, double *__restrict__ out, double *__restrict__ stats) {
    __shared__ double sred[NS_FIN_THREADS / 64];
    __shared__ unsigned long long sarg[2];
    const int DT = S.DT, W = DT + 2, tid = (int)threadIdx.x;
    const long p = blockIdx.x;
    const long n = off[p + 1] - off[p], nl = ns_n(S, (int)p), nd = n - nl;
    double *o = out + off[p] * W;
    double *st = stats + p * (6 + 4 * DT);
    // maxima of the log weights (dead, live) and of lnL
    double md = -INFINITY, ml = -INFINITY, lmax = -INFINITY, llive = -INFINITY;
    for (long r = tid; r < n; r += NS_FIN_THREADS) {
        const double lw = o[r * W + DT + 1], L = -0.5 * o[r * W + DT];
        if (r < nd) md = fmax(md, lw); else { ml = fmax(ml, lw); llive = fmax(llive, L); }
        lmax = fmax(lmax, L);
    }
    md = ns_fin_max(md, sred); ml = ns_fin_max(ml, sred); lmax = ns_fin_max(lmax, sred); llive = ns_fin_max(llive, sred);
    double sd = 0.0, sl = 0.0;
    for (long r = tid; r < n; r += NS_FIN_THREADS) {
        const double lw = o[r * W + DT + 1];
        if (r < nd) sd += isfinite(md) ? exp(lw - md) : 0.0; else sl += isfinite(ml) ? exp(lw - ml) : 0.0;
    }
    sd = ns_fin_sum(sd, sred); sl = ns_fin_sum(sl, sred);
    const double lnz_dead = nd > 0 ? (isfinite(md) ? md + log(sd) : md) : -INFINITY;
    const double lnz_live = isfinite(ml) ? ml + log(sl) : ml;
    const double lnz = ns_logaddexp(lnz_dead, lnz_live);
    // weights in place, information, sum of the weights; rows of the largest likelihood and of the largest weight
    double H = 0.0, sw = 0.0, bestL = -INFINITY, bestw = -1.0;
    long ibest = 0, imap = 0;
    for (long r = tid; r < n; r += NS_FIN_THREADS) {
        const double lw = o[r * W + DT + 1], L = -0.5 * o[r * W + DT];
        const double w = exp(lw - lnz);
        o[r * W + DT + 1] = w;
        if (w > 0.0) H += w * (L - lnz);
        sw += w;
        if (L > bestL) { bestL = L; ibest = r; }
        if (w > bestw) { bestw = w; imap = r; }
    }
    H = ns_fin_sum(H, sred); sw = ns_fin_sum(sw, sred);
    // (arg-maxima over the workgroup: the value's bits -- order-preserving for the finite values that matter -- with the row packed below)
    if (tid == 0) { sarg[0] = 0ull; sarg[1] = 0ull; }
    __syncthreads();
    const double gL = ns_fin_max(bestL, sred), gw = ns_fin_max(bestw, sred);
    if (bestL == gL) atomicMax(&sarg[0], ~(unsigned long long)ibest);        // the FIRST row among equals: largest complement
    if (bestw == gw) atomicMax(&sarg[1], ~(unsigned long long)imap);
    __syncthreads();
    const long rb = (long)~sarg[0], rm = (long)~sarg[1];
    if (tid == 0) { st[0] = lnz; st[1] = lnz_dead; st[2] = H; st[3] = lmax; st[4] = llive; st[5] = sw; }
    for (int j = tid; j < DT; j += NS_FIN_THREADS) { st[6 + 2 * DT + j] = o[rb * W + j]; st[6 + 3 * DT + j] = o[rm * W + j]; }
    // weighted first and second moments of every column, taken about the row of the largest weight (a point inside the
    // posterior's bulk: raw moments cancel where |mean| >> sigma): stored are the mean, sum w t, and sum w (t - c)^2
    for (int j = 0; j < DT; ++j) {
        const double c = o[rm * W + j];
        double m1 = 0.0, m2 = 0.0;
        for (long r = tid; r < n; r += NS_FIN_THREADS) {
            const double w = o[r * W + DT + 1], d = o[r * W + j] - c;
            m1 += w * d; m2 += w * (d * d);
        }
        m1 = ns_fin_sum(m1, sred); m2 = ns_fin_sum(m2, sred);
        if (tid == 0) { st[6 + j] = m1 + c * sw; st[6 + DT + j] = m2; }
    }
}

// ---- host side -----------------------------------------------------------------------------
#ifndef NS_KMAX
#define NS_KMAX 65536           // most proposals one pixel gets in a round
#endif
#define NS_PARTS 4              // at most this many groups of pixels, each on its own stream lane
struct nfa_sampler {
    nfa_runner *r = nullptr;
    NsDev d = {};
    long b_target = 0;          // candidates per round the sampler aims for (all pixels together)
    int *d_pixmap = nullptr, *d_actlist = nullptr, *d_livepix = nullptr, *d_fmap = nullptr;
    int *d_nlive = nullptr, *d_updp = nullptr;
    long *d_capp = nullptr;
    double *d_frames = nullptr;
    int set_frames = -2;        // nfa_sampler_set_boxes: -2 = the default (NS_FRAMES above NS_ME_MAXD sampled dimensions), -1 = no boxes
    double set_margin = 0.0;    // ... 0 = the default
    double set_pairs = -1.0;    // nfa_sampler_set_pairs: < 0 = the default, 0 = off, >= 1 = the safety factor on the ellipses' areas
    double set_shear = -1.0;    // nfa_sampler_set_shear: < 0 = the default (engine option sampler_shear_pct), 0 = off, >= 1 = the safety factor
    int *d_sh_mono = nullptr, *d_sh_start = nullptr;
    std::vector<int> fm;        // the sampled dimensions' slots
    size_t k_alloc = 0;         // proposal rows allocated per pixel
    long ratio_max = NS_RATIO_MAX, kmax = NS_KMAX;   // options sampler_ratio_max / sampler_kmax as they stood at creation
    long raw_sum = 0, val_sum = 0;   // proposals drawn / evaluated since the last look at the active pixels
    std::vector<int> h_nlive;   // per-pixel live points (empty: d.N for everybody)
    int max_ell = 0;            // nfa_sampler_set_ellipsoids (0: the default)
    std::vector<int> h_active, h_act;
    long rounds = 0;
    int  n_act = 0, check_every = 8;
    size_t lds = 0, lds_refit = 0;
    bool ran = false;
    // host memory the device writes (mapped): per part, [rows of the round][sequence number of the round], 16 bytes each
    unsigned long long *h_pub = nullptr, *d_pub = nullptr;
    unsigned long long seq = 0;  // sequence number of the last proposing launch
};

extern "C" {

int nfa_sampler_destroy(nfa_sampler *s) {
    if (!s) return NFA_OK;
    NsDev &d = s->d;
    if (d.dbg) {
        long h[64];
        if (hipMemcpy(h, d.dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[ns timing, first listed pixel of part 0] update launches %ld: prologue %.1f us, loads %.1f us (%ld batches), counts+compaction %.1f us, "
                    "wave-0 pass %.1f us (%ld survivors, %ld replacements), tail %.1f us per launch\n", h[8], 0.01 * h[0] / h[8], 0.01 * h[1] / h[8], h[9],
                    0.01 * h[2] / h[8], 0.01 * h[3] / h[8], h[12], h[11], 0.01 * h[4] / h[8]);
        if (hipMemcpy(h, d.dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[14] > 0)
            fprintf(stderr, "[ns timing, all update workgroups] %ld: mean %.1f us, longest %.1f us; walking ones %ld: mean %.1f us\n", h[14], 0.01 * h[13] / h[14],
                    0.01 * h[15], h[6], h[6] ? 0.01 * h[5] / h[6] : 0.0);
        if (h[57]) fprintf(stderr, "[ns timing, refit workgroup 0] %ld refits: standardise %.1f, Gram %.1f, Cholesky %.1f, back substitution %.1f, w %.1f, covariance %.1f, "
                           "its Cholesky + axis box %.1f, y %.1f, frames' boxes %.1f us\n", h[57], 0.01 * h[48] / h[57], 0.01 * h[49] / h[57], 0.01 * h[50] / h[57], 0.01 * h[51] / h[57],
                           0.01 * h[52] / h[57], 0.01 * h[53] / h[57], 0.01 * h[54] / h[57], 0.01 * h[55] / h[57], 0.01 * h[56] / h[57]);
        for (int b = 0; b < 16; ++b) if (h[16 + b]) fprintf(stderr, "    <= %5ld us: %9ld workgroups, %8.1f ms in all\n", 1l << b, h[16 + b], 1e-5 * h[32 + b]);
        (void)hipFree(d.dbg);
    }
    void *ptrs[] = {d.Ulive, d.Tlive, d.Llive, d.centre, d.axes, d.n_iter, d.n_evals, d.cand_base, d.lnZ, d.active, d.use_cube,
                    d.since_fit, d.refit_due, d.deadT, d.deadL, d.deadlnw, d.candU, d.candT, d.candL, d.candpix, d.valid, d.slot, d.count,
                    d.walk, d.wstep, d.wW, d.wscale, d.wLthr, d.wacc_sum, d.wtot_sum, d.wU, d.wT, d.wL, d.wnacc, d.lnvol, d.elnv, d.nell,
                    s->d_pixmap, s->d_actlist, s->d_livepix, s->d_fmap, s->d_nlive, s->d_updp, s->d_capp,
                    s->d_frames, d.ubox, d.fbox, d.rj_scan, d.rj_acc, d.rj_raw, d.rj_val, d.ln_pass,
                    s->d_sh_mono, s->d_sh_start, d.sh_mu, d.sh_sg, d.sh_beta, d.Kp, d.pair_tab};
    for (void *p : ptrs) (void)hipFree(p);
    if (s->h_pub) (void)hipHostFree(s->h_pub);
    delete s;
    return NFA_OK;
}

// Device-resident nested sampling of n_pix pixels of the runner's spectra set in lock-step.
// pix[n_pix] = cube pixel per run (NULL: 0..n_pix-1 must all be the runner's pixel 0 -> only
// n_pix = 1 makes sense then).  cap_iter = dead-point slots per pixel (a run stops there).
// batch_target = candidates per round over all pixels the sampler aims for.
int nfa_sampler_create(nfa_sampler **out, nfa_runner *r, const int32_t *pix, int64_t n_pix, int nlive,
                       int n_cand, int64_t batch_target, int64_t cap_iter, const int32_t *free_mask) {
    if (!out || !r) return fail(NFA_ERR_ARG, "null argument");
    if (!r->pr) return fail(NFA_ERR_STATE, "runner has no priors (predict-only)");
    if (n_pix < 1 || n_pix > (1 << 24)) return fail(NFA_ERR_ARG, "n_pix out of range");
    if (nlive < r->ndim + 2 || nlive > 8192) return fail(NFA_ERR_ARG, "nlive must be in ndim+2..8192");
    if (n_cand < 1 || n_cand > 1024) return fail(NFA_ERR_ARG, "n_cand must be in 1..1024");
    if (cap_iter < 1) return fail(NFA_ERR_ARG, "cap_iter must be >= 1");
    if (batch_target < 1 || batch_target > (1 << 26)) return fail(NFA_ERR_ARG, "batch_target out of range");
    if (r->ndim > NS_MAXD) return fail(NFA_ERR_ARG, "too many dimensions");
    int rc = check_pix(r, pix, n_pix); if (rc) return rc;
    nfa_sampler *s = new nfa_sampler();
    s->r = r;
    NsDev &d = s->d;
    // sampled dimensions: the unit-cube slots the likelihood depends on (free_mask[ndim], NULL = all);
    // a constant or duplicated parameter's slot is integrated out exactly by not sampling it
    std::vector<int> fm;
    for (int j = 0; j < r->ndim; ++j) if (!free_mask || free_mask[j]) fm.push_back(j);
    if (fm.empty()) { delete s; return fail(NFA_ERR_ARG, "no free dimension to sample"); }
    d.P = (int)n_pix; d.N = nlive; d.D = (int)fm.size(); d.DT = r->ndim; d.K = n_cand; d.cap = (long)cap_iter;
    const size_t P = (size_t)n_pix, N = (size_t)nlive, D = fm.size(), DT = (size_t)r->ndim, C = (size_t)cap_iter;
    s->b_target = std::max<long>((long)n_pix * n_cand, (long)batch_target);
    // rows of the candidate buffers: n_act * Kr <= max(b_target, n_act * K) <= b_target -- times NS_RATIO_MAX where boxes
    // may veto proposals for free (one-ellipsoid bounds: more than NS_ME_MAXD sampled dimensions, or on request)
    // (the two process options are read ONCE, here: the buffers are sized for them, and a value changed while the sampler
    // lives must not outrun the buffers)
    const size_t ratio_alloc = g_eng.sampler_ratio_max > 0 ? g_eng.sampler_ratio_max : NS_RATIO_MAX;
    s->ratio_max = (long)ratio_alloc;
    s->kmax = g_eng.sampler_kmax > 0 ? g_eng.sampler_kmax : NS_KMAX;
    const size_t K = ((size_t)s->b_target * ratio_alloc + P - 1) / P;       // so that P * K >= ratio_max * b_target
    s->k_alloc = K;
    std::vector<int> pm(P);
    for (size_t p = 0; p < P; ++p) pm[p] = pix ? pix[p] : 0;
#define NS_ALLOC(ptr, type, count) \
    if (hipMalloc((void **)&(ptr), sizeof(type) * (count)) != hipSuccess) { \
        nfa_sampler_destroy(s); return fail(NFA_ERR_DEVICE, "out of device memory for the sampler state"); }
    NS_ALLOC(s->d_pixmap, int, P); NS_ALLOC(s->d_actlist, int, P); NS_ALLOC(s->d_livepix, int, P * N);
    NS_ALLOC(s->d_fmap, int, D);
    NS_ALLOC(d.Ulive, double, P * N * D); NS_ALLOC(d.Tlive, double, P * N * DT); NS_ALLOC(d.Llive, double, P * N);
    NS_ALLOC(d.centre, double, P * NS_ME * D); NS_ALLOC(d.axes, double, P * NS_ME * D * D);
    NS_ALLOC(d.elnv, double, P * NS_ME); NS_ALLOC(d.nell, int, P);
    NS_ALLOC(d.n_iter, long, P); NS_ALLOC(d.n_evals, long, P); NS_ALLOC(d.cand_base, long, P); NS_ALLOC(d.lnZ, double, P);
    NS_ALLOC(d.active, int, P); NS_ALLOC(d.since_fit, int, P); NS_ALLOC(d.use_cube, int, P); NS_ALLOC(d.refit_due, int, P);
    NS_ALLOC(d.deadT, double, P * C * DT); NS_ALLOC(d.deadL, double, P * C); NS_ALLOC(d.deadlnw, double, P * C);
    NS_ALLOC(d.candU, double, P * K * D); NS_ALLOC(d.candT, double, P * K * DT); NS_ALLOC(d.candL, double, P * K);
    NS_ALLOC(d.candpix, int, P * K); NS_ALLOC(d.valid, int, P * K); NS_ALLOC(d.slot, int, P * K); NS_ALLOC(d.count, int, NS_PARTS);
    NS_ALLOC(d.walk, int, P); NS_ALLOC(d.wstep, int, P); NS_ALLOC(d.wW, int, P); NS_ALLOC(d.wscale, double, P);
    NS_ALLOC(d.wLthr, double, P); NS_ALLOC(d.wacc_sum, long, P); NS_ALLOC(d.wtot_sum, long, P);
    d.w_fixed = g_eng.sampler_walkers;
    d.w_stride = d.w_fixed > 0 ? d.w_fixed : ns_walkers_for(N);      // (a pixel's own count can only be smaller than N)
    NS_ALLOC(d.wU, double, P * d.w_stride * D); NS_ALLOC(d.wT, double, P * d.w_stride * DT); NS_ALLOC(d.wL, double, P * d.w_stride);
    NS_ALLOC(d.wnacc, int, P * d.w_stride); NS_ALLOC(d.lnvol, double, P);
    NS_ALLOC(d.ubox, double, P * D * 2); NS_ALLOC(d.fbox, double, P * (NS_FRAMES_MAX + 1) * D * 2);
    NS_ALLOC(d.rj_scan, long, P); NS_ALLOC(d.rj_acc, long, P); NS_ALLOC(d.rj_raw, long, P); NS_ALLOC(d.rj_val, long, P);
    NS_ALLOC(d.ln_pass, double, P);
    NS_ALLOC(s->d_frames, double, (size_t)NS_FRAMES_MAX * D * D);
    NS_ALLOC(s->d_sh_mono, int, NS_SHEAR_MMAX * 2); NS_ALLOC(s->d_sh_start, int, D);
    NS_ALLOC(d.Kp, int, P); NS_ALLOC(d.pair_tab, double, P * (D * (D - 1) / 2 + 1) * 5);
    NS_ALLOC(d.sh_mu, double, P * D); NS_ALLOC(d.sh_sg, double, P * D); NS_ALLOC(d.sh_beta, double, P * D * NS_SHEAR_MMAX);
    s->fm = fm;
    if (getenv("NFA_NS_TIMING")) { NS_ALLOC(d.dbg, long, 64); HIP_TRY(hipMemset(d.dbg, 0, sizeof(long) * 64)); }
#undef NS_ALLOC
    HIP_TRY(hipMemcpy(s->d_pixmap, pm.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_fmap, fm.data(), sizeof(int) * D, hipMemcpyHostToDevice));
    d.pixmap = s->d_pixmap; d.actlist = s->d_actlist; d.fmap = s->d_fmap;
    *out = s;
    return NFA_OK;
}

// Pixels with numbers of live points of their own (call between create and begin): nlive[p] in ndim+2 .. the nlive given
// to nfa_sampler_create (the stride of the live arrays), cap[p] in 1 .. cap_iter dead-point slots, upd[p] >= 1
// replacements between refits.  The cube driver's pixels differ by a few live points each (main.py:445-447); one
// lock-step group for all of them instead of a group per count.
int nfa_sampler_set_pixel_nlive(nfa_sampler *s, const int32_t *nlive, const int64_t *cap, const int32_t *upd) {
    if (!s || !nlive || !cap || !upd) return fail(NFA_ERR_ARG, "null argument");
    if (s->ran) return fail(NFA_ERR_STATE, "call nfa_sampler_set_pixel_nlive before nfa_sampler_begin");
    NsDev &d = s->d;
    const size_t P = (size_t)d.P;
    std::vector<int> hn(P), hu(P);
    std::vector<long> hc(P);
    for (size_t p = 0; p < P; ++p) {
        if (nlive[p] < s->r->ndim + 2 || nlive[p] > d.N) return fail(NFA_ERR_ARG, "a pixel's nlive must be in ndim+2 .. the sampler's nlive");
        if (cap[p] < 1 || cap[p] > d.cap || upd[p] < 1) return fail(NFA_ERR_ARG, "a pixel's cap / upd is out of range");
        hn[p] = nlive[p]; hc[p] = (long)cap[p]; hu[p] = upd[p];
    }
    if (!s->d_nlive) {
        HIP_TRY(hipMalloc((void **)&s->d_nlive, sizeof(int) * P));
        HIP_TRY(hipMalloc((void **)&s->d_updp, sizeof(int) * P));
        HIP_TRY(hipMalloc((void **)&s->d_capp, sizeof(long) * P));
    }
    HIP_TRY(hipMemcpy(s->d_nlive, hn.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_updp, hu.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_capp, hc.data(), sizeof(long) * P, hipMemcpyHostToDevice));
    d.nlive = s->d_nlive; d.updp = s->d_updp; d.capp = s->d_capp;
    s->h_nlive = hn;
    return NFA_OK;
}

// Free rejections by boxes (one-ellipsoid bounds): n_frames rotated frames beside the unit cube's axes and the ellipsoid's
// own (-2: the default -- none; -1: no boxes; 0..64; NS_FRAMES = 32 is the measured choice), margin = the
// factor c of a face's distance beyond the extreme live point (0: the default, NS_MARGIN_C).  Before nfa_sampler_begin.
int nfa_sampler_set_boxes(nfa_sampler *s, int n_frames, double margin) {
    if (!s || n_frames < -2 || n_frames > NS_FRAMES_MAX || !(margin >= 0.0) || margin > 100.0) return fail(NFA_ERR_ARG, "boxes: frames -2 (default), -1 (none) .. 64; margin >= 0");
    if (s->ran) return fail(NFA_ERR_STATE, "call nfa_sampler_set_boxes before nfa_sampler_begin");
    s->set_frames = n_frames;
    s->set_margin = margin;
    return NFA_OK;
}

// The shear in front of a one-ellipsoid bound (10 or 15 sampled dimensions = all five free parameters of two or three
// components; elsewhere the call is accepted and changes nothing): enlarge = the safety factor on the sheared ellipsoid's
// enclosing volume (>= 1; NS_SHEAR_ENLARGE = 3: profiles/r05/sampler_bias.txt), 0 = off, < 0 = the default.  Before nfa_sampler_begin.
int nfa_sampler_set_shear(nfa_sampler *s, double enlarge) {
    if (!s || (enlarge > 0.0 && enlarge < 1.0) || enlarge > 1e6 || enlarge != enlarge) return fail(NFA_ERR_ARG, "shear: 0 (off), < 0 (default) or a safety factor >= 1");
    if (s->ran) return fail(NFA_ERR_STATE, "call nfa_sampler_set_shear before nfa_sampler_begin");
    s->set_shear = enlarge;
    return NFA_OK;
}

// The pair ellipses (with the shear and the boxes): enlarge = the safety factor on their areas (>= 1; NS_PAIRS_ENLARGE = 2),
// 0 = off, < 0 = the default.  Before nfa_sampler_begin.
int nfa_sampler_set_pairs(nfa_sampler *s, double enlarge) {
    if (!s || (enlarge > 0.0 && enlarge < 1.0) || enlarge > 1e6 || enlarge != enlarge) return fail(NFA_ERR_ARG, "pairs: 0 (off), < 0 (default) or a safety factor >= 1");
    if (s->ran) return fail(NFA_ERR_STATE, "call nfa_sampler_set_pairs before nfa_sampler_begin");
    s->set_pairs = enlarge;
    return NFA_OK;
}

int nfa_sampler_set_ellipsoids(nfa_sampler *s, int max_ellipsoids) {
    if (!s || max_ellipsoids < 0 || max_ellipsoids > NS_ME) return fail(NFA_ERR_ARG, "ellipsoids per pixel: 0 (default) .. 4");
    if (s->ran) return fail(NFA_ERR_STATE, "call nfa_sampler_set_ellipsoids before nfa_sampler_begin");
    s->max_ell = max_ellipsoids;
    return NFA_OK;
}

// tol, efr, seed, maxiter as run_multinest (core.pyx:727-744); upd = replacements between
// ellipsoid refits; check_every = rounds between two looks at the set of active pixels.
// nfa_sampler_begin draws and evaluates the live points and fits the first ellipsoids;
// nfa_sampler_advance runs up to max_chunks groups of check_every rounds (0 = until every pixel
// has stopped) and reports how many pixels are still running, so the caller can show progress
// or give up; nfa_sampler_run = begin + advance to the end.
int nfa_sampler_begin(nfa_sampler *s, double tol, double efr, int64_t seed, int64_t maxiter, int upd,
                      double log_zero, int check_every, double enlarge, int method, int n_steps) {
    if (!s) return fail(NFA_ERR_ARG, "null sampler");
    if (!(tol > 0) || !(efr > 0 && efr <= 1) || maxiter < 0 || upd < 1 || check_every < 1 || !(enlarge >= 1) ||
        method < 0 || method > 2 || n_steps < 1)
        return fail(NFA_ERR_ARG, "bad sampler options");
    nfa_runner *r = s->r;
    RUNNER_LOCK(r);
    { int rcf = sync_all_lanes(r); if (rcf) return rcf; }    // the lanes are the sampler's from here on
    NsDev &d = s->d;
    const int P = d.P, N = d.N, D = d.D;
    d.ln_enlarge = log(enlarge);
    d.method = method; d.n_steps = n_steps;
    d.tol = tol; d.maxiter = (long)maxiter; d.upd = upd; d.seed = (uint64_t)seed; d.log_zero = log_zero;
    d.ln_shrink = log1p(-exp(-1.0 / N));
    d.ln_efr = log(efr);
    d.ln_vball = 0.5 * D * log(M_PI) - lgamma(0.5 * D + 1.0);
    s->check_every = check_every;
    hipStream_t st = r->lanes[0];
    HIP_TRY(hipMemsetAsync(d.n_iter, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.cand_base, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.since_fit, 0, sizeof(int) * P, st));
    HIP_TRY(hipMemsetAsync(d.refit_due, 0, sizeof(int) * P, st));
    HIP_TRY(hipMemsetAsync(d.walk, 0, sizeof(int) * P, st));
    HIP_TRY(hipMemsetAsync(d.wstep, 0, sizeof(int) * P, st));
    HIP_TRY(hipMemsetAsync(d.wacc_sum, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.wtot_sum, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.rj_scan, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.rj_acc, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.rj_raw, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.rj_val, 0, sizeof(long) * P, st));
    HIP_TRY(hipMemsetAsync(d.ln_pass, 0, sizeof(double) * P, st));
    // (a pixel starts with a small share and doubles it while its rounds accept little: started at the round's Kr, a run of
    // two pixels drew 65 k proposals per pixel in its first round, where every second one is accepted, and halved from there)
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d.Kp, NS_KP_START, P, st));
    d.k_target = g_eng.sampler_ktarget >= 0 ? g_eng.sampler_ktarget : NS_K_TARGET;
    {   // live points
        const long tot = (long)P * N;
        hipLaunchKernelGGL(ns_init_live_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d, s->d_livepix);
        HIP_TRY(hipGetLastError());
        int rc = run_batch(r, s->d_livepix, d.Tlive, d.Llive, nullptr, (int64_t)P * N, true, 0, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(ns_sanitize_kernel, dim3((unsigned)(((long)P * N + 255) / 256)), dim3(256), 0, st,
                           d.Llive, (long)P * N, log_zero);
        HIP_TRY(hipGetLastError());
    }
    std::vector<long> h_evals((size_t)P, (long)N);
    for (size_t p = 0; p < s->h_nlive.size(); ++p) h_evals[p] = s->h_nlive[p];      // a pixel's own live points were its first evaluations
    std::vector<double> h_lnz((size_t)P, -INFINITY);
    s->h_active.assign((size_t)P, maxiter > 0 ? 1 : 0);
    HIP_TRY(hipMemcpyAsync(d.n_evals, h_evals.data(), sizeof(long) * P, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d.lnZ, h_lnz.data(), sizeof(double) * P, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d.active, s->h_active.data(), sizeof(int) * P, hipMemcpyHostToDevice, st));
    s->lds = sizeof(double) * ns_upd_lds(N);                                // the update workgroup: the live log-likelihoods, a segment's survivors
    s->lds_refit = sizeof(double) * (8 + (size_t)D * D + (size_t)((D + 1) & ~1));  // the refit workgroup: reductions, ...
    d.stage_live = (size_t)N * D * sizeof(double) <= 96 * 1024 ? 1 : 0;
    d.refit_every = g_eng.sampler_refit_every;
    // When does a pixel give up rejection sampling for constrained walks?  Measured on config 5 (profiles/r03/
    // sweep_walk_factor.txt): with ten sampled dimensions the walks win from an acceptance of ~1 in 2 n_steps down (the
    // run takes 7.0-7.2 s for factors 1..4, 8.6 s at 32, 10.9 s at 64); with five they hardly ever do -- a rejection
    // round is one large batch, a walk cycle n_steps small ones, and the run goes from 1.16 s (factor 2) to 0.84 s (64;
    // rejection only: 0.79 s).  The walks stay as the way out of a bound that has become hopeless.
    d.walk_factor = g_eng.sampler_walk_factor > 0 ? g_eng.sampler_walk_factor : (D <= NS_WALK_LOWD ? NS_WALK_FACTOR_LOWD : NS_WALK_FACTOR);
    if (d.stage_live) s->lds_refit += sizeof(double) * (size_t)N * D;       // ... the live points ...
    d.max_ell = s->max_ell > 0 ? s->max_ell : (g_eng.sampler_ellipsoids == 1 ? 1 : NS_ME);
    d.multi = (d.stage_live && D <= NS_ME_MAXD && d.max_ell > 1) ? 1 : 0;
    if (d.multi) s->lds_refit += sizeof(double) * (size_t)((NS_ME + 2) * ns_me_slot(D)) + sizeof(int) * (size_t)((N + 3) & ~3);   // ... fit slots, labels
    {   // the shear: one-ellipsoid bounds of all five free parameters of two or three components
        const double enl = s->set_shear >= 0.0 ? s->set_shear : g_eng.sampler_shear_pct >= 0 ? 0.01 * g_eng.sampler_shear_pct : NS_SHEAR_ENLARGE;
        const int nc = D / 5;
        bool shape = (D == 10 || D == 15) && d.DT == 6 * nc;
        for (int j = 0; shape && j < D; ++j) shape = (s->fm[(size_t)j] % nc) == (j % nc);
        d.shear = (enl >= 1.0 && shape && !d.multi && d.stage_live) ? 1 : 0;
        d.sh_M = 0;
        if (d.shear) {
            std::vector<int> mono, start;
            ns_shear_monomials(D, nc, mono, start);
            d.sh_M = (int)mono.size() / 2;
            if (d.sh_M > NS_SHEAR_MMAX) return fail(NFA_ERR_STATE, "shear: too many monomials");
            HIP_TRY(hipMemcpyAsync(s->d_sh_mono, mono.data(), sizeof(int) * mono.size(), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(s->d_sh_start, start.data(), sizeof(int) * start.size(), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            d.ln_enlarge_shear = log(enl);
            s->lds_refit += sizeof(double) * ((size_t)d.sh_M * d.sh_M + (size_t)D * d.sh_M + 2 * (size_t)D + 2);
        }
        d.sh_mono = s->d_sh_mono; d.sh_start = s->d_sh_start;
    }
    {   // free rejections by boxes: one-ellipsoid bounds whose live points are staged in LDS
        // (off unless asked for: on BASELINE config 5 the boxes save a quarter of the evaluations of the two-component runs
        // and cost more than that in longer rounds -- DESIGN section 10; a bright pixel alone needs a third of the walks' evaluations)
        // (by default: NS_FRAMES frames where the bound is sheared -- there the pair halves the evaluations of config 5's
        // two-component runs in less time than the walks take -- and none elsewhere)
        int nf = s->set_frames != -2 ? s->set_frames : g_eng.sampler_frames != -2 ? g_eng.sampler_frames : (d.shear ? NS_FRAMES : -1);
        d.boxes = (!d.multi && d.stage_live && nf >= 0) ? 1 : 0;
        d.n_frames = d.boxes ? nf : 0;
        d.margin_c = s->set_margin > 0.0 ? s->set_margin : g_eng.sampler_margin_pct > 0 ? 0.01 * g_eng.sampler_margin_pct : NS_MARGIN_C;
        if (d.boxes && d.n_frames > 0) {
            std::vector<double> Q;
            ns_make_frames(D, d.n_frames, Q);
            HIP_TRY(hipMemcpyAsync(s->d_frames, Q.data(), sizeof(double) * Q.size(), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));                   // (Q goes out of scope)
        }
        d.frames = s->d_frames;
        s->raw_sum = s->val_sum = 0;
        const double pe = s->set_pairs >= 0.0 ? s->set_pairs : g_eng.sampler_pairs_pct >= 0 ? 0.01 * g_eng.sampler_pairs_pct : NS_PAIRS_ENLARGE;
        d.pairs = (d.shear && d.boxes && pe >= 1.0) ? 1 : 0;
        d.pairs_enlarge = pe;
        if (d.pairs && (size_t)(D * (D - 1) / 2) * 4 > (size_t)d.sh_M * d.sh_M) d.pairs = 0;      // (they are fitted in the shear's scratch)
    }
    if (s->lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)ns_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds));
    if (s->lds_refit > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)ns_refit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_refit));
    hipLaunchKernelGGL(ns_refit_kernel, dim3((unsigned)P), dim3(d.multi ? 64 : NS_REFIT_THREADS), s->lds_refit, st, d, P, 1);   // first ellipsoids
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    s->rounds = 0;
    s->n_act = maxiter > 0 ? P : 0;
    s->h_act.resize((size_t)P);
    for (int p = 0; p < P; ++p) s->h_act[p] = p;
    if (s->n_act) HIP_TRY(hipMemcpy(s->d_actlist, s->h_act.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    s->ran = true;
    return NFA_OK;
}

int nfa_sampler_advance(nfa_sampler *s, int64_t max_chunks, int64_t *n_active_out) {
    if (!s || !s->ran) return fail(NFA_ERR_STATE, "nfa_sampler_begin has not been called");
    nfa_runner *r = s->r;
    RUNNER_LOCK(r);
    { int rcf = flush_pending(r); if (rcf) return rcf; }
    NsDev &d = s->d;
    const int P = d.P, K = d.K, D = d.D;
    // The active pixels are split in groups that run on different stream lanes: proposing and
    // updating one group (latency-bound, few waves) overlaps the likelihood batch of another.  A pixel's random stream and decisions do not depend on its company, so the
    // split changes nothing in the results.
    const int n_half = std::max(1, std::min(std::min(r->n_lanes, NS_PARTS), g_eng.sampler_parts));
    if (!s->h_pub) {
        HIP_TRY(hipHostMalloc((void **)&s->h_pub, sizeof(unsigned long long) * 2 * NS_PARTS, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer((void **)&s->d_pub, s->h_pub, 0));
        memset(s->h_pub, 0, sizeof(unsigned long long) * 2 * NS_PARTS);
        // (a blocking copy: a memset on the null stream is not ordered before work on the runner's non-blocking lanes)
        const int zeros[NS_PARTS] = {};
        HIP_TRY(hipMemcpy(d.count, zeros, sizeof(int) * NS_PARTS, hipMemcpyHostToDevice));   // the publishing launches leave them at zero
    }
    for (int64_t chunk = 0; s->n_act > 0 && (max_chunks <= 0 || chunk < max_chunks); ++chunk) {
        // candidates per pixel: the round's batch stays near b_target however few pixels are left
        const int n_act = s->n_act;
        // with boxes most proposals are vetoed for free: so many more are drawn that a round still evaluates ~b_target
        long ratio = 1;
        const long ratio_max = s->ratio_max;
        if (d.boxes && s->raw_sum > 0) ratio = std::min<long>(ratio_max, std::max<long>(1, (s->raw_sum + s->val_sum / 2) / std::max<long>(s->val_sum, 1)));
        s->raw_sum = s->val_sum = 0;
        const long kmax = s->kmax;
        const int Kr = (int)std::min<long>(kmax, std::max<long>(K, (s->b_target * ratio) / n_act));    // (n_act * Kr rows <= NS_RATIO_MAX * b_target: what nfa_sampler_create allocated)
        int n_pix_h[NS_PARTS];
        NsDev dh[NS_PARTS];
        {   // every part works on its own slices of the proposal / compact-row buffers
            long first = 0;
            for (int h = 0; h < NS_PARTS; ++h) {
                n_pix_h[h] = h < n_half ? (n_act * (h + 1)) / n_half - (n_act * h) / n_half : 0;
                dh[h] = d;
                const long off = first * Kr;
                dh[h].candU += off * D; dh[h].candT += off * d.DT; dh[h].candL += off;
                dh[h].candpix += off; dh[h].valid += off; dh[h].slot += off;
                dh[h].count += h; dh[h].actlist += first;
                dh[h].host_rows = (int *)(s->d_pub + 2 * h); dh[h].host_seq = s->d_pub + 2 * h + 1;
                first += n_pix_h[h];
            }
        }
        for (int c = 0; c < s->check_every; ++c) {
            for (int h = 0; h < NS_PARTS; ++h) {
                if (n_pix_h[h] == 0) continue;
                hipStream_t st = r->lanes[h];
                const long B = (long)n_pix_h[h] * Kr;
                (void)B;
                const dim3 pg((unsigned)((Kr + 127) / 128), (unsigned)std::min(n_pix_h[h], 65535), (unsigned)((n_pix_h[h] + 65534) / 65535));
                switch (D) {                                        // compile-time dimensions where they are common
                case 5: hipLaunchKernelGGL(ns_propose_kernel<5>, pg, dim3(NS_PROPOSE_THREADS), 0, st, dh[h], n_pix_h[h], Kr); break;
                case 10: hipLaunchKernelGGL(ns_propose_kernel_v128<10>, pg, dim3(NS_PROPOSE_THREADS), 0, st, dh[h], n_pix_h[h], Kr); break;
                case 15: hipLaunchKernelGGL(ns_propose_kernel<15>, pg, dim3(NS_PROPOSE_THREADS), 0, st, dh[h], n_pix_h[h], Kr); break;
                default: hipLaunchKernelGGL(ns_propose_kernel<0>, pg, dim3(NS_PROPOSE_THREADS), 0, st, dh[h], n_pix_h[h], Kr); break;
                }
                hipLaunchKernelGGL(ns_publish_kernel, dim3(1), dim3(1), 0, st, dh[h], s->seq + 1);
                HIP_TRY(hipGetLastError());
            }
            s->seq += 1;
            s->raw_sum += (long)n_act * Kr;
            for (int h = 0; h < NS_PARTS; ++h) {
                if (n_pix_h[h] == 0) continue;
                hipStream_t st = r->lanes[h];
                // proposals inside the prior: the only ones worth a likelihood.  Their number is the launch's last store
                // into the mapped buffer, behind it the round's sequence number
                volatile unsigned long long *pub = s->h_pub + 2 * h;
                const auto t_start = std::chrono::steady_clock::now();
                for (uint64_t spins = 0; __atomic_load_n(pub + 1, __ATOMIC_ACQUIRE) != s->seq; ++spins) {
                    if ((spins & 0x3fff) == 0x3fff && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) {
                        // nothing came back: a fault surfaces here, a very slow launch finishes
                        HIP_TRY(hipStreamSynchronize(st));
                    }
                }
                const int n_rows = (int)(unsigned)(pub[0] & 0xffffffffull);
                s->val_sum += n_rows;
                dh[h].part = nullptr;
                if (n_rows > 0) {
                    r->part_only = true;                                   // (the update wave sums the parts of a row)
                    int rc = run_batch(r, dh[h].candpix, dh[h].candT, dh[h].candL, nullptr, n_rows, true, h, nullptr);
                    r->part_only = false;
                    if (rc) return rc;
                    dh[h].part = r->d_part[h];                             // (after the batch: its buffers may have grown)
                    dh[h].noise = r->ss->dev.noise; dh[h].nspec = r->ss->dev.n_spec;
                }
                hipLaunchKernelGGL(ns_update_kernel, dim3((unsigned)n_pix_h[h]), dim3(NS_UPD_THREADS), s->lds, st, dh[h], n_pix_h[h], Kr, s->rounds);
                // the refit wave of the pixels the update marked, in the rounds where a pixel can be due: rejection-mode
                // pixels every refit_every-th round, walking ones at a cycle's end -- and in a cycle's first round, where a
                // pixel that has just turned to walks brings along what it collected before
                if ((s->rounds + 1) % d.refit_every == 0 || (d.method != 0 && ((s->rounds + 1) % d.n_steps == 0 || s->rounds % d.n_steps == 0)))
                    hipLaunchKernelGGL(ns_refit_kernel, dim3((unsigned)n_pix_h[h]), dim3(dh[h].multi ? 64 : NS_REFIT_THREADS), s->lds_refit, st, dh[h], n_pix_h[h], 0);
                HIP_TRY(hipGetLastError());
            }
            s->rounds += 1;
        }
        for (int h = 0; h < n_half; ++h) HIP_TRY(hipStreamSynchronize(r->lanes[h]));
        HIP_TRY(hipMemcpy(s->h_active.data(), d.active, sizeof(int) * P, hipMemcpyDeviceToHost));
        s->n_act = 0;
        for (int p = 0; p < P; ++p) if (s->h_active[p]) s->h_act[s->n_act++] = p;
        if (s->n_act) HIP_TRY(hipMemcpy(s->d_actlist, s->h_act.data(), sizeof(int) * s->n_act, hipMemcpyHostToDevice));
    }
    for (int h = 0; h < n_half; ++h) HIP_TRY(hipStreamSynchronize(r->lanes[h]));
    if (n_active_out) *n_active_out = s->n_act;
    return NFA_OK;
}

int nfa_sampler_run(nfa_sampler *s, double tol, double efr, int64_t seed, int64_t maxiter, int upd,
                    double log_zero, int check_every) {
    int rc = nfa_sampler_begin(s, tol, efr, seed, maxiter, upd, log_zero, check_every, 1.5, 1, 10 * s->d.D);
    if (rc) return rc;
    return nfa_sampler_advance(s, 0, nullptr);
}

// n_iter[P], n_evals[P], rounds (scalar)
int nfa_sampler_counts(nfa_sampler *s, int64_t *n_iter, int64_t *n_evals, int64_t *rounds) {
    if (!s || !s->ran || !n_iter || !n_evals) return fail(NFA_ERR_ARG, "sampler has not run");
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    HIP_TRY(hipMemcpy(n_iter, s->d.n_iter, sizeof(long) * s->d.P, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(n_evals, s->d.n_evals, sizeof(long) * s->d.P, hipMemcpyDeviceToHost));
    if (rounds) *rounds = s->rounds;
    return NFA_OK;
}

// dead points of pixel p: theta[n][D], lnL[n], lnw[n] with n = min(n_iter[p], cap)
int nfa_sampler_dead(nfa_sampler *s, int64_t p, int64_t n, double *theta, double *lnL, double *lnw) {
    if (!s || !s->ran || p < 0 || p >= s->d.P || n < 0 || n > s->d.cap) return fail(NFA_ERR_ARG, "bad argument");
    if (n == 0) return NFA_OK;
    const NsDev &d = s->d;
    HIP_TRY(hipMemcpy(theta, d.deadT + (size_t)p * d.cap * d.DT, sizeof(double) * n * d.DT, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lnL, d.deadL + (size_t)p * d.cap, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lnw, d.deadlnw + (size_t)p * d.cap, sizeof(double) * n, hipMemcpyDeviceToHost));
    return NFA_OK;
}

// dead points of every pixel at once: offsets[P + 1] (host; offsets[0] = 0, offsets[p + 1] - offsets[p] <= min(n_iter[p],
// cap) rows of pixel p), theta[offsets[P]][DT], lnL[offsets[P]], lnw[offsets[P]]
int nfa_sampler_dead_packed(nfa_sampler *s, const int64_t *offsets, double *theta, double *lnL, double *lnw) {
    if (!s || !s->ran || !offsets || !theta || !lnL || !lnw) return fail(NFA_ERR_ARG, "bad argument");
    const NsDev &d = s->d;
    const int P = d.P;
    if (offsets[0] != 0) return fail(NFA_ERR_ARG, "offsets must start at 0");
    for (int p = 0; p < P; ++p)
        if (offsets[p + 1] < offsets[p] || offsets[p + 1] - offsets[p] > d.cap) return fail(NFA_ERR_ARG, "bad offsets");
    const int64_t total = offsets[P];
    if (total == 0) return NFA_OK;
    long *d_off = nullptr;
    double *d_T = nullptr, *d_L = nullptr, *d_W = nullptr;
    auto release = [&]() { (void)hipFree(d_off); (void)hipFree(d_T); (void)hipFree(d_L); (void)hipFree(d_W); };
    if (hipMalloc((void **)&d_off, sizeof(long) * (P + 1)) != hipSuccess || hipMalloc((void **)&d_T, sizeof(double) * total * d.DT) != hipSuccess
        || hipMalloc((void **)&d_L, sizeof(double) * total) != hipSuccess || hipMalloc((void **)&d_W, sizeof(double) * total) != hipSuccess) {
        release();
        return fail(NFA_ERR_DEVICE, "out of device memory for the packed dead points");
    }
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    hipStream_t st = s->r->lanes[0];
    bool ok = hipMemcpyAsync(d_off, offsets, sizeof(long) * (P + 1), hipMemcpyHostToDevice, st) == hipSuccess;
    hipLaunchKernelGGL(ns_pack_dead_kernel, dim3(16, (unsigned)std::min(P, 32768)), dim3(256), 0, st, d, (const long *)d_off, d_T, d_L, d_W);
    ok = ok && hipGetLastError() == hipSuccess;
    ok = ok && hipMemcpyAsync(theta, d_T, sizeof(double) * total * d.DT, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(lnL, d_L, sizeof(double) * total, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(lnw, d_W, sizeof(double) * total, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    release();
    return ok ? NFA_OK : fail(NFA_ERR_DEVICE, "copying the dead points failed");
}

// The posterior tables of every pixel at once (ns_pack_post_kernel): offsets[P + 1] (host; rows of pixel p =
// min(n_iter[p], cap) dead points + its live points), live_off[P] (host), out[offsets[P]][DT + 2] (host).
int nfa_sampler_posterior_packed(nfa_sampler *s, const int64_t *offsets, const double *live_off, double *out, double *stats) {
    if (!s || !s->ran || !offsets || !live_off || !out) return fail(NFA_ERR_ARG, "bad argument");
    const NsDev &d = s->d;
    const int P = d.P;
    if (offsets[0] != 0) return fail(NFA_ERR_ARG, "offsets must start at 0");
    for (int p = 0; p < P; ++p) {
        const int64_t nl = s->h_nlive.empty() ? d.N : s->h_nlive[(size_t)p];
        if (offsets[p + 1] - offsets[p] < nl || offsets[p + 1] - offsets[p] - nl > d.cap) return fail(NFA_ERR_ARG, "bad offsets");
    }
    const int64_t total = offsets[P];
    long *d_off = nullptr;
    double *d_lo = nullptr, *d_out = nullptr, *d_st = nullptr;
    const size_t n_st = (size_t)P * (6 + 4 * d.DT);
    auto release = [&]() { (void)hipFree(d_off); (void)hipFree(d_lo); (void)hipFree(d_out); (void)hipFree(d_st); };
    if (hipMalloc((void **)&d_off, sizeof(long) * (P + 1)) != hipSuccess || hipMalloc((void **)&d_lo, sizeof(double) * P) != hipSuccess
        || hipMalloc((void **)&d_out, sizeof(double) * total * (d.DT + 2)) != hipSuccess
        || (stats && hipMalloc((void **)&d_st, sizeof(double) * n_st) != hipSuccess)) {
        release();
        return fail(NFA_ERR_DEVICE, "out of device memory for the packed posterior tables");
    }
    static_assert(sizeof(long) == sizeof(int64_t), "LP64");
    hipStream_t st = s->r->lanes[0];
    bool ok = hipMemcpyAsync(d_off, offsets, sizeof(long) * (P + 1), hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(d_lo, live_off, sizeof(double) * P, hipMemcpyHostToDevice, st) == hipSuccess;
    hipLaunchKernelGGL(ns_pack_post_kernel, dim3(16, (unsigned)std::min(P, 32768)), dim3(256), 0, st, d, (const long *)d_off, (const double *)d_lo, d_out);
    ok = ok && hipGetLastError() == hipSuccess;
    if (stats) {            // weights, evidence, information and moments formed where the tables are
        hipLaunchKernelGGL(ns_finish_kernel, dim3((unsigned)P), dim3(NS_FIN_THREADS), 0, st, d, (const long *)d_off, d_out, d_st);
        ok = ok && hipGetLastError() == hipSuccess;
        ok = ok && hipMemcpyAsync(stats, d_st, sizeof(double) * n_st, hipMemcpyDeviceToHost, st) == hipSuccess;
    }
    ok = ok && hipMemcpyAsync(out, d_out, sizeof(double) * total * (d.DT + 2), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    release();
    return ok ? NFA_OK : fail(NFA_ERR_DEVICE, "copying the posterior tables failed");
}

// final live points: theta[P][N][D], lnL[P][N]
int nfa_sampler_live(nfa_sampler *s, double *theta, double *lnL) {
    if (!s || !s->ran || !theta || !lnL) return fail(NFA_ERR_ARG, "sampler has not run");
    const NsDev &d = s->d;
    HIP_TRY(hipMemcpy(theta, d.Tlive, sizeof(double) * (size_t)d.P * d.N * d.DT, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lnL, d.Llive, sizeof(double) * (size_t)d.P * d.N, hipMemcpyDeviceToHost));
    return NFA_OK;
}

}  // extern "C"
