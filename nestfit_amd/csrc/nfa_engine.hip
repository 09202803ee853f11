// nfa_engine.hip -- host side (C ABI of include/nestfit_amd.h) of the MI355X
// NH3 log-likelihood engine; the kernels live in nfa_device.h.
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared (see
// nestfit_amd/build.py).  One process per GPU; every runner owns a HIP stream.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/nestfit_amd.h"

#define NFA_DATA_QUAL static const
#include "nh3_data.h"
#include "n2hp_data.h"
#include "nfa_device.h"

// ---------------------------------------------------------------------------
//  error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess)                                                      \
            return fail(NFA_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------
//  host side
// ---------------------------------------------------------------------------
struct Engine {
    bool   init = false;
    int    device = 0;
    int    n_cu = 256;
    int    exp_mode = 2;             // "fast": see include/nestfit_amd.h, nfa_set_exp_mode
    int    wpb = 1;                  // waves per workgroup of the likelihood kernel (fast / poly mode): one -- a workgroup
                                     // retires, and its slot is refilled, wave by wave (4: -1.5 %, 8: -9 %, profiles/r02/sweep_lanes.txt)
    int    wpb_table = 0;            // the same in table mode; 0 = chosen per spectra set (launch_lnl_t)
    int    lnl_split = 0;            // waves per (item, spectrum) unit of the likelihood kernel: 1, 2, 4, or
                                     // 0 = by launch size (resolve_split).  The chi^2 of a unit is a sum of LNL_PARTS
                                     // row blocks in a fixed order whatever the split, so every evaluation is bitwise
                                     // independent of it and of the batch it travels in (the sampler twin relies on that)
    int    lnl_queue = 1;            // table mode: launches of two and more units per wave slot run as resident workgroups
                                     // whose waves draw the units from a queue (lnl_kernel_queue); 0 = one unit per wave always
    unsigned long long *d_trace = nullptr;   // test library: the queue kernel's per-wave records (nfa_test_queue_trace)
    int    lnl_queue_wg = 0;         // A/B: workgroups per CU of a queue launch (0 = 2)
    int    lnl_cap = 0;              // fast / poly mode: workgroups of the likelihood kernel resident per CU at most
                                     // (LDS padding; 0 = no cap).  A/B knob: leaving one slot per CU to the set-up
                                     // kernels of the next batch paid off (+7 %) until those kernels got a raised wave
                                     // priority of their own; with it the cap only costs occupancy (-6 %).
    int    ablate = 0;
    int    streams = 0;              // stream lanes of new runners; 0 = six, of which a batch uses four or six (run_batch)
    int    sampler_parts = 3;        // groups of pixels the device sampler pipelines over the lanes
    int    sampler_refit_every = 4;  // rejection-mode pixels refit their bound in every n-th round (A/B knob)
    int    sampler_walkers = 0;      // walkers per pixel of a walk cycle (A/B knob: 64, 128, 192, 256); 0 = by the live points
    int    sampler_ellipsoids = 0;   // 1: one bounding ellipsoid per pixel whatever the dimension (A/B knob; 0: several where it pays)
    int    sampler_frames = -2;      // rotated box frames of a one-ellipsoid bound: -2 = by the sampled dimensions, -1 = no boxes, 0..64
    int    sampler_margin_pct = 0;   // the boxes' margin factor c in hundredths (0 = NS_MARGIN_C)
    int    profile_skip = 0;         // nfa_runner_get_profile leaves the first calls out (warm-up launches behind an idle gap)
    int    sampler_ktarget = -1;     // replacements per pixel and rejection round its share of proposals aims at (-1 = NS_K_TARGET, 0 = everybody the round's Kr)
    int    sampler_pairs_pct = -1;   // the pair ellipses' safety factor in hundredths (-1 = NS_PAIRS_ENLARGE where the bound is sheared and boxed, 0 = off)
    int    sampler_ratio_max = 0;    // proposals drawn per round: at most this multiple of the evaluations aimed for (0 = NS_RATIO_MAX)
    int    sampler_kmax = 0;         // most proposals one pixel gets in a round (0 = NS_KMAX)
    int    sampler_shear_pct = -1;   // the shear's safety factor in hundredths (-1 = the default, NS_SHEAR_ENLARGE where the shape allows; 0 = no shear)
    int    sampler_walk_factor = 0;  // a pixel turns to walks when rejection accepts fewer than 1 in factor * n_steps; 0 = by the
                                     // number of sampled dimensions (nfa_sampler_begin)
    int    graph = -1;               // single-point graph replay: -1 = decide at first use, 0 off, 1 on
    int    coalesce = NFA_GROUP_MAX; // device-pointer batches of one shape enqueued back to back travel as one launch, up to
                                     // this many (1 = every batch its own launches)
    int    prior_stage = 1;          // prior tables staged in LDS by the set-up kernel (priors created afterwards)
    int    setup_ti = 0, setup_threads = 0;   // set-up kernel: items and threads per workgroup (0 = 64 / 256)
    int    setup_sub = 0;                     // table mode: 1 = one group of items per set-up workgroup (0 = two where the batch allows)
    int    point = 1;                // single points: 1 = the one-launch point kernel, 0 = the batch kernels (graph replay)
    bool   have_t0 = false;
    double *d_tabs = nullptr;                      // SM_END_TABLE doubles
    double t0_xmin = 0, t0_xmax = 0, t0_inv_dx = 0;
};
static Engine g_eng;
struct nfa_runner;
static int flush_pending(nfa_runner *r);
static int flush_all_runners();
static std::mutex g_runners_m;
static std::vector<nfa_runner *> g_runners;         // live runners: nfa_device_synchronize / nfa_set_exp_mode flush them all

// The HIP runtime multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4);
// streams that share a queue run in order with each other.  A runner's stream lanes only overlap when
// every lane has a queue of its own, so ask for 8 -- before the runtime reads the variable, i.e. when this
// library is loaded, and only if the user has not set it.
__attribute__((constructor)) static void nfa_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static int engine_init_once();
// HIP's current device is per host thread (default 0): every public entry point that allocates or
// launches binds the calling thread to the engine's device first (broker threads, sampler threads
// of a host application, MultiNest's own thread).
static int engine_init() {
    if (!g_eng.init) { int rc = engine_init_once(); if (rc) return rc; }
    static thread_local int bound = -1;
    if (bound != g_eng.device) {
        HIP_TRY(hipSetDevice(g_eng.device));
        bound = g_eng.device;
    }
    return NFA_OK;
}
static int engine_init_once() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_eng.init) return NFA_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(NFA_ERR_DEVICE, "no HIP device available (the engine has no CPU fallback)");
    HIP_TRY(hipSetDevice(g_eng.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_eng.device));
    g_eng.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // transition tables of all models in one index space (nfa_device.h)
        static int h_nhf[NFA_T_ALL];
        static double h_nu[NFA_T_ALL], h_voff[NFA_T_ALL][NFA_MAX_HF_N], h_tauw[NFA_T_ALL][NFA_MAX_HF_N];
        memset(h_voff, 0, sizeof(h_voff)); memset(h_tauw, 0, sizeof(h_tauw));
        for (int t = 0; t < NFA_N_LEVELS; ++t) {
            h_nhf[t] = nfa_nhf[t]; h_nu[t] = nfa_nu[t];
            memcpy(h_voff[t], nfa_voff[t], sizeof(h_voff[t])); memcpy(h_tauw[t], nfa_tau_wts[t], sizeof(h_tauw[t]));
        }
        for (int t = 0; t < NFA_N2HP_LEVELS; ++t) {
            const int g = NFA_T_N2HP + t;
            h_nhf[g] = nfa_n2hp_nhf[t]; h_nu[g] = nfa_n2hp_nu[t];
            memcpy(h_voff[g], nfa_n2hp_voff[t], sizeof(h_voff[g])); memcpy(h_tauw[g], nfa_n2hp_tau_wts[t], sizeof(h_tauw[g]));
        }
        h_nhf[NFA_T_GAUSS] = 1; h_nu[NFA_T_GAUSS] = 0.0; h_tauw[NFA_T_GAUSS][0] = 1.0;
        static double h_hfreq[NFA_T_ALL][NFA_MAX_HF_N];
        for (int t = 0; t < NFA_T_ALL; ++t)
            for (int i = 0; i < NFA_MAX_HF_N; ++i) {
                volatile double q = h_voff[t][i] / NFA_CKMS;      // hyperfine.pyx:71, one rounding per operation
                volatile double f = 1.0 - q;
                h_hfreq[t][i] = f * h_nu[t];
            }
        // rank of every line when a transition's lines are ordered by velocity offset (stable; nfa_device.h: c_rank)
        static unsigned char h_rank[NFA_T_ALL][NFA_MAX_HF_N];
        for (int t = 0; t < NFA_T_ALL; ++t) {
            std::vector<int> order(h_nhf[t]);
            for (int i = 0; i < h_nhf[t]; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h_voff[t][a] > h_voff[t][b]; });
            for (int i = 0; i < NFA_MAX_HF_N; ++i) h_rank[t][i] = (unsigned char)i;
            for (int k = 0; k < h_nhf[t]; ++k) h_rank[t][order[k]] = (unsigned char)k;
        }
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_rank), h_rank, sizeof(h_rank)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_hfreq), h_hfreq, sizeof(h_hfreq)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_nhf), h_nhf, sizeof(h_nhf)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_nu), h_nu, sizeof(h_nu)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_ea), nfa_ea, sizeof(nfa_ea)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_voff), h_voff, sizeof(h_voff)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_tauw), h_tauw, sizeof(h_tauw)));
    }
    std::vector<double> tabs(SM_END_TABLE, 0.0);
    // 2^(i/256)
    for (int i = 0; i < NFA_EXP2_N; ++i) tabs[SM_EXP2 + i] = (double)exp2l((long double)i / (long double)NFA_EXP2_N);
    // FastExp product tables: host libm exp() of exactly representable
    // arguments, as the reference fills them (fastexp.c:203-226)
    for (int l = 0; l < 10; ++l) {
        for (int j = 0; j < 128; ++j) tabs[SM_FEA + l * 128 + j] = exp(-ldexp((double)(128 + j), l - 12));
        for (int j = 0; j < 256; ++j) {
            tabs[SM_FEB + l * 256 + j] = exp(-ldexp((double)j, l - 20));
            tabs[SM_FEC + l * 256 + j] = exp(-ldexp((double)j, l - 28));
        }
    }
    HIP_TRY(hipMalloc(&g_eng.d_tabs, sizeof(double) * SM_END_TABLE));
    HIP_TRY(hipMemcpy(g_eng.d_tabs, tabs.data(), sizeof(double) * SM_END_TABLE, hipMemcpyHostToDevice));
    g_eng.init = true;
    return NFA_OK;
}

struct nfa_specset {
    SpecDev dev{};
    int64_t n_pix = 0;
    int     nhf_max = 0;
    double *d_xarr = nullptr, *d_t0 = nullptr, *d_tbg = nullptr, *d_data = nullptr, *d_noise = nullptr;
    double *d_t0tbg = nullptr, *d_rowsq = nullptr, *d_totsq = nullptr;
};

struct nfa_priors {
    PriorProg prog{};
    PriorProg *d_prog = nullptr;       // device copy handed to the kernels
    std::vector<double *> d_arrays;
};

#define NFA_MAX_LANES 8
struct nfa_runner {
    nfa_specset *ss = nullptr;
    nfa_priors  *pr = nullptr;
    int ncomp = 1, cold = 0, lte = 0, ndim = 6;
    // numerical mode: -1 = the process default at call time (nfa_set_exp_mode), 0..2 = pinned to
    // this runner (nfa_runner_set_exp_mode): runners of different modes may then work side by side
    int exp_mode = -1;
    int wpb = 1, wpb_table = 0, lnl_cap = 0, lnl_split = 0;   // launch geometry, taken from the process options at creation
    // Stream lanes: consecutive batches go to different HIP streams (round robin), so the
    // tail of one batch (few workgroups left, SIMDs draining) overlaps the start of the
    // next; inside a lane the set-up kernel and the likelihood kernel run in order and own
    // the lane's derived-parameter records.
    int         n_lanes = 1;
    bool        lanes_auto = false;      // four lanes, six once batches of about one wave per slot have come by (run_batch)
    hipStream_t lanes[NFA_MAX_LANES] = {};
    double     *d_D[NFA_MAX_LANES] = {};
    double     *d_part[NFA_MAX_LANES] = {};  // per (item, spectrum) log-likelihood terms
    unsigned   *d_queue[NFA_MAX_LANES] = {}; // unit counters of the lane's table-mode launches (lnl_kernel_queue), zero between launches
    int64_t     cap_D[NFA_MAX_LANES] = {};
    hipStream_t stream = nullptr;            // lane 0: also the stream of the host-pointer entry points
    uint64_t    n_calls = 0;
    unsigned    lane_busy = 0;               // lanes with work enqueued and not yet synchronised
    double *d_U = nullptr, *d_lnL = nullptr, *d_spec = nullptr;
    int    *d_pix = nullptr;
    int64_t cap_B = 0, cap_spec = 0;
    // single-point calls (MultiNest's LogLike): the whole H2D -> 5 kernels -> D2H sequence as one
    // captured graph, replayed per call (built on the third single-point call in a given mode)
    hipGraphExec_t g1 = nullptr;
    int     g1_mode = -1;
    double *h_pin = nullptr;         // pinned staging: ndim + 1 doubles
    uint64_t n_single = 0;
    bool    part_only = false;       // a batch leaves the per-spectrum chi^2 parts; whoever set this sums them (the device sampler's update)
    double *h_point = nullptr;       // mapped host buffer of the point kernel: theta[ndim], lnL, sequence number
    double *d_point = nullptr;       // the same buffer as the device sees it
    unsigned *d_point_done = nullptr; // workgroups of a point launch that have finished
    uint64_t pt_seq = 0;
    // optional per-kernel timing (HIP events on the runner's stream)
    bool profiling = false;
    std::vector<hipEvent_t> ev;      // per call: start and stop of the set-up kernel's dispatch, start and stop of lnl_kernel's
    size_t ev_used = 0;
    hipEvent_t *ev_cur = nullptr;        // profiling: [set-up start, stop, likelihood start, stop] of the call under way
    BatchGroup  cur_group = {};          // the batches of the launches being enqueued (run_group)
    BatchGroup  pending = {};            // device-pointer batches accepted but not yet launched (coalescing)
    bool        pending_prior = true;    // ... loglike batches (unit cube, prior transform) or predict batches (physical parameters)
    // One in-flight call per runner is the contract (include/nestfit_amd.h); the process-wide calls
    // (nfa_device_synchronize, nfa_set_exp_mode) walk every live runner from whatever thread makes them, so the state a
    // launch touches -- pending, cur_group, n_calls, lane_busy, the lanes themselves -- is guarded: every public entry
    // point of a runner and the global flush take this lock (recursive: entry points call each other).
    std::recursive_mutex mu;
};
#define RUNNER_LOCK(r) std::lock_guard<std::recursive_mutex> runner_lock_((r)->mu)

extern "C" {

const char *nfa_last_error(void) { return g_err.c_str(); }
int nfa_version(void) { return 100; }

int nfa_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(NFA_ERR_DEVICE, hipGetErrorString(e)); }
    *count = n;
    return NFA_OK;
}

int nfa_set_device(int device) {
    if (g_eng.init && device != g_eng.device)
        return fail(NFA_ERR_STATE, "nfa_set_device must be called before any other engine call");
    g_eng.device = device;
    return engine_init();
}

int nfa_device_synchronize(void) {
    int rc = flush_all_runners(); if (rc) return rc;          // batches held for coalescing are launched first
    HIP_TRY(hipDeviceSynchronize());
    return NFA_OK;
}

int nfa_device_name(char *buf, int buflen) {
    int rc = engine_init(); if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_eng.device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return NFA_OK;
}

int nfa_device_uuid(char *buf, int buflen) {
    if (!buf || buflen < 33) return fail(NFA_ERR_ARG, "buffer too small for a UUID (33 bytes)");
    int rc = engine_init(); if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_eng.device));
    for (int k = 0; k < 16; ++k) snprintf(buf + 2 * k, 3, "%02x", (unsigned)(unsigned char)prop.uuid.bytes[k]);
    return NFA_OK;
}

int nfa_set_exp_mode(int mode) {
    if (mode != 0 && mode != 2) return fail(NFA_ERR_ARG, "exp mode must be 0 (table) or 2 (fast)");
    { int rc = flush_all_runners(); if (rc) return rc; }       // what is held was enqueued under the old mode
    g_eng.exp_mode = mode;
    return NFA_OK;
}
int nfa_get_exp_mode(void) { return g_eng.exp_mode; }

int nfa_set_option(const char *key, int value) {
    if (key && !strcmp(key, "lnl_cap") && value >= 0 && value <= 8) { g_eng.lnl_cap = value; return NFA_OK; }
    if (key && !strcmp(key, "lnl_queue_wg") && value >= 0 && value <= 2) { g_eng.lnl_queue_wg = value; return NFA_OK; }
    if (key && !strcmp(key, "lnl_queue") && (value == 0 || value == 1)) { g_eng.lnl_queue = value; return NFA_OK; }
    if (key && !strcmp(key, "lnl_split") && (value == 0 || value == 1 || value == 2 || value == 4)) { g_eng.lnl_split = value; return NFA_OK; }
    if (key && !strcmp(key, "graph") && (value == 0 || value == 1)) { g_eng.graph = value; return NFA_OK; }
    if (key && !strcmp(key, "point") && (value == 0 || value == 1)) { g_eng.point = value; return NFA_OK; }
    if (key && !strcmp(key, "coalesce") && value >= 1 && value <= NFA_GROUP_MAX) { g_eng.coalesce = value; return NFA_OK; }
    if (key && !strcmp(key, "prior_stage") && (value == 0 || value == 1)) { g_eng.prior_stage = value; return NFA_OK; }
    if (key && !strcmp(key, "setup_ti") && (value == 0 || value == 8 || value == 16 || value == 32 || value == 64)) { g_eng.setup_ti = value; return NFA_OK; }
    if (key && !strcmp(key, "setup_sub") && (value == 0 || value == 1)) { g_eng.setup_sub = value; return NFA_OK; }
    if (key && !strcmp(key, "setup_threads") && (value == 0 || value == 256 || value == 320 || value == 384 || value == 448 || value == 512)) { g_eng.setup_threads = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_parts") && value >= 1 && value <= 4) { g_eng.sampler_parts = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_refit_every") && value >= 1 && value <= 16) { g_eng.sampler_refit_every = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_walkers") && value >= 0 && value <= 256 && value % 64 == 0) { g_eng.sampler_walkers = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_ellipsoids") && value >= 0 && value <= 1) { g_eng.sampler_ellipsoids = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_walk_factor") && value >= 0 && value <= 1024) { g_eng.sampler_walk_factor = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_frames") && value >= -2 && value <= 64) { g_eng.sampler_frames = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_margin_pct") && value >= 0 && value <= 1000) { g_eng.sampler_margin_pct = value; return NFA_OK; }
    if (key && !strcmp(key, "profile_skip") && value >= 0 && value <= 100000) { g_eng.profile_skip = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_ktarget") && value >= -1 && value <= 4096) { g_eng.sampler_ktarget = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_pairs_pct") && (value == -1 || value == 0 || (value >= 100 && value <= 100000))) { g_eng.sampler_pairs_pct = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_ratio_max") && value >= 0 && value <= 64) { g_eng.sampler_ratio_max = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_kmax") && (value == 0 || (value >= 64 && value <= 262144))) { g_eng.sampler_kmax = value; return NFA_OK; }
    if (key && !strcmp(key, "sampler_shear_pct") && (value == -1 || value == 0 || (value >= 100 && value <= 100000))) { g_eng.sampler_shear_pct = value; return NFA_OK; }
    if (key && !strcmp(key, "wpb_table") && value >= 0 && value <= 16) { g_eng.wpb_table = value; return NFA_OK; }
    if (key && !strcmp(key, "wpb") && value >= 1 && value <= 16) { g_eng.wpb = value; return NFA_OK; }
#ifdef NFA_ABLATE
    if (key && !strcmp(key, "ablate") && value >= 0 && value <= 127) { g_eng.ablate = value; return NFA_OK; }
#endif
    if (key && !strcmp(key, "streams") && value >= 0 && value <= NFA_MAX_LANES) { g_eng.streams = value; return NFA_OK; }
    return fail(NFA_ERR_ARG, "unknown option, or value out of range");
}

int nfa_set_iemtex_table(const double *t0_x, const double *t0_y, int64_t n) {
    if (n != T0_SIZE || !t0_x || !t0_y) return fail(NFA_ERR_ARG, "iemtex table must have 1000 points");
    int rc = engine_init(); if (rc) return rc;
    HIP_TRY(hipMemcpy(g_eng.d_tabs + SM_T0X, t0_x, sizeof(double) * T0_SIZE, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g_eng.d_tabs + SM_T0Y, t0_y, sizeof(double) * T0_SIZE, hipMemcpyHostToDevice));
    g_eng.t0_xmin = (NFA_H * 23.0e9 / NFA_KB) / 8.0;           // hyperfine.pyx:13-16
    g_eng.t0_xmax = (NFA_H * 28.0e9 / NFA_KB) / 2.7;
    g_eng.t0_inv_dx = 1.0 / (t0_x[1] - t0_x[0]);               // hyperfine.pyx:20
    g_eng.have_t0 = true;
    return NFA_OK;
}

// ---- spectra ---------------------------------------------------------------
int nfa_specset_create(nfa_specset **out, int n_spec, const int64_t *sizes,
                       const int32_t *trans_ids, const double *const *xarr,
                       int64_t n_pix, const double *data, const double *noise) {
    return nfa_specset_create_model(out, NFA_MODEL_AMMONIA, n_spec, sizes, trans_ids, nullptr, xarr, n_pix,
                                    data, noise);
}

// sums of data^2 per row of 64 channels (chi^2 of the rows without a line window) of pixels
// [pix0, pix0 + n)
static int launch_rowsq(nfa_specset *ss, int64_t pix0, int64_t n) {
    const int64_t waves = n * ss->dev.rows_tot;
    hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, 0, ss->dev,
                       (long)pix0, (long)n, ss->d_rowsq);
    hipLaunchKernelGGL(totsq_kernel, dim3((unsigned)((n * ss->dev.n_spec + 255) / 256)), dim3(256), 0, 0, ss->dev,
                       (long)pix0, (long)n, (const double *)ss->d_rowsq, ss->d_totsq);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return NFA_OK;
}

static int specset_fill(nfa_specset *ss, int model, int n_spec, const int64_t *sizes, const int32_t *trans_ids,
                        const double *rest_freqs, const double *const *xarr, int64_t n_pix, const double *data,
                        const double *noise) {
    SpecDev &d = ss->dev;
    d.n_spec = n_spec;
    d.model = model;
    d.npar = model == NFA_MODEL_DIAZENYLIUM ? NFA_N2HP_PARAMS : model == NFA_MODEL_GAUSSIAN ? NFA_GAUSS_PARAMS
                                                                                          : NFA_N_PARAMS;
    int64_t tot = 0, rows = 0;
    for (int s = 0; s < n_spec; ++s) {
        if (sizes[s] < 2 || sizes[s] > (1 << 24)) return fail(NFA_ERR_ARG, "spectrum size out of range");
        int tglob;                                                      // index into the device tables
        if (model == NFA_MODEL_AMMONIA) {
            if (trans_ids[s] < 1 || trans_ids[s] > NFA_N_LEVELS)        // ammonia.pyx:268
                return fail(NFA_ERR_ARG, "trans_id must be in 1..9");
            tglob = trans_ids[s] - 1;
            d.rest[s] = nfa_nu[tglob];
            ss->nhf_max = std::max(ss->nhf_max, nfa_nhf[tglob]);
        } else if (model == NFA_MODEL_DIAZENYLIUM) {
            if (trans_ids[s] < 1 || trans_ids[s] > NFA_N2HP_LEVELS)     // diazenylium.pyx:128
                return fail(NFA_ERR_ARG, "trans_id must be in 1..3");
            tglob = NFA_T_N2HP + trans_ids[s] - 1;
            d.rest[s] = nfa_n2hp_nu[trans_ids[s] - 1];
            ss->nhf_max = std::max(ss->nhf_max, nfa_n2hp_nhf[trans_ids[s] - 1]);
        } else {
            tglob = NFA_T_GAUSS;
            d.rest[s] = rest_freqs ? rest_freqs[s] : 0.0;               // core.pyx:510
            ss->nhf_max = std::max(ss->nhf_max, 1);
        }
        const double nu_chan = xarr[s][1] - xarr[s][0];
        if (!(nu_chan > 0)) return fail(NFA_ERR_ARG, "frequency axis must be ascending");   // core.pyx:503-504
        d.size[s] = (int)sizes[s];
        d.trans[s] = tglob + 1;
        d.off[s] = (int)tot;
        d.row_off[s] = (int)rows;
        d.nu_min[s] = xarr[s][0];
        d.nu_chan[s] = nu_chan;
        {   // the reciprocal nf_line divides with (0: a width that is not an ordinary number, or whose mantissa is all ones)
            uint64_t bits; memcpy(&bits, &nu_chan, sizeof bits);
            const bool ordinary = std::isnormal(nu_chan) && nu_chan > 1e-100 && nu_chan < 1e100;
            d.r_chan[s] = ordinary && (bits & 0xfffffffffffffull) != 0xfffffffffffffull ? 1.0 / nu_chan : 0.0;
        }
        tot += sizes[s];
        rows += (sizes[s] + 63) / 64;
    }
    for (int64_t i = 0; i < n_pix * n_spec; ++i)
        if (!(noise[i] > 0)) return fail(NFA_ERR_ARG, "noise must be > 0");                 // core.pyx:502
    d.chan_tot = tot;
    d.rows_tot = rows;
    ss->n_pix = n_pix;
    std::vector<double> xcat(tot);
    for (int s = 0; s < n_spec; ++s) memcpy(xcat.data() + d.off[s], xarr[s], sizeof(double) * sizes[s]);
    HIP_TRY(hipMalloc(&ss->d_xarr, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_t0, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_tbg, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_t0tbg, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_data, sizeof(double) * tot * n_pix));
    HIP_TRY(hipMalloc(&ss->d_noise, sizeof(double) * n_spec * n_pix));
    HIP_TRY(hipMalloc(&ss->d_rowsq, sizeof(double) * rows * n_pix));
    HIP_TRY(hipMalloc(&ss->d_totsq, sizeof(double) * n_spec * n_pix));
    HIP_TRY(hipMemcpy(ss->d_xarr, xcat.data(), sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ss->d_data, data, sizeof(double) * tot * n_pix, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ss->d_noise, noise, sizeof(double) * n_spec * n_pix, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0,
                       ss->d_xarr, ss->d_t0, ss->d_tbg, ss->d_t0tbg, (long)tot);
    HIP_TRY(hipGetLastError());
    d.xarr = ss->d_xarr; d.t0 = ss->d_t0; d.tbg = ss->d_tbg; d.data = ss->d_data; d.noise = ss->d_noise;
    d.t0tbg = ss->d_t0tbg; d.rowsq = ss->d_rowsq; d.totsq = ss->d_totsq;
    return launch_rowsq(ss, 0, n_pix);
}

int nfa_specset_create_model(nfa_specset **out, int model, int n_spec, const int64_t *sizes,
                             const int32_t *trans_ids, const double *rest_freqs,
                             const double *const *xarr, int64_t n_pix, const double *data,
                             const double *noise) {
    if (!out || !sizes || !xarr || !data || !noise) return fail(NFA_ERR_ARG, "null argument");
    if (model < NFA_MODEL_AMMONIA || model > NFA_MODEL_GAUSSIAN) return fail(NFA_ERR_ARG, "unknown model");
    if (model != NFA_MODEL_GAUSSIAN && !trans_ids) return fail(NFA_ERR_ARG, "null argument");
    if (model == NFA_MODEL_GAUSSIAN && n_spec != 1)                     // gaussian.pyx:57-89
        return fail(NFA_ERR_ARG, "the Gaussian model takes one spectrum");
    if (n_spec < 1 || n_spec > MAXSPEC) return fail(NFA_ERR_ARG, "n_spec must be in 1..16");
    if (n_pix < 1) return fail(NFA_ERR_ARG, "n_pix must be >= 1");
    int rc = engine_init(); if (rc) return rc;
    nfa_specset *ss = new nfa_specset();
    rc = specset_fill(ss, model, n_spec, sizes, trans_ids, rest_freqs, xarr, n_pix, data, noise);
    if (rc) { nfa_specset_destroy(ss); return rc; }          // frees whatever was allocated
    *out = ss;
    return NFA_OK;
}

int nfa_specset_destroy(nfa_specset *ss) {
    if (!ss) return NFA_OK;
    (void)hipFree(ss->d_xarr); (void)hipFree(ss->d_t0); (void)hipFree(ss->d_tbg); (void)hipFree(ss->d_data); (void)hipFree(ss->d_noise);
    (void)hipFree(ss->d_t0tbg); (void)hipFree(ss->d_rowsq); (void)hipFree(ss->d_totsq);
    delete ss;
    return NFA_OK;
}

int nfa_specset_set_data(nfa_specset *ss, int64_t pix, const double *data) {
    if (!ss || !data || pix < 0 || pix >= ss->n_pix) return fail(NFA_ERR_ARG, "bad pixel index");
    int rc = engine_init(); if (rc) return rc;
    HIP_TRY(hipMemcpy(ss->d_data + pix * ss->dev.chan_tot, data, sizeof(double) * ss->dev.chan_tot,
                      hipMemcpyHostToDevice));
    return launch_rowsq(ss, pix, 1);
}

int nfa_specset_null_lnz(const nfa_specset *ss, double *out) {
    if (!ss || !out) return fail(NFA_ERR_ARG, "null argument");
    const int64_t n = ss->n_pix * ss->dev.n_spec;
    double *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, sizeof(double) * n));
    hipLaunchKernelGGL(null_lnz_kernel, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, 0,
                       ss->dev, (long)ss->n_pix, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);                                     // on every path
    if (e != hipSuccess) return fail(NFA_ERR_DEVICE, std::string("nfa_specset_null_lnz: ") + hipGetErrorString(e));
    return NFA_OK;
}

int nfa_specset_tbg(const nfa_specset *ss, double *out) {
    if (!ss || !out) return fail(NFA_ERR_ARG, "null argument");
    HIP_TRY(hipMemcpy(out, ss->d_tbg, sizeof(double) * ss->dev.chan_tot, hipMemcpyDeviceToHost));
    return NFA_OK;
}

int64_t nfa_specset_chan_tot(const nfa_specset *ss) { return ss ? ss->dev.chan_tot : 0; }

// ---- priors ----------------------------------------------------------------
static int priors_fill(nfa_priors *p, const nfa_prior_desc *priors, int n_prior, const nfa_dist_desc *dists,
                       int n_dist, int n_param) {
    PriorProg &g = p->prog;
    g.n_prior = n_prior; g.n_dist = n_dist; g.n_param = n_param; g.max_size = 2;
    for (int k = 0; k < n_prior; ++k) {
        g.pr[k] = priors[k];
        const int dd[3] = {priors[k].dist0, priors[k].dist1, priors[k].dist2};
        for (int q = 0; q < 3; ++q)
            if (dd[q] >= n_dist) return fail(NFA_ERR_ARG, "distribution index out of range");
        if (priors[k].p_ix < 0) return fail(NFA_ERR_ARG, "p_ix must be >= 0");   // core.pyx:186
    }
    auto upload = [&](const double *src, int64_t n, const double **dst) -> int {
        double *dp = nullptr;
        HIP_TRY(hipMalloc(&dp, sizeof(double) * n));
        p->d_arrays.push_back(dp);                 // owned from here on: nfa_priors_destroy frees it
        HIP_TRY(hipMemcpy(dp, src, sizeof(double) * n, hipMemcpyHostToDevice));
        *dst = dp;
        return NFA_OK;
    };
    std::vector<std::array<std::vector<double>, 3>> moments;      // host copies, for the staging image below
    for (int k = 0; k < n_dist; ++k) {
        const nfa_dist_desc &s = dists[k];
        if (s.size < 2 || s.size > 65536) return fail(NFA_ERR_ARG, "distribution size out of range");
        DistDev &d = g.ds[k];
        d.size = (int)s.size; d.du = s.du; d.dx = s.dx; d.xmin = s.xmin; d.xmax = s.xmax;
        const double *src[4] = {s.xax, s.pdf, s.cdf, s.ppf};
        const double **dst[4] = {&d.xax, &d.pdf, &d.cdf, &d.ppf};
        for (int q = 0; q < 4; ++q) { int rc = upload(src[q], s.size, dst[q]); if (rc) return rc; }
        // prefix moments of the trapezoid terms (long double accumulation, one rounding each)
        std::vector<double> m0(s.size), m1(s.size), m2(s.size);
        long double a0 = 0, a1 = 0, a2 = 0;
        m0[0] = m1[0] = m2[0] = 0.0;
        for (int64_t i = 1; i < s.size; ++i) {
            const long double ti = 0.5L * ((long double)s.pdf[i] + (long double)s.pdf[i - 1]);
            const long double ic = (long double)(i - s.size / 2);
            a0 += ti; a1 += ti * ic; a2 += ti * ic * ic;
            m0[i] = (double)a0; m1[i] = (double)a1; m2[i] = (double)a2;
        }
        const double *msrc[3] = {m0.data(), m1.data(), m2.data()};
        const double **mdst[3] = {&d.m0, &d.m1, &d.m2};
        for (int q = 0; q < 3; ++q) { int rc = upload(msrc[q], s.size, mdst[q]); if (rc) return rc; }
        g.max_size = std::max(g.max_size, (int)s.size);
        moments.push_back({std::move(m0), std::move(m1), std::move(m2)});
    }
    // tables the set-up kernel keeps in LDS: ppf of every distribution a prior interpolates, and the
    // abscissa + prefix moments (+ pdf, for more than three components) of a placement prior's distribution
    {
        bool need[MAXDIST][6] = {};
        for (int k = 0; k < n_prior; ++k) {
            const nfa_prior_desc &q = priors[k];
            const bool composite = q.kind == NFA_PRIOR_RESOLVED_CENSEP || q.kind == NFA_PRIOR_RESOLVED_PLACEMENT;
            if (q.kind != NFA_PRIOR_CONSTANT && q.dist0 >= 0) need[q.dist0][ST_PPF] = true;
            if ((q.kind == NFA_PRIOR_SPACED || q.kind == NFA_PRIOR_CENSEP || q.kind == NFA_PRIOR_RESOLVED_CENSEP) && q.dist1 >= 0)
                need[q.dist1][ST_PPF] = true;
            if (composite && q.sub_kind != NFA_PRIOR_CONSTANT && q.dist2 >= 0) need[q.dist2][ST_PPF] = true;
            if (q.kind == NFA_PRIOR_RESOLVED_PLACEMENT && q.dist0 >= 0)
                for (int f : {ST_XAX, ST_PDF, ST_M0, ST_M1, ST_M2}) need[q.dist0][f] = true;
        }
        int n = 0, off = 0;
        for (int d = 0; d < n_dist; ++d)
            for (int f = 0; f < 6; ++f)
                if (need[d][f] && n < MAXSTAGE) { g.stage[n++] = StageItem{d, f, g.ds[d].size, off}; off += g.ds[d].size; }
        g.n_stage = n; g.stage_doubles = off;
        if (off * sizeof(double) > 72 * 1024 || !g_eng.prior_stage) { g.n_stage = 0; g.stage_doubles = 0; }     // too big (or option prior_stage 0): stay in global memory
        // one contiguous image of the staged tables, in LDS order: staging is a flat copy (every load of a
        // workgroup in flight at once) instead of a walk over the table list
        g.stage_image = nullptr;
        if (g.n_stage > 0) {
            std::vector<double> image((size_t)g.stage_doubles);
            for (int q = 0; q < g.n_stage; ++q) {
                const StageItem &it = g.stage[q];
                const nfa_dist_desc &s = dists[it.dist];
                const double *src = it.field == ST_XAX ? s.xax : it.field == ST_PDF ? s.pdf : it.field == ST_PPF ? s.ppf
                                  : moments[it.dist][it.field - ST_M0].data();
                std::copy(src, src + it.n, image.begin() + it.off);
            }
            int rc = upload(image.data(), (int64_t)image.size(), &g.stage_image); if (rc) return rc;
        }
    }
    // Priors that write disjoint parameter slots can be interpreted side by side (one wave each): the
    // reference applies them one after the other (core.pyx:459-476), which only matters if two of them share a slot.
    {
        unsigned seen = 0;
        g.parallel = 1;
        for (int k = 0; k < n_prior; ++k) {
            const nfa_prior_desc &q = priors[k];
            unsigned mine = 1u << (q.p_ix & 31);
            const bool two = q.kind == NFA_PRIOR_DUPLICATE || q.kind == NFA_PRIOR_RESOLVED_CENSEP || q.kind == NFA_PRIOR_RESOLVED_PLACEMENT;
            if (two && q.p_ix2 >= 0) mine |= 1u << (q.p_ix2 & 31);
            if ((seen & mine) || q.p_ix >= 32 || (two && q.p_ix2 >= 32)) g.parallel = 0;
            seen |= mine;
        }
    }
    HIP_TRY(hipMalloc(&p->d_prog, sizeof(PriorProg)));
    HIP_TRY(hipMemcpy(p->d_prog, &p->prog, sizeof(PriorProg), hipMemcpyHostToDevice));
    return NFA_OK;
}

int nfa_priors_create(nfa_priors **out, const nfa_prior_desc *priors, int n_prior,
                      const nfa_dist_desc *dists, int n_dist, int n_param) {
    if (!out || !priors || n_prior < 1 || n_prior > MAXPRIOR || n_dist < 0 || n_dist > MAXDIST)
        return fail(NFA_ERR_ARG, "prior program out of range (<=16 priors, <=16 distributions)");
    int rc = engine_init(); if (rc) return rc;
    nfa_priors *p = new nfa_priors();
    rc = priors_fill(p, priors, n_prior, dists, n_dist, n_param);
    if (rc) { nfa_priors_destroy(p); return rc; }            // frees whatever was uploaded
    *out = p;
    return NFA_OK;
}

int nfa_priors_destroy(nfa_priors *p) {
    if (!p) return NFA_OK;
    (void)hipFree(p->d_prog);
    for (double *d : p->d_arrays) (void)hipFree(d);
    delete p;
    return NFA_OK;
}

static int launch_priors(const nfa_priors *p, double *d_U, int64_t B, int ncomp, hipStream_t st) {
    const int ndim = p->prog.n_param * ncomp;
    const size_t lds = sizeof(double) * 64 * (size_t)ndim;          // theta transposed, one lane per item
    if (lds > 64 * 1024) return fail(NFA_ERR_ARG, "too many parameters for the prior kernel");
    hipLaunchKernelGGL(prior_items_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), lds, st,
                       (const PriorProg *)p->d_prog, d_U, (long)B, ncomp);
    HIP_TRY(hipGetLastError());
    return NFA_OK;
}

int nfa_priors_transform_batch(const nfa_priors *p, double *U, int64_t B, int ncomp, int ndim) {
    if (!p || !U) return fail(NFA_ERR_ARG, "null argument");
    if (ncomp < 1 || p->prog.n_param * ncomp != ndim) {                 // core.pyx:479-483
        char msg[96];
        snprintf(msg, sizeof msg, "Invalid shape for ncomp=%d: %d", ncomp, ndim);
        return fail(NFA_ERR_ARG, msg);
    }
    if (B <= 0) return NFA_OK;
    double *d_U = nullptr;
    HIP_TRY(hipMalloc(&d_U, sizeof(double) * B * ndim));
    HIP_TRY(hipMemcpy(d_U, U, sizeof(double) * B * ndim, hipMemcpyHostToDevice));
    int rc = launch_priors(p, d_U, B, ncomp, 0);
    if (rc) { (void)hipFree(d_U); return rc; }
    HIP_TRY(hipMemcpy(U, d_U, sizeof(double) * B * ndim, hipMemcpyDeviceToHost));
    (void)hipFree(d_U);
    return NFA_OK;
}

// ---- runner ----------------------------------------------------------------
int nfa_runner_create(nfa_runner **out, nfa_specset *ss, nfa_priors *priors, int ncomp,
                      int cold, int lte) {
    if (!out || !ss) return fail(NFA_ERR_ARG, "null argument");
    if (ncomp < 1 || ncomp > MAXCOMP) return fail(NFA_ERR_ARG, "ncomp must be in 1..10");   // ammonia.pyx:401
    if (priors && priors->prog.n_param != ss->dev.npar)
        return fail(NFA_ERR_ARG, "prior program must cover the model's parameters (6 NH3, 4 N2H+, 3 Gaussian)");
    int rc = engine_init(); if (rc) return rc;
    nfa_runner *r = new nfa_runner();
    r->ss = ss; r->pr = priors; r->ncomp = ncomp; r->cold = cold ? 1 : 0; r->lte = lte ? 1 : 0;
    r->ndim = ss->dev.npar * ncomp;
    r->lanes_auto = g_eng.streams == 0;
    r->n_lanes = r->lanes_auto ? 4 : std::max(1, std::min(g_eng.streams, NFA_MAX_LANES));     // automatic: two more on demand (run_batch)
    r->wpb = g_eng.wpb; r->wpb_table = g_eng.wpb_table; r->lnl_cap = g_eng.lnl_cap; r->lnl_split = g_eng.lnl_split;
    for (int k = 0; k < r->n_lanes; ++k) HIP_TRY(hipStreamCreateWithFlags(&r->lanes[k], hipStreamNonBlocking));
    r->stream = r->lanes[0];
    { std::lock_guard<std::mutex> lk(g_runners_m); g_runners.push_back(r); }
    *out = r;
    return NFA_OK;
}

int nfa_runner_destroy(nfa_runner *r) {
    if (!r) return NFA_OK;
    // out of the list first: from here on no global call walks this runner
    { std::lock_guard<std::mutex> lk(g_runners_m); g_runners.erase(std::remove(g_runners.begin(), g_runners.end(), r), g_runners.end()); }
    // batches still held for coalescing are dropped, not launched: their buffers are the caller's, who may have freed
    // them already (whoever wants the results synchronises, and that launches what is held)
    { RUNNER_LOCK(r); r->pending.n = 0; }
    for (int k = 0; k < r->n_lanes; ++k) (void)hipStreamSynchronize(r->lanes[k]);
    (void)hipFree(r->d_U); (void)hipFree(r->d_lnL); (void)hipFree(r->d_pix); (void)hipFree(r->d_spec);
    for (int k = 0; k < r->n_lanes; ++k) { (void)hipFree(r->d_D[k]); (void)hipFree(r->d_part[k]); (void)hipFree(r->d_queue[k]); }
    if (r->g1) (void)hipGraphExecDestroy(r->g1);
    if (r->h_pin) (void)hipHostFree(r->h_pin);
    if (r->h_point) (void)hipHostFree(r->h_point);
    if (r->d_point_done) (void)hipFree(r->d_point_done);
    for (hipEvent_t x : r->ev) (void)hipEventDestroy(x);
    for (int k = 0; k < r->n_lanes; ++k) (void)hipStreamDestroy(r->lanes[k]);
    delete r;
    return NFA_OK;
}

int nfa_runner_ndim(const nfa_runner *r) { return r ? r->ndim : 0; }

int nfa_runner_set_exp_mode(nfa_runner *r, int mode) {
    if (!r) return fail(NFA_ERR_ARG, "null runner");
    RUNNER_LOCK(r);
    if (mode != -1 && mode != 0 && mode != 2) return fail(NFA_ERR_ARG, "exp mode must be -1 (process default), 0 (table) or 2 (fast)");
    { int rc = flush_pending(r); if (rc) return rc; }
    r->exp_mode = mode;
    return NFA_OK;
}
int nfa_runner_get_exp_mode(const nfa_runner *r) { return !r ? -1 : r->exp_mode >= 0 ? r->exp_mode : g_eng.exp_mode; }

static int runner_reserve(nfa_runner *r, int64_t B, bool spec) {
    if (B > r->cap_B) {
        if (r->g1) { (void)hipGraphExecDestroy(r->g1); r->g1 = nullptr; }
        (void)hipFree(r->d_U); (void)hipFree(r->d_lnL); (void)hipFree(r->d_pix);
        r->d_U = nullptr; r->d_lnL = nullptr; r->d_pix = nullptr; r->cap_B = 0;
        const int64_t cap = std::max<int64_t>(B, 64);
        HIP_TRY(hipMalloc(&r->d_U, sizeof(double) * cap * r->ndim));
        HIP_TRY(hipMalloc(&r->d_lnL, sizeof(double) * cap));
        HIP_TRY(hipMalloc(&r->d_pix, sizeof(int) * cap));
        r->cap_B = cap;
    }
    if (spec && B > r->cap_spec) {
        (void)hipFree(r->d_spec); r->d_spec = nullptr; r->cap_spec = 0;
        HIP_TRY(hipMalloc(&r->d_spec, sizeof(double) * B * r->ss->dev.chan_tot));
        r->cap_spec = B;
    }
    return NFA_OK;
}

}  // extern "C" (templates need C++ linkage)

static SpecDev runner_specdev(const nfa_runner *r) {
    SpecDev S = r->ss->dev;
    S.ncomp = r->ncomp; S.cold = r->cold; S.lte = r->lte;
    S.t0_xmin = g_eng.t0_xmin; S.t0_xmax = g_eng.t0_xmax; S.t0_inv_dx = g_eng.t0_inv_dx;
    return S;
}

// Set-up stage of a batch on stream lane `slot`: [unit cube -> theta in place] -> partition sums ->
// derived records r->d_D[slot], one launch (setup_kernel, nfa_setup.h)
// A kernel that wants more than 64 KB of dynamic LDS has to be told so -- once per kernel and size, not on every
// launch (the attribute call is a trip into the runtime: 1-2 us of the ~10 the host spends on enqueueing a step).
static int ensure_dynamic_lds(const void *kernel, size_t lds) {
    if (lds <= 64 * 1024) return NFA_OK;
    struct Grant { const void *kernel; int device; size_t lds; };        // the attribute belongs to a kernel on a device
    static std::mutex m;
    static std::vector<Grant> granted;
    std::lock_guard<std::mutex> lk(m);
    for (auto &g : granted)
        if (g.kernel == kernel && g.device == g_eng.device) {
            if (g.lds >= lds) return NFA_OK;
            HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            g.lds = lds;
            return NFA_OK;
        }
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    granted.push_back(Grant{kernel, g_eng.device, lds});
    return NFA_OK;
}

// derived records and chi^2 parts of stream lane `slot` for B items
static int reserve_lane(nfa_runner *r, int slot, int64_t B) {
    if (B <= r->cap_D[slot]) return NFA_OK;  // grown outside any timed loop
    const int n_spec = r->ss->dev.n_spec;
    if (slot == 0 && r->g1) { (void)hipGraphExecDestroy(r->g1); r->g1 = nullptr; }
    HIP_TRY(hipStreamSynchronize(r->lanes[slot]));
    (void)hipFree(r->d_D[slot]); (void)hipFree(r->d_part[slot]);
    r->d_D[slot] = nullptr; r->d_part[slot] = nullptr; r->cap_D[slot] = 0;
    const int64_t cap = std::max<int64_t>(B, 4096);
    HIP_TRY(hipMalloc(&r->d_D[slot], sizeof(double) * cap * drec_size(r->ncomp, n_spec)));
    HIP_TRY(hipMalloc(&r->d_part[slot], sizeof(double) * cap * n_spec));
    if (!r->d_queue[slot]) {
        HIP_TRY(hipMalloc(&r->d_queue[slot], sizeof(unsigned) * NFA_QUEUE_WORDS));
        HIP_TRY(hipMemset(r->d_queue[slot], 0, sizeof(unsigned) * NFA_QUEUE_WORDS));
    }
    r->cap_D[slot] = cap;
    return NFA_OK;
}

// LDS of the set-up stage: exponential tables, theta + partition records + the prior program and its tables
static bool setup_uses_tables(const nfa_runner *r, int mode) { return mode == 0 && r->ss->dev.model == NFA_MODEL_AMMONIA; }
static size_t setup_lds_bytes(const nfa_runner *r, int mode, bool has_prior, int nsub = 1) {
    const size_t work = (size_t)nsub * ((size_t)64 * r->ndim + (size_t)SETUP_TI * r->ncomp * QREC) + sizeof(PriorProg) / sizeof(double) + 1
                        + (has_prior ? (size_t)r->pr->prog.stage_doubles : 0);
    return sizeof(double) * ((setup_uses_tables(r, mode) ? (SM_END_TABLE - SM_EXP2) : NFA_EXP2_N) + work);
}

static int launch_setup(nfa_runner *r, double *d_U, int64_t B, bool has_prior, int slot, int mode) {
    const SpecDev S = runner_specdev(r);
    hipStream_t st = r->lanes[slot];
    int rc = reserve_lane(r, slot, B); if (rc) return rc;
    if (has_prior && !r->pr) return fail(NFA_ERR_STATE, "runner has no priors (predict-only)");
    const PriorProg *prog = has_prior ? (const PriorProg *)r->pr->d_prog : nullptr;
    // items per workgroup and waves per workgroup (options setup_ti, setup_threads: A/B knobs)
    const int ti = g_eng.setup_ti > 0 ? g_eng.setup_ti : SETUP_TI;
    const bool tables = setup_uses_tables(r, mode);
    // Eight waves per workgroup where the partition sums go through FastExp's tables (52 KB of LDS per workgroup: two per
    // CU whatever their size, and the sums are eight rounds of a four-wave workgroup): 57.7 -> 48.8 us per 32768 items,
    // 88.4 -> 90.5 M evaluations/s on the metric shape.  The polynomial's set-up (fast mode) is faster with four
    // (156.5 against 149.8 M): its workgroups are many per CU (scripts/gpu_setup_shape.sh).
    int threads = g_eng.setup_threads > 0 ? g_eng.setup_threads : tables ? 2 * SETUP_THREADS : SETUP_THREADS;
    // ... and two such groups per workgroup behind one copy of the tables (115 KB of LDS for one group: one workgroup per CU
    // and two rounds of them for 32768 items; 133 KB for two: one round): 48.4 -> see profiles/r05/ab_table_linestep.txt.
    // Every batch of a group must hold whole workgroups; a launch of ONE batch may have any size (the last workgroup's
    // second group then has fewer items, or none: the sampler's batches).  Small launches keep one group per workgroup:
    // they are latency, not rounds.
    int nsub = 1;
    const bool whole = r->cur_group.n <= 1 || (B % (2 * ti) == 0 && r->cur_group.each % (2 * ti) == 0);
    if (tables && g_eng.setup_threads == 0 && g_eng.setup_sub != 1 && ti == SETUP_TI && whole && B > (int64_t)ti * g_eng.n_cu
        && setup_lds_bytes(r, mode, has_prior, 2) <= 160 * 1024) {
        nsub = 2;
        threads = 1024;
    }
    const unsigned blocks = (unsigned)((B + (int64_t)ti * nsub - 1) / ((int64_t)ti * nsub));
    const size_t lds = setup_lds_bytes(r, mode, has_prior, nsub);
    if (lds > 160 * 1024) return fail(NFA_ERR_ARG, "too many parameters for the set-up kernel");
    auto kern = nsub == 2 ? setup_kernel<0, false, 2> : tables ? setup_kernel<0, false> : mode == 2 ? setup_kernel<1, true> : setup_kernel<1, false>;
    { int rc2 = ensure_dynamic_lds((const void *)kern, lds); if (rc2) return rc2; }
    (void)d_U;                                               // the batches' arrays travel in r->cur_group
    if (r->ev_cur)      // profiling: the events ride on the dispatch itself -- its own start and stop, as a tracer sees them
        hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, st, r->ev_cur[0], r->ev_cur[1], 0, prog, S, r->cur_group, r->d_D[slot], (long)B,
                              has_prior ? 1 : 0, (const double *)g_eng.d_tabs, g_eng.ablate, ti);
    else
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, st, prog, S, r->cur_group, r->d_D[slot], (long)B,
                           has_prior ? 1 : 0, (const double *)g_eng.d_tabs, g_eng.ablate, ti);
    HIP_TRY(hipGetLastError());
    return NFA_OK;
}

// the fast mode's narrow form (FastRec records, fp32 window test): at most 26 lines per transition (a 32-bit line
// mask per component) and channel indices that fp32 holds to the half (nfa_device.h: FastRec)
static bool lnl_wide(const nfa_runner *r) {
    int max_size = 0;
    for (int k = 0; k < r->ss->dev.n_spec; ++k) max_size = std::max(max_size, r->ss->dev.size[k]);
    return r->ss->nhf_max > 26 || max_size > (1 << 22);
}
// LDS doubles per (item, spectrum) unit: the line table (32-byte records, nhf_max per component) followed by the
// windows (two ints per line)
static int lnl_wave_doubles(const nfa_runner *r) {
    const int per_line = (int)(sizeof(LineRec) / sizeof(double)) + 1;
    return (r->ncomp * r->ss->nhf_max * per_line + 1) & ~1;          // 16-byte records: an even number of doubles
}

// waves per workgroup of a table-mode launch with one wave per unit: the workgroup stages 51 KB of product tables, so it
// is made as fat as keeps the most waves resident per CU (ties: more workgroups, so that one stages while another computes)
static int table_waves(const nfa_runner *r) {
    if (r->wpb_table > 0) return r->wpb_table;
    const int n_shared = SM_END_TABLE - SM_EXP2, wave_doubles = lnl_wave_doubles(r);
    int best = -1, best_blocks = 0, waves = 16;
    for (int w = 4; w <= 16; w += 2) {
        const size_t need = sizeof(double) * ((size_t)n_shared + (size_t)wave_doubles * w);
        const int blocks = (int)((160 * 1024) / need);
        const int resident = std::min(32, blocks * w);
        if (resident > best || (resident == best && blocks > best_blocks)) { best = resident; best_blocks = blocks; waves = w; }
    }
    return waves;
}

// waves per unit of a launch of B items (runner option lnl_split; 0 = by the size of the launch)
static int resolve_split(const nfa_runner *r, const SpecDev &S, int64_t B) {
    int split = r->lnl_split;
    if (split == 0) {
        const int64_t slots = (int64_t)g_eng.n_cu * 32;
        split = 1;
        while (split < LNL_PARTS && B * S.n_spec * split * 2 <= slots) split *= 2;
    }
    int min_rows = 1 << 30;
    for (int k = 0; k < S.n_spec; ++k) min_rows = std::min(min_rows, (S.size[k] + 63) / 64);
    while (split > 1 && split > min_rows) split /= 2;
    return split;
}

// table mode, one wave per unit, at least two units per wave slot of the device: as many workgroups as are resident at
// once, the units drawn from the launch's queue (lnl_kernel_queue)
// workgroups of `waves` waves of the table mode that a CU holds at once (LDS: the tables, the waves' line tables, the queue's words)
static int table_wg_per_cu(const nfa_runner *r, int waves) {
    const size_t need = sizeof(double) * ((size_t)(SM_END_TABLE - SM_EXP2) + (size_t)lnl_wave_doubles(r) * waves) + 16;
    return std::max(1, std::min((int)((160 * 1024) / need), 32 / waves));
}
static bool lnl_uses_queue(const nfa_runner *r, const SpecDev &S, int64_t B, int mode) {
    if (mode != 0 || g_eng.lnl_queue == 0 || lnl_wide(r)) return false;
    if (resolve_split(r, S, B) != 1) return false;
    // (short units -- config 1's 256 channels are four rows -- finish before the draw has paid: 348 M evaluations/s one
    // unit per wave against 335 M through the queue; from eight rows per spectrum on)
    for (int k = 0; k < S.n_spec; ++k) if (S.size[k] < 512) return false;
    const int waves = table_waves(r);
    return B * S.n_spec >= 2 * ((int64_t)g_eng.n_cu * table_wg_per_cu(r, waves)) * waves;     // two units per resident wave and more
}

template <int MODE, bool WS, bool WIDE, int NCOMP>
static int launch_lnl_t(nfa_runner *r, const int *d_pix, int slot, double *d_lnL,
                        double *d_spec, int64_t B) {
    const SpecDev S = runner_specdev(r);
    LnlGeom G;
    G.ablate = g_eng.ablate;
    G.nhf_max = r->ss->nhf_max;
    G.inv_nspec = S.n_spec == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)S.n_spec) + 1u;
    G.inv_nhf = G.nhf_max == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)G.nhf_max) + 1u;
    if (B * S.n_spec * 8 >= (1LL << 28)) return fail(NFA_ERR_ARG, "batch too large for one launch");
    // Waves per unit.  A launch with fewer units than a few per wave slot is latency bound: its waves are
    // placed once and every SIMD waits for its own longest; splitting the rows of a unit over 2 or 4 waves
    // gives the hardware shorter waves to place as slots free up (a single point: 2 units -> 8 waves).
    const int split = resolve_split(r, S, B);
    G.split = split;
    G.wave_doubles = lnl_wave_doubles(r);
    const int n_shared = (MODE == 0) ? (SM_END_TABLE - SM_EXP2) : 0;
    // waves per workgroup.  Table mode stages 51 KB of product tables per workgroup, so the
    // workgroup is made as fat as keeps the most waves resident per CU (ties: more workgroups,
    // so that one stages while another computes).
    int waves = std::max(1, std::min(r->wpb, 16));
    waves = std::max(waves, split);
    waves -= waves % split;
    if (MODE == 0 && split > 1) waves = std::max(8, split);
    if (MODE == 0 && split == 1) waves = table_waves(r);
    G.queue = nullptr;
#ifdef NFA_TEST_HOOKS
    G.trace = g_eng.d_trace;
#endif
    const int64_t n_units = B * S.n_spec, wg_resident = (int64_t)g_eng.n_cu * (g_eng.lnl_queue_wg > 0 ? g_eng.lnl_queue_wg : table_wg_per_cu(r, waves));
    if (MODE == 0 && !WIDE && r->d_queue[slot] && lnl_uses_queue(r, S, B, 0)) G.queue = r->d_queue[slot];
    size_t lds = sizeof(double) * ((size_t)n_shared + ((size_t)G.wave_doubles + (split > 1 ? LNL_PARTS * 64 : 0)) * (waves / split))
               + (G.queue ? 16 : 0);
    if (MODE == 0) lds = std::max(lds, sizeof(double) * (size_t)(n_shared + SM_TABLE_TAIL));
    if (lds > 160 * 1024) return fail(NFA_ERR_ARG, "ncomp too large for the LDS line table");
    if (MODE != 0 && r->lnl_cap > 0 && waves * r->lnl_cap < 32)      // residency cap: see Engine::lnl_cap
        lds = std::max(lds, (size_t)((160 * 1024) / r->lnl_cap) & ~(size_t)15);
    void (*kern)(SpecDev, BatchGroup, const double *, double *, double *, long, LnlGeom, const double *) = lnl_kernel<MODE, WS, WIDE, NCOMP>;
    if constexpr (MODE == 0 && WS) kern = lnl_kernel_w8<MODE, WS, WIDE, NCOMP>;
    if constexpr (MODE == 0 && !WIDE) { if (G.queue) kern = lnl_kernel_queue<WS, NCOMP>; }
    { int rc2 = ensure_dynamic_lds((const void *)kern, lds); if (rc2) return rc2; }
    const int64_t units = B * S.n_spec;
    const int64_t upw = waves / split;
    const int64_t blocks = G.queue ? wg_resident : (units + upw - 1) / upw;
    if (blocks > 0x7fffffffLL) return fail(NFA_ERR_ARG, "batch too large for one launch");
    hipStream_t st = r->lanes[slot];
    (void)d_pix;
    if (r->ev_cur) {
        hipExtLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * waves), lds, st, r->ev_cur[2], r->ev_cur[3], 0, S, r->cur_group,
                              (const double *)r->d_D[slot], d_lnL ? r->d_part[slot] : nullptr, d_spec, (long)B, G,
                              (const double *)g_eng.d_tabs);
        r->ev_cur = nullptr;
    } else
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * waves), lds, st, S, r->cur_group,
                           (const double *)r->d_D[slot], d_lnL ? r->d_part[slot] : nullptr, d_spec, (long)B, G,
                           (const double *)g_eng.d_tabs);
    HIP_TRY(hipGetLastError());
    if (d_lnL && !r->part_only) {
        hipLaunchKernelGGL(lnl_sum_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st,
                           (const double *)r->d_part[slot], S.noise, r->cur_group, (long)B, S.n_spec);
        HIP_TRY(hipGetLastError());
    }
    return NFA_OK;
}

// component count: 1..3 are compiled with the component loop unrolled, anything else takes the general form
template <int MODE, bool WS, bool WIDE>
static int launch_lnl_n(nfa_runner *r, const int *d_pix, int slot, double *d_lnL, double *d_spec, int64_t B) {
    switch (r->ncomp) {
    case 1: return launch_lnl_t<MODE, WS, WIDE, 1>(r, d_pix, slot, d_lnL, d_spec, B);
    case 2: return launch_lnl_t<MODE, WS, WIDE, 2>(r, d_pix, slot, d_lnL, d_spec, B);
    case 3: return launch_lnl_t<MODE, WS, WIDE, 3>(r, d_pix, slot, d_lnL, d_spec, B);
    default: return launch_lnl_t<MODE, WS, WIDE, 0>(r, d_pix, slot, d_lnL, d_spec, B);
    }
}

static int launch_lnl(nfa_runner *r, const int *d_pix, int slot, double *d_lnL, double *d_spec, int64_t B,
                      int mode) {
    switch (mode) {
    case 0:
        if (lnl_wide(r))                // more than 26 lines in a transition (N2H+ 2-1, 3-2): 64-bit line masks, a mask per component
            return d_spec ? launch_lnl_n<0, true, true>(r, d_pix, slot, d_lnL, d_spec, B)
                          : launch_lnl_n<0, false, true>(r, d_pix, slot, d_lnL, d_spec, B);
        return d_spec ? launch_lnl_n<0, true, false>(r, d_pix, slot, d_lnL, d_spec, B)
                      : launch_lnl_n<0, false, false>(r, d_pix, slot, d_lnL, d_spec, B);
    default:
        if (lnl_wide(r))                // more lines than any NH3 transition (or 2^22 channels): fp64 running sum of tau
            return d_spec ? launch_lnl_n<2, true, true>(r, d_pix, slot, d_lnL, d_spec, B)
                          : launch_lnl_n<2, false, true>(r, d_pix, slot, d_lnL, d_spec, B);
        return d_spec ? launch_lnl_n<2, true, false>(r, d_pix, slot, d_lnL, d_spec, B)
                      : launch_lnl_n<2, false, false>(r, d_pix, slot, d_lnL, d_spec, B);
    }
}

// One batch on the next stream lane: set-up kernel, then likelihood kernel.  `lane_out`
// receives the lane (stream) the batch was enqueued on.
static int run_group(nfa_runner *r, const BatchGroup &grp, double *d_spec, bool has_prior, int force_lane, int *lane_out);

static int run_batch(nfa_runner *r, const int *d_pix, double *d_U, double *d_lnL, double *d_spec,
                     int64_t B, bool has_prior, int force_lane, int *lane_out) {
    BatchGroup g = {};
    g.pix[0] = d_pix; g.U[0] = d_U; g.lnL[0] = d_lnL; g.each = (long)B; g.n = 1;
    return run_group(r, g, d_spec, has_prior, force_lane, lane_out);
}

// The batches of `grp` (one, or several of the same shape coalesced) as one set of launches on the next lane.
static int run_group(nfa_runner *r, const BatchGroup &grp, double *d_spec, bool has_prior, int force_lane, int *lane_out) {
    const int64_t B = (int64_t)grp.each * grp.n;
    const int *d_pix = grp.pix[0];
    double *d_U = grp.U[0], *d_lnL = grp.lnL[0];
    int rc0 = engine_init(); if (rc0) return rc0;            // binds the calling thread to the device
    if (!g_eng.have_t0) return fail(NFA_ERR_STATE, "nfa_set_iemtex_table has not been called");
    // Lanes a sequence of batches rotates over.  Four overlap the draining tail of one batch with the next;
    // a batch of about one wave per wave slot (the metric's 4096 rows x 2 spectra) leaves the longest tail and
    // gains another 3 % from six, smaller and larger ones lose with more than four (profiles/r02/sweep_lanes.txt).
    int n_use = r->n_lanes;
    if (r->lanes_auto) {
        const int64_t units = B * r->ss->dev.n_spec, slots = (int64_t)g_eng.n_cu * 32;
        n_use = (4 * units >= 3 * slots && 2 * units <= 3 * slots) ? 6 : 4;
        // The fifth and sixth stream exist only once a batch of that size has come by: idle streams are not free --
        // with six streams mapped the small launches of a sampler round trip 25 % slower even on the three they use
        // (config 5, one component: 1.25 -> 1.56 s).
        while (r->n_lanes < n_use && force_lane < 0) {
            HIP_TRY(hipStreamCreateWithFlags(&r->lanes[r->n_lanes], hipStreamNonBlocking));
            r->n_lanes += 1;
        }
        n_use = std::min(n_use, r->n_lanes);
    }
    const int slot = force_lane >= 0 ? force_lane : (int)(r->n_calls % (uint64_t)n_use);
    hipStream_t st = r->lanes[slot];
    hipEvent_t *e = nullptr;
    if (r->profiling) {
        if (r->ev_used + 4 > r->ev.size()) {
            for (int k = 0; k < 4; ++k) { hipEvent_t x; HIP_TRY(hipEventCreate(&x)); r->ev.push_back(x); }
        }
        e = &r->ev[r->ev_used];
        r->ev_used += 4;
        r->ev_cur = e;
    }
    const int mode = r->exp_mode >= 0 ? r->exp_mode : g_eng.exp_mode;      // read once per batch
    r->cur_group = grp;
    int rc = launch_setup(r, d_U, B, has_prior, slot, mode);
    if (rc) return rc;
    rc = launch_lnl(r, d_pix, slot, d_lnL, d_spec, B, mode);
    if (rc) return rc;
    if (r->ev_cur) {        // (a likelihood launcher that does not carry events: the pair goes behind it, an empty interval)
        HIP_TRY(hipEventRecord(r->ev_cur[2], st)); HIP_TRY(hipEventRecord(r->ev_cur[3], st));
        r->ev_cur = nullptr;
    }
    r->n_calls++;
    r->lane_busy |= 1u << slot;
    if (lane_out) *lane_out = slot;
    return NFA_OK;
}

// Coalescing of device-pointer batches.  nfa_runner_loglike_batch_dev returns before anything runs anyway; batches
// of one shape that arrive back to back are held (at most `coalesce` of them) and launched together: a launch of
// four times 4096 rows keeps the vector ALUs busy 94 % of the time, four launches of 4096 rows overlapping on
// their lanes 79 %.  Everything that looks at results, changes how launches are made or uses the lanes
// itself launches what is held first (flush_pending).
static int flush_pending(nfa_runner *r) {
    if (r->pending.n == 0) return NFA_OK;
    const BatchGroup g = r->pending;
    r->pending.n = 0;
    return run_group(r, g, g.spec[0], r->pending_prior, -1, nullptr);
}
static int flush_all_runners() {
    std::lock_guard<std::mutex> lk(g_runners_m);
    for (nfa_runner *r : g_runners) {
        RUNNER_LOCK(r);
        int rc = flush_pending(r); if (rc) return rc;
    }
    return NFA_OK;
}

static int sync_all_lanes(nfa_runner *r) {
    { int rc = flush_pending(r); if (rc) return rc; }
    // only lanes that had work enqueued since their last synchronisation (a synchronise call on an
    // idle stream still costs a couple of microseconds, and single-point callers pay it per point)
    for (int k = 0; k < r->n_lanes; ++k)
        if (r->lane_busy & (1u << k)) HIP_TRY(hipStreamSynchronize(r->lanes[k]));
    r->lane_busy = 0;
    return NFA_OK;
}

static int check_pix(const nfa_runner *r, const int32_t *pix, int64_t B) {
    if (!pix) return NFA_OK;
    for (int64_t b = 0; b < B; ++b)
        if (pix[b] < 0 || pix[b] >= r->ss->n_pix) return fail(NFA_ERR_ARG, "pixel index out of range");
    return NFA_OK;
}

extern "C" {

// a device-pointer batch: launched with its neighbours of the same kind and shape, or on its own
static int enqueue_dev(nfa_runner *r, const int32_t *d_pix, double *d_U, double *d_lnL, double *d_spec, int64_t B, bool has_prior) {
    const int ti = g_eng.setup_ti > 0 ? g_eng.setup_ti : SETUP_TI;
    const int64_t units = B * r->ss->dev.n_spec, slots = (int64_t)g_eng.n_cu * 32;
    BatchGroup &p = r->pending;
    const int group = g_eng.coalesce;                  // read per call: a knob, not part of a runner's identity
    const bool fits = group > 1 && !r->profiling && B % ti == 0 && 2 * units <= NFA_GROUP_MAX * slots;   // (a group stays below NFA_GROUP_MAX waves per slot)
    if (p.n > 0 && (!fits || p.each != (long)B || r->pending_prior != has_prior || (p.pix[0] == nullptr) != (d_pix == nullptr) ||
                    (p.lnL[0] == nullptr) != (d_lnL == nullptr) || (p.spec[0] == nullptr) != (d_spec == nullptr) ||
                    (int64_t)(p.n + 1) * units > NFA_GROUP_MAX * slots)) {
        int rc = flush_pending(r); if (rc) return rc;
    }
    if (!fits) return run_batch(r, d_pix, d_U, d_lnL, d_spec, B, has_prior, -1, nullptr);
    p.pix[p.n] = d_pix; p.U[p.n] = d_U; p.lnL[p.n] = d_lnL; p.spec[p.n] = d_spec; p.each = (long)B; p.n += 1;
    r->pending_prior = has_prior;
    if (p.n >= group || (int64_t)(p.n + 1) * units > NFA_GROUP_MAX * slots) return flush_pending(r);
    return NFA_OK;
}

int nfa_runner_loglike_batch_dev(nfa_runner *r, const int32_t *d_pix, double *d_U, double *d_lnL,
                                 int64_t B) {
    if (!r || !d_U || !d_lnL) return fail(NFA_ERR_ARG, "null argument");
    if (!r->pr) return fail(NFA_ERR_STATE, "runner has no priors (predict-only)");
    if (B <= 0) return NFA_OK;
    RUNNER_LOCK(r);
    return enqueue_dev(r, d_pix, d_U, d_lnL, nullptr, B, true);
}

int nfa_runner_set_profiling(nfa_runner *r, int on) {
    if (!r) return fail(NFA_ERR_ARG, "null runner");
    RUNNER_LOCK(r);
    int rc = sync_all_lanes(r); if (rc) return rc;
    r->profiling = on != 0;
    r->ev_used = 0;
    return NFA_OK;
}

// length of the union of intervals [a_k, b_k] (milliseconds)
static double union_length(std::vector<std::pair<double, double>> iv) {
    std::sort(iv.begin(), iv.end());
    double tot = 0, cur_a = 0, cur_b = -1;
    for (auto &p : iv) {
        if (cur_b < cur_a || p.first > cur_b) {
            if (cur_b >= cur_a) tot += cur_b - cur_a;
            cur_a = p.first; cur_b = p.second;
        } else if (p.second > cur_b) cur_b = p.second;
    }
    if (cur_b >= cur_a) tot += cur_b - cur_a;
    return tot;
}

int nfa_runner_get_profile(nfa_runner *r, double *out, int64_t *calls) {
    if (!r || !out || !calls) return fail(NFA_ERR_ARG, "null argument");
    RUNNER_LOCK(r);
    int rc = sync_all_lanes(r); if (rc) return rc;
    double a = 0, b = 0;
    std::vector<std::pair<double, double>> iv_setup, iv_lnl;
    const size_t n = std::min(r->ev_used, r->ev.size()) / 4;
    // (the first profile_skip calls are left out: launches behind an idle gap run at the clocks the chip had idled at, a
    // few milliseconds of load later the same launch is 10 % shorter -- a timed block is long, a probe of 15 launches is not)
    const size_t k0 = std::min<size_t>(n, (size_t)std::max(0, g_eng.profile_skip));
    for (size_t k = k0; k < n; ++k) {
        float t0 = 0, t1 = 0, t2 = 0, t3 = 0;  // times since the first recorded event
        HIP_TRY(hipEventElapsedTime(&t0, r->ev[0], r->ev[4 * k]));
        HIP_TRY(hipEventElapsedTime(&t1, r->ev[0], r->ev[4 * k + 1]));
        HIP_TRY(hipEventElapsedTime(&t2, r->ev[0], r->ev[4 * k + 2]));
        HIP_TRY(hipEventElapsedTime(&t3, r->ev[0], r->ev[4 * k + 3]));
        a += t1 - t0; b += t3 - t2;
        iv_setup.emplace_back(t0, t1);
        iv_lnl.emplace_back(t2, t3);
    }
    out[0] = a;                          // sum of set-up kernel durations
    out[1] = b;                          // sum of lnl_kernel durations (lnl_sum_kernel not included)
    out[2] = union_length(iv_setup);     // time during which >= 1 set-up kernel was running
    out[3] = union_length(iv_lnl);       // time during which >= 1 likelihood kernel was running
    *calls = (int64_t)(n - k0);
    r->ev_used = 0;
    return NFA_OK;
}

int nfa_runner_synchronize(nfa_runner *r) {
    if (!r) return fail(NFA_ERR_ARG, "null runner");
    RUNNER_LOCK(r);
    return sync_all_lanes(r);
}

}  // extern "C"

// One point, or the few a broker gathered, through the point kernel (nfa_setup.h); returns 1 when the call
// was served, 0 when another path has to do it, a negative value on a device error.
#define POINT_HOST_DOUBLES (NFA_POINT_MAXB * (2 * NFA_POINT_MAXDIM + 2) + 8)
template <int MODE, int NCOMP>
static void launch_point_t(nfa_runner *r, const SpecDev &S, const PointIn &in, const LnlGeom &G, size_t lds) {
    auto kern = point_kernel<MODE, NCOMP>;
    (void)ensure_dynamic_lds((const void *)kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)in.n), dim3(POINT_THREADS), lds, r->lanes[0], (const PriorProg *)r->pr->d_prog, S, in,
                       r->d_pix, r->d_U, r->d_D[0], r->d_part[0], r->d_point, r->d_point_done, G,
                       (const double *)g_eng.d_tabs);
}
template <int MODE>
static void launch_point_n(nfa_runner *r, const SpecDev &S, const PointIn &in, const LnlGeom &G, size_t lds) {
    switch (r->ncomp) {
    case 1: return launch_point_t<MODE, 1>(r, S, in, G, lds);
    case 2: return launch_point_t<MODE, 2>(r, S, in, G, lds);
    case 3: return launch_point_t<MODE, 3>(r, S, in, G, lds);
    default: return launch_point_t<MODE, 0>(r, S, in, G, lds);
    }
}

static int few_points_kernel(nfa_runner *r, const int32_t *pix, double *U, double *lnL, int64_t B) {
    const int ndim = r->ndim;
    if (!g_eng.point || r->profiling || ndim > NFA_POINT_MAXDIM || lnl_wide(r) || B > NFA_POINT_MAXB) return 0;
    const int mode = r->exp_mode >= 0 ? r->exp_mode : g_eng.exp_mode;
    const SpecDev S = runner_specdev(r);
    LnlGeom G;
    G.ablate = 0;
    G.nhf_max = r->ss->nhf_max;
    G.inv_nspec = S.n_spec == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)S.n_spec) + 1u;
    G.inv_nhf = G.nhf_max == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)G.nhf_max) + 1u;
    G.split = resolve_split(r, S, 1);
    if (G.split > POINT_WAVES) return 0;
    G.wave_doubles = lnl_wave_doubles(r);
    const int upw = POINT_WAVES / G.split;                       // units per pass of the workgroup
    const int n_shared = mode == 0 ? (SM_END_TABLE - SM_EXP2) : 0;
    // the set-up stage and the likelihood waves use the same LDS one after the other, behind the staged tables
    const size_t n_staged = mode == 0 ? (SM_END_TABLE - SM_EXP2) : NFA_EXP2_N;
    const size_t lds = std::max(setup_lds_bytes(r, 1, true) + sizeof(double) * (n_staged - NFA_EXP2_N),
                                sizeof(double) * ((size_t)n_shared + ((size_t)G.wave_doubles + (G.split > 1 ? LNL_PARTS * 64 : 0)) * upw));
    if (lds > 160 * 1024) return 0;
    if (reserve_lane(r, 0, B) != NFA_OK) return -1;
    if (!r->h_point) {
        bool ok = hipHostMalloc((void **)&r->h_point, sizeof(double) * POINT_HOST_DOUBLES, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess
                  && hipHostGetDevicePointer((void **)&r->d_point, r->h_point, 0) == hipSuccess
                  && hipMalloc((void **)&r->d_point_done, sizeof(unsigned)) == hipSuccess
                  && hipMemset(r->d_point_done, 0, sizeof(unsigned)) == hipSuccess
                  && hipDeviceSynchronize() == hipSuccess;        // (a null-stream memset is not ordered before the lanes' work)
        if (!ok) {
            (void)hipGetLastError();
            if (r->h_point) (void)hipHostFree(r->h_point);
            if (r->d_point_done) (void)hipFree(r->d_point_done);
            r->h_point = nullptr; r->d_point_done = nullptr;
            g_eng.point = 0;
            return 0;
        }
        memset(r->h_point, 0, sizeof(double) * POINT_HOST_DOUBLES);
    }
    PointIn in;
    memset(&in, 0, sizeof in);
    in.seq = ++r->pt_seq;
    in.n = (int)B;
    in.n_blocks = (S.n_spec + upw - 1) / upw;
    if (B == 1) {
        memcpy(in.u, U, sizeof(double) * ndim);
        in.pix = pix ? pix[0] : -1;
    } else {                                                     // the unit cubes travel through the mapped buffer
        double *in_u = r->h_point + B * (ndim + 1) + 1;
        memcpy(in_u, U, sizeof(double) * B * ndim);
        if (pix) memcpy(in_u + B * ndim, pix, sizeof(int32_t) * B);
        in.pix = pix ? 0 : -1;
    }
    volatile unsigned long long *flag = (volatile unsigned long long *)(r->h_point + B * (ndim + 1));
    *flag = 0;                                                   // the slot holds other data when B changes
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    switch (mode) {
    case 0: launch_point_n<0>(r, S, in, G, lds); break;
    default: launch_point_n<2>(r, S, in, G, lds); break;
    }
    if (hipGetLastError() != hipSuccess) { fail(NFA_ERR_DEVICE, "point kernel launch failed"); return -1; }
    // the kernel's last store is the sequence number; the host reads it straight from the mapped buffer
    const auto t_start = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == in.seq) break;
        if ((spins & 0xffff) == 0xffff
            && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) {
            // nothing came back: let the runtime say why (a fault surfaces here), or finish a very slow kernel
            if (hipStreamSynchronize(r->lanes[0]) != hipSuccess) { fail(NFA_ERR_DEVICE, "point kernel failed"); return -1; }
        }
    }
    memcpy(U, r->h_point, sizeof(double) * B * ndim);
    memcpy(lnL, r->h_point + B * ndim, sizeof(double) * B);
    return 1;
}

// One point through a captured graph; returns 1 when the call was served, 0 when the plain path
// has to do it (first calls, table mode whose launch sets a function attribute, profiling on).
static int single_point_graph(nfa_runner *r, double *U, double *lnL) {
    if (g_eng.graph < 0) {
        // Stream capture under the rocprofiler-sdk tool library (rocprofv3) has crashed the process
        // here: with a profiler attached single points take the plain path unless asked otherwise.
        bool profiled = false;
        for (const char *name : {"ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIB", "HSA_TOOLS_LIB"}) {
            const char *v = getenv(name);
            profiled = profiled || (v && *v);
        }
        const char *pre = getenv("LD_PRELOAD");
        profiled = profiled || (pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer")));
        g_eng.graph = profiled ? 0 : 1;
    }
    const int mode = r->exp_mode >= 0 ? r->exp_mode : g_eng.exp_mode;
    if (!g_eng.graph || r->profiling || mode == 0 || r->cap_B < 1 || r->cap_D[0] < 1 || r->n_single < 2) return 0;
    const int ndim = r->ndim;
    hipStream_t st = r->lanes[0];
    if (!r->h_pin && hipHostMalloc((void **)&r->h_pin, sizeof(double) * (ndim + 1)) != hipSuccess) return 0;
    if (r->g1 && r->g1_mode != mode) { (void)hipGraphExecDestroy(r->g1); r->g1 = nullptr; }
    if (!r->g1) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) return 0;
        bool ok = hipMemcpyAsync(r->d_U, r->h_pin, sizeof(double) * ndim, hipMemcpyHostToDevice, st) == hipSuccess;
        ok = ok && run_batch(r, nullptr, r->d_U, r->d_lnL, nullptr, 1, true, 0, nullptr) == NFA_OK;
        ok = ok && hipMemcpyAsync(r->h_pin, r->d_U, sizeof(double) * ndim, hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipMemcpyAsync(r->h_pin + ndim, r->d_lnL, sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess;
        const bool ended = hipStreamEndCapture(st, &graph) == hipSuccess && graph;
        if (!ok || !ended || hipGraphInstantiate(&r->g1, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            r->g1 = nullptr;
            (void)hipGetLastError();
            r->n_single = 0;                         // do not try again right away
            return 0;
        }
        (void)hipGraphDestroy(graph);
        r->g1_mode = mode;
    }
    memcpy(r->h_pin, U, sizeof(double) * ndim);
    if (hipGraphLaunch(r->g1, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0;
    r->lane_busy &= ~1u;
    memcpy(U, r->h_pin, sizeof(double) * ndim);
    *lnL = r->h_pin[ndim];
    return 1;
}

extern "C" {

// The device's view of host memory it can address (pinned and mapped: nfa_host_alloc, hipHostMalloc,
// hipHostRegister), nullptr for ordinary pageable memory.
static void *mapped_view(const void *host) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    return a.devicePointer;
}

// Pinned, device-addressable host memory for the buffers of the host-pointer entry points.
int nfa_host_alloc(void **out, size_t bytes) {
    if (!out || bytes == 0) return fail(NFA_ERR_ARG, "null argument");
    int rc = engine_init(); if (rc) return rc;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocPortable));
    return NFA_OK;
}
int nfa_host_free(void *p) {
    if (!p) return NFA_OK;
    HIP_TRY(hipHostFree(p));
    return NFA_OK;
}

int nfa_runner_loglike_batch(nfa_runner *r, const int32_t *pix, double *U, double *lnL, int64_t B) {
    if (!r || !U || !lnL) return fail(NFA_ERR_ARG, "null argument");
    if (B <= 0) return NFA_OK;
    int rc = check_pix(r, pix, B); if (rc) return rc;
    if (!r->pr) return fail(NFA_ERR_STATE, "runner has no priors (predict-only)");
    RUNNER_LOCK(r);
    rc = sync_all_lanes(r); if (rc) return rc;           // the staging buffers are shared
    rc = runner_reserve(r, B, false); if (rc) return rc;
    if (B <= NFA_POINT_MAXB) {                           // MultiNest-style single points, a broker's handful: one launch, no copies
        const int served = few_points_kernel(r, pix, U, lnL, B);
        if (served < 0) return NFA_ERR_DEVICE;
        if (served) return NFA_OK;
    }
    if (B == 1 && !pix) {                                // ... or the batch kernels replayed as a graph
        r->n_single += 1;
        if (single_point_graph(r, U, lnL)) return NFA_OK;
    }
    // Large batches go through the stream lanes in chunks: the kernels of chunk c run while the
    // host copies chunk c+1 in, and the results of chunk c come back while c+1 computes.  (Every
    // per-item result is independent of the batch it travels in.)
    // A buffer the device can address itself (nfa_host_alloc, or memory the caller registered with the
    // runtime) is not copied at all: the set-up kernel reads the unit cube over the bus and writes theta back
    // in place, the sum kernel writes lnL there.
    double *vU = (double *)mapped_view(U), *vL = (double *)mapped_view(lnL);
    const int32_t *vP = pix ? (const int32_t *)mapped_view(pix) : nullptr;
    const int n_chunks = (B >= 16384 && r->n_lanes > 1) ? (int)std::min<int64_t>(r->lanes_auto ? 4 : r->n_lanes, B / 4096) : 1;
    const int64_t per = ((B + n_chunks - 1) / n_chunks + 63) / 64 * 64;
    const int ndim = r->ndim;
    for (int c = 0; c < n_chunks; ++c) {
        const int64_t b0 = c * per, nb = std::min<int64_t>(per, B - b0);
        if (nb <= 0) break;
        hipStream_t st = r->lanes[c];
        if (!vU) HIP_TRY(hipMemcpyAsync(r->d_U + b0 * ndim, U + b0 * ndim, sizeof(double) * nb * ndim, hipMemcpyHostToDevice, st));
        if (pix && !vP) HIP_TRY(hipMemcpyAsync(r->d_pix + b0, pix + b0, sizeof(int) * nb, hipMemcpyHostToDevice, st));
        rc = run_batch(r, pix ? (vP ? vP + b0 : r->d_pix + b0) : nullptr, vU ? vU + b0 * ndim : r->d_U + b0 * ndim,
                       vL ? vL + b0 : r->d_lnL + b0, nullptr, nb, true, c, nullptr);
        if (rc) return rc;
    }
    for (int c = 0; c < n_chunks; ++c) {
        const int64_t b0 = c * per, nb = std::min<int64_t>(per, B - b0);
        if (nb <= 0) break;
        hipStream_t st = r->lanes[c];
        if (!vU) HIP_TRY(hipMemcpyAsync(U + b0 * ndim, r->d_U + b0 * ndim, sizeof(double) * nb * ndim, hipMemcpyDeviceToHost, st));
        if (!vL) HIP_TRY(hipMemcpyAsync(lnL + b0, r->d_lnL + b0, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
    }
    for (int c = 0; c < n_chunks; ++c) HIP_TRY(hipStreamSynchronize(r->lanes[c]));
    r->lane_busy &= ~((1u << n_chunks) - 1u);
    return NFA_OK;
}

int nfa_runner_predict_batch(nfa_runner *r, const int32_t *pix, const double *theta, int64_t B,
                             double *spectra_out, double *lnL_out) {
    if (!r || !theta) return fail(NFA_ERR_ARG, "null argument");
    if (B <= 0) return NFA_OK;
    int rc = check_pix(r, pix, B); if (rc) return rc;
    RUNNER_LOCK(r);
    rc = sync_all_lanes(r); if (rc) return rc;
    // output buffers the device can address (nfa_host_alloc) are written by the kernels themselves: the spectra
    // -- B x chan_tot doubles, the bulk of this call's traffic -- then cross the bus once, without a staging copy
    double *vS = spectra_out ? (double *)mapped_view(spectra_out) : nullptr;
    double *vL = lnL_out ? (double *)mapped_view(lnL_out) : nullptr;
    rc = runner_reserve(r, B, spectra_out != nullptr && !vS); if (rc) return rc;
    hipStream_t st = r->lanes[0];
    HIP_TRY(hipMemcpyAsync(r->d_U, theta, sizeof(double) * B * r->ndim, hipMemcpyHostToDevice, st));
    if (pix) HIP_TRY(hipMemcpyAsync(r->d_pix, pix, sizeof(int) * B, hipMemcpyHostToDevice, st));
    rc = run_batch(r, pix ? r->d_pix : nullptr, r->d_U, vL ? vL : r->d_lnL, spectra_out ? (vS ? vS : r->d_spec) : nullptr, B,
                   false, 0, nullptr);
    if (rc) return rc;
    if (spectra_out && !vS)
        HIP_TRY(hipMemcpyAsync(spectra_out, r->d_spec, sizeof(double) * B * r->ss->dev.chan_tot,
                               hipMemcpyDeviceToHost, st));
    if (lnL_out && !vL)
        HIP_TRY(hipMemcpyAsync(lnL_out, r->d_lnL, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    r->lane_busy &= ~1u;
    return NFA_OK;
}

// The spectra-out call for a caller whose buffers live in HBM: theta in, model spectra (and, where asked for, lnL)
// out, nothing copied, nothing waited for -- what deblend_hf_intensity / generate_predicted_profiles
// (nestfit/main.py:1106-1113, 1182-1188) become when the MAP cube and the profile cube stay on the device.
// Consecutive calls rotate over the runner's stream lanes like nfa_runner_loglike_batch_dev's.
int nfa_runner_predict_batch_dev(nfa_runner *r, const int32_t *d_pix, const double *d_theta, int64_t B,
                                 double *d_spectra, double *d_lnL) {
    if (!r || !d_theta || (!d_spectra && !d_lnL)) return fail(NFA_ERR_ARG, "null argument");
    if (B <= 0) return NFA_OK;
    RUNNER_LOCK(r);
    // (coalesced like the likelihood's batches: a launch of one 4096-row batch is one wave per wave slot and as long as its
    // longest wave, 37.8 us against 26 per batch in a launch of eight)
    return enqueue_dev(r, d_pix, const_cast<double *>(d_theta), d_lnL, d_spectra, B, false);
}

void nfa_loglike_callback(double *Cube, int *ndim, int *npars, double *lnew, void *ctx) {
    (void)npars;
    nfa_runner *r = (nfa_runner *)ctx;
    if (!r || !Cube || !lnew || !ndim || *ndim != r->ndim) {
        if (lnew) *lnew = NAN;
        return;
    }
    if (nfa_runner_loglike_batch(r, nullptr, Cube, lnew, 1) != NFA_OK) *lnew = NAN;
}

// ---- device memory + events -------------------------------------------------
int nfa_malloc(void **dptr, int64_t bytes) {
    int rc = engine_init(); if (rc) return rc;
    HIP_TRY(hipMalloc(dptr, (size_t)bytes));
    return NFA_OK;
}
int nfa_free(void *dptr) { HIP_TRY(hipFree(dptr)); return NFA_OK; }
int nfa_memcpy_h2d(void *dst, const void *src, int64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice)); return NFA_OK;
}
int nfa_memcpy_d2h(void *dst, const void *src, int64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost)); return NFA_OK;
}
int nfa_memcpy_d2d(void *dst, const void *src, int64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice)); return NFA_OK;
}
int nfa_event_create(void **ev) {
    int rc = engine_init(); if (rc) return rc;
    hipEvent_t e; HIP_TRY(hipEventCreate(&e)); *ev = (void *)e; return NFA_OK;
}
int nfa_event_destroy(void *ev) { HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return NFA_OK; }
int nfa_event_record(void *ev, nfa_runner *r) {
    if (r) {                 // batches held for coalescing are launched first: the event stands behind everything enqueued so far
        RUNNER_LOCK(r);
        int rc = flush_pending(r); if (rc) return rc;
        HIP_TRY(hipEventRecord((hipEvent_t)ev, r->stream));
        return NFA_OK;
    }
    HIP_TRY(hipEventRecord((hipEvent_t)ev, 0)); return NFA_OK;
}
int nfa_event_synchronize(void *ev) { HIP_TRY(hipEventSynchronize((hipEvent_t)ev)); return NFA_OK; }
int nfa_event_elapsed_ms(void *start, void *stop, float *ms) {
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop)); return NFA_OK;
}

}  // extern "C"

#include "nfa_broker.h"
#include "nfa_ring.h"
#include "nfa_ring_serve.h"
#include "nfa_sampler.h"
#include "nfa_comm.h"
#ifdef NFA_TEST_HOOKS
#include "nfa_testhooks.h"      // libnestfit_amd_test.so only
#endif
