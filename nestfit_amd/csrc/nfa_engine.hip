// nfa_engine.hip -- MI355X (gfx950) NH3 hyperfine log-likelihood engine.
//
// Hand-written HIP for the hot path of autocorr/nestfit v0.2
// (AmmoniaRunner.c_loglikelihood, nestfit/models/ammonia.pyx:423-432):
//   prior transform        core/core.pyx:459-476      -> prior_kernel
//   model spectra          models/ammonia.pyx:326-361,
//                          models/hyperfine.pyx:52-118 -> lnl_kernel
//   chi^2 reduction        core/core.pyx:522-530      -> lnl_kernel (wave shuffle)
//
// Execution model: one 64-lane wavefront owns one (theta, pixel) work item.
// Lanes are frequency channels in the hot loops (coalesced 512-B row loads of
// data/x/t0/tbg), (component, hyperfine line) pairs while line constants are
// formed, and J levels in the partition sums.  Read-only tables (1/(e^x-1)
// interpolation, FastExp product tables or the 2^(i/32) table) are staged once
// per workgroup in LDS; every wave also owns a private LDS slice holding its
// theta and per-line window constants.  No MFMA: the path is elementwise fp64
// plus reductions.
//
// Compile: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off (FMA only where
// written explicitly, so window indices / table indices match the reference's
// plain double arithmetic bit for bit).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nestfit_amd.h"

#define NFA_DATA_QUAL static const
#include "nh3_data.h"

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
//  device constants / shared-table layout
// ---------------------------------------------------------------------------
#define MAXSPEC   16
#define MAXCOMP   10          // ResolvedPlacementPrior's own limit (core.pyx:399)
#define T0_SIZE   1000
// LDS table layout, in doubles
#define SM_T0X    0
#define SM_T0Y    1000
#define SM_EXP2   2000        // 2^(i/32), i = 0..31          (poly mode)
#define SM_FEA    2032        // exp(-(128+j) 2^(l-12)) [10][128] (table mode)
#define SM_FEB    (SM_FEA + 1280)   // exp(-j 2^(l-20)) [10][256]
#define SM_FEC    (SM_FEB + 2560)   // exp(-j 2^(l-28)) [10][256]
#define SM_END_POLY   2032
#define SM_END_TABLE  (SM_FEC + 2560)   // 8432 doubles = 67,456 B

__constant__ int    c_nhf[NFA_N_LEVELS];
__constant__ double c_nu[NFA_N_LEVELS];
__constant__ double c_ea[NFA_N_LEVELS];
__constant__ double c_voff[NFA_N_LEVELS][NFA_MAX_HF_N];
__constant__ double c_tauw[NFA_N_LEVELS][NFA_MAX_HF_N];

struct SpecDev {
    int     n_spec, ncomp, cold, lte;
    int     size[MAXSPEC], trans[MAXSPEC], off[MAXSPEC];
    double  nu_min[MAXSPEC], nu_chan[MAXSPEC];
    int64_t chan_tot;
    const double *xarr, *t0, *tbg, *data, *noise;
    double  t0_xmin, t0_xmax, t0_inv_dx;
};

// ---------------------------------------------------------------------------
//  wave-level helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the
    // compiler from moving LDS accesses across the hand-off point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ double wave_excl_scan(double v, int lane, double *total) {
    double inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    *total = __shfl(inc, 63, 64);
    return inc - v;
}

// ---------------------------------------------------------------------------
//  FastExp replacement (reference: nestfit/core/fastexp.c:234-283, entered with
//  a double narrowed to float, nestfit/core/math.pxd:17)
// ---------------------------------------------------------------------------
// exp(-t) for t = (double)float in [2^-5, 32): n = rint(-t*32/ln2),
// exp(-t) = 2^(n>>5) * 2^((n&31)/32) * exp(r), |r| <= ln2/64.
__device__ __forceinline__ double exp_neg_poly(double t, const double *sm) {
    const double C32 = 46.16624130844682903551758979206054;   // 32/ln2
    const double L_HI = 6.93147180369123816490e-01 / 32.0;    // fdlibm ln2 split
    const double L_LO = 1.90821492927058770002e-10 / 32.0;
    double n = __builtin_rint(-t * C32);
    double r = __builtin_fma(-n, L_HI, -t);
    r = __builtin_fma(-n, L_LO, r);
    int ni = (int)n;
    int m = ni & 31, q = ni >> 5;
    double p = 1.0 / 720.0;
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    double v = sm[SM_EXP2 + m] * p;
    // multiply by 2^q (result stays normal: q >= -47)
    long long bits = __double_as_longlong(v) + ((long long)q << 52);
    return __longlong_as_double(bits);
}

template <int MODE>
__device__ __forceinline__ double nf_fastexp(double xd, const double *sm) {
    const float x = (float)xd;                               // math.pxd:17 narrowing
    const uint32_t bits = __float_as_uint(x);
    const int l = (int)((bits & 0x7f800000u) >> 23) - 122;    // fastexp.c:262
    double r;
    if (MODE == 0) {
        const int lc = min(max(l, 0), 9);
        const int j0 = (bits & 0x007f0000u) >> 16;            // fastexp.c:276-278
        const int j1 = (bits & 0x0000ff00u) >> 8;
        const int j2 = (bits & 0x000000ffu);
        r = sm[SM_FEA + lc * 128 + j0] * sm[SM_FEB + lc * 256 + j1] * sm[SM_FEC + lc * 256 + j2];
    } else {
        // clamp the argument so the polynomial path stays in range for lanes
        // that are overridden below
        const float xc = fminf(fmaxf(x, 0.03125f), 31.999998f);
        r = exp_neg_poly((double)xc, sm);
    }
    if (__ballot(l < 0) != 0ull) {                            // fastexp.c:264-270
        const double t = (double)x;
        double ty = 1.0 - t * (1.0 / 3.0);
        ty = 1.0 - (t * ty) * 0.5;
        ty = 1.0 - (t * ty);
        r = (l < 0) ? ty : r;
    }
    r = (l >= 10) ? 0.0 : r;                                  // fastexp.c:272-273
    r = (x == 0.0f) ? 1.0 : r;                                // fastexp.c:260
    if (x < 0.0f) r = exp(-(double)x);                        // fastexp.c:259
    return r;
}

// 1/(e^x-1): nestfit/models/hyperfine.pyx:23-45
__device__ __forceinline__ double nf_iemtex(double x, const double *sm, double xmin,
                                            double xmax, double inv_dx) {
    const bool in_tab = (xmin < x) && (x < xmax);
    double res;
    {
        long i_lo = in_tab ? (long)((x - xmin) * inv_dx) : 0;
        i_lo = i_lo > T0_SIZE - 2 ? T0_SIZE - 2 : i_lo;       // never taken inside the table
        const double x_lo = sm[SM_T0X + i_lo];
        const double y_lo = sm[SM_T0Y + i_lo];
        const double y_hi = sm[SM_T0Y + i_lo + 1];
        const double slope = (y_hi - y_lo) * inv_dx;
        res = slope * (x - x_lo) + y_lo;
    }
    if (!in_tab) res = 1.0 / expm1(x);
    return res;
}

__device__ __forceinline__ double nf_swift(double tkin) {    // ammonia.pyx:280-286
    return tkin / (1.0 + (tkin / 41.18) * log(1.0 + 0.6 * exp(-15.7 / tkin)));
}

template <int MODE>
__device__ __forceinline__ double nf_partition_level(int j, double trot, const double *sm) {
    // ammonia.pyx:289-295
    const double dj = (double)j;
    const double arg = NFA_H * (NFA_BROT * dj * (double)(j + 1) + (NFA_CROT - NFA_BROT) * dj * dj)
                       / (NFA_KB * trot);
    return (double)(2 * j + 1) * nf_fastexp<MODE>(arg, sm);
}


// Line centre, width and channel window of hyperfine line i of transition t
// (reference: nestfit/models/hyperfine.pyx:70-91).  Plain double arithmetic,
// no contraction: the floor() arguments must round like the reference's.
struct LineConst { double nucen, idenom; int lo, hi; };
__device__ __forceinline__ LineConst nf_line(int t, int i, double voff, double sigm, double nu0,
                                             double nu_min, double nu_chan, int N) {
    LineConst r;
    const double hf_freq   = (1.0 - c_voff[t][i] / NFA_CKMS) * nu0;
    const double hf_width  = sigm / NFA_CKMS * hf_freq;
    const double hf_offset = voff / NFA_CKMS * hf_freq;
    const double hf_nucen  = hf_freq - hf_offset;
    const double hf_idenom = 0.5 / (hf_width * hf_width);
    const double nu_cutoff = sqrt(12.5 / hf_idenom);
    const double nu_lo = (hf_nucen - nu_min - nu_cutoff);
    const double nu_hi = (hf_nucen - nu_min + nu_cutoff);
    long lo = (long)floor(nu_lo / nu_chan);
    long hi = (long)floor(nu_hi / nu_chan);
    if (hi < 0 || lo > N - 1) { lo = 0; hi = 0; }             // `continue`: empty window
    else {
        lo = lo < 0 ? 0 : lo;
        hi = hi > N - 1 ? N - 1 : hi;
    }
    r.nucen = hf_nucen; r.idenom = hf_idenom; r.lo = (int)lo; r.hi = (int)hi;
    return r;
}

__device__ __forceinline__ void load_shared_tables(double *sm, const double *g_tabs, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = g_tabs[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------
//  lnl_kernel: model spectra + chi^2 for B (theta, pixel) items
// ---------------------------------------------------------------------------
template <int MODE, bool WRITE_SPEC>
__global__ void __launch_bounds__(MODE == 0 ? 512 : 256) lnl_kernel(SpecDev S, const int *__restrict__ pix,
                           const double *__restrict__ theta, double *__restrict__ lnL,
                           double *__restrict__ spec_out, long B, int nhf_max,
                           int wave_doubles, const double *__restrict__ g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n_shared = (MODE == 0) ? SM_END_TABLE : SM_END_POLY;
    load_shared_tables(sm, g_tabs, n_shared);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int ncomp = S.ncomp, ndim = NFA_N_PARAMS * ncomp;
    const int P = ncomp * nhf_max;
    double *w_theta = sm + n_shared + (size_t)wave * wave_doubles;
    double *w_trot  = w_theta + ndim;
    double *w_tex   = w_trot + ncomp;
    double *w_qpara = w_tex + ncomp;
    double *w_qorth = w_qpara + ncomp;
    double *w_tmain = w_qorth + ncomp;
    double *w_zlev  = w_tmain + ncomp;            // [ncomp][9]
    double *w_nucen = w_zlev + ncomp * NFA_N_LEVELS;
    double *w_idenom = w_nucen + P;
    double *w_htau  = w_idenom + P;
    int2   *w_lohi  = (int2 *)(w_htau + P);

    const long gw = (long)blockIdx.x * waves + wave, nw = (long)gridDim.x * waves;
    for (long b = gw; b < B; b += nw) {
        const long p_ix = pix ? (long)pix[b] : 0;
        if (lane < ndim) w_theta[lane] = theta[b * ndim + lane];
        wave_lds_sync();
        // --- per-component temperatures (ammonia.pyx:337-346)
        if (lane < ncomp) {
            double trot = w_theta[ncomp + lane];
            double tex  = w_theta[2 * ncomp + lane];
            if (S.cold) trot = nf_swift(trot);
            if (S.lte) tex = trot;
            w_trot[lane] = trot;
            w_tex[lane] = tex;
        }
        wave_lds_sync();
        // --- partition sums, lanes = J levels (ammonia.pyx:304-315, 347-348)
        for (int c = 0; c < ncomp; ++c) {
            const double trot = w_trot[c];
            const int j = lane;
            double lev = 0.0;
            if (j < NFA_NPART) lev = nf_partition_level<MODE>(j, trot, sm);
            const bool is_orth = (j % 3) == 0;
            const double qp = wave_sum((j < NFA_NPART && !is_orth) ? lev : 0.0);
            const double qo = wave_sum((j < NFA_NPART && is_orth) ? 2 * lev : 0.0);
            if (lane == 0) { w_qpara[c] = qp; w_qorth[c] = qo; }
            if (j >= 1 && j <= NFA_N_LEVELS) w_zlev[c * NFA_N_LEVELS + (j - 1)] = lev;
        }
        wave_lds_sync();

        double lnl_tot = 0.0;
        for (int s = 0; s < S.n_spec; ++s) {
            const int t = S.trans[s] - 1, N = S.size[s], off = S.off[s];
            const int nhf = c_nhf[t];
            const double nu0 = c_nu[t];
            const bool para = ((t + 1) % 3) != 0;
            // --- main-line optical depth, lanes = components (ammonia.pyx:349-361)
            if (lane < ncomp) {
                const int c = lane;
                const double tex  = w_tex[c];
                const double ntot = w_theta[3 * ncomp + c];
                const double sigm = w_theta[4 * ncomp + c];
                const double orth = w_theta[5 * ncomp + c];
                const double zlev = w_zlev[c * NFA_N_LEVELS + t];
                const double qtot = para ? w_qpara[c] : w_qorth[c];
                const double species_frac = para ? 1.0 - orth : orth;
                const double pop_rotstate = pow(10.0, ntot) * species_frac * zlev / qtot;
                const double ex = exp(-NFA_H * nu0 / (NFA_KB * tex));
                const double expterm = (1.0 - ex) / (1.0 + ex);
                const double fracterm = (NFA_CCMS * NFA_CCMS) * c_ea[t] / (8 * M_PI * (nu0 * nu0));
                const double widthterm = NFA_CKMS / (sigm * nu0 * sqrt(2 * M_PI));
                const double tau_main = pop_rotstate * fracterm * expterm * widthterm;
                // log10 -> 10** round trip (ammonia.pyx:361, hyperfine.pyx:63)
                w_tmain[c] = pow(10.0, log10(tau_main));
            }
            wave_lds_sync();
            // --- line constants + windows, lanes = (component, line) pairs
            //     (hyperfine.pyx:68-91)
            for (int p = lane; p < ncomp * nhf; p += 64) {
                const int c = p / nhf, i = p - c * nhf;
                const LineConst lc = nf_line(t, i, w_theta[c], w_theta[4 * ncomp + c], nu0,
                                             S.nu_min[s], S.nu_chan[s], N);
                const double hf_nucen = lc.nucen, hf_idenom = lc.idenom;
                const double hf_tau = w_tmain[c] * c_tauw[t][i];
                int2 lh; lh.x = lc.lo; lh.y = lc.hi;
                const int q = c * nhf_max + i;
                w_nucen[q] = hf_nucen;
                w_idenom[q] = hf_idenom;
                w_htau[q] = hf_tau;
                w_lohi[q] = lh;
            }
            wave_lds_sync();
            // --- rows of 64 channels: tau profile, Tb, chi^2
            //     (hyperfine.pyx:93-113, core.pyx:522-530)
            const double *xs = S.xarr + off, *t0s = S.t0 + off, *tbgs = S.tbg + off;
            const double *ds = S.data + p_ix * S.chan_tot + off;
            double acc = 0.0;
            for (int r0 = 0; r0 < N; r0 += 64) {
                const int j = r0 + lane;
                const bool valid = j < N;
                const int jj = valid ? j : N - 1;
                const double xj = xs[jj];
                const double dj = ds[jj];
                double pred = 0.0;
                for (int c = 0; c < ncomp; ++c) {
                    int2 lh = make_int2(0, 0);
                    if (lane < nhf) lh = w_lohi[c * nhf_max + lane];
                    unsigned long long mask = __ballot(lh.y > lh.x && lh.x < r0 + 64 && lh.y > r0);
                    if (mask == 0ull) continue;
                    double tau = 0.0;
                    while (mask) {
                        const int i = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const int q = c * nhf_max + i;
                        const double nucen = w_nucen[q], idenom = w_idenom[q], htau = w_htau[q];
                        const int2 w = w_lohi[q];
                        const double nu = xj - nucen;
                        const double tau_exp = nu * nu * idenom;
                        const double e = nf_fastexp<MODE>(tau_exp, sm);
                        if (j >= w.x && j < w.y) tau = __builtin_fma(htau, e, tau);
                    }
                    const bool live = valid && !(tau == 0.0);         // hyperfine.pyx:104-105
                    if (__ballot(live) != 0ull) {
                        const double T0 = t0s[jj];
                        const double tbg = tbgs[jj];
                        const double y = nf_iemtex(T0 / w_tex[c], sm, S.t0_xmin, S.t0_xmax, S.t0_inv_dx);
                        const double tb = (T0 * (y - tbg)) * (1.0 - nf_fastexp<MODE>(tau, sm));
                        if (live) pred += tb;
                    }
                }
                if (WRITE_SPEC) { if (valid) spec_out[b * S.chan_tot + off + j] = pred; }
                const double dev = dj - pred;
                if (valid) acc = __builtin_fma(dev, dev, acc);
            }
            acc = wave_sum(acc);
            const double noise = S.noise[p_ix * S.n_spec + s];
            lnl_tot += -acc / (2 * (noise * noise));                   // core.pyx:530
            wave_lds_sync();
        }
        if (lane == 0 && lnL) lnL[b] = lnl_tot;
    }
}

// ---------------------------------------------------------------------------
//  prior_kernel: PriorTransformer.c_transform (core.pyx:459-476), one wave per
//  unit-cube row, in place.
// ---------------------------------------------------------------------------
struct DistDev {
    int     size, pad;
    double  du, dx, xmin, xmax;
    const double *xax, *pdf, *cdf, *ppf;
};
#define MAXPRIOR 16
#define MAXDIST  16
struct PriorProg {
    int n_prior, n_dist, n_param, max_size;
    nfa_prior_desc pr[MAXPRIOR];
    DistDev        ds[MAXDIST];
};

__device__ __forceinline__ double d_ppf_interp(const DistDev &d, double u) {   // core.pyx:47-63
    long i_lo = (long)((double)(d.size - 1) * u);
    long i_hi = i_lo + 1;
    i_lo = i_lo < 0 ? 0 : (i_lo > d.size - 1 ? d.size - 1 : i_lo);   // u==1 reads past the end
    i_hi = i_hi > d.size - 1 ? d.size - 1 : (i_hi < 0 ? 0 : i_hi);   // in the reference; clamp
    const double x_lo = (double)i_lo * d.du;
    const double y_lo = d.ppf[i_lo];
    const double y_hi = d.ppf[i_hi];
    const double slope = (y_hi - y_lo) / d.du;
    return slope * (u - x_lo) + y_lo;
}

// `prior.interp(utheta, n)` of the simple kinds, lanes = components
__device__ __forceinline__ void d_simple_interp(const PriorProg &pp, int kind, int dist, int p_ix,
                                                double value, double *u, int n, int lane) {
    const int ix = p_ix * n;
    if (kind == NFA_PRIOR_CONSTANT) {                         // core.pyx:233-238
        if (lane < n) u[ix + lane] = value;
    } else if (kind == NFA_PRIOR_ORDERED) {                   // core.pyx:242-258
        if (lane == 0) {
            double umin = 0.0;
            for (int i = 0; i < n; ++i) {
                const double uu = umin + (1 - umin) * u[ix + i];
                umin = uu;
                u[ix + i] = d_ppf_interp(pp.ds[dist], uu);
            }
        }
    } else {                                                  // core.pyx:192-197
        if (lane < n) u[ix + lane] = d_ppf_interp(pp.ds[dist], u[ix + lane]);
    }
    wave_lds_sync();
}

// Distribution.cdf_over_interval + cdf_interp (core.pyx:65-161) on a private
// LDS copy of the CDF; lanes own contiguous chunks of the table.
__device__ double d_placement_draw(const DistDev &d, double *cdf, double x_lo, double x_hi,
                                   double sfact, double u, int lane) {
    if (x_lo > x_hi) { const double t = x_lo; x_lo = x_hi; x_hi = t; }
    const int size = d.size;
    long i_lo = (long)((x_lo - d.xmin) / d.dx);
    if (i_lo >= size) i_lo = size - 1; else if (i_lo < 0) i_lo = 0;
    long i_hi = (long)((x_hi - d.xmin) / d.dx);
    if (i_hi == i_lo) i_hi = i_lo + 1;
    if (i_hi > size) i_hi = size; else if (i_hi < 0) i_hi = 1;
    const int ilo = (int)i_lo, ihi = (int)i_hi;
    // trapezoid terms i = ilo+1 .. ihi-1, chunked over lanes
    const int L = ihi - ilo - 1;
    const int ch = (L + 63) / 64;
    const int k0 = ilo + 1 + lane * ch;
    const int k1 = min(k0 + ch, ihi);
    const double inv_delta_i = 1.0 / (double)(ihi - ilo);
    double local = 0.0;
    for (int i = k0; i < k1; ++i) {
        double scale;
        const double base = 1.0 - (double)(i - ilo) * inv_delta_i;
        if (sfact == 0.0) scale = 1.0;
        else if (sfact == 1.0) scale = base;
        else if (sfact == 2.0) scale = base * base;
        else scale = pow(base, sfact);
        local += 0.5 * (d.pdf[i] + d.pdf[i - 1]) * scale;
    }
    double csum;
    double run = wave_excl_scan(local, lane, &csum);
    // materialise the rewritten, normalised CDF
    for (int i = lane; i < ilo; i += 64) cdf[i] = 0.0;
    for (int i = ihi + lane; i < size; i += 64) cdf[i] = 1.0;
    if (L <= 0) {
        if (lane == 0) cdf[ilo] = 1.0 / csum;                 // csum == 0: inf like the reference
    } else {
        if (lane == 0) cdf[ilo] = 0.0 / csum;
        for (int i = k0; i < k1; ++i) {
            double scale;
            const double base = 1.0 - (double)(i - ilo) * inv_delta_i;
            if (sfact == 0.0) scale = 1.0;
            else if (sfact == 1.0) scale = base;
            else if (sfact == 2.0) scale = base * base;
            else scale = pow(base, sfact);
            run += 0.5 * (d.pdf[i] + d.pdf[i - 1]) * scale;
            cdf[i] = run / csum;
        }
    }
    wave_lds_sync();
    // cdf_interp (core.pyx:65-107)
    if (u <= cdf[0]) u = 1e-64;
    int i;
    const bool regular = (L > 0) && (csum > 0.0) && (csum < INFINITY);
    if (regular) {
        // monotone table: the bisection lands on (#entries below u) - 1
        int cnt = 0;
        for (int k = lane; k < size; k += 64) cnt += (u > cdf[k]) ? 1 : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
        i = cnt >= 1 ? cnt - 1 : 0;
    } else {
        int lo = 0, hi = size;
        i = hi / 2;
        while (i != lo) {
            if (u > cdf[i]) lo = i; else hi = i;
            i = (hi + lo) / 2;
        }
    }
    int j_lo = i < size ? i : size - 1;
    int j_hi = j_lo + 1;
    if (j_hi > size - 1) j_hi = size - 1;                     // reference reads cdf[size] here
    const double xl = d.xax[j_lo];
    const double y_lo = cdf[j_lo];
    const double y_hi = cdf[j_hi];
    const double slope = (y_hi - y_lo) / d.dx;
    const double res = 1 / slope * (u - y_lo) + xl;
    wave_lds_sync();
    return res;
}

__global__ void __launch_bounds__(256) prior_kernel(PriorProg pp, double *__restrict__ U, long B, int n, int wave_doubles) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int ndim = pp.n_param * n;
    double *u = sm + (size_t)wave * wave_doubles;       // [ndim]
    double *cdf = u + ((ndim + 1) & ~1);                 // [max_size]
    const long gw = (long)blockIdx.x * waves + wave, nw = (long)gridDim.x * waves;
    for (long b = gw; b < B; b += nw) {
        for (int k = lane; k < ndim; k += 64) u[k] = U[b * ndim + k];
        wave_lds_sync();
        for (int k = 0; k < pp.n_prior; ++k) {
            const nfa_prior_desc &p = pp.pr[k];
            const int ix = p.p_ix * n;
            switch (p.kind) {
            case NFA_PRIOR_SIMPLE:
            case NFA_PRIOR_CONSTANT:
            case NFA_PRIOR_ORDERED:
                d_simple_interp(pp, p.kind, p.dist0, p.p_ix, p.value, u, n, lane);
                break;
            case NFA_PRIOR_DUPLICATE:                         // core.pyx:211-221
                if (lane < n) {
                    const double v = d_ppf_interp(pp.ds[p.dist0], u[ix + lane]);
                    u[ix + lane] = v;
                    u[p.p_ix2 * n + lane] = v;
                }
                wave_lds_sync();
                break;
            case NFA_PRIOR_SPACED:                            // core.pyx:280-292
                if (lane == 0) {
                    double v = d_ppf_interp(pp.ds[p.dist0], u[ix]);
                    u[ix] = v;
                    for (int i = 1; i < n; ++i) {
                        v = v + d_ppf_interp(pp.ds[p.dist1], u[ix + i]);
                        u[ix + i] = v;
                    }
                }
                wave_lds_sync();
                break;
            case NFA_PRIOR_CENSEP:                            // core.pyx:305-318
                if (lane == 0) {
                    const double vcen = d_ppf_interp(pp.ds[p.dist0], u[ix]);
                    if (n == 1) u[ix] = vcen;
                    else if (n == 2) {
                        const double vsep = d_ppf_interp(pp.ds[p.dist1], u[ix + 1]);
                        u[ix]     = vcen - 0.5 * vsep;
                        u[ix + 1] = vcen + 0.5 * vsep;
                    }
                }
                wave_lds_sync();
                break;
            case NFA_PRIOR_RESOLVED_CENSEP: {                 // core.pyx:347-366
                const int ix_s = p.p_ix2 * n;
                d_simple_interp(pp, p.sub_kind, p.dist2, p.p_ix2, p.value, u, n, lane);
                if (lane == 0) {
                    const double vcen = d_ppf_interp(pp.ds[p.dist0], u[ix]);
                    if (n == 1) u[ix] = vcen;
                    else if (n == 2) {
                        double vsep = d_ppf_interp(pp.ds[p.dist1], u[ix + 1]);
                        const double min_sep = p.sep_scale * sqrt(u[ix_s] * u[ix_s + 1]);
                        if (min_sep > vsep) vsep = min_sep;
                        u[ix]     = vcen - 0.5 * vsep;
                        u[ix + 1] = vcen + 0.5 * vsep;
                    }
                }
                wave_lds_sync();
            } break;
            case NFA_PRIOR_RESOLVED_PLACEMENT: {              // core.pyx:391-435
                if (n > MAXCOMP) break;
                const DistDev &vd = pp.ds[p.dist0];
                const int ix_s = p.p_ix2 * n;
                double v_lo = vd.xmin, v_hi = vd.xmax;
                d_simple_interp(pp, p.sub_kind, p.dist2, p.p_ix2, p.value, u, n, lane);
                if (n == 1) {
                    if (lane == 0) u[ix] = d_ppf_interp(vd, u[ix]);
                    wave_lds_sync();
                    break;
                }
                // every lane carries the same scalars
                double min_seps[MAXCOMP];
                double sep_tot = 0.0;
                min_seps[0] = 0.0;
#pragma unroll
                for (int i = 1; i < MAXCOMP; ++i) {
                    double sep = 0.0;
                    if (i < n) {
                        sep = p.sep_scale * sqrt(u[ix_s + i] * u[ix_s + i - 1]);
                        sep_tot += sep;
                    }
                    min_seps[i] = sep;
                }
                if (sep_tot > v_hi - v_lo) {
                    const double overf = (v_hi - v_lo) / sep_tot;
                    sep_tot = 0.0;
#pragma unroll
                    for (int i = 0; i < MAXCOMP; ++i) {
                        if (i < n) { min_seps[i] *= overf; sep_tot += min_seps[i]; }
                    }
                }
                v_hi -= sep_tot;
#pragma unroll
                for (int i = 0; i < MAXCOMP; ++i) {
                    if (i < n) {
                        const double sep = min_seps[i];
                        v_lo += sep;
                        v_hi += sep;
                        const double uu = u[ix + i];
                        v_lo = d_placement_draw(vd, cdf, v_lo, v_hi, (double)(n - 1 - i), uu, lane);
                        if (lane == 0) u[ix + i] = v_lo;
                    }
                }
                wave_lds_sync();
            } break;
            default: break;
            }
        }
        for (int k = lane; k < ndim; k += 64) U[b * ndim + k] = u[k];
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------
//  set-up kernels
// ---------------------------------------------------------------------------
__global__ void prep_kernel(const double *__restrict__ x, double *__restrict__ t0,
                            double *__restrict__ tbg, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double T0 = NFA_H * x[i] / NFA_KB;                  // hyperfine.pyx:106
    t0[i] = T0;
    tbg[i] = 1.0 / expm1(T0 / NFA_TCMB);                      // ammonia.pyx:274-277
}

// null_lnZ[pix][spec] = -sum(data^2)/(2 noise^2): Spectrum.c_loglikelihood with
// pred == 0 (core.pyx:517-530).  One wave per (pixel, spectrum).
__global__ void null_lnz_kernel(SpecDev S, long n_pix, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n_pix * S.n_spec) return;
    const long p = w / S.n_spec;
    const int s = (int)(w - p * S.n_spec);
    const double *d = S.data + p * S.chan_tot + S.off[s];
    double acc = 0.0;
    for (int j = lane; j < S.size[s]; j += 64) { const double dev = d[j] - 0.0; acc += dev * dev; }
    acc = wave_sum(acc);
    const double noise = S.noise[p * S.n_spec + s];
    if (lane == 0) out[w] = -acc / (2 * (noise * noise));
}

// unit-test kernels
template <int MODE>
__global__ void test_fastexp_kernel(const double *x, double *out, long n, const double *g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    load_shared_tables(sm, g_tabs, MODE == 0 ? SM_END_TABLE : SM_END_POLY);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ((n + 63) & ~63L);
         i += (long)gridDim.x * blockDim.x) {
        const double v = nf_fastexp<MODE>(i < n ? x[i] : 1.0, sm);
        if (i < n) out[i] = v;
    }
}

__global__ void test_iemtex_kernel(const double *x, double *out, long n, const double *g_tabs,
                                   double xmin, double xmax, double inv_dx) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    load_shared_tables(sm, g_tabs, SM_END_POLY);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        out[i] = nf_iemtex(x[i], sm, xmin, xmax, inv_dx);
}

template <int MODE>
__global__ void test_partition_kernel(const double *trot, double *qpara, double *qorth, long n,
                                      const double *g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    load_shared_tables(sm, g_tabs, MODE == 0 ? SM_END_TABLE : SM_END_POLY);
    const int lane = threadIdx.x & 63;
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    double lev = 0.0;
    if (lane < NFA_NPART) lev = nf_partition_level<MODE>(lane, trot[w], sm);
    const bool is_orth = (lane % 3) == 0;
    const double qp = wave_sum((lane < NFA_NPART && !is_orth) ? lev : 0.0);
    const double qo = wave_sum((lane < NFA_NPART && is_orth) ? 2 * lev : 0.0);
    if (lane == 0) { qpara[w] = qp; qorth[w] = qo; }
}


__global__ void test_windows_kernel(SpecDev S, int s, double voff, double sigm, int *lo, int *hi) {
    const int t = S.trans[s] - 1, i = threadIdx.x;
    if (i >= c_nhf[t]) return;
    const LineConst lc = nf_line(t, i, voff, sigm, c_nu[t], S.nu_min[s], S.nu_chan[s], S.size[s]);
    lo[i] = lc.lo; hi[i] = lc.hi;
}

// ---------------------------------------------------------------------------
//  host side
// ---------------------------------------------------------------------------
struct Engine {
    bool   init = false;
    int    device = 0;
    int    n_cu = 256;
    int    exp_mode = 0;
    bool   have_t0 = false;
    double *d_tabs = nullptr;                      // SM_END_TABLE doubles
    double t0_xmin = 0, t0_xmax = 0, t0_inv_dx = 0;
};
static Engine g_eng;

static int engine_init() {
    if (g_eng.init) return NFA_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(NFA_ERR_DEVICE, "no HIP device available (the engine has no CPU fallback)");
    HIP_TRY(hipSetDevice(g_eng.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_eng.device));
    g_eng.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_nhf), nfa_nhf, sizeof(nfa_nhf)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_nu), nfa_nu, sizeof(nfa_nu)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_ea), nfa_ea, sizeof(nfa_ea)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_voff), nfa_voff, sizeof(nfa_voff)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_tauw), nfa_tau_wts, sizeof(nfa_tau_wts)));
    std::vector<double> tabs(SM_END_TABLE, 0.0);
    // 2^(i/32)
    for (int i = 0; i < 32; ++i) tabs[SM_EXP2 + i] = (double)exp2l((long double)i / 32.0L);
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
};

struct nfa_priors {
    PriorProg prog{};
    std::vector<double *> d_arrays;
};

struct nfa_runner {
    nfa_specset *ss = nullptr;
    nfa_priors  *pr = nullptr;
    int ncomp = 1, cold = 0, lte = 0, ndim = 6;
    hipStream_t stream = nullptr;
    double *d_U = nullptr, *d_lnL = nullptr, *d_spec = nullptr;
    int    *d_pix = nullptr;
    int64_t cap_B = 0, cap_spec = 0;
    // optional per-kernel timing (HIP events on the runner's stream)
    bool profiling = false;
    std::vector<hipEvent_t> ev;      // triples: before priors, before lnl, after lnl
    size_t ev_used = 0;
};

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

int nfa_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return NFA_OK; }

int nfa_device_name(char *buf, int buflen) {
    int rc = engine_init(); if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_eng.device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return NFA_OK;
}

int nfa_set_exp_mode(int mode) {
    if (mode != 0 && mode != 1) return fail(NFA_ERR_ARG, "exp mode must be 0 (table) or 1 (poly)");
    g_eng.exp_mode = mode;
    return NFA_OK;
}
int nfa_get_exp_mode(void) { return g_eng.exp_mode; }

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
    if (!out || !sizes || !trans_ids || !xarr || !data || !noise)
        return fail(NFA_ERR_ARG, "null argument");
    if (n_spec < 1 || n_spec > MAXSPEC) return fail(NFA_ERR_ARG, "n_spec must be in 1..16");
    if (n_pix < 1) return fail(NFA_ERR_ARG, "n_pix must be >= 1");
    int rc = engine_init(); if (rc) return rc;
    nfa_specset *ss = new nfa_specset();
    SpecDev &d = ss->dev;
    d.n_spec = n_spec;
    int64_t tot = 0;
    for (int s = 0; s < n_spec; ++s) {
        if (sizes[s] < 2 || sizes[s] > (1 << 24)) { delete ss; return fail(NFA_ERR_ARG, "spectrum size out of range"); }
        if (trans_ids[s] < 1 || trans_ids[s] > NFA_N_LEVELS) {          // ammonia.pyx:268
            delete ss; return fail(NFA_ERR_ARG, "trans_id must be in 1..9");
        }
        const double nu_chan = xarr[s][1] - xarr[s][0];
        if (!(nu_chan > 0)) {                                           // core.pyx:503-504
            delete ss; return fail(NFA_ERR_ARG, "frequency axis must be ascending");
        }
        d.size[s] = (int)sizes[s];
        d.trans[s] = trans_ids[s];
        d.off[s] = (int)tot;
        d.nu_min[s] = xarr[s][0];
        d.nu_chan[s] = nu_chan;
        ss->nhf_max = std::max(ss->nhf_max, nfa_nhf[trans_ids[s] - 1]);
        tot += sizes[s];
    }
    for (int64_t i = 0; i < n_pix * n_spec; ++i)
        if (!(noise[i] > 0)) { delete ss; return fail(NFA_ERR_ARG, "noise must be > 0"); }   // core.pyx:502
    d.chan_tot = tot;
    ss->n_pix = n_pix;
    std::vector<double> xcat(tot);
    for (int s = 0; s < n_spec; ++s) memcpy(xcat.data() + d.off[s], xarr[s], sizeof(double) * sizes[s]);
    HIP_TRY(hipMalloc(&ss->d_xarr, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_t0, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_tbg, sizeof(double) * tot));
    HIP_TRY(hipMalloc(&ss->d_data, sizeof(double) * tot * n_pix));
    HIP_TRY(hipMalloc(&ss->d_noise, sizeof(double) * n_spec * n_pix));
    HIP_TRY(hipMemcpy(ss->d_xarr, xcat.data(), sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ss->d_data, data, sizeof(double) * tot * n_pix, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ss->d_noise, noise, sizeof(double) * n_spec * n_pix, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0,
                       ss->d_xarr, ss->d_t0, ss->d_tbg, (long)tot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    d.xarr = ss->d_xarr; d.t0 = ss->d_t0; d.tbg = ss->d_tbg; d.data = ss->d_data; d.noise = ss->d_noise;
    *out = ss;
    return NFA_OK;
}

int nfa_specset_destroy(nfa_specset *ss) {
    if (!ss) return NFA_OK;
    (void)hipFree(ss->d_xarr); (void)hipFree(ss->d_t0); (void)hipFree(ss->d_tbg); (void)hipFree(ss->d_data); (void)hipFree(ss->d_noise);
    delete ss;
    return NFA_OK;
}

int nfa_specset_set_data(nfa_specset *ss, int64_t pix, const double *data) {
    if (!ss || !data || pix < 0 || pix >= ss->n_pix) return fail(NFA_ERR_ARG, "bad pixel index");
    HIP_TRY(hipMemcpy(ss->d_data + pix * ss->dev.chan_tot, data, sizeof(double) * ss->dev.chan_tot,
                      hipMemcpyHostToDevice));
    return NFA_OK;
}

int nfa_specset_null_lnz(const nfa_specset *ss, double *out) {
    if (!ss || !out) return fail(NFA_ERR_ARG, "null argument");
    const int64_t n = ss->n_pix * ss->dev.n_spec;
    double *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, sizeof(double) * n));
    hipLaunchKernelGGL(null_lnz_kernel, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, 0,
                       ss->dev, (long)ss->n_pix, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(d_out);
    return NFA_OK;
}

int nfa_specset_tbg(const nfa_specset *ss, double *out) {
    if (!ss || !out) return fail(NFA_ERR_ARG, "null argument");
    HIP_TRY(hipMemcpy(out, ss->d_tbg, sizeof(double) * ss->dev.chan_tot, hipMemcpyDeviceToHost));
    return NFA_OK;
}

int64_t nfa_specset_chan_tot(const nfa_specset *ss) { return ss ? ss->dev.chan_tot : 0; }

// ---- priors ----------------------------------------------------------------
int nfa_priors_create(nfa_priors **out, const nfa_prior_desc *priors, int n_prior,
                      const nfa_dist_desc *dists, int n_dist, int n_param) {
    if (!out || !priors || n_prior < 1 || n_prior > MAXPRIOR || n_dist < 0 || n_dist > MAXDIST)
        return fail(NFA_ERR_ARG, "prior program out of range (<=16 priors, <=16 distributions)");
    int rc = engine_init(); if (rc) return rc;
    nfa_priors *p = new nfa_priors();
    PriorProg &g = p->prog;
    g.n_prior = n_prior; g.n_dist = n_dist; g.n_param = n_param; g.max_size = 2;
    for (int k = 0; k < n_prior; ++k) {
        g.pr[k] = priors[k];
        const int dd[3] = {priors[k].dist0, priors[k].dist1, priors[k].dist2};
        for (int q = 0; q < 3; ++q)
            if (dd[q] >= n_dist) { delete p; return fail(NFA_ERR_ARG, "distribution index out of range"); }
        if (priors[k].p_ix < 0) { delete p; return fail(NFA_ERR_ARG, "p_ix must be >= 0"); }   // core.pyx:186
    }
    for (int k = 0; k < n_dist; ++k) {
        const nfa_dist_desc &s = dists[k];
        if (s.size < 2 || s.size > 65536) { delete p; return fail(NFA_ERR_ARG, "distribution size out of range"); }
        DistDev &d = g.ds[k];
        d.size = (int)s.size; d.du = s.du; d.dx = s.dx; d.xmin = s.xmin; d.xmax = s.xmax;
        const double *src[4] = {s.xax, s.pdf, s.cdf, s.ppf};
        const double **dst[4] = {&d.xax, &d.pdf, &d.cdf, &d.ppf};
        for (int q = 0; q < 4; ++q) {
            double *dp = nullptr;
            HIP_TRY(hipMalloc(&dp, sizeof(double) * s.size));
            HIP_TRY(hipMemcpy(dp, src[q], sizeof(double) * s.size, hipMemcpyHostToDevice));
            p->d_arrays.push_back(dp);
            *dst[q] = dp;
        }
        g.max_size = std::max(g.max_size, (int)s.size);
    }
    *out = p;
    return NFA_OK;
}

int nfa_priors_destroy(nfa_priors *p) {
    if (!p) return NFA_OK;
    for (double *d : p->d_arrays) (void)hipFree(d);
    delete p;
    return NFA_OK;
}

static int launch_priors(const nfa_priors *p, double *d_U, int64_t B, int ncomp, hipStream_t st) {
    const int ndim = p->prog.n_param * ncomp;
    const int wave_doubles = ((ndim + 1) & ~1) + ((p->prog.max_size + 1) & ~1);
    const int waves = 4;
    const size_t lds = sizeof(double) * wave_doubles * waves;
    if (lds > 160 * 1024) return fail(NFA_ERR_ARG, "distribution tables too large for LDS scratch");
    int64_t blocks = (B + waves - 1) / waves;
    const int64_t cap = (int64_t)g_eng.n_cu * 8;
    if (blocks > cap) blocks = cap;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)prior_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(prior_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, st, p->prog, d_U,
                       (long)B, ncomp, wave_doubles);
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
    if (priors && priors->prog.n_param != NFA_N_PARAMS)
        return fail(NFA_ERR_ARG, "prior program must cover the 6 ammonia parameters");
    int rc = engine_init(); if (rc) return rc;
    nfa_runner *r = new nfa_runner();
    r->ss = ss; r->pr = priors; r->ncomp = ncomp; r->cold = cold ? 1 : 0; r->lte = lte ? 1 : 0;
    r->ndim = NFA_N_PARAMS * ncomp;
    HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    *out = r;
    return NFA_OK;
}

int nfa_runner_destroy(nfa_runner *r) {
    if (!r) return NFA_OK;
    (void)hipStreamSynchronize(r->stream);
    (void)hipFree(r->d_U); (void)hipFree(r->d_lnL); (void)hipFree(r->d_pix); (void)hipFree(r->d_spec);
    for (hipEvent_t x : r->ev) (void)hipEventDestroy(x);
    (void)hipStreamDestroy(r->stream);
    delete r;
    return NFA_OK;
}

int nfa_runner_ndim(const nfa_runner *r) { return r ? r->ndim : 0; }

static int runner_reserve(nfa_runner *r, int64_t B, bool spec) {
    if (B > r->cap_B) {
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

template <int MODE, bool WS>
static int launch_lnl_t(nfa_runner *r, const int *d_pix, const double *d_theta, double *d_lnL,
                        double *d_spec, int64_t B) {
    SpecDev S = r->ss->dev;
    S.ncomp = r->ncomp; S.cold = r->cold; S.lte = r->lte;
    S.t0_xmin = g_eng.t0_xmin; S.t0_xmax = g_eng.t0_xmax; S.t0_inv_dx = g_eng.t0_inv_dx;
    const int nhf_max = r->ss->nhf_max;
    const int P = r->ncomp * nhf_max;
    int wave_doubles = r->ndim + r->ncomp * 5 + r->ncomp * NFA_N_LEVELS + P * 4;
    wave_doubles = (wave_doubles + 1) & ~1;
    const int n_shared = MODE == 0 ? SM_END_TABLE : SM_END_POLY;
    // table mode shares 67 KB of product tables: use fat workgroups
    int waves = MODE == 0 ? 8 : 4;     // must match lnl_kernel's __launch_bounds__
    size_t lds = sizeof(double) * ((size_t)n_shared + (size_t)wave_doubles * waves);
    while (lds > 160 * 1024 && waves > 1) { waves >>= 1; lds = sizeof(double) * ((size_t)n_shared + (size_t)wave_doubles * waves); }
    if (lds > 160 * 1024) return fail(NFA_ERR_ARG, "ncomp too large for the LDS line table");
    int64_t blocks = (B + waves - 1) / waves;
    const int per_cu = std::max<int>(1, std::min<int>(32 / waves, (int)((160 * 1024) / lds)));
    const int64_t cap = (int64_t)g_eng.n_cu * per_cu;
    if (blocks > cap) blocks = cap;
    auto kern = lnl_kernel<MODE, WS>;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * waves), lds, r->stream, S, d_pix, d_theta,
                       d_lnL, d_spec, (long)B, nhf_max, wave_doubles, (const double *)g_eng.d_tabs);
    HIP_TRY(hipGetLastError());
    return NFA_OK;
}

static int launch_lnl(nfa_runner *r, const int *d_pix, const double *d_theta, double *d_lnL,
                      double *d_spec, int64_t B) {
    if (!g_eng.have_t0) return fail(NFA_ERR_STATE, "nfa_set_iemtex_table has not been called");
    if (g_eng.exp_mode == 0)
        return d_spec ? launch_lnl_t<0, true>(r, d_pix, d_theta, d_lnL, d_spec, B)
                      : launch_lnl_t<0, false>(r, d_pix, d_theta, d_lnL, d_spec, B);
    return d_spec ? launch_lnl_t<1, true>(r, d_pix, d_theta, d_lnL, d_spec, B)
                  : launch_lnl_t<1, false>(r, d_pix, d_theta, d_lnL, d_spec, B);
}

extern "C" {

static int check_pix(const nfa_runner *r, const int32_t *pix, int64_t B) {
    if (!pix) return NFA_OK;
    for (int64_t b = 0; b < B; ++b)
        if (pix[b] < 0 || pix[b] >= r->ss->n_pix) return fail(NFA_ERR_ARG, "pixel index out of range");
    return NFA_OK;
}

int nfa_runner_loglike_batch_dev(nfa_runner *r, const int32_t *d_pix, double *d_U, double *d_lnL,
                                 int64_t B) {
    if (!r || !d_U || !d_lnL) return fail(NFA_ERR_ARG, "null argument");
    if (!r->pr) return fail(NFA_ERR_STATE, "runner has no priors (predict-only)");
    if (B <= 0) return NFA_OK;
    hipEvent_t *e = nullptr;
    if (r->profiling) {
        if (r->ev_used + 3 > r->ev.size()) {
            for (int k = 0; k < 3; ++k) { hipEvent_t x; HIP_TRY(hipEventCreate(&x)); r->ev.push_back(x); }
        }
        e = &r->ev[r->ev_used];
        r->ev_used += 3;
        HIP_TRY(hipEventRecord(e[0], r->stream));
    }
    int rc = launch_priors(r->pr, d_U, B, r->ncomp, r->stream);
    if (rc) return rc;
    if (e) HIP_TRY(hipEventRecord(e[1], r->stream));
    rc = launch_lnl(r, d_pix, d_U, d_lnL, nullptr, B);
    if (rc) return rc;
    if (e) HIP_TRY(hipEventRecord(e[2], r->stream));
    return NFA_OK;
}

int nfa_runner_set_profiling(nfa_runner *r, int on) {
    if (!r) return fail(NFA_ERR_ARG, "null runner");
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->profiling = on != 0;
    r->ev_used = 0;
    return NFA_OK;
}

int nfa_runner_get_profile(nfa_runner *r, double *prior_ms, double *lnl_ms, int64_t *calls) {
    if (!r || !prior_ms || !lnl_ms || !calls) return fail(NFA_ERR_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(r->stream));
    double a = 0, b = 0;
    for (size_t k = 0; k + 2 < r->ev_used + 0 && k + 2 < r->ev.size() + 0; k += 3) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, r->ev[k], r->ev[k + 1])); a += t;
        HIP_TRY(hipEventElapsedTime(&t, r->ev[k + 1], r->ev[k + 2])); b += t;
    }
    *prior_ms = a; *lnl_ms = b; *calls = (int64_t)(r->ev_used / 3);
    r->ev_used = 0;
    return NFA_OK;
}

int nfa_runner_synchronize(nfa_runner *r) {
    if (!r) return fail(NFA_ERR_ARG, "null runner");
    HIP_TRY(hipStreamSynchronize(r->stream));
    return NFA_OK;
}

int nfa_runner_loglike_batch(nfa_runner *r, const int32_t *pix, double *U, double *lnL, int64_t B) {
    if (!r || !U || !lnL) return fail(NFA_ERR_ARG, "null argument");
    if (B <= 0) return NFA_OK;
    int rc = check_pix(r, pix, B); if (rc) return rc;
    rc = runner_reserve(r, B, false); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(r->d_U, U, sizeof(double) * B * r->ndim, hipMemcpyHostToDevice, r->stream));
    if (pix) HIP_TRY(hipMemcpyAsync(r->d_pix, pix, sizeof(int) * B, hipMemcpyHostToDevice, r->stream));
    rc = nfa_runner_loglike_batch_dev(r, pix ? r->d_pix : nullptr, r->d_U, r->d_lnL, B);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(U, r->d_U, sizeof(double) * B * r->ndim, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipMemcpyAsync(lnL, r->d_lnL, sizeof(double) * B, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return NFA_OK;
}

int nfa_runner_predict_batch(nfa_runner *r, const int32_t *pix, const double *theta, int64_t B,
                             double *spectra_out, double *lnL_out) {
    if (!r || !theta) return fail(NFA_ERR_ARG, "null argument");
    if (B <= 0) return NFA_OK;
    int rc = check_pix(r, pix, B); if (rc) return rc;
    rc = runner_reserve(r, B, spectra_out != nullptr); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(r->d_U, theta, sizeof(double) * B * r->ndim, hipMemcpyHostToDevice, r->stream));
    if (pix) HIP_TRY(hipMemcpyAsync(r->d_pix, pix, sizeof(int) * B, hipMemcpyHostToDevice, r->stream));
    rc = launch_lnl(r, pix ? r->d_pix : nullptr, r->d_U, r->d_lnL, spectra_out ? r->d_spec : nullptr, B);
    if (rc) return rc;
    if (spectra_out)
        HIP_TRY(hipMemcpyAsync(spectra_out, r->d_spec, sizeof(double) * B * r->ss->dev.chan_tot,
                               hipMemcpyDeviceToHost, r->stream));
    if (lnL_out)
        HIP_TRY(hipMemcpyAsync(lnL_out, r->d_lnL, sizeof(double) * B, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return NFA_OK;
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
int nfa_event_create(void **ev) {
    int rc = engine_init(); if (rc) return rc;
    hipEvent_t e; HIP_TRY(hipEventCreate(&e)); *ev = (void *)e; return NFA_OK;
}
int nfa_event_destroy(void *ev) { HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return NFA_OK; }
int nfa_event_record(void *ev, nfa_runner *r) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, r ? r->stream : 0)); return NFA_OK;
}
int nfa_event_synchronize(void *ev) { HIP_TRY(hipEventSynchronize((hipEvent_t)ev)); return NFA_OK; }
int nfa_event_elapsed_ms(void *start, void *stop, float *ms) {
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop)); return NFA_OK;
}

// ---- unit-test hooks ---------------------------------------------------------
int nfa_test_fastexp(const double *x, double *out, int64_t n, int mode) {
    int rc = engine_init(); if (rc) return rc;
    if (n <= 0) return NFA_OK;
    double *dx = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&dx, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dout, sizeof(double) * n));
    HIP_TRY(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    if (mode == 0) {
        const size_t lds = sizeof(double) * SM_END_TABLE;
        HIP_TRY(hipFuncSetAttribute((const void *)test_fastexp_kernel<0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(test_fastexp_kernel<0>, dim3(blocks), dim3(256), lds, 0, dx, dout, (long)n,
                           (const double *)g_eng.d_tabs);
    } else {
        hipLaunchKernelGGL(test_fastexp_kernel<1>, dim3(blocks), dim3(256), sizeof(double) * SM_END_POLY, 0,
                           dx, dout, (long)n, (const double *)g_eng.d_tabs);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout);
    return NFA_OK;
}

int nfa_test_iemtex(const double *x, double *out, int64_t n) {
    int rc = engine_init(); if (rc) return rc;
    if (!g_eng.have_t0) return fail(NFA_ERR_STATE, "nfa_set_iemtex_table has not been called");
    if (n <= 0) return NFA_OK;
    double *dx = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&dx, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dout, sizeof(double) * n));
    HIP_TRY(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(test_iemtex_kernel, dim3(blocks), dim3(256), sizeof(double) * SM_END_POLY, 0, dx, dout,
                       (long)n, (const double *)g_eng.d_tabs, g_eng.t0_xmin, g_eng.t0_xmax, g_eng.t0_inv_dx);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout);
    return NFA_OK;
}

int nfa_test_partition(const double *trot, double *qpara, double *qorth, int64_t n) {
    int rc = engine_init(); if (rc) return rc;
    if (n <= 0) return NFA_OK;
    double *dt = nullptr, *dp = nullptr, *dq = nullptr;
    HIP_TRY(hipMalloc(&dt, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dp, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dq, sizeof(double) * n));
    HIP_TRY(hipMemcpy(dt, trot, sizeof(double) * n, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((n * 64 + 255) / 256);
    if (g_eng.exp_mode == 0) {
        const size_t lds = sizeof(double) * SM_END_TABLE;
        HIP_TRY(hipFuncSetAttribute((const void *)test_partition_kernel<0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(test_partition_kernel<0>, dim3(blocks), dim3(256), lds, 0, dt, dp, dq, (long)n,
                           (const double *)g_eng.d_tabs);
    } else {
        hipLaunchKernelGGL(test_partition_kernel<1>, dim3(blocks), dim3(256), sizeof(double) * SM_END_POLY, 0,
                           dt, dp, dq, (long)n, (const double *)g_eng.d_tabs);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(qpara, dp, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(qorth, dq, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dt); (void)hipFree(dp); (void)hipFree(dq);
    return NFA_OK;
}

int nfa_test_windows(nfa_runner *r, int spec, double voff, double sigm, int32_t *lo, int32_t *hi) {
    if (!r || spec < 0 || spec >= r->ss->dev.n_spec) return fail(NFA_ERR_ARG, "bad spectrum index");
    int *dl = nullptr, *dh = nullptr;
    const int nhf = nfa_nhf[r->ss->dev.trans[spec] - 1];
    HIP_TRY(hipMalloc(&dl, sizeof(int) * 64));
    HIP_TRY(hipMalloc(&dh, sizeof(int) * 64));
    hipLaunchKernelGGL(test_windows_kernel, dim3(1), dim3(64), 0, 0, r->ss->dev, spec, voff, sigm, dl, dh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(lo, dl, sizeof(int) * nhf, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hi, dh, sizeof(int) * nhf, hipMemcpyDeviceToHost));
    (void)hipFree(dl); (void)hipFree(dh);
    return NFA_OK;
}

}  // extern "C"
