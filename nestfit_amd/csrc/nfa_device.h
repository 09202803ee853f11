// nfa_device.h -- device code of the MI355X (gfx950) NH3 log-likelihood engine.
//
// Hot path of autocorr/nestfit v0.2 (AmmoniaRunner.c_loglikelihood,
// nestfit/models/ammonia.pyx:423-432):
//
//   set-up stage    unit cube -> theta            core/core.pyx:459-476
//   (nfa_setup.h)   theta -> channel-independent  models/ammonia.pyx:337-361
//                   scalars (Trot/Tex, partition sums, main-line tau)
//   lnl_kernel      model spectrum + chi^2        models/hyperfine.pyx:52-118,
//                                                 core/core.pyx:522-530
//
// lnl_kernel execution model (wave = 64 lanes):
//   * a work item is one (theta, pixel); each of its spectra is one independent
//     wavefront (a 4096-row batch of 2 spectra puts 8 waves on every SIMD);
//   * inside a wave lanes are frequency channels: a row is 64 consecutive
//     channels (coalesced 512-B loads of x, data, T0, tbg); the optical depth of
//     the row lives in one register per lane;
//   * the (component, hyperfine line) constants are formed by the set-up stage with
//     lanes = lines (lines_kernel) and reach the row loop through scalar loads: line
//     constants are SGPR operands, the line's window is the EXEC mask; a ballot over
//     the line windows selects the lines that touch a row;
//   * chi^2 is reduced with DPP lane permutes; the per-spectrum terms of an item are
//     added in spectrum order by lnl_sum_kernel (bitwise reproducible).
// No MFMA: the path is elementwise fp64/fp32 plus reductions.
//
// Numerical modes (template MODE): 0 "table" evaluates FastExp like the reference
// (float-narrowed argument, Taylor below 2^-5, zero from 32, the three-table product
// with the same table indices) in fp64; 2 "fast" keeps every index computation in fp64
// but evaluates the exponentials in fp32 with split exponents (<= 3e-7 relative on
// Tb, tolerance of the metric: 1e-6).  (MODE 1, the fp64 polynomial form of the same
// exponential, survives in the set-up stage of the fast mode -- nf_fastexp<1> for the
// partition sums -- and in the test hooks; as a likelihood mode it was slower than the
// table mode and less faithful, and is gone.)
//
// Compile with -ffp-contract=off: FMA only where written, so window and table
// indices round like the reference's plain double arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MAXSPEC   16
#define MAXCOMP   10          // ResolvedPlacementPrior's own limit (core.pyx:399)
#define T0_SIZE   1000
// table layout inside g_tabs (doubles); the LDS copy starts at SM_EXP2
#define SM_T0X    0
#define SM_T0Y    1000
#define NFA_EXP2_N 256        // entries of the polynomial mode's table
#define SM_EXP2   2000        // 2^(i/256), i = 0..255        (poly)
// (table; C first, A last: a lane outside the tables' range -- it is given its value by the branch for such
// arguments -- still forms an address from its exponent bits, up to row 15 of a 10-row table: behind C lies B,
// behind B lies A, behind A at least SM_TABLE_TAIL doubles of whatever the kernel keeps there)
#define SM_FEC    (SM_EXP2 + NFA_EXP2_N)   // exp(-j 2^(l-28)) [10][256]
#define SM_FEB    (SM_FEC + 2560)   // exp(-j 2^(l-20)) [10][256]
#define SM_FEA    (SM_FEB + 2560)   // exp(-(128+j) 2^(l-12)) [10][128]
#define SM_END_POLY   (SM_EXP2 + NFA_EXP2_N)
#define SM_END_TABLE  (SM_FEA + 1280)   // 8432 doubles
#define SM_TABLE_TAIL 768               // doubles that must follow the staged tables in LDS (row 15 of A ends there)

// Transition tables of every model in one index space: 0..8 NH3 (1,1)..(9,9), 9..11 N2H+
// 1-0, 2-1, 3-2, 12 the Gaussian model's single "line" (offset 0, weight 1, rest frequency
// from the spectrum).  SpecDev.trans holds index + 1.
#define NFA_T_N2HP   NFA_N_LEVELS
#define NFA_T_GAUSS  (NFA_N_LEVELS + NFA_N2HP_LEVELS)
#define NFA_T_ALL    (NFA_T_GAUSS + 1)
__constant__ int    c_nhf[NFA_T_ALL];
__constant__ double c_nu[NFA_T_ALL];
__constant__ double c_ea[NFA_N_LEVELS];
__constant__ double c_voff[NFA_T_ALL][NFA_MAX_HF_N];
__constant__ double c_hfreq[NFA_T_ALL][NFA_MAX_HF_N];       // (1 - voff / CKMS) * nu of every line
__constant__ double c_tauw[NFA_T_ALL][NFA_MAX_HF_N];
// position of line i of transition t when the lines are ordered by their velocity offset (stable): in that order
// the lines whose windows touch a row of channels form ONE run of neighbours (fast mode's line table, lnl_body)
__constant__ unsigned char c_rank[NFA_T_ALL][NFA_MAX_HF_N];

struct SpecDev {
    int     n_spec, ncomp, cold, lte;
    int     model, npar;                 // NFA_MODEL_*, parameters per component (6 / 4 / 3)
    double  rest[MAXSPEC];               // line rest frequency (tables, or Spectrum.rest_freq)
    int     size[MAXSPEC], trans[MAXSPEC], off[MAXSPEC];
    double  nu_min[MAXSPEC], nu_chan[MAXSPEC];
    double  r_chan[MAXSPEC];             // correctly rounded 1 / nu_chan (0: take the division, nf_line)
    int64_t chan_tot;
    const double *xarr, *t0, *tbg, *data, *noise;
    const double *t0tbg;                 // T0 * tbg per channel (fast mode: g = B0x x^2 + A0x x - T0 tbg)
    const double *rowsq;                 // [n_pix][rows_tot]: sum of data^2 over each row of 64 channels
    const double *totsq;                 // [n_pix][n_spec]: sum of data^2 over a spectrum (its rows added in order)
    int     row_off[MAXSPEC];            // first row of spectrum s inside a pixel's rowsq slice
    int64_t rows_tot;                    // sum over the spectra of ceil(size / 64)
    double  t0_xmin, t0_xmax, t0_inv_dx;
};

// derived-parameter record of one item (doubles), written by setup_kernel (nfa_setup.h):
//   [c*4 + 0] tex  [c*4 + 1] sigm/CKMS  [c*4 + 2] voff/CKMS  [c*4 + 3] 1/tex
//   [4*ncomp + (c*nspec + s)*DREC_CS + 0] main-line optical depth of (component, spectrum)
//                                    + 1 kind, + 2.. the y(T0) = 1/(e^(T0/tex)-1) model:
// x = T0/tex is monotonic in the channel, so the table cells of the first and last channel
// of the spectrum bound the cells of all its channels.  The fast mode evaluates
//      up = !(T0 < split);  dT = T0 - m;  y = (up ? A1 : A0) + ((up ? B1 : B0) + q dT) dT
//   kind 1  one table cell:      A0 + B0 T0                         (split = inf, m = q = 0)
//   kind 2  two adjacent cells:  second cell from T0 >= split       (m = q = 0)
//   kind 3  outside the table:   Taylor of 1/expm1 about the band centre m (split = inf)
//   kind 0  anything else:       per-channel evaluation of hyperfine.pyx:23-45
// For kind 1 the record also holds the cell written in the frequency x of the channel
// (T0 = kappa x, kappa = h/k): A0X = A0 kappa, B0X = B0 kappa^2, so that
//      T0 (y - tbg) = B0X x^2 + A0X x - T0 tbg          (two fused multiply-adds per channel)
// The exact modes evaluate hyperfine.pyx:23-45 per channel, x = T0 / tex by division; where the first and
// the last channel fall in ONE table cell (XKIND = 1: the usual case) the cell -- slope XS, x_lo, y_lo -- is
// in the record and a channel computes slope (x - x_lo) + y_lo without finding the cell again: the same
// operations on the same operands, hence the same bits (division is monotonic: the cells of the two ends
// bound the cells of every channel between them).
#define DREC_CS 16
#define DK_TMAIN 0
#define DK_KIND  1
#define DK_A0X 2
#define DK_B0X 3
#define DK_A0 4
#define DK_B0 5
#define DK_A1 6
#define DK_B1 7
#define DK_SPLIT 8
#define DK_M 9
#define DK_Q 10
#define DK_XKIND 11
#define DK_XS 12
#define DK_XLO 13
#define DK_YLO 14
__host__ __device__ inline int drec_size(int ncomp, int nspec) { return 4 * ncomp + ncomp * nspec * DREC_CS; }

// Up to NFA_GROUP_MAX batches of `each` rows that a caller enqueues one after the other travel as ONE launch (the
// engine coalesces them, nfa_engine.hip): item b of the launch is row b - c * each of batch c = b / each, and every
// batch keeps its own pixel, unit-cube and result arrays.
// (Eight since round 5 -- four before: the table mode's launch of 16384 evaluations spends its last ~80 of ~200 us with
// fewer and fewer waves per SIMD, profiles/r05/queue_timeline.txt; eight batches in a launch halve that share:
// 86.3 -> 87.8 M evaluations/s on the metric shape, 26.0 -> 27.2 M on config 4, the fast mode unchanged.)
#ifndef NFA_GROUP_MAX
#define NFA_GROUP_MAX 8
#endif
struct BatchGroup {
    const int *pix[NFA_GROUP_MAX];
    double    *U[NFA_GROUP_MAX];
    double    *lnL[NFA_GROUP_MAX];
    double    *spec[NFA_GROUP_MAX];          // spectra out: the batch's B x chan_tot array (null: the launch writes none)
    long       each;
    int        n;
};
__device__ __forceinline__ int group_of(const BatchGroup &g, long b) {
    if (g.n <= 1) return 0;
    int c = 0;
#pragma unroll
    for (int k = 1; k < NFA_GROUP_MAX; ++k) c += (int)(b >= k * g.each);
    return c;
}

#define LNL_PARTS 4      // row parts of a unit: the fixed shape of its chi^2 sum
struct LnlGeom {
    int nhf_max;       // lines per component slot in the LDS line table
    int wave_doubles;  // LDS doubles per wave
    unsigned inv_nspec; // floor(2^32 / nspec) + 1: unit / nspec = mulhi(unit, inv_nspec) for unit < 2^28; 0: nspec == 1
    unsigned inv_nhf;   // floor(2^32 / nhf_max) + 1: p / nhf_max = mulhi(p, inv_nhf) for the few hundred (component, line) slots; 0: nhf_max == 1
    int split;          // waves that share one (item, spectrum) unit (1, 2, 4), each taking LNL_PARTS / split row parts
#ifdef NFA_TEST_HOOKS
    unsigned long long *trace;   // measurement (test library): per wave of the queue kernel 8 records {start, end, unit, position} in 10 ns ticks
#endif
    unsigned *queue;    // table mode, split == 1, launches of several units per wave slot (lnl_kernel_queue): the launch's
                        // chunk counter and, a 128-byte line behind it, its count of workgroups that have left
                        // (NFA_QUEUE_WORDS words, zero between launches); nullptr: one unit per wave (lnl_kernel)
    int ablate;        // timing experiments only: 1 skip Tb, 2 skip the line loop, 4 skip rows, 8 skip line set-up
};

// ---------------------------------------------------------------------------
//  wave-level helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the
    // compiler from moving LDS accesses across the hand-off point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Wave-wide sum with DPP lane permutations (no LDS traffic): butterfly inside each row of
// 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row sums
// are read through SGPRs.  Fixed order: bitwise reproducible; every lane gets the total.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
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

// Rarely taken libm paths kept out of line so they do not inflate the register
// budget of the hot loops.
__device__ __attribute__((noinline)) double slow_inv_expm1(double x) { return 1.0 / expm1(x); }
__device__ __attribute__((noinline)) double slow_exp(double x) { return exp(x); }
__device__ __attribute__((noinline)) double slow_pow(double x, double y) { return pow(x, y); }
__device__ __attribute__((noinline)) double slow_log10(double x) { return log10(x); }
__device__ __attribute__((noinline)) double slow_log(double x) { return log(x); }

// ---------------------------------------------------------------------------
//  FastExp replacement (reference: nestfit/core/fastexp.c:234-283, entered with
//  a double narrowed to float, nestfit/core/math.pxd:17)
// ---------------------------------------------------------------------------
// exp(-t) for t = (double)float in [2^-5, 32): n = rint(-t*256/ln2),
// exp(-t) = 2^(n>>8) * 2^((n&255)/256) * exp(r), |r| <= ln2/512: a degree-4 polynomial is exact to 4e-17 there
// (round 2 had 32 table entries and degree 6: two more fused multiply-adds in the dependent chain of every exponential).
__device__ __forceinline__ double exp_neg_poly(double t, const double *sm) {
    const double C256 = 369.3299304675746322841407183364843;  // 256/ln2
    const double L_HI = 6.93147180369123816490e-01 / 256.0;   // fdlibm ln2 split
    const double L_LO = 1.90821492927058770002e-10 / 256.0;
    const double n = __builtin_rint(-t * C256);
    double r = __builtin_fma(-n, L_HI, -t);
    r = __builtin_fma(-n, L_LO, r);
    const int ni = (int)n;
    const int m = ni & (NFA_EXP2_N - 1), q = ni >> 8;
    double p = 1.0 / 24.0;
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double v = sm[SM_EXP2 + m] * p;
    // multiply by 2^q (result stays normal: q >= -47 for t < 32)
    return __longlong_as_double(__double_as_longlong(v) + ((long long)q << 52));
}

// MODE 0: the reference's three-table product; MODE 1: polynomial.  The
// branches of fastexp.c (negative, zero, Taylor, >= 32) are handled under one wave-uniform test.
// NONNEG: the caller knows x >= +0 or NaN (a square times a positive number; a sum of such terms).
// BOUNDED: the caller also knows x < 32 and not NaN (the line loop: the window set-up sees to it).
// Table addresses come straight from the bits of the float: u = bits - (122 << 23) holds l = exponent - 122
// (fastexp.c:262) in bits 23..26 where l is in range, so (u >> 16) & 0x7ff = l * 128 + j0 and the rows of B and
// C start at l << 11 bytes.
template <int MODE, bool NONNEG = false, bool BOUNDED = false>
__device__ __forceinline__ double nf_fastexp(double xd, const double *sm) {
    const float x = (float)xd;                               // math.pxd:17 narrowing
    const uint32_t bits = __float_as_uint(x);
    const uint32_t u = (NONNEG ? bits : (bits & 0x7fffffffu)) - (122u << 23);
    double r;
    if (MODE == 0) {
        // bit-field extract + shift-add pairs, the LDS base folded into the adds (the compiler's own form of
        // this spends two more instructions on adding the base); B sits a constant 2560 doubles behind C
        typedef const __attribute__((address_space(3))) double *lds_dbl_p;
        const uint32_t base_a = (uint32_t)(uintptr_t)(sm + SM_FEA), base_c = (uint32_t)(uintptr_t)(sm + SM_FEC);
        uint32_t oa, ob, oc, rc;
        asm("v_bfe_u32 %[oa], %[u], 16, 11\n\t"             // l * 128 + j0, fastexp.c:276
            "v_lshl_add_u32 %[oa], %[oa], 3, %[ba]\n\t"
            "v_bfe_u32 %[rc], %[u], 23, 4\n\t"              // row l of C
            "v_lshl_add_u32 %[rc], %[rc], 11, %[bc]\n\t"
            "v_bfe_u32 %[ob], %[bits], 8, 8\n\t"            // j1, fastexp.c:277
            "v_lshl_add_u32 %[ob], %[ob], 3, %[rc]\n\t"
            "v_and_b32 %[oc], 0xff, %[bits]\n\t"            // j2, fastexp.c:278
            "v_lshl_add_u32 %[oc], %[oc], 3, %[rc]"
            : [oa] "=&v"(oa), [ob] "=&v"(ob), [oc] "=&v"(oc), [rc] "=&v"(rc)
            : [u] "v"(u), [bits] "v"(bits), [ba] "s"(base_a), [bc] "s"(base_c));
        r = *(lds_dbl_p)(uintptr_t)oa * *((lds_dbl_p)(uintptr_t)ob + (SM_FEB - SM_FEC)) * *(lds_dbl_p)(uintptr_t)oc;
    } else {
        r = exp_neg_poly((double)x, sm);                      // out-of-range lanes are replaced below
    }
    bool special = BOUNDED ? (int32_t)u < 0 : u >= (10u << 23);      // l < 0 [or l >= 10 (also NaN, inf)]
    if (!NONNEG) special = special || (x < 0.0f);
    if (__builtin_amdgcn_ballot_w64(special) != 0ull) {
        const double t = (double)x;                           // fastexp.c:264-270; x == 0 gives exactly 1
        double ty = 1.0 - t * (1.0 / 3.0);
        ty = 1.0 - (t * ty) * 0.5;
        ty = 1.0 - (t * ty);
        const bool small = (int32_t)u < 0;                    // l < 0
        r = small ? ty : r;
        if (!BOUNDED) r = (special && !small) ? 0.0 : r;      // fastexp.c:272-273
        if (!NONNEG) { if (x < 0.0f) r = slow_exp(-(double)x); }        // fastexp.c:259
    }
    return r;
}

// 1 - FastExp(tau) for the Tb pass of the exact modes (hyperfine.pyx:109-113), tau >= +0 or NaN: the same value,
// bit for bit, as 1.0 - nf_fastexp<MODE, true>(tau, sm).  Most rows lie in the line wings, where every lane is
// below 2^-5 and FastExp is its Taylor branch (fastexp.c:264-270): that branch is evaluated first, and the table
// (or polynomial) form with its three gathers only when a lane of the row needs it.
template <int MODE>
__device__ __forceinline__ double nf_one_minus_fastexp_row(double tau, const double *sm) {
    const float x = (float)tau;                               // math.pxd:17 narrowing
    const double t = (double)x;
    double r = 1.0 - t * (1.0 / 3.0);                         // x == 0 gives exactly 1
    r = 1.0 - (t * r) * 0.5;
    r = 1.0 - (t * r);
    const uint32_t u = __float_as_uint(x) - (122u << 23);
    const bool small = (int32_t)u < 0;                        // l < 0
    if (__builtin_amdgcn_ballot_w64(!small) != 0ull) {
        asm volatile("" ::: "memory");                        // keep it a branch
        const double big = nf_fastexp<MODE, true>(tau, sm);   // the general form (rows at a line centre)
        r = small ? r : big;
    }
    return 1.0 - r;
}

// exp(-x) in fp32 with the exponent split in two floats (MODE 2).  Like FastExp:
// exactly 0 from x = 32 (and for NaN), exp(|x|) for negative x.
__device__ __forceinline__ float exp_neg_f32(float x) {
    const float NEG_L2E_HI = -1.44269502162933349609375f;         // -(float)log2(e)
    const float NEG_L2E_LO = -1.925963033500011e-08f;             // -(log2(e) - hi)
    const float LN2F = 0.693147180559945f;
    const float yh = x * NEG_L2E_HI;
    float yl = __builtin_fmaf(x, NEG_L2E_HI, -yh);
    yl = __builtin_fmaf(x, NEG_L2E_LO, yl);
    float e = __builtin_amdgcn_exp2f(yh);
    e = __builtin_fmaf(e * LN2F, yl, e);
    return (x < 32.0f) ? e : 0.0f;
}

// 1/(e^x-1) (nestfit/models/hyperfine.pyx:23-45), tables in global memory:
// the index is nearly uniform over a row, the reads stay in L1.
__device__ __forceinline__ double nf_iemtex(double x, const double *__restrict__ t0x,
                                            const double *__restrict__ t0y, double xmin,
                                            double xmax, double inv_dx) {
    const bool in_tab = (xmin < x) && (x < xmax);
    long i_lo = in_tab ? (long)((x - xmin) * inv_dx) : 0;
    i_lo = i_lo > T0_SIZE - 2 ? T0_SIZE - 2 : i_lo;           // never taken inside the table
    const double x_lo = t0x[i_lo];
    const double y_lo = t0y[i_lo];
    const double y_hi = t0y[i_lo + 1];
    const double slope = (y_hi - y_lo) * inv_dx;
    double res = slope * (x - x_lo) + y_lo;
    if (!in_tab) res = slow_inv_expm1(x);
    return res;
}

__device__ __forceinline__ double nf_swift(double tkin) {    // ammonia.pyx:280-286
    return tkin / (1.0 + (tkin / 41.18) * slow_log(1.0 + 0.6 * slow_exp(-15.7 / tkin)));
}

// E_J / (k trot) of the metastable level J (ammonia.pyx:289-295): grows with J (C j^2 + B j)
__device__ __forceinline__ double nf_partition_arg(int j, double trot) {
    const double dj = (double)j;
    return NFA_H * (NFA_BROT * dj * (double)(j + 1) + (NFA_CROT - NFA_BROT) * dj * dj) / (NFA_KB * trot);
}
template <int MODE>
__device__ __forceinline__ double nf_partition_level(int j, double trot, const double *sm) {
    return (double)(2 * j + 1) * nf_fastexp<MODE>(nf_partition_arg(j, trot), sm);
}

// Line centre, width and channel window of hyperfine line i of transition t
// (reference: nestfit/models/hyperfine.pyx:70-91).  Plain double arithmetic,
// no contraction: the floor() arguments must round like the reference's.
struct LineConst { double nucen, idenom; int lo, hi; };
__device__ __forceinline__ LineConst nf_line(int t, int i, double v_over_c, double s_over_c, double nu0,
                                             double nu_min, double nu_chan, int N, double r_chan = 0.0) {
    LineConst r;
    // hf_freq = (1 - voff_i / CKMS) nu0 (hyperfine.pyx:71) is a constant of the line: the host forms it
    // with the same two IEEE operations at start-up (c_hfreq); the Gaussian model's line has voff = 0
    const double hf_freq   = t == NFA_T_GAUSS ? nu0 : c_hfreq[t][i];
    const double hf_width  = s_over_c * hf_freq;             // (sigm / CKMS) * hf_freq
    const double hf_offset = v_over_c * hf_freq;             // (voff / CKMS) * hf_freq
    // the Gaussian model forms its centre as rest_freq * (1 - voff / CKMS) (gaussian.pyx:33)
    const double hf_nucen  = t == NFA_T_GAUSS ? nu0 * (1 - v_over_c) : hf_freq - hf_offset;
    const double hf_idenom = 0.5 / (hf_width * hf_width);
    const double nu_cutoff = sqrt(12.5 / hf_idenom);
    const double nu_lo = (hf_nucen - nu_min - nu_cutoff);
    const double nu_hi = (hf_nucen - nu_min + nu_cutoff);
    // the two quotients by the channel width (hyperfine.pyx:78-79), correctly rounded without a division each: with
    // r = RN(1 / nu_chan), q = a r, e = a - nu_chan q (exact, fused), x = q + e r is RN(a / nu_chan) -- Markstein's
    // final step, as in the Tb pass (the same bits as the division; the host passes r = 0 for a width whose mantissa
    // is all ones, the one case the theorem leaves out; a centre that is not finite ends as an empty window either way)
    double q_lo, q_hi;
    if (r_chan != 0.0) {
        const double ql = nu_lo * r_chan, qh = nu_hi * r_chan;
        q_lo = __builtin_fma(__builtin_fma(-nu_chan, ql, nu_lo), r_chan, ql);
        q_hi = __builtin_fma(__builtin_fma(-nu_chan, qh, nu_hi), r_chan, qh);
    } else {
        q_lo = nu_lo / nu_chan;
        q_hi = nu_hi / nu_chan;
    }
    // (long) floor(.) of both, `continue` on an empty window, clipping to the spectrum (hyperfine.pyx:78-86) -- in
    // doubles, then ONE conversion each: the integers are below 2^31 once they are clipped, and a double converts to a
    // 64-bit integer through six instructions here.  (A quotient that is not finite belongs to a centre or width that is
    // not: the caller empties that window.)
    const double flo = floor(q_lo), fhi = floor(q_hi), last = (double)(N - 1);
    int lo = 0, hi = 0;
    if (!(fhi < 0.0 || flo > last)) {
        lo = (int)fmax(flo, 0.0);
        hi = (int)fmin(fhi, last);
    }
    r.nucen = hf_nucen; r.idenom = hf_idenom; r.lo = lo; r.hi = hi;
    return r;
}

// The LDS copy of the exponential tables starts at smem[0]; helper functions
// index with the absolute g_tabs offsets, hence the shifted base pointer.
template <int MODE>
__device__ __forceinline__ const double *stage_exp_tables(double *smem, const double *g_tabs, int *n_shared) {
    const int n = (MODE == 0) ? (SM_END_TABLE - SM_EXP2) : NFA_EXP2_N;
    for (int i = threadIdx.x; i < n; i += blockDim.x) smem[i] = g_tabs[SM_EXP2 + i];
    __syncthreads();
    *n_shared = n;
    return smem - SM_EXP2;
}

// ---------------------------------------------------------------------------
//  line records (hyperfine.pyx:68-91) of one (item, spectrum) unit, in the wave's LDS slice
// ---------------------------------------------------------------------------
// 32 bytes = two 16-byte broadcast reads.  The table of a component is kept in the order of the lines' velocity
// offsets (c_rank), so that the lines whose windows touch a row of channels are ONE run of neighbours; the windows
// themselves, [lo, hi), lie behind the table as an array of their own (the row loop's hit masks).
//   nucen   line centre (hyperfine.pyx:73)
//   idenom  0.5 / width^2 (hyperfine.pyx:75)
//   w       weight hf_tau (hyperfine.pyx:74): a double in the exact modes, a float in the low word in the fast mode
//   mid, half   the window [lo, lo + len) as |j - mid| < half, mid = lo + (len - 1) / 2, half = len / 2: exact in
//           fp32 for spectra below 2^22 channels (longer ones take the wide form, which tests integers)
struct __attribute__((aligned(16))) LineRec {
    double nucen, idenom;                        // first 16-byte read
    double w;                                    // second 16-byte read
    float mid, half;
};
typedef const __attribute__((address_space(4))) double *k_dbl_p;     // constant address space: a
                                                                     // uniform index gives an s_load

// exp(-x), x = float in [0, 32): 2^yh * (1 + r) with yh = fl(-x log2 e) and r = -x - yh ln 2
// carried in two fused steps (|r| < 4e-6, so e^r = 1 + r to 1e-11); <= 2e-7 relative.
__device__ __forceinline__ float exp_neg_core_f32(float x) {
    const float yh = x * -1.44269502162933349609375f;
    float r = __builtin_fmaf(yh, -0.693147182464599609375f, -x);        // -x - yh ln2_hi (exact product)
    r = __builtin_fmaf(yh, 1.904654299957e-09f, r);                     //    - yh ln2_lo
    const float e0 = __builtin_amdgcn_exp2f(yh);
    return __builtin_fmaf(e0, r, e0);
}

// lane masks straight from one compare (a ballot of a combined bool costs a select and a second compare)
#define NF_ICMP_SGT 38
#define NF_ICMP_SLT 40
#define NF_FCMP_OLT 4
#define NF_FCMP_UGE 11
#define NF_FCMP_UNE 14
__device__ __forceinline__ unsigned long long lanes_lt(int a, int b) { return __builtin_amdgcn_sicmp(a, b, NF_ICMP_SLT); }
__device__ __forceinline__ unsigned long long lanes_gt(int a, int b) { return __builtin_amdgcn_sicmp(a, b, NF_ICMP_SGT); }

// 1 - FastExp(tau) of the reference for fp32 tau (MODE 2):
//   tau < 2^-5   the reference's cubic  tau (1 - tau/2 (1 - tau/3))   (fastexp.c:264-270)
//   tau < 0.25   tau * P4(tau), a degree-4 fit of (1 - e^-tau)/tau on [2^-5, 1/4]
//   otherwise    1 - exp(-tau) (6e-8 e/(1-e) <= 2.2e-7; exactly 1 from 32 and for NaN)
// Below 1e-8 the reference's own evaluation 1 - (1 - q) is quantised in steps of 2^-53 (it
// even returns exactly 0 below 1.1e-16, which decides the zero pattern of faint channels):
// those lanes repeat that rounding in fp64, under a wave-uniform branch that is rarely taken.
// The two upper ranges are one instruction block in which the range IS the EXEC mask (v_cmpx: the lanes
// of a range compute and write, the others idle; a range no lane is in is jumped over): a select per
// range costs a compare and a v_cndmask, four cycles each against 2.3 for an fp32 multiply-add
// (profiles/r02/ubench_valu.txt).  EXEC must be all ones on entry (every branch around a call is
// wave-uniform) and is all ones on exit.
// `live`: the lanes whose tau is not zero (a lane outside every window of its row holds an exact 0: without the
// mask such a lane alone would send most passes through the sub-1e-8 branch, which leaves its 0 a 0).
__device__ __forceinline__ double one_minus_fastexp_f32(float t, unsigned long long live = ~0ull) {
    float pc = __builtin_fmaf(t, 1.0f / 6.0f, -0.5f);            // 1 - t/2 + t^2/6: the reference's cubic / t
    pc = __builtin_fmaf(t, pc, 1.0f);
    float wf = t * pc;
    // Most rows lie in the line wings where every lane is below 2^-5 and the cubic is all there is.
    float p, q, e, r;
    asm volatile("v_cmpx_ngt_f32 0x3d000000, %[t]\n\t"                  // lanes at or above 2^-5 (and NaN)
                 "s_cbranch_execz 1f\n\t"
                 // (1 - e^-t)/t on [2^-5, 1/4]: degree-4 interpolant at the Chebyshev nodes of the interval
                 // (1.2e-7 relative in fp32 arithmetic, the rounding floor; the degree-6 Taylor sum was no better)
                 "v_fmamk_f32 %[p], %[t], 0x3bf30129, %[c3]\n\t"         //   0.00741590978577733 t - 0.04143298789858818
                 "v_fmaak_f32 %[p], %[p], %[t], 0x3e2aa388\n\t"          //   ... t + 0.16663944721221924
                 "v_fmaak_f32 %[p], %[p], %[t], 0xbeffffd1\n\t"          //   ... t - 0.4999985992908478
                 "v_fma_f32 %[p], %[p], %[t], 1.0\n\t"
                 "v_mul_f32 %[wf], %[t], %[p]\n\t"
                 "v_cmpx_ngt_f32 0x3e800000, %[t]\n\t"                  // of those, the lanes at or above 1/4
                 "s_cbranch_execz 1f\n\t"
                 // 1 - exp(-t) like exp_neg_f32; t capped at 64: from 32 on (and for NaN, which v_min drops)
                 // 2^yh is below 2^-46 and the difference is exactly 1
                 "v_min_f32 %[p], 0x42800000, %[t]\n\t"
                 "v_mul_f32 %[e], 0xbfb8aa3b, %[p]\n\t"                  // yh = -x log2(e)_hi
                 // (the result of a transcendental instruction must not be read by the next vector instruction on
                 // gfx940+: the compiler pads its own code, nobody pads this block -- two instructions lie between)
                 "v_exp_f32 %[r], %[e]\n\t"
                 "v_fma_f32 %[q], %[p], %[l2e], -%[e]\n\t"               // yl = (-x log2(e)_hi - yh)
                 "v_fmac_f32 %[q], 0xb2a57060, %[p]\n\t"                 //      - x log2(e)_lo
                 "v_mul_f32 %[p], 0x3f317218, %[r]\n\t"                  // 2^yh ln 2
                 "v_fmac_f32 %[r], %[p], %[q]\n\t"                       // 2^yh (1 + yl ln 2)
                 "v_sub_f32 %[wf], 1.0, %[r]\n\t"
                 "1:\n\t"
                 "s_mov_b64 exec, -1"
                 : [wf] "+v"(wf), [p] "=&v"(p), [q] "=&v"(q), [e] "=&v"(e), [r] "=&v"(r)
                 : [t] "v"(t), [c3] "v"(-0.04143298789858818f), [l2e] "s"(-1.44269502162933349609375f)
                 : "vcc");
    double w = (double)wf;
    const bool tiny = t < 1e-8f;
    if ((__builtin_amdgcn_fcmpf(t, 1e-8f, NF_FCMP_OLT) & live) != 0ull) {
        asm volatile("" ::: "memory");            // keep the rare path a branch (no if-conversion)
        const double r1 = 1.0 - w;                 // (a tiny lane is below 2^-5: its w is the cubic t * pc)
        w = tiny ? 1.0 - r1 : w;
    }
    return w;
}

// ---------------------------------------------------------------------------
//  lnl_kernel: one wavefront per (item, spectrum) unit, no coupling between waves.
//  Lanes = channels of a row of 64; the optical depth of the row lives in one register per
//  lane.  The wave forms the line constants of its spectrum with lanes = (component, line)
//  into its LDS slice (32-byte records); in the row loop a ballot over the windows selects
//  the lines touching the row, their records are broadcast-read (two ds_read_b128) and the
//  window of the line is applied as the EXEC mask: lanes outside it are idle instead of
//  multiplied by zero.  Per line x row: thirteen vector instructions -- the LDS address, two
//  for the window test, three fp64 operations for the float-narrowed FastExp argument
//  (math.pxd:17), the conversion, five fp32 operations for exp, one multiply-add -- and six
//  scalar ones (find the line, clear its bit, compare, restore EXEC, branch, wait; the scalar unit is
//  shared by the four SIMDs of a CU: a scalar instruction costs the wave as much as an fp64 one,
//  scripts/ubench_lineloop.hip).
//  Rows without any line window (about 40 % at the metric shape) cost two compares per
//  component: their chi^2 term is the precomputed sum of data^2 of the row (SpecDev.rowsq).
//  NCOMP > 0: the number of components is a compile-time constant, the component loop is
//  unrolled, line windows and the constants of the Tb pass live in registers; NCOMP == 0 is
//  the general form (any ncomp up to MAXCOMP, windows re-read from LDS, constants through
//  scalar loads).  lnl_sum_kernel adds the terms of an item in spectrum order
//  (ammonia.pyx:429-432).
// ---------------------------------------------------------------------------
// The fast mode's line x row step as one instruction block: window test -> EXEC, float-narrowed Gaussian argument
// formed exactly as the reference forms it ((x - nucen)^2 * idenom in fp64, hyperfine.pyx:94, then narrowed,
// math.pxd:17), exp, tau += w e; EXEC is all ones on entry (every branch around it is wave-uniform) and on exit.
//   * the window as |j - mid| < half in fp32: an fp32 subtraction runs at twice the rate of an integer one.
//   * exp as 2^yh (1 + r), r = -x - yh ln2 in ONE fused step (the exact product with the float nearest ln 2;
//     what is dropped, yh (ln 2 - fl(ln 2)), is below 3.5e-8 relative inside a window, x <= 12.5).
// (Tried and dropped: the argument as ((x - nucen) sqrt(idenom))^2 -- it is two multiplications either way.)
__device__ __forceinline__ void line_step_fastz(float &tau, float jf, double xj, double nucen, double idenom,
                                                float htau, float mid, float half) {
    double d;
    float t0, t1, t2;
    asm volatile("v_sub_f32 %[t0], %[jf], %[mid]\n\t"
                 "v_cmpx_lt_f32_e64 vcc, |%[t0]|, %[half]\n\t"
                 "v_add_f64 %[d], %[xj], -%[nucen]\n\t"
                 "v_mul_f64 %[d], %[d], %[d]\n\t"
                 "v_mul_f64 %[d], %[d], %[idenom]\n\t"
                 "v_cvt_f32_f64 %[t0], %[d]\n\t"                       // math.pxd:17 narrowing
                 "v_mul_f32 %[t1], 0xbfb8aa3b, %[t0]\n\t"               // yh = -x log2(e)
                 "v_exp_f32 %[t2], %[t1]\n\t"
                 "v_fma_f32 %[t0], %[t1], %[kln2], -%[t0]\n\t"          // r = -x - yh ln2 (also the wait state behind v_exp)
                 "v_fmac_f32 %[t2], %[t2], %[t0]\n\t"                   // e = 2^yh (1 + r)
                 "v_fmac_f32 %[tau], %[htau], %[t2]\n\t"
                 "s_mov_b64 exec, -1"
                 : [tau] "+v"(tau), [d] "=&v"(d), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
                 : [jf] "v"(jf), [mid] "v"(mid), [half] "v"(half), [xj] "v"(xj), [nucen] "v"(nucen),
                   [idenom] "v"(idenom), [htau] "v"(htau), [kln2] "s"(-0.693147182464599609375f)
                 : "vcc");
}

// The staged product tables start at LDS address 0 (the kernels hold no static LDS; stage_exp_tables copies
// g_tabs[SM_EXP2 ...] to smem[0 ...]): their byte offsets are instruction immediates.
#define NFA_LDS_OFF_C ((SM_FEC - SM_EXP2) * 8)     //  2048
#define NFA_LDS_OFF_B ((SM_FEB - SM_EXP2) * 8)     // 22528
#define NFA_LDS_OFF_A ((SM_FEA - SM_EXP2) * 8)     // 43008
#define NFA_STR2(x) #x
#define NFA_STR(x) NFA_STR2(x)

// FastExp's table product for the float X (its bits in a register), all lanes of EXEC: the three gathers issued, nothing
// waited for.  u = bits - (122 << 23) holds l = exponent - 122 (fastexp.c:262) in bits 23..26 and (l, j0) = the A index
// as its upper half-word; j1 and j2 are bytes 1 and 0 of the bits (fastexp.c:276-278).  Sub-dword operand selects do
// the extractions inside the shifts.  Six address instructions (round 3: nine -- a bit-field extract and a shift-add per
// index; round 4: seven -- a row register and a shift-add per table): u << 1 has l as its top byte (bits 27..30 of u are
// clear for a float in the table's range); ONE byte permute puts (l, j1) and (l, j2) into the two half-words of a
// register, and each half-word shifted by three is an address.  A lane outside the table's range (u negative: the Taylor
// form's; 32 and more, NaN: an exact zero instead) forms addresses beyond the workgroup's LDS for all three: such a read
// returns nothing and faults nothing.
// (The C factor gathered through the vector-memory path instead: profiles/r05/ab_table_cgather_vmem.txt; that build's
// macros are in this file as of commit 4af2f27.)
#define NFA_TABLE_ADDR(X, G0, G1, G2) NFA_TABLE_ADDR_U(t0, X, G0, G1, G2)        /* u in t0 */
#define NFA_TABLE_ADDR_U(U, X, G0, G1, G2)                                                                 \
        "v_lshlrev_b32_sdwa %[t1], 3, %[" #U "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" /* (l, j0) * 8 */ \
        "ds_read_b64 %[" #G0 "], %[t1] offset:" NFA_STR(NFA_LDS_OFF_A) "\n\t"                              \
        "v_lshlrev_b32 %[t0], 1, %[" #U "]\n\t"                   /* top byte: l */                        \
        "v_perm_b32 %[t0], %[t0], %[" #X "], %[psel]\n\t"          /* bytes (l, j1, l, j2) */               \
        "v_lshlrev_b32_sdwa %[t1], 3, %[t0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" /* (l, j1) * 8 */ \
        "ds_read_b64 %[" #G1 "], %[t1] offset:" NFA_STR(NFA_LDS_OFF_B) "\n\t"                              \
        "v_lshlrev_b32_sdwa %[t1], 3, %[t0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t" /* (l, j2) * 8 */ \
        "ds_read_b64 %[" #G2 "], %[t1] offset:" NFA_STR(NFA_LDS_OFF_C) "\n\t"
#define NFA_TABLE_GATHER(X, G0, G1, G2)                                                                    \
        "v_add_u32 %[t0], 0xc3000000, %[" #X "]\n\t"              /* bits - (122 << 23) */                 \
        NFA_TABLE_ADDR(X, G0, G1, G2)
#define NFA_PSEL_OPERAND , [psel] "s"(0x07010700u)             // v_perm_b32 D, S0, S1: selector bytes 4..7 = S0's, 0..3 = S1's
// The line blocks' form: the subtraction's carry -- set where the bits are at least 122 << 23, i.e. where x >= 2^-5 -- lands
// in VCC, and the lanes of the Taylor form (fastexp.c:264) are EXEC without VCC: one scalar instruction where round 4 spent
// a vector compare per line x row step.
#define NFA_W_A01 "s_waitcnt lgkmcnt(4)\n\t"
#define NFA_W_A2  "s_waitcnt lgkmcnt(3)\n\t"
#define NFA_W_B01 "s_waitcnt lgkmcnt(1)\n\t"
#define NFA_W_B2  "s_waitcnt lgkmcnt(0)\n\t"
#define NFA_W_S01 "s_waitcnt lgkmcnt(1)\n\t"
#define NFA_W_S2  "s_waitcnt lgkmcnt(0)\n\t"
#define NFA_TABLE_LOOKUP(X, G0, G1, G2, NUC, ID)                                                           \
        "v_add_f64 %[" #G0 "], %[xj], -%[" #NUC "]\n\t"                                                    \
        "v_mul_f64 %[" #G0 "], %[" #G0 "], %[" #G0 "]\n\t"                                                 \
        "v_mul_f64 %[" #G0 "], %[" #G0 "], %[" #ID "]\n\t"                                                 \
        "v_cvt_f32_f64 %[" #X "], %[" #G0 "]\n\t"                 /* math.pxd:17 narrowing */             \
        "v_add_co_u32 %[t0], vcc, 0xc3000000, %[" #X "]\n\t"      /* bits - (122 << 23); VCC: x >= 2^-5 */ \
        NFA_TABLE_ADDR(X, G0, G1, G2)
// FastExp's Taylor form (fastexp.c:264-270: 1 - t (1 - t/2 (1 - t/3)), one IEEE operation per operation of the
// reference) for the lanes in VCC.  The middle step 1 - (t ty) 0.5 is ONE fused multiply-add: a product with 0.5 is
// exact, so fma(t ty, -0.5, 1) rounds once, where the reference's multiplication and subtraction round once too.
#define NFA_TABLE_TAYLOR(X, G0, G1, G2, LBL, M, BR, TM)                                                              \
        BR " " LBL "%=\n\t"                                                                     \
        "s_mov_b64 exec, " TM "\n\t"                                 /* (a subset of the window) */                \
        "v_cvt_f64_f32 %[" #G1 "], %[" #X "]\n\t"                                                          \
        "v_mul_f64 %[" #G2 "], %[" #G1 "], %[nthird]\n\t"          /* 1 - t / 3 */                         \
        "v_add_f64 %[" #G2 "], %[" #G2 "], 1.0\n\t"                                                        \
        "v_mul_f64 %[" #G2 "], %[" #G2 "], %[" #G1 "]\n\t"         /* 1 - (t ty) / 2 */                    \
        "v_fma_f64 %[" #G2 "], %[" #G2 "], -0.5, 1.0\n\t"                                                  \
        "v_mul_f64 %[" #G1 "], %[" #G2 "], %[" #G1 "]\n\t"         /* 1 - t ty */                          \
        "v_add_f64 %[" #G0 "], -%[" #G1 "], 1.0\n\t"                                                       \
        "s_mov_b64 exec, %[" #M "]\n\t"                           /* back to the window */                \
        LBL "%=:\n\t"

// Two line x row steps of the table mode as one instruction block: both lines' FastExp arguments and table addresses
// are formed and all six gathers issued before the first product is taken, each line under its own window as the
// EXEC mask.  (Compiled from C++ the two steps of a pair run one after the other, each waiting for its own gathers:
// the wave then sits through two trips to LDS per pair with nothing of its own to issue, and the kernel was bound by
// neither the vector ALUs (75 % busy) nor the LDS array.)  The arithmetic is nf_fastexp<0, true, true>'s, operation
// for operation: (x - nucen)^2 idenom (hyperfine.pyx:94), the float narrowing (math.pxd:17), A B C in that order
// (fastexp.c:276-279), the Taylor form below 2^-5 (fastexp.c:264-270) where a lane of the line needs it, tau += w e.
// EXEC is all ones on entry and on exit.
__device__ __forceinline__ void line_pair_table(double &tau, float jf, double xj,
                                                double nucA, double idA, double wA, float midA, float halfA,
                                                double nucB, double idB, double wB, float midB, float halfB) {
    float xA, xB;
    uint32_t t0, t1;
    double a0, a1, a2, b0, b1, b2;
    unsigned long long mA, mB, tA;
    asm volatile(
        "v_sub_f32 %[t0], %[jf], %[midA]\n\t"
        "v_cmp_lt_f32_e64 %[mA], |%[t0]|, %[halfA]\n\t"
        "v_sub_f32 %[t0], %[jf], %[midB]\n\t"
        "v_cmp_lt_f32_e64 %[mB], |%[t0]|, %[halfB]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        NFA_TABLE_LOOKUP(xA, a0, a1, a2, nucA, idA)
        "s_andn2_b64 %[tA], exec, vcc\n\t"                       // line A's lanes of the Taylor form
        "s_mov_b64 exec, %[mB]\n\t"
        NFA_TABLE_LOOKUP(xB, b0, b1, b2, nucB, idB)
        "s_andn2_b64 vcc, exec, vcc\n\t"                         // line B's
        // line A: the product as its gathers land, the Taylor form where x < 2^-5, tau += w e
        "s_mov_b64 exec, %[mA]\n\t"
        NFA_W_A01
        "v_mul_f64 %[a0], %[a0], %[a1]\n\t"
        NFA_W_A2
        "v_mul_f64 %[a0], %[a0], %[a2]\n\t"
        "s_cmp_lg_u64 %[tA], 0\n\t"
        NFA_TABLE_TAYLOR(xA, a0, a1, a2, ".Lnfa_tpa_", mA, "s_cbranch_scc0", "%[tA]")
        "v_fmac_f64 %[tau], %[wA], %[a0]\n\t"
        // line B
        "s_mov_b64 exec, %[mB]\n\t"
        NFA_W_B01
        "v_mul_f64 %[b0], %[b0], %[b1]\n\t"
        NFA_W_B2
        "v_mul_f64 %[b0], %[b0], %[b2]\n\t"
        NFA_TABLE_TAYLOR(xB, b0, b1, b2, ".Lnfa_tpb_", mB, "s_cbranch_vccz", "vcc")
        "v_fmac_f64 %[tau], %[wB], %[b0]\n\t"
        "s_mov_b64 exec, -1"
        : [tau] "+v"(tau), [xA] "=&v"(xA), [xB] "=&v"(xB), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2),
          [mA] "=&s"(mA), [mB] "=&s"(mB), [tA] "=&s"(tA)
        : [jf] "v"(jf), [xj] "v"(xj), [nucA] "v"(nucA), [idA] "v"(idA), [wA] "v"(wA), [midA] "v"(midA), [halfA] "v"(halfA),
          [nucB] "v"(nucB), [idB] "v"(idB), [wB] "v"(wB), [midB] "v"(midB), [halfB] "v"(halfB),
          [nthird] "s"(-(1.0 / 3.0)) NFA_PSEL_OPERAND
        : "vcc", "scc");
}

// One line x row step of the table mode (the odd line of a run), the same arithmetic.
__device__ __forceinline__ void line_single_table(double &tau, float jf, double xj,
                                                  double nucA, double idA, double wA, float midA, float halfA) {
    float xA;
    uint32_t t0, t1;
    double a0, a1, a2;
    unsigned long long mA;
    asm volatile(
        "v_sub_f32 %[t0], %[jf], %[midA]\n\t"
        "v_cmp_lt_f32_e64 %[mA], |%[t0]|, %[halfA]\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        NFA_TABLE_LOOKUP(xA, a0, a1, a2, nucA, idA)
        "s_andn2_b64 vcc, exec, vcc\n\t"                         // the lanes of the Taylor form
        NFA_W_S01
        "v_mul_f64 %[a0], %[a0], %[a1]\n\t"
        NFA_W_S2
        "v_mul_f64 %[a0], %[a0], %[a2]\n\t"
        NFA_TABLE_TAYLOR(xA, a0, a1, a2, ".Lnfa_tps_", mA, "s_cbranch_vccz", "vcc")
        "v_fmac_f64 %[tau], %[wA], %[a0]\n\t"
        "s_mov_b64 exec, -1"
        : [tau] "+v"(tau), [xA] "=&v"(xA), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [mA] "=&s"(mA)
        : [jf] "v"(jf), [xj] "v"(xj), [nucA] "v"(nucA), [idA] "v"(idA), [wA] "v"(wA), [midA] "v"(midA), [halfA] "v"(halfA),
          [nthird] "s"(-(1.0 / 3.0)) NFA_PSEL_OPERAND
        : "vcc", "scc");
}

// a where the lane's bit of `m` is set, b elsewhere (the mask is already a scalar register pair: no compare)
__device__ __forceinline__ double nf_select64(unsigned long long m, double a, double b) {
    int lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(b)), "v"(__double2loint(a)), "s"(m));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(m));
    return __hiloint2double(hi, lo);
}

// 1 - FastExp(tau) for the Tb pass of the table mode (hyperfine.pyx:109-113), tau >= +0 or NaN, every lane of a full
// EXEC: bit for bit 1.0 - nf_fastexp<0, true>(tau).  The Taylor form (fastexp.c:264-270) for every lane first -- in the
// line wings, most rows, every lane is below 2^-5 and that is all there is --; the table product only when a lane of
// the row needs it, ONE select between the two, and the exact zero from 32 on (and for NaN: fastexp.c:272-273) under a
// branch of its own that optically thin rows never take.  (Round 3 went through the general nf_fastexp there: a second
// Taylor evaluation and three more selects per pass.)
__device__ __forceinline__ double one_minus_fastexp_table_row(double tau) {
    const float x = (float)tau;                               // math.pxd:17 narrowing
    const double t = (double)x;
    double r = 1.0 - t * (1.0 / 3.0);                         // x == 0 gives exactly 1
    r = __builtin_fma(t * r, -0.5, 1.0);                      // = 1 - (t r) 0.5: the product with 0.5 is exact
    r = 1.0 - (t * r);
    // u = bits - (122 << 23); the subtraction's carry marks the lanes with l >= 0: the table's range and beyond (one
    // instruction where a subtraction and two compares stood)
    uint32_t u, t1;
    unsigned long long big;
    asm volatile("v_add_co_u32 %[u], vcc, 0xc3000000, %[x]\n\t"
                 "s_mov_b64 %[big], vcc"
                 : [u] "=v"(u), [big] "=s"(big) : [x] "v"(x) : "vcc");
    if (big != 0ull) {
        uint32_t t0;
        double a0, a1, a2;
        asm volatile(NFA_TABLE_ADDR_U(u, x, a0, a1, a2)
                     "s_waitcnt lgkmcnt(1)\n\t"
                     "v_mul_f64 %[a0], %[a0], %[a1]\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_mul_f64 %[a0], %[a0], %[a2]"
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2)
                     : [x] "v"(x), [u] "v"(u) NFA_PSEL_OPERAND);
        r = nf_select64(big, a0, r);
        if (__builtin_amdgcn_sicmp((int32_t)u, (int32_t)(10u << 23), 39 /* sge */) != 0ull) {      // l >= 10: x >= 32, inf, NaN
            asm volatile("" ::: "memory");
            r = (int32_t)u >= (int32_t)(10u << 23) ? 0.0 : r;
        }
    }
    return 1.0 - r;
}

// The body of the likelihood kernel for workgroup `block_id` of a launch (lnl_kernel: the hardware's
// workgroup; point_kernel: the one workgroup walks the few of a single point).  `sm` = the staged
// exponential tables (n_shared doubles at the start of smem), the line tables follow them.
template <int MODE, bool WRITE_SPEC, bool WIDE, int NCOMP, bool DYN = false>
__device__ __forceinline__ void lnl_body(const SpecDev &S, const int *__restrict__ pix, const double *__restrict__ D,
                                         double *__restrict__ part, double *__restrict__ spec_out, long B,
                                         const LnlGeom &G, const double *__restrict__ g_tabs, double *smem,
                                         const double *sm, int n_shared, unsigned block_id,
                                         const BatchGroup *grp = nullptr, long unit_dyn = -1) {
    typedef typename std::conditional<MODE == 2, float, double>::type tau_t;
    constexpr int NC = NCOMP > 0 ? NCOMP : 1;
    // the fast mode's narrow form: at most 26 lines per transition (32-bit line masks), fp32 optical depth, the
    // line step as one instruction block (line_step_fastz)
    constexpr bool FASTN = MODE == 2 && !WIDE;
    // the window test in fp32 (LineRec.mid / half); the wide form may hold spectra of 2^22 channels and more and
    // tests the integers of the window array instead
    constexpr bool FWIN = MODE == 0 || !WIDE;      // (the table mode's line blocks test the window in fp32 whatever the line count)
    const double *g_t0x = g_tabs + SM_T0X, *g_t0y = g_tabs + SM_T0Y;

#ifdef NFA_ABLATE
    const int ablate = G.ablate;      // timing experiments (build with -DNFA_ABLATE)
#else
    const int ablate = 0;
#endif
    int lane_ = threadIdx.x & 63;
    // the queue form runs this body in a loop: what depends only on the lane must not move out of it (the row loop has
    // no registers to spare for values of the unit's prologue)
    if (DYN) asm volatile("" : "+v"(lane_));
    const int lane = lane_, waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncomp = NCOMP > 0 ? NCOMP : S.ncomp, nspec = S.n_spec;
    const int drec = drec_size(ncomp, nspec);

    // One unit per wave, or `split` = 2 or 4 waves per unit (small launches: more, shorter waves than wave
    // slots, so that the hardware places them as slots free up; the waves of a unit share its line table).
    // The rows of a unit form LNL_PARTS interleaved parts (rows h, h + LNL_PARTS, ...: the hyperfine groups
    // sit in a few neighbouring rows, so interleaved parts carry like work); chi^2 is, per lane, the sum of the
    // parts' sums taken in part order, whatever the number of waves that worked on them: the result does
    // not depend on `split` (nor on the size of the batch an item travels in) to the last bit.
    // The grid covers all units (the host keeps B * nspec * split below 2^28).  Waves
    // of a workgroup land on the SIMDs of a CU in order, so the unit -> wave assignment is rotated per
    // workgroup: otherwise one SIMD would only ever see the spectrum with the most hyperfine lines.
    const int split = DYN ? 1 : G.split;                            // (the queue form: one wave per unit)
    const unsigned units = (unsigned)B * (unsigned)nspec;
    const unsigned rot = __builtin_amdgcn_readfirstlane((block_id * 0x9E3779B1u) >> 28);
    const unsigned wsel0 = (unsigned)wave + rot;
    // (no integer division in here: the compiler builds one from two dozen vector instructions, and a wave
    // of the metric shape is only ~2200 long; split is 1, 2 or 4, the waves of a workgroup mostly a power of two)
    // (unit_dyn >= 0: the unit was drawn from the workgroup's queue, lnl_kernel; the wave keeps its own LDS slice)
    const unsigned wsel = DYN ? (unsigned)wave
                        : (waves & (waves - 1)) == 0 ? (wsel0 & (unsigned)(waves - 1)) : wsel0 % (unsigned)waves;
    const int split_log2 = split >> 1;                              // 1, 2, 4 -> 0, 1, 2
    const unsigned upw = (unsigned)waves >> split_log2;             // units per workgroup
    const unsigned ulocal = wsel >> split_log2;
    const int rpart = (int)(wsel & (unsigned)(split - 1));
    const unsigned unit = DYN ? (unsigned)unit_dyn : block_id * upw + ulocal;
    LineRec *w_line = (LineRec *)(smem + n_shared + (size_t)ulocal * G.wave_doubles);
    int2 *w_win = (int2 *)(w_line + (NCOMP > 0 ? NCOMP : S.ncomp) * G.nhf_max);   // the windows [lo, hi) follow the table
    // split > 1: the parts' per-lane sums meet here, [unit of the workgroup][part][lane]
    double *w_part = smem + n_shared + (size_t)upw * G.wave_doubles + (size_t)ulocal * (LNL_PARTS * 64);
    if (unit >= units) {
        if (split > 1) { __syncthreads(); __syncthreads(); }         // the two barriers of the waves at work
        return;
    }
    const unsigned bu = G.inv_nspec ? __umulhi(unit, G.inv_nspec) : unit;      // unit / nspec without a division
    const long b = (long)bu;
    const int s = (int)(unit - bu * (unsigned)nspec);
    const int t = S.trans[s] - 1, N = S.size[s], off = S.off[s];
    const int nhf = __builtin_amdgcn_readfirstlane(c_nhf[t]);
    long p_ix = 0;
    double *so = WRITE_SPEC ? spec_out + b * S.chan_tot + off : nullptr;     // spectra out: the unit's model spectrum
    if (grp) {                                                    // batch kernels: the item's batch of the group has the pixels
        const int c = group_of(*grp, b);
        const int *pp = grp->pix[c];
        if (pp) p_ix = (long)__builtin_amdgcn_readfirstlane(pp[b - c * grp->each]);
        if (WRITE_SPEC && grp->spec[0]) so = grp->spec[c] + (b - c * grp->each) * S.chan_tot + off;     // every batch its own array
    } else if (pix) {
        p_ix = (long)__builtin_amdgcn_readfirstlane(pix[b]);
    }
    const double nu0 = S.rest[s];
    const double *xs = S.xarr + off;
    const k_dbl_p Dk = (k_dbl_p)(D + b * drec);                    // the item's record: scalar loads
    // --- line constants + windows, lanes = (component, line) pairs (hyperfine.pyx:68-91)
    for (int p = lane; p < ncomp * G.nhf_max && !(ablate & 8) && rpart == 0; p += 64) {
        const int c = G.inv_nhf ? (int)__umulhi((unsigned)p, G.inv_nhf) : p, i = p - c * G.nhf_max;     // inv_nhf == 0: one line per component
        // slots beyond the last line: empty windows
        double r_nucen = 0.0, r_idenom = 0.0, r_htau = 0.0;
        int r_lo = 0, r_len = 0, slot = p;
        if (i < nhf) {
            const LineConst lc = nf_line(t, i, D[b * drec + c * 4 + 2], D[b * drec + c * 4 + 1], nu0, S.nu_min[s],
                                         S.nu_chan[s], N, S.r_chan[s]);
            int lo = lc.lo;
            const int hi = lc.hi;
            // Only the first channel of a window can lie beyond the point where FastExp
            // returns exactly 0 (float argument >= 32, fastexp.c:272-273; also NaN): it then
            // adds nothing, so the hot loop starts one channel later and needs no cut-off test.
            if (hi > lo) {
                const double nu = xs[lo] - lc.nucen;
                const float a = (float)(nu * nu * lc.idenom);
                if (!(a < 32.0f)) lo += 1;
            }
            // a centre or a width that is not a finite number gives FastExp NaN or inf at every channel: 0 each
            // time (fastexp.c:272-273), i.e. a line that adds nothing -- an empty window says the same
            if (!(fabs(lc.nucen) < INFINITY) || !(lc.idenom < INFINITY)) lo = hi;
            r_nucen = lc.nucen;
            r_idenom = lc.idenom;
            r_htau = D[b * drec + 4 * ncomp + (c * nspec + s) * DREC_CS + DK_TMAIN] * c_tauw[t][i];
            r_lo = lo;
            r_len = hi > lo ? hi - lo : 0;
            slot = c * G.nhf_max + c_rank[t][i];                 // velocity order: the lines of a row are neighbours
        }
        LineRec rec;
        rec.nucen = r_nucen;
        rec.idenom = r_idenom;
        // fast mode keeps the weight as a float in the low word (no union store: that goes through scratch)
        rec.w = MODE == 2 ? __longlong_as_double((long long)__float_as_uint((float)r_htau)) : r_htau;
        rec.mid = (float)r_lo + 0.5f * (float)(r_len - 1);
        rec.half = 0.5f * (float)r_len;                          // an empty window: half = 0, no channel passes
        w_line[slot] = rec;
        w_win[slot] = make_int2(r_lo, r_lo + r_len);
    }
    if (split > 1) __syncthreads(); else wave_lds_sync();
    // windows [lo, hi) of the lines of each component, lane = line (an empty window is [0, 0):
    // it fails `hi > r0` for every row), and the component's constants of the Tb pass
    int wlo[NC], whi[NC];
    // fast mode, two components (at most 26 lines each): both components' windows in one register pair, lanes
    // 0..31 the first component's lines, lanes 32..63 the second's -- two compares per row instead of four
    // (a compare costs as much as an fp64 operation, profiles/r02/ubench_valu.txt)
    // (round 5: the table mode too, where no transition has more than 26 lines -- WIDE is then the host's word for "more" --:
    // two compares per row less of the ~80 a row costs)
    constexpr bool PACK2 = (MODE == 2 || MODE == 0) && !WIDE && NCOMP == 2;
    constexpr bool HOISTX = false;                             // (round 4: the exact modes' Tb constants hoisted per unit; round 5: the cell form below needs none)
    int wlo2 = 0, whi2 = 0;
    double ck_a0x[NC], ck_b0x[NC];
    // how a component's Tb pass goes, one scalar register formed once per unit (read where it is used, the model is a
    // scalar load and a wait in the dependent chain of every (row, component); as compares of the record's doubles the
    // classes are two register pairs per component)
    int ck_cls[NC];
    double ck_kind[NC];                                            // (fast mode: left to the compiler, which keeps the classes as lane masks)
    // exact modes: the constants of a component's Tb pass (excitation temperature, its reciprocal, the band's cell of
    // the 1/(e^x - 1) table) are read once per unit -- read where they are used, each is a scalar load and a wait in
    // the dependent chain of every (row, component)
    double cx_tex[NC], cx_rtex[NC], cx_xkind[NC], cx_xs[NC], cx_xlo[NC], cx_ylo[NC];
    // the LDS address of each component's line table as a per-lane value, formed once: a line's record address
    // is then ONE vector shift-add of the scalar line index (left to itself the compiler forms it with two
    // scalar instructions and a move per step; the scalar unit is the shared resource, see the header)
    typedef const __attribute__((address_space(3))) char *lds_char_p;
    unsigned lbase_c[NC];
    // window [lo, hi) of line `l` of component `c` (0, 0 for l beyond the table)
    auto window_of = [&](int c, int l, int &lo, int &hi) {
        const int k = c * G.nhf_max + (l < G.nhf_max ? l : 0);
        const int2 w = w_win[k];
        lo = w.x; hi = w.y;
        if (!(l < G.nhf_max) || (ablate & 8)) { lo = 0; hi = 0; }
    };
    // 3: no Tb pass (gaussian.pyx:50); 1: y(T0) one table cell over the band; 2: two cells; 0: the general form
    auto tb_class = [&](double kind) {
        return ((ablate & 1) || S.model == NFA_MODEL_GAUSSIAN) ? 3 : kind == 1.0 ? 1 : kind != 0.0 ? 2 : 0;
    };
    if (NCOMP > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!PACK2) window_of(c, lane, wlo[c], whi[c]);
            lbase_c[c] = (unsigned)(uintptr_t)(lds_char_p)(w_line + c * G.nhf_max);
            asm volatile("" : "+v"(lbase_c[c]));
            const int dko = 4 * ncomp + (c * nspec + s) * DREC_CS;
            ck_kind[c] = Dk[dko + DK_KIND]; ck_a0x[c] = Dk[dko + DK_A0X]; ck_b0x[c] = Dk[dko + DK_B0X];
            if (MODE != 2) {
                ck_cls[c] = __builtin_amdgcn_readfirstlane(tb_class(ck_kind[c]));
                asm volatile("" : "+s"(ck_cls[c]));
            }
            if (HOISTX) {
                cx_tex[c] = Dk[c * 4]; cx_rtex[c] = Dk[c * 4 + 3];
                cx_xkind[c] = Dk[dko + DK_XKIND]; cx_xs[c] = Dk[dko + DK_XS]; cx_xlo[c] = Dk[dko + DK_XLO]; cx_ylo[c] = Dk[dko + DK_YLO];
            }
        }
    }
    if (PACK2) window_of(lane >> 5, lane & 31, wlo2, whi2);
    // --- rows of 64 channels: tau profile, Tb, chi^2 (hyperfine.pyx:93-113, core.pyx:522-530)
    const double *t0s = S.t0 + off, *tbgs = S.tbg + off, *p3s = S.t0tbg + off;
    const double *ds = S.data + p_ix * S.chan_tot + off;
    // chi^2 = sum (d - pred)^2 = sum d^2 + sum pred (pred - 2 d): the first sum is a constant of the (pixel,
    // spectrum), formed once (SpecDev.totsq); the second has a term only where the model is not zero -- rows no
    // line window touches are never read, and a lane outside every window adds pred (...) = 0 to its row's sum
    double acc = 0.0;
    const int n_rows = (ablate & 4) ? 0 : (N + 63) >> 6;
    const int parts_per_wave = LNL_PARTS >> split_log2;
    double tot = 0.0;
    constexpr bool SPEC_DEFER = WRITE_SPEC && MODE == 2;
    double pend_v = 0.0;                                           // spectra out: the row whose store is still to be issued
    int pend_j = -1;
    for (int hp = 0; hp < parts_per_wave; ++hp) {
    const int h = rpart * parts_per_wave + hp;                 // part h = rows h, h + LNL_PARTS, h + 2 LNL_PARTS, ...
    acc = 0.0;
    for (int row = h; row < n_rows; row += LNL_PARTS) {
        const int r0 = row << 6;
        const int j = r0 + lane;
        const float jf = (float)j;                                 // FASTN: the window test runs in fp32
        // lines of each component that touch this row
        unsigned long long hitm[NC];
        bool any = false;
        if (PACK2) {
            const unsigned long long hit = lanes_lt(wlo2, r0 + 64) & lanes_gt(whi2, r0);
            // (the halves as 32-bit scalars of their own: the compiler tests "the upper half is not zero" as a 64-bit
            // VECTOR compare of the pair against a constant otherwise)
            unsigned h0 = (unsigned)hit, h1 = (unsigned)(hit >> 32);
            asm volatile("" : "+s"(h0), "+s"(h1));
            hitm[0] = h0;
            hitm[NC - 1] = h1;
            any = hit != 0ull;
        } else if (NCOMP > 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                hitm[c] = lanes_lt(wlo[c], r0 + 64) & lanes_gt(whi[c], r0);
                any = any || hitm[c] != 0ull;
            }
        } else {
            for (int c = 0; c < ncomp && !any; ++c) {
                int lo, hi;
                window_of(c, lane, lo, hi);
                any = __builtin_amdgcn_ballot_w64((lo < r0 + 64) & (hi > r0) & (hi > lo)) != 0ull;
            }
        }
        // Spectra out (main.py:1106-1113, 1182-1188): a row no line window touches is zeros, written without reading the row
        // (write-once data, past the caches: non-temporal).
        if (WRITE_SPEC && !any) {
            if (j < N) __builtin_nontemporal_store(0.0, so + j);
            continue;
        }
        if (any) {
            const bool valid = j < N;
            const unsigned jo = (unsigned)(valid ? j : N - 1) * 8u;           // byte offset of the lane's channel
            const double xj = *(const double *)((const char *)xs + jo);
            const double dj = *(const double *)((const char *)ds + jo);
            double p3;
            // T0 tbg of the channel; T0 and tbg themselves are read inside the rare pass that needs them (y(T0) not a single
            // table cell): carried through the row as "maybe loaded" values they cost two register copies per row
            p3 = *(const double *)((const char *)p3s + jo);
            // Spectra out: the PREVIOUS row's store is issued here, behind this row's loads.  Vector-memory operations of a
            // wave complete in order (one counter, vmcnt): issued at the end of its own row, a store stood between the next
            // row's loads and their first use, and every row waited for the acknowledgement of 512 bytes written to HBM
            // (round 4: 43 us per 4096 rows with the store against 28 without).  Behind the loads it has the whole row to
            // complete.
            // (Fast mode; the table mode's sixteen-wave workgroups have no registers for the held row -- 64 for two of them
            // per CU -- and store where the row ends.)
            if (SPEC_DEFER) {
                asm volatile("" ::: "memory");                        // (the loads above stay above, the store below)
                if (pend_j >= 0) __builtin_nontemporal_store(pend_v, so + pend_j);
                asm volatile("" ::: "memory");
            }
            double pred = 0.0;
            // one component: the lines in `mask` add their optical depths, then the Tb pass
            auto component = [&](int c, unsigned long long mask, double kind, int cls_unit, double a0x, double b0x) {
                unsigned lbase;
                if (NCOMP > 0) {
                    lbase = lbase_c[c];
                } else {
                    lbase = (unsigned)(uintptr_t)(lds_char_p)(w_line + c * G.nhf_max);
                    asm volatile("" : "+v"(lbase));
                }
                tau_t tau = 0;
                double td = 0.0;                                      // WIDE: fp64 running sum
                typedef double v2d __attribute__((ext_vector_type(2)));
                typedef int v4i __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(3))) v2d *lds_v2d_p;
                typedef const __attribute__((address_space(3))) v4i *lds_v4i_p;
                // one line x row step with the record (nucen, k | w, mid, half) already read; window index `wi` (WIDE)
                auto step = [&](const v2d ab, const v4i hw, int wi) {
                    if constexpr (FASTN) {
                        line_step_fastz(tau, jf, xj, ab.x, ab.y, __int_as_float(hw.x), __int_as_float(hw.z), __int_as_float(hw.w));
                    } else {
                        // the whole record is read before the window test (left alone the compiler reads the window, tests,
                        // and only then reads the rest: one more trip to LDS in the dependent chain of every step)
                        double nucen = ab.x, idenom = ab.y;
                        asm volatile("" : "+v"(nucen), "+v"(idenom));
                        bool inside;
                        if constexpr (FWIN) {
                            inside = __builtin_fabsf(jf - __int_as_float(hw.z)) < __int_as_float(hw.w);
                        } else {
                            const int2 w = w_win[wi];
                            inside = (unsigned)(j - w.x) < (unsigned)(w.y - w.x);
                        }
                        if (inside) {                                     // the window is the EXEC mask
                            asm volatile("" ::: "memory");                // keep it a branch (no if-conversion)
                            const double nu = xj - nucen;
                            const double tau_exp = nu * nu * idenom;      // hyperfine.pyx:94
                            if constexpr (MODE == 2) {
                                const float e = exp_neg_core_f32((float)tau_exp);          // math.pxd:17 narrowing
                                td = __builtin_fma((double)__int_as_float(hw.x), (double)e, td);
                            } else {
                                const double e = nf_fastexp<MODE, true, true>(tau_exp, sm);
                                tau = __builtin_fma(__hiloint2double(hw.y, hw.x), e, (double)tau);
                            }
                        }
                    }
                };
                auto rec_ab = [&](unsigned a) { return *(lds_v2d_p)(uintptr_t)a; };
                auto rec_hw = [&](unsigned a) { return *(lds_v4i_p)(uintptr_t)(a + 16); };
                if (ablate & 2) { tau = (tau_t)(1e-3 * (lane + 1)); }
                else {
                    // The table is in velocity order, so the lines of this row are the run from the lowest to the highest
                    // set bit of the mask (a line inside the run whose window misses the row -- widths differ by 1e-4 from
                    // line to line -- finds no lane in its window and adds nothing).  The run is walked two lines at a
                    // time: the records of a pair are four reads off one address, which advances once per pair.
                    int first, n;
                    if (FASTN || (MODE == 0 && !WIDE) || nhf <= 32) { // every NH3 transition: 32-bit mask arithmetic
                        const unsigned m = (unsigned)mask;
                        first = __builtin_ctz(m);
                        n = 32 - __builtin_clz(m) - first;
                    } else {
                        first = __builtin_ctzll(mask);
                        n = 64 - __builtin_clzll(mask) - first;
                    }
                    unsigned va = lbase + ((unsigned)first << 5);
                    asm volatile("" : "+v"(va));
                    int wi = c * G.nhf_max + first;
                    if (n & 1) {
                        const v2d ab = rec_ab(va);
                        if constexpr (MODE == 0 && FWIN) {
                            const v2d wm = rec_ab(va + 16);
                            line_single_table(tau, jf, xj, ab.x, ab.y, wm.x, __int_as_float(__double2loint(wm.y)),
                                              __int_as_float(__double2hiint(wm.y)));
                        } else {
                            const v4i hw = rec_hw(va);
                            step(ab, hw, wi);
                        }
                        va += 32;
                        asm volatile("" : "+v"(va));
                        wi += 1;
                        n -= 1;
                    }
                    while (n) {                                        // both records of a pair are read before the first step
                        const v2d ab0 = rec_ab(va), ab1 = rec_ab(va + 32);
                        if constexpr (MODE == 0 && FWIN) {
                            // (w | mid, half) read as two doubles: the weight is then a register pair as it stands
                            const v2d wm0 = rec_ab(va + 16), wm1 = rec_ab(va + 48);
                            line_pair_table(tau, jf, xj, ab0.x, ab0.y, wm0.x, __int_as_float(__double2loint(wm0.y)),
                                            __int_as_float(__double2hiint(wm0.y)), ab1.x, ab1.y, wm1.x,
                                            __int_as_float(__double2loint(wm1.y)), __int_as_float(__double2hiint(wm1.y)));
                        } else {
                            const v4i hw0 = rec_hw(va), hw1 = rec_hw(va + 32);
                            step(ab0, hw0, wi);
                            step(ab1, hw1, wi + 1);
                        }
                        va += 64;
                        asm volatile("" : "+v"(va));
                        wi += 2;
                        n -= 2;
                    }
                }
                if (MODE == 2 && WIDE) tau = (tau_t)td;
                // hyperfine.pyx:104-105: channels with tau == 0 are skipped (lanes beyond the last channel
                // are in no window: tau == 0 there too)
                const unsigned long long livem = MODE == 2 ? __builtin_amdgcn_fcmpf((float)tau, 0.0f, NF_FCMP_UNE)
                                                           : __builtin_amdgcn_fcmp((double)tau, 0.0, NF_FCMP_UNE);
                if (livem == 0ull) return;
                // (the class is tested through a copy the compiler cannot see through: left to itself it makes one switch of
                // the chain below and the structured form of that costs a dozen scalar instructions on the way to the usual case)
                // (fast mode: the class from the record's double where it is used, as round 4 had it -- one scalar register per
                // component formed per unit measured 1.5 % slower there)
                const int cls = MODE == 2 ? tb_class(kind) : cls_unit;
                int cls_hot = __builtin_amdgcn_readfirstlane(cls);
                if (MODE != 2) asm volatile("" : "+s"(cls_hot));
                if (cls_hot == 1) {
                    // The band lies in ONE cell of the 1/(e^x - 1) table (hyperfine.pyx:23-45; the usual case, first in the
                    // chain: one scalar compare and a branch between the optical depth and the pass): the cell's
                    // straight line in x = T0 / tex is a straight line in the channel's frequency, and
                    //     T0 (y - tbg) = B0x x^2 + A0x x - T0 tbg          (Horner: no x^2 per row)
                    // with the two coefficients formed once per (item, component, spectrum) by the set-up stage -- two
                    // fused multiply-adds per channel where the reference's order of operations (division, cell, slope,
                    // difference, product: eight in the exact modes, with the quotient by Markstein's step) rounds
                    // differently in the sixteenth digit; the reference itself is built with -ffast-math and differs from
                    // its own strict build by 1e-11.  FastExp of the optical depth, the factor with the table
                    // indices, is the reference's to the bit in the table mode.  Lanes with tau == 0 (skipped by the
                    // reference, hyperfine.pyx:104-105) get g * (1 - 1) = +-0: the sum needs no per-lane select.
                    const double g = __builtin_fma(xj, __builtin_fma(b0x, xj, a0x), -p3);
                    if (MODE == 2) pred = __builtin_fma(g, one_minus_fastexp_f32((float)tau, livem), pred);
                    else pred = __builtin_fma(g, MODE == 0 ? one_minus_fastexp_table_row((double)tau) : nf_one_minus_fastexp_row<MODE>((double)tau, sm), pred);
                    return;
                }
                if (cls == 3) {                                       // gaussian.pyx:50: pred += peak * e
                    pred += (double)tau;                              // tau == 0 adds nothing
                    return;
                }
                const int dko = 4 * ncomp + (c * nspec + s) * DREC_CS;
                if (MODE == 2) {
                    double g;                                         // T0 (y(T0) - tbg)
                    if (cls == 2) {
                        unsigned jr = jo;
                        asm volatile("" : "+v"(jr));                  // the two addresses are formed here, not in every row's head
                        const double T0 = *(const double *)((const char *)t0s + jr), tbg = *(const double *)((const char *)tbgs + jr);
                        const bool up = !(T0 < Dk[dko + DK_SPLIT]);
                        const double dT = T0 - Dk[dko + DK_M];
                        const double ya = up ? Dk[dko + DK_A1] : Dk[dko + DK_A0];
                        const double yb = up ? Dk[dko + DK_B1] : Dk[dko + DK_B0];
                        const double y = __builtin_fma(__builtin_fma(Dk[dko + DK_Q], dT, yb), dT, ya);
                        g = T0 * (y - tbg);
                    } else {
                        unsigned jr = jo;
                        asm volatile("" : "+v"(jr));
                        const double T0 = *(const double *)((const char *)t0s + jr), tbg = *(const double *)((const char *)tbgs + jr);
                        const double y = nf_iemtex(T0 / Dk[c * 4], g_t0x, g_t0y, S.t0_xmin, S.t0_xmax, S.t0_inv_dx);
                        g = T0 * (y - tbg);
                    }
                    pred = __builtin_fma(g, one_minus_fastexp_f32((float)tau, livem), pred);
                } else {
                    unsigned jr = jo;
                    asm volatile("" : "+v"(jr));                      // the two addresses are formed here, not in every row's head
                    const double T0 = *(const double *)((const char *)t0s + jr), tbg = *(const double *)((const char *)tbgs + jr);
                    // x = T0 / tex (hyperfine.pyx:107) without a division per channel: from the correctly rounded
                    // reciprocal the set-up stage left in the record, q = T0 r, e = T0 - tex q (exact, fused),
                    // x = q + e r is the correctly rounded quotient (Markstein's final step; the one exception,
                    // a tex whose mantissa is all ones, has probability 2^-52): the same bits as the division
                    // (400 M random pairs on the host: 0 differences).  A tex that is not an ordinary positive number
                    // takes the division.
                    const double tex = (HOISTX && NCOMP > 0) ? cx_tex[c] : Dk[c * 4], rtex = (HOISTX && NCOMP > 0) ? cx_rtex[c] : Dk[c * 4 + 3];
                    double x;
                    if (tex > 1e-100 && tex < 1e100) {
                        const double q = T0 * rtex;
                        x = __builtin_fma(__builtin_fma(-tex, q, T0), rtex, q);
                    } else {
                        x = T0 / tex;
                    }
                    // the reference's order: pred[i] += T0 * (y - tbg) * (1 - FastExp(tau)), only where tau != 0
                    // (tau is a sum of non-negative terms, or NaN)
                    if (((HOISTX && NCOMP > 0) ? cx_xkind[c] : Dk[dko + DK_XKIND]) == 1.0) {
                        // the band's table cell: y is finite, and where tau == 0 the last factor is exactly 1 - 1 = 0 --
                        // the product adds +0 there and needs no select
                        const double y = (HOISTX && NCOMP > 0) ? cx_xs[c] * (x - cx_xlo[c]) + cx_ylo[c]
                                                   : Dk[dko + DK_XS] * (x - Dk[dko + DK_XLO]) + Dk[dko + DK_YLO];
                        pred += (T0 * (y - tbg)) * (MODE == 0 ? one_minus_fastexp_table_row((double)tau) : nf_one_minus_fastexp_row<MODE>((double)tau, sm));
                    } else {
                        const double y = nf_iemtex(x, g_t0x, g_t0y, S.t0_xmin, S.t0_xmax, S.t0_inv_dx);
                        const double tb = (T0 * (y - tbg)) * (MODE == 0 ? one_minus_fastexp_table_row((double)tau) : nf_one_minus_fastexp_row<MODE>((double)tau, sm));
                        pred += !(tau == 0) ? tb : 0.0;
                    }
                }
            };
            if (NCOMP > 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (hitm[c] != 0ull) component(c, hitm[c], ck_kind[c], MODE != 2 ? ck_cls[c] : 0, ck_a0x[c], ck_b0x[c]);
            } else {
                for (int c = 0; c < ncomp; ++c) {
                    int lo, hi;
                    window_of(c, lane, lo, hi);
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64((lo < r0 + 64) & (hi > r0) & (hi > lo));
                    if (mask == 0ull) continue;
                    const int dko = 4 * ncomp + (c * nspec + s) * DREC_CS;
                    component(c, mask, Dk[dko + DK_KIND], tb_class(Dk[dko + DK_KIND]), Dk[dko + DK_A0X], Dk[dko + DK_B0X]);
                }
            }
            if (SPEC_DEFER) { pend_v = pred; pend_j = valid ? j : -1; }
            else if (WRITE_SPEC) { if (valid) __builtin_nontemporal_store(pred, so + j); }
            if (any) acc = __builtin_fma(pred, __builtin_fma(-2.0, dj, pred), acc);       // lanes beyond N: pred = 0
        }
    }
    if (split == 1) tot += acc; else w_part[h * 64 + lane] = acc;
    }
    if (SPEC_DEFER && pend_j >= 0) __builtin_nontemporal_store(pend_v, so + pend_j);
    if (split > 1) {
        __syncthreads();
        if (rpart != 0) return;
        for (int h = 0; h < LNL_PARTS; ++h) tot += w_part[h * 64 + lane];
    }
    tot = wave_sum(tot);
    // sum of squared deviations of the unit; lnl_sum_kernel scales and adds
    if (lane == 0 && part) part[unit] = S.totsq[p_ix * nspec + s] + tot;
}

// WIDE (fast mode only): the spectra set holds a transition with more than 26 lines (N2H+)
template <int MODE, bool WRITE_SPEC, bool WIDE, int NCOMP>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80)))
lnl_kernel(SpecDev S, BatchGroup grp, const double *__restrict__ D, double *__restrict__ part,
           double *__restrict__ spec_out, long B, LnlGeom G, const double *__restrict__ g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int n_shared = 0;
    const double *sm = smem;
    if (MODE != 2) sm = stage_exp_tables<MODE == 2 ? 1 : MODE>(smem, g_tabs, &n_shared);   // fast: no tables
    lnl_body<MODE, WRITE_SPEC, WIDE, NCOMP>(S, nullptr, D, part, spec_out, B, G, g_tabs, smem, sm, n_shared, blockIdx.x, &grp);
}

// lnl_kernel held to 64 vector registers (eight waves per SIMD = two sixteen-wave workgroups of the table mode per CU): the
// table mode with spectra out asks for 66 left alone.
template <int MODE, bool WRITE_SPEC, bool WIDE, int NCOMP>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) __attribute__((amdgpu_waves_per_eu(8, 8)))
lnl_kernel_w8(SpecDev S, BatchGroup grp, const double *__restrict__ D, double *__restrict__ part,
              double *__restrict__ spec_out, long B, LnlGeom G, const double *__restrict__ g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int n_shared = 0;
    const double *sm = smem;
    if (MODE != 2) sm = stage_exp_tables<MODE == 2 ? 1 : MODE>(smem, g_tabs, &n_shared);
    lnl_body<MODE, WRITE_SPEC, WIDE, NCOMP>(S, nullptr, D, part, spec_out, B, G, g_tabs, smem, sm, n_shared, blockIdx.x, &grp);
}

// Table mode, one wave per unit, the units drawn from a queue.  A workgroup of the table mode is sixteen waves behind one
// 51 KB copy of the product tables, and in lnl_kernel the wave slots of the waves that are done stay empty until the
// longest of the sixteen is: units differ by the widths of their lines, and at the metric's shape 4.5 of 8 waves per
// SIMD were resident on average (profiles/r04/pmc_lnl_table.json).  Here the launch is as many workgroups as the device
// holds at once (two per CU), each stages the tables once, and every wave takes units one at a time until none is
// left.  The launch is cut into chunks of NFA_QUEUE_CHUNK units that a workgroup draws from the launch's counter in global memory (one returning
// atomic per chunk: a unit per wave and draw there stands in line, ~11 ns per atomic on one word, profiles/r05) and
// hands out to its waves from a word in LDS: {first unit of the chunk, units handed out} as ONE 64-bit word, so that a
// wave's returning add is a snapshot of both.  The wave whose add finds the chunk just used up draws the next one and
// rewrites the word; waves that come while it does wait on the word.  Every unit is computed exactly once; which wave
// computes it does not enter its result.  The last workgroup to leave zeroes the launch's counters for the next launch
// on this stream lane.
#define NFA_QUEUE_CHUNK  16u
#define NFA_QUEUE_END    0xffffffffu
#define NFA_QUEUE_STRIDE 32                                   // words between the two counters: a 128-byte line each
#define NFA_QUEUE_WORDS  (2 * NFA_QUEUE_STRIDE)
struct QueueArgs { SpecDev S; BatchGroup grp; const double *D; double *part; double *spec_out; long B; LnlGeom G; const double *g_tabs; };
// (the argument segment places every argument at the next multiple of its alignment: 8 for all of these, so the struct's
// layout is the segment's -- .offset of the kernel's .args in the code object: 0, 896, 1008, 1016, 1024, 1032, 1040, 1080)
static_assert(sizeof(SpecDev) % 8 == 0 && sizeof(BatchGroup) % 8 == 0 && sizeof(LnlGeom) % 8 == 0 && alignof(SpecDev) == 8 &&
              alignof(BatchGroup) == 8 && alignof(LnlGeom) == 8, "QueueArgs must mirror the kernel argument segment");
template <bool WRITE_SPEC, int NCOMP>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) __attribute__((amdgpu_waves_per_eu(8, 8)))
lnl_kernel_queue(SpecDev S, BatchGroup grp, const double *__restrict__ D, double *__restrict__ part,
                 double *__restrict__ spec_out, long B, LnlGeom G, const double *__restrict__ g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int n_shared = 0;
    // the workgroup's queue word and its count of waves that have left, behind the waves' line tables (the barrier of
    // the staging publishes them)
    unsigned long long *const wq = (unsigned long long *)(smem + (SM_END_TABLE - SM_EXP2) + (size_t)(blockDim.x >> 6) * G.wave_doubles);
    unsigned *const left = (unsigned *)(wq + 1);
    if (threadIdx.x == 0) { *wq = (unsigned long long)NFA_QUEUE_CHUNK; *left = 0u; }      // "used up": the first comer draws
    const double *sm = stage_exp_tables<0>(smem, g_tabs, &n_shared);
    const unsigned units = (unsigned)B * (unsigned)S.n_spec;
    // (every unit comes from the queue, a wave's first one too: a workgroup that becomes resident late -- the launch
    // shares the device with another lane's -- finds what is left, or nothing, and does not hold units of its own)
    const unsigned queued = units;
    const int lane_q = threadIdx.x & 63;
    unsigned *const q = G.queue;
    unsigned unit = 0;
    bool have = false, first_draw = true;
#ifdef NFA_TEST_HOOKS
    int n_rec = 0;
#endif
#pragma nounroll
    while (have || first_draw) {
      if (!first_draw) {
        const unsigned u_item = unit;
#ifdef NFA_TEST_HOOKS
        const unsigned long long t_start = wall_clock64();
#endif
        // The kernel's arguments are read where the unit needs them, through the scalar cache, from the kernel's
        // argument segment: through an address the compiler cannot see through, so that it does not
        // load the unit-independent fields ONCE before the loop -- they then live in scalar registers across the whole
        // body, overflow the 80 the launch may have, and come back from lanes of a vector register with ~85 v_readlane
        // per unit, vector instructions on the pipe that bounds the kernel.
        // (QueueArgs: the kernel's arguments as the segment holds them -- every one of them 8-byte aligned.)
        typedef const __attribute__((address_space(4))) QueueArgs *k_args_p;
        k_args_p A = (k_args_p)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(A));
        lnl_body<0, WRITE_SPEC, false, NCOMP, true>(*(const SpecDev *)&A->S, nullptr, A->D, A->part, A->spec_out, A->B, *(const LnlGeom *)&A->G,
                                                    A->g_tabs, smem, sm, n_shared, blockIdx.x, (const BatchGroup *)&A->grp, (long)u_item);
#ifdef NFA_TEST_HOOKS
        if (G.trace && lane_q == 0 && n_rec < 8) {
            unsigned long long *t = G.trace + ((size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + n_rec) * 4;
            t[0] = t_start; t[1] = wall_clock64(); t[2] = u_item; t[3] = unit;
        }
        n_rec += 1;
#endif
      }
        first_draw = false;
        have = false;
#pragma nounroll
        for (;;) {
            unsigned long long old = 0;
            if (lane_q == 0) old = __hip_atomic_fetch_add(wq, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned first = __builtin_amdgcn_readfirstlane((unsigned)(old >> 32));
            const unsigned k = __builtin_amdgcn_readfirstlane((unsigned)old);
            if (first == NFA_QUEUE_END) break;
            if (k < NFA_QUEUE_CHUNK) {
                if (first + k >= queued) continue;                 // beyond the launch's last unit: the next draw ends it
                unit = first + k; have = true;
                break;
            }
            if (k == NFA_QUEUE_CHUNK) {                              // this wave draws the workgroup's next chunk
                unsigned g = 0;
                if (lane_q == 0) g = __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned nf = __builtin_amdgcn_readfirstlane(g) * NFA_QUEUE_CHUNK;
                have = nf < queued;                                  // and takes the chunk's first unit itself
                if (lane_q == 0)
                    __hip_atomic_store(wq, have ? ((unsigned long long)nf << 32) | 1ull : (unsigned long long)NFA_QUEUE_END << 32,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                unit = nf;
                break;
            }
            __builtin_amdgcn_s_sleep(8);                             // a chunk is being drawn
        }
    }
    if (lane_q == 0) {
        const unsigned w = blockDim.x >> 6;
        if (__hip_atomic_fetch_add(left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == w - 1) {
            unsigned *const out = q + NFA_QUEUE_STRIDE;
            if (__hip_atomic_fetch_add(out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// chi^2 parts of one item -> its log-likelihood: the sum over the spectra, in order (ammonia.pyx:425-432), of
// -chi2_s / (2 noise_s^2) (core.pyx:530)
__device__ __forceinline__ double lnl_of_item(const double *__restrict__ part, const double *__restrict__ noise,
                                              long p_ix, long b, int nspec) {
    double tot = 0.0;
    for (int s = 0; s < nspec; ++s) {
        const double sigma = noise[p_ix * nspec + s];
        tot += -part[b * nspec + s] / (2 * (sigma * sigma));
    }
    return tot;
}

// lnL of the items of a batch, lanes = items (the division happens here instead of once per likelihood wave)
__global__ void lnl_sum_kernel(const double *__restrict__ part, const double *__restrict__ noise,
                               BatchGroup grp, long B, int nspec) {
    __builtin_amdgcn_s_setprio(3);
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int c = group_of(grp, b);
    const long row = b - c * grp.each;
    const int *pix = grp.pix[c];
    grp.lnL[c][row] = lnl_of_item(part, noise, pix ? (long)pix[row] : 0, b, nspec);
}

#include "nfa_setup.h"

// ---------------------------------------------------------------------------
//  set-up kernels
// ---------------------------------------------------------------------------
__global__ void prep_kernel(const double *__restrict__ x, double *__restrict__ t0,
                            double *__restrict__ tbg, double *__restrict__ t0tbg, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double T0 = NFA_H * x[i] / NFA_KB;                  // hyperfine.pyx:106
    t0[i] = T0;
    const double bg = 1.0 / expm1(T0 / NFA_TCMB);             // ammonia.pyx:274-277
    tbg[i] = bg;
    t0tbg[i] = T0 * bg;
}

// rowsq[pix][row_off[s] + r] = sum over the channels of row r of spectrum s of data^2: the chi^2
// term of a row without any line window (pred == 0 there, core.pyx:522-530).  One wave per row.
__global__ void rowsq_kernel(SpecDev S, long pix0, long n_pix, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n_pix * S.rows_tot) return;
    const long p = pix0 + w / S.rows_tot;
    const int rr = (int)(w % S.rows_tot);
    int s = 0;
    while (s + 1 < S.n_spec && rr >= S.row_off[s + 1]) ++s;
    const int j = (rr - S.row_off[s]) * 64 + lane;
    double v = 0.0;
    if (j < S.size[s]) { const double d = S.data[p * S.chan_tot + S.off[s] + j]; v = d * d; }
    v = wave_sum(v);
    if (lane == 0) out[p * S.rows_tot + rr] = v;
}

// totsq[pix][spec] = sum over the rows of the spectrum, in order, of rowsq: the constant part of chi^2
__global__ void totsq_kernel(SpecDev S, long pix0, long n_pix, const double *__restrict__ rowsq, double *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * S.n_spec) return;
    const long p = pix0 + i / S.n_spec;
    const int s = (int)(i % S.n_spec);
    const int n_rows = (S.size[s] + 63) >> 6;
    const double *r = rowsq + p * S.rows_tot + S.row_off[s];
    double v = 0.0;
    for (int k = 0; k < n_rows; ++k) v += r[k];
    out[p * S.n_spec + s] = v;
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
