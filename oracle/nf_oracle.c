/* nf_oracle.c -- CPU restatement of the nestfit NH3 log-likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * engine in nestfit_amd/csrc; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped product never calls
 * into oracle/.
 *
 * It is a fresh, scalar, plain-C restatement written from the algorithm of the
 * reference (autocorr/nestfit v0.2).  Each function cites the reference
 * file:line it follows.  Operation order follows the reference expression by
 * expression so that, built with strict IEEE flags (-O2 -ffp-contract=off),
 * integer indices / window supports are identical and floating point agrees
 * with the reference's own build-flag noise (~1e-11 rel).
 *
 * PARITY PIN (see DESIGN.md "Oracle"):
 *   * FastExp: checked bit-for-bit against oracle/_ref/libfastexp_ref.so, which
 *     is compiled from the reference's own nestfit/core/fastexp.c (tests/).
 *   * iemtex / partition / spectra / lnL / prior transform: checked against
 *     the known answers captured from the compiled reference in SURVEY.md 8c
 *     (tests/golden/survey_kat.json) and the reference's own in-module tests.
 *   * Anything outside those vectors (trans 4..9, cold/lte, CenSep priors):
 *     parity unpinned beyond a line-by-line restatement.
 */
#include "nf_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NFA_DATA_QUAL static const
#include "../nestfit_amd/csrc/nh3_data.h"
#include "../nestfit_amd/csrc/n2hp_data.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- *
 *  FastExp  (reference: nestfit/core/fastexp.c:177-231 table fill,
 *            :234-283 evaluation; called as fast_expn, core/math.pxd:17)
 *
 *  exp(-x) for a float32 argument.  The 23 mantissa bits are split in three
 *  fields (7 | 8 | 8 bits).  For binary exponent e = l-5, l in 0..9:
 *     x = (128+j0) 2^(l-12) + j1 2^(l-20) + j2 2^(l-28)
 *  so exp(-x) is the product of three tabulated exponentials.  Below 2^-5 a
 *  third-order Horner series is used, at/above 2^5 the result is exactly 0.
 * ------------------------------------------------------------------------- */
static double g_tab_a[10][128];   /* exp(-(128+j) 2^(l-12)) */
static double g_tab_b[10][256];   /* exp(-j 2^(l-20))       */
static double g_tab_c[10][256];   /* exp(-j 2^(l-28))       */
static double g_inv_i[4];
static int    g_fastexp_ready = 0;

void nfo_fastexp_init(void) {
    /* The reference evaluates libm exp() on float bit patterns; every table
     * argument below is exactly representable, so building it from ldexp is
     * the same number (fastexp.c:203-226). */
    for (int l = 0; l < 10; ++l) {
        for (int j = 0; j < 128; ++j)
            g_tab_a[l][j] = exp(-ldexp((double)(128 + j), l - 12));
        for (int j = 0; j < 256; ++j) {
            g_tab_b[l][j] = exp(-ldexp((double)j, l - 20));
            g_tab_c[l][j] = exp(-ldexp((double)j, l - 28));
        }
    }
    g_inv_i[0] = 0.0;
    for (int i = 1; i <= 3; ++i) g_inv_i[i] = 1.0 / (1.0 * i);
    g_fastexp_ready = 1;
}

void nfo_fastexp_indices(float x, int *l, int *j0, int *j1, int *j2) {
    uint32_t bits;
    memcpy(&bits, &x, sizeof bits);
    *l  = (int)((bits & 0x7f800000u) >> 23) - 122;   /* fastexp.c:262 */
    *j0 = (int)((bits & 0x007f0000u) >> 16);         /* fastexp.c:276 */
    *j1 = (int)((bits & 0x0000ff00u) >> 8);
    *j2 = (int)(bits & 0x000000ffu);
}

double nfo_fastexp(float x) {
    if (!g_fastexp_ready) nfo_fastexp_init();
    if (x < 0.0f) return exp(-(double)x);            /* fastexp.c:259 */
    if (x == 0.0f) return 1.0;                       /* fastexp.c:260 */
    int l, j0, j1, j2;
    nfo_fastexp_indices(x, &l, &j0, &j1, &j2);
    if (l < 0) {                                     /* fastexp.c:264-270 */
        double r = 1.0;
        for (int i = 3; i > 0; --i) r = 1.0 - (double)x * r * g_inv_i[i];
        return r;
    }
    if (l >= 10) return 0.0;                         /* fastexp.c:272-273 */
    return g_tab_a[l][j0] * g_tab_b[l][j1] * g_tab_c[l][j2];  /* :280-282 */
}

/* Cython passes a C double to `double FastExp(const float)`: implicit
 * double->float narrowing (core/math.pxd:17). */
static inline double fast_expn(double x) { return nfo_fastexp((float)x); }

/* ------------------------------------------------------------------------- *
 *  iemtex: 1/(exp(x)-1) by linear interpolation of a 1000-point table
 *  (reference: nestfit/models/hyperfine.pyx:12-45).  The reference fills the
 *  table with numpy at import; callers install the same arrays through
 *  nfo_iemtex_set_table so both sides index identical numbers.
 * ------------------------------------------------------------------------- */
#define T0_SIZE 1000
static double g_t0_x[T0_SIZE], g_t0_y[T0_SIZE];
static double g_t0_xmin, g_t0_xmax, g_t0_inv_dx;
static int    g_t0_ready = 0;

double nfo_t0_xmin(void) { return (NFA_H * 23.0e9 / NFA_KB) / 8.0; }  /* hyperfine.pyx:13-16 */
double nfo_t0_xmax(void) { return (NFA_H * 28.0e9 / NFA_KB) / 2.7; }

void nfo_iemtex_set_table(const double *t0_x, const double *t0_y, long n) {
    if (n != T0_SIZE) return;
    memcpy(g_t0_x, t0_x, sizeof g_t0_x);
    memcpy(g_t0_y, t0_y, sizeof g_t0_y);
    g_t0_xmin = nfo_t0_xmin();
    g_t0_xmax = nfo_t0_xmax();
    g_t0_inv_dx = 1.0 / (g_t0_x[1] - g_t0_x[0]);     /* hyperfine.pyx:20 */
    g_t0_ready = 1;
}

static void t0_default_table(void) {
    /* np.linspace semantics: start + i*step, last point = stop. */
    double xmin = nfo_t0_xmin(), xmax = nfo_t0_xmax();
    double step = (xmax - xmin) / (double)(T0_SIZE - 1);
    double x[T0_SIZE], y[T0_SIZE];
    for (int i = 0; i < T0_SIZE; ++i) {
        x[i] = (i == T0_SIZE - 1) ? xmax : (double)i * step + xmin;
        y[i] = 1.0 / (exp(x[i]) - 1.0);
    }
    nfo_iemtex_set_table(x, y, T0_SIZE);
}

long nfo_iemtex_index(double x) {
    if (!g_t0_ready) t0_default_table();
    if (g_t0_xmin < x && x < g_t0_xmax) return (long)((x - g_t0_xmin) * g_t0_inv_dx);
    return -1;
}

double nfo_iemtex_interp(double x) {
    if (!g_t0_ready) t0_default_table();
    if (g_t0_xmin < x && x < g_t0_xmax) {            /* hyperfine.pyx:36-43 */
        long i_lo = (long)((x - g_t0_xmin) * g_t0_inv_dx);
        long i_hi = i_lo + 1;
        double x_lo = g_t0_x[i_lo];
        double y_lo = g_t0_y[i_lo];
        double y_hi = g_t0_y[i_hi];
        double slope = (y_hi - y_lo) * g_t0_inv_dx;
        return slope * (x - x_lo) + y_lo;
    }
    return 1.0 / expm1(x);                           /* hyperfine.pyx:45 */
}

/* ------------------------------------------------------------------------- *
 *  Ammonia scalar physics (reference: nestfit/models/ammonia.pyx:280-315)
 * ------------------------------------------------------------------------- */
double nfo_swift_convert(double tkin) {               /* ammonia.pyx:280-286 */
    return tkin / (1.0 + (tkin / 41.18) * log(1.0 + 0.6 * exp(-15.7 / tkin)));
}

double nfo_partition_level(long j, double trot) {     /* ammonia.pyx:289-295 */
    return (double)(2 * j + 1)
         * fast_expn(NFA_H * (NFA_BROT * (double)j * (double)(j + 1)
                              + (NFA_CROT - NFA_BROT) * (double)j * (double)j)
                     / (NFA_KB * trot));
}

double nfo_partition_func(int para, double trot) {    /* ammonia.pyx:304-315 */
    double q = 0.0;
    if (para) {
        for (long j = 0; j < NFA_NPART; ++j)
            if (j % 3 != 0) q += nfo_partition_level(j, trot);
    } else {
        for (long j = 0; j < NFA_NPART; ++j)
            if (j % 3 == 0) q += 2 * nfo_partition_level(j, trot);
    }
    return q;
}

/* ------------------------------------------------------------------------- *
 *  Spectrum (reference: nestfit/core/core.pyx:486-530 Spectrum,
 *            nestfit/models/ammonia.pyx:244-277 AmmoniaSpectrum)
 * ------------------------------------------------------------------------- */
struct nfo_spectrum {
    long    size;
    int     model;         /* NFO_MODEL_* */
    int     trans_id;      /* 1..9 (NH3), 1..3 (N2H+), unused (Gaussian) */
    double  rest_freq;     /* Gaussian model only (core.pyx:510) */
    double  noise, nu_chan, nu_min, nu_max, null_lnZ;
    double *xarr, *data, *pred, *tarr, *tbg;
};

double nfo_spectrum_loglike(const nfo_spectrum *s) {  /* core.pyx:522-530 */
    double acc = 0.0;
    for (long i = 0; i < s->size; ++i) {
        double dev = s->data[i] - s->pred[i];
        acc += dev * dev;
    }
    return -acc / (2 * (s->noise * s->noise));
}

nfo_spectrum *nfo_spectrum_new_model(const double *xarr, const double *data, long n,
                                     double noise, int model, int trans_id, double rest_freq) {
    if (n < 2 || !(noise > 0)) return NULL;
    if (model == NFO_MODEL_AMMONIA && (trans_id < 1 || trans_id > NFA_N_LEVELS)) return NULL;
    if (model == NFO_MODEL_DIAZENYLIUM && (trans_id < 1 || trans_id > NFA_N2HP_LEVELS)) return NULL;
    if (model < 0 || model > NFO_MODEL_GAUSSIAN) return NULL;
    if (!(xarr[1] - xarr[0] > 0)) return NULL;        /* core.pyx:502-504 */
    nfo_spectrum *s = (nfo_spectrum *)calloc(1, sizeof *s);
    s->size = n;
    s->model = model;
    s->trans_id = trans_id;
    s->rest_freq = rest_freq;
    s->noise = noise;
    s->xarr = (double *)malloc(sizeof(double) * n);
    s->data = (double *)malloc(sizeof(double) * n);
    s->pred = (double *)calloc(n, sizeof(double));
    s->tarr = (double *)calloc(n, sizeof(double));
    s->tbg  = (double *)malloc(sizeof(double) * n);
    memcpy(s->xarr, xarr, sizeof(double) * n);
    memcpy(s->data, data, sizeof(double) * n);
    s->nu_chan = xarr[1] - xarr[0];                    /* core.pyx:503,512 */
    s->nu_min = xarr[0];
    s->nu_max = xarr[n - 1];
    s->null_lnZ = nfo_spectrum_loglike(s);             /* core.pyx:520 */
    for (long i = 0; i < n; ++i) {                     /* ammonia.pyx:273-277, diazenylium.pyx:132-136 */
        double T0 = NFA_H * xarr[i] / NFA_KB;
        s->tbg[i] = 1.0 / expm1(T0 / NFA_TCMB);
    }
    return s;
}

nfo_spectrum *nfo_spectrum_new(const double *xarr, const double *data, long n,
                               double noise, int trans_id) {
    return nfo_spectrum_new_model(xarr, data, n, noise, NFO_MODEL_AMMONIA, trans_id, 0.0);
}

/* hyperfine tables of the spectrum's transition (ammonia.pyx:232-241, diazenylium.pyx:97-105) */
typedef struct { int nhf; double nu; const double *voff, *wts; } trans_tab;
static trans_tab spectrum_trans(const nfo_spectrum *s) {
    trans_tab t;
    int k = s->trans_id - 1;
    if (s->model == NFO_MODEL_DIAZENYLIUM) {
        t.nhf = nfa_n2hp_nhf[k]; t.nu = nfa_n2hp_nu[k]; t.voff = nfa_n2hp_voff[k]; t.wts = nfa_n2hp_tau_wts[k];
    } else {
        t.nhf = nfa_nhf[k]; t.nu = nfa_nu[k]; t.voff = nfa_voff[k]; t.wts = nfa_tau_wts[k];
    }
    return t;
}

void nfo_spectrum_free(nfo_spectrum *s) {
    if (!s) return;
    free(s->xarr); free(s->data); free(s->pred); free(s->tarr); free(s->tbg);
    free(s);
}

long nfo_spectrum_size(const nfo_spectrum *s) { return s->size; }
double nfo_spectrum_null_lnZ(const nfo_spectrum *s) { return s->null_lnZ; }
const double *nfo_spectrum_pred(const nfo_spectrum *s) { return s->pred; }
const double *nfo_spectrum_tarr(const nfo_spectrum *s) { return s->tarr; }
const double *nfo_spectrum_tbg(const nfo_spectrum *s) { return s->tbg; }
void nfo_spectrum_set_data(nfo_spectrum *s, const double *data) {
    memcpy(s->data, data, sizeof(double) * s->size);
}

/* Window of one hyperfine line: [lo, hi) after clamping, or lo=hi=-1 when the
 * line is skipped (reference: nestfit/models/hyperfine.pyx:70-93). */
static int hf_window(const nfo_spectrum *s, double hf_nucen, double hf_idenom,
                     long *lo_out, long *hi_out) {
    double nu_cutoff = sqrt(12.5 / hf_idenom);
    double nu_lo = (hf_nucen - s->nu_min - nu_cutoff);
    double nu_hi = (hf_nucen - s->nu_min + nu_cutoff);
    long lo = (long)floor(nu_lo / s->nu_chan);
    long hi = (long)floor(nu_hi / s->nu_chan);
    if (hi < 0 || lo > s->size - 1) return 0;
    if (lo < 0) lo = 0;
    if (hi > s->size - 1) hi = s->size - 1;
    *lo_out = lo; *hi_out = hi;
    return 1;
}

void nfo_hf_windows(const nfo_spectrum *s, double voff, double sigm,
                    long *lo, long *hi) {
    trans_tab t = spectrum_trans(s);
    for (int i = 0; i < t.nhf; ++i) {
        double hf_freq   = (1.0 - t.voff[i] / NFA_CKMS) * t.nu;
        double hf_width  = sigm / NFA_CKMS * hf_freq;
        double hf_offset = voff / NFA_CKMS * hf_freq;
        double hf_nucen  = hf_freq - hf_offset;
        double hf_idenom = 0.5 / (hf_width * hf_width);
        lo[i] = hi[i] = -1;
        hf_window(s, hf_nucen, hf_idenom, &lo[i], &hi[i]);
    }
}

/* reference: nestfit/models/hyperfine.pyx:52-118 (c_hf_predict, __APPROX) */
static void hf_predict(nfo_spectrum *s, double voff, double tex,
                       double ltau_main, double sigm) {
    trans_tab t = spectrum_trans(s);
    double tau_main = pow(10.0, ltau_main);            /* hyperfine.pyx:63 */
    for (long i = 0; i < s->size; ++i) s->tarr[i] = 0.0;
    for (int i = 0; i < t.nhf; ++i) {
        double hf_freq   = (1.0 - t.voff[i] / NFA_CKMS) * t.nu;
        double hf_width  = sigm / NFA_CKMS * hf_freq;
        double hf_offset = voff / NFA_CKMS * hf_freq;
        double hf_nucen  = hf_freq - hf_offset;
        double hf_tau    = tau_main * t.wts[i];
        double hf_idenom = 0.5 / (hf_width * hf_width);
        long lo, hi;
        if (!hf_window(s, hf_nucen, hf_idenom, &lo, &hi)) continue;
        for (long j = lo; j < hi; ++j) {               /* hyperfine.pyx:93-96 */
            double nu = s->xarr[j] - hf_nucen;
            double tau_exp = nu * nu * hf_idenom;
            s->tarr[j] += hf_tau * fast_expn(tau_exp);
        }
    }
    for (long i = 0; i < s->size; ++i) {               /* hyperfine.pyx:103-113 */
        if (s->tarr[i] == 0.0) continue;
        double T0 = NFA_H * s->xarr[i] / NFA_KB;
        s->pred[i] += (T0 * (nfo_iemtex_interp(T0 / tex) - s->tbg[i])
                       * (1.0 - fast_expn(s->tarr[i])));
    }
}

/* reference: nestfit/models/ammonia.pyx:326-361 (c_amm_predict) */
void nfo_amm_predict(nfo_spectrum *s, const double *params, long ndim,
                     int cold, int lte) {
    long ncomp = ndim / NFA_N_PARAMS;
    int t = s->trans_id - 1;
    int para = ((t + 1) % 3) != 0;                     /* ammonia.pyx:237 */
    double nu0 = nfa_nu[t];
    for (long i = 0; i < s->size; ++i) s->pred[i] = 0.0;
    for (long i = 0; i < ncomp; ++i) {
        double voff = params[i];
        double trot = params[ncomp + i];
        double tex  = params[2 * ncomp + i];
        double ntot = params[3 * ncomp + i];
        double sigm = params[4 * ncomp + i];
        double orth = params[5 * ncomp + i];
        if (cold) trot = nfo_swift_convert(trot);
        if (lte) tex = trot;
        double zlev = nfo_partition_level(t + 1, trot);
        double qtot = nfo_partition_func(para, trot);
        double species_frac = para ? 1.0 - orth : orth;
        double pop_rotstate = pow(10.0, ntot) * species_frac * zlev / qtot;
        double expterm = ((1.0 - exp(-NFA_H * nu0 / (NFA_KB * tex)))
                        / (1.0 + exp(-NFA_H * nu0 / (NFA_KB * tex))));
        double fracterm = (NFA_CCMS * NFA_CCMS) * nfa_ea[t] / (8 * M_PI * (nu0 * nu0));
        double widthterm = NFA_CKMS / (sigm * nu0 * sqrt(2 * M_PI));
        double tau_main = pop_rotstate * fracterm * expterm * widthterm;
        hf_predict(s, voff, tex, log10(tau_main), sigm);
    }
}

/* reference: nestfit/models/diazenylium.pyx:138-154 (c_nnhp_predict) */
void nfo_nnhp_predict(nfo_spectrum *s, const double *params, long ndim) {
    long ncomp = ndim / NFA_N2HP_PARAMS;
    for (long i = 0; i < s->size; ++i) s->pred[i] = 0.0;
    for (long i = 0; i < ncomp; ++i) {
        double voff = params[i];
        double tex  = params[ncomp + i];
        double ltau = params[2 * ncomp + i];
        double sigm = params[3 * ncomp + i];
        hf_predict(s, voff, tex, ltau, sigm);
    }
}

/* reference: nestfit/models/gaussian.pyx:17-50 (c_gauss_predict) */
void nfo_gauss_predict(nfo_spectrum *s, const double *params, long ndim) {
    int ncomp = (int)(ndim / NFA_GAUSS_PARAMS);
    for (long i = 0; i < s->size; ++i) s->pred[i] = 0.0;
    for (int i = 0; i < ncomp; ++i) {
        double voff = params[i];
        double sigm = params[ncomp + i];
        double peak = params[2 * ncomp + i];
        double nu_width = sigm / NFA_CKMS * s->rest_freq;
        double nu_cen   = s->rest_freq * (1 - voff / NFA_CKMS);
        double nu_denom = 0.5 / (nu_width * nu_width);
        double nu_cutoff = sqrt(12.5 / nu_denom);
        double nu_lo = (nu_cen - s->nu_min - nu_cutoff);
        double nu_hi = (nu_cen - s->nu_min + nu_cutoff);
        int lo = (int)floor(nu_lo / s->nu_chan);
        int hi = (int)floor(nu_hi / s->nu_chan);
        if (hi < 0 || lo > s->size - 1) continue;
        if (lo < 0) lo = 0;
        if (hi > s->size - 1) hi = (int)(s->size - 1);
        for (int j = lo; j < hi; ++j) {
            double nu = s->xarr[j] - nu_cen;
            s->pred[j] += peak * fast_expn(nu * nu * nu_denom);
        }
    }
}

/* ------------------------------------------------------------------------- *
 *  Priors (reference: nestfit/core/core.pyx:23-161 Distribution,
 *          :169-476 Prior family and PriorTransformer.c_transform)
 * ------------------------------------------------------------------------- */
#define FWHM 2.3548200450309493                        /* core.pyx:20 */

static double dist_ppf_interp(const nfo_dist *d, double u) {  /* core.pyx:47-63 */
    long i_lo = (long)((double)(d->size - 1) * u);
    long i_hi = i_lo + 1;
    /* The reference reads ppf[size] for u == 1 (out of bounds);
     * parity is defined for u in [0,1).  Clamp instead of reading past the end. */
    if (i_lo < 0) i_lo = 0;
    if (i_lo > d->size - 1) i_lo = d->size - 1;
    if (i_hi > d->size - 1) i_hi = d->size - 1;
    double x_lo = (double)i_lo * d->du;
    double y_lo = d->ppf[i_lo];
    double y_hi = d->ppf[i_hi];
    double slope = (y_hi - y_lo) / d->du;
    return slope * (u - x_lo) + y_lo;
}

static double dist_cdf_interp(const nfo_dist *d, const double *cdf, double u) {
    /* core.pyx:65-107 */
    if (u <= cdf[0]) u = 1e-64;
    long i_lo = 0, i_hi = d->size, i = i_hi / 2;
    while (i != i_lo) {
        if (u > cdf[i]) i_lo = i; else i_hi = i;
        i = (i_hi + i_lo) / 2;
    }
    i_lo = (i < d->size) ? i : d->size - 1;
    i_hi = i_lo + 1;
    if (i_hi > d->size - 1) i_hi = d->size - 1;        /* guard the OOB read */
    double x_lo = d->xax[i_lo];
    double y_lo = cdf[i_lo];
    double y_hi = cdf[i_hi];
    double slope = (y_hi - y_lo) / d->dx;
    return 1 / slope * (u - y_lo) + x_lo;
}

static void dist_cdf_over_interval(const nfo_dist *d, double *cdf, double x_lo,
                                   double x_hi, double sfact) {
    /* core.pyx:109-161; writes every entry of `cdf`, so a private scratch
     * copy gives the same numbers as the reference's in-place rewrite. */
    double csum = 0.0;
    if (x_lo > x_hi) { double t = x_lo; x_lo = x_hi; x_hi = t; }
    long i_lo = (long)((x_lo - d->xmin) / d->dx);
    if (i_lo >= d->size) i_lo = d->size - 1;
    else if (i_lo < 0) i_lo = 0;
    long i_hi = (long)((x_hi - d->xmin) / d->dx);
    if (i_hi == i_lo) i_hi = i_lo + 1;
    if (i_hi > d->size) i_hi = d->size;
    else if (i_hi < 0) i_hi = 1;
    for (long i = 0; i < i_lo; ++i) cdf[i] = 0.0;
    for (long i = i_hi; i < d->size; ++i) cdf[i] = 1.0;
    if (i_hi - i_lo == 1) {
        cdf[i_lo] = 1.0;
    } else {
        cdf[i_lo] = 0.0;
        double inv_delta_i = 1.0 / (double)(i_hi - i_lo);
        for (long i = i_lo + 1; i < i_hi; ++i) {
            double scale;
            if (sfact == 0.0) scale = 1.0;
            else if (sfact == 1.0) scale = (1.0 - (double)(i - i_lo) * inv_delta_i);
            else if (sfact == 2.0) {
                scale = (1.0 - (double)(i - i_lo) * inv_delta_i);
                scale *= scale;
            } else scale = pow(1.0 - (double)(i - i_lo) * inv_delta_i, sfact);
            csum += 0.5 * (d->pdf[i] + d->pdf[i - 1]) * scale;
            cdf[i] = csum;
        }
    }
    for (long i = i_lo; i < i_hi; ++i) cdf[i] /= csum;
}

/* `sigm_prior.interp(utheta, n)` is polymorphic in the reference; the simple
 * kinds are what its constructors use (prior_constructors.py:62-66,122-127;
 * core.pyx:849). */
static void simple_interp(const nfo_priorset *ps, int kind, int dist, int p_ix,
                          double value, double *u, long n) {
    long ix = (long)p_ix * n;
    if (kind == NFO_PRIOR_CONSTANT) {                  /* core.pyx:233-238 */
        for (long i = 0; i < n; ++i) u[ix + i] = value;
    } else if (kind == NFO_PRIOR_ORDERED) {            /* core.pyx:242-258 */
        double umin = 0.0;
        for (long i = 0; i < n; ++i) {
            double uu = umin + (1 - umin) * u[ix + i];
            umin = uu;
            u[ix + i] = dist_ppf_interp(&ps->dists[dist], uu);
        }
    } else {                                           /* core.pyx:192-197 */
        for (long i = 0; i < n; ++i)
            u[ix + i] = dist_ppf_interp(&ps->dists[dist], u[ix + i]);
    }
}

void nfo_transform(const nfo_priorset *ps, double *u, long n) {
    /* core.pyx:459-476: every prior in order, in place. */
    for (int k = 0; k < ps->n_prior; ++k) {
        const nfo_prior *p = &ps->priors[k];
        long ix = (long)p->p_ix * n;
        switch (p->kind) {
        case NFO_PRIOR_SIMPLE:
        case NFO_PRIOR_CONSTANT:
        case NFO_PRIOR_ORDERED:
            simple_interp(ps, p->kind, p->dist0, p->p_ix, p->value, u, n);
            break;
        case NFO_PRIOR_DUPLICATE: {                    /* core.pyx:211-221 */
            long ix_dup = (long)p->p_ix2 * n;
            for (long i = 0; i < n; ++i) {
                double v = dist_ppf_interp(&ps->dists[p->dist0], u[ix + i]);
                u[ix + i] = v;
                u[ix_dup + i] = v;
            }
        } break;
        case NFO_PRIOR_SPACED: {                       /* core.pyx:280-292 */
            double v = dist_ppf_interp(&ps->dists[p->dist0], u[ix]);
            u[ix] = v;
            for (long i = 1; i < n; ++i) {
                v = v + dist_ppf_interp(&ps->dists[p->dist1], u[ix + i]);
                u[ix + i] = v;
            }
        } break;
        case NFO_PRIOR_CENSEP: {                       /* core.pyx:305-318 */
            double vcen = dist_ppf_interp(&ps->dists[p->dist0], u[ix]);
            if (n == 1) u[ix] = vcen;
            else if (n == 2) {
                double vsep = dist_ppf_interp(&ps->dists[p->dist1], u[ix + 1]);
                u[ix]     = vcen - 0.5 * vsep;
                u[ix + 1] = vcen + 0.5 * vsep;
            }
        } break;
        case NFO_PRIOR_RESOLVED_CENSEP: {              /* core.pyx:347-366 */
            long ix_s = (long)p->p_ix2 * n;
            simple_interp(ps, p->sub_kind, p->dist2, p->p_ix2, p->value, u, n);
            double vcen = dist_ppf_interp(&ps->dists[p->dist0], u[ix]);
            if (n == 1) u[ix] = vcen;
            else if (n == 2) {
                double vsep = dist_ppf_interp(&ps->dists[p->dist1], u[ix + 1]);
                double min_sep = p->sep_scale * sqrt(u[ix_s] * u[ix_s + 1]);
                if (min_sep > vsep) vsep = min_sep;
                u[ix]     = vcen - 0.5 * vsep;
                u[ix + 1] = vcen + 0.5 * vsep;
            }
        } break;
        case NFO_PRIOR_RESOLVED_PLACEMENT: {           /* core.pyx:391-435 */
            if (n > 10) break;
            const nfo_dist *vd = &ps->dists[p->dist0];
            long ix_s = (long)p->p_ix2 * n;
            double v_lo = vd->xmin, v_hi = vd->xmax;
            double min_seps[10];
            simple_interp(ps, p->sub_kind, p->dist2, p->p_ix2, p->value, u, n);
            if (n == 1) { u[ix] = dist_ppf_interp(vd, u[ix]); break; }
            double sep_tot = 0.0;
            min_seps[0] = 0.0;
            for (long i = 1; i < n; ++i) {
                double sep = p->sep_scale * sqrt(u[ix_s + i] * u[ix_s + i - 1]);
                sep_tot += sep;
                min_seps[i] = sep;
            }
            if (sep_tot > v_hi - v_lo) {
                double overf = (v_hi - v_lo) / sep_tot;
                sep_tot = 0.0;
                for (long i = 0; i < n; ++i) {
                    min_seps[i] *= overf;
                    sep_tot += min_seps[i];
                }
            }
            v_hi -= sep_tot;
            double *cdf = (double *)malloc(sizeof(double) * vd->size);
            for (long i = 0; i < n; ++i) {
                double sep = min_seps[i];
                v_lo += sep;
                v_hi += sep;
                dist_cdf_over_interval(vd, cdf, v_lo, v_hi, (double)(n - 1 - i));
                v_lo = dist_cdf_interp(vd, cdf, u[ix + i]);
                u[ix + i] = v_lo;
            }
            free(cdf);
        } break;
        default: break;
        }
    }
}

double nfo_dist_ppf_interp(const nfo_dist *d, double u) { return dist_ppf_interp(d, u); }
double nfo_dist_cdf_interp(const nfo_dist *d, double u) { return dist_cdf_interp(d, d->cdf, u); }
long nfo_dist_ppf_index(const nfo_dist *d, double u) { return (long)((double)(d->size - 1) * u); }

/* ------------------------------------------------------------------------- *
 *  Runner (reference: nestfit/models/ammonia.pyx:423-432 c_loglikelihood)
 * ------------------------------------------------------------------------- */
static long model_npar(int model) {
    return model == NFO_MODEL_DIAZENYLIUM ? NFA_N2HP_PARAMS : model == NFO_MODEL_GAUSSIAN ? NFA_GAUSS_PARAMS
                                                                                        : NFA_N_PARAMS;
}

/* ammonia.pyx:423-432, diazenylium.pyx:207-216, gaussian.pyx:96-100: the model is the spectra's */
double nfo_runner_loglike(nfo_spectrum **spectra, int n_spec,
                          const nfo_priorset *ps, double *utheta, long ncomp,
                          int cold, int lte) {
    double lnL = 0.0;
    int model = spectra[0]->model;
    long ndim = model_npar(model) * ncomp;
    if (ps) nfo_transform(ps, utheta, ncomp);
    for (int i = 0; i < n_spec; ++i) {
        if (model == NFO_MODEL_DIAZENYLIUM) nfo_nnhp_predict(spectra[i], utheta, ndim);
        else if (model == NFO_MODEL_GAUSSIAN) nfo_gauss_predict(spectra[i], utheta, ndim);
        else nfo_amm_predict(spectra[i], utheta, ndim, cold, lte);
        lnL += nfo_spectrum_loglike(spectra[i]);
    }
    return lnL;
}

void nfo_runner_loglike_batch(nfo_spectrum **spectra, int n_spec,
                              const nfo_priorset *ps, double *U, double *lnL,
                              long B, long ncomp, int cold, int lte) {
    long ndim = model_npar(spectra[0]->model) * ncomp;
    for (long b = 0; b < B; ++b)
        lnL[b] = nfo_runner_loglike(spectra, n_spec, ps, U + b * ndim, ncomp, cold, lte);
}

int nfo_n2hp_nhf(int trans_id) { return nfa_n2hp_nhf[trans_id - 1]; }
double nfo_n2hp_nu(int trans_id) { return nfa_n2hp_nu[trans_id - 1]; }
double nfo_n2hp_voff(int trans_id, int i) { return nfa_n2hp_voff[trans_id - 1][i]; }
double nfo_n2hp_tau_wt(int trans_id, int i) { return nfa_n2hp_tau_wts[trans_id - 1][i]; }

/* Static line data accessors (for the data cross-check test). */
int nfo_trans_nhf(int trans_id) { return nfa_nhf[trans_id - 1]; }
double nfo_trans_nu(int trans_id) { return nfa_nu[trans_id - 1]; }
double nfo_trans_ea(int trans_id) { return nfa_ea[trans_id - 1]; }
double nfo_trans_voff(int trans_id, int i) { return nfa_voff[trans_id - 1][i]; }
double nfo_trans_tau_wt(int trans_id, int i) { return nfa_tau_wts[trans_id - 1][i]; }

/* Batch helpers for the test-suite (plain loops over the functions above). */
void nfo_fastexp_many(const float *x, double *out, long n) {
    for (long i = 0; i < n; ++i) out[i] = nfo_fastexp(x[i]);
}
void nfo_fast_expn_many(const double *x, double *out, long n) {
    for (long i = 0; i < n; ++i) out[i] = fast_expn(x[i]);
}
void nfo_iemtex_many(const double *x, double *out, long n) {
    for (long i = 0; i < n; ++i) out[i] = nfo_iemtex_interp(x[i]);
}
