/* nf_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * See nf_oracle.c for the reference file:line each function restates. */
#ifndef NF_ORACLE_H
#define NF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* FastExp */
void   nfo_fastexp_init(void);
double nfo_fastexp(float x);
void   nfo_fastexp_indices(float x, int *l, int *j0, int *j1, int *j2);
void   nfo_fastexp_many(const float *x, double *out, long n);
void   nfo_fast_expn_many(const double *x, double *out, long n); /* double arg narrowed to float, math.pxd:17 */

/* 1/(e^x - 1) interpolation */
double nfo_t0_xmin(void);
double nfo_t0_xmax(void);
void   nfo_iemtex_set_table(const double *t0_x, const double *t0_y, long n);
long   nfo_iemtex_index(double x);
double nfo_iemtex_interp(double x);
void   nfo_iemtex_many(const double *x, double *out, long n);

/* scalar NH3 physics */
double nfo_swift_convert(double tkin);
double nfo_partition_level(long j, double trot);
double nfo_partition_func(int para, double trot);

/* spectra */
enum { NFO_MODEL_AMMONIA = 0, NFO_MODEL_DIAZENYLIUM = 1, NFO_MODEL_GAUSSIAN = 2 };
typedef struct nfo_spectrum nfo_spectrum;
nfo_spectrum *nfo_spectrum_new_model(const double *xarr, const double *data, long n,
                                     double noise, int model, int trans_id, double rest_freq);
nfo_spectrum *nfo_spectrum_new(const double *xarr, const double *data, long n,
                               double noise, int trans_id);
void   nfo_spectrum_free(nfo_spectrum *s);
long   nfo_spectrum_size(const nfo_spectrum *s);
double nfo_spectrum_null_lnZ(const nfo_spectrum *s);
const double *nfo_spectrum_pred(const nfo_spectrum *s);
const double *nfo_spectrum_tarr(const nfo_spectrum *s);
const double *nfo_spectrum_tbg(const nfo_spectrum *s);
void   nfo_spectrum_set_data(nfo_spectrum *s, const double *data);
double nfo_spectrum_loglike(const nfo_spectrum *s);
void   nfo_hf_windows(const nfo_spectrum *s, double voff, double sigm,
                      long *lo, long *hi);
void   nfo_amm_predict(nfo_spectrum *s, const double *params, long ndim,
                       int cold, int lte);
void   nfo_nnhp_predict(nfo_spectrum *s, const double *params, long ndim);   /* diazenylium.pyx:138-154 */
void   nfo_gauss_predict(nfo_spectrum *s, const double *params, long ndim);  /* gaussian.pyx:17-50 */

/* priors */
enum {
    NFO_PRIOR_SIMPLE = 0,             /* Prior                  core.pyx:169-197 */
    NFO_PRIOR_DUPLICATE = 1,          /* DuplicatePrior         core.pyx:200-221 */
    NFO_PRIOR_CONSTANT = 2,           /* ConstantPrior          core.pyx:224-238 */
    NFO_PRIOR_ORDERED = 3,            /* OrderedPrior           core.pyx:241-258 */
    NFO_PRIOR_SPACED = 4,             /* SpacedPrior            core.pyx:261-292 */
    NFO_PRIOR_CENSEP = 5,             /* CenSepPrior            core.pyx:295-318 */
    NFO_PRIOR_RESOLVED_CENSEP = 6,    /* ResolvedCenSepPrior    core.pyx:321-366 */
    NFO_PRIOR_RESOLVED_PLACEMENT = 7, /* ResolvedPlacementPrior core.pyx:369-435 */
};

typedef struct {
    long   size;
    double du, dx, xmin, xmax;
    double *xax, *pdf, *cdf, *ppf;
} nfo_dist;

typedef struct {
    int    kind;
    int    p_ix;      /* parameter slot (vcen slot for the composite kinds) */
    int    p_ix2;     /* duplicate slot / sigm slot */
    int    dist0;     /* main / vcen / independent distribution */
    int    dist1;     /* vsep / dependent distribution */
    int    dist2;     /* sigm distribution */
    int    sub_kind;  /* kind of the sigm sub-prior (SIMPLE/CONSTANT/ORDERED) */
    int    pad_;
    double value;     /* constant value (CONSTANT, or constant sigm sub-prior) */
    double sep_scale; /* FWHM * scale */
} nfo_prior;

typedef struct {
    int        n_prior, n_dist;
    nfo_prior *priors;
    nfo_dist  *dists;
} nfo_priorset;

void   nfo_transform(const nfo_priorset *ps, double *utheta, long ncomp);
double nfo_dist_ppf_interp(const nfo_dist *d, double u);
double nfo_dist_cdf_interp(const nfo_dist *d, double u);
long   nfo_dist_ppf_index(const nfo_dist *d, double u);

/* runner */
double nfo_runner_loglike(nfo_spectrum **spectra, int n_spec,
                          const nfo_priorset *ps, double *utheta, long ncomp,
                          int cold, int lte);
void   nfo_runner_loglike_batch(nfo_spectrum **spectra, int n_spec,
                                const nfo_priorset *ps, double *U, double *lnL,
                                long B, long ncomp, int cold, int lte);

/* static line data */
int    nfo_trans_nhf(int trans_id);
double nfo_trans_nu(int trans_id);
double nfo_trans_ea(int trans_id);
double nfo_trans_voff(int trans_id, int i);
double nfo_trans_tau_wt(int trans_id, int i);
int    nfo_n2hp_nhf(int trans_id);
double nfo_n2hp_nu(int trans_id);
double nfo_n2hp_voff(int trans_id, int i);
double nfo_n2hp_tau_wt(int trans_id, int i);

#ifdef __cplusplus
}
#endif
#endif
