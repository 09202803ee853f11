/* nestfit_amd.h -- C ABI of the MI355X-native NH3 log-likelihood engine.
 *
 * Drop-in boundary for the hot path of autocorr/nestfit v0.2: everything that
 * runs inside `AmmoniaRunner.c_loglikelihood` (prior transform -> model spectra
 * -> chi^2), batched over live points and map pixels on one gfx950 device.
 * Plain pointers and sizes only; no torch / numpy types.  All functions return
 * 0 on success and a non-zero code on failure (text via nfa_last_error());
 * the one exception is nfa_loglike_callback, which has MultiNest's `LogLike`
 * signature and therefore no error channel (it writes NaN into *lnew).
 *
 * Each entry point names the reference interface it replaces (file:line in
 * /root/reference).  INTEGRATION.md shows the ctypes / Cython stubs a
 * maintainer of the reference would add to bind them.
 */
#ifndef NESTFIT_AMD_H
#define NESTFIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFA_OK            0
#define NFA_ERR_ARG       1   /* invalid argument (reference: assert / ValueError) */
#define NFA_ERR_DEVICE    2   /* HIP runtime error, or no gfx950 device            */
#define NFA_ERR_STATE     3   /* call order / missing table                        */

typedef struct nfa_specset nfa_specset;  /* device-resident spectra of 1..n_pix pixels */
typedef struct nfa_priors  nfa_priors;   /* device-resident prior program              */
typedef struct nfa_runner  nfa_runner;   /* (specset, priors, ncomp, cold, lte) bundle */

/* ---- process / device ---------------------------------------------------- */
const char *nfa_last_error(void);
int nfa_version(void);
int nfa_device_count(int *count);
int nfa_set_device(int device);              /* one process per GPU: call first */
int nfa_device_synchronize(void);
int nfa_device_name(char *buf, int buflen);
/* the 16-byte UUID of the engine's device as 32 hex digits (two ranks that report the same one share a GPU) */
int nfa_device_uuid(char *buf, int buflen);

/* Numerical mode of the FastExp replacement (process default, 2 unless set; a runner can pin
 * its own with nfa_runner_set_exp_mode):
 *   0 = "table": the reference's three-table product held in LDS
 *       (nestfit/core/fastexp.c:234-283), bit-identical table indices;
 *   (1, an fp64 polynomial form of the same exponential, was a likelihood mode until round 3:
 *       slower than the table mode and less faithful, it is refused now);
 *   2 = "fast" : window / table indices and the float-narrowed FastExp argument
 *       in fp64 exactly as above, exponentials in fp32 with split exponents;
 *       <= 1e-6 relative on brightness temperature (the metric's tolerance). */
int nfa_set_exp_mode(int mode);
int nfa_get_exp_mode(void);
/* Engine tuning knobs for A/B measurements (key, value):
 *   "wpb"           waves per workgroup of the likelihood kernel (1..16, default 1), and
 *   "wpb_table"     the same in table mode (0 = chosen per spectra set, the default): both are
 *                   taken over by runners created afterwards;
 *   "lnl_split"     waves that share one (item, spectrum) unit of the likelihood kernel: 1, 2, 4, or 0 = chosen per
 *                   launch (the default: 4 for single points and small batches, 1 for batches that fill the
 *                   GPU's wave slots).  More, shorter waves balance a small launch better at a few per cent more
 *                   instructions.  The chi^2 of a unit is always the sum of four interleaved row parts (rows h,
 *                   h + 4, ...) taken in order, whoever computed them, so log-likelihoods are bitwise independent
 *                   of this option and of the batch an evaluation travels in;
 *   "lnl_queue"     1 (default) / 0: table mode, launches of two and more (item, spectrum) units per wave slot of the
 *                   device (the coalesced steps of nfa_runner_loglike_batch_dev; spectra of 512 channels and more) run as
 *                   resident workgroups whose waves draw their units from a queue, or every wave owns one unit.
 *                   Bitwise the same results; "lnl_queue_wg" 1 / 2 (0 = as many as fit): workgroups per CU of such a
 *                   launch (A/B knob);
 *   "lnl_cap"       workgroups of the likelihood kernel resident per CU at most (fast and poly mode,
 *                   0 = no cap, the default; 1..8): A/B knob, see DESIGN.md;
 *   "streams"       number of HIP streams ("lanes", 1..8) that runners created afterwards spread consecutive
 *                   nfa_runner_loglike_batch_dev calls over; 0 (default) = six streams of which a batch of
 *                   about one wavefront per wave slot of the GPU uses all six and any other batch four;
 *   "sampler_parts" groups of pixels the device sampler pipelines over the lanes (1..4, default 3);
 *   "coalesce"      1..8 (default 8; 4 until round 5): nfa_runner_loglike_batch_dev calls of one shape (same number of rows, a
 *                   multiple of 64; pixels given or not) that follow each other are held and launched together,
 *                   at most this many (1 = every call its own launches).  Results are bitwise the same; anything
 *                   that looks at them or changes the way launches are made (nfa_runner_synchronize,
 *                   nfa_device_synchronize, the host-pointer calls, a mode change, the sampler) launches what is
 *                   held first.  Read at every call;
 *   "prior_stage"   1 / 0: the set-up kernel stages the prior tables in LDS (default) or reads them from global
 *                   memory; taken over by priors created afterwards (A/B knob: no measurable difference in the
 *                   pipelined rates, the staged form is 5 us shorter when the stage runs alone);
 *   "setup_ti", "setup_threads"  items (8..64, default 64) and threads (256 .. 512 in steps of 64; default 256, 512 in the table mode) per workgroup of
 *                   the set-up kernel: A/B knobs, see DESIGN.md;
 *   "setup_sub"     1: one group of 64 items per set-up workgroup in the table mode; 0 (default): two groups behind one
 *                   copy of the tables where every batch of the launch is a multiple of 128 rows;
 *   "point"         1 / 0: single points and small batches (nfa_runner_loglike_batch with B <= 128,
 *                   nfa_loglike_callback) go through the one-launch point kernel (default: one workgroup per
 *                   point, the result written to a mapped host buffer) or through the batch kernels;
 *   "graph"         1 / 0: with "point" 0, replay single-point calls as one captured hipGraph or not (default:
 *                   on, off when the rocprofiler tool library is attached: capture crashed under it);
 *   "sampler_parts"       1..4 (default 3): groups of pixels the device sampler pipelines over the stream lanes;
 *   "sampler_ellipsoids"  1: one bounding ellipsoid per pixel whatever the dimension; 0 (default): up to four where at
 *                   most six dimensions are sampled (per sampler: nfa_sampler_set_ellipsoids);
 *   "sampler_walk_factor" a pixel turns from rejection rounds to constrained walks when a round accepts fewer than
 *                   1 in factor * n_steps candidates (and back above 8 times that); 0 (default): 64 up to six sampled
 *                   dimensions, 2 above (the numpy twin takes `walk_factor=`: the two must agree to run alike);
 *   "sampler_walkers"     walkers per pixel of a walk cycle, 64 / 128 / 192 / 256; 0 (default): by the live points
 *                   (128 from 384, 256 from 768);
 *   "sampler_frames"      rotated box frames of a one-ellipsoid bound (nfa_sampler_set_boxes): -2 (default) and -1
 *                   none, 0..64; "sampler_margin_pct": the boxes' margin factor in hundredths (0 = 250);
 *   "sampler_shear_pct"   the shear in front of one-ellipsoid bounds (nfa_sampler_set_shear): its safety factor in
 *                   hundredths, -1 (default) = 300 where the shape allows, 0 = off, 100..100000;
 *   "sampler_pairs_pct"   the pair ellipses of a sheared and boxed bound (nfa_sampler_set_pairs): their safety factor in
 *                   hundredths, -1 (default) = 200, 0 = off, 100..100000 (the host side names sets of these three:
 *                   nestfit_amd.sampler.PRECISION, 'speed' = 250 / 175 / 175, round 4's defaults);
 *   "sampler_ktarget"     replacements per pixel and rejection round a pixel's own share of the round's proposals aims
 *                   at: halved after a round with more than twice as many, doubled after one with fewer than half
 *                   (-1 = the default, 16; 0 = every pixel the round's number); "sampler_ratio_max": proposals drawn
 *                   per round with vetoes on, at most this multiple of the evaluations aimed for (0 = the default,
 *                   32; the candidate buffers are sized by it when a sampler is created); "sampler_kmax": most
 *                   proposals a pixel gets in a round (0 = 65536).  The twin: `k_target=`, `ratio_max=`, `kmax=`;
 *   "sampler_refit_every" rejection-mode pixels refit their bound in rounds that are multiples of this (default 4);
 *                   the sampler_* keys are read when a sampler is created / begun, A/B knobs like the rest;
 *   "ablate"        only in builds with -DNFA_ABLATE (timing experiments, results invalid; the
 *                   shipped library rejects the key): bit mask, 1 skip the Tb pass, 2 skip the
 *                   hyperfine-line loop, 4 skip the rows, 8 skip the line set-up.
 * Unknown keys and values out of range return NFA_ERR_ARG. */
int nfa_set_option(const char *key, int value);

/* 1/(e^x-1) interpolation table.  The reference builds T0_X, T0_Y with numpy at
 * import (nestfit/models/hyperfine.pyx:12-20); the host passes the same
 * arrays so device and reference index identical numbers.  n must be 1000. */
int nfa_set_iemtex_table(const double *t0_x, const double *t0_y, int64_t n);

/* ---- spectra --------------------------------------------------------------
 * Replaces AmmoniaSpectrum.__init__ / Spectrum.__init__
 * (nestfit/models/ammonia.pyx:244-277, nestfit/core/core.pyx:486-520) for all
 * spectra of one pixel or of a whole cube.  Inputs are copied to the device;
 * the caller keeps ownership of its arrays.
 *   sizes[n_spec], trans_ids[n_spec] (1..9), xarr[s] -> sizes[s] doubles
 *   (ascending Hz, shared by all pixels), data[n_pix][sum(sizes)] channel-
 *   contiguous per pixel with spectra concatenated in order, noise[n_pix][n_spec].
 */
int nfa_specset_create(nfa_specset **out, int n_spec, const int64_t *sizes,
                       const int32_t *trans_ids, const double *const *xarr,
                       int64_t n_pix, const double *data, const double *noise);
/* The same for the sibling models on the same kernels:
 *   model 0 = ammonia (as above);
 *   model 1 = N2H+, DiazenyliumSpectrum.__init__ (nestfit/models/diazenylium.pyx:108-136),
 *             trans_ids 1..3 = J 1-0, 2-1, 3-2; parameters per component voff, tex, ltau, sigm
 *             (c_nnhp_predict, diazenylium.pyx:138-154);
 *   model 2 = Gaussian on a plain Spectrum (nestfit/core/core.pyx:486-520 with rest_freq;
 *             c_gauss_predict, nestfit/models/gaussian.pyx:17-50): n_spec must be 1,
 *             trans_ids is ignored, rest_freqs[0] = Spectrum.rest_freq in Hz (NULL = 0 like
 *             the reference's default); parameters per component voff, sigm, peak. */
#define NFA_MODEL_AMMONIA      0
#define NFA_MODEL_DIAZENYLIUM  1
#define NFA_MODEL_GAUSSIAN     2
int nfa_specset_create_model(nfa_specset **out, int model, int n_spec, const int64_t *sizes,
                             const int32_t *trans_ids, const double *rest_freqs,
                             const double *const *xarr, int64_t n_pix, const double *data,
                             const double *noise);
int nfa_specset_destroy(nfa_specset *ss);
int nfa_specset_set_data(nfa_specset *ss, int64_t pix, const double *data);
/* null_lnZ[n_pix][n_spec] = -sum(data^2)/(2 noise^2)   (core.pyx:517-520) */
int nfa_specset_null_lnz(const nfa_specset *ss, double *out);
/* tbg[sum(sizes)] = 1/expm1(h nu / (k TCMB))           (ammonia.pyx:273-277) */
int nfa_specset_tbg(const nfa_specset *ss, double *out);
int64_t nfa_specset_chan_tot(const nfa_specset *ss);

/* ---- priors ---------------------------------------------------------------
 * Replaces PriorTransformer + the Prior family (nestfit/core/core.pyx:169-476).
 * A prior program is the ordered list of priors (executed in order, in place,
 * like c_transform, core.pyx:459-476) plus the Distribution tables
 * (core.pyx:23-45: xax, pdf, cdf, ppf of `size` doubles each).
 */
enum {
    NFA_PRIOR_SIMPLE = 0,             /* Prior                  core.pyx:169-197 */
    NFA_PRIOR_DUPLICATE = 1,          /* DuplicatePrior         core.pyx:200-221 */
    NFA_PRIOR_CONSTANT = 2,           /* ConstantPrior          core.pyx:224-238 */
    NFA_PRIOR_ORDERED = 3,            /* OrderedPrior           core.pyx:241-258 */
    NFA_PRIOR_SPACED = 4,             /* SpacedPrior            core.pyx:261-292 */
    NFA_PRIOR_CENSEP = 5,             /* CenSepPrior            core.pyx:295-318 */
    NFA_PRIOR_RESOLVED_CENSEP = 6,    /* ResolvedCenSepPrior    core.pyx:321-366 */
    NFA_PRIOR_RESOLVED_PLACEMENT = 7  /* ResolvedPlacementPrior core.pyx:369-435 */
};

typedef struct {
    int64_t size;
    double  du, dx, xmin, xmax;
    const double *xax, *pdf, *cdf, *ppf;
} nfa_dist_desc;

typedef struct {
    int32_t kind;
    int32_t p_ix;      /* parameter slot (vcen slot for the composite kinds) */
    int32_t p_ix2;     /* duplicate slot / sigm slot                         */
    int32_t dist0;     /* main / vcen / independent distribution             */
    int32_t dist1;     /* vsep / dependent distribution                      */
    int32_t dist2;     /* sigm distribution                                  */
    int32_t sub_kind;  /* kind of the sigm sub-prior (SIMPLE/CONSTANT/ORDERED) */
    int32_t pad_;
    double  value;     /* constant value                                     */
    double  sep_scale; /* FWHM * scale                                       */
} nfa_prior_desc;

int nfa_priors_create(nfa_priors **out, const nfa_prior_desc *priors, int n_prior,
                      const nfa_dist_desc *dists, int n_dist, int n_param);
int nfa_priors_destroy(nfa_priors *p);
/* PriorTransformer.transform (core.pyx:478-483) over B rows of U[B][n_param*ncomp],
 * host memory, in place.  NFA_ERR_ARG when ndim != n_param*ncomp. */
int nfa_priors_transform_batch(const nfa_priors *p, double *U, int64_t B,
                               int ncomp, int ndim);

/* ---- runner ---------------------------------------------------------------
 * Replaces AmmoniaRunner (nestfit/models/ammonia.pyx:369-447).  `priors` may be
 * NULL for predict-only use (amm_predict, ammonia.pyx:364-366).
 */
int nfa_runner_create(nfa_runner **out, nfa_specset *ss, nfa_priors *priors,
                      int ncomp, int cold, int lte);
int nfa_runner_destroy(nfa_runner *r);
int nfa_runner_ndim(const nfa_runner *r);
/* Numerical mode of this runner: -1 (default) = whatever nfa_set_exp_mode says when a batch is
 * launched; 0..2 = pinned, so that runners of different modes can work side by side (threads of a
 * broker, a table-mode checker next to a fast-mode sampler). */
int nfa_runner_set_exp_mode(nfa_runner *r, int mode);
int nfa_runner_get_exp_mode(const nfa_runner *r);

/* AmmoniaRunner.c_loglikelihood (ammonia.pyx:423-432) for B unit-cube rows of
 * pixel `pix[b]` (pix == NULL: pixel 0).  U[B][ndim] host memory, overwritten
 * in place with the physical parameters exactly like the reference mutates
 * `utheta`; lnL[B] out.  Synchronous.  Up to 128 rows take one launch (point kernel), more go through
 * the batch kernels with copies in and out; the values do not depend on the route. */
int nfa_runner_loglike_batch(nfa_runner *r, const int32_t *pix, double *U,
                             double *lnL, int64_t B);
/* Host buffers the device can address (pinned and mapped).  A U / lnL / pix buffer of
 * nfa_runner_loglike_batch that lies in such memory (from here, or registered with the HIP runtime by the
 * caller) is not copied: the kernels read the unit cube over the bus and write theta and lnL in place
 * (no reference counterpart: the reference's arrays never leave the host). */
int nfa_host_alloc(void **out, size_t bytes);
int nfa_host_free(void *p);

/* AmmoniaRunner.predict / amm_predict (ammonia.pyx:437-447, 364-366) for B
 * parameter rows theta[B][ndim] (no priors).  spectra_out[B][chan_tot] and/or
 * lnL_out[B] may be NULL; output buffers from nfa_host_alloc are written by the kernels themselves. */
int nfa_runner_predict_batch(nfa_runner *r, const int32_t *pix, const double *theta,
                             int64_t B, double *spectra_out, double *lnL_out);

/* Same as nfa_runner_loglike_batch with every buffer already resident in device
 * memory (from nfa_malloc); enqueued, returns without synchronising.  d_pix may be
 * NULL.  Consecutive calls of one shape may be launched together (option "coalesce");
 * consecutive launches go to different HIP streams of the runner (round robin) and may
 * overlap on the device: the buffers of calls that are in flight together must not
 * alias, and the caller's buffers must stay valid until nfa_runner_synchronize (or
 * nfa_device_synchronize), which launches what is still held and waits for all of it. */
int nfa_runner_loglike_batch_dev(nfa_runner *r, const int32_t *d_pix, double *d_U,
                                 double *d_lnL, int64_t B);
/* Spectra-out with device pointers (what AmmoniaRunner.predict is to deblend_hf_intensity /
 * generate_predicted_profiles, nestfit/main.py:1106-1113, 1182-1188, for a caller whose MAP cube and
 * profile cube stay in HBM): d_theta [B x ndim] physical parameters (read only), d_spectra
 * [B x chan_tot] or NULL, d_lnL [B] or NULL (not both NULL).  Asynchronous and rotating over the
 * runner's streams like nfa_runner_loglike_batch_dev, and like its batches consecutive calls of one
 * shape (same B, a multiple of 64; the same of d_pix / d_spectra / d_lnL given) travel as one launch,
 * every batch writing its own arrays (option "coalesce"; since round 5: a launch of one 4096-row batch is
 * as long as its longest wave); same results as nfa_runner_predict_batch. */
int nfa_runner_predict_batch_dev(nfa_runner *r, const int32_t *d_pix, const double *d_theta, int64_t B,
                                 double *d_spectra, double *d_lnL);
int nfa_runner_synchronize(nfa_runner *r);
/* Per-kernel timing of nfa_runner_loglike_batch_dev with HIP events recorded on
 * the stream each kernel is launched on, over the calls made since profiling was
 * switched on.  out[0], out[1]: summed milliseconds of the set-up kernel and of the
 * likelihood kernel (lnl_kernel alone: the interval rocprofv3 reports for it when the
 * runner has one stream lane); out[2], out[3]: milliseconds during which at least one set-up /
 * likelihood kernel was running (union of the launch intervals: with several stream
 * lanes launches overlap and the plain sum counts that time more than once). */
int nfa_runner_set_profiling(nfa_runner *r, int on);
int nfa_runner_get_profile(nfa_runner *r, double *out, int64_t *calls);

/* MultiNest `LogLike` (nestfit/core/cmultinest.pxd:27-28; the reference's
 * trampoline is mn_loglikelihood, nestfit/core/core.pyx:622-624).  Pass the
 * nfa_runner* as MultiNest's `context`. */
void nfa_loglike_callback(double *Cube, int *ndim, int *npars, double *lnew, void *ctx);

/* ---- callback-coalescing broker -------------------------------
 * MultiNest evaluates one point per LogLike call (cmultinest.pxd:27-28); the broker
 * gathers concurrent calls of many sampler threads into GPU batches.
 * nfa_broker_loglike blocks the calling thread until its batch has run: `cube` (ndim
 * doubles in the unit cube) is overwritten with the physical parameters and *lnew set,
 * exactly like mn_loglikelihood -> c_loglikelihood (core.pyx:622-624,
 * ammonia.pyx:423-432); pix < 0 = the runner's only pixel.  A generation's first caller
 * leads it: it launches when n_clients requests (0 = unknown) or max_batch are queued, or
 * after max_wait_us.  Results are bitwise those of nfa_runner_loglike_batch.
 * nfa_broker_callback has MultiNest's LogLike signature; its context is an
 * nfa_broker_client (broker + pixel).  The runner must not be used directly while a
 * broker serves it. */
typedef struct nfa_broker nfa_broker;
typedef struct { nfa_broker *broker; int32_t pix; } nfa_broker_client;
int  nfa_broker_create(nfa_broker **out, nfa_runner *r, int max_batch, int64_t max_wait_us, int n_clients);
int  nfa_broker_destroy(nfa_broker *b);
int  nfa_broker_set_clients(nfa_broker *b, int n_clients);
int  nfa_broker_loglike(nfa_broker *b, int32_t pix, double *cube, double *lnew);
void nfa_broker_callback(double *Cube, int *ndim, int *npars, double *lnew, void *ctx);
/* out[0] batches launched, out[1] evaluations served, out[2] largest batch */
int  nfa_broker_stats(nfa_broker *b, int64_t *out);

/* ---- cross-process transport for the broker --------------------
 * The reference runs one process per stripe (nestfit/main.py:516-523), each with its own
 * MultiNest (one instance per process) calling LogLike point by point
 * (cmultinest.pxd:27-28).  With a ring those processes never touch the GPU: each attaches to a
 * POSIX shared-memory ring (`name`), owns one slot, and nfa_ring_loglike / nfa_ring_callback
 * (MultiNest's LogLike signature; context = nfa_ring_client) post the point there and sleep
 * until the ONE serving process has evaluated it together with the other processes' points:
 * nfa_ring_serve = nfa_ring_poll -> nfa_runner_loglike_batch (up to 128 points: one launch;
 * up to 1024 through the batch kernels) -> nfa_ring_complete, until nfa_ring_stop, max_batches
 * (> 0) or idle_ms without a post.  A request that names a pixel the runner does not have fails
 * alone (NFA_ERR_ARG to its client); a device error stops the ring.
 * Results are bitwise those of nfa_runner_loglike_batch.  poll / complete are public so that a
 * server can put its own evaluator between them.  All entry points except nfa_ring_serve are
 * also exported by libnestfit_amd_ring.so, which has no HIP in it: the one library a sampler
 * process loads. */
typedef struct nfa_ring nfa_ring;
typedef struct { nfa_ring *ring; int32_t pix; } nfa_ring_client;
int  nfa_ring_create(nfa_ring **out, const char *name, int n_slots, int ndim);      /* server; one point per slot */
/* slots that hold up to max_points (<= 64) points each: for samplers that post several points per call */
int  nfa_ring_create_multi(nfa_ring **out, const char *name, int n_slots, int ndim, int max_points);
int  nfa_ring_attach(nfa_ring **out, const char *name, int wait_ms);                /* client: takes a slot */
int  nfa_ring_close(nfa_ring *r);            /* client: gives the slot back; server: removes the ring */
int  nfa_ring_stop(nfa_ring *r);             /* everybody leaves: blocked clients get NFA_ERR_STATE */
int  nfa_ring_ndim(const nfa_ring *r);
int  nfa_ring_slot(const nfa_ring *r);
int  nfa_ring_max_points(const nfa_ring *r);
/* blocking; NFA_ERR_STATE when the ring was stopped, its creator is gone, or no serving loop has been on it for 5 s */
int  nfa_ring_loglike(nfa_ring *r, int32_t pix, double *cube, double *lnew);
/* k points in one call (k <= max_points, all against pixel pix): cubes[k][ndim] in: unit cube, out: theta; lnew[k].
 * MultiNest asks for one point per call; a sampler whose next k proposals do not depend on each other's
 * likelihoods (uniform draws from the current bounding ellipsoid) posts them together and uses them in order */
int  nfa_ring_loglike_many(nfa_ring *r, int32_t pix, double *cubes, double *lnew, int k);
void nfa_ring_callback(double *Cube, int *ndim, int *npars, double *lnew, void *ctx);
/* slots[k], pix[k], U[k*ndim..]: row k of the *n gathered points (at most max_batch; the points of one
 * request are neighbouring rows with the same slot); returns when every attached client has posted, or
 * max_wait_us after the first post; *n = 0 after idle_ms without a post or when the ring was stopped
 * (*stopped).  nfa_ring_complete takes whole requests back, in one call or several. */
int  nfa_ring_poll(nfa_ring *r, int max_batch, int64_t max_wait_us, int idle_ms, int32_t *slots,
                   int32_t *pix, double *U, int *n, int *stopped);
int  nfa_ring_complete(nfa_ring *r, int n, const int32_t *slots, const double *U, const double *lnL, int rc);
int  nfa_ring_serve(nfa_ring *r, nfa_runner *run, int64_t max_wait_us, int64_t max_batches, int idle_ms);
/* The same service from a RESIDENT kernel (engine library only; one point per slot: rings made with nfa_ring_create).
 * The ring's mapping is registered with the runtime; the kernel's workgroups poll the slots themselves, claim a posted
 * point, run the point kernel's path and write theta, lnL and the DONE state into the slot the client spins on: a
 * LogLike call (mn_loglikelihood, nestfit/core/core.pyx:622-624) costs the path itself, no launch and no host thread.
 * Every kernel instance ends after lifetime_ms (0 = 20) or when the ring is stopped, and is launched again while there is
 * work; the call returns when the ring was stopped or nothing was served for idle_ms.  Bitwise nfa_runner_loglike_batch. */
int  nfa_ring_serve_device(nfa_ring *r, nfa_runner *run, int lifetime_ms, int idle_ms);
/* out[0] batches served, out[1] evaluations served, out[2] largest batch, out[3] clients attached */
int  nfa_ring_stats(nfa_ring *r, int64_t *out);
/* ---- device-resident batched nested sampler --------------------
 * Stand-in for one serial MultiNest run per pixel (run_multinest, nestfit/core/core.pyx:727-823,
 * pixel loop nestfit/main.py:452-469) when libmultinest is absent: all pixels' runs advance in
 * lock-step rounds with their state in HBM; a round = n_cand candidates per active pixel from
 * its bounding ellipsoid, one likelihood batch over all of them, one wave per pixel doing the
 * replace / evidence / stop / refit step.  pix[n_pix]: cube pixel of each run.  n_cand: candidates
 * per pixel and round at least; the number is raised (up to 16384) so that a round proposes about
 * batch_target candidates however few pixels are still running; only the proposals inside the
 * unit cube (the prior's support) are compacted and sent to the likelihood.  cap_iter: dead
 * point slots per pixel (a run stops when they are full).  free_mask[ndim] (NULL = all ones): unit-cube
 * slots the likelihood depends on; the others (constant or duplicated parameters) are not sampled --
 * integrating a uniform dummy dimension out exactly -- and stay at u = 0.5.  tol, efr, seed, maxiter as
 * run_multinest; upd = replacements between ellipsoid refits; log_zero replaces non-finite
 * likelihoods; check_every = rounds between two compactions of the active-pixel list.
 * Outputs are the raw material of what mn_dump stores (core.pyx:627-687): dead points with
 * their ln-weights, the final live points, iteration and evaluation counts; nestfit_amd/sampler.py
 * assembles posteriors / lnZ from them and holds the bit-compatible host twin of the algorithm. */
typedef struct nfa_sampler nfa_sampler;
int nfa_sampler_create(nfa_sampler **out, nfa_runner *r, const int32_t *pix, int64_t n_pix, int nlive,
                       int n_cand, int64_t batch_target, int64_t cap_iter, const int32_t *free_mask);
int nfa_sampler_destroy(nfa_sampler *s);
/* Optional, between create and begin: pixels with numbers of live points of their own (the cube driver gives every
 * pixel nlive + int(5 SNR), nestfit/main.py:445-447) in ONE lock-step group: nlive[p] <= the nlive of
 * nfa_sampler_create (which is then the stride of the live arrays nfa_sampler_live returns: pixel p's points are
 * the first nlive[p] of its slice), cap[p] <= cap_iter dead-point slots, upd[p] replacements between refits. */
int nfa_sampler_set_pixel_nlive(nfa_sampler *s, const int32_t *nlive, const int64_t *cap, const int32_t *upd);
/* Optional, between create and begin: bounding ellipsoids per pixel at most -- MultiNest's `mmodal` / `maxModes`
 * (nestfit/core/core.pyx:727-760) as far as this sampler has them: 1 = one ellipsoid around all live points
 * (mmodal = False), 2..4 = clusters of live points get ellipsoids of their own where at most six dimensions are
 * sampled, 0 = the default (4 there, 1 above). */
int nfa_sampler_set_ellipsoids(nfa_sampler *s, int max_ellipsoids);
/* Free rejections of a one-ellipsoid bound (between create and begin).  Above six sampled dimensions no ellipsoid bounds
 * the live region of a fit well; but every superset of the region may veto a proposal before its likelihood is evaluated:
 * the bounding boxes of the live points in the unit cube's axes, in the ellipsoid's own frame and in n_frames fixed
 * rotations of it (-2: the default = 32 where the bound is sheared, none elsewhere; -1: none; 0..64).  margin: a face lies
 * beyond the extreme live point by margin * max(0.1 s, extreme - mean - 1.5 s), s the spread along its direction
 * (0: the default, 1.75).  MultiNest has no counterpart; `efr` keeps its meaning for the ellipsoid the proposals are drawn
 * from (nestfit/core/core.pyx:727-732). */
int nfa_sampler_set_boxes(nfa_sampler *s, int n_frames, double margin);
/* A volume-preserving shear in front of a one-ellipsoid bound (between create and begin).  The live region of a faint
 * pixel is a curved ridge (tex against ntot) that an ellipsoid holds badly.  The bound is therefore fitted AFTER the map
 * w_j = z_j - q_j(z_0 .. z_j-1), z = (u - mean) / spread, q_j the least-squares quadratic of the live points (squares and,
 * inside one velocity component, products of the earlier coordinates): an additive triangular map has a unit Jacobian, so
 * a point drawn uniformly in the w-ellipsoid and mapped back is uniform over its curved image in the unit cube.  Boxes, if
 * on, are fitted and tested in the w frame.  enlarge: the safety factor on the sheared ellipsoid's enclosing volume (>= 1;
 * 2.5 is the measured choice), 0 = off, < 0 = the default (engine option "sampler_shear_pct": 2.5 unless changed).  Applies
 * where all five free parameters of two or three components are sampled (10 or 15 dimensions); accepted and without effect
 * elsewhere.  MultiNest has no counterpart (its answer to curved regions is more ellipsoids: nestfit/core/core.pyx:727-760). */
int nfa_sampler_set_shear(nfa_sampler *s, double enlarge);
/* With the shear and the boxes on: every pair (i, j) of the sheared coordinates has the bounding ellipse of the live points'
 * projection onto (w_i, w_j) -- the covariance ellipse scaled to enclose them, its area times `enlarge` -- as one more free
 * veto: the region lies inside the cylinder over each of its projections.  enlarge >= 1 (1.75 is the measured choice), 0 = off,
 * < 0 = the default (engine option "sampler_pairs_pct", hundredths).  Between create and begin. */
int nfa_sampler_set_pairs(nfa_sampler *s, double enlarge);
/* After a run: every pixel's table of posterior samples in one copy (what MultiNest leaves in its post files and mn_dump
 * reads back, nestfit/core/core.pyx:627-687) -- rows [offsets[p], offsets[p+1]) of out[offsets[P]][ndim + 2] are pixel p's
 * dead points (min(n_iter[p], cap) of them) followed by its live points; a row = theta[ndim], -2 lnL, ln(prior mass x
 * likelihood): lnw + lnL of a dead point, lnL + live_off[p] of a live one (live_off[p] = -n_iter / nlive - ln nlive, given by
 * the caller, who normalises the last column into weights) -- or, with stats[P][6 + 4 ndim] given, the device does: the
 * last column comes back as weights exp(. - lnZ) and stats[p] = lnZ, lnZ of the dead points alone, information H, largest
 * lnL, largest lnL of the live points, sum of the weights, then weighted mean[ndim], weighted raw second moment[ndim],
 * theta of the largest likelihood[ndim], theta of the largest weight[ndim] (what mn_dump derives from MultiNest's files). */
int nfa_sampler_posterior_packed(nfa_sampler *s, const int64_t *offsets, const double *live_off, double *out, double *stats);
int nfa_sampler_run(nfa_sampler *s, double tol, double efr, int64_t seed, int64_t maxiter, int upd,
                    double log_zero, int check_every);
/* method: how a pixel finds its next point above the threshold.  0 = rejection sampling in the
 * bounding ellipsoid only; 1 = a pixel whose rejection round accepted fewer than 1 in 2 n_steps of
 * the evaluated candidates switches to constrained random walks: 64 walkers start from random live
 * points and take n_steps Metropolis steps inside {L > threshold}; a step is a scaled difference of
 * two random live points (differential evolution, ter Braak 2006), the scale tuned to an acceptance
 * of one half;
 * 2 = walks from the first round.  nfa_sampler_run uses method 1, n_steps 10 x sampled dimensions
 * (walks that are too short bias lnZ upwards: +0.13 with 40 steps in 10 dimensions, +0.014 with 120).
 * (nfa_sampler_run uses enlarge = 1.5: safety factor on the volume of the ellipsoid that just
 * encloses the live points, before MultiNest's floor X / efr is applied)
 * the same in two steps: begin = live points + first ellipsoids; advance = up to max_chunks groups
 * of check_every rounds (0 = to the end); *n_active = pixels still running (progress, time limits) */
int nfa_sampler_begin(nfa_sampler *s, double tol, double efr, int64_t seed, int64_t maxiter, int upd,
                      double log_zero, int check_every, double enlarge, int method, int n_steps);
int nfa_sampler_advance(nfa_sampler *s, int64_t max_chunks, int64_t *n_active);
int nfa_sampler_counts(nfa_sampler *s, int64_t *n_iter, int64_t *n_evals, int64_t *rounds);
int nfa_sampler_dead(nfa_sampler *s, int64_t p, int64_t n, double *theta, double *lnL, double *lnw);
/* The same for all pixels in one call: offsets[P + 1] with offsets[0] = 0 and offsets[p + 1] - offsets[p] the
 * number of dead points wanted of pixel p (at most min(n_iter[p], cap_iter)); rows offsets[p] .. offsets[p + 1]
 * of theta[.][ndim], lnL[.], lnw[.] are pixel p's. */
int nfa_sampler_dead_packed(nfa_sampler *s, const int64_t *offsets, double *theta, double *lnL, double *lnw);
int nfa_sampler_live(nfa_sampler *s, double *theta, double *lnL);

/* ---- the exchange step of a sharded cube fit ---------------------
 * The reference stripes pixels over processes, (lon_ix[i::nproc], lat_ix[i::nproc])
 * (nestfit/main.py:565-571), and exchanges nothing while sampling; its processes meet through chunk
 * files (main.py:516-523, docs/store_spec.rst:12-32).  One process per GPU here; these entry points
 * are the end-of-run exchange: RCCL over xGMI (librccl.so, opened at run time).  Rank 0 calls
 * nfa_comm_unique_id and the host carries the 128 bytes to the other ranks; every rank then calls
 * nfa_comm_create (after nfa_set_device).  All buffers are host memory; calls are blocking.
 * op: 0 sum, 2 max, 3 min. */
typedef struct nfa_comm nfa_comm;
int nfa_comm_unique_id(unsigned char *id128);
int nfa_comm_create(nfa_comm **out, const unsigned char *id128, int rank, int world);
int nfa_comm_destroy(nfa_comm *c);
int nfa_comm_rank(const nfa_comm *c);
int nfa_comm_world(const nfa_comm *c);
int nfa_comm_allgather(nfa_comm *c, const double *send, int64_t count, double *recv /* [world][count] */);
int nfa_comm_allreduce(nfa_comm *c, double *values, int64_t count, int op);
int nfa_comm_barrier(nfa_comm *c);

/* ---- device memory + events (for harnesses that keep inputs in HBM) ------- */
int nfa_malloc(void **dptr, int64_t bytes);
int nfa_free(void *dptr);
int nfa_memcpy_h2d(void *dst, const void *src, int64_t bytes);
int nfa_memcpy_d2h(void *dst, const void *src, int64_t bytes);
int nfa_memcpy_d2d(void *dst, const void *src, int64_t bytes);
int nfa_event_create(void **ev);
int nfa_event_destroy(void *ev);
int nfa_event_record(void *ev, nfa_runner *r);       /* on the runner's stream */
int nfa_event_synchronize(void *ev);
int nfa_event_elapsed_ms(void *start, void *stop, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* NESTFIT_AMD_H */
