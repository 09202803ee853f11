/* nestfit_amd_test.h -- entry points of libnestfit_amd_test.so only: the engine built with
 * -DNFA_TEST_HOOKS -DNFA_ABLATE (unit-test hooks, the "ablate" timing option).  The product library
 * libnestfit_amd.so exports none of these.  The test library is a second, independent instance of the
 * engine (its own tables and options); handles created by the product library may be passed to the hooks
 * below that take one (they only read the handle's device pointers).
 */
#ifndef NESTFIT_AMD_TEST_H
#define NESTFIT_AMD_TEST_H

#include "nestfit_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* device evaluation of the scalar building blocks */
int nfa_test_fastexp(const double *x, double *out, int64_t n, int mode);    /* fastexp.c:234-283 via math.pxd:17; mode 3: 1 - FastExp(x) of the fast mode */
int nfa_test_iemtex(const double *x, double *out, int64_t n);               /* hyperfine.pyx:23-45 */
int nfa_test_partition(const double *trot, double *qpara, double *qorth, int64_t n); /* ammonia.pyx:304-315 */
int nfa_test_windows(nfa_runner *r, int spec, double voff, double sigm,
                     int32_t *lo, int32_t *hi);                             /* hyperfine.pyx:70-93 */

/* n_threads native threads (one per would-be serial sampler, thread k bound to pixel pix[k], NULL = the
 * only pixel) each make n_calls blocking calls of `loglike` -- the address of the product library's
 * nfa_broker_loglike, whose broker `b` is -- on their own rows of U[n_threads][n_calls][ndim]
 * (overwritten); lnL[n_threads][n_calls]; wall time out. */
typedef int (*nfa_broker_loglike_fn)(nfa_broker *b, int32_t pix, double *cube, double *lnew);
int nfa_test_broker_storm(nfa_broker *b, nfa_broker_loglike_fn loglike, int n_threads, int n_calls,
                          const int32_t *pix, double *U, double *lnL, double *seconds_out);

/* n_calls serial calls of `callback` -- the address of the product library's nfa_loglike_callback, the
 * LogLike a serial MultiNest would be handed (wrapped.pyx:56-99), with `runner` as its context -- on copies
 * of the unit-cube point u[ndim]: the per-call latency of the drop-in with no host language in the way. */
typedef void (*nfa_loglike_callback_fn)(double *cube, int *ndim, int *npars, double *lnew, void *ctx);
int nfa_test_callback_latency(nfa_loglike_callback_fn callback, void *runner, int ndim, const double *u,
                              int n_calls, double *lnew_out, double *seconds_out);

/* Timeline of the queue form of the table-mode likelihood kernel (csrc/nfa_device.h, lnl_kernel_queue): on = 1 attaches
 * a buffer (and clears it), 0 frees it; with it attached every wave of a launch records up to 8 units as
 * {start, end, item * nspec + spectrum, position in the launch's order}, ticks of 10 ns.  `out` takes
 * 8192 waves x 8 records x 4 words of the last launch. */
int nfa_test_queue_trace(int on);
int nfa_test_queue_trace_read(unsigned long long *out);

#ifdef __cplusplus
}
#endif
#endif /* NESTFIT_AMD_TEST_H */
