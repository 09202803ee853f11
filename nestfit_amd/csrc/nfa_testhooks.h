// nfa_testhooks.h -- unit-test and measurement hooks: device evaluation of the scalar building blocks
// (FastExp, iemtex, partition sums, hyperfine windows) and a native-thread storm on a broker.
// Compiled only into libnestfit_amd_test.so (-DNFA_TEST_HOOKS -DNFA_ABLATE, nestfit_amd/build.py); the
// product library carries none of it.  Declarations: include/nestfit_amd_test.h.
#pragma once
#include <chrono>
#include <thread>
#include "../../include/nestfit_amd_test.h"

// ---------------------------------------------------------------------------
//  unit-test kernels (device evaluation of the scalar building blocks)
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void test_fastexp_kernel(const double *x, double *out, long n, const double *g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int n_shared;
    const double *sm = stage_exp_tables<MODE == 2 ? 1 : MODE>(smem, g_tabs, &n_shared);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ((n + 63) & ~63L);
         i += (long)gridDim.x * blockDim.x) {
        const double xi = i < n ? x[i] : 1.0;
        double v;
        if (MODE == 2) v = (double)exp_neg_f32((float)xi);
        else v = nf_fastexp<MODE == 2 ? 1 : MODE>(xi, sm);
        if (i < n) out[i] = v;
    }
}

// 1 - FastExp(tau) as the fast mode evaluates it
__global__ void test_one_minus_fastexp_kernel(const double *x, double *out, long n) {
    // whole waves walk the array (the function sets the EXEC mask itself and wants it full on entry)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ((n + 63) & ~63L); i += (long)gridDim.x * blockDim.x) {
        const double v = one_minus_fastexp_f32((float)(i < n ? x[i] : 1.0));
        if (i < n) out[i] = v;
    }
}

__global__ void test_iemtex_kernel(const double *x, double *out, long n, const double *g_tabs,
                                   double xmin, double xmax, double inv_dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        out[i] = nf_iemtex(x[i], g_tabs + SM_T0X, g_tabs + SM_T0Y, xmin, xmax, inv_dx);
}

template <int MODE>
__global__ void test_partition_kernel(const double *trot, double *qpara, double *qorth, long n,
                                      const double *g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int n_shared;
    const double *sm = stage_exp_tables<MODE>(smem, g_tabs, &n_shared);
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
    const LineConst lc = nf_line(t, i, voff / NFA_CKMS, sigm / NFA_CKMS, S.rest[s], S.nu_min[s], S.nu_chan[s],
                                 S.size[s]);
    lo[i] = lc.lo; hi[i] = lc.hi;
}

extern "C" {

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
        const size_t lds = sizeof(double) * (SM_END_TABLE - SM_EXP2 + SM_TABLE_TAIL);
        HIP_TRY(hipFuncSetAttribute((const void *)test_fastexp_kernel<0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(test_fastexp_kernel<0>, dim3(blocks), dim3(256), lds, 0, dx, dout, (long)n,
                           (const double *)g_eng.d_tabs);
    } else if (mode == 1) {
        hipLaunchKernelGGL(test_fastexp_kernel<1>, dim3(blocks), dim3(256), sizeof(double) * NFA_EXP2_N, 0,
                           dx, dout, (long)n, (const double *)g_eng.d_tabs);
    } else if (mode == 2) {
        hipLaunchKernelGGL(test_fastexp_kernel<2>, dim3(blocks), dim3(256), sizeof(double) * NFA_EXP2_N, 0,
                           dx, dout, (long)n, (const double *)g_eng.d_tabs);
    } else {   // 3: 1 - FastExp(x) as the fast mode's Tb pass evaluates it
        hipLaunchKernelGGL(test_one_minus_fastexp_kernel, dim3(blocks), dim3(256), 0, 0, dx, dout, (long)n);
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
    hipLaunchKernelGGL(test_iemtex_kernel, dim3(blocks), dim3(256), 0, 0, dx, dout,
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
        const size_t lds = sizeof(double) * (SM_END_TABLE - SM_EXP2 + SM_TABLE_TAIL);
        HIP_TRY(hipFuncSetAttribute((const void *)test_partition_kernel<0>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(test_partition_kernel<0>, dim3(blocks), dim3(256), lds, 0, dt, dp, dq, (long)n,
                           (const double *)g_eng.d_tabs);
    } else {
        hipLaunchKernelGGL(test_partition_kernel<1>, dim3(blocks), dim3(256), sizeof(double) * NFA_EXP2_N, 0,
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
    const int tg = r->ss->dev.trans[spec] - 1;          // index into the combined tables
    const int nhf = tg < NFA_T_N2HP ? nfa_nhf[tg] : tg < NFA_T_GAUSS ? nfa_n2hp_nhf[tg - NFA_T_N2HP] : 1;
    HIP_TRY(hipMalloc(&dl, sizeof(int) * 64));
    HIP_TRY(hipMalloc(&dh, sizeof(int) * 64));
    hipLaunchKernelGGL(test_windows_kernel, dim3(1), dim3(64), 0, 0, r->ss->dev, spec, voff, sigm, dl, dh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(lo, dl, sizeof(int) * nhf, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hi, dh, sizeof(int) * nhf, hipMemcpyDeviceToHost));
    (void)hipFree(dl); (void)hipFree(dh);
    return NFA_OK;
}


// Measurement / test support: n_threads native threads, each a stand-in for one serial sampler,
// make n_calls blocking broker calls (`loglike` = the address of the PRODUCT library's
// nfa_broker_loglike, whose broker `b` is) on their own rows of U[n_threads][n_calls][ndim]
// (overwritten with theta); lnL[n_threads][n_calls].  Returns the wall time in *seconds_out.
int nfa_test_broker_storm(nfa_broker *b, nfa_broker_loglike_fn loglike, int n_threads, int n_calls,
                          const int32_t *pix, double *U, double *lnL, double *seconds_out) {
    if (!b || !loglike || !U || !lnL || n_threads < 1 || n_calls < 1) return fail(NFA_ERR_ARG, "bad argument");
    const int ndim = b->r->ndim;
    std::vector<std::thread> th;
    std::vector<int> rcs((size_t)n_threads, NFA_OK);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n_threads; ++k)
        th.emplace_back([=, &rcs] {
            for (int j = 0; j < n_calls; ++j) {
                const size_t row = (size_t)k * n_calls + j;
                const int rc = loglike(b, pix ? pix[k] : -1, U + row * ndim, lnL + row);
                if (rc != NFA_OK) rcs[k] = rc;
            }
        });
    for (auto &t : th) t.join();
    if (seconds_out)
        *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int rc : rcs) if (rc != NFA_OK) return rc;
    return NFA_OK;
}

int nfa_test_callback_latency(nfa_loglike_callback_fn callback, void *runner, int ndim, const double *u,
                              int n_calls, double *lnew_out, double *seconds_out) {
    if (!callback || !runner || !u || ndim < 1 || ndim > 64 || n_calls < 1) return fail(NFA_ERR_ARG, "bad argument");
    double cube[64], lnew = 0.0;
    int nd = ndim, npars = ndim;
    const auto t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < n_calls; ++j) {
        memcpy(cube, u, sizeof(double) * ndim);
        callback(cube, &nd, &npars, &lnew, runner);
    }
    if (seconds_out)
        *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (lnew_out) *lnew_out = lnew;
    return NFA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
//  timeline of the queue form of the table-mode likelihood kernel (lnl_kernel_queue): with a buffer attached every
//  wave of a launch records {start, end, item * nspec + spectrum, position in the order} of up to 8 units, in ticks of
//  10 ns (s_memrealtime); the buffer holds the last launch
// ---------------------------------------------------------------------------
#define NFA_TRACE_WAVES 8192
int nfa_test_queue_trace(int on) {
    int rc0 = engine_init(); if (rc0) return rc0;
    if (on && !g_eng.d_trace) {
        HIP_TRY(hipMalloc(&g_eng.d_trace, sizeof(unsigned long long) * NFA_TRACE_WAVES * 8 * 4));
    }
    if (on) HIP_TRY(hipMemset(g_eng.d_trace, 0, sizeof(unsigned long long) * NFA_TRACE_WAVES * 8 * 4));
    if (!on && g_eng.d_trace) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(g_eng.d_trace); g_eng.d_trace = nullptr; }
    return NFA_OK;
}
int nfa_test_queue_trace_read(unsigned long long *out) {
    if (!g_eng.d_trace) return fail(NFA_ERR_STATE, "no trace buffer");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, g_eng.d_trace, sizeof(unsigned long long) * NFA_TRACE_WAVES * 8 * 4, hipMemcpyDeviceToHost));
    return NFA_OK;
}
