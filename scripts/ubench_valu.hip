// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU
// instructions the likelihood kernel is made of, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define R8(B) B B B B B B B B
typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, T, BODY)                                                         \
    __global__ void __launch_bounds__(256) NAME(double *out, const double *in) {      \
        T a0 = (T)in[0], a1 = (T)in[1], a2 = (T)in[2], a3 = (T)in[3];                 \
        const T c = (T)in[4], d = (T)in[5];                                           \
        for (int it = 0; it < ITER; ++it) { R8(BODY) }                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + c * d); \
    }

#define B_FMA32 a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
#define B_FMA64 a0 = __builtin_fma(a0, c, d); a1 = __builtin_fma(a1, c, d); a2 = __builtin_fma(a2, c, d); a3 = __builtin_fma(a3, c, d);
#define B_FMA64D a0 = __builtin_fma(a0, c, d); a0 = __builtin_fma(a0, c, d); a0 = __builtin_fma(a0, c, d); a0 = __builtin_fma(a0, c, d);
#define B_MUL a0 = a0 * c; a1 = a1 * c; a2 = a2 * c; a3 = a3 * c;
#define B_ADD a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c;
#define B_EXP a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
#define B_CVT a0 = (double)(float)a0; a1 = (double)(float)a1; a2 = (double)(float)a2; a3 = (double)(float)a3; asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#define B_CND a0 = (a1 > c) ? a2 : a0; a1 = (a2 > c) ? a3 : a1; a2 = (a3 > c) ? a0 : a2; a3 = (a0 > c) ? a1 : a3;
#define B_INT a0 = (a0 + c) ^ a1; a1 = (a1 + c) ^ a2; a2 = (a2 + c) ^ a3; a3 = (a3 + c) ^ a0;
#define B_RCP a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
#define B_RND a0 = __builtin_rint(a0) + c; a1 = __builtin_rint(a1) + c; a2 = __builtin_rint(a2) + c; a3 = __builtin_rint(a3) + c;

KERNEL(k_fma32, float, B_FMA32)
KERNEL(k_mul32, float, B_MUL)
KERNEL(k_fma64, double, B_FMA64)
KERNEL(k_fma64d, double, B_FMA64D)
KERNEL(k_mul64, double, B_MUL)
KERNEL(k_add64, double, B_ADD)
KERNEL(k_exp32, float, B_EXP)
KERNEL(k_cvt, double, B_CVT)
KERNEL(k_cnd32, float, B_CND)
KERNEL(k_cnd64, double, B_CND)
KERNEL(k_int, int, B_INT)
KERNEL(k_rcp64, double, B_RCP)
KERNEL(k_rnd64, double, B_RND)

__global__ void __launch_bounds__(256) k_pkfma(double *out, const double *in) {
    f2 a0 = {(float)in[0], (float)in[1]}, a1 = {(float)in[2], (float)in[3]}, a2 = a0 + 1.f, a3 = a1 + 1.f;
    const f2 c = {(float)in[4], (float)in[4]}, d = {(float)in[5], (float)in[5]};
    for (int it = 0; it < ITER; ++it) {
        R8(a0 = __builtin_elementwise_fma(a0, c, d); a1 = __builtin_elementwise_fma(a1, c, d);
           a2 = __builtin_elementwise_fma(a2, c, d); a3 = __builtin_elementwise_fma(a3, c, d);)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y;
}

// LDS broadcast reads (same address in every lane), the pattern of the line table
__global__ void __launch_bounds__(256) k_ldsb128(double *out, const double *in) {
    __shared__ double4 t[64];
    t[threadIdx.x & 63] = make_double4(in[0], in[1], in[2], in[3]);
    __syncthreads();
    double acc = 0;
    int idx = (int)in[6];
    for (int it = 0; it < ITER; ++it) {
        R8({ const double4 v = t[idx & 63]; acc += v.x; idx += (int)v.w + 1; })
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(double *, const double *);
static void run(const char *name, kern_t kf, int ops_per_body, double *d_out, double *d_in, int w) {
    const int blocks = 256 * w;               // 256 CUs x 4 SIMDs: w waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, d_out, d_in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, d_out, d_in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)ITER * 8 * ops_per_body * w;
    printf("%-18s w=%d %8.3f ms  %6.2f cyc/inst/SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    double *d_out, *d_in; hipMalloc(&d_out, 8 * 256 * 256 * 16); hipMalloc(&d_in, 64);
    double h[8] = {1.0000001, 0.9999999, 1.0000002, 0.9999998, 1.0000001, 1e-9, 0, 0};
    hipMemcpy(d_in, h, 64, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 8}) {
        run("v_fma_f32", k_fma32, 4, d_out, d_in, w);
        run("v_mul_f32", k_mul32, 4, d_out, d_in, w);
        run("v_pk_fma_f32", k_pkfma, 4, d_out, d_in, w);
        run("v_fma_f64", k_fma64, 4, d_out, d_in, w);
        run("v_fma_f64 dep", k_fma64d, 4, d_out, d_in, w);
        run("v_mul_f64", k_mul64, 4, d_out, d_in, w);
        run("v_add_f64", k_add64, 4, d_out, d_in, w);
        run("v_exp_f32", k_exp32, 4, d_out, d_in, w);
        run("cvt f64>f32>f64", k_cvt, 8, d_out, d_in, w);
        run("cmp+cnd b32", k_cnd32, 8, d_out, d_in, w);
        run("cmp f64+2cnd", k_cnd64, 12, d_out, d_in, w);
        run("add+xor i32", k_int, 8, d_out, d_in, w);
        run("v_rcp_f64", k_rcp64, 4, d_out, d_in, w);
        run("rndne+add f64", k_rnd64, 8, d_out, d_in, w);
        run("ds_read_b128 bcast", k_ldsb128, 3, d_out, d_in, w);
    }
    return 0;
}
