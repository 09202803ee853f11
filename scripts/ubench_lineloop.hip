// Micro-benchmark of candidate inner loops of lnl_kernel (hyperfine-line loop, fast mode):
// cycles per (line x row) iteration per SIMD at 8 waves per SIMD, with synthetic line records.
//   VAR 0  round-1 form: 32-B records broadcast-read from LDS, window test as a predicated weight
//   VAR 1  records through scalar loads (SGPRs), lane mask of the window built on the scalar unit,
//          applied as EXEC
//   VAR 2  records through scalar loads, window test on the vector unit (v_sub + v_cmp -> EXEC)
//   VAR 3  records in LDS as two 16-byte arrays (s_lshl4_add addressing), window applied as EXEC by the
//          compiler (v_cmp + s_and_saveexec), do-while loop on a 32-bit mask
//   VAR 4  the same with the masked body as one asm block (v_cmpx ... s_mov exec,-1)
//   then pure SALU chains (throughput of the scalar unit per CU)
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/ubench_lineloop.hip -o scripts/ubench_lineloop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) v8i *crec_t;
#define NLINES 21
#define NROWS  16

__device__ __forceinline__ double mkd(int lo, int hi) { return __hiloint2double(hi, lo); }
__device__ __forceinline__ unsigned long long lanemask(int a, int b) {   // lanes [a, b) of 0..63
    a = a < 0 ? 0 : a;
    b = b > 64 ? 64 : b;
    return (~0ull << a) & (~0ull >> (64 - b));
}
__device__ __forceinline__ float exp_neg(float xf) {
    const float yh = xf * -1.44269502162933349609375f;
    float r = __builtin_fmaf(yh, -0.693147182464599609375f, -xf);
    r = __builtin_fmaf(yh, 1.904654299957e-09f, r);
    const float e0 = __builtin_amdgcn_exp2f(yh);
    return __builtin_fmaf(e0, r, e0);
}

struct __attribute__((aligned(16))) LineRec { double nucen, idenom; float htau; int pad; int lo, len; };

template <int VAR>
__global__ void __launch_bounds__(256) k_loop(const v8i *__restrict__ recs, const unsigned *__restrict__ rowmask,
                                              const double *__restrict__ x, float *__restrict__ out, int reps) {
    __shared__ LineRec lds[4][NLINES + 3];
    __shared__ __attribute__((aligned(16))) char lds2[4][2048];          // [A: 64 x 16 B][B: 64 x 16 B]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long unit = (long)blockIdx.x * 4 + wave;
    const v8i *mine = recs + unit * 32;
    if (VAR == 0) {
        if (lane < NLINES) {
            const v8i r = mine[lane];
            LineRec q;
            q.nucen = mkd(r[0], r[1]); q.idenom = mkd(r[2], r[3]); q.htau = __int_as_float(r[4]); q.pad = 0;
            q.lo = r[5]; q.len = r[6] - r[5];
            lds[wave][lane] = q;
        }
        __syncthreads();
    }
    if (VAR == 3 || VAR == 4) {
        if (lane < NLINES) {
            const v8i r = mine[lane];
            *(double2 *)(lds2[wave] + lane * 16) = make_double2(mkd(r[0], r[1]), mkd(r[2], r[3]));
            *(int4 *)(lds2[wave] + 1024 + lane * 16) = make_int4(r[4], r[5], r[6] - r[5], 0);
        }
        __syncthreads();
    }
    crec_t R = (crec_t)mine;
    float total = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        for (int row = 0; row < NROWS; ++row) {
            const int r0 = row * 64;
            const double xj = x[(unit & 255) * 1024 + r0 + lane];
            unsigned mask = __builtin_amdgcn_readfirstlane(rowmask[(unit & 1023) * NROWS + row]);
            float tau = 0.f;
            if (VAR == 0) {
                while (mask) {
                    const int i = __builtin_ctz(mask);
                    mask &= mask - 1;
                    const LineRec rec = lds[wave][i];
                    const double nu = xj - rec.nucen;
                    const float xf = (float)(nu * nu * rec.idenom);
                    const float e = exp_neg(xf);
                    const bool inwin = (unsigned)(r0 + lane - rec.lo) < (unsigned)rec.len;
                    const float h = inwin ? rec.htau : 0.0f;
                    tau = __builtin_fmaf(h, e, tau);
                }
            } else if (VAR == 1 || VAR == 2) {
                if (mask) {
                    v8i nxt = R[__builtin_ctz(mask)];
                    while (true) {
                        const v8i rec = nxt;
                        mask &= mask - 1;
                        if (mask) nxt = R[__builtin_ctz(mask)];
                        bool in;
                        if (VAR == 1) in = __builtin_amdgcn_inverse_ballot_w64(lanemask(rec[5] - r0, rec[6] - r0));
                        else in = (unsigned)(lane - (rec[5] - r0)) < (unsigned)(rec[6] - rec[5]);
                        if (in) {
                            const double nu = xj - mkd(rec[0], rec[1]);
                            const float xf = (float)(nu * nu * mkd(rec[2], rec[3]));
                            tau = __builtin_fmaf(__int_as_float(rec[4]), exp_neg(xf), tau);
                        }
                        if (!mask) break;
                    }
                }
            }
            if (VAR == 3 || VAR == 4) {
                if (mask) {
                    const char *base = lds2[wave];
                    const int j = r0 + lane;
                    do {
                        const int i = __builtin_ctz(mask);
                        const char *p = base + (i << 4);
                        double2 ab = *(const double2 *)p;
                        int4 q = *(const int4 *)(p + 1024);
                        mask &= mask - 1;
                        if (VAR == 3) {
                            asm volatile("" : "+v"(ab.x), "+v"(ab.y), "+v"(q.x));
                            if ((unsigned)(j - q.y) < (unsigned)q.z) {
                                asm volatile("" ::: "memory");
                                const double nu = xj - ab.x;
                                const float xf = (float)(nu * nu * ab.y);
                                tau = __builtin_fmaf(__int_as_float(q.x), exp_neg(xf), tau);
                            }
                        } else {
                            double d; float t0, t1, t2;
                            asm volatile("v_sub_u32 %[t0], %[j], %[lo]\n\t"
                                         "v_cmpx_lt_u32 %[t0], %[len]\n\t"
                                         "v_add_f64 %[d], %[xj], -%[nucen]\n\t"
                                         "v_mul_f64 %[d], %[d], %[d]\n\t"
                                         "v_mul_f64 %[d], %[d], %[idenom]\n\t"
                                         "v_cvt_f32_f64 %[t0], %[d]\n\t"
                                         "v_mul_f32 %[t1], 0xbfb8aa3b, %[t0]\n\t"
                                         "v_exp_f32 %[t2], %[t1]\n\t"
                                         "v_fma_f32 %[t0], %[t1], %[kln2], -%[t0]\n\t"
                                         "v_fmac_f32 %[t0], 0x3102e308, %[t1]\n\t"
                                         "v_fmac_f32 %[t2], %[t2], %[t0]\n\t"
                                         "v_fmac_f32 %[tau], %[htau], %[t2]\n\t"
                                         "s_mov_b64 exec, -1"
                                         : [tau] "+v"(tau), [d] "=&v"(d), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
                                         : [j] "v"(j), [lo] "v"(q.y), [len] "v"(q.z), [xj] "v"(xj), [nucen] "v"(ab.x),
                                           [idenom] "v"(ab.y), [htau] "v"(q.x), [kln2] "s"(-0.693147182464599609375f)
                                         : "vcc");
                        }
                    } while (mask);
                }
            }
            total += tau;
        }
    }
    out[unit * 64 + lane] = total;
}

// scalar-unit throughput: independent s_add chains, 8 per loop trip
__global__ void __launch_bounds__(256) k_salu(int *out, int n) {
    int a = __builtin_amdgcn_readfirstlane(n), b = a + 1, c = a + 2, d = a + 3;
    for (int i = 0; i < n; ++i) {
        asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                     "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                     "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                     "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d) :: "scc");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = a + b + c + d;
}
// the same beside a VALU stream in the same wave: 16 SALU + 8 v_fma_f32 per trip
__global__ void __launch_bounds__(256) k_salu_valu(int *out, int n, float f) {
    int a = __builtin_amdgcn_readfirstlane(n), b = a + 1, c = a + 2, d = a + 3;
    float v0 = f, v1 = f + 1, v2 = f + 2, v3 = f + 3;
    for (int i = 0; i < n; ++i) {
        asm volatile("s_add_u32 %0, %0, 3\n v_fma_f32 %4, %4, %4, %4\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n v_fma_f32 %5, %5, %5, %5\n s_add_u32 %3, %3, 9\n"
                     "s_add_u32 %0, %0, 3\n v_fma_f32 %6, %6, %6, %6\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n v_fma_f32 %7, %7, %7, %7\n s_add_u32 %3, %3, 9\n"
                     "s_add_u32 %0, %0, 3\n v_fma_f32 %4, %4, %4, %4\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n v_fma_f32 %5, %5, %5, %5\n s_add_u32 %3, %3, 9\n"
                     "s_add_u32 %0, %0, 3\n v_fma_f32 %6, %6, %6, %6\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n v_fma_f32 %7, %7, %7, %7\n s_add_u32 %3, %3, 9\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "scc");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = a + b + c + d + (int)(v0 + v1 + v2 + v3);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int units = 8192;                       // 8 waves on every SIMD
    std::vector<v8i> recs((size_t)units * 32);
    std::vector<unsigned> rowmask(1024 * NROWS, 0u);
    std::vector<double> x(256 * 1024);
    long iters = 0;
    for (int u = 0; u < 1024; ++u) {
        srand(1 + u);
        // windows of ~90 channels at random places: each line touches 2-3 rows
        for (int i = 0; i < NLINES; ++i) {
            const int lo = rand() % (1024 - 100), len = 60 + rand() % 60;
            for (int r = lo / 64; r <= (lo + len - 1) / 64; ++r) rowmask[u * NROWS + r] |= 1u << i;
        }
    }
    for (int u = 0; u < units; ++u) {
        srand(1 + (u & 1023));
        for (int i = 0; i < NLINES; ++i) {
            const int lo = rand() % (1024 - 100), len = 60 + rand() % 60;
            const double nucen = 2.3e10 + (lo + len * 0.5) * 2300.0, idenom = 12.5 / ((len * 0.5 * 2300.0) * (len * 0.5 * 2300.0));
            v8i r;
            int2 a = *(const int2 *)&nucen, b = *(const int2 *)&idenom;
            const float h = 0.1f;
            r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y; r[4] = *(const int *)&h; r[5] = lo; r[6] = lo + len; r[7] = 0;
            recs[(size_t)u * 32 + i] = r;
        }
    }
    for (int u = 0; u < 1024; ++u) for (int r = 0; r < NROWS; ++r) iters += __builtin_popcount(rowmask[u * NROWS + r]);
    const double iters_per_unit = (double)iters / 1024;
    for (int p = 0; p < 256; ++p) for (int j = 0; j < 1024; ++j) x[p * 1024 + j] = 2.3e10 + j * 2300.0;
    v8i *d_recs; unsigned *d_mask; double *d_x; float *d_out; int *d_i;
    hipMalloc(&d_recs, recs.size() * sizeof(v8i)); hipMalloc(&d_mask, rowmask.size() * 4);
    hipMalloc(&d_x, x.size() * 8); hipMalloc(&d_out, (size_t)units * 64 * 4); hipMalloc(&d_i, 4096 * 4);
    hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(v8i), hipMemcpyHostToDevice);
    hipMemcpy(d_mask, rowmask.data(), rowmask.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_x, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 8;
    printf("line x row iterations per wave and pass: %.1f\n", iters_per_unit);
    for (int round = 0; round < 3; ++round) {
        for (int var = 0; var < 5; ++var) {
            auto kf = var == 0 ? k_loop<0> : var == 1 ? k_loop<1> : var == 2 ? k_loop<2> : var == 3 ? k_loop<3> : k_loop<4>;
            hipLaunchKernelGGL(kf, dim3(units / 4), dim3(256), 0, 0, d_recs, d_mask, d_x, d_out, reps);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kf, dim3(units / 4), dim3(256), 0, 0, d_recs, d_mask, d_x, d_out, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> h(64);
            hipMemcpy(h.data(), d_out, 64 * 4, hipMemcpyDeviceToHost);
            const double it_simd = iters_per_unit * reps * 8;      // 8 waves per SIMD
            printf("VAR %d  %8.3f ms  %6.1f cycles/iteration/SIMD @2.4GHz   (check %.6g)\n", var, ms,
                   ms * 1e-3 * 2.4e9 / it_simd, (double)h[5]);
        }
    }
    for (int w : {1, 2, 4, 8}) {
        const int n = 4096;
        hipLaunchKernelGGL(k_salu, dim3(256 * w), dim3(256), 0, 0, d_i, n);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_salu, dim3(256 * w), dim3(256), 0, 0, d_i, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 4 SIMDs x w waves x n x 16 scalar instructions
        printf("s_add_u32 w=%d %8.3f ms  %6.3f SALU/cycle/CU @2.4GHz\n", w, ms, 4.0 * w * n * 16 / (ms * 1e-3 * 2.4e9));
        hipLaunchKernelGGL(k_salu_valu, dim3(256 * w), dim3(256), 0, 0, d_i, n, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_salu_valu, dim3(256 * w), dim3(256), 0, 0, d_i, n, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("16 s_add + 8 v_fma w=%d %8.3f ms  %6.3f SALU/cycle/CU, %6.2f cycles per v_fma per SIMD\n", w, ms,
               4.0 * w * n * 16 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ((double)w * n * 8));
    }
    return 0;
}
