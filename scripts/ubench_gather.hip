// Micro-benchmark: what one 8-byte table gather per lane costs a CU through the LDS (ds_read_b64) and through the
// vector-memory path (global_load_dwordx2 off an SGPR base, the table L1/L2 resident), with the index statistics of
// FastExp's B / C tables (fastexp.c:276-278: a byte of the float's low mantissa = as good as random inside a 2 KB row,
// a handful of rows per wave) and of its A table (neighbouring lanes share or neighbour an entry).
//   build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_gather scripts/ubench_gather.hip
// Output: cycles per wave-instruction and CU (2.4 GHz assumed; s_memtime ticks printed beside it), 32 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 4096
#define NTAB 2560                     // doubles: 10 rows of 256

// MODE bit 0: an LDS gather with random index; bit 1: a global gather with random index;
// bit 2 / bit 4: a second / third LDS gather (the table mode has three); bit 3: smooth index instead of random (A-like)
template <int MODE>
__global__ void __launch_bounds__(1024) k_gather(double *out, const double *__restrict__ tab, unsigned long long *ticks, int rows) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < NTAB; i += blockDim.x) lds[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned r0 = s >> 24, r1 = (s >> 16) & 255u, r2 = (s >> 8) & 255u;
        // a handful of rows per wave: the row from the lane's distance to a moving centre
        const unsigned row = (unsigned)((lane + it) & 63) * (unsigned)rows >> 6;
        if (MODE & 8) { r0 = ((unsigned)lane + (unsigned)it) >> 1 & 255u; }
        const unsigned i0 = row * 256u + r0, i1 = row * 256u + r1, i2 = row * 256u + r2;
        double v = 0.0;
        if (MODE & 1) v += lds[i0];
        if (MODE & 4) v += lds[i1];
        if (MODE & 16) v += lds[i2];
        if (MODE & 2) {
            double g;
            const unsigned off = i2 * 8u;
            asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(off), "s"(tab) : "memory");
            v += g;
        }
        acc += v;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// the same with the global gather waited for one iteration late (a wave keeps one in flight)
template <int MODE>
__global__ void __launch_bounds__(1024) k_gather_late(double *out, const double *__restrict__ tab, unsigned long long *ticks, int rows) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < NTAB; i += blockDim.x) lds[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0, g0 = 0.0, g1 = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it += 2) {
        double v = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            s = s * 1664525u + 1013904223u;
            const unsigned r0 = s >> 24, r1 = (s >> 16) & 255u, r2 = (s >> 8) & 255u;
            const unsigned row = (unsigned)((lane + it + h) & 63) * (unsigned)rows >> 6;
            const unsigned i0 = row * 256u + r0, i1 = row * 256u + r1, i2 = row * 256u + r2;
            if (MODE & 1) v += lds[i0];
            if (MODE & 4) v += lds[i1];
            const unsigned off = i2 * 8u;
            if (h == 0) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(g0) : "v"(off), "s"(tab) : "memory");
            else        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(g1) : "v"(off), "s"(tab) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(g1) :: "memory");
        acc += v + g0 + g1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename K>
static void run(const char *name, K kern, int n_gathers, double *out, const double *tab, unsigned long long *ticks, int rows) {
    const int blocks = 512;                      // two workgroups of 16 waves per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 64 * 1024, 0, out, tab, ticks, rows);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 64 * 1024, 0, out, tab, ticks, rows);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(blocks);
    CK(hipMemcpy(t.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double tick_mean = 0; for (auto v : t) tick_mean += (double)v; tick_mean /= blocks;
    const double us = ms * 1e3 / reps;
    // one CU ran 32 waves x ITER steps in `us`
    const double cyc_step_cu = us * 2400.0 / (32.0 * ITER);
    printf("%-44s rows %2d  %8.1f us  %6.2f cycles per step and CU (%5.2f per gather)   s_memtime/step/wave %7.1f\n",
           name, rows, us, cyc_step_cu, cyc_step_cu / n_gathers, tick_mean / ITER);
}

int main() {
    double *tab, *out; unsigned long long *ticks;
    CK(hipMalloc(&tab, NTAB * 8 + 65536)); CK(hipMalloc(&out, 512 * 1024 * 8)); CK(hipMalloc(&ticks, 512 * 8));
    std::vector<double> h(NTAB + 8192, 1.0);
    CK(hipMemcpy(tab, h.data(), NTAB * 8 + 65536, hipMemcpyHostToDevice));
    for (int rows : {1, 4, 9}) {
        run("index arithmetic only", k_gather<0>, 1, out, tab, ticks, rows);
        run("1 LDS gather (random in row)", k_gather<1>, 1, out, tab, ticks, rows);
        run("1 LDS gather (smooth index)", k_gather<9>, 1, out, tab, ticks, rows);
        run("3 LDS gathers", k_gather<21>, 3, out, tab, ticks, rows);
        run("1 global gather, waited at once", k_gather<2>, 1, out, tab, ticks, rows);
        run("2 LDS + 1 global gather, waited at once", k_gather<7>, 3, out, tab, ticks, rows);
        run("1 global gather, two in flight", k_gather_late<0>, 1, out, tab, ticks, rows);
        run("2 LDS + 1 global gather, two in flight", k_gather_late<5>, 3, out, tab, ticks, rows);
    }
    return 0;
}
