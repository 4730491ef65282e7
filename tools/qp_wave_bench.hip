// Development micro-benchmark: where does one interior-point iteration of the wave-cooperative path QP spend
// its time?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iemplanner_carla_amd/csrc tools/qp_wave_bench.hip -o /tmp/qpb && /tmp/qpb [blocks] [G]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ unsigned long long g_prof[16];
__device__ unsigned long long g_last;
#define EMP_QP_PROF(i)                                       \
    do {                                                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) {           \
            const unsigned long long t_ = clock64();         \
            g_prof[i] += t_ - g_last;                        \
            g_last = t_;                                     \
        }                                                    \
    } while (0)
#ifdef QPB_DEBUG
#define EMP_QP_DEBUG(...) do { if (threadIdx.x == 0 && blockIdx.x == 0) printf(__VA_ARGS__); } while (0)
#endif
#include "emp_qp_wave.h"

using namespace emp;

template <int G>
__global__ __launch_bounds__(64) void bench_kernel(int n, PathQpParams prm, double* out, int* iters, unsigned long long* total,
                                                   unsigned long long* trace) {
    extern __shared__ double lds_all[];
    constexpr int GPW = 64 / G;
    const int lane = threadIdx.x & 63, grp = lane / G, gl = lane & (G - 1);
    const int per_group = 3 * n + path_qp_words(n);
    double* lds = lds_all + grp * per_group;
    double *lmin = lds, *lmax = lds + n, *ql = lds + 2 * n;
    const int scene = blockIdx.x * GPW + grp;
    for (int i = gl; i < n; i += G) {  // a corridor with two obstacles that the path has to weave through
        double lo = -10.0, hi = 10.0;
        if (i >= 5 && i <= 8) hi = -0.6 - 0.01 * (scene % 7);
        if (i >= 13 && i <= 16) lo = 0.4 + 0.01 * (scene % 5);
        lmin[i] = lo;
        lmax[i] = hi;
    }
    __syncthreads();
    int it = 0;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) g_last = t0;
    const int rc = path_qp_group<G>(lds + 3 * n, lmin, lmax, n, 0.1, 0.0, 0.0, prm, ql, nullptr, nullptr, &it, true, 0);
    const unsigned long long t1 = clock64();
    if (gl == 0) {
        iters[scene] = rc == 0 ? it : -rc;
        out[scene] = ql[n / 2];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = t1 - t0;
    if (threadIdx.x == 0 && trace) {
        trace[3 * blockIdx.x] = w0;
        trace[3 * blockIdx.x + 1] = wall_clock64();
        trace[3 * blockIdx.x + 2] = t1 - t0;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;
    const int G = argc > 2 ? atoi(argv[2]) : 32;
    const int n = argc > 3 ? atoi(argv[3]) : 21;
    PathQpParams prm;
    prm.ds = 5.0; prm.w_l = 1000; prm.w_ddl = 3000; prm.w_dddl = 150; prm.w_centre = 250;
    prm.d1 = 3; prm.d2 = 3; prm.host_w = 3;
    const int gpw = 64 / G;
    double* out; int* iters; unsigned long long* total;
    hipMalloc(&out, sizeof(double) * blocks * gpw);
    hipMalloc(&iters, sizeof(int) * blocks * gpw);
    hipMalloc(&total, sizeof(unsigned long long));
    unsigned long long* trace;
    hipMalloc(&trace, sizeof(unsigned long long) * 3 * blocks);
    const size_t lds = (size_t)gpw * (3 * n + path_qp_words(n)) * sizeof(double);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof(zero));
        hipEventRecord(e0);
        if (G == 32) hipLaunchKernelGGL(bench_kernel<32>, dim3(blocks), dim3(64), lds, 0, n, prm, out, iters, total, trace);
        else hipLaunchKernelGGL(bench_kernel<64>, dim3(blocks), dim3(64), lds, 0, n, prm, out, iters, total, trace);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long prof[16], tot;
        int it0;
        double o0;
        hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof), sizeof(prof));
        hipMemcpy(&tot, total, sizeof(tot), hipMemcpyDeviceToHost);
        hipMemcpy(&it0, iters, sizeof(int), hipMemcpyDeviceToHost);
        hipMemcpy(&o0, out, sizeof(double), hipMemcpyDeviceToHost);
        printf("blocks %d G %d n %d: kernel %.1f us, wave0 total %llu ticks, iters %d, l_mid %.6f\n", blocks, G, n, ms * 1e3, tot, it0, o0);
        const char* names[10] = {"loop top", "1 stations", "2 matrix row", "reductions+test", "3 cholesky", "4 predictor rhs",
                                 "predictor solve", "5 affine/corrector rhs", "corrector solve", "6 step+update"};
        if (rep == 2) {
            unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 3 * blocks);
            hipMemcpy(h, trace, sizeof(unsigned long long) * 3 * blocks, hipMemcpyDeviceToHost);
            unsigned long long w_min = ~0ull, w_max = 0, s_max = 0, tk_min = ~0ull, tk_max = 0;
            double dur = 0;
            for (int b = 0; b < blocks; ++b) {
                if (h[3 * b] < w_min) w_min = h[3 * b];
                if (h[3 * b] > s_max) s_max = h[3 * b];
                if (h[3 * b + 1] > w_max) w_max = h[3 * b + 1];
                if (h[3 * b + 2] < tk_min) tk_min = h[3 * b + 2];
                if (h[3 * b + 2] > tk_max) tk_max = h[3 * b + 2];
                dur += (double)(h[3 * b + 1] - h[3 * b]);
            }
            printf("   wall (100 MHz): first start -> last start %.1f us, first start -> last end %.1f us, mean wave duration %.1f us; ticks min %llu max %llu\n",
                   (s_max - w_min) / 100.0, (w_max - w_min) / 100.0, dur / blocks / 100.0, tk_min, tk_max);
        }
        if (rep == 2)
            for (int i = 0; i < 10; ++i) printf("   %-24s %8.1f ticks/iter\n", names[i], (double)prof[i] / (it0 > 0 ? it0 + 1 : 1));
    }
    return 0;
}
