// Development micro-benchmark of the smoothing QP pair (x, y) of one polyline per wavefront.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iemplanner_carla_amd/csrc [-DEMP_SMOOTH_FORCE_LDS] tools/smooth_wave_bench.hip -o tools/_build/smb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "emp_qp_wave.h"
using namespace emp;

__global__ __launch_bounds__(64) void bench_kernel(int m, SmoothQpParams sx, SmoothQpParams sy, double* out, int* iters,
                                                   unsigned long long* ticks) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    double* xy = lds;                                   // [m][2]
    double* qmem = lds + 2 * m;
    const double R = 300.0 + blockIdx.x % 97;
    if (lane < m) {                                     // a noisy arc, 2.5 m spacing
        const double a = 2.5 * lane / R;
        const double noise = 0.12 * (((lane * 2654435761u + blockIdx.x * 40503u) >> 7) % 200 / 100.0 - 1.0);
        xy[2 * lane] = R * sin(a) + noise;
        xy[2 * lane + 1] = R * (1.0 - cos(a)) - noise;
    }
    __syncthreads();
    double *px, *py;
    int it = 0;
    const unsigned long long t0 = clock64();
    const int rc = smooth_pair_wave<true>(qmem, xy, 2, m, sx, sy, &px, &py, &it);
    const unsigned long long t1 = clock64();
    if (lane == 0) {
        iters[blockIdx.x] = rc ? -rc : it;
        out[blockIdx.x] = rc ? 0.0 : px[m / 2] + py[m / 2];
        ticks[blockIdx.x] = t1 - t0;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;
    const int m = argc > 2 ? atoi(argv[2]) : 42;
    SmoothQpParams s{0.4, 0.3, 0.3, 0.2};
    double* out; int* iters; unsigned long long* ticks;
    hipMalloc(&out, 8 * blocks); hipMalloc(&iters, 4 * blocks); hipMalloc(&ticks, 8 * blocks);
    const size_t lds = (2 * m + 2 * BoxRangeQp::words(m, m)) * sizeof(double);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench_kernel, dim3(blocks), dim3(64), lds, 0, m, s, s, out, iters, ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        int it0; double o0; unsigned long long t0;
        hipMemcpy(&it0, iters, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&o0, out, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&t0, ticks, 8, hipMemcpyDeviceToHost);
        if (rep == 2) printf("blocks %d m %d: kernel %.1f us, wave0 %llu ticks, iters %d (%.0f ticks/iter), check %.9f, lds %zu B\n", blocks, m,
                             ms * 1e3, t0, it0, it0 > 0 ? (double)t0 / it0 : 0.0, o0, lds);
    }
    return 0;
}
