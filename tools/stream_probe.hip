// What can 586 single-wave blocks stream?  (experiment for the DP sweep: same geometry, no arithmetic)
//   mode 0: 8 B per lane per load (what the sweep does), ring of PD*9 loads in flight
//   mode 1: 16 B per lane per load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int PD, int ROW, bool NT = false>
__global__ __launch_bounds__(64) void stream8(const double* __restrict__ src, double* out, int rows_per_wave) {
    const double* p = src + (size_t)blockIdx.x * rows_per_wave * 64 + threadIdx.x;
    double acc[ROW];
#pragma unroll
    for (int k = 0; k < ROW; ++k) acc[k] = 0;
    double ring[PD][ROW];
    const int ncol = rows_per_wave / ROW;
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int k = 0; k < ROW; ++k) ring[d][k] = NT ? __builtin_nontemporal_load(&p[(size_t)(min(d, ncol - 1) * ROW + k) * 64]) : p[(size_t)(min(d, ncol - 1) * ROW + k) * 64];
    for (int j0 = 0; j0 < ncol; j0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
#pragma unroll
            for (int k = 0; k < ROW; ++k) acc[k] = fmin(acc[k], ring[d][k]);
            const int j = min(j0 + d + PD, ncol - 1);
#pragma unroll
            for (int k = 0; k < ROW; ++k) ring[d][k] = NT ? __builtin_nontemporal_load(&p[(size_t)(j * ROW + k) * 64]) : p[(size_t)(j * ROW + k) * 64];
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < ROW; ++k) s += acc[k];
    if (s == 12345.678) out[blockIdx.x] = s;
}

template <int PD, int ROW2>
__global__ __launch_bounds__(64) void stream16(const double2* __restrict__ src, double* out, int rows_per_wave) {
    const double2* p = src + (size_t)blockIdx.x * rows_per_wave * 64 + threadIdx.x;
    double acc[ROW2];
#pragma unroll
    for (int k = 0; k < ROW2; ++k) acc[k] = 0;
    double2 ring[PD][ROW2];
    const int ncol = rows_per_wave / ROW2;
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int k = 0; k < ROW2; ++k) ring[d][k] = p[(size_t)(min(d, ncol - 1) * ROW2 + k) * 64];
    for (int j0 = 0; j0 < ncol; j0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
#pragma unroll
            for (int k = 0; k < ROW2; ++k) acc[k] = fmin(acc[k], fmin(ring[d][k].x, ring[d][k].y));
            const int j = min(j0 + d + PD, ncol - 1);
#pragma unroll
            for (int k = 0; k < ROW2; ++k) ring[d][k] = p[(size_t)(j * ROW2 + k) * 64];
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < ROW2; ++k) s += acc[k];
    if (s == 12345.678) out[blockIdx.x] = s;
}

template <class F>
static double time_us(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 586;
    const int rows = 39 * 9;                       // 512-byte rows per wave (8 B mode)
    const size_t bytes = (size_t)waves * rows * 512;
    // three buffers rotated so that nothing is re-read from L2 / MALL
    double* buf[3]; double* out;
    for (auto& b : buf) { CK(hipMalloc(&b, bytes + 4096)); CK(hipMemset(b, 0, bytes + 4096)); }
    CK(hipMalloc(&out, waves * 8));
    int r = 0;
    printf("%d waves, %.1f MB per launch\n", waves, bytes / 1e6);
    auto report = [&](const char* name, double us) { printf("%-28s %7.2f us  %6.0f GB/s\n", name, us, bytes / us / 1e3); };
    report("8B/lane  PD=4", time_us([&] { hipLaunchKernelGGL((stream8<4, 9>), dim3(waves), dim3(64), 0, 0, buf[r++ % 3], out, rows); }, 30));
    report("8B/lane  PD=8", time_us([&] { hipLaunchKernelGGL((stream8<8, 9>), dim3(waves), dim3(64), 0, 0, buf[r++ % 3], out, rows); }, 30));
    report("8B/lane  PD=8 nontemporal", time_us([&] { hipLaunchKernelGGL((stream8<8, 9, true>), dim3(waves), dim3(64), 0, 0, buf[r++ % 3], out, rows); }, 30));
    report("8B/lane  PD=4 nontemporal", time_us([&] { hipLaunchKernelGGL((stream8<4, 9, true>), dim3(waves), dim3(64), 0, 0, buf[r++ % 3], out, rows); }, 30));
    report("8B/lane  PD=13", time_us([&] { hipLaunchKernelGGL((stream8<13, 9>), dim3(waves), dim3(64), 0, 0, buf[r++ % 3], out, rows); }, 30));
    // 16 B mode: 39*9 rows of 512 B = 175.5 rows of 1024 B -> 20 "columns" of 9 rows (2.5 % more bytes)
    const int rows16 = 20 * 9;
    const size_t bytes16 = (size_t)waves * rows16 * 1024;
    double2* b16[3];
    for (auto& b : b16) { CK(hipMalloc(&b, bytes16 + 4096)); CK(hipMemset(b, 0, bytes16 + 4096)); }
    auto report16 = [&](const char* name, double us) { printf("%-28s %7.2f us  %6.0f GB/s (of %.1f MB)\n", name, us, bytes16 / us / 1e3, bytes16 / 1e6); };
    report16("16B/lane PD=2", time_us([&] { hipLaunchKernelGGL((stream16<2, 9>), dim3(waves), dim3(64), 0, 0, b16[r++ % 3], out, rows16); }, 30));
    report16("16B/lane PD=4", time_us([&] { hipLaunchKernelGGL((stream16<4, 9>), dim3(waves), dim3(64), 0, 0, b16[r++ % 3], out, rows16); }, 30));
    report16("16B/lane PD=6", time_us([&] { hipLaunchKernelGGL((stream16<6, 9>), dim3(waves), dim3(64), 0, 0, b16[r++ % 3], out, rows16); }, 30));
    return 0;
}
