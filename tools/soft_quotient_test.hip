// The edge-cost kernel's 5000 / d2 (emp_dp_kernels.h: soft_cost_quotient, one Newton step) against the compiler's IEEE binary64
// division on 2^33 operands of the range it is used on, 16 < d2 < 36: an even sweep of the interval and a hashed one (all
// mantissa bits in play).  Prints the number of results that differ in any bit; 0 expected.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I emplanner_carla_amd/csrc tools/soft_quotient_test.hip -o /tmp/sq && /tmp/sq
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "emp_dp_kernels.h"

__global__ void check(unsigned long long per_thread, unsigned long long* bad, unsigned long long* done) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long total = (unsigned long long)gridDim.x * blockDim.x * per_thread;
    unsigned long long mine = 0;
    for (unsigned long long k = 0; k < per_thread; ++k) {
        const unsigned long long i = tid * per_thread + k;
        // even sweep
        double d2 = 16.0 + ((double)i + 0.5) * (20.0 / (double)total);
        // hashed: splitmix64 bits into the mantissa of a value in [16, 32) or [32, 36)
        unsigned long long z = i + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        double h = __longlong_as_double(0x4030000000000000ull | (z & 0x000FFFFFFFFFFFFFull));   // [16, 32)
        if ((z >> 60) == 0) h = 32.0 + (h - 16.0) * 0.25;                                         // some in [32, 36)
        for (int v = 0; v < 2; ++v) {
            const double x = v ? h : d2;
            if (!(x > 16.0 && x < 36.0)) continue;
            const double a = emp::soft_cost_quotient(x), b = 5000.0 / x;
            if (__double_as_longlong(a) != __double_as_longlong(b)) ++mine;
        }
    }
    if (mine) atomicAdd(bad, mine);
    if (threadIdx.x == 0) atomicAdd(done, 1ull);      // the host checks that every block ran to its end
}

#define CHECK(call)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            printf("%s failed: %s\n", #call, hipGetErrorString(e_));                       \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

int main() {
    unsigned long long *bad, *done;
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&done, 8));
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipMemset(done, 0, 8));         // a kernel that never ran cannot read as "0 differences": `done` counts its blocks
    const unsigned long long per_thread = 1ull << 12;
    const unsigned blocks = 1u << 12;
    hipLaunchKernelGGL(check, dim3(blocks), dim3(256), 0, 0, per_thread, bad, done);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    unsigned long long h = ~0ull, d = 0;
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&d, done, 8, hipMemcpyDeviceToHost));
    if (d != blocks) {
        printf("only %llu of %u blocks ran\n", d, blocks);
        return 2;
    }
    printf("operands checked: %llu, results differing from IEEE division: %llu\n", 2ull * blocks * 256 * per_thread, h);
    return h != 0;
}
