// fp64_pipe_bench.hip - what one wave64 instruction costs the SIMD's vector pipe on gfx950, measured (development aid).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/fp64_pipe_bench tools/fp64_pipe_bench.hip && tools/_build/fp64_pipe_bench
//
// One workgroup on one CU, W waves per SIMD (block = 256 W threads).  Every wave runs the same unrolled loop of N
// INDEPENDENT instructions of one kind (8 accumulators, so nothing waits on a result) between two s_memtime reads;
// printed: shader cycles per instruction PER SIMD = (cycles of the slowest wave) / (instructions x W).  If the pipe is
// the bound, that number is the instruction's issue cost whatever W is; where W = 1 reads more than W = 4, latency shows.
// The edge-cost and speed-DP kernels are priced with these figures (bench.py roofline_step).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024) void pipe_kernel(double* out, unsigned long long* cyc, int iters, double seed) {
    double a[8], b = seed + threadIdx.x * 1e-9, c = 1.0 + seed * 1e-3;
    float fa[8], fb = (float)b, fc = (float)c;
    for (int k = 0; k < 8; ++k) {
        a[k] = seed + k + threadIdx.x * 1e-6;
        fa[k] = (float)a[k];
    }
    unsigned long long acc_mask = 0;
    __shared__ double lds[1024 * 2];
    lds[threadIdx.x] = b;
    lds[threadIdx.x + 1024] = c;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(threadIdx.x * 8);
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) {
#define X(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(k) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                REP8(X)
#undef X
            } else if (KIND == 5) {
#define X(k) { unsigned long long m; asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b)); acc_mask ^= m; }
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[k]) : "v"(fb), "v"(fc));
                REP8(X)
#undef X
            } else if (KIND == 7) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(fa[k]));
                REP8(X)
#undef X
            } else if (KIND == 8) {
#define X(k) asm volatile("ds_read_b64 %0, %1" : "=v"(a[k]) : "v"(lds_addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (KIND == 9) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(fa[k]) : "v"(fb));
                REP8(X)
#undef X
            } else if (KIND == 10) {
#define X(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (KIND == 11) {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "=v"(fa[k]) : "v"(fb));
                REP8(X)
#undef X
            } else if (KIND == 12) {
#define X(k) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(fa[k]) : "v"(fb));
                REP8(X)
#undef X
            }
        }
    }
    unsigned long long t1 = clock64();
    double s = 0;
    float fs = 0;
    for (int k = 0; k < 8; ++k) {
        s += a[k];
        fs += fa[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + fs + (double)(acc_mask & 1);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int blocks) {
    const int iters = 2000;
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * 1024 * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * 16 * blocks);
    printf("%-22s", name);
    for (int W : {1, 2, 4}) {
        const int threads = 256 * W;
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(pipe_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.25);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16 * blocks);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * (threads / 64) * blocks, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int k = 0; k < (threads / 64) * blocks; ++k) mx = h[k] > mx ? h[k] : mx;
        printf("  W=%d: %8.4f", W, (double)mx / ((double)iters * 32.0 * W));
    }
    printf("   shader cycles (clock64) per instruction and SIMD\n");
    hipFree(out);
    hipFree(cyc);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;      // 1 = one CU; 256+ = the whole chip (clock under load)
    printf("blocks = %d (one per CU up to 256); W = waves per SIMD\n", blocks);
    run<11>("v_mov_b32", blocks);
    run<6>("v_fma_f32", blocks);
    run<9>("v_cndmask_b32", blocks);
    run<7>("v_rcp_f32", blocks);
    run<0>("v_fma_f64", blocks);
    run<1>("v_add_f64", blocks);
    run<12>("v_mov_b32 dpp row_shr", blocks);
    run<2>("v_mul_f64", blocks);
    run<4>("v_min_f64", blocks);
    run<5>("v_cmp_lt_f64 -> sgpr", blocks);
    run<3>("v_rcp_f64", blocks);
    run<10>("v_rsq_f64", blocks);
    run<8>("ds_read_b64 (x8, wait)", blocks);
    return 0;
}
