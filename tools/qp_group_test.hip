// Development check: band_chol_group / band_solve_group (lane-shift sweeps) against the scalar band_chol /
// band_solve of emp_qp_core.h on random SPD band matrices, both groups of a wavefront.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iemplanner_carla_amd/csrc tools/qp_group_test.hip -o tools/_build/qgt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "emp_qp_wave.h"
using namespace emp;

template <int G, int KD>
__global__ void k_test(const double* band, const double* rhs, const int* Ns, double* x, double* shifts, int* okout) {
    const int lane = threadIdx.x & 63, grp = lane / G, gl = lane & (G - 1);
    const int N = Ns[grp];
    double a[KD + 1], low[KD + 1], rinv;
    for (int d = 0; d <= KD; ++d) a[d] = gl < N ? band[(grp * 64 + gl) * 4 + d] : 0.0;
    const bool ok = band_chol_group<G, KD>(a, rinv, low, N, gl, true);
    double b = gl < N ? rhs[grp * 64 + gl] : 0.0;
    band_solve_group<G, KD>(a, rinv, low, b, N, gl);
    x[lane] = b;
    okout[lane] = ok;
    shifts[lane] = lane_up1((double)lane);
    shifts[64 + lane] = lane_dn1((double)lane);
}

template <int G, int KD>
int run(int N0, int N1) {
    double hb[2 * 64 * 4] = {0}, hr[2 * 64] = {0}, ref[2 * 64] = {0};
    int hN[2] = {N0, N1};
    for (int g = 0; g < 64 / G; ++g) {
        const int N = hN[g];
        double M[64 * 4] = {0};
        for (int i = 0; i < N; ++i) {
            for (int d = 0; d <= KD; ++d) {
                double v = (d == 0) ? 1317.0 + (rand() % 100) : (d == 1) ? 510.0 : (d == 2) ? 82.0 : -0.48;   // like the path-QP Hessian: not diagonally dominant
                if (i + d >= N) v = 0.0;
                hb[(g * 64 + i) * 4 + d] = v;
                M[i * (KD + 1) + d] = v;
            }
            hr[g * 64 + i] = (rand() % 2000) / 100.0 - 10.0;
            ref[g * 64 + i] = hr[g * 64 + i];
        }
        if (!band_chol<KD>(M, N)) printf("scalar chol failed\n");
        band_solve<KD>(M, ref + g * 64, N);
    }
    double *db, *dr, *dx, *ds; int *dN, *dok;
    hipMalloc(&db, sizeof(hb)); hipMalloc(&dr, sizeof(hr)); hipMalloc(&dx, 64 * 8); hipMalloc(&ds, 128 * 8);
    hipMalloc(&dN, 8); hipMalloc(&dok, 64 * 4);
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dr, hr, sizeof(hr), hipMemcpyHostToDevice);
    hipMemcpy(dN, hN, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_test<G, KD>), dim3(1), dim3(64), 0, 0, db, dr, dN, dx, ds, dok);
    double hx[64], hs[128]; int hok[64];
    hipMemcpy(hx, dx, sizeof(hx), hipMemcpyDeviceToHost);
    hipMemcpy(hs, ds, sizeof(hs), hipMemcpyDeviceToHost);
    hipMemcpy(hok, dok, sizeof(hok), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int g = 0; g < 64 / G; ++g)
        for (int i = 0; i < hN[g]; ++i) worst = fmax(worst, fabs(hx[g * G + i] - ref[g * 64 + i]) / fmax(1.0, fabs(ref[g * 64 + i])));
    printf("G %d KD %d N %d/%d: worst rel err %.3e, ok flags %d %d; up1[0..2] %.0f %.0f %.0f up1[32] %.0f dn1[0] %.0f dn1[63] %.0f\n",
           G, KD, N0, N1, worst, hok[0], hok[63], hs[0], hs[1], hs[2], hs[32], hs[64], hs[127]);
    return worst < 1e-11 ? 0 : 1;
}

__global__ void k_red(const double* v, double* out) {
    const double x = v[threadIdx.x];
    out[threadIdx.x] = group_max<32>(x);
    out[64 + threadIdx.x] = group_min<32>(x);
    out[128 + threadIdx.x] = group_sum<32>(x);
    out[192 + threadIdx.x] = group_max<64>(x);
    out[256 + threadIdx.x] = group_sum<64>(x);
}

static int test_reductions() {
    double h[64], o[320];
    for (int i = 0; i < 64; ++i) h[i] = (rand() % 2001) - 1000;            // integers: sums are exact in any order
    double *dv, *dout;
    hipMalloc(&dv, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(dv, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_red, dim3(1), dim3(64), 0, 0, dv, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int g = 0; g < 2; ++g) {
        double mx = -1e300, mn = 1e300, sm = 0;
        for (int i = 0; i < 32; ++i) { mx = fmax(mx, h[g * 32 + i]); mn = fmin(mn, h[g * 32 + i]); sm += h[g * 32 + i]; }
        for (int i = 0; i < 32; ++i) bad += (o[g * 32 + i] != mx) + (o[64 + g * 32 + i] != mn) + (o[128 + g * 32 + i] != sm);
    }
    double mx = -1e300, sm = 0;
    for (int i = 0; i < 64; ++i) { mx = fmax(mx, h[i]); sm += h[i]; }
    for (int i = 0; i < 64; ++i) bad += (o[192 + i] != mx) + (o[256 + i] != sm);
    printf("group reductions: %d mismatches\n", bad);
    return bad ? 1 : 0;
}

int main() {
    int bad = test_reductions();
    bad += run<32, 3>(17, 17);
    bad += run<32, 3>(32, 5);
    bad += run<32, 3>(1, 30);
    bad += run<32, 2>(32, 19);
    bad += run<64, 3>(17, 0);
    bad += run<64, 3>(64, 0);
    bad += run<64, 2>(41, 0);
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
