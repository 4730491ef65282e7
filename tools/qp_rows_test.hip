// Development check: band_chol_rows / band_solve_rows (R rows per lane on 8-lane groups, emp_qp_rows.h) against the scalar band_chol /
// band_solve of emp_qp_core.h on random SPD band matrices, eight groups of different sizes per wavefront.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iemplanner_carla_amd/csrc tools/qp_rows_test.hip -o tools/_build/qrt
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "emp_qp_rows.h"
using namespace emp;

template <int GP, int R>
__global__ void k_rows(const double* band, const double* rhs, const int* Ns, double* x, int* okout, double* fac) {
    const int lane = threadIdx.x & 63, grp = lane / GP, gl = lane & (GP - 1);
    const int N = Ns[grp];
    double a[R][4], low[R][4], rinv[R], b[R];
    for (int r = 0; r < R; ++r) {
        const int j = gl * R + r;
        for (int d = 0; d < 4; ++d) a[r][d] = (j < N && j + d < N) ? band[(grp * 64 + j) * 4 + d] : 0.0;
        b[r] = j < N ? rhs[grp * 64 + j] : 0.0;
    }
    const int steps = oct_wave_max<GP>((N + R - 1) / R);
    const bool ok = band_chol_rows<GP, R>(a, rinv, low, N, gl, N > 0, steps);
    band_solve_rows<R>(a, rinv, low, b, steps);
    for (int r = 0; r < R; ++r) {
        x[grp * 64 + gl * R + r] = b[r];
        for (int d = 0; d < 4; ++d) fac[((grp * 64) + gl * R + r) * 4 + d] = a[r][d];
    }
    okout[lane] = ok;
}

template <int GP, int R>
int run(const int* hN) {
    constexpr int NG = 64 / GP;
    double hb[NG * 64 * 4] = {0}, hr[NG * 64] = {0}, ref[NG * 64] = {0};
    for (int g = 0; g < NG; ++g) {
        const int N = hN[g];
        double M[64 * 4] = {0};
        for (int i = 0; i < N; ++i) {
            for (int d = 0; d <= 3; ++d) {
                double v = (d == 0) ? 1317.0 + (rand() % 100) : (d == 1) ? 510.0 : (d == 2) ? 82.0 : -0.48;
                if (i + d >= N) v = 0.0;
                hb[(g * 64 + i) * 4 + d] = v;
                M[i * 4 + d] = v;
            }
            hr[g * 64 + i] = (rand() % 2000) / 100.0 - 10.0;
            ref[g * 64 + i] = hr[g * 64 + i];
        }
        if (N && !band_chol<3>(M, N)) printf("scalar chol failed\n");
        if (N) band_solve<3>(M, ref + g * 64, N);
    }
    double *db, *dr, *dx, *df;
    int *dN, *dok;
    hipMalloc(&db, sizeof(hb)); hipMalloc(&dr, sizeof(hr)); hipMalloc(&dx, NG * 64 * 8); hipMalloc(&df, NG * 64 * 4 * 8);
    hipMalloc(&dN, NG * 4); hipMalloc(&dok, 64 * 4);
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dr, hr, sizeof(hr), hipMemcpyHostToDevice);
    hipMemcpy(dN, hN, NG * 4, hipMemcpyHostToDevice);
    hipMemset(dx, 0, NG * 64 * 8);
    hipLaunchKernelGGL((k_rows<GP, R>), dim3(1), dim3(64), 0, 0, db, dr, dN, dx, dok, df);
    double hx[NG * 64], hf[NG * 64 * 4];
    int hok[64];
    hipMemcpy(hx, dx, sizeof(hx), hipMemcpyDeviceToHost);
    hipMemcpy(hf, df, sizeof(hf), hipMemcpyDeviceToHost);
    hipMemcpy(hok, dok, sizeof(hok), hipMemcpyDeviceToHost);
    double worst = 0;
    int nanfac = 0, oks = 0;
    for (int g = 0; g < NG; ++g) {
        oks += hok[g * GP];
        for (int i = 0; i < hN[g]; ++i) {
            const double e = fabs(hx[g * 64 + i] - ref[g * 64 + i]) / fmax(1.0, fabs(ref[g * 64 + i]));
            worst = (e == e) ? fmax(worst, e) : 1e300;
        }
        for (int i = 0; i < GP * R * 4; ++i) nanfac += !(hf[g * 64 * 4 + i] == hf[g * 64 * 4 + i]);
    }
    printf("GP %d R %d N", GP, R);
    for (int g = 0; g < NG; ++g) printf(" %d", hN[g]);
    printf(": worst rel err %.3e, ok groups %d of %d, NaN factor entries %d\n", worst, oks, NG, nanfac);
    return worst < 1e-11 ? 0 : 1;
}

int main() {
    int bad = 0;
    { const int n[8] = {17, 17, 17, 17, 17, 17, 17, 17}; bad += run<8, 3>(n); }
    { const int n[8] = {17, 1, 22, 0, 18, 3, 24, 9}; bad += run<8, 3>(n); }
    { const int n[8] = {24, 24, 24, 24, 24, 24, 24, 24}; bad += run<8, 3>(n); }
    { const int n[8] = {30, 17, 32, 1, 0, 29, 4, 31}; bad += run<8, 4>(n); }
    { const int n[4] = {57, 64, 17, 0}; bad += run<16, 4>(n); }
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad;
}
