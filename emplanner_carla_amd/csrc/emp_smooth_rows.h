// emp_smooth_rows.h - the smoothing QP (ref smooth_reference_line, planning_utils.py:262-361) with EIGHT problems per
// wavefront: the x and the y problem of four polylines (device only).
//
// Same idea as emp_qp_rows.h: a problem takes a group of GP = 8 lanes (16 for polylines of up to 64 points: four problems
// per wavefront) and every lane owns R CONSECUTIVE coordinates (GP = 8: R = 3 up to 24 points, R = 4 up to 32), so a sweep
// of the pentadiagonal Cholesky factorisation / substitution serves eight problems where the half-wave form of
// emp_qp_wave.h (box_qp_active_set_lanes<32>) serves two.  Same algorithm - the primal-dual active-set iteration - and the
// same operation order per coordinate, so the fixed point it stops at is the same KKT point: the trajectories of the
// benchmark batch come out bit-identical.
#pragma once

#include "emp_qp_rows.h"

namespace emp {

#pragma clang fp contract(fast)      // see emp_qp_wave.h

// Banded Cholesky, half bandwidth 2, R consecutive rows per lane (row j = gl * R + r); see band_chol_rows.
// a[r][0..2] = A[j][j..j+2] on entry, the factor row on return; low[r][e] = U[j-e][e] (e = 1, 2).
template <int GP, int R>
__device__ __forceinline__ bool band_chol_rows2(double (&a)[R][3], double (&rinv)[R], double (&low)[R][3], int N, int gl,
                                                bool active, int steps) {
    static_assert(R >= 2, "a lane must hold at least KD = 2 rows");
    double A[R][3];
    bool rowv[R], inband[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = gl * R + r;
        rowv[r] = active && j < N;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            A[r][d] = rowv[r] ? a[r][d] : (d == 0 ? 1.0 : 0.0);
            a[r][d] = A[r][d];
            inband[r][d] = rowv[r] && j + d < N;
        }
        rinv[r] = 1.0;
    }
    double diag[R];
#pragma unroll
    for (int r = 0; r < R; ++r) diag[r] = 1.0;
    for (int k = 0; k < steps; ++k) {
        const double n1_1 = lane_up1(a[R - 1][1]), n1_2 = lane_up1(a[R - 1][2]);      // row j0-1
        const double n2_2 = lane_up1(a[R - 2][2]);                                    // row j0-2
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double u1_1 = (r >= 1) ? a[r >= 1 ? r - 1 : 0][1] : n1_1;
            const double u1_2 = (r >= 1) ? a[r >= 1 ? r - 1 : 0][2] : n1_2;
            const double u2_2 = (r >= 2) ? a[r >= 2 ? r - 2 : 0][2] : (r == 1 ? n1_2 : n2_2);
            double acc0 = A[r][0], acc1 = A[r][1];
            const double acc2 = A[r][2];
            acc0 = __builtin_fma(-u1_1, u1_1, acc0);
            acc1 = __builtin_fma(-u1_1, u1_2, acc1);
            acc0 = __builtin_fma(-u2_2, u2_2, acc0);
            diag[r] = acc0;
            const double rs = fast_rsqrt(acc0 > 0.0 ? acc0 : 1.0);
            a[r][0] = acc0 * rs;
            a[r][1] = inband[r][1] ? acc1 * rs : 0.0;
            a[r][2] = inband[r][2] ? acc2 * rs : 0.0;
            rinv[r] = rs;
        }
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double offd = fabs(a[r][1]) + fabs(a[r][2]);
        bad = bad || (rowv[r] && !(diag[r] > 0.0 && diag[r] < 1e300 && offd < 1e300));
    }
    const bool failed = oct_any<GP>(bad);
    if (failed) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            a[r][0] = 1.0;
            a[r][1] = a[r][2] = 0.0;
            rinv[r] = 1.0;
        }
    }
    {
        const double n1 = lane_up1(a[R - 1][1]), n1b = lane_up1(a[R - 1][2]), n2b = lane_up1(a[R - 2][2]);
        const bool first = gl == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            low[r][0] = 0.0;
            low[r][1] = (r >= 1) ? a[r >= 1 ? r - 1 : 0][1] : (first ? 0.0 : n1);
            low[r][2] = (r >= 2) ? a[r >= 2 ? r - 2 : 0][2] : (r == 1 ? (first ? 0.0 : n1b) : (first ? 0.0 : n2b));
        }
    }
    return !failed;
}

template <int R>
__device__ __forceinline__ void band_solve_rows2(const double (&a)[R][3], const double (&rinv)[R], const double (&low)[R][3],
                                                 double (&b)[R], int steps) {
    double rhs[R], y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rhs[r] = (fabs(b[r]) < 1e300) ? b[r] : 0.0;
        y[r] = rhs[r] * rinv[r];
    }
    for (int k = 0; k < steps; ++k) {                               // U' y = b
        const double p1 = lane_up1(y[R - 1]), p2 = lane_up1(y[R - 2]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double y1 = (r >= 1) ? y[r >= 1 ? r - 1 : 0] : p1;
            const double y2 = (r >= 2) ? y[r >= 2 ? r - 2 : 0] : (r == 1 ? p1 : p2);
            double acc = __builtin_fma(-low[r][1], y1, rhs[r]);
            acc = __builtin_fma(-low[r][2], y2, acc);
            y[r] = acc * rinv[r];
        }
    }
    double x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = y[r] * rinv[r];
    for (int k = 0; k < steps; ++k) {                               // U x = y
        const double q1 = lane_dn1(x[0]), q2 = lane_dn1(x[1]);
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const double x1 = (r + 1 < R) ? x[r + 1 < R ? r + 1 : 0] : q1;
            const double x2 = (r + 2 < R) ? x[r + 2 < R ? r + 2 : 0] : (r + 2 == R ? q1 : q2);
            double acc = __builtin_fma(-a[r][1], x1, y[r]);
            acc = __builtin_fma(-a[r][2], x2, acc);
            x[r] = acc * rinv[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) b[r] = x[r];
}

// One box QP per group of GP lanes (8 or 16) by the primal-dual active-set iteration (box_qp_active_set_lanes, same arithmetic per
// coordinate): ref[r] = the lane's reference coordinates j = gl * R + r (anything for j >= m), box ref +- thr.
// Every lane of the wavefront must call it.  Returns (per group) 0 settled (x holds the minimiser), -1 classification
// still changing after kBoxAsMaxIter rounds (the caller falls back to the interior point), 2 bad input.
template <int GP, int R>
__device__ inline int box_qp_active_set_rows(const double (&ref)[R], int m, const SmoothQpParams& prm, double (&x)[R],
                                             int* iters_out) {
    const int gl = (threadIdx.x & 63) & (GP - 1), base = gl * R;
    const bool valid = m >= 2 && m <= GP * R && prm.thr > 0.0;
    *iters_out = 0;
    bool has[R];
    double Prow[R][3], Plow[R][3], q[R], lo[R], hi[R];
    int code[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = base + r;
        has[r] = valid && j < m;
#pragma unroll
        for (int d = 0; d < 3; ++d) {                    // ref planning_utils.py:313-344 cost matrices, as box_qp_active_set_lanes
            double e = 0.0;
            if (has[r] && j + d < m) {
                for (int rr = max(0, j + d - 2); rr <= min(m - 3, j); ++rr) {
                    const double a = (j - rr == 1) ? -2.0 : 1.0, b = (j + d - rr == 1) ? -2.0 : 1.0;
                    e += 2.0 * prm.w_smooth * a * b;
                }
                for (int rr = max(0, j + d - 1); rr <= min(m - 2, j); ++rr) {
                    const double a = (j - rr == 0) ? 1.0 : -1.0, b = (j + d - rr == 0) ? 1.0 : -1.0;
                    e += 2.0 * prm.w_length * a * b;
                }
                if (d == 0) e += 2.0 * prm.w_ref;
            }
            Prow[r][d] = e;
        }
        q[r] = has[r] ? -2.0 * prm.w_ref * ref[r] : 0.0;      // ref :346
        lo[r] = ref[r] - prm.thr;                           // ref :308-311
        hi[r] = ref[r] + prm.thr;
        x[r] = ref[r];
        code[r] = 0;                                        // 0 free, 1 fixed at hi, 2 fixed at lo
    }
    {
        const double t1 = lane_up1(Prow[R - 1][1]), t1b = lane_up1(Prow[R - 1][2]), t2b = lane_up1(Prow[R - 2][2]);
        const bool first = gl == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            Plow[r][0] = 0.0;
            Plow[r][1] = (r >= 1) ? Prow[r >= 1 ? r - 1 : 0][1] : (first ? 0.0 : t1);                                   // P[j-1][j]
            Plow[r][2] = (r >= 2) ? Prow[r >= 2 ? r - 2 : 0][2] : (r == 1 ? (first ? 0.0 : t1b) : (first ? 0.0 : t2b));  // P[j-2][j]
        }
    }
    const double c = 2.0 * (6.0 * prm.w_smooth + 2.0 * prm.w_length + prm.w_ref);   // the Hessian's interior diagonal
    int state = valid ? 1 : 0, iters = 0;                  // 1 running, 0 done, -1 gave up
    const int steps = oct_wave_max<GP>(valid ? (m + R - 1) / R : 0);
    const bool g_first = gl == 0, g_last = gl == GP - 1;
    // value i of the lane's row window [-2, R + 1]: own rows, the left neighbour's last two, the right neighbour's first two
    auto window = [&](const double (&v)[R], double (&w)[R + 4]) {
        const double p2 = lane_up1(v[R - 2]), p1 = lane_up1(v[R - 1]), n0 = lane_dn1(v[0]), n1 = lane_dn1(v[1]);
        w[0] = g_first ? 0.0 : p2;
        w[1] = g_first ? 0.0 : p1;
#pragma unroll
        for (int r = 0; r < R; ++r) w[2 + r] = v[r];
        w[R + 2] = g_last ? 0.0 : n0;
        w[R + 3] = g_last ? 0.0 : n1;
    };
    while (__any(state == 1)) {
        const bool go = state == 1;
        bool act[R];
        double af[R], ab[R], afw[R + 4], abw[R + 4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            act[r] = go && has[r] && code[r] != 0;
            const double bnd = (code[r] == 1) ? hi[r] : lo[r];
            af[r] = act[r] ? 1.0 : 0.0;
            ab[r] = act[r] ? bnd : 0.0;
        }
        window(af, afw);
        window(ab, abw);
        double fa[R][3], flow[R][3], frinv[R], rhs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            fa[r][0] = act[r] ? 1.0 : ((go && has[r]) ? Prow[r][0] : 0.0);
            fa[r][1] = (go && !act[r] && afw[r + 3] == 0.0) ? Prow[r][1] : 0.0;
            fa[r][2] = (go && !act[r] && afw[r + 4] == 0.0) ? Prow[r][2] : 0.0;
            const double free_rhs = -q[r] - (((Prow[r][1] * abw[r + 3] + Prow[r][2] * abw[r + 4]) + Plow[r][1] * abw[r + 1]) + Plow[r][2] * abw[r]);
            rhs[r] = act[r] ? ab[r] : ((go && has[r]) ? free_rhs : 0.0);
        }
        const bool okf = band_chol_rows2<GP, R>(fa, frinv, flow, m, gl, go, steps);
        band_solve_rows2<R>(fa, frinv, flow, rhs, steps);
        double xs[R], xw[R + 4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (go && has[r]) x[r] = rhs[r];
            xs[r] = (go && has[r]) ? x[r] : 0.0;
        }
        if (go) ++iters;
        window(xs, xw);
        // gradient of the full problem at x and the new classification
        bool changed_here = false;
        int ncode[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double g = (q[r] + Prow[r][0] * xs[r]) + (((Prow[r][1] * xw[r + 3] + Prow[r][2] * xw[r + 4]) + Plow[r][1] * xw[r + 1]) + Plow[r][2] * xw[r]);
            const double lam = -g;
            int nc = 0;
            if (lam + c * (xs[r] - hi[r]) > 0.0) nc = 1;
            else if (lam + c * (xs[r] - lo[r]) < 0.0) nc = 2;
            if (!(go && has[r])) nc = code[r];
            ncode[r] = nc;
            changed_here = changed_here || (nc != code[r]);
        }
        const bool changed = oct_any<GP>(changed_here);
        if (go) {
#pragma unroll
            for (int r = 0; r < R; ++r) code[r] = ncode[r];
            if (!okf) state = -1;
            else if (!changed) state = 0;
            else if (iters >= kBoxAsMaxIter) state = -1;
        }
    }
    *iters_out = iters;
    return valid ? state : 2;
}

#pragma clang fp contract(off)

}  // namespace emp
