// emp_qp_rows.h - the path QP with EIGHT (or four) problems per wavefront (device only).
//
// emp_qp_wave.h solves the banded range QP with one station and one unknown per lane: two 32-lane groups per wavefront,
// and every step of the banded Cholesky / substitution sweeps is executed by all 64 lanes for the benefit of ONE row per
// problem.  On the benchmark's path QPs (n = 21 stations: 17 unknowns, 19 stations with two range forms each) that is
// ~2500 wave-level instructions per interior-point iteration for two problems, 44 M per 4096 scenes - a third of all the
// vector instructions of a planning step, in a step that is bound by FP64 instruction issue over all its kernels together.
//
// Here a problem takes a group of GP = 8 lanes (half a DPP row; GP = 16, a whole row, for BASELINE configs[4]'s 61
// stations) and every lane owns R CONSECUTIVE stations and R consecutive unknowns (GP = 8: R = 3 up to 26 stations, R = 4 up
// to 34; GP = 16, R = 4: up to 66).  The station / unknown phases are the same arithmetic, R times per lane; the sweeps become
// ceil(N / R) lane-steps in each of which a lane runs its R rows one after the other from its left neighbour's last KD rows
// (KD <= R), so a sweep costs about what it cost before - for eight problems instead of two.  Same algorithm, stopping rule
// and failure handling as range_qp_solve_wave_fast (Mehrotra predictor-corrector on the reduced normal equations); sums are
// associated differently, so results agree to round-off (2e-9 on the benchmark batch; QP outputs are compared at 1e-6 and
// certified against the reference-built KKT system).  Measured: 15.7 M vector instructions per 4096
// scenes (HISTORY.md section 3.3).
//
// The problems of a wavefront iterate in lock step until the slowest has converged (finished groups idle through the
// barriers): mean 10 iterations per wavefront of eight where a pair took 8.7.  Small batches (under 1024 scenes), which
// cannot fill the chip and are served by latency, keep the two-per-wavefront kernel.
#pragma once

#include "emp_qp_wave.h"

#ifndef EMP_QP_DEBUG_ROWS
#define EMP_QP_DEBUG_ROWS(...)
#endif

namespace emp {

// (floating-point contraction: see emp_qp_wave.h)
#pragma clang fp contract(fast)

// ---- reductions over an aligned group of GP = 8 or 16 lanes: two quad permutes, the half-row mirror, (16:) the row mirror
template <int GP, class Op>
__device__ __forceinline__ double oct_reduce(double v, Op op) {
    static_assert(GP == 8 || GP == 16, "groups of 8 or 16 lanes");
    v = op(v, dpp_move<0xB1>(v));      // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E>(v));      // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141>(v));     // row_half_mirror
    if constexpr (GP == 16) v = op(v, dpp_move<0x140>(v));     // row_mirror
    return v;
}
template <int GP>
__device__ __forceinline__ double oct_max(double v) { return oct_reduce<GP>(v, [](double a, double b) { return vmax(a, b); }); }
template <int GP>
__device__ __forceinline__ double oct_min(double v) { return oct_reduce<GP>(v, [](double a, double b) { return vmin(a, b); }); }
template <int GP>
__device__ __forceinline__ double oct_sum(double v) { return oct_reduce<GP>(v, [](double a, double b) { return a + b; }); }
template <int GP>
__device__ __forceinline__ bool oct_any(bool p) {
    const unsigned long long m = __ballot(p);
    return ((m >> ((threadIdx.x & 63) & ~(GP - 1))) & ((1ull << GP) - 1ull)) != 0ull;
}
// largest value of a group-uniform int over the groups of the wavefront (scalar result)
template <int GP>
__device__ __forceinline__ int oct_wave_max(int v) {
    int r = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int g = 1; g < 64 / GP; ++g) r = max(r, __builtin_amdgcn_readlane(v, GP * g));
    return r;
}

// ---------------------------------------------------------------------------------------------
// Banded Cholesky (half bandwidth 3) with R consecutive rows per lane, row j = gl * R + r.
// a[r][0..3] = A[j][j..j+3] on entry, the factor row U[j][..] on return; rinv[r] = 1 / U[j][j]; low[r][e] = U[j-e][e].
// A lane-step fetches the left neighbour's last three rows (the entries a row below them can need: six values) and then
// runs the lane's R rows in order - row r from rows r-1..r-3, its own where they exist, the neighbour's otherwise.  After
// lane-step k the rows of lanes 0..k are final; `steps` = ceil(N / R) of the largest problem in the wavefront.
// Entries past column N-1 are written as exact zeros at every step, rows past N-1 are identity rows: what a shift
// carries over a group's end (into idle lanes, or into the first lane of the next group) is zero, and a failed group
// is left with the identity factor, as in band_chol_group.
// ---------------------------------------------------------------------------------------------
template <int GP, int R>
__device__ __forceinline__ bool band_chol_rows(double (&a)[R][4], double (&rinv)[R], double (&low)[R][4], int N, int gl,
                                               bool active, int steps) {
    static_assert(R >= 3, "a lane must hold at least KD = 3 rows");
    double A[R][4];
    bool rowv[R], inband[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = gl * R + r;
        rowv[r] = active && j < N;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            A[r][d] = rowv[r] ? a[r][d] : (d == 0 ? 1.0 : 0.0);
            a[r][d] = A[r][d];
            inband[r][d] = rowv[r] && j + d < N;
        }
        rinv[r] = 1.0;
    }
    double diag[R];
    for (int k = 0; k < steps; ++k) {
        // neighbour's rows: n1 = row j0-1 (entries 1..3), n2 = row j0-2 (entries 2..3), n3 = row j0-3 (entry 3)
        const double n1_1 = lane_up1(a[R - 1][1]), n1_2 = lane_up1(a[R - 1][2]), n1_3 = lane_up1(a[R - 1][3]);
        const double n2_2 = lane_up1(a[R - 2][2]), n2_3 = lane_up1(a[R - 2][3]);
        const double n3_3 = lane_up1(a[R - 3][3]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // U[j-1][1..3], U[j-2][2..3], U[j-3][3]
            double u1_1, u1_2, u1_3, u2_2, u2_3, u3_3;
            if (r >= 1) { u1_1 = a[r - 1][1]; u1_2 = a[r - 1][2]; u1_3 = a[r - 1][3]; }
            else        { u1_1 = n1_1; u1_2 = n1_2; u1_3 = n1_3; }
            if (r >= 2)      { u2_2 = a[r - 2][2]; u2_3 = a[r - 2][3]; }
            else if (r == 1) { u2_2 = n1_2; u2_3 = n1_3; }
            else             { u2_2 = n2_2; u2_3 = n2_3; }
            if (r >= 3)      u3_3 = a[r - 3][3];
            else if (r == 2) u3_3 = n1_3;
            else if (r == 1) u3_3 = n2_3;
            else             u3_3 = n3_3;
            double acc0 = A[r][0], acc1 = A[r][1], acc2 = A[r][2];
            const double acc3 = A[r][3];
            acc0 = __builtin_fma(-u1_1, u1_1, acc0);
            acc1 = __builtin_fma(-u1_1, u1_2, acc1);
            acc2 = __builtin_fma(-u1_1, u1_3, acc2);
            acc0 = __builtin_fma(-u2_2, u2_2, acc0);
            acc1 = __builtin_fma(-u2_2, u2_3, acc1);
            acc0 = __builtin_fma(-u3_3, u3_3, acc0);
            diag[r] = acc0;
            const double rs = fast_rsqrt(acc0 > 0.0 ? acc0 : 1.0);
            // The selects below are what keeps the eight problems of a wavefront apart (round 6 tried without: A holds exact
            // zeros past column N-1 and every product that meets them is zero by induction - in exact arithmetic.  But a lane
            // that is not final yet iterates on garbage that squares itself every lane-step and overflows after seven of them,
            // 0 x NaN is NaN, and the first lane of the NEXT group reads these entries at every step: tools/qp_rows_test.hip).
            a[r][0] = acc0 * rs;
            a[r][1] = inband[r][1] ? acc1 * rs : 0.0;
            a[r][2] = inband[r][2] ? acc2 * rs : 0.0;
            a[r][3] = inband[r][3] ? acc3 * rs : 0.0;
            rinv[r] = rs;
        }
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double offd = fabs(a[r][1]) + fabs(a[r][2]) + fabs(a[r][3]);
        bad = bad || (rowv[r] && !(diag[r] > 0.0 && diag[r] < 1e300 && offd < 1e300));
    }
    const bool failed = oct_any<GP>(bad);
    if (failed) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            a[r][0] = 1.0;
            a[r][1] = a[r][2] = a[r][3] = 0.0;
            rinv[r] = 1.0;
        }
    }
    // column entries for the forward substitution: low[r][e] = U[j-e][e]
    {
        const double n1 = lane_up1(a[R - 1][1]), n1b = lane_up1(a[R - 1][2]), n1c = lane_up1(a[R - 1][3]);
        const double n2b = lane_up1(a[R - 2][2]), n2c = lane_up1(a[R - 2][3]);
        const double n3c = lane_up1(a[R - 3][3]);
        const bool first = gl == 0;                          // no row above row 0
#pragma unroll
        for (int r = 0; r < R; ++r) {
            low[r][0] = 0.0;
            low[r][1] = (r >= 1) ? a[r - 1][1] : (first ? 0.0 : n1);
            low[r][2] = (r >= 2) ? a[r - 2][2] : (r == 1 ? (first ? 0.0 : n1b) : (first ? 0.0 : n2b));
            low[r][3] = (r >= 3) ? a[r - 3][3] : (r == 2 ? (first ? 0.0 : n1c) : (r == 1 ? (first ? 0.0 : n2c) : (first ? 0.0 : n3c)));
        }
    }
    return !failed;
}

// solve U'U x = b with the factor of band_chol_rows; b[r] is overwritten with x.  Rows past N-1 must carry b = 0.
template <int R>
__device__ __forceinline__ void band_solve_rows(const double (&a)[R][4], const double (&rinv)[R], const double (&low)[R][4],
                                                double (&b)[R], int steps) {
    double rhs[R], y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rhs[r] = (fabs(b[r]) < 1e300) ? b[r] : 0.0;              // a NaN must not travel into the neighbouring group
        y[r] = rhs[r] * rinv[r];
    }
    for (int k = 0; k < steps; ++k) {                               // U' y = b, lanes 0..k final after lane-step k
        const double p1 = lane_up1(y[R - 1]), p2 = lane_up1(y[R - 2]), p3 = lane_up1(y[R - 3]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double y1 = (r >= 1) ? y[r - 1] : p1;
            const double y2 = (r >= 2) ? y[r - 2] : (r == 1 ? p1 : p2);
            const double y3 = (r >= 3) ? y[r - 3] : (r == 2 ? p1 : (r == 1 ? p2 : p3));
            double acc = __builtin_fma(-low[r][1], y1, rhs[r]);
            acc = __builtin_fma(-low[r][2], y2, acc);
            acc = __builtin_fma(-low[r][3], y3, acc);
            y[r] = acc * rinv[r];
        }
    }
    double x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = y[r] * rinv[r];
    for (int k = 0; k < steps; ++k) {                               // U x = y, from the last lane with rows down to lane 0
        const double q1 = lane_dn1(x[0]), q2 = lane_dn1(x[1]), q3 = lane_dn1(x[2]);
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const double x1 = (r + 1 < R) ? x[r + 1] : q1;
            const double x2 = (r + 2 < R) ? x[r + 2] : (r + 2 == R ? q1 : q2);
            const double x3 = (r + 3 < R) ? x[r + 3] : (r + 3 == R ? q1 : (r + 3 == R + 1 ? q2 : q3));
            double acc = __builtin_fma(-a[r][1], x1, y[r]);
            acc = __builtin_fma(-a[r][2], x2, acc);
            acc = __builtin_fma(-a[r][3], x3, acc);
            x[r] = acc * rinv[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) b[r] = x[r];
}

// ---------------------------------------------------------------------------------------------
// Interior point for the path QP (RangeQp<3, 2, 3>, window offset -2), R stations and R unknowns per lane.
// Q must be bound with bind_fast(mem, GP R, GP R, N, ns) behind at least 12 readable doubles (the coefficient slots of
// path_qp_group_rows): windows are read with plain lane offsets, up to three rows outside their arrays, and what lies
// outside a problem is replaced by zero after the load.  Q.g must be the same for every lane of the wavefront.
// returns (per group) 0 converged, 2 failed / infeasible.
// ---------------------------------------------------------------------------------------------
template <int GP, int R>
__device__ int path_qp_solve_rows(PathRangeQp& Q, int gl, bool live, int iter_cap, double* keep) {
    constexpr int F = 2, W = 3;
    const int N = Q.N, ns = Q.ns, rows = ns * F * 2;
    int state = (live && N > 0) ? 1 : 0;
    int iters = 0;
    bool acceptable = false;
    const int base = gl * R;                                   // first station / first unknown of this lane
    const int steps = oct_wave_max<GP>((state == 1) ? (N + R - 1) / R : 0);
    // form weights and their products, wave-uniform: scalar registers
    double g[F][W], gg0[W][F], gg1[W - 1][F], gg2[W - 2][F];   // gg_d[p][f] = g[f][p] g[f][p + d]
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int p = 0; p < W; ++p) g[f][p] = wave_uniform(Q.g[f][p]);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int p = 0; p < W; ++p) gg0[p][f] = wave_uniform(g[f][p] * g[f][p]);
#pragma unroll
        for (int p = 0; p < W - 1; ++p) gg1[p][f] = wave_uniform(g[f][p] * g[f][p + 1]);
        gg2[0][f] = wave_uniform(g[f][0] * g[f][2]);
    }
    // existence masks (bit i of a window = that entry belongs to the problem)
    unsigned tmask = 0, mmask = 0, uwm = 0, swm = 0, uxm = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (live && base + r < ns) tmask |= 1u << r;
        if (live && base + r < N) mmask |= 1u << r;
    }
#pragma unroll
    for (int i = 0; i < R + 2; ++i) {
        const int k = base - 2 + i;                            // unknown window of the lane's stations
        if (live && k >= 0 && k < N) uwm |= 1u << i;
        if (live && base + i < ns) swm |= 1u << i;             // station window of the lane's unknowns: stations base..base+R+1
    }
#pragma unroll
    for (int i = 0; i < R + 6; ++i) {
        const int k = base - 3 + i;                            // unknowns base-3 .. base+R+2 (Hessian rows)
        if (live && k >= 0 && k < N) uxm |= 1u << i;
    }
    auto ld = [](const double* a, int i, bool ok) {           // unconditional load, then select
        const double raw = a[i];
        return ok ? raw : 0.0;
    };
    // sum_p g[f][p] vec[t - 2 + p] for the lane's R stations
    auto win = [&](const double* vec, double (&out)[R][F]) {
        double vals[R + 2];
#pragma unroll
        for (int i = 0; i < R + 2; ++i) vals[i] = ld(vec, base - 2 + i, (uwm >> i) & 1u);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int f = 0; f < F; ++f) out[r][f] = (g[f][0] * vals[r] + g[f][1] * vals[r + 1]) + g[f][2] * vals[r + 2];
        }
    };
    // sum over the stations t = m + 2 - p whose windows contain unknown m, both forms: g[f][p] coef[t][f]
    auto gather = [&](const double* coef, double (&out)[R]) {
        double cw[R + 2][F];
#pragma unroll
        for (int i = 0; i < R + 2; ++i) {
#pragma unroll
            for (int f = 0; f < F; ++f) cw[i][f] = ld(coef, (base + i) * F + f, (swm >> i) & 1u);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < W; ++p) {
#pragma unroll
                for (int f = 0; f < F; ++f) acc += g[f][p] * cw[r + 2 - p][f];
            }
            out[r] = acc;
        }
    };
    auto bounds = [&](double (&c_it)[R][F], double (&lo_it)[R][F], double (&hi_it)[R][F]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = (tmask >> r) & 1u;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const double c = Q.c[(base + r) * F + f], lo = Q.lo[(base + r) * F + f], hi = Q.hi[(base + r) * F + f];
                c_it[r][f] = ok ? c : 0.0;
                lo_it[r][f] = ok ? lo : -1e300;
                hi_it[r][f] = ok ? hi : 1e300;
            }
        }
    };

    double su[R][F], sl[R][F], zu[R][F], zl[R][F];
    double pdiag = 0.0, qabs = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = (mmask >> r) & 1u;
        pdiag = fmax(pdiag, ld(Q.P, (base + r) * 4, ok));
        qabs = fmax(qabs, fabs(ld(Q.q, base + r, ok)));
    }
    const double pscale = oct_max<GP>(pdiag);                     // largest Hessian diagonal
    const double qscale = fmax(oct_max<GP>(qabs), 1.0);
    {
        const double z0 = Q.initial_multiplier(pscale);
        double v[R][F], c_it[R][F], lo_it[R][F], hi_it[R][F];
        bounds(c_it, lo_it, hi_it);
        win(Q.u, v);
        double smin = 1e300;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                zu[r][f] = zl[r][f] = z0;
                su[r][f] = hi_it[r][f] - (c_it[r][f] + v[r][f]);
                sl[r][f] = (c_it[r][f] + v[r][f]) - lo_it[r][f];
                if ((tmask >> r) & 1u) smin = fmin(smin, fmin(su[r][f], sl[r][f]));
            }
        }
        smin = oct_min<GP>(smin);
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;   // slacks pushed to >= 1 (infeasible start)
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                su[r][f] += shift;
                sl[r][f] += shift;
            }
        }
    }

    bool restore = false;
    while (__any(state == 1)) {
        const bool run = state == 1;
        bool acc_now = false;
        // ---- 1: stations: residuals, reciprocals, coefficients of the dual residual and of the normal matrix
        double rpu[R][F], rpl[R][F], isu[R][F], isl[R][F], izu[R][F], izl[R][F];
        double rp_max = 0.0, zmax = 0.0, mu = 0.0;
        {
            double v[R][F], c_it[R][F], lo_it[R][F], hi_it[R][F];
            bounds(c_it, lo_it, hi_it);
            win(Q.u, v);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = run && ((tmask >> r) & 1u);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    rpu[r][f] = (c_it[r][f] + v[r][f]) - hi_it[r][f] + su[r][f];
                    rpl[r][f] = lo_it[r][f] - (c_it[r][f] + v[r][f]) + sl[r][f];
                    isu[r][f] = fast_rcp(su[r][f]);
                    isl[r][f] = fast_rcp(sl[r][f]);
                    izu[r][f] = fast_rcp(zu[r][f]);
                    izl[r][f] = fast_rcp(zl[r][f]);
                    if (ok) {
                        Q.tmp[(base + r) * F + f] = zu[r][f] - zl[r][f];
                        Q.wgt[(base + r) * F + f] = zu[r][f] * isu[r][f] + zl[r][f] * isl[r][f];
                        rp_max = vmax(rp_max, vmax_abs(rpu[r][f], rpl[r][f]));
                        zmax = vmax(zmax, vmax(zu[r][f], zl[r][f]));
                        mu += su[r][f] * zu[r][f] + sl[r][f] * zl[r][f];
                    }
                }
            }
        }
        __syncthreads();
        // ---- 2: unknowns: dual residual and the rows of the normal matrix P + G' diag(w) G
        double rd_m[R], fa[R][4], flow[R][4], frinv[R];
        {
            double ux[R + 6], gz[R];
#pragma unroll
            for (int i = 0; i < R + 6; ++i) ux[i] = ld(Q.u, base - 3 + i, (uxm >> i) & 1u);
            gather(Q.tmp, gz);
            double ww[R + 2][F];
#pragma unroll
            for (int i = 0; i < R + 2; ++i) {
#pragma unroll
                for (int f = 0; f < F; ++f) ww[i][f] = ld(Q.wgt, (base + i) * F + f, (swm >> i) & 1u);
            }
            double rd_max = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = run && ((mmask >> r) & 1u);
                const int m = base + r;
                // P u: the row's own band and the three rows above it (all stored bands have zeros past column N-1)
                double acc = Q.q[m];
#pragma unroll
                for (int d = 0; d < 4; ++d) acc += Q.P[m * 4 + d] * ux[r + 3 + d];
#pragma unroll
                for (int d = 1; d < 4; ++d) acc += ld(Q.P, (m - d) * 4 + d, m - d >= 0) * ux[r + 3 - d];
                rd_m[r] = ok ? acc + gz[r] : 0.0;
                rd_max = vmax_abs(rd_max, rd_m[r]);
                double e0 = Q.P[m * 4 + 0], e1 = Q.P[m * 4 + 1], e2 = Q.P[m * 4 + 2];
                const double e3 = Q.P[m * 4 + 3];
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    // station m + 2 - p carries weight index p for unknown m and p + d for unknown m + d
                    e0 += (ww[r + 2][f] * gg0[0][f] + ww[r + 1][f] * gg0[1][f]) + ww[r][f] * gg0[2][f];
                    e1 += ww[r + 2][f] * gg1[0][f] + ww[r + 1][f] * gg1[1][f];
                    e2 += ww[r + 2][f] * gg2[0][f];
                }
                fa[r][0] = ok ? e0 : 0.0;
                fa[r][1] = (ok && m + 1 < N) ? e1 : 0.0;
                fa[r][2] = (ok && m + 2 < N) ? e2 : 0.0;
                fa[r][3] = (ok && m + 3 < N) ? e3 : 0.0;
            }
            rd_max = oct_max<GP>(rd_max);
            rp_max = oct_max<GP>(rp_max);
            zmax = oct_max<GP>(zmax);
            mu = oct_sum<GP>(mu) / (double)rows;
            if (run) {
                const double dscale = fmax(qscale, zmax);
                if (rd_max <= Q.eps_d_rel * dscale && rp_max <= Q.eps_p && mu <= Q.eps_mu) state = 0;
                else if (!(mu == mu) || mu > 1e30 || (iters >= kQpStallIter && rp_max > kQpStallResidual)) state = 2;
                else if (rp_max > kQpStallResidual && zmax > kQpInfeasibleZ * pscale) state = 2;    // see range_qp_solve_wave_fast
                else if (iters >= kQpMaxIter || iters >= iter_cap) state = acceptable ? 0 : 2;
                acc_now = rd_max <= 100.0 * Q.eps_d_rel * dscale && rp_max <= 10.0 * Q.eps_p && mu <= 1000.0 * Q.eps_mu;
                if (acc_now) acceptable = true;
                // Converged complementarity with the dual residual inside the acceptable band: at this mu the normal
                // matrix carries weights z / s of 1e15 and more, another iteration adds rounding noise to the residual
                // instead of removing it (and a few more destroy the iterate) - stop here.
                if (state == 1 && acc_now && rp_max <= Q.eps_p && mu <= Q.eps_mu) state = 0;
                if (state == 0 && acceptable && !acc_now) restore = true;      // left through the fallback: last good iterate
                EMP_QP_DEBUG_ROWS("R it %d rd %.3e rp %.3e mu %.3e zmax %.3e -> state %d acc %d\n", iters, rd_max, rp_max, mu, zmax, state, (int)acceptable);
            }
        }
        if (run && acc_now) {                                 // remember the iterate the fallback exits return (the problem's own
#pragma unroll                                                // unknowns only: `keep` shares the coefficient slots, whose entries past N stay 0)
            for (int r = 0; r < R; ++r)
                if ((mmask >> r) & 1u) keep[base + r] = Q.u[base + r];
        }
        // Nobody left to iterate (the pass in which the last group of the wavefront converges or gives up): leave here.  The rest of
        // the body - factorisation, two solves, the step - would run fully masked: 4.5 us of a 6.6-us pass, once per wavefront
        // (round 6, found by tools/qp_phase_probe.py: a solve capped at K iterations cost K + 1 passes).  Wave-uniform: one
        // wavefront per block, so every lane of the block leaves together.
        if (!__any(state == 1)) break;
        const bool go = state == 1;
        // ---- 3: factorisation
#ifdef EMP_QP_PROBE_SKIP_CHOL
        bool okf = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            frinv[r] = 1.0 / (fa[r][0] + 1.0);
#pragma unroll
            for (int d = 0; d < 4; ++d) flow[r][d] = 0.0;
        }
#else
        const bool okf = band_chol_rows<GP, R>(fa, frinv, flow, N, gl, go, steps);
#endif
        EMP_QP_DEBUG_ROWS("R    chol ok %d\n", (int)okf);
        if (go && !okf) {
            state = acceptable ? 0 : 2;
            if (acceptable && !acc_now) restore = true;
        }
        const bool go2 = state == 1;
        // ---- 4: predictor
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (go2 && ((tmask >> r) & 1u)) {
#pragma unroll
                for (int f = 0; f < F; ++f)
                    Q.tmp[(base + r) * F + f] = -(((zu[r][f] * isu[r][f]) * rpu[r][f] - zu[r][f]) - ((zl[r][f] * isl[r][f]) * rpl[r][f] - zl[r][f]));
            }
        }
        __syncthreads();
        double dua[R];
        gather(Q.tmp, dua);
#pragma unroll
        for (int r = 0; r < R; ++r) dua[r] = (go2 && ((mmask >> r) & 1u)) ? (-rd_m[r] + dua[r]) : 0.0;
#ifndef EMP_QP_PROBE_SKIP_SOLVE
        band_solve_rows<R>(fa, frinv, flow, dua, steps);
#endif
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (go2 && ((mmask >> r) & 1u)) Q.dua[base + r] = dua[r];
        __syncthreads();
        // ---- 5: affine step length, centring parameter, corrector coefficients
        double rcu[R][F], rcl[R][F];
        {
            double gda[R][F], dsua[R][F], dsla[R][F], dzua[R][F], dzla[R][F];
            win(Q.dua, gda);
            double ratio = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = go2 && ((tmask >> r) & 1u);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    dsua[r][f] = -rpu[r][f] - gda[r][f];
                    dsla[r][f] = -rpl[r][f] + gda[r][f];
                    dzua[r][f] = -zu[r][f] - (zu[r][f] * isu[r][f]) * dsua[r][f];
                    dzla[r][f] = -zl[r][f] - (zl[r][f] * isl[r][f]) * dsla[r][f];
                    if (ok)
                        ratio = vmax(ratio, vmax(vmax(-dsua[r][f] * isu[r][f], -dsla[r][f] * isl[r][f]),
                                                 vmax(-dzua[r][f] * izu[r][f], -dzla[r][f] * izl[r][f])));
                }
            }
            ratio = oct_max<GP>(ratio);
            const double a_aff = (ratio > 1.0) ? fast_rcp(ratio) : 1.0;
            double mu_aff = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (go2 && ((tmask >> r) & 1u)) {
#pragma unroll
                    for (int f = 0; f < F; ++f)
                        mu_aff += (su[r][f] + a_aff * dsua[r][f]) * (zu[r][f] + a_aff * dzua[r][f]) +
                                  (sl[r][f] + a_aff * dsla[r][f]) * (zl[r][f] + a_aff * dzla[r][f]);
                }
            }
            mu_aff = oct_sum<GP>(mu_aff) / (double)rows;
            double sigma = (mu > 0.0) ? mu_aff * fast_rcp(mu) : 0.0;
            sigma = sigma * sigma * sigma;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = go2 && ((tmask >> r) & 1u);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    rcu[r][f] = su[r][f] * zu[r][f] + dsua[r][f] * dzua[r][f] - sigma * mu;
                    rcl[r][f] = sl[r][f] * zl[r][f] + dsla[r][f] * dzla[r][f] - sigma * mu;
                }
                if (ok) {
#pragma unroll
                    for (int f = 0; f < F; ++f)
                        Q.tmp[(base + r) * F + f] = -((zu[r][f] * rpu[r][f] - rcu[r][f]) * isu[r][f] - (zl[r][f] * rpl[r][f] - rcl[r][f]) * isl[r][f]);
                }
            }
        }
        __syncthreads();
        double du[R];
        gather(Q.tmp, du);
#pragma unroll
        for (int r = 0; r < R; ++r) du[r] = (go2 && ((mmask >> r) & 1u)) ? (-rd_m[r] + du[r]) : 0.0;
#ifndef EMP_QP_PROBE_SKIP_SOLVE
        band_solve_rows<R>(fa, frinv, flow, du, steps);
#endif
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (go2 && ((mmask >> r) & 1u)) Q.rhs[base + r] = du[r];
        __syncthreads();
        // ---- 6: step length and update
        {
            double gd[R][F], dsu[R][F], dsl[R][F], dzu[R][F], dzl[R][F];
            win(Q.rhs, gd);
            double ratio = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = go2 && ((tmask >> r) & 1u);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    dsu[r][f] = -rpu[r][f] - gd[r][f];
                    dsl[r][f] = -rpl[r][f] + gd[r][f];
                    dzu[r][f] = -(rcu[r][f] + zu[r][f] * dsu[r][f]) * isu[r][f];
                    dzl[r][f] = -(rcl[r][f] + zl[r][f] * dsl[r][f]) * isl[r][f];
                    if (ok)
                        ratio = vmax(ratio, vmax(vmax(-dsu[r][f] * isu[r][f], -dsl[r][f] * isl[r][f]),
                                                 vmax(-dzu[r][f] * izu[r][f], -dzl[r][f] * izl[r][f])));
                }
            }
            ratio = oct_max<GP>(ratio);
            const double tau = qp_step_fraction(mu);
            const double alpha = (ratio > tau) ? tau * fast_rcp(ratio) : 1.0;   // min(1, tau / ratio)
            EMP_QP_DEBUG_ROWS("R    alpha %.6e\n", alpha);
            if (go2) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((tmask >> r) & 1u) {
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            su[r][f] += alpha * dsu[r][f];
                            sl[r][f] += alpha * dsl[r][f];
                            zu[r][f] += alpha * dzu[r][f];
                            zl[r][f] += alpha * dzl[r][f];
                        }
                    }
                    if ((mmask >> r) & 1u) Q.u[base + r] = Q.u[base + r] + alpha * du[r];      // (u is re-read, not kept: registers)
                }
                ++iters;
            }
        }
        __syncthreads();
    }
    if (restore) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((mmask >> r) & 1u) Q.u[base + r] = keep[base + r];
    }
    __syncthreads();
    Q.iters = iters;
    return state;
}

// doubles of LDS one problem of path_qp_group_rows<GP, R> needs: GP R + 4 coefficient slots (n + 2 of them are used) and the
// solver's arrays at capacity GP R
// (round 5: the last acceptable iterate lives in the coefficient slots 3 .. GP R + 2, which hold nothing between the set-up - it
// reads the three fixed start coefficients only - and the read-out that fills them from the final iterate: GP R doubles less)
template <int GP, int R>
__host__ __device__ constexpr int path_qp_words_rows() { return (GP * R + 4) + PathRangeQp::words_fast(GP * R, GP * R); }

// ---------------------------------------------------------------------------------------------
// Path QP on one group of GP lanes (64 / GP scenes per wavefront); n <= GP R + 2 stations (GP = 8: R = 3: 26, R = 4: 34;
// GP = 16, R = 4: 66 - BASELINE configs[4]'s 61).  Same contract as path_qp_group: every lane of the wavefront must call
// it; returns (per group) 0 ok, 1 infeasible, 2 failed.  lds: this group's path_qp_words_rows<GP, R>() doubles.
// ---------------------------------------------------------------------------------------------
template <int GP, int R>
__device__ inline int path_qp_group_rows(double* lds, const double* l_min, const double* l_max, int n, double l0, double dl0,
                                         double ddl0, const PathQpParams& prm, double* out_l, int* iters_out, bool live,
                                         int debug_stage = 0) {
    constexpr int kCc = GP * R + 4;
    const int gl = (threadIdx.x & 63) & (GP - 1);
    *iters_out = 0;
    PathRangeQp Q;
    double* cc = lds;
    const int nn = live ? n : 4;
    Q.bind_fast(lds + kCc, GP * R, GP * R, nn - 4 > 0 ? nn - 4 : 0, nn - 2 > 0 ? nn - 2 : 0);
    int rc = path_qp_setup_group<GP>(Q, cc, l_min, l_max, n, l0, dl0, ddl0, prm, gl, live);
    if (!live) rc = 2;
    if (EMP_DEV_HOOKS && debug_stage == 2) rc = 2;          // development timing: stop behind the set-up (tools/qp_phase_probe.py)
    __syncthreads();
    bool ok = rc == 0;
    const int base = gl * R;
    // ---- start from the unconstrained minimiser P u = -q
    {
        double fa[R][4], flow[R][4], frinv[R], b0[R];
        const bool act = ok && Q.N > 0;
        const int steps = oct_wave_max<GP>(act ? (Q.N + R - 1) / R : 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool has = act && base + r < Q.N;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const double raw = Q.P[(base + r) * 4 + d];
                fa[r][d] = has ? raw : 0.0;
            }
            const double qr = Q.q[base + r];
            b0[r] = has ? -qr : 0.0;
        }
        const bool okc = band_chol_rows<GP, R>(fa, frinv, flow, Q.N, gl, act, steps);
        band_solve_rows<R>(fa, frinv, flow, b0, steps);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (act && base + r < Q.N) Q.u[base + r] = b0[r];
        if (ok && !okc) rc = 2;
    }
    if (EMP_DEV_HOOKS && debug_stage == 3) rc = 2;          // ... behind the start point
    __syncthreads();
    ok = rc == 0;
    const int cap_it = debug_stage >= 10 ? debug_stage - 10 : 1000;
    const int rs = path_qp_solve_rows<GP, R>(Q, gl, ok && Q.N > 0, cap_it, cc + 3);
    if (ok && Q.N > 0) {
        *iters_out = Q.iters;
        if (rs && (debug_stage < 10 || !EMP_DEV_HOOKS)) rc = rs;   // a failed solve is never masked in a product build
    }
    ok = rc == 0;
    if (ok && Q.N == 0) {                                // nothing free: only check the constant forms
        bool bad = false;
        for (int it = gl; it < Q.ns * 2; it += GP)
            if (Q.c[it] > Q.hi[it] + 1e-9 || Q.c[it] < Q.lo[it] - 1e-9) bad = true;
        if (oct_any<GP>(bad)) rc = 1;
    }
    ok = rc == 0;
    for (int m = gl; m < (ok ? Q.N : 0); m += GP) cc[m + 3] = Q.u[m];
    __syncthreads();
    for (int i = gl; i < (ok ? n : 0); i += GP) out_l[i] = (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0;
    __syncthreads();
    return rc;
}

#pragma clang fp contract(off)

}  // namespace emp
