// emp_qp_wave.h - wave-cooperative solvers for the banded range QP of emp_qp_core.h (device only).
//
// One problem per GROUP of G lanes (G = 64: one problem per wavefront; G = 32: two problems side by side - two
// scenes of the path QP, or the x and y smoothing problems of one polyline).  Three tiers, same algorithm
// (Mehrotra predictor-corrector on the reduced normal equations) and stopping rule as RangeQp::solve_scalar:
//   * range_qp_solve_wave_fast - one station and one unknown per lane (N, ns <= G): per-station and per-unknown
//     state, the normal matrix and its factor in registers; vectors that lanes of different roles exchange go
//     through LDS.  The path QP of the cycle runs here (two scenes per wavefront: LIN / UNI variants, 163 VGPRs).
//   * box_qp_lanes / smooth_pair_lanes - the smoothing QP (G = I): registers only.
//   * range_qp_solve_wave - any size: arrays in LDS, serial factorisation by lane 0 of the group.
// The banded Cholesky and the substitutions of the first two tiers are lane-shift sweeps (band_chol_group /
// band_solve_group); norms and ratio tests are group reductions without LDS traffic (group_reduce).
// Blocks must consist of exactly one wavefront so that __syncthreads() is a cheap wave-level LDS fence.
//
// Different summation order than the scalar solver, so results agree to round-off, not bitwise (QP outputs are
// compared at 1e-6).
#pragma once

#include <hip/hip_runtime.h>

#include "emp_qp_core.h"

// Development hook: tools/qp_wave_bench.hip defines EMP_QP_PROF(i) to accumulate clock ticks per solver section.
#ifndef EMP_QP_PROF
#define EMP_QP_PROF(i)
#endif
#ifndef EMP_QP_DEBUG
#define EMP_QP_DEBUG(...)
#endif

namespace emp {

// Floating-point contraction is ON from here to the end of this header (and in emp_qp_rows.h, emp_smooth_rows.h): the library is
// compiled with -ffp-contract=off because the DP is specified operation by operation and compared bit for bit (emp_core.h), but an
// interior-point iteration is not - its result is a fixed point that does not depend on how a step is rounded, QP outputs are
// judged at 1e-6 and certified against the KKT system - and a fused multiply-add is one instruction instead of two and rounds once.
// Round 6: the rows form's loop went from 423 mul + 394 add + 178 fma to 291 + 452 of 1794 -> 1477 vector instructions, its spills to
// accumulation registers from 101 to 36; results moved by 1e-9 at most (tools/lib_equal.py --diff).
#pragma clang fp contract(fast)

// Hardware reciprocal / reciprocal square root seeds (v_rcp_f64, v_rsq_f64: ~2^-27 relative) plus Newton steps.
// Not correctly rounded (a few ulp): used only inside the interior-point iteration, whose result is a fixed
// point that does not depend on how the steps are rounded.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);     // (round 6 measured ONE step: same results to 1e-9, same duration - kept at two)
    return r;
}
// max / min of two doubles as ONE v_max_f64 / v_min_f64.  fmax() compiles to the same instruction behind a canonicalisation
// (v_max_f64 x, x, x) of every operand that might be a signalling NaN - 71 of the path QP loop's 163 v_max_f64.  The IPM's norms
// and ratio tests need no NaN semantics beyond the instruction's own (a NaN operand loses against a number, like fmax; the
// callers test `mu == mu` for a poisoned iterate).
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_abs(double a, double b) {     // max(|a|, |b|)
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    // Newton for 1/sqrt(x): r <- r + r * (1 - x r^2) / 2
    // ONE step: the seed's 2^-27 becomes ~2^-53, a factor accurate to an ulp or two.  The second step this had
    // sat on the critical path of every Cholesky sweep (path QP -2 %, smoothing QP -5 %) without changing a result.
    const double e = __builtin_fma(-x * r, r, 1.0);
    r = __builtin_fma(0.5 * r, e, r);
    return r;
}

constexpr double kQpInfeasibleZ = 1e5;    // multipliers beyond this many Hessian diagonals with a stalled primal residual: infeasible

// All-lanes reductions over a group of G = 32 or 64 lanes without LDS traffic: four DPP butterfly levels inside
// each row of 16 lanes (quad_perm x2, row_half_mirror, row_mirror: after level k every aligned block of 2^k
// lanes holds its own reduction, so a mirror works as the next exchange), then gfx950's row swaps
// (v_permlane16_swap / v_permlane32_swap) across rows.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xF, 0xF, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xF, 0xF, true);
    return r.d;
}
// the partner value of the xor-16 (ROWS = 16) or xor-32 (ROWS = 32) exchange, for values uniform per row
template <int ROWS>
__device__ __forceinline__ void row_swap(double v, double* mine, double* other) {
    union { double d; int i[2]; } a, m, o;
    a.d = v;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        if constexpr (ROWS == 16) {
            const auto r = __builtin_amdgcn_permlane16_swap(a.i[w], a.i[w], false, false);
            m.i[w] = r[0];
            o.i[w] = r[1];
        } else {
            const auto r = __builtin_amdgcn_permlane32_swap(a.i[w], a.i[w], false, false);
            m.i[w] = r[0];
            o.i[w] = r[1];
        }
    }
    *mine = m.d;
    *other = o.d;
}
template <int G, class Op>
__device__ __forceinline__ double group_reduce(double v, Op op) {
    static_assert(G == 32 || G == 64, "group size must be 32 or 64");
    v = op(v, dpp_move<0xB1>(v));      // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E>(v));      // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141>(v));     // row_half_mirror
    v = op(v, dpp_move<0x140>(v));     // row_mirror
    double x, y;
    row_swap<16>(v, &x, &y);
    v = op(x, y);
    if constexpr (G == 64) {
        row_swap<32>(v, &x, &y);
        v = op(x, y);
    }
    return v;
}
template <int G>
__device__ __forceinline__ double group_max(double v) {
    return group_reduce<G>(v, [](double a, double b) { return fmax(a, b); });
}
template <int G>
__device__ __forceinline__ double group_min(double v) {
    return group_reduce<G>(v, [](double a, double b) { return fmin(a, b); });
}
template <int G>
__device__ __forceinline__ bool group_any(bool p) {             // true if p holds on any lane of MY group
    const unsigned long long m = __ballot(p);
    if constexpr (G == 64) return m != 0ull;
    else return ((m >> ((threadIdx.x & 63) & ~(G - 1))) & ((1ull << G) - 1ull)) != 0ull;     // G = 32, 16, 8: aligned groups
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    return group_reduce<G>(v, [](double a, double b) { return a + b; });
}

// ---------------------------------------------------------------------------------------------
// Group-cooperative banded Cholesky / triangular solves for N <= G: lane gl of the group owns row gl of the
// band in registers.  Each elimination step broadcasts one row with v_readlane (uniform lane index), so the
// serial recurrence runs at register latency instead of LDS round trips.
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ double group_bcast(double v, int k) {     // value of lane k of MY group (k uniform)
    union { double d; int i[2]; } a, r;
    a.d = v;
    if constexpr (G == 64) {
        r.i[0] = __builtin_amdgcn_readlane(a.i[0], k);
        r.i[1] = __builtin_amdgcn_readlane(a.i[1], k);
        return r.d;
    } else {
        static_assert(G == 32, "group size must be 32 or 64");
        // two groups per wavefront need two different source lanes: one ds_bpermute per dword
        const int src = (((threadIdx.x & 63) & 32) + k) << 2;
        r.i[0] = __builtin_amdgcn_ds_bpermute(src, a.i[0]);
        r.i[1] = __builtin_amdgcn_ds_bpermute(src, a.i[1]);
        return r.d;
    }
}

template <int G>
__device__ __forceinline__ int group_nmax(int N) {                 // largest N over the groups of this wavefront
    int n0 = __builtin_amdgcn_readlane(N, 0);
    if constexpr (G == 32) n0 = max(n0, __builtin_amdgcn_readlane(N, 32));
    return n0;
}

// Whole-wavefront shifts by one lane (DPP wave_shr:1 / wave_shl:1, a VALU move: no LDS traffic).  Lanes without
// a source read 0.  With G = 32 the shift crosses from one group into the next; every use below multiplies the
// shifted-in value by a band entry that is exactly 0 there, or selects it away.
__device__ __forceinline__ double lane_up1(double v) {               // lane i <- lane i - 1
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x138, 0xF, 0xF, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x138, 0xF, 0xF, true);
    return r.d;
}
__device__ __forceinline__ double lane_dn1(double v) {               // lane i <- lane i + 1
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x130, 0xF, 0xF, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x130, 0xF, 0xF, true);
    return r.d;
}

// Band row in registers: a[0..KD] = A[gl][gl..gl+KD] (entries past column N-1 are 0; lanes gl >= N and groups
// that are not `active` carry zeros).  On return a[] holds the factor row U[gl][..], rinv = 1/U[gl][gl] and
// low[e] = U[gl-e][e] (the column entries the forward substitution needs).  ok == false if a pivot of this
// group was <= 0.
//
// Left-looking Cholesky as a lane-parallel sweep: row j of U only needs rows j-1..j-KD,
//     U[j][j+d] = (A[j][j+d] - sum_{e=1..KD-d} U[j-e][e] U[j-e][e+d]) / U[j][j],
// so EVERY lane recomputes its row from A and its neighbours' current rows (fetched with lane shifts) at every
// step; after step k rows 0..k are final.  No broadcasts, no selects on the pivot index, no LDS.  Rows that
// are not final yet hold garbage (possibly NaN) that is overwritten, never accumulated.  Entries past column
// N-1 are written as exact zeros at every step: those are the entries a shift carries across the end of a row
// block (into idle lanes, or with G = 32 into the first rows of the next group), so nothing else needs a mask.
template <int G, int KD>
__device__ __forceinline__ bool band_chol_group(double (&a)[KD + 1], double& rinv, double (&low)[KD + 1], int N, int gl,
                                                bool active) {
    static_assert(KD >= 1 && KD <= 3, "band_chol_group handles half bandwidths 1..3");
    const bool row = active && gl < N;
    double A[KD + 1];
#pragma unroll
    for (int d = 0; d <= KD; ++d) A[d] = row ? a[d] : (d == 0 ? 1.0 : 0.0);     // idle lanes: identity rows
#pragma unroll
    for (int d = 0; d <= KD; ++d) a[d] = A[d];
    bool inband[KD + 1];
#pragma unroll
    for (int d = 0; d <= KD; ++d) inband[d] = row && gl + d < N;
    double diag = A[0];
    rinv = 1.0;
    const int nmax = group_nmax<G>(N);                             // groups of one wavefront may differ in size
    for (int k = 0; k < nmax; ++k) {
        // s[e][c] = U[gl-e][e+c] (c = 0..KD-e): row gl-e shifted up by e lanes
        double s1[KD], s2[KD > 1 ? KD - 1 : 1], s3[1];
#pragma unroll
        for (int c = 0; c < KD; ++c) s1[c] = lane_up1(a[1 + c]);
        if constexpr (KD >= 2) {
#pragma unroll
            for (int c = 0; c < KD - 1; ++c) s2[c] = lane_up1(s1[1 + c]);
        }
        if constexpr (KD >= 3) s3[0] = lane_up1(s2[1]);
        const double h1 = s1[0];
        double acc[KD + 1];
#pragma unroll
        for (int d = 0; d <= KD; ++d) acc[d] = A[d];
#pragma unroll
        for (int c = 0; c < KD; ++c) acc[c] = __builtin_fma(-h1, s1[c], acc[c]);
        if constexpr (KD >= 2) {
            const double h2 = s2[0];
#pragma unroll
            for (int c = 0; c < KD - 1; ++c) acc[c] = __builtin_fma(-h2, s2[c], acc[c]);
        }
        if constexpr (KD >= 3) {
            acc[0] = __builtin_fma(-s3[0], s3[0], acc[0]);
        }
        diag = acc[0];
        const double r = fast_rsqrt(diag > 0.0 ? diag : 1.0);
        a[0] = diag * r;
#pragma unroll
        for (int d = 1; d <= KD; ++d) a[d] = inband[d] ? acc[d] * r : 0.0;
        rinv = r;
    }
    // A failed factorisation (pivot <= 0, or anything non-finite) must not leave its garbage in the registers:
    // with G = 32 the substitutions' shifts would carry a NaN into the neighbouring group, where 0 * NaN poisons a
    // healthy problem.  The failed group gets the identity factor (the caller stops it anyway).
    double offd = 0.0;
#pragma unroll
    for (int d = 1; d <= KD; ++d) offd += fabs(a[d]);
    const bool bad = row && !(diag > 0.0 && diag < 1e300 && offd < 1e300);
    const bool failed = group_any<G>(bad);
    if (failed) {
        a[0] = 1.0;
#pragma unroll
        for (int d = 1; d <= KD; ++d) a[d] = 0.0;
        rinv = 1.0;
    }
    // column entries for the forward substitution: low[e] = U[gl-e][e]
    low[0] = 0.0;
    {
        double t = lane_up1(a[1]);
        low[1] = (gl >= 1) ? t : 0.0;
        if constexpr (KD >= 2) {
            t = lane_up1(lane_up1(a[2]));
            low[2] = (gl >= 2) ? t : 0.0;
        }
        if constexpr (KD >= 3) {
            t = lane_up1(lane_up1(lane_up1(a[3])));
            low[3] = (gl >= 3) ? t : 0.0;
        }
    }
    return !failed;
}

// solve U'U x = b with the factor from band_chol_group; b (one entry per lane) is overwritten with x.
// Same sweep idea: y_j = (b_j - sum_e U[j-e][e] y_{j-e}) / U[j][j] is recomputed by every lane at every step from
// its neighbours' current values; after step k entries 0..k are final (backward: N-1..N-1-k).  All operands are
// finite (band_chol_group replaces a failed factor by the identity, a non-finite right-hand side is zeroed), and
// shifted-in values from outside the group meet a zero coefficient.
template <int G, int KD>
__device__ __forceinline__ void band_solve_group(const double (&a)[KD + 1], double rinv, const double (&low)[KD + 1],
                                                 double& b, int N, int gl) {
    const int nmax = group_nmax<G>(N);
    const double rhs = (fabs(b) < 1e300) ? b : 0.0;               // a NaN must not travel into the neighbouring group
    double y = rhs * rinv;
    for (int k = 1; k < nmax; ++k) {                               // U' y = b
        const double y1 = lane_up1(y);
        double acc = __builtin_fma(-low[1], y1, rhs);
        if constexpr (KD >= 2) {
            const double y2 = lane_up1(y1);
            acc = __builtin_fma(-low[2], y2, acc);
            if constexpr (KD >= 3) acc = __builtin_fma(-low[3], lane_up1(y2), acc);
        }
        y = acc * rinv;
    }
    double x = y * rinv;
    for (int k = 1; k < nmax; ++k) {                               // U x = y
        const double x1 = lane_dn1(x);
        double acc = __builtin_fma(-a[1], x1, y);
        if constexpr (KD >= 2) {
            const double x2 = lane_dn1(x1);
            acc = __builtin_fma(-a[2], x2, acc);
            if constexpr (KD >= 3) acc = __builtin_fma(-a[3], lane_dn1(x2), acc);
        }
        x = acc * rinv;
    }
    b = x;
}

// NP independent problems in the same lanes (same N): one loop, NP interleaved dependency chains.
template <int G, int KD, int NP>
__device__ __forceinline__ bool band_chol_group_n(double (&a)[NP][KD + 1], double (&rinv)[NP], double (&low)[NP][KD + 1],
                                                  int N, int gl, const bool (&active)[NP]) {
    static_assert(KD >= 1 && KD <= 3, "band_chol_group_n handles half bandwidths 1..3");
    double A[NP][KD + 1], diag[NP];
    bool row[NP], inband[NP][KD + 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        row[p] = active[p] && gl < N;
#pragma unroll
        for (int d = 0; d <= KD; ++d) {
            A[p][d] = row[p] ? a[p][d] : (d == 0 ? 1.0 : 0.0);
            a[p][d] = A[p][d];
            inband[p][d] = row[p] && gl + d < N;
        }
        diag[p] = A[p][0];
        rinv[p] = 1.0;
    }
    const int nmax = group_nmax<G>(N);
    for (int k = 0; k < nmax; ++k) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double s1[KD], s2[KD > 1 ? KD - 1 : 1], s3[1];
#pragma unroll
            for (int c = 0; c < KD; ++c) s1[c] = lane_up1(a[p][1 + c]);
            if constexpr (KD >= 2) {
#pragma unroll
                for (int c = 0; c < KD - 1; ++c) s2[c] = lane_up1(s1[1 + c]);
            }
            if constexpr (KD >= 3) s3[0] = lane_up1(s2[1]);
            double acc[KD + 1];
#pragma unroll
            for (int d = 0; d <= KD; ++d) acc[d] = A[p][d];
#pragma unroll
            for (int c = 0; c < KD; ++c) acc[c] = __builtin_fma(-s1[0], s1[c], acc[c]);
            if constexpr (KD >= 2) {
#pragma unroll
                for (int c = 0; c < KD - 1; ++c) acc[c] = __builtin_fma(-s2[0], s2[c], acc[c]);
            }
            if constexpr (KD >= 3) acc[0] = __builtin_fma(-s3[0], s3[0], acc[0]);
            diag[p] = acc[0];
            const double r = fast_rsqrt(diag[p] > 0.0 ? diag[p] : 1.0);
            a[p][0] = diag[p] * r;
#pragma unroll
            for (int d = 1; d <= KD; ++d) a[p][d] = inband[p][d] ? acc[d] * r : 0.0;
            rinv[p] = r;
        }
    }
    bool bad = false;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (row[p] && !(diag[p] > 0.0)) {
            bad = true;
            rinv[p] = -1.0;                              // marks this problem's failed pivot for the caller
        }
        low[p][0] = 0.0;
        double t = lane_up1(a[p][1]);
        low[p][1] = (gl >= 1) ? t : 0.0;
        if constexpr (KD >= 2) {
            t = lane_up1(lane_up1(a[p][2]));
            low[p][2] = (gl >= 2) ? t : 0.0;
        }
        if constexpr (KD >= 3) {
            t = lane_up1(lane_up1(lane_up1(a[p][3])));
            low[p][3] = (gl >= 3) ? t : 0.0;
        }
    }
    return !group_any<G>(bad);
}

template <int G, int KD, int NP>
__device__ __forceinline__ void band_solve_group_n(const double (&a)[NP][KD + 1], const double (&rinv)[NP],
                                                   const double (&low)[NP][KD + 1], double (&b)[NP], int N, int gl) {
    const int nmax = group_nmax<G>(N);
    double rhs[NP], y[NP], x[NP], ri[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        ri[p] = rinv[p] > 0.0 ? rinv[p] : 1.0;
        rhs[p] = b[p];
        y[p] = rhs[p] * ri[p];
    }
    for (int k = 1; k < nmax; ++k) {                               // U' y = b
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double y1 = lane_up1(y[p]);
            double acc = __builtin_fma(-low[p][1], y1, rhs[p]);
            if constexpr (KD >= 2) {
                const double y2 = lane_up1(y1);
                acc = __builtin_fma(-low[p][2], y2, acc);
                if constexpr (KD >= 3) acc = __builtin_fma(-low[p][3], lane_up1(y2), acc);
            }
            y[p] = acc * ri[p];
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) x[p] = y[p] * ri[p];
    for (int k = 1; k < nmax; ++k) {                               // U x = y
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double x1 = lane_dn1(x[p]);
            double acc = __builtin_fma(-a[p][1], x1, y[p]);
            if constexpr (KD >= 2) {
                const double x2 = lane_dn1(x1);
                acc = __builtin_fma(-a[p][2], x2, acc);
                if constexpr (KD >= 3) acc = __builtin_fma(-a[p][3], lane_dn1(x2), acc);
            }
            x[p] = acc * ri[p];
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) b[p] = x[p];
}

// gather of per-row coefficients onto unknown m:  sum over (t, f, p) with t + off0 + p == m of g[f][p] * coef[t][f]
template <int KD, int F, int W>
__device__ __forceinline__ double gather_rows(const RangeQp<KD, F, W>& Q, int m, const double* coef) {
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < W; ++p) {
        const int t = m - Q.off0 - p;
        if (t >= 0 && t < Q.ns) {
#pragma unroll
            for (int f = 0; f < F; ++f) acc += Q.g[f][p] * coef[t * F + f];
        }
    }
    return acc;
}

// Q: this group's problem (arrays in LDS, u = starting guess).  gl = lane index inside the group.  `live` is
// false for a group without a problem (it still takes part in every barrier).  Returns 0 ok / 2 failed (per
// group; only meaningful where live).
template <int G, int KD, int F, int W>
__device__ int range_qp_solve_wave(RangeQp<KD, F, W>& Q, int gl, bool live, int iter_cap = 1000) {
    constexpr int B = KD + 1;
    const int N = Q.N, ns = Q.ns, items = ns * F, rows = items * 2;
    int state = (live && N > 0) ? 1 : 0;          // 1 running, 0 finished ok, 2 failed
    int iters = 0;
    bool acceptable = false;
    double qscale = 1.0;
    // ---- initial slacks / multipliers
    if (state == 1) {
        for (int m = gl; m < N; m += G) qscale = fmax(qscale, fabs(Q.q[m]));
        double smin = 1e300;
        for (int it = gl; it < items; it += G) {
            const int t = it / F, f = it - t * F;
            const double v = Q.c[it] + Q.form_val(t, f, Q.u);
            Q.s[it * 2] = Q.hi[it] - v;
            Q.s[it * 2 + 1] = v - Q.lo[it];
            smin = fmin(smin, fmin(Q.s[it * 2], Q.s[it * 2 + 1]));
        }
        qscale = group_max<G>(qscale);
        smin = group_min<G>(smin);
        double pscale = 0.0;
        for (int m = gl; m < N; m += G) pscale = fmax(pscale, Q.P[m * B]);
        const double z0 = Q.initial_multiplier(group_max<G>(pscale));
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;
        for (int it = gl; it < items; it += G) {
            Q.s[it * 2] += shift;
            Q.s[it * 2 + 1] += shift;
            Q.z[it * 2] = z0;
            Q.z[it * 2 + 1] = z0;
        }
    }
    __syncthreads();
    while (__any(state == 1)) {
        const bool run = state == 1;
        double mu = 0.0, sigma = 0.0, alpha = 1.0;
        // ---- A: per row item: residual pieces, gather coefficient for rd, barrier weight
        double rp_max = 0.0, zmax = 0.0;
        if (run) {
            for (int it = gl; it < items; it += G) {
                const int t = it / F, f = it - t * F;
                const double v = Q.c[it] + Q.form_val(t, f, Q.u);
                const double su = Q.s[it * 2], sl = Q.s[it * 2 + 1], zu = Q.z[it * 2], zl = Q.z[it * 2 + 1];
                rp_max = fmax(rp_max, fmax(fabs(v - Q.hi[it] + su), fabs(Q.lo[it] - v + sl)));
                mu += su * zu + sl * zl;
                zmax = fmax(zmax, fmax(zu, zl));
                Q.tmp[it] = zu - zl;
                Q.wgt[it] = zu / su + zl / sl;
            }
        }
        __syncthreads();
        // ---- B: per unknown: rd = P u + q + G'z, normal matrix row M[m][0..KD]
        double rd_max = 0.0;
        if (run) {
            for (int m = gl; m < N; m += G) {
                double acc = Q.q[m];
#pragma unroll
                for (int d = 0; d <= KD; ++d)
                    if (m + d < N) acc += Q.P[m * B + d] * Q.u[m + d];
#pragma unroll
                for (int d = 1; d <= KD; ++d)
                    if (m - d >= 0) acc += Q.P[(m - d) * B + d] * Q.u[m - d];
                acc += gather_rows(Q, m, Q.tmp);
                Q.rhs[m] = acc;
                rd_max = fmax(rd_max, fabs(acc));
#pragma unroll
                for (int d = 0; d <= KD; ++d) {
                    double e = Q.P[m * B + d];
                    if (m + d < N) {
#pragma unroll
                        for (int p = 0; p + d < W; ++p) {
                            const int t = m - Q.off0 - p;
                            if (t >= 0 && t < ns) {
#pragma unroll
                                for (int f = 0; f < F; ++f) e += Q.wgt[t * F + f] * Q.g[f][p] * Q.g[f][p + d];
                            }
                        }
                    }
                    Q.M[m * B + d] = e;
                }
            }
            rd_max = group_max<G>(rd_max);
            rp_max = group_max<G>(rp_max);
            zmax = group_max<G>(zmax);
            mu = group_sum<G>(mu) / (double)rows;
            const double dscale = fmax(qscale, zmax);
            if (rd_max <= Q.eps_d_rel * dscale && rp_max <= Q.eps_p && mu <= Q.eps_mu) state = 0;
            else if (!(mu == mu) || mu > 1e30 || (iters >= kQpStallIter && rp_max > kQpStallResidual)) state = 2;
            else if (iters >= kQpMaxIter || iters >= iter_cap) state = acceptable ? 0 : 2;
            const bool acc_now = rd_max <= 100.0 * Q.eps_d_rel * dscale && rp_max <= 10.0 * Q.eps_p && mu <= 1000.0 * Q.eps_mu;
            if (acc_now) acceptable = true;
            if (state == 1 && acc_now && rp_max <= Q.eps_p && mu <= Q.eps_mu) state = 0;     // see range_qp_solve_wave_fast
        }
        __syncthreads();
        if (!__any(state == 1)) break;          // nobody left to iterate: the rest of the pass would run fully masked (emp_qp_rows.h; one wavefront per block)
        const bool go = state == 1;
        // ---- C: factorisation.  N <= G: register-resident, one band row per lane; otherwise lane 0 out of LDS.
        int ok = 1;
        double fa[KD + 1], flow[KD + 1], frinv = 0.0;
        if (N <= G) {
#pragma unroll
            for (int d = 0; d <= KD; ++d) fa[d] = (go && gl < N) ? Q.M[gl * B + d] : 0.0;
            ok = band_chol_group<G, KD>(fa, frinv, flow, N, gl, go) ? 1 : 0;
            // (every lane of a group sees the same broadcast pivots, so `ok` is already uniform within the group)
        } else {
            if (go && gl == 0) ok = band_chol<KD>(Q.M, N) ? 1 : 0;
            ok = __shfl(ok, (threadIdx.x & 63) & ~(G - 1), 64);
        }
        if (go && !ok) state = acceptable ? 0 : 2;
        const bool go2 = state == 1;
        // ---- D: predictor coefficients  -(w rp - z)_upper + (w rp - z)_lower
        if (go2) {
            for (int it = gl; it < items; it += G) {
                const int t = it / F, f = it - t * F;
                const double v = Q.c[it] + Q.form_val(t, f, Q.u);
                const double su = Q.s[it * 2], sl = Q.s[it * 2 + 1], zu = Q.z[it * 2], zl = Q.z[it * 2 + 1];
                const double tu = (zu / su) * (v - Q.hi[it] + su) - zu, tl = (zl / sl) * (Q.lo[it] - v + sl) - zl;
                Q.tmp[it] = -(tu - tl);
            }
        }
        __syncthreads();
        if (go2)
            for (int m = gl; m < N; m += G) Q.dua[m] = -Q.rhs[m] + gather_rows(Q, m, Q.tmp);
        __syncthreads();
        if (N <= G) {
            double bb = (go2 && gl < N) ? Q.dua[gl] : 0.0;
            band_solve_group<G, KD>(fa, frinv, flow, bb, N, gl);
            if (go2 && gl < N) Q.dua[gl] = bb;
        } else if (go2 && gl == 0) {
            band_solve<KD>(Q.M, Q.dua, N);
        }
        __syncthreads();
        // ---- G: affine step length and centring parameter
        if (go2) {
            double a_loc = 1.0;
            for (int it = gl; it < items; it += G) {
                const int t = it / F, f = it - t * F;
                const double v = Q.c[it] + Q.form_val(t, f, Q.u), gd = Q.form_val(t, f, Q.dua);
                const double su = Q.s[it * 2], sl = Q.s[it * 2 + 1], zu = Q.z[it * 2], zl = Q.z[it * 2 + 1];
                const double dsu = -(v - Q.hi[it] + su) - gd, dsl = -(Q.lo[it] - v + sl) + gd;
                const double dzu = -zu - (zu / su) * dsu, dzl = -zl - (zl / sl) * dsl;
                if (dsu < 0.0) a_loc = fmin(a_loc, -su / dsu);
                if (dsl < 0.0) a_loc = fmin(a_loc, -sl / dsl);
                if (dzu < 0.0) a_loc = fmin(a_loc, -zu / dzu);
                if (dzl < 0.0) a_loc = fmin(a_loc, -zl / dzl);
            }
            const double a_aff = group_min<G>(a_loc);
            double mu_aff = 0.0;
            for (int it = gl; it < items; it += G) {
                const int t = it / F, f = it - t * F;
                const double v = Q.c[it] + Q.form_val(t, f, Q.u), gd = Q.form_val(t, f, Q.dua);
                const double su = Q.s[it * 2], sl = Q.s[it * 2 + 1], zu = Q.z[it * 2], zl = Q.z[it * 2 + 1];
                const double rpu = v - Q.hi[it] + su, rpl = Q.lo[it] - v + sl;
                const double dsu = -rpu - gd, dsl = -rpl + gd;
                const double dzu = -zu - (zu / su) * dsu, dzl = -zl - (zl / sl) * dsl;
                mu_aff += (su + a_aff * dsu) * (zu + a_aff * dzu) + (sl + a_aff * dsl) * (zl + a_aff * dzl);
            }
            mu_aff = group_sum<G>(mu_aff) / (double)rows;
            sigma = mu_aff / mu;
            sigma = sigma * sigma * sigma;
            // ---- H: corrector coefficients
            for (int it = gl; it < items; it += G) {
                const int t = it / F, f = it - t * F;
                const double v = Q.c[it] + Q.form_val(t, f, Q.u), gd = Q.form_val(t, f, Q.dua);
                const double su = Q.s[it * 2], sl = Q.s[it * 2 + 1], zu = Q.z[it * 2], zl = Q.z[it * 2 + 1];
                const double rpu = v - Q.hi[it] + su, rpl = Q.lo[it] - v + sl;
                const double dsu = -rpu - gd, dsl = -rpl + gd;
                const double dzu = -zu - (zu / su) * dsu, dzl = -zl - (zl / sl) * dsl;
                const double rcu = su * zu + dsu * dzu - sigma * mu, rcl = sl * zl + dsl * dzl - sigma * mu;
                Q.tmp[it] = -((zu * rpu - rcu) / su - (zl * rpl - rcl) / sl);
            }
        }
        __syncthreads();
        if (go2)
            for (int m = gl; m < N; m += G) Q.rhs[m] = -Q.rhs[m] + gather_rows(Q, m, Q.tmp);
        __syncthreads();
        if (N <= G) {                                           // rhs = du
            double bb = (go2 && gl < N) ? Q.rhs[gl] : 0.0;
            band_solve_group<G, KD>(fa, frinv, flow, bb, N, gl);
            if (go2 && gl < N) Q.rhs[gl] = bb;
        } else if (go2 && gl == 0) {
            band_solve<KD>(Q.M, Q.rhs, N);
        }
        __syncthreads();
        // ---- K/L: step length, then update rows (s, z) and, after a barrier, the unknowns
        if (go2) {
            double a_loc = 1e300;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) alpha = fmin(1.0, qp_step_fraction(mu) * group_min<G>(a_loc));
                for (int it = gl; it < items; it += G) {
                    const int t = it / F, f = it - t * F;
                    const double v = Q.c[it] + Q.form_val(t, f, Q.u);
                    const double gda = Q.form_val(t, f, Q.dua), gd = Q.form_val(t, f, Q.rhs);
                    const double su = Q.s[it * 2], sl = Q.s[it * 2 + 1], zu = Q.z[it * 2], zl = Q.z[it * 2 + 1];
                    const double rpu = v - Q.hi[it] + su, rpl = Q.lo[it] - v + sl;
                    const double dsua = -rpu - gda, dsla = -rpl + gda;
                    const double dzua = -zu - (zu / su) * dsua, dzla = -zl - (zl / sl) * dsla;
                    const double rcu = su * zu + dsua * dzua - sigma * mu, rcl = sl * zl + dsla * dzla - sigma * mu;
                    const double dsu = -rpu - gd, dsl = -rpl + gd;
                    const double dzu = -(rcu + zu * dsu) / su, dzl = -(rcl + zl * dsl) / sl;
                    if (pass == 0) {
                        if (dsu < 0.0) a_loc = fmin(a_loc, -su / dsu);
                        if (dsl < 0.0) a_loc = fmin(a_loc, -sl / dsl);
                        if (dzu < 0.0) a_loc = fmin(a_loc, -zu / dzu);
                        if (dzl < 0.0) a_loc = fmin(a_loc, -zl / dzl);
                    } else {
                        Q.z[it * 2] = zu + alpha * dzu;
                        Q.z[it * 2 + 1] = zl + alpha * dzl;
                        Q.s[it * 2] = su + alpha * dsu;
                        Q.s[it * 2 + 1] = sl + alpha * dsl;
                    }
                }
            }
        }
        __syncthreads();                                   // every row has read the old u
        if (go2) {
            for (int m = gl; m < N; m += G) Q.u[m] += alpha * Q.rhs[m];
            ++iters;
        }
        __syncthreads();
    }
    Q.iters = iters;
    return state;
}

// ---------------------------------------------------------------------------------------------
// Fast path: N <= G and ns <= G, i.e. one station (with its F forms) and one unknown per lane.  Per-station
// state (slacks, multipliers, bounds) and per-unknown state (Hessian row, factor row) stay in registers for the
// whole solve; only the vectors that lanes exchange (u, directions, per-row coefficients) go through LDS, read
// with clamped indices and zeroed weights instead of bounds branches.  Divisions: four reciprocals per row per
// iteration (1/s, 1/z); ratio tests use max(-d/x) with those reciprocals.
// ---------------------------------------------------------------------------------------------
// A value every lane of the wavefront holds alike, moved to scalar registers (there is no scalar FP64 arithmetic, so a
// uniform product computed by the vector unit would otherwise stay in a vector register for the whole solve).
__device__ __forceinline__ double wave_uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// LIN: Q was bound with bind_fast at a compile-time capacity and may be read up to three rows outside its arrays
// (the values are discarded): window and neighbour indices are then plain lane index + constant, not clamped, and
// every LDS access of the iteration is one of three per-lane base registers plus an immediate offset.
// UNI: Q.g is the same for every lane of the wavefront (path QP: a function of the launch parameters; NOT the speed
// QP, whose time step is per scene).
template <int G, bool LIN = false, bool UNI = false, int KD, int F, int W>
__device__ int range_qp_solve_wave_fast(RangeQp<KD, F, W>& Q, int gl, bool live, int iter_cap) {
    constexpr int B = KD + 1;
    const int N = Q.N, ns = Q.ns, rows = ns * F * 2;
    int state = (live && N > 0) ? 1 : 0;
    int iters = 0;
    bool acceptable = false;
    // ---- station role: window of unknowns t+off0 .. t+off0+W-1 (entries outside 0..N-1 are loaded from a valid address
    // and replaced by zero)
    const bool has_t = live && gl < ns;
    const int t = (LIN || has_t) ? gl : 0;
    auto pick = [](const double* a, int i, bool ok, double other) {      // unconditional load, then select
        const double raw = a[i];
        return ok ? raw : other;
    };
    // When the form weights g[f][p] are the same for every lane (UNI) they and their products stay in scalar registers;
    // what differs per lane is WHICH window entries exist, kept as bit masks and applied to the loaded values
    // (0 x g == g x 0: the sums below are the ones a masked-weight form produces).
    double g[F][W], gg[B][W][F];          // gg[d][p][f] = g[f][p] g[f][p+d]: normal-matrix weights
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int p = 0; p < W; ++p) g[f][p] = UNI ? wave_uniform(Q.g[f][p]) : Q.g[f][p];
    }
#pragma unroll
    for (int d = 0; d <= KD; ++d) {
#pragma unroll
        for (int p = 0; p < W; ++p) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const double pr = (p + d < W) ? g[f][p] * g[f][(p + d < W) ? p + d : 0] : 0.0;
                gg[d][p][f] = (UNI && p + d < W) ? wave_uniform(pr) : pr;
            }
        }
    }
    // Indices, bounds and Hessian rows are re-derived / re-read from LDS where an iteration needs them rather than
    // held across it: the solve is a chain of dependent instructions, and what it keeps live decides how many other
    // wavefronts (the next batch's front stage) fit beside it on the SIMD.
    const int k0 = t + Q.off0;
    unsigned kin = 0;
#pragma unroll
    for (int p = 0; p < W; ++p) kin |= (has_t && k0 + p >= 0 && k0 + p < N) ? (1u << p) : 0u;
    auto kc = [&](int p) { return LIN ? k0 + p : min(max(k0 + p, 0), max(N - 1, 0)); };
    double su[F], sl[F], zu[F], zl[F];
    auto bounds = [&](double (&c_it)[F], double (&lo_it)[F], double (&hi_it)[F]) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            c_it[f] = pick(Q.c, t * F + f, has_t, 0.0);
            lo_it[f] = pick(Q.lo, t * F + f, has_t, -1e300);
            hi_it[f] = pick(Q.hi, t * F + f, has_t, 1e300);
        }
    };
    auto win = [&](const double* vec, double (&out)[F]) {
        double vals[W];
#pragma unroll
        for (int p = 0; p < W; ++p) vals[p] = pick(vec, kc(p), (kin >> p) & 1u, 0.0);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < W; ++p) v += g[f][p] * vals[p];
            out[f] = v;
        }
    };
    // ---- unknown role: Hessian row, symmetric partners, and the stations whose windows contain unknown m
    const bool has_m = live && gl < N;
    const int m = (LIN || has_m) ? gl : 0;
    auto p_row = [&](int d) { return pick(Q.P, m * B + d, has_m && m + d < N, 0.0); };                                  // P[m][m+d]
    auto p_low = [&](int d) { return pick(Q.P, (LIN ? m - d : max(m - d, 0)) * B + d, has_m && m - d >= 0, 0.0); };     // P[m-d][m]
    auto u_up = [&](int d) { return pick(Q.u, (LIN || m + d < N) ? m + d : m, has_m && m + d < N, 0.0); };             // u[m+d]
    auto u_dn = [&](int d) { return pick(Q.u, (LIN || m - d >= 0) ? m - d : m, has_m && m - d >= 0, 0.0); };           // u[m-d]
    const int t0 = m - Q.off0;
    unsigned tin = 0;           // station m-off0-p exists
#pragma unroll
    for (int p = 0; p < W; ++p) tin |= (has_m && t0 - p >= 0 && t0 - p < ns) ? (1u << p) : 0u;
    auto ti = [&](int p) { return LIN ? t0 - p : min(max(t0 - p, 0), max(ns - 1, 0)); };
    const double q_m = pick(Q.q, m, has_m, 0.0);
    double u_m = pick(Q.u, m, has_m, 0.0);
    const double pscale = group_max<G>(p_row(0));     // largest Hessian diagonal
    {
        const double z0 = Q.initial_multiplier(pscale);
#pragma unroll
        for (int f = 0; f < F; ++f) zu[f] = zl[f] = z0;
    }
    auto gather = [&](const double* coef) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < W; ++p) {
#pragma unroll
            for (int f = 0; f < F; ++f) acc += g[f][p] * pick(coef, ti(p) * F + f, (tin >> p) & 1u, 0.0);
        }
        return acc;
    };
    // ---- initial slacks (pushed to >= 1) and unit multipliers
    double qscale = fmax(group_max<G>(has_m ? fabs(q_m) : 0.0), 1.0);
    {
        double v[F], c_it[F], lo_it[F], hi_it[F];
        bounds(c_it, lo_it, hi_it);
        win(Q.u, v);
        double smin = 1e300;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            su[f] = hi_it[f] - (c_it[f] + v[f]);
            sl[f] = (c_it[f] + v[f]) - lo_it[f];
            if (has_t) smin = fmin(smin, fmin(su[f], sl[f]));
        }
        smin = group_min<G>(smin);
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            su[f] += shift;
            sl[f] += shift;
        }
    }
    while (__any(state == 1)) {
        const bool run = state == 1;
        EMP_QP_PROF(0);
        // ---- 1: stations: residuals, reciprocals, rd gather coefficient, barrier weight
        double v[F], rpu[F], rpl[F], isu[F], isl[F], izu[F], izl[F], wu[F], wl[F];
        {
            double c_it[F], lo_it[F], hi_it[F];
            bounds(c_it, lo_it, hi_it);
            win(Q.u, v);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                rpu[f] = (c_it[f] + v[f]) - hi_it[f] + su[f];
                rpl[f] = lo_it[f] - (c_it[f] + v[f]) + sl[f];
            }
        }
        double rp_max = 0.0, zmax = 0.0, mu = 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            isu[f] = fast_rcp(su[f]);
            isl[f] = fast_rcp(sl[f]);
            izu[f] = fast_rcp(zu[f]);
            izl[f] = fast_rcp(zl[f]);
            wu[f] = zu[f] * isu[f];
            wl[f] = zl[f] * isl[f];
            if (run && has_t) {
                Q.tmp[t * F + f] = zu[f] - zl[f];
                Q.wgt[t * F + f] = wu[f] + wl[f];
                rp_max = fmax(rp_max, fmax(fabs(rpu[f]), fabs(rpl[f])));
                zmax = fmax(zmax, fmax(zu[f], zl[f]));
                mu += su[f] * zu[f] + sl[f] * zl[f];
            }
        }
        __syncthreads();
        EMP_QP_PROF(1);
        // ---- 2: unknowns: rd and the normal-matrix row (registers)
        double rd_m = 0.0;
        double fa[B], flow[B], frinv = 0.0;
#pragma unroll
        for (int d = 0; d <= KD; ++d) fa[d] = 0.0;
        if (run && has_m) {
            double acc = q_m;
#pragma unroll
            for (int d = 0; d <= KD; ++d) acc += p_row(d) * u_up(d);
#pragma unroll
            for (int d = 1; d <= KD; ++d) acc += p_low(d) * u_dn(d);
            rd_m = acc + gather(Q.tmp);
            double wv[W][F];
#pragma unroll
            for (int p = 0; p < W; ++p)
#pragma unroll
                for (int f = 0; f < F; ++f) wv[p][f] = pick(Q.wgt, ti(p) * F + f, (tin >> p) & 1u, 0.0);
#pragma unroll
            for (int d = 0; d <= KD; ++d) {
                double e = p_row(d);
#pragma unroll
                for (int p = 0; p < W; ++p)
#pragma unroll
                    for (int f = 0; f < F; ++f) e += wv[p][f] * gg[d][p][f];
                fa[d] = (m + d < N) ? e : 0.0;
            }
        }
        EMP_QP_PROF(2);
        {
            double rd_max = (run && has_m) ? fabs(rd_m) : 0.0;
            rd_max = group_max<G>(rd_max);
            rp_max = group_max<G>(rp_max);
            zmax = group_max<G>(zmax);
            mu = group_sum<G>(mu);
            mu /= (double)rows;
            if (run) {
                const double dscale = fmax(qscale, zmax);
                EMP_QP_DEBUG("it %d rd %.3e rp %.3e mu %.3e zmax %.3e\n", iters, rd_max, rp_max, mu, zmax);
                if (rd_max <= Q.eps_d_rel * dscale && rp_max <= Q.eps_p && mu <= Q.eps_mu) state = 0;
                else if (!(mu == mu) || mu > 1e30 || (iters >= kQpStallIter && rp_max > kQpStallResidual)) state = 2;
                // An infeasible problem shows long before that: its primal residual stalls while the multipliers grow
                // without bound (x3..100 per iteration, 1e9 and more by iteration 8-12); those of the feasible benchmark
                // problems stay below 250 Hessian diagonals.  Most of the batch's slowest problems were infeasible ones
                // running into the iteration-16 rule above, and a kernel lasts as long as its slowest problem.
                else if (rp_max > kQpStallResidual && zmax > kQpInfeasibleZ * pscale) state = 2;
                else if (iters >= kQpMaxIter || iters >= iter_cap) state = acceptable ? 0 : 2;
                const bool acc_now = rd_max <= 100.0 * Q.eps_d_rel * dscale && rp_max <= 10.0 * Q.eps_p && mu <= 1000.0 * Q.eps_mu;
                if (acc_now) acceptable = true;
                // Converged complementarity with the dual residual inside the acceptable band: at this mu the normal matrix
                // carries weights z / s of 1e15 and more, another iteration adds rounding noise to the residual instead of
                // removing it (benchmark scene 446: rd 5.6e-6, 1.1e-5 in another summation order, then 2e-2, 2e+4) - stop.
                if (state == 1 && acc_now && rp_max <= Q.eps_p && mu <= Q.eps_mu) state = 0;
            }
        }
        if (!__any(state == 1)) break;          // nobody left to iterate: the rest of the pass would run fully masked (emp_qp_rows.h; one wavefront per block)
        const bool go = state == 1;
        EMP_QP_PROF(3);
        // ---- 3: factorisation in registers
        const bool okf = band_chol_group<G, KD>(fa, frinv, flow, N, gl, go);
        EMP_QP_DEBUG("   chol ok %d\n", (int)okf);
        if (go && !okf) state = acceptable ? 0 : 2;
        const bool go2 = state == 1;
        EMP_QP_PROF(4);
        // ---- 4: predictor
        if (go2 && has_t) {
#pragma unroll
            for (int f = 0; f < F; ++f) Q.tmp[t * F + f] = -((wu[f] * rpu[f] - zu[f]) - (wl[f] * rpl[f] - zl[f]));
        }
        __syncthreads();
        double dua_m = (go2 && has_m) ? (-rd_m + gather(Q.tmp)) : 0.0;
        EMP_QP_PROF(5);
        band_solve_group<G, KD>(fa, frinv, flow, dua_m, N, gl);
        if (go2 && has_m) Q.dua[m] = dua_m;
        __syncthreads();
        EMP_QP_PROF(6);
        // ---- 5: affine step length, centring parameter, corrector coefficients
        double gda[F], dsua[F], dsla[F], dzua[F], dzla[F], rcu[F], rcl[F];
        win(Q.dua, gda);
        double ratio = 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            dsua[f] = -rpu[f] - gda[f];
            dsla[f] = -rpl[f] + gda[f];
            dzua[f] = -zu[f] - wu[f] * dsua[f];
            dzla[f] = -zl[f] - wl[f] * dsla[f];
            if (go2 && has_t)
                ratio = fmax(ratio, fmax(fmax(-dsua[f] * isu[f], -dsla[f] * isl[f]), fmax(-dzua[f] * izu[f], -dzla[f] * izl[f])));
        }
        ratio = group_max<G>(ratio);
        const double a_aff = (ratio > 1.0) ? fast_rcp(ratio) : 1.0;
        double mu_aff = 0.0;
        if (go2 && has_t) {
#pragma unroll
            for (int f = 0; f < F; ++f)
                mu_aff += (su[f] + a_aff * dsua[f]) * (zu[f] + a_aff * dzua[f]) + (sl[f] + a_aff * dsla[f]) * (zl[f] + a_aff * dzla[f]);
        }
        mu_aff = group_sum<G>(mu_aff) / (double)rows;
        double sigma = (mu > 0.0) ? mu_aff * fast_rcp(mu) : 0.0;
        sigma = sigma * sigma * sigma;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            rcu[f] = su[f] * zu[f] + dsua[f] * dzua[f] - sigma * mu;
            rcl[f] = sl[f] * zl[f] + dsla[f] * dzla[f] - sigma * mu;
            if (go2 && has_t) Q.tmp[t * F + f] = -((zu[f] * rpu[f] - rcu[f]) * isu[f] - (zl[f] * rpl[f] - rcl[f]) * isl[f]);
        }
        __syncthreads();
        double du_m = (go2 && has_m) ? (-rd_m + gather(Q.tmp)) : 0.0;
        EMP_QP_PROF(7);
        band_solve_group<G, KD>(fa, frinv, flow, du_m, N, gl);
        if (go2 && has_m) Q.rhs[m] = du_m;
        __syncthreads();
        EMP_QP_PROF(8);
        // ---- 6: step length and update
        double gd[F], dsu[F], dsl[F], dzu[F], dzl[F];
        win(Q.rhs, gd);
        ratio = 0.0;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            dsu[f] = -rpu[f] - gd[f];
            dsl[f] = -rpl[f] + gd[f];
            dzu[f] = -(rcu[f] + zu[f] * dsu[f]) * isu[f];
            dzl[f] = -(rcl[f] + zl[f] * dsl[f]) * isl[f];
            if (go2 && has_t)
                ratio = fmax(ratio, fmax(fmax(-dsu[f] * isu[f], -dsl[f] * isl[f]), fmax(-dzu[f] * izu[f], -dzl[f] * izl[f])));
        }
        ratio = group_max<G>(ratio);
        const double tau = qp_step_fraction(mu);
        const double alpha = (ratio > tau) ? tau * fast_rcp(ratio) : 1.0;   // min(1, tau / ratio)
        EMP_QP_DEBUG("   alpha %.6e\n", alpha);
        if (go2) {
            if (has_t) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    su[f] += alpha * dsu[f];
                    sl[f] += alpha * dsl[f];
                    zu[f] += alpha * dzu[f];
                    zl[f] += alpha * dzl[f];
                }
            }
            if (has_m) {
                u_m += alpha * du_m;
                Q.u[m] = u_m;
            }
            ++iters;
        }
        __syncthreads();
        EMP_QP_PROF(9);
    }
    Q.iters = iters;
    return state;
}

// ---------------------------------------------------------------------------------------------
// Path QP set-up, one station / one unknown per lane (same mathematics as path_qp_setup in emp_qp_core.h,
// written as gathers so that no lane has to accumulate into another lane's entry).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double bsp_a(int p) { return (p == 1) ? 4.0 / 6.0 : 1.0 / 6.0; }
__device__ __forceinline__ double bsp_b(int p, double ids2) { return (p == 1) ? -2.0 * ids2 : ids2; }
__device__ __forceinline__ double bsp_j(int p, double ids2) {
    return ((p == 0) ? -1.0 : (p == 1) ? 3.0 : (p == 2) ? -3.0 : 1.0) * ids2;
}
// Hessian entry between coefficient indices jp and jp + d (0 <= d <= 3) of the full (n+2)-coefficient problem
__device__ __forceinline__ double path_hess_entry(int jp, int d, int n, double wl2, double wd2, double wj2, double ids2) {
    double acc = 0.0;
    for (int i = max(0, jp + d - 2); i <= min(n - 1, jp); ++i)
        acc += wl2 * bsp_a(jp - i) * bsp_a(jp + d - i) + wd2 * bsp_b(jp - i, ids2) * bsp_b(jp + d - i, ids2);
    for (int i = max(0, jp + d - 3); i <= min(n - 2, jp); ++i) acc += wj2 * bsp_j(jp - i, ids2) * bsp_j(jp + d - i, ids2);
    return acc;
}

template <int G>
__device__ inline int path_qp_setup_group(PathRangeQp& Q, double* cc, const double* l_min, const double* l_max, int n,
                                          double l0, double dl0, double ddl0, const PathQpParams& prm, int gl, bool live) {
    path_qp_forms(Q, prm);
    if (!live) n = 0;                       // a group without a problem runs every loop zero times
    if (live && n < 4) return 2;
    const int N = n - 4;
    const double ds = prm.ds, ids2 = 1.0 / (ds * ds);
    const double hw = fabs(prm.host_w) / 2.0;
    const int fwd = (int)ceil(prm.d1 / ds), back = (int)ceil(prm.d2 / ds);       // ref :126-127
    const double c0 = l0 - ds * ds * ddl0 / 6.0;
    const double cf[3] = {c0 + ds * ds * ddl0 / 2.0 - ds * dl0, c0, c0 + ds * ds * ddl0 / 2.0 + ds * dl0};   // c_{-1}, c_0, c_1
    for (int j = gl; j < (live ? n + 2 : 0); j += G) cc[j] = (j == 0) ? cf[0] : (j == 1) ? cf[1] : (j == 2) ? cf[2] : 0.0;
    const double tol = 1e-9;
    int bad = 0;
    for (int i = gl; i < n; i += G) {
        const int i1 = (i + fwd < n - 1) ? i + fwd : n - 1;                       // ref :130
        const int i2 = (i - back > 0) ? i - back : 0;                             // ref :131
        const double ub = l_max[i1] - hw, lb = l_min[i2] + hw;
        if (lb > ub + tol) bad = 1;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            double v = 0.0;                                                       // fixed part of the form
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int j = i + p;
                const double cj = (j == 0) ? cf[0] : (j == 1) ? cf[1] : (j == 2) ? cf[2] : 0.0;
                v += Q.g[f][p] * cj;
            }
            if (i == 0 || i == n - 1) {
                if (v > ub + tol || v < lb - tol) bad = 1;
            } else {
                Q.c[(i - 1) * 2 + f] = v;
                Q.lo[(i - 1) * 2 + f] = lb;
                Q.hi[(i - 1) * 2 + f] = ub;
            }
        }
    }
    const double wl2 = 2.0 * (prm.w_l + prm.w_centre), wd2 = 2.0 * prm.w_ddl, wj2 = 2.0 * prm.w_dddl;
    for (int m = gl; m < (live ? N : 0); m += G) {
        const int j = m + 3;
#pragma unroll
        for (int d = 0; d < 4; ++d) Q.P[m * 4 + d] = (m + d < N) ? path_hess_entry(j, d, n, wl2, wd2, wj2, ids2) : 0.0;
        double qm = 0.0;
        for (int i = max(0, j - 2); i <= min(n - 1, j); ++i)
            qm += (-2.0 * prm.w_centre * ((l_min[i] + l_max[i]) / 2.0)) * bsp_a(j - i);   // ref :201-205
        for (int jp = max(0, j - 3); jp <= 2; ++jp)                               // fixed start coefficients
            qm += path_hess_entry(jp, j - jp, n, wl2, wd2, wj2, ids2) * ((jp == 0) ? cf[0] : (jp == 1) ? cf[1] : cf[2]);
        Q.q[m] = qm;
        Q.u[m] = 0.0;
    }
    return group_any<G>(bad != 0) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Path QP on one GROUP of G lanes (G = 64: one scene per wavefront, G = 32: two scenes side by side).
// lds: this group's path_qp_words(n) doubles (G = 32: path_qp_words_pair()).  l_min / l_max / outputs may be LDS or global.  EVERY lane of the
// wavefront must call this (it contains barriers); a group with live == false only takes part in them.
// G = 32 requires n <= 34 (N, ns <= 32).  returns (per group) 0 ok, 1 infeasible, 2 failed.
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ inline int path_qp_group(double* lds, const double* l_min, const double* l_max, int n, double l0, double dl0,
                                    double ddl0, const PathQpParams& prm, double* out_l, double* out_dl,
                                    double* out_ddl, int* iters_out, bool live, int debug_stage = 0) {
    const int gl = (threadIdx.x & 63) & (G - 1);
    *iters_out = 0;
    PathRangeQp Q;
    double* cc = lds;
    const int nn = live ? n : 4;
    if constexpr (G == 32) Q.bind_fast(lds + 36, 32, 32, nn - 4 > 0 ? nn - 4 : 0, nn - 2 > 0 ? nn - 2 : 0);   // path_qp_words_pair()
    else Q.bind(lds + nn + 2, nn - 4 > 0 ? nn - 4 : 0, nn - 2 > 0 ? nn - 2 : 0);
    int rc = path_qp_setup_group<G>(Q, cc, l_min, l_max, n, l0, dl0, ddl0, prm, gl, live);
    if (!live) rc = 2;
    if (debug_stage == 2) rc = 2;
    __syncthreads();
    bool ok = rc == 0;
    const bool small = Q.N <= G && Q.ns <= G;          // G = 32: guaranteed by the launcher
    // ---- start from the unconstrained minimiser P u = -q
    if (small || G == 32) {
        double fa[4], flow[4], frinv = 0.0;
#pragma unroll
        for (int d = 0; d < 4; ++d) fa[d] = (ok && gl < Q.N) ? Q.P[gl * 4 + d] : 0.0;
        const bool okc = band_chol_group<G, 3>(fa, frinv, flow, Q.N, gl, ok && Q.N > 0);
        double b0 = (ok && gl < Q.N) ? -Q.q[gl] : 0.0;
        EMP_QP_DEBUG("init: N %d P0 %.4e %.4e %.4e %.4e q0 %.4e | U0 %.4e %.4e rinv %.4e okc %d\n", Q.N, Q.P[0], Q.P[1], Q.P[2],
                     Q.P[3], Q.q[0], fa[0], fa[1], frinv, (int)okc);
        band_solve_group<G, 3>(fa, frinv, flow, b0, Q.N, gl);
        EMP_QP_DEBUG("init: u0 %.6e\n", b0);
        if (ok && gl < Q.N) Q.u[gl] = b0;
        if (ok && !okc) rc = 2;
    } else {
        for (int m = gl; m < (ok ? Q.N * 4 : 0); m += G) Q.M[m] = Q.P[m];
        for (int m = gl; m < (ok ? Q.N : 0); m += G) Q.u[m] = -Q.q[m];
        __syncthreads();
        int okc = 1;
        if (ok && gl == 0 && Q.N > 0) {
            okc = band_chol<3>(Q.M, Q.N) ? 1 : 0;
            if (okc) band_solve<3>(Q.M, Q.u, Q.N);
        }
        okc = __shfl(okc, 0, 64);
        if (ok && !okc) rc = 2;
    }
    if (debug_stage == 3) rc = 2;
    __syncthreads();
    ok = rc == 0;
    const int cap_it = debug_stage >= 10 ? debug_stage - 10 : 1000;
    int rs;
    if (small || G == 32) rs = range_qp_solve_wave_fast<G, G == 32, true>(Q, gl, ok && Q.N > 0, cap_it);
    else rs = range_qp_solve_wave<G>(Q, gl, ok && Q.N > 0, cap_it);
    if (ok && Q.N > 0) {
        *iters_out = Q.iters;
        if (rs && (debug_stage < 10 || !EMP_DEV_HOOKS)) rc = rs;   // (-DEMP_DEV_HOOKS only: a capped solve hands its last iterate on, for timing)
    }
    ok = rc == 0;
    if (ok && Q.N == 0) {                                // nothing free: only check the constant forms
        bool bad = false;
        for (int it = gl; it < Q.ns * 2; it += G)
            if (Q.c[it] > Q.hi[it] + 1e-9 || Q.c[it] < Q.lo[it] - 1e-9) bad = true;
        if (group_any<G>(bad)) rc = 1;
    }
    ok = rc == 0;
    for (int m = gl; m < (ok ? Q.N : 0); m += G) cc[m + 3] = Q.u[m];
    __syncthreads();
    const double ds = prm.ds;
    for (int i = gl; i < (ok ? n : 0); i += G) {
        out_l[i] = (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0;
        if (out_dl) out_dl[i] = (cc[i + 2] - cc[i]) / (2.0 * ds);
        if (out_ddl) out_ddl[i] = (cc[i] - 2.0 * cc[i + 1] + cc[i + 2]) / (ds * ds);
    }
    __syncthreads();
    return rc;
}

// Box-QP set-up, one point per lane of the 32-lane group (same mathematics as box_qp_setup in emp_qp_core.h)
__device__ inline int box_qp_setup_group(BoxRangeQp& Q, const double* ref, int stride, int m, const SmoothQpParams& prm,
                                         int gl) {
    box_qp_forms(Q);
    if (m < 2 || !(prm.thr > 0.0)) return 2;
    for (int i = gl; i < m; i += 32) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double e = 0.0;
            if (i + d < m) {
                for (int r = max(0, i + d - 2); r <= min(m - 3, i); ++r) {       // second-difference rows (ws)
                    const double a = (i - r == 1) ? -2.0 : 1.0, b = (i + d - r == 1) ? -2.0 : 1.0;
                    e += 2.0 * prm.w_smooth * a * b;
                }
                for (int r = max(0, i + d - 1); r <= min(m - 2, i); ++r) {       // first-difference rows (wl)
                    const double a = (i - r == 0) ? 1.0 : -1.0, b = (i + d - r == 0) ? 1.0 : -1.0;
                    e += 2.0 * prm.w_length * a * b;
                }
                if (d == 0) e += 2.0 * prm.w_ref;
            }
            Q.P[i * 3 + d] = e;
        }
        const double r = ref[i * stride];
        Q.q[i] = -2.0 * prm.w_ref * r;                                             // ref :346
        Q.c[i] = 0.0;
        Q.lo[i] = r - prm.thr;                                                     // ref :308-311
        Q.hi[i] = r + prm.thr;
        Q.u[i] = r;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// One smoothing (box) QP of m <= G points per GROUP of G lanes, one point per lane, registers only.  G = 32: the x
// problem of a polyline on lanes 0-31 and its y problem on lanes 32-63.  The box QP has G = I (W = 1, F = 1,
// off0 = 0): a station IS its unknown, so nothing is exchanged through LDS - neighbours for P u and for the
// Cholesky / substitution sweeps come from lane shifts (a shift that crosses into the other group meets a band
// entry that is exactly 0), norms and ratio tests from the group reductions.  Same algorithm, stopping rule and
// constants as range_qp_solve_wave_fast with BoxRangeQp (emp_qp_core.h: box_qp_forms / box_qp_setup).
// r: this lane's reference coordinate (lanes >= m of the group: anything).  Returns 0 ok / 2 failed per group.
// ---------------------------------------------------------------------------------------------
// The same box QP by a primal-dual active set iteration (Hintermueller / Ito / Kunisch): fix the coordinates whose
// bound is active, solve the banded system of the others, re-classify from the gradient, until the classification
// repeats.  A fixed point is the KKT point, i.e. THE minimiser, exact to rounding - and it is reached in two to
// five factorisations where the interior-point iteration needs nine to fourteen with two solves each.  There is no
// convergence guarantee for this Hessian (D2'D2 has positive off-diagonals: not an M-matrix), so the caller falls
// back to box_qp_lanes when this returns -1 (classification still changing after kBoxAsMaxIter rounds).
// Same lane layout as box_qp_lanes: one coordinate per lane, r = reference value, box r +- thr.
constexpr int kBoxAsMaxIter = 16;
template <int G>
__device__ inline int box_qp_active_set_lanes(double r, int m, const SmoothQpParams& prm, double* out_u, int* iters_out) {
    constexpr int KD = 2;
    const int gl = (threadIdx.x & 63) & (G - 1);
    const bool has = gl < m;
    *iters_out = 0;
    *out_u = r;
    if (m < 2 || !(prm.thr > 0.0)) return 2;
    double Prow[KD + 1], Plow[KD + 1];
#pragma unroll
    for (int d = 0; d <= KD; ++d) {                      // ref planning_utils.py:262-361 cost matrices, as box_qp_lanes
        double e = 0.0;
        if (has && gl + d < m) {
            for (int rr = max(0, gl + d - 2); rr <= min(m - 3, gl); ++rr) {
                const double a = (gl - rr == 1) ? -2.0 : 1.0, b = (gl + d - rr == 1) ? -2.0 : 1.0;
                e += 2.0 * prm.w_smooth * a * b;
            }
            for (int rr = max(0, gl + d - 1); rr <= min(m - 2, gl); ++rr) {
                const double a = (gl - rr == 0) ? 1.0 : -1.0, b = (gl + d - rr == 0) ? 1.0 : -1.0;
                e += 2.0 * prm.w_length * a * b;
            }
            if (d == 0) e += 2.0 * prm.w_ref;
        }
        Prow[d] = e;
    }
    {
        const double t1 = lane_up1(Prow[1]), t2 = lane_up1(lane_up1(Prow[2]));
        Plow[0] = 0.0;
        Plow[1] = (gl >= 1) ? t1 : 0.0;                  // P[gl-1][gl]
        Plow[2] = (gl >= 2) ? t2 : 0.0;                  // P[gl-2][gl]
    }
    const double q = has ? -2.0 * prm.w_ref * r : 0.0;   // ref :346
    const double lo = r - prm.thr, hi = r + prm.thr;     // ref :308-311
    const double c = 2.0 * (6.0 * prm.w_smooth + 2.0 * prm.w_length + prm.w_ref);   // the Hessian's interior diagonal
    int code = 0;                                        // 0 free, 1 fixed at hi, 2 fixed at lo
    double x = r;
    int state = 1, iters = 0;                            // 1 running, 0 converged, -1 gave up
    while (__any(state == 1)) {
        const bool go = state == 1;
        const bool act = go && has && code != 0;
        const double bnd = (code == 1) ? hi : lo;
        const double af = act ? 1.0 : 0.0, ab = act ? bnd : 0.0;
        // neighbours' flags and fixed values (0 beyond the ends of the group: Prow / Plow are 0 there as well)
        const double f_p1 = lane_dn1(af), f_p2 = lane_dn1(f_p1);      // (the band is stored by its upper half)
        const double b_p1 = lane_dn1(ab), b_p2 = lane_dn1(b_p1), b_m1 = lane_up1(ab), b_m2 = lane_up1(b_m1);
        double fa[KD + 1], flow[KD + 1], frinv = 1.0;
        fa[0] = act ? 1.0 : ((go && has) ? Prow[0] : 0.0);
        fa[1] = (go && !act && f_p1 == 0.0) ? Prow[1] : 0.0;
        fa[2] = (go && !act && f_p2 == 0.0) ? Prow[2] : 0.0;
        double rhs = -q - (((Prow[1] * b_p1 + Prow[2] * b_p2) + Plow[1] * b_m1) + Plow[2] * b_m2);
        rhs = act ? bnd : ((go && has) ? rhs : 0.0);
        const bool okf = band_chol_group<G, KD>(fa, frinv, flow, m, gl, go);
        band_solve_group<G, KD>(fa, frinv, flow, rhs, m, gl);
        if (go) {
            if (has) x = rhs;
            ++iters;
        }
        // gradient of the full problem at x and the new classification
        const double xs = (go && has) ? x : 0.0;
        const double x_p1 = lane_dn1(xs), x_p2 = lane_dn1(x_p1), x_m1 = lane_up1(xs), x_m2 = lane_up1(x_m1);
        const double g = (q + Prow[0] * xs) + (((Prow[1] * x_p1 + Prow[2] * x_p2) + Plow[1] * x_m1) + Plow[2] * x_m2);
        const double lam = -g;
        int ncode = 0;
        if (lam + c * (xs - hi) > 0.0) ncode = 1;
        else if (lam + c * (xs - lo) < 0.0) ncode = 2;
        if (!(go && has)) ncode = code;
        const bool changed = group_any<G>(ncode != code);
        if (go) {
            code = ncode;
            if (!okf) state = -1;
            else if (!changed) state = 0;
            else if (iters >= kBoxAsMaxIter) state = -1;
        }
    }
    *iters_out = iters;
    if (state == 0) *out_u = x;
    return state;
}

template <int G>
__device__ inline int box_qp_lanes(double r, int m, const SmoothQpParams& prm, double* out_u, int* iters_out) {
    constexpr int KD = 2;
    const int gl = (threadIdx.x & 63) & (G - 1);
    const bool has = gl < m;
    *iters_out = 0;
    *out_u = r;
    if (m < 2 || !(prm.thr > 0.0)) return 2;
    const double eps_p = 1e-10, eps_mu = 1e-13, eps_d_rel = 1e-10;   // box_qp_forms
    double Prow[KD + 1], Plow[KD + 1];
#pragma unroll
    for (int d = 0; d <= KD; ++d) {                      // ref planning_utils.py:262-361 cost matrices, as box_qp_setup_group
        double e = 0.0;
        if (has && gl + d < m) {
            for (int rr = max(0, gl + d - 2); rr <= min(m - 3, gl); ++rr) {
                const double a = (gl - rr == 1) ? -2.0 : 1.0, b = (gl + d - rr == 1) ? -2.0 : 1.0;
                e += 2.0 * prm.w_smooth * a * b;
            }
            for (int rr = max(0, gl + d - 1); rr <= min(m - 2, gl); ++rr) {
                const double a = (gl - rr == 0) ? 1.0 : -1.0, b = (gl + d - rr == 0) ? 1.0 : -1.0;
                e += 2.0 * prm.w_length * a * b;
            }
            if (d == 0) e += 2.0 * prm.w_ref;
        }
        Prow[d] = e;
    }
    {
        const double t1 = lane_up1(Prow[1]), t2 = lane_up1(lane_up1(Prow[2]));
        Plow[0] = 0.0;
        Plow[1] = (gl >= 1) ? t1 : 0.0;                  // P[gl-1][gl]
        Plow[2] = (gl >= 2) ? t2 : 0.0;                  // P[gl-2][gl]
    }
    const double q = has ? -2.0 * prm.w_ref * r : 0.0;   // ref :346
    const double lo = has ? r - prm.thr : -1e300;        // ref :308-311
    const double hi = has ? r + prm.thr : 1e300;
    double u = has ? r : 0.0, zu = 1.0, zl = 1.0;
    double su = hi - u, sl = u - lo;
    {
        const double smin = group_min<G>(has ? fmin(su, sl) : 1e300);
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;
        su += shift;
        sl += shift;
    }
    const double qscale = fmax(group_max<G>(has ? fabs(q) : 0.0), 1.0);
    const int rows = m * 2;
    int state = 1, iters = 0;
    bool acceptable = false;
    while (__any(state == 1)) {
        const double rpu = u - hi + su, rpl = lo - u + sl;
        const double isu = fast_rcp(su), isl = fast_rcp(sl), izu = fast_rcp(zu), izl = fast_rcp(zl);
        const double wu = zu * isu, wl = zl * isl;
        const double u1 = lane_dn1(u), u2 = lane_dn1(u1), d1 = lane_up1(u), d2 = lane_up1(d1);
        double acc = q + Prow[0] * u;
        acc += Prow[1] * u1 + Prow[2] * u2 + Plow[1] * d1 + Plow[2] * d2;
        const double rd = has ? acc + (zu - zl) : 0.0;
        const double rd_max = group_max<G>(fabs(rd));
        const double rp_max = group_max<G>(has ? fmax(fabs(rpu), fabs(rpl)) : 0.0);
        const double zmax = group_max<G>(has ? fmax(zu, zl) : 0.0);
        const double mu = group_sum<G>(has ? su * zu + sl * zl : 0.0) / (double)rows;
        if (state == 1) {
            const double dscale = fmax(qscale, zmax);
            if (rd_max <= eps_d_rel * dscale && rp_max <= eps_p && mu <= eps_mu) state = 0;
            else if (!(mu == mu) || mu > 1e30 || (iters >= kQpStallIter && rp_max > kQpStallResidual)) state = 2;
            else if (iters >= kQpMaxIter) state = acceptable ? 0 : 2;
            if (rd_max <= 100.0 * eps_d_rel * dscale && rp_max <= 10.0 * eps_p && mu <= 1000.0 * eps_mu) acceptable = true;
        }
        if (!__any(state == 1)) break;          // nobody left to iterate: the rest of the pass would run fully masked (emp_qp_rows.h; one wavefront per block)
        const bool go = state == 1;
        double fa[KD + 1], flow[KD + 1], frinv = 1.0;
        fa[0] = (go && has) ? Prow[0] + (wu + wl) : 0.0;
        fa[1] = go ? Prow[1] : 0.0;
        fa[2] = go ? Prow[2] : 0.0;
        const bool okf = band_chol_group<G, KD>(fa, frinv, flow, m, gl, go);
        if (go && !okf) state = acceptable ? 0 : 2;
        const bool go2 = state == 1;
        double dua = (go2 && has) ? -rd - ((wu * rpu - zu) - (wl * rpl - zl)) : 0.0;
        band_solve_group<G, KD>(fa, frinv, flow, dua, m, gl);
        const double dsua = -rpu - dua, dsla = -rpl + dua;
        const double dzua = -zu - wu * dsua, dzla = -zl - wl * dsla;
        double ratio = (go2 && has) ? fmax(fmax(-dsua * isu, -dsla * isl), fmax(-dzua * izu, -dzla * izl)) : 0.0;
        ratio = group_max<G>(ratio);
        const double a_aff = (ratio > 1.0) ? fast_rcp(ratio) : 1.0;
        double mu_aff = (go2 && has) ? (su + a_aff * dsua) * (zu + a_aff * dzua) + (sl + a_aff * dsla) * (zl + a_aff * dzla) : 0.0;
        mu_aff = group_sum<G>(mu_aff) / (double)rows;
        double sigma = (mu > 0.0) ? mu_aff * fast_rcp(mu) : 0.0;
        sigma = sigma * sigma * sigma;
        const double rcu = su * zu + dsua * dzua - sigma * mu, rcl = sl * zl + dsla * dzla - sigma * mu;
        double du = (go2 && has) ? -rd - ((zu * rpu - rcu) * isu - (zl * rpl - rcl) * isl) : 0.0;
        band_solve_group<G, KD>(fa, frinv, flow, du, m, gl);
        const double dsu = -rpu - du, dsl = -rpl + du;
        const double dzu = -(rcu + zu * dsu) * isu, dzl = -(rcl + zl * dsl) * isl;
        ratio = (go2 && has) ? fmax(fmax(-dsu * isu, -dsl * isl), fmax(-dzu * izu, -dzl * izl)) : 0.0;
        ratio = group_max<G>(ratio);
        const double tau = qp_step_fraction(mu);
        const double alpha = (ratio > tau) ? tau * fast_rcp(ratio) : 1.0;   // min(1, tau / ratio)
        if (go2) {
            if (has) {
                su += alpha * dsu;
                sl += alpha * dsl;
                zu += alpha * dzu;
                zl += alpha * dzl;
                u += alpha * du;
            }
            ++iters;
        }
    }
    *iters_out = iters;
    *out_u = u;
    return state;
}

// ---------------------------------------------------------------------------------------------
// Smoothing QPs of one polyline of m <= 64 points on one wavefront, one POINT per lane; the x and the y
// problem (same structure, independent data) run side by side in the same lanes so that their two serial
// chains interleave.  The box QP has G = I (W = 1, F = 1, off0 = 0): a station IS its unknown, so the whole
// interior-point iteration lives in registers - neighbours for P u and for the Cholesky / substitution sweeps
// come from lane shifts, norms and ratio tests from butterflies.  Same algorithm, stopping rule and constants
// as range_qp_solve_wave_fast with BoxRangeQp (emp_qp_core.h: box_qp_forms / box_qp_setup).
// rx, ry: this lane's reference point (lanes >= m: anything).  Returns 0 ok / 2 failed (wave-uniform).
// ---------------------------------------------------------------------------------------------
__device__ inline int smooth_pair_lanes(double rx, double ry, int m, const SmoothQpParams& sx, const SmoothQpParams& sy,
                                        double* out_x, double* out_y, int* iters_out) {
    constexpr int NP = 2, KD = 2;
    const int gl = threadIdx.x & 63;
    const bool has = gl < m;
    *iters_out = 0;
    if (m < 2 || !(sx.thr > 0.0) || !(sy.thr > 0.0)) return 2;
    const SmoothQpParams prm[NP] = {sx, sy};
    const double eps_p = 1e-10, eps_mu = 1e-13, eps_d_rel = 1e-10;   // box_qp_forms
    double Prow[NP][KD + 1], Plow[NP][KD + 1], q[NP], lo[NP], hi[NP], u[NP], su[NP], sl[NP], zu[NP], zl[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const double r = p == 0 ? rx : ry;
#pragma unroll
        for (int d = 0; d <= KD; ++d) {                  // ref planning_utils.py:262-361 cost matrices, as box_qp_setup_group
            double e = 0.0;
            if (has && gl + d < m) {
                for (int rr = max(0, gl + d - 2); rr <= min(m - 3, gl); ++rr) {
                    const double a = (gl - rr == 1) ? -2.0 : 1.0, b = (gl + d - rr == 1) ? -2.0 : 1.0;
                    e += 2.0 * prm[p].w_smooth * a * b;
                }
                for (int rr = max(0, gl + d - 1); rr <= min(m - 2, gl); ++rr) {
                    const double a = (gl - rr == 0) ? 1.0 : -1.0, b = (gl + d - rr == 0) ? 1.0 : -1.0;
                    e += 2.0 * prm[p].w_length * a * b;
                }
                if (d == 0) e += 2.0 * prm[p].w_ref;
            }
            Prow[p][d] = e;
        }
        Plow[p][0] = 0.0;
        Plow[p][1] = lane_up1(Prow[p][1]);               // P[gl-1][gl]; 0 for gl = 0 (lane 0 has no source)
        Plow[p][2] = lane_up1(lane_up1(Prow[p][2]));
        q[p] = has ? -2.0 * prm[p].w_ref * r : 0.0;      // ref :346
        lo[p] = has ? r - prm[p].thr : -1e300;           // ref :308-311
        hi[p] = has ? r + prm[p].thr : 1e300;
        u[p] = has ? r : 0.0;
        zu[p] = zl[p] = 1.0;
    }
    const int rows = m * 2;
    int state[NP], iters[NP];
    bool acceptable[NP];
    double qscale[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        state[p] = 1;
        iters[p] = 0;
        acceptable[p] = false;
        qscale[p] = fmax(group_max<64>(has ? fabs(q[p]) : 0.0), 1.0);
        su[p] = hi[p] - u[p];                            // = thr: no shift needed when thr >= 1, else push to >= 1
        sl[p] = u[p] - lo[p];
        double smin = has ? fmin(su[p], sl[p]) : 1e300;
        smin = group_min<64>(smin);
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;
        su[p] += shift;
        sl[p] += shift;
    }
    while (state[0] == 1 || state[1] == 1) {
        double rpu[NP], rpl[NP], isu[NP], isl[NP], izu[NP], izl[NP], wu[NP], wl[NP], rd[NP], mu[NP];
        double fa[NP][KD + 1], flow[NP][KD + 1], frinv[NP];
        double red[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            rpu[p] = u[p] - hi[p] + su[p];
            rpl[p] = lo[p] - u[p] + sl[p];
            isu[p] = fast_rcp(su[p]);
            isl[p] = fast_rcp(sl[p]);
            izu[p] = fast_rcp(zu[p]);
            izl[p] = fast_rcp(zl[p]);
            wu[p] = zu[p] * isu[p];
            wl[p] = zl[p] * isl[p];
            const double u1 = lane_dn1(u[p]), u2 = lane_dn1(u1), d1 = lane_up1(u[p]), d2 = lane_up1(d1);
            double acc = q[p] + Prow[p][0] * u[p];
            acc += Prow[p][1] * u1 + Prow[p][2] * u2 + Plow[p][1] * d1 + Plow[p][2] * d2;
            rd[p] = has ? acc + (zu[p] - zl[p]) : 0.0;
            fa[p][0] = has ? Prow[p][0] + (wu[p] + wl[p]) : 0.0;
            fa[p][1] = Prow[p][1];
            fa[p][2] = Prow[p][2];
            red[p][0] = fabs(rd[p]);
            red[p][1] = has ? fmax(fabs(rpu[p]), fabs(rpl[p])) : 0.0;
            red[p][2] = has ? fmax(zu[p], zl[p]) : 0.0;
            red[p][3] = has ? su[p] * zu[p] + sl[p] * zl[p] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            red[p][0] = group_max<64>(red[p][0]);
            red[p][1] = group_max<64>(red[p][1]);
            red[p][2] = group_max<64>(red[p][2]);
            red[p][3] = group_sum<64>(red[p][3]);
        }
        bool go[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            mu[p] = red[p][3] / (double)rows;
            if (state[p] == 1) {
                const double rd_max = red[p][0], rp_max = red[p][1], dscale = fmax(qscale[p], red[p][2]);
                if (rd_max <= eps_d_rel * dscale && rp_max <= eps_p && mu[p] <= eps_mu) state[p] = 0;
                else if (!(mu[p] == mu[p]) || mu[p] > 1e30 || (iters[p] >= kQpStallIter && rp_max > kQpStallResidual)) state[p] = 2;
                else if (iters[p] >= kQpMaxIter) state[p] = acceptable[p] ? 0 : 2;
                if (rd_max <= 100.0 * eps_d_rel * dscale && rp_max <= 10.0 * eps_p && mu[p] <= 1000.0 * eps_mu)
                    acceptable[p] = true;
            }
            go[p] = state[p] == 1;
        }
        if (!go[0] && !go[1]) break;
        const bool okf = band_chol_group_n<64, KD, NP>(fa, frinv, flow, m, gl, go);
        (void)okf;
        double pivot_bad[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) pivot_bad[p] = (go[p] && has && !(frinv[p] == frinv[p] && frinv[p] > 0.0 && frinv[p] < 1e300)) ? 1.0 : 0.0;
        const double any_bad0 = group_max<64>(pivot_bad[0]), any_bad1 = group_max<64>(pivot_bad[1]);
        if (go[0] && any_bad0 != 0.0) state[0] = acceptable[0] ? 0 : 2;
        if (go[1] && any_bad1 != 0.0) state[1] = acceptable[1] ? 0 : 2;
        bool go2[NP];
        double dua[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            go2[p] = state[p] == 1;
            dua[p] = (go2[p] && has) ? -rd[p] - ((wu[p] * rpu[p] - zu[p]) - (wl[p] * rpl[p] - zl[p])) : 0.0;
        }
        band_solve_group_n<64, KD, NP>(fa, frinv, flow, dua, m, gl);
        double dsua[NP], dsla[NP], dzua[NP], dzla[NP], ratio[NP], mu_aff[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            dsua[p] = -rpu[p] - dua[p];
            dsla[p] = -rpl[p] + dua[p];
            dzua[p] = -zu[p] - wu[p] * dsua[p];
            dzla[p] = -zl[p] - wl[p] * dsla[p];
            ratio[p] = (go2[p] && has) ? fmax(fmax(-dsua[p] * isu[p], -dsla[p] * isl[p]), fmax(-dzua[p] * izu[p], -dzla[p] * izl[p])) : 0.0;
        }
        ratio[0] = group_max<64>(ratio[0]);
        ratio[1] = group_max<64>(ratio[1]);
        double a_aff[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            a_aff[p] = (ratio[p] > 1.0) ? fast_rcp(ratio[p]) : 1.0;
            mu_aff[p] = (go2[p] && has) ? (su[p] + a_aff[p] * dsua[p]) * (zu[p] + a_aff[p] * dzua[p]) +
                                              (sl[p] + a_aff[p] * dsla[p]) * (zl[p] + a_aff[p] * dzla[p])
                                        : 0.0;
        }
        mu_aff[0] = group_sum<64>(mu_aff[0]);
        mu_aff[1] = group_sum<64>(mu_aff[1]);
        double rcu[NP], rcl[NP], du[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            mu_aff[p] /= (double)rows;
            double sigma = (mu[p] > 0.0) ? mu_aff[p] * fast_rcp(mu[p]) : 0.0;
            sigma = sigma * sigma * sigma;
            rcu[p] = su[p] * zu[p] + dsua[p] * dzua[p] - sigma * mu[p];
            rcl[p] = sl[p] * zl[p] + dsla[p] * dzla[p] - sigma * mu[p];
            du[p] = (go2[p] && has) ? -rd[p] - ((zu[p] * rpu[p] - rcu[p]) * isu[p] - (zl[p] * rpl[p] - rcl[p]) * isl[p]) : 0.0;
        }
        band_solve_group_n<64, KD, NP>(fa, frinv, flow, du, m, gl);
        double dsu[NP], dsl[NP], dzu[NP], dzl[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            dsu[p] = -rpu[p] - du[p];
            dsl[p] = -rpl[p] + du[p];
            dzu[p] = -(rcu[p] + zu[p] * dsu[p]) * isu[p];
            dzl[p] = -(rcl[p] + zl[p] * dsl[p]) * isl[p];
            ratio[p] = (go2[p] && has) ? fmax(fmax(-dsu[p] * isu[p], -dsl[p] * isl[p]), fmax(-dzu[p] * izu[p], -dzl[p] * izl[p])) : 0.0;
        }
        ratio[0] = group_max<64>(ratio[0]);
        ratio[1] = group_max<64>(ratio[1]);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double tau = qp_step_fraction(mu[p]);
            const double alpha = (ratio[p] > tau) ? tau * fast_rcp(ratio[p]) : 1.0;   // min(1, tau / ratio)
            if (go2[p]) {
                if (has) {
                    su[p] += alpha * dsu[p];
                    sl[p] += alpha * dsl[p];
                    zu[p] += alpha * dzu[p];
                    zl[p] += alpha * dzl[p];
                    u[p] += alpha * du[p];
                }
                ++iters[p];
            }
        }
    }
    *iters_out = max(iters[0], iters[1]);
    *out_x = u[0];
    *out_y = u[1];
    return (state[0] == 0 && state[1] == 0) ? 0 : 2;
}

// Smoothing of one polyline on one wavefront.  m <= 32: lanes 0-31 solve x, lanes 32-63 solve y (register fast
// path); 32 < m <= 64: smooth_pair_lanes (one point per lane, x and y side by side, registers only); longer
// polylines: the two half-waves with the LDS-resident solver.
// lds: 2 * BoxRangeQp::words(m, m) doubles.  xy: [m][stride] with x at +0, y at +1 (LDS or global).
// On success the smoothed coordinates are Q.u of each half: returned through out_x / out_y pointers INTO lds.
// WIDE == false: the caller guarantees m <= 32 and only the half-wave register path is compiled (fewer VGPRs).
template <bool WIDE>
__device__ inline int smooth_pair_wave(double* lds, const double* xy, int stride, int m, const SmoothQpParams& sx,
                                       const SmoothQpParams& sy, double** out_x, double** out_y, int* iters_out) {
    const int lane = threadIdx.x & 63, grp = lane >> 5, gl = lane & 31;
    *iters_out = 0;
    if (m < 2) return 2;
    if constexpr (!WIDE) {
        if (m > 32) return 2;
    }
#ifndef EMP_SMOOTH_FORCE_LDS
    if (WIDE && m > 32 && m <= 64) {                    // one point per lane, x and y side by side, all in registers
        double ux = 0.0, uy = 0.0;
        const double rx = lane < m ? xy[(size_t)lane * stride] : 0.0, ry = lane < m ? xy[(size_t)lane * stride + 1] : 0.0;
        // active-set iteration for x, then for y; the paired interior-point solver if either did not settle
        int itx = 0, ity = 0;
        int rc = box_qp_active_set_lanes<64>(rx, m, sx, &ux, &itx);
        if (rc == 0) rc = box_qp_active_set_lanes<64>(ry, m, sy, &uy, &ity);
        *iters_out = max(itx, ity);
        if (rc < 0) {
            int it2 = 0;
            rc = smooth_pair_lanes(rx, ry, m, sx, sy, &ux, &uy, &it2);
            *iters_out += it2;
        }
        __syncthreads();
        if (lane < m) {
            lds[lane] = ux;
            lds[m + lane] = uy;
        }
        __syncthreads();
        *out_x = lds;
        *out_y = lds + m;
        return rc;
    }
#endif
#ifndef EMP_SMOOTH_FORCE_LDS
    if (m <= 32) {                                      // x on lanes 0-31, y on lanes 32-63, registers only
        const double r = gl < m ? xy[(size_t)gl * stride + grp] : 0.0;
        double uu = 0.0;
        int it_mine = 0;
        // active-set iteration first; the interior-point solver takes over for a coordinate whose classification
        // did not settle (every lane of the wavefront walks through it then, the settled half keeps its result)
        int rc = box_qp_active_set_lanes<32>(r, m, grp ? sy : sx, &uu, &it_mine);
        if (__any(rc < 0)) {
            double u2 = 0.0;
            int it2 = 0;
            const int rc2 = box_qp_lanes<32>(r, m, grp ? sy : sx, &u2, &it2);
            if (rc < 0) {
                rc = rc2;
                uu = u2;
                it_mine += it2;
            }
        }
        __syncthreads();
        if (gl < m) lds[grp * m + gl] = uu;
        __syncthreads();
        *iters_out = max(__shfl(it_mine, 0, 64), __shfl(it_mine, 32, 64));
        *out_x = lds;
        *out_y = lds + m;
        return __any(rc != 0) ? 2 : 0;
    }
#endif
    BoxRangeQp Q;
    const int words = BoxRangeQp::words(m, m);
    Q.bind(lds + grp * words, m, m);
    int rc = box_qp_setup_group(Q, xy + grp, stride, m, grp ? sy : sx, gl);
    __syncthreads();
    const int bad_setup = __any(rc != 0);
    if (bad_setup) return 2;
    if (m <= 32) rc = range_qp_solve_wave_fast<32>(Q, gl, true, 1000);
    else if constexpr (WIDE) rc = range_qp_solve_wave<32>(Q, gl, true);
    const int it_mine = Q.iters;
    *iters_out = max(__shfl(it_mine, 0, 64), __shfl(it_mine, 32, 64));
    BoxRangeQp Q0, Q1;
    Q0.bind(lds, m, m);
    Q1.bind(lds + words, m, m);
    *out_x = Q0.u;
    *out_y = Q1.u;
    return __any(rc != 0) ? 2 : 0;
}

#pragma clang fp contract(off)

}  // namespace emp
