// emp_qp_wave.h - wave-cooperative solver for the banded range QP of emp_qp_core.h (device only).
//
// One problem per GROUP of G lanes (G = 64: one problem per wavefront; G = 32: two problems side by side,
// e.g. the x and y smoothing problems of one scene).  All problem arrays live in LDS.  The O(n) parts of an
// interior-point iteration (residuals, normal-matrix assembly, ratio tests, updates) run one station /
// unknown per lane with butterfly reductions; only the banded Cholesky and the two triangular solves are a
// serial recurrence, run by lane 0 of the group out of LDS.  The block must consist of exactly one
// wavefront so that __syncthreads() is a cheap wave-level LDS fence.
//
// Same algorithm and stopping rule as RangeQp::solve_scalar (emp_qp_core.h); different summation order, so
// results agree to round-off, not bitwise (QP outputs are compared at 1e-6).
#pragma once

#include <hip/hip_runtime.h>

#include "emp_qp_core.h"

namespace emp {

template <int G>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G>
__device__ __forceinline__ double group_min(double v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
__device__ int range_qp_solve_wave(RangeQp<KD, F, W>& Q, int gl, bool live) {
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
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;
        for (int it = gl; it < items; it += G) {
            Q.s[it * 2] += shift;
            Q.s[it * 2 + 1] += shift;
            Q.z[it * 2] = 1.0;
            Q.z[it * 2 + 1] = 1.0;
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
            else if (!(mu == mu) || mu > 1e30) state = 2;
            else if (iters >= kQpMaxIter) state = acceptable ? 0 : 2;
            if (rd_max <= 100.0 * Q.eps_d_rel * dscale && rp_max <= 10.0 * Q.eps_p && mu <= 1000.0 * Q.eps_mu)
                acceptable = true;
        }
        __syncthreads();
        const bool go = state == 1;
        // ---- C: factorisation (serial recurrence, lane 0 of the group)
        int ok = 1;
        if (go && gl == 0) ok = band_chol<KD>(Q.M, N) ? 1 : 0;
        ok = __shfl(ok, (threadIdx.x & 63) & ~(G - 1), 64);
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
        if (go2 && gl == 0) band_solve<KD>(Q.M, Q.dua, N);
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
        if (go2 && gl == 0) band_solve<KD>(Q.M, Q.rhs, N);      // rhs = du
        __syncthreads();
        // ---- K/L: step length, then update rows (s, z) and, after a barrier, the unknowns
        if (go2) {
            double a_loc = 1e300;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) alpha = fmin(1.0, ((mu < 1e-6) ? 0.999 : 0.99) * group_min<G>(a_loc));
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
// Path QP on one wavefront.  lds: path_qp_words(n) doubles.  l_min / l_max / outputs may be LDS or global.
// Every lane of the wavefront must call this (it contains barriers).  returns 0 ok, 1 infeasible, 2 failed.
// ---------------------------------------------------------------------------------------------
__device__ inline int path_qp_wave(double* lds, const double* l_min, const double* l_max, int n, double l0, double dl0,
                                   double ddl0, const PathQpParams& prm, double* out_l, double* out_dl,
                                   double* out_ddl, int* iters_out) {
    const int lane = threadIdx.x & 63;
    *iters_out = 0;
    if (n < 4) return 2;
    PathRangeQp Q;
    double* cc = lds;
    Q.bind(lds + n + 2, n - 4, n - 2);
    int rc = 0;
    if (lane == 0) rc = path_qp_setup(Q, cc, l_min, l_max, n, l0, dl0, ddl0, prm);
    rc = __shfl(rc, 0, 64);
    path_qp_forms(Q, prm);          // per-thread constants (lane 0's setup only filled its own copy)
    __syncthreads();
    if (rc) return rc;
    if (Q.N > 0) {
        // start from the unconstrained minimiser P u = -q
        for (int m = lane; m < Q.N * 4; m += 64) Q.M[m] = Q.P[m];
        for (int m = lane; m < Q.N; m += 64) Q.u[m] = -Q.q[m];
        __syncthreads();
        int ok = 1;
        if (lane == 0) {
            ok = band_chol<3>(Q.M, Q.N) ? 1 : 0;
            if (ok) band_solve<3>(Q.M, Q.u, Q.N);
        }
        ok = __shfl(ok, 0, 64);
        __syncthreads();
        if (!ok) return 2;
        rc = range_qp_solve_wave<64>(Q, lane, true);
        *iters_out = Q.iters;
        if (rc) return rc;
        for (int m = lane; m < Q.N; m += 64) cc[m + 3] = Q.u[m];
    } else {
        int bad = 0;
        for (int it = lane; it < Q.ns * 2; it += 64)
            if (Q.c[it] > Q.hi[it] + 1e-9 || Q.c[it] < Q.lo[it] - 1e-9) bad = 1;
        if (__any(bad)) return 1;
    }
    __syncthreads();
    const double ds = prm.ds;
    for (int i = lane; i < n; i += 64) {
        out_l[i] = (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0;
        if (out_dl) out_dl[i] = (cc[i + 2] - cc[i]) / (2.0 * ds);
        if (out_ddl) out_ddl[i] = (cc[i] - 2.0 * cc[i + 1] + cc[i + 2]) / (ds * ds);
    }
    __syncthreads();
    return 0;
}

// Smoothing of one polyline on one wavefront: lanes 0-31 solve x, lanes 32-63 solve y.
// lds: 2 * BoxRangeQp::words(m, m) doubles.  xy: [m][stride] with x at +0, y at +1 (LDS or global).
// On success the smoothed coordinates are Q.u of each half: returned through out_x / out_y pointers INTO lds.
__device__ inline int smooth_pair_wave(double* lds, const double* xy, int stride, int m, const SmoothQpParams& sx,
                                       const SmoothQpParams& sy, double** out_x, double** out_y, int* iters_out) {
    const int lane = threadIdx.x & 63, grp = lane >> 5, gl = lane & 31;
    *iters_out = 0;
    if (m < 2) return 2;
    BoxRangeQp Q;
    const int words = BoxRangeQp::words(m, m);
    Q.bind(lds + grp * words, m, m);
    int rc = 0;
    if (gl == 0) rc = box_qp_setup(Q, xy + grp, stride, m, grp ? sy : sx);
    rc = __shfl(rc, grp * 32, 64);
    box_qp_forms(Q);
    __syncthreads();
    const int bad_setup = __any(rc != 0);
    if (bad_setup) return 2;
    rc = range_qp_solve_wave<32>(Q, gl, true);
    const int it_mine = Q.iters;
    *iters_out = max(__shfl(it_mine, 0, 64), __shfl(it_mine, 32, 64));
    BoxRangeQp Q0, Q1;
    Q0.bind(lds, m, m);
    Q1.bind(lds + words, m, m);
    *out_x = Q0.u;
    *out_y = Q1.u;
    return __any(rc != 0) ? 2 : 0;
}

}  // namespace emp
