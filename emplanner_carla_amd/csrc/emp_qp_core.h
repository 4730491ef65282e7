// emp_qp_core.h - the two QPs of the EM-Planner path as banded interior-point solves.
//
// The reference builds DENSE matrices and hands them to cvxopt (ref: path_planning.py:103-214,
// planning_utils.py:300-353).  Both problems have a unique minimiser, so we are free to pose them in
// the coordinates where they are banded and positive definite:
//
//  * Path QP (ref Quadratic_planning).  The 2n-2 equality rows (path_planning.py:106-112) say exactly
//    that l(s) is a C2 cubic spline on uniform knots (piecewise-linear second derivative).  In the
//    uniform cubic B-spline basis c_{-1..n}
//        l_i = (c_{i-1} + 4 c_i + c_{i+1}) / 6,  dl_i = (c_{i+1} - c_{i-1}) / (2 ds),
//        ddl_i = (c_{i-1} - 2 c_i + c_{i+1}) / ds^2
//    the equalities vanish identically, the pinned start state (lb == ub rows, :148-153) fixes
//    c_{-1}, c_0, c_1, the pinned end state (:155-160) fixes c_{n-2} = c_{n-1} = c_n = 0, and the n-4
//    remaining coefficients carry a heptadiagonal SPD Hessian.  Of the 8 corner rows per station
//    (:117-142) four are parallel to and looser than the other four; the +-1e5 box rows (:145-147)
//    cannot be active.  What remains per station is  lb_i <= l_i +- d * dl_i <= ub_i.
//  * Smoothing QP (ref smooth_reference_line).  x and y decouple into two box-constrained problems
//    with the same constant pentadiagonal SPD Hessian 2 (ws D2'D2 + wl D1'D1 + wr I).
//
// Solver: Mehrotra predictor-corrector on the reduced normal equations (H + G' diag(z/s) G) du = r,
// banded Cholesky (half bandwidth 3 / 2), one scene per lane, all state in private arrays.
#pragma once

#include "emp_core.h"

#ifndef EMP_QP_TRACE
#define EMP_QP_TRACE(...)
#endif

namespace emp {

constexpr int kQpMaxIter = 60;

// ---------------------------------------------------------------------------------------------
// banded SPD Cholesky: A = U'U, upper band storage a[i*(KD+1)+d] = A[i][i+d]
// ---------------------------------------------------------------------------------------------
template <int KD>
EMP_HD bool band_chol(double* a, int n) {
    constexpr int W = KD + 1;
    for (int i = 0; i < n; ++i) {
        for (int d = 0; d <= KD && i + d < n; ++d) {
            double sum = a[i * W + d];
            const int j = i + d;
            const int k0 = (j - KD > 0) ? j - KD : 0;
            for (int k = k0; k < i; ++k) sum -= a[k * W + (i - k)] * a[k * W + (j - k)];
            if (d == 0) {
                if (!(sum > 0.0)) return false;
                a[i * W] = sqrt(sum);
            } else {
                a[i * W + d] = sum / a[i * W];
            }
        }
    }
    return true;
}

template <int KD>
EMP_HD void band_solve(const double* u, double* x, int n) {
    constexpr int W = KD + 1;
    for (int i = 0; i < n; ++i) {          // U' y = b
        double sum = x[i];
        const int k0 = (i - KD > 0) ? i - KD : 0;
        for (int k = k0; k < i; ++k) sum -= u[k * W + (i - k)] * x[k];
        x[i] = sum / u[i * W];
    }
    for (int i = n - 1; i >= 0; --i) {     // U x = y
        double sum = x[i];
        for (int d = 1; d <= KD && i + d < n; ++d) sum -= u[i * W + d] * x[i + d];
        x[i] = sum / u[i * W];
    }
}

struct PathQpParams {
    double ds, w_l, w_ddl, w_dddl, w_centre, d1, d2, host_w;
};

// ---------------------------------------------------------------------------------------------
// Path QP.  NMAX = maximum number of stations n.
// ---------------------------------------------------------------------------------------------
template <int NMAX>
struct PathQp {
    int n = 0, N = 0;                   // stations, free coefficients (n - 4)
    double cc[NMAX + 2];                // B-spline coefficients c_{-1..n} at index +1
    double lb[NMAX], ub[NMAX];          // per-station range of both corner forms
    double gp[3], gm[3];                // corner forms l + d1 dl, l - d2 dl on a coefficient window
    double P[NMAX * 4], q[NMAX];        // free-variable Hessian (upper band, constant) and linear term
    double M[NMAX * 4];                 // P + G'WG, then its Cholesky factor
    double s[4 * NMAX], z[4 * NMAX];    // slacks / multipliers, row = 4*i + {0: gp upper, 1: gp lower, 2: gm upper, 3: gm lower}
    double rhs[NMAX], dua[NMAX];
    int iters = 0;

    EMP_HD double form(const double* g, int i) const { return g[0] * cc[i] + g[1] * cc[i + 1] + g[2] * cc[i + 2]; }
    // G-row value on a direction vector over the free coefficients (fixed ones contribute 0)
    EMP_HD double form_dir(const double* g, int i, const double* du) const {
        double v = 0.0;
        for (int p = 0; p < 3; ++p) {
            const int f = i + p - 3;
            if (f >= 0 && f < N) v += g[p] * du[f];
        }
        return v;
    }
    EMP_HD void scatter(const double* g, int i, double coef, double* vec) const {
        for (int p = 0; p < 3; ++p) {
            const int f = i + p - 3;
            if (f >= 0 && f < N) vec[f] += coef * g[p];
        }
    }
    // residual g.c - h + s of row (i, r)
    EMP_HD double row_resid(int i, int r) const {
        const double v = form(r < 2 ? gp : gm, i);
        return ((r & 1) == 0) ? (v - ub[i] + s[4 * i + r]) : (lb[i] - v + s[4 * i + r]);
    }
    EMP_HD double row_sign(int r) const { return ((r & 1) == 0) ? 1.0 : -1.0; }

    // returns 0 ok, 1 infeasible by inspection, 2 not converged / numerical failure
    EMP_HD int solve(const double* l_min, const double* l_max, int n_in, double l0, double dl0, double ddl0,
                     const PathQpParams& prm, double* out_l, double* out_dl, double* out_ddl) {
        n = n_in;
        iters = 0;
        if (n < 4 || n > NMAX) return 2;
        N = n - 4;
        const double ds = prm.ds;
        const double hw = fabs(prm.host_w) / 2.0;
        const int fwd = (int)ceil(prm.d1 / ds), back = (int)ceil(prm.d2 / ds);   // ref :126-127
        for (int i = 0; i < n; ++i) {
            const int i1 = (i + fwd < n - 1) ? i + fwd : n - 1;                   // ref :130
            const int i2 = (i - back > 0) ? i - back : 0;                         // ref :131
            ub[i] = l_max[i1] - hw;
            lb[i] = l_min[i2] + hw;
        }
        gp[0] = 1.0 / 6.0 - prm.d1 / (2.0 * ds);
        gp[1] = 4.0 / 6.0;
        gp[2] = 1.0 / 6.0 + prm.d1 / (2.0 * ds);
        gm[0] = 1.0 / 6.0 + prm.d2 / (2.0 * ds);
        gm[1] = 4.0 / 6.0;
        gm[2] = 1.0 / 6.0 - prm.d2 / (2.0 * ds);
        // pinned start / end states -> fixed coefficients
        for (int j = 0; j < n + 2; ++j) cc[j] = 0.0;
        const double c0 = l0 - ds * ds * ddl0 / 6.0;
        cc[1] = c0;
        cc[2] = c0 + ds * ds * ddl0 / 2.0 + ds * dl0;
        cc[0] = c0 + ds * ds * ddl0 / 2.0 - ds * dl0;
        // feasibility by inspection: empty ranges, fixed stations 0 and n-1
        const double tol = 1e-9;
        for (int i = 0; i < n; ++i)
            if (lb[i] > ub[i] + tol) return 1;
        {
            const double vs[4] = {form(gp, 0), form(gm, 0), form(gp, n - 1), form(gm, n - 1)};
            if (vs[0] > ub[0] + tol || vs[0] < lb[0] - tol || vs[1] > ub[0] + tol || vs[1] < lb[0] - tol) return 1;
            if (vs[2] > ub[n - 1] + tol || vs[2] < lb[n - 1] - tol || vs[3] > ub[n - 1] + tol ||
                vs[3] < lb[n - 1] - tol)
                return 1;
        }
        if (N > 0) {
            int rc = interior_point(l_min, l_max, prm);
            if (rc) return rc;
        } else {
            for (int i = 1; i <= n - 2; ++i) {
                const double a = form(gp, i), b = form(gm, i);
                if (a > ub[i] + tol || a < lb[i] - tol || b > ub[i] + tol || b < lb[i] - tol) return 1;
            }
        }
        for (int i = 0; i < n; ++i) {
            out_l[i] = (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0;
            out_dl[i] = (cc[i + 2] - cc[i]) / (2.0 * ds);
            out_ddl[i] = (cc[i] - 2.0 * cc[i + 1] + cc[i + 2]) / (ds * ds);
        }
        return 0;
    }

    EMP_HD void assemble_objective(const double* l_min, const double* l_max, const PathQpParams& prm) {
        // full-index Hessian band Pf[j][d], j = coefficient index 0..n+1; we only keep rows that matter:
        // accumulate directly into the free system and fold fixed coefficients into q.
        const double ds2 = prm.ds * prm.ds;
        const double a[3] = {1.0 / 6.0, 4.0 / 6.0, 1.0 / 6.0};
        const double b[3] = {1.0 / ds2, -2.0 / ds2, 1.0 / ds2};
        const double jk[4] = {-1.0 / ds2, 3.0 / ds2, -3.0 / ds2, 1.0 / ds2};
        const double wl2 = 2.0 * (prm.w_l + prm.w_centre), wd2 = 2.0 * prm.w_ddl, wj2 = 2.0 * prm.w_dddl;
        for (int m = 0; m < N; ++m) {
            q[m] = 0.0;
            for (int d = 0; d < 4; ++d) P[m * 4 + d] = 0.0;
        }
        auto add = [&](int jp, int jq, double val) {     // symmetric entry (jp <= jq), coefficient indices
            const int fp = jp - 3, fq = jq - 3;
            const bool p_free = fp >= 0 && fp < N, q_free = fq >= 0 && fq < N;
            if (p_free && q_free) {
                P[fp * 4 + (fq - fp)] += val;
            } else if (p_free) {
                q[fp] += val * cc[jq];                    // fixed partner -> linear term
            } else if (q_free) {
                q[fq] += val * cc[jp];
            }
        };
        for (int i = 0; i < n; ++i) {
            for (int p = 0; p < 3; ++p)
                for (int r = p; r < 3; ++r) add(i + p, i + r, wl2 * a[p] * a[r] + wd2 * b[p] * b[r]);
            if (i + 1 < n)
                for (int p = 0; p < 4; ++p)
                    for (int r = p; r < 4; ++r) add(i + p, i + r, wj2 * jk[p] * jk[r]);
            const double ctr = (l_min[i] + l_max[i]) / 2.0;                        // ref :201
            const double lin = -2.0 * prm.w_centre * ctr;                          // ref :204-205
            for (int p = 0; p < 3; ++p) {
                const int f = i + p - 3;
                if (f >= 0 && f < N) q[f] += lin * a[p];
            }
        }
    }

    EMP_HD void apply_P(const double* u, double* out) const {   // out = P u (free system, symmetric band)
        for (int m = 0; m < N; ++m) out[m] = 0.0;
        for (int m = 0; m < N; ++m) {
            out[m] += P[m * 4] * u[m];
            for (int d = 1; d < 4 && m + d < N; ++d) {
                out[m] += P[m * 4 + d] * u[m + d];
                out[m + d] += P[m * 4 + d] * u[m];
            }
        }
    }

    EMP_HD int interior_point(const double* l_min, const double* l_max, const PathQpParams& prm) {
        assemble_objective(l_min, l_max, prm);
        double* u = cc + 3;                                // free coefficients live inside cc
        const int i_lo = 1, i_hi = n - 2;                  // stations that involve a free coefficient
        const int m_rows = 4 * (i_hi - i_lo + 1);
        // ---- starting point: min 1/2 u'Pu + q'u + 1/2 sum (g.c - mid)^2 keeps u inside wide ranges
        for (int m = 0; m < N; ++m) {
            rhs[m] = -q[m];
            for (int d = 0; d < 4; ++d) M[m * 4 + d] = P[m * 4 + d];
        }
        if (!band_chol<3>(M, N)) return 2;
        for (int m = 0; m < N; ++m) u[m] = rhs[m];
        band_solve<3>(M, u, N);
        double qscale = 1.0;
        for (int m = 0; m < N; ++m) qscale = fmax(qscale, fabs(q[m]));
        double smin = 1e300, smax = 0.0;
        for (int i = i_lo; i <= i_hi; ++i) {
            const double vp = form(gp, i), vm = form(gm, i);
            s[4 * i + 0] = ub[i] - vp;
            s[4 * i + 1] = vp - lb[i];
            s[4 * i + 2] = ub[i] - vm;
            s[4 * i + 3] = vm - lb[i];
            for (int r = 0; r < 4; ++r) {
                smin = fmin(smin, s[4 * i + r]);
                smax = fmax(smax, s[4 * i + r]);
            }
        }
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;   // push every slack to >= 1
        for (int i = i_lo; i <= i_hi; ++i)
            for (int r = 0; r < 4; ++r) {
                s[4 * i + r] += shift;
                z[4 * i + r] = 1.0;
            }
        // Stopping rule.  mu = s'z/m is driven to 1e-12 (a weakly active row then sits within ~1e-6 of its
        // bound in BOTH slack and multiplier, i.e. x is good to ~1e-9 relative of the metre scale); the dual
        // residual is measured against the size of the multipliers it is made of.  Late factorizations see
        // z/s ~ 1e15 and may lose positive definiteness in round-off: an iterate that already met the
        // "acceptable" thresholds is then returned instead of an error.
        const double eps_mu = 1e-12, eps_p = 1e-9;
        bool acceptable = false;
        for (iters = 0; iters < kQpMaxIter; ++iters) {
            // ---- residuals
            apply_P(u, rhs);                               // rhs = P u
            double rd_max = 0.0, rp_max = 0.0, mu = 0.0, zmax = 0.0;
            for (int m = 0; m < N; ++m) rhs[m] += q[m];
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    scatter(r < 2 ? gp : gm, i, row_sign(r) * z[4 * i + r], rhs);
                    rp_max = fmax(rp_max, fabs(row_resid(i, r)));
                    mu += s[4 * i + r] * z[4 * i + r];
                    zmax = fmax(zmax, z[4 * i + r]);
                }
            mu /= (double)m_rows;
            for (int m = 0; m < N; ++m) rd_max = fmax(rd_max, fabs(rhs[m]));   // rhs = rd
            const double dscale = fmax(qscale, zmax);
            EMP_QP_TRACE("it %d rd %.3e rp %.3e mu %.3e\n", iters, rd_max, rp_max, mu);
            if (rd_max <= 1e-9 * dscale && rp_max <= eps_p && mu <= eps_mu) return 0;
            if (rd_max <= 1e-7 * dscale && rp_max <= 1e-8 && mu <= 1e-9) acceptable = true;
            if (!(mu == mu) || mu > 1e30) return 2;
            // ---- M = P + G'WG
            for (int m = 0; m < N; ++m)
                for (int d = 0; d < 4; ++d) M[m * 4 + d] = P[m * 4 + d];
            for (int i = i_lo; i <= i_hi; ++i) {
                const double wp = z[4 * i + 0] / s[4 * i + 0] + z[4 * i + 1] / s[4 * i + 1];
                const double wm = z[4 * i + 2] / s[4 * i + 2] + z[4 * i + 3] / s[4 * i + 3];
                for (int p = 0; p < 3; ++p) {
                    const int fp = i + p - 3;
                    if (fp < 0 || fp >= N) continue;
                    for (int r = p; r < 3; ++r) {
                        const int fr = i + r - 3;
                        if (fr >= N) continue;
                        M[fp * 4 + (fr - fp)] += wp * gp[p] * gp[r] + wm * gm[p] * gm[r];
                    }
                }
            }
            if (!band_chol<3>(M, N)) return acceptable ? 0 : 2;
            // ---- predictor: rc = s z
            for (int m = 0; m < N; ++m) dua[m] = -rhs[m];
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * i + r;
                    const double w = z[k] / s[k];
                    scatter(r < 2 ? gp : gm, i, -row_sign(r) * (w * row_resid(i, r) - z[k]), dua);
                }
            band_solve<3>(M, dua, N);
            double alpha = 1.0;
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * i + r;
                    const double dsr = -row_resid(i, r) - row_sign(r) * form_dir(r < 2 ? gp : gm, i, dua);
                    const double dzr = -z[k] - (z[k] / s[k]) * dsr;
                    if (dsr < 0.0) alpha = fmin(alpha, -s[k] / dsr);
                    if (dzr < 0.0) alpha = fmin(alpha, -z[k] / dzr);
                }
            double mu_aff = 0.0;
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * i + r;
                    const double dsr = -row_resid(i, r) - row_sign(r) * form_dir(r < 2 ? gp : gm, i, dua);
                    const double dzr = -z[k] - (z[k] / s[k]) * dsr;
                    mu_aff += (s[k] + alpha * dsr) * (z[k] + alpha * dzr);
                }
            mu_aff /= (double)m_rows;
            double sigma = mu_aff / mu;
            sigma = sigma * sigma * sigma;
            // ---- corrector: rc = s z + ds_a dz_a - sigma mu ; rhs still holds rd
            for (int m = 0; m < N; ++m) rhs[m] = -rhs[m];
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * i + r;
                    const double rp = row_resid(i, r);
                    const double dsa = -rp - row_sign(r) * form_dir(r < 2 ? gp : gm, i, dua);
                    const double dza = -z[k] - (z[k] / s[k]) * dsa;
                    const double rc = s[k] * z[k] + dsa * dza - sigma * mu;
                    scatter(r < 2 ? gp : gm, i, -row_sign(r) * ((z[k] * rp - rc) / s[k]), rhs);
                }
            band_solve<3>(M, rhs, N);                      // rhs = du
            alpha = 1e300;
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * i + r;
                    const double rp = row_resid(i, r);
                    const double dsa = -rp - row_sign(r) * form_dir(r < 2 ? gp : gm, i, dua);
                    const double dza = -z[k] - (z[k] / s[k]) * dsa;
                    const double rc = s[k] * z[k] + dsa * dza - sigma * mu;
                    const double dsr = -rp - row_sign(r) * form_dir(r < 2 ? gp : gm, i, rhs);
                    const double dzr = -(rc + z[k] * dsr) / s[k];
                    if (dsr < 0.0) alpha = fmin(alpha, -s[k] / dsr);
                    if (dzr < 0.0) alpha = fmin(alpha, -z[k] / dzr);
                }
            const double tau = (mu < 1e-6) ? 0.999 : 0.99;
            alpha = fmin(1.0, tau * alpha);
            EMP_QP_TRACE("      sigma %.3e alpha %.3e\n", sigma, alpha);
            // ---- update (slack/multiplier directions recomputed row by row BEFORE u moves)
            for (int i = i_lo; i <= i_hi; ++i)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * i + r;
                    const double rp = row_resid(i, r);
                    const double dsa = -rp - row_sign(r) * form_dir(r < 2 ? gp : gm, i, dua);
                    const double dza = -z[k] - (z[k] / s[k]) * dsa;
                    const double rc = s[k] * z[k] + dsa * dza - sigma * mu;
                    const double dsr = -rp - row_sign(r) * form_dir(r < 2 ? gp : gm, i, rhs);
                    const double dzr = -(rc + z[k] * dsr) / s[k];
                    z[k] += alpha * dzr;
                    // row_resid(i, r) reads only this row's own slack, so in-place updates are safe
                    s[k] += alpha * dsr;
                }
            for (int m = 0; m < N; ++m) u[m] += alpha * rhs[m];
        }
        return acceptable ? 0 : 2;
    }
};

// ---------------------------------------------------------------------------------------------
// Smoothing QP for ONE coordinate (ref smooth_reference_line, planning_utils.py:262-353).
// ---------------------------------------------------------------------------------------------
struct SmoothQpParams {
    double w_smooth, w_length, w_ref, thr;
};

template <int MMAX>
struct BoxQp {
    int m = 0;
    double x[MMAX], f[MMAX], lo[MMAX], hi[MMAX];
    double H[MMAX * 3], M[MMAX * 3];
    double su[MMAX], sl[MMAX], zu[MMAX], zl[MMAX];
    double rhs[MMAX], dxa[MMAX];
    int iters = 0;

    EMP_HD void apply_H(const double* v, double* out) const {
        for (int i = 0; i < m; ++i) out[i] = 0.0;
        for (int i = 0; i < m; ++i) {
            out[i] += H[i * 3] * v[i];
            for (int d = 1; d < 3 && i + d < m; ++d) {
                out[i] += H[i * 3 + d] * v[i + d];
                out[i + d] += H[i * 3 + d] * v[i];
            }
        }
    }

    // ref[i] = reference coordinate; result in x.  returns 0 ok, 2 not converged.
    EMP_HD int solve(const double* ref, int stride, int m_in, const SmoothQpParams& prm) {
        m = m_in;
        iters = 0;
        if (m < 2 || m > MMAX) return 2;
        // H = 2 (ws D2'D2 + wl D1'D1 + wr I)   (ref :313-344)
        for (int i = 0; i < m * 3; ++i) H[i] = 0.0;
        for (int i = 0; i < m; ++i) H[i * 3] += 2.0 * prm.w_ref;
        const double d2[3] = {1.0, -2.0, 1.0};
        for (int r = 0; r + 2 < m; ++r)
            for (int p = 0; p < 3; ++p)
                for (int c = p; c < 3; ++c) H[(r + p) * 3 + (c - p)] += 2.0 * prm.w_smooth * d2[p] * d2[c];
        const double d1[2] = {1.0, -1.0};
        for (int r = 0; r + 1 < m; ++r)
            for (int p = 0; p < 2; ++p)
                for (int c = p; c < 2; ++c) H[(r + p) * 3 + (c - p)] += 2.0 * prm.w_length * d1[p] * d1[c];
        double fscale = 1.0;
        for (int i = 0; i < m; ++i) {
            const double r = ref[i * stride];
            f[i] = -2.0 * prm.w_ref * r;                                            // ref :346
            lo[i] = r - prm.thr;                                                    // ref :308-311
            hi[i] = r + prm.thr;
            x[i] = r;                                   // strictly interior start (slack = thr both sides)
            su[i] = prm.thr;
            sl[i] = prm.thr;
            zu[i] = 1.0;
            zl[i] = 1.0;
            fscale = fmax(fscale, fabs(f[i]));
        }
        if (!(prm.thr > 0.0)) return 2;
        const double eps_mu = 1e-13, eps_p = 1e-10;        // box of 0.2 m: same rule as the path QP, tighter scale
        bool acceptable = false;
        for (iters = 0; iters < kQpMaxIter; ++iters) {
            apply_H(x, rhs);
            double rd_max = 0.0, rp_max = 0.0, mu = 0.0, zmax = 0.0;
            for (int i = 0; i < m; ++i) {
                rhs[i] += f[i] + zu[i] - zl[i];                                     // rd
                rd_max = fmax(rd_max, fabs(rhs[i]));
                rp_max = fmax(rp_max, fmax(fabs(x[i] - hi[i] + su[i]), fabs(lo[i] - x[i] + sl[i])));
                mu += su[i] * zu[i] + sl[i] * zl[i];
                zmax = fmax(zmax, fmax(zu[i], zl[i]));
            }
            mu /= (double)(2 * m);
            const double dscale = fmax(fscale, zmax);
            if (rd_max <= 1e-10 * dscale && rp_max <= eps_p && mu <= eps_mu) return 0;
            if (rd_max <= 1e-8 * dscale && rp_max <= 1e-9 && mu <= 1e-10) acceptable = true;
            if (!(mu == mu) || mu > 1e30) return 2;
            for (int i = 0; i < m * 3; ++i) M[i] = H[i];
            for (int i = 0; i < m; ++i) M[i * 3] += zu[i] / su[i] + zl[i] / sl[i];
            if (!band_chol<2>(M, m)) return acceptable ? 0 : 2;
            // predictor
            for (int i = 0; i < m; ++i) {
                const double rpu = x[i] - hi[i] + su[i], rpl = lo[i] - x[i] + sl[i];
                dxa[i] = -rhs[i] - ((zu[i] / su[i]) * rpu - zu[i]) + ((zl[i] / sl[i]) * rpl - zl[i]);
            }
            band_solve<2>(M, dxa, m);
            double alpha = 1.0, mu_aff = 0.0;
            for (int i = 0; i < m; ++i) {
                const double dsu = -(x[i] - hi[i] + su[i]) - dxa[i], dsl = -(lo[i] - x[i] + sl[i]) + dxa[i];
                const double dzu = -zu[i] - (zu[i] / su[i]) * dsu, dzl = -zl[i] - (zl[i] / sl[i]) * dsl;
                if (dsu < 0.0) alpha = fmin(alpha, -su[i] / dsu);
                if (dsl < 0.0) alpha = fmin(alpha, -sl[i] / dsl);
                if (dzu < 0.0) alpha = fmin(alpha, -zu[i] / dzu);
                if (dzl < 0.0) alpha = fmin(alpha, -zl[i] / dzl);
            }
            for (int i = 0; i < m; ++i) {
                const double dsu = -(x[i] - hi[i] + su[i]) - dxa[i], dsl = -(lo[i] - x[i] + sl[i]) + dxa[i];
                const double dzu = -zu[i] - (zu[i] / su[i]) * dsu, dzl = -zl[i] - (zl[i] / sl[i]) * dsl;
                mu_aff += (su[i] + alpha * dsu) * (zu[i] + alpha * dzu) + (sl[i] + alpha * dsl) * (zl[i] + alpha * dzl);
            }
            mu_aff /= (double)(2 * m);
            double sigma = mu_aff / mu;
            sigma = sigma * sigma * sigma;
            // corrector
            for (int i = 0; i < m; ++i) {
                const double rpu = x[i] - hi[i] + su[i], rpl = lo[i] - x[i] + sl[i];
                const double dsu = -rpu - dxa[i], dsl = -rpl + dxa[i];
                const double dzu = -zu[i] - (zu[i] / su[i]) * dsu, dzl = -zl[i] - (zl[i] / sl[i]) * dsl;
                const double rcu = su[i] * zu[i] + dsu * dzu - sigma * mu;
                const double rcl = sl[i] * zl[i] + dsl * dzl - sigma * mu;
                rhs[i] = -rhs[i] - (zu[i] * rpu - rcu) / su[i] + (zl[i] * rpl - rcl) / sl[i];
            }
            band_solve<2>(M, rhs, m);                       // rhs = dx
            alpha = 1e300;
            for (int i = 0; i < m; ++i) {
                const double rpu = x[i] - hi[i] + su[i], rpl = lo[i] - x[i] + sl[i];
                const double dsua = -rpu - dxa[i], dsla = -rpl + dxa[i];
                const double dzua = -zu[i] - (zu[i] / su[i]) * dsua, dzla = -zl[i] - (zl[i] / sl[i]) * dsla;
                const double rcu = su[i] * zu[i] + dsua * dzua - sigma * mu;
                const double rcl = sl[i] * zl[i] + dsla * dzla - sigma * mu;
                const double dsu = -rpu - rhs[i], dsl = -rpl + rhs[i];
                const double dzu = -(rcu + zu[i] * dsu) / su[i], dzl = -(rcl + zl[i] * dsl) / sl[i];
                if (dsu < 0.0) alpha = fmin(alpha, -su[i] / dsu);
                if (dsl < 0.0) alpha = fmin(alpha, -sl[i] / dsl);
                if (dzu < 0.0) alpha = fmin(alpha, -zu[i] / dzu);
                if (dzl < 0.0) alpha = fmin(alpha, -zl[i] / dzl);
            }
            const double tau = (mu < 1e-6) ? 0.999 : 0.99;
            alpha = fmin(1.0, tau * alpha);
            for (int i = 0; i < m; ++i) {
                const double rpu = x[i] - hi[i] + su[i], rpl = lo[i] - x[i] + sl[i];
                const double dsua = -rpu - dxa[i], dsla = -rpl + dxa[i];
                const double dzua = -zu[i] - (zu[i] / su[i]) * dsua, dzla = -zl[i] - (zl[i] / sl[i]) * dsla;
                const double rcu = su[i] * zu[i] + dsua * dzua - sigma * mu;
                const double rcl = sl[i] * zl[i] + dsla * dzla - sigma * mu;
                const double dsu = -rpu - rhs[i], dsl = -rpl + rhs[i];
                const double dzu = -(rcu + zu[i] * dsu) / su[i], dzl = -(rcl + zl[i] * dsl) / sl[i];
                zu[i] += alpha * dzu;
                zl[i] += alpha * dzl;
                su[i] += alpha * dsu;
                sl[i] += alpha * dsl;
                x[i] += alpha * rhs[i];
            }
        }
        return acceptable ? 0 : 2;
    }
};

}  // namespace emp
