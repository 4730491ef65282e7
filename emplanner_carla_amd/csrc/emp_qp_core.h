// emp_qp_core.h - the two QPs of the EM-Planner path as ONE generic banded interior-point problem.
//
// The reference builds DENSE matrices and hands them to cvxopt (ref: path_planning.py:103-214,
// planning_utils.py:300-353).  Both problems have a unique minimiser, so we pose them in the
// coordinates where they are banded and positive definite:
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
// Generic form ("banded range QP"):
//      minimise 1/2 u'Pu + q'u   s.t.   lo[t][f] <= c[t][f] + sum_p g[f][p] * u[t + off0 + p] <= hi[t][f]
// for stations t = 0..ns-1, forms f = 0..F-1, window p = 0..W-1 (window entries outside [0, N) belong to
// fixed variables and are folded into c).  P is SPD with half bandwidth KD, stored as its upper band.
//   path QP:      KD = 3, F = 2, W = 3, off0 = -2, ns = n - 2 (stations 1..n-2), N = n - 4
//   smoothing QP: KD = 2, F = 1, W = 1, off0 = 0,  ns = N = m
//
// Solver: Mehrotra predictor-corrector on the reduced normal equations (P + G' diag(z/s) G) du = r with a
// banded Cholesky.  This header holds the problem set-up (shared by host checks and kernels) and the SCALAR
// solver (one thread; used by the CPU logic checks).  emp_qp_wave.h holds the wave-cooperative solver the
// kernels run; both operate on the same arrays.
#pragma once

#include "emp_core.h"

#ifndef EMP_QP_TRACE
#define EMP_QP_TRACE(...)
#endif

namespace emp {

constexpr int kQpMaxIter = 60;
// Fraction of the step to the boundary.  It approaches 1 with the complementarity gap (as in Mehrotra-type
// codes: tau = max(0.99, 1 - mu)), which turns the endgame from a fixed 100x reduction of mu per iteration into
// a superlinear one.
#ifndef EMP_QP_TAU
#define EMP_QP_TAU(mu) fmax(0.99, 1.0 - (mu))
#endif
EMP_HD double qp_step_fraction(double mu) { return EMP_QP_TAU(mu); }
// Infeasible problems never close the primal residual: a full Newton step removes it entirely, and feasible
// problems of this family take one early (on the 902 feasible benchmark scenes of tools/qp_iters_probe.py the
// residual is below 1e-3 m by iteration 8 at the latest).  A residual still above 1e-3 m after kQpStallIter
// iterations is reported as failure instead of iterating until the multipliers overflow.
constexpr int kQpStallIter = 16;
constexpr double kQpStallResidual = 1e-3;

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

// ---------------------------------------------------------------------------------------------
// generic problem view: plain pointers into caller-owned storage (private arrays, LDS or host memory)
// ---------------------------------------------------------------------------------------------
template <int KD, int F, int W>
struct RangeQp {
    int N = 0, ns = 0, off0 = 0;
    double g[F][W];
    double* P = nullptr;     // [N][KD+1] upper band
    double* q = nullptr;     // [N]
    double* u = nullptr;     // [N] unknowns (in/out)
    double* c = nullptr;     // [ns][F] constant part of each form
    double* lo = nullptr;    // [ns][F]
    double* hi = nullptr;    // [ns][F]
    // work
    double* M = nullptr;     // [N][KD+1]
    double* s = nullptr;     // [ns][F][2]  slack of (upper, lower)
    double* z = nullptr;     // [ns][F][2]
    double* rhs = nullptr;   // [N]
    double* dua = nullptr;   // [N]
    double* tmp = nullptr;   // [ns][F][2] scratch (wave solver); unused by the scalar solver
    double* wgt = nullptr;   // [ns][F]    scratch (wave solver)
    double eps_p = 1e-9, eps_mu = 1e-12, eps_d_rel = 1e-9;   // stopping thresholds (see solve_scalar)
    // Starting multipliers: max(1, z0_rel * max_m P[m][m]).  With z0_rel = 1 they start at the scale of the
    // Hessian - the size a multiplier needs to move an unknown by one unit; path QPs through hard corridors
    // otherwise spend ten iterations growing z by three orders of magnitude (mean 9.6 -> 8.0 iterations, worst
    // case 25 -> 16 on the benchmark scenes).  The smoothing QP (0.2 m boxes, Hessian ~7) is best left at 1.
    double z0_rel = 0.0;
    EMP_HD double initial_multiplier(double pscale) const { return fmax(1.0, z0_rel * pscale); }
    int iters = 0;

    static constexpr int words(int N_, int ns_) {            // doubles of storage the pointers need
        return N_ * (KD + 1) * 2 + N_ * 4 + ns_ * F * 3 + ns_ * F * 2 * 3 + ns_ * F;
    }
    // carve the arrays out of one block of `words(N, ns)` doubles
    EMP_HD void bind(double* mem, int N_, int ns_) {
        N = N_;
        ns = ns_;
        P = mem; mem += N * (KD + 1);
        M = mem; mem += N * (KD + 1);
        q = mem; mem += N;
        u = mem; mem += N;
        rhs = mem; mem += N;
        dua = mem; mem += N;
        c = mem; mem += ns * F;
        lo = mem; mem += ns * F;
        hi = mem; mem += ns * F;
        s = mem; mem += ns * F * 2;
        z = mem; mem += ns * F * 2;
        tmp = mem; mem += ns * F * 2;
        wgt = mem; mem += ns * F;
    }

    // Storage of the one-row-per-lane solver (emp_qp_wave.h, range_qp_solve_wave_fast): arrays at offsets that depend
    // on the CAPACITY only, so that with a compile-time capacity every access is one per-lane base register plus an
    // immediate offset.  M, s and z (scalar / strided solvers only) are not bound.
    static constexpr int words_fast(int capN, int capS) { return capN * (KD + 1) + capN * 4 + capS * F * 5 + 4 * F; }   // + slack read past wgt
    EMP_HD void bind_fast(double* mem, int capN, int capS, int N_, int ns_) {
        N = N_;
        ns = ns_;
        P = mem; mem += capN * (KD + 1);
        q = mem; mem += capN;
        u = mem; mem += capN;
        rhs = mem; mem += capN;
        dua = mem; mem += capN;
        c = mem; mem += capS * F;
        lo = mem; mem += capS * F;
        hi = mem; mem += capS * F;
        tmp = mem; mem += capS * F;
        wgt = mem; mem += capS * F;
    }

    EMP_HD double form_val(int t, int f, const double* vec) const {      // sum_p g[f][p] vec[t+off0+p]
        double v = 0.0;
        for (int p = 0; p < W; ++p) {
            const int k = t + off0 + p;
            if (k >= 0 && k < N) v += g[f][p] * vec[k];
        }
        return v;
    }
    EMP_HD void scatter(int t, int f, double coef, double* vec) const {
        for (int p = 0; p < W; ++p) {
            const int k = t + off0 + p;
            if (k >= 0 && k < N) vec[k] += coef * g[f][p];
        }
    }
    EMP_HD void apply_P(const double* v, double* out) const {
        constexpr int B = KD + 1;
        for (int m = 0; m < N; ++m) out[m] = 0.0;
        for (int m = 0; m < N; ++m) {
            out[m] += P[m * B] * v[m];
            for (int d = 1; d <= KD && m + d < N; ++d) {
                out[m] += P[m * B + d] * v[m + d];
                out[m + d] += P[m * B + d] * v[m];
            }
        }
    }

    // ---- scalar Mehrotra predictor-corrector.  `u` must hold a starting guess.  returns 0 ok, 2 failed.
    // Stopping rule.  mu = s'z/m is driven to eps_mu (1e-12: a weakly active row then sits within ~1e-6 of
    // its bound in both slack and multiplier, i.e. u is good to ~1e-9 on the metre scale); the dual residual is
    // measured against the size of the multipliers it is made of.  Late factorizations see z/s ~ 1e15 and may
    // lose positive definiteness in round-off: an iterate that already met the "acceptable" thresholds is then
    // returned instead of an error.
    EMP_HD int solve_scalar() {
        constexpr int B = KD + 1;
        iters = 0;
        if (N <= 0) return 0;
        const int rows = ns * F * 2;
        double qscale = 1.0;
        for (int m = 0; m < N; ++m) qscale = fmax(qscale, fabs(q[m]));
        // slacks from the starting guess, pushed to >= 1 (infeasible-start IPM), unit multipliers
        double smin = 1e300;
        for (int t = 0; t < ns; ++t)
            for (int f = 0; f < F; ++f) {
                const double v = c[t * F + f] + form_val(t, f, u);
                s[(t * F + f) * 2 + 0] = hi[t * F + f] - v;
                s[(t * F + f) * 2 + 1] = v - lo[t * F + f];
                smin = fmin(smin, fmin(s[(t * F + f) * 2], s[(t * F + f) * 2 + 1]));
            }
        const double shift = (smin < 1.0) ? (1.0 - smin) : 0.0;
        double pscale = 0.0;
        for (int m = 0; m < N; ++m) pscale = fmax(pscale, P[m * B]);
        const double z0 = initial_multiplier(pscale);
        for (int r = 0; r < rows; ++r) {
            s[r] += shift;
            z[r] = z0;
        }
        bool acceptable = false;
        for (iters = 0; iters < kQpMaxIter; ++iters) {
            // ---- residuals: rhs <- rd = P u + q + G'z ; rp per row recomputed on the fly
            apply_P(u, rhs);
            double rd_max = 0.0, rp_max = 0.0, mu = 0.0, zmax = 0.0;
            for (int m = 0; m < N; ++m) rhs[m] += q[m];
            for (int t = 0; t < ns; ++t)
                for (int f = 0; f < F; ++f) {
                    const int k = (t * F + f) * 2;
                    const double v = c[t * F + f] + form_val(t, f, u);
                    scatter(t, f, z[k] - z[k + 1], rhs);
                    rp_max = fmax(rp_max, fmax(fabs(v - hi[t * F + f] + s[k]), fabs(lo[t * F + f] - v + s[k + 1])));
                    mu += s[k] * z[k] + s[k + 1] * z[k + 1];
                    zmax = fmax(zmax, fmax(z[k], z[k + 1]));
                }
            mu /= (double)rows;
            for (int m = 0; m < N; ++m) rd_max = fmax(rd_max, fabs(rhs[m]));
            const double dscale = fmax(qscale, zmax);
            EMP_QP_TRACE("it %d rd %.3e rp %.3e mu %.3e\n", iters, rd_max, rp_max, mu);
            if (rd_max <= eps_d_rel * dscale && rp_max <= eps_p && mu <= eps_mu) return 0;
            if (rd_max <= 100.0 * eps_d_rel * dscale && rp_max <= 10.0 * eps_p && mu <= 1000.0 * eps_mu) {
                acceptable = true;
                // complementarity converged, dual residual inside the acceptable band: further iterations at this mu
                // only add rounding noise to it (z / s weights of 1e15 and more in the normal matrix)
                if (rp_max <= eps_p && mu <= eps_mu) return 0;
            }
            if (!(mu == mu) || mu > 1e30) return 2;
            if (iters >= kQpStallIter && rp_max > kQpStallResidual) return 2;
            // ---- M = P + G'WG
            for (int m = 0; m < N * B; ++m) M[m] = P[m];
            for (int t = 0; t < ns; ++t)
                for (int f = 0; f < F; ++f) {
                    const int k = (t * F + f) * 2;
                    const double w = z[k] / s[k] + z[k + 1] / s[k + 1];
                    for (int p = 0; p < W; ++p) {
                        const int a = t + off0 + p;
                        if (a < 0 || a >= N) continue;
                        for (int r = p; r < W; ++r) {
                            const int b = t + off0 + r;
                            if (b >= N) continue;
                            M[a * B + (b - a)] += w * g[f][p] * g[f][r];
                        }
                    }
                }
            if (!band_chol<KD>(M, N)) return acceptable ? 0 : 2;
            // ---- predictor (rc = s z):  rhs_a = -rd - G'(w rp - z)
            for (int m = 0; m < N; ++m) dua[m] = -rhs[m];
            for (int t = 0; t < ns; ++t)
                for (int f = 0; f < F; ++f) {
                    const int k = (t * F + f) * 2;
                    const double v = c[t * F + f] + form_val(t, f, u);
                    const double rpu = v - hi[t * F + f] + s[k], rpl = lo[t * F + f] - v + s[k + 1];
                    const double tu = (z[k] / s[k]) * rpu - z[k], tl = (z[k + 1] / s[k + 1]) * rpl - z[k + 1];
                    scatter(t, f, -(tu - tl), dua);
                }
            band_solve<KD>(M, dua, N);
            double alpha = 1.0, mu_aff = 0.0;
            for (int pass = 0; pass < 2; ++pass)
                for (int t = 0; t < ns; ++t)
                    for (int f = 0; f < F; ++f) {
                        const int k = (t * F + f) * 2;
                        const double v = c[t * F + f] + form_val(t, f, u);
                        const double gd = form_val(t, f, dua);
                        const double dsu = -(v - hi[t * F + f] + s[k]) - gd, dsl = -(lo[t * F + f] - v + s[k + 1]) + gd;
                        const double dzu = -z[k] - (z[k] / s[k]) * dsu, dzl = -z[k + 1] - (z[k + 1] / s[k + 1]) * dsl;
                        if (pass == 0) {
                            if (dsu < 0.0) alpha = fmin(alpha, -s[k] / dsu);
                            if (dsl < 0.0) alpha = fmin(alpha, -s[k + 1] / dsl);
                            if (dzu < 0.0) alpha = fmin(alpha, -z[k] / dzu);
                            if (dzl < 0.0) alpha = fmin(alpha, -z[k + 1] / dzl);
                        } else {
                            mu_aff += (s[k] + alpha * dsu) * (z[k] + alpha * dzu) +
                                      (s[k + 1] + alpha * dsl) * (z[k + 1] + alpha * dzl);
                        }
                    }
            mu_aff /= (double)rows;
            double sigma = mu_aff / mu;
            sigma = sigma * sigma * sigma;
            // ---- corrector (rc = s z + ds_a dz_a - sigma mu): rhs <- -rd - G'((z rp - rc)/s)
            for (int m = 0; m < N; ++m) rhs[m] = -rhs[m];
            for (int t = 0; t < ns; ++t)
                for (int f = 0; f < F; ++f) {
                    const int k = (t * F + f) * 2;
                    const double v = c[t * F + f] + form_val(t, f, u);
                    const double gd = form_val(t, f, dua);
                    const double rpu = v - hi[t * F + f] + s[k], rpl = lo[t * F + f] - v + s[k + 1];
                    const double dsu = -rpu - gd, dsl = -rpl + gd;
                    const double dzu = -z[k] - (z[k] / s[k]) * dsu, dzl = -z[k + 1] - (z[k + 1] / s[k + 1]) * dsl;
                    const double rcu = s[k] * z[k] + dsu * dzu - sigma * mu;
                    const double rcl = s[k + 1] * z[k + 1] + dsl * dzl - sigma * mu;
                    scatter(t, f, -((z[k] * rpu - rcu) / s[k] - (z[k + 1] * rpl - rcl) / s[k + 1]), rhs);
                }
            band_solve<KD>(M, rhs, N);                      // rhs = du
            alpha = 1e300;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) alpha = fmin(1.0, qp_step_fraction(mu) * alpha);
                for (int t = 0; t < ns; ++t)
                    for (int f = 0; f < F; ++f) {
                        const int k = (t * F + f) * 2;
                        const double v = c[t * F + f] + form_val(t, f, u);
                        const double gda = form_val(t, f, dua), gd = form_val(t, f, rhs);
                        const double rpu = v - hi[t * F + f] + s[k], rpl = lo[t * F + f] - v + s[k + 1];
                        const double dsua = -rpu - gda, dsla = -rpl + gda;
                        const double dzua = -z[k] - (z[k] / s[k]) * dsua, dzla = -z[k + 1] - (z[k + 1] / s[k + 1]) * dsla;
                        const double rcu = s[k] * z[k] + dsua * dzua - sigma * mu;
                        const double rcl = s[k + 1] * z[k + 1] + dsla * dzla - sigma * mu;
                        const double dsu = -rpu - gd, dsl = -rpl + gd;
                        const double dzu = -(rcu + z[k] * dsu) / s[k], dzl = -(rcl + z[k + 1] * dsl) / s[k + 1];
                        if (pass == 0) {
                            if (dsu < 0.0) alpha = fmin(alpha, -s[k] / dsu);
                            if (dsl < 0.0) alpha = fmin(alpha, -s[k + 1] / dsl);
                            if (dzu < 0.0) alpha = fmin(alpha, -z[k] / dzu);
                            if (dzl < 0.0) alpha = fmin(alpha, -z[k + 1] / dzl);
                        } else {                            // rows only read their own s/z: update in place
                            z[k] += alpha * dzu;
                            z[k + 1] += alpha * dzl;
                            s[k] += alpha * dsu;
                            s[k + 1] += alpha * dsl;
                        }
                    }
            }
            for (int m = 0; m < N; ++m) u[m] += alpha * rhs[m];
        }
        return acceptable ? 0 : 2;
    }
};

// ---------------------------------------------------------------------------------------------
// Dual active-set solver (Goldfarb & Idnani 1983) for the banded range QP - scalar form.
// ---------------------------------------------------------------------------------------------
// The interior-point solver above spends 6-19 factorisations on a path QP whose solution has two or three active
// rows.  The Hessian P does not depend on the iterate, so ONE factorisation P = U'U serves the whole solve: start at
// the unconstrained minimiser, and while some row is violated add the most violated one to the active set by a step
// that keeps every active row active and the multipliers non-negative (dropping the row whose multiplier reaches zero
// first).  In the coordinates y = U u the objective is |y|^2 / 2 + ..., the active normals n~ = U^-T n are kept
// through an orthonormal basis Q1 and a triangular R (Q1 R = [n~_a]); a step needs one forward substitution (n~_p),
// k dot products, one back substitution (the step in u) - against a banded factorisation and two full solves per
// interior-point iteration - and the method ends after (rows added + rows dropped) steps with the exact minimiser,
// or proves infeasibility when a violated row can be neither reached nor traded (returns 1).
// Row id = (station * F + form) * 2 + side: side 0 is the upper bound c + g u <= hi, side 1 the lower bound.
// `M` must hold the band factor of P (band_chol) and `u` the unconstrained minimiser.  work: gi_words(N) doubles.
// Where it is used: as the second, independent solver the host tests hold the interior point against
// (tests/test_host_logic.py, `solver="gi"`); the kernels keep the interior point.  The wave-parallel form of this
// solver (tools/experiments/path_qp_dual_active_set_wave_r02.diff.txt) passed every GPU test but lost on the clock:
// its dependent chain per step (substitution sweep, k reductions, k-step triangular solve, sweep back) is as long as an
// interior-point iteration's, its slowest benchmark scene takes 20 steps, and Q1/R cost 31 KB of LDS per wavefront.
constexpr double kGiTolViolation = 1e-10;     // a row counts as violated below -1e-10 (metres)
constexpr double kGiTolRank = 1e-12;          // |z~|^2 <= 1e-12 |n~|^2: the row's normal lies in the span of the active ones
EMP_HD constexpr int gi_words(int N) { return 2 * N * N + 8 * N; }

template <int KD>
EMP_HD void band_fwd(const double* uf, double* x, int n) {     // U' y = b
    constexpr int W = KD + 1;
    for (int i = 0; i < n; ++i) {
        double sum = x[i];
        const int k0 = (i - KD > 0) ? i - KD : 0;
        for (int k = k0; k < i; ++k) sum -= uf[k * W + (i - k)] * x[k];
        x[i] = sum / uf[i * W];
    }
}
template <int KD>
EMP_HD void band_bwd(const double* uf, double* x, int n) {     // U x = y
    constexpr int W = KD + 1;
    for (int i = n - 1; i >= 0; --i) {
        double sum = x[i];
        for (int d = 1; d <= KD && i + d < n; ++d) sum -= uf[i * W + d] * x[i + d];
        x[i] = sum / uf[i * W];
    }
}

template <int KD, int F, int W>
EMP_HD int range_qp_gi_scalar(RangeQp<KD, F, W>& Q, double* work, int iter_cap = 0) {
    const int N = Q.N, ns = Q.ns;
    Q.iters = 0;
    if (N <= 0) return 0;
    double* Qm = work;                 // [N][N]  column a of Q1 at Qm[a*N ..]
    double* R = Qm + N * N;            // [N][N]  R[i*N + j], upper triangular
    double* lam = R + N * N;           // [N]
    double* nt = lam + N;              // n~_p
    double* zt = nt + N;               // (I - Q1 Q1') n~_p
    double* dv = zt + N;               // Q1' n~_p
    double* rv = dv + N;               // R^-1 d
    double* zu = rv + N;               // step in u
    double* idsd = zu + N;             // [N] active row ids (as doubles: one storage class)
    double* spare = idsd + N;
    (void)spare;
    const int cap = iter_cap > 0 ? iter_cap : 4 * N + 2 * ns * F + 20;
    int k = 0;
    auto slack = [&](int id) {
        const int side = id & 1, tf = id >> 1, t = tf / F, f = tf - t * F;
        const double v = Q.c[tf] + Q.form_val(t, f, Q.u);
        return side == 0 ? Q.hi[tf] - v : v - Q.lo[tf];
    };
    auto is_active = [&](int id) {
        for (int a = 0; a < k; ++a)
            if ((int)idsd[a] == id) return true;
        return false;
    };
    for (;;) {
        // ---- the most violated row that is not active
        int p = -1;
        double worst = -kGiTolViolation;
        for (int id = 0; id < ns * F * 2; ++id) {
            const double sl = slack(id);
            if (sl < worst && !is_active(id)) {
                worst = sl;
                p = id;
            }
        }
        if (p < 0) return 0;                                   // every row holds: u is the minimiser
        // n~_p = U^-T n_p with n_p = +-g on the row's window (as inequality n'u >= b)
        const int side = p & 1, tf = p >> 1, t = tf / F, f = tf - t * F;
        const double sign = side == 0 ? -1.0 : 1.0;
        for (int m = 0; m < N; ++m) nt[m] = 0.0;
        for (int w = 0; w < W; ++w) {
            const int m = t + Q.off0 + w;
            if (m >= 0 && m < N) nt[m] = sign * Q.g[f][w];
        }
        band_fwd<KD>(Q.M, nt, N);
        double nn = 0.0;
        for (int m = 0; m < N; ++m) nn += nt[m] * nt[m];
        double lam_p = 0.0;
        for (;;) {                                             // partial steps until row p is active (or proves infeasible)
            if (++Q.iters > cap) return 2;
            for (int a = 0; a < k; ++a) {
                double acc = 0.0;
                for (int m = 0; m < N; ++m) acc += Qm[a * N + m] * nt[m];
                dv[a] = acc;
            }
            double zz = 0.0;
            for (int m = 0; m < N; ++m) {
                double v = nt[m];
                for (int a = 0; a < k; ++a) v -= dv[a] * Qm[a * N + m];
                zt[m] = v;
                zz += v * v;
            }
            for (int a = k - 1; a >= 0; --a) {                 // r = R^-1 d
                double acc = dv[a];
                for (int j = a + 1; j < k; ++j) acc -= R[a * N + j] * rv[j];
                rv[a] = acc / R[a * N + a];
            }
            double t1 = INFINITY;
            int kd = -1;
            for (int a = 0; a < k; ++a)
                if (rv[a] > kGiTolRank) {
                    const double ta = lam[a] / rv[a];
                    if (ta < t1) {
                        t1 = ta;
                        kd = a;
                    }
                }
            const bool full = zz > kGiTolRank * nn;
            const double t2 = full ? -slack(p) / zz : INFINITY;
            if (!(t1 < INFINITY) && !full) return 1;           // the row can be neither reached nor traded: infeasible
            const double tt = t1 < t2 ? t1 : t2;
            if (full) {
                for (int m = 0; m < N; ++m) zu[m] = zt[m];
                band_bwd<KD>(Q.M, zu, N);
                for (int m = 0; m < N; ++m) Q.u[m] += tt * zu[m];
            }
            for (int a = 0; a < k; ++a) lam[a] -= tt * rv[a];
            lam_p += tt;
            if (full && t2 <= t1) {                            // full step: row p joins the active set
                const double nz = sqrt(zz);
                for (int m = 0; m < N; ++m) Qm[k * N + m] = zt[m] / nz;
                for (int a = 0; a < k; ++a) R[a * N + k] = dv[a];
                R[k * N + k] = nz;
                idsd[k] = (double)p;
                lam[k] = lam_p;
                ++k;
                break;
            }
            // partial step: active row kd leaves; column kd of R goes, Givens rotations restore the triangle
            for (int a = kd; a + 1 < k; ++a) {
                idsd[a] = idsd[a + 1];
                lam[a] = lam[a + 1];
                for (int i = 0; i < k; ++i) R[i * N + a] = R[i * N + a + 1];
            }
            --k;
            for (int j = kd; j < k; ++j) {
                const double a0 = R[j * N + j], b0 = R[(j + 1) * N + j];
                const double hh = sqrt(a0 * a0 + b0 * b0);
                const double cg = a0 / hh, sg = b0 / hh;
                for (int col = j; col < k; ++col) {
                    const double x0 = R[j * N + col], x1 = R[(j + 1) * N + col];
                    R[j * N + col] = cg * x0 + sg * x1;
                    R[(j + 1) * N + col] = -sg * x0 + cg * x1;
                }
                for (int m = 0; m < N; ++m) {
                    const double x0 = Qm[j * N + m], x1 = Qm[(j + 1) * N + m];
                    Qm[j * N + m] = cg * x0 + sg * x1;
                    Qm[(j + 1) * N + m] = -sg * x0 + cg * x1;
                }
            }
        }
    }
}

using PathRangeQp = RangeQp<3, 2, 3>;
using BoxRangeQp = RangeQp<2, 1, 1>;

// ---------------------------------------------------------------------------------------------
// Path QP set-up / finish (ref Quadratic_planning, path_planning.py:78-219)
// ---------------------------------------------------------------------------------------------
struct PathQpParams {
    double ds, w_l, w_ddl, w_dddl, w_centre, d1, d2, host_w;
};

// window offset and the two corner forms l + d1 dl, l - d2 dl on a coefficient window (c_{i-1}, c_i, c_{i+1});
// these live in the (per-thread) struct, so every lane that uses Q must call this
EMP_HD void path_qp_forms(PathRangeQp& Q, const PathQpParams& prm) {
    const double ds = prm.ds;
    Q.off0 = -2;
    Q.z0_rel = 1.0;
    Q.g[0][0] = 1.0 / 6.0 - prm.d1 / (2.0 * ds);
    Q.g[0][1] = 4.0 / 6.0;
    Q.g[0][2] = 1.0 / 6.0 + prm.d1 / (2.0 * ds);
    Q.g[1][0] = 1.0 / 6.0 + prm.d2 / (2.0 * ds);
    Q.g[1][1] = 4.0 / 6.0;
    Q.g[1][2] = 1.0 / 6.0 - prm.d2 / (2.0 * ds);
}

// Fills Q (bound to storage for N = n-4, ns = n-2) and cc[n+2] (fixed B-spline coefficients, free ones = 0).
// returns 0 ok, 1 infeasible by inspection, 2 unsupported size.  Q.u is set to 0 (caller picks the start).
EMP_HD int path_qp_setup(PathRangeQp& Q, double* cc, const double* l_min, const double* l_max, int n, double l0,
                         double dl0, double ddl0, const PathQpParams& prm) {
    if (n < 4) return 2;
    const int N = n - 4;
    const double ds = prm.ds;
    const double hw = fabs(prm.host_w) / 2.0;
    const int fwd = (int)ceil(prm.d1 / ds), back = (int)ceil(prm.d2 / ds);       // ref :126-127
    path_qp_forms(Q, prm);
    // pinned start / end states -> fixed coefficients (index j+1 holds c_j)
    for (int j = 0; j < n + 2; ++j) cc[j] = 0.0;
    const double c0 = l0 - ds * ds * ddl0 / 6.0;
    cc[1] = c0;
    cc[2] = c0 + ds * ds * ddl0 / 2.0 + ds * dl0;
    cc[0] = c0 + ds * ds * ddl0 / 2.0 - ds * dl0;
    const double tol = 1e-9;
    // station ranges (ref :130-142), fixed stations 0 and n-1 checked by inspection
    for (int i = 0; i < n; ++i) {
        const int i1 = (i + fwd < n - 1) ? i + fwd : n - 1;                       // ref :130
        const int i2 = (i - back > 0) ? i - back : 0;                             // ref :131
        const double ub = l_max[i1] - hw, lb = l_min[i2] + hw;
        if (lb > ub + tol) return 1;
        for (int f = 0; f < 2; ++f) {
            const double v = Q.g[f][0] * cc[i] + Q.g[f][1] * cc[i + 1] + Q.g[f][2] * cc[i + 2];   // fixed part
            if (i == 0 || i == n - 1) {
                if (v > ub + tol || v < lb - tol) return 1;
            } else {
                const int t = i - 1;
                Q.c[t * 2 + f] = v;
                Q.lo[t * 2 + f] = lb;
                Q.hi[t * 2 + f] = ub;
            }
        }
    }
    // objective: F = sum (w_l + w_c) l_i^2 + w_ddl ddl_i^2 - 2 w_c ctr_i l_i + w_dddl sum (ddl_{i+1} - ddl_i)^2
    // (ref :193-205 with H = 2 * (...); the w_cost_dl product is identically zero, :193; end terms act on the
    // pinned end state only)
    const double ds2 = ds * ds;
    const double a[3] = {1.0 / 6.0, 4.0 / 6.0, 1.0 / 6.0};
    const double b[3] = {1.0 / ds2, -2.0 / ds2, 1.0 / ds2};
    const double jk[4] = {-1.0 / ds2, 3.0 / ds2, -3.0 / ds2, 1.0 / ds2};
    const double wl2 = 2.0 * (prm.w_l + prm.w_centre), wd2 = 2.0 * prm.w_ddl, wj2 = 2.0 * prm.w_dddl;
    for (int m = 0; m < N; ++m) {
        Q.q[m] = 0.0;
        Q.u[m] = 0.0;
        for (int d = 0; d < 4; ++d) Q.P[m * 4 + d] = 0.0;
    }
    auto add = [&](int jp, int jq, double val) {        // symmetric entry, coefficient indices jp <= jq
        const int fp = jp - 3, fq = jq - 3;
        const bool pf = fp >= 0 && fp < N, qf = fq >= 0 && fq < N;
        if (pf && qf) Q.P[fp * 4 + (fq - fp)] += val;
        else if (pf) Q.q[fp] += val * cc[jq];           // fixed partner -> linear term
        else if (qf) Q.q[fq] += val * cc[jp];
    };
    for (int i = 0; i < n; ++i) {
        for (int p = 0; p < 3; ++p)
            for (int r = p; r < 3; ++r) add(i + p, i + r, wl2 * a[p] * a[r] + wd2 * b[p] * b[r]);
        if (i + 1 < n)
            for (int p = 0; p < 4; ++p)
                for (int r = p; r < 4; ++r) add(i + p, i + r, wj2 * jk[p] * jk[r]);
        const double lin = -2.0 * prm.w_centre * ((l_min[i] + l_max[i]) / 2.0);   // ref :201-205
        for (int p = 0; p < 3; ++p) {
            const int f = i + p - 3;
            if (f >= 0 && f < N) Q.q[f] += lin * a[p];
        }
    }
    return 0;
}

// cc (with the solved free coefficients copied in) -> l, dl, ddl per station
EMP_HD void path_qp_finish(const double* cc, int n, double ds, double* out_l, double* out_dl, double* out_ddl) {
    for (int i = 0; i < n; ++i) {
        out_l[i] = (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0;
        if (out_dl) out_dl[i] = (cc[i + 2] - cc[i]) / (2.0 * ds);
        if (out_ddl) out_ddl[i] = (cc[i] - 2.0 * cc[i + 1] + cc[i + 2]) / (ds * ds);
    }
}

// doubles of scratch path_qp_solve_scalar needs for n stations
// two scenes per wavefront (n <= 34 stations: N, ns <= 32): fixed layout, 36 coefficient slots + the solver's arrays
EMP_HD constexpr int path_qp_words_pair() { return 36 + PathRangeQp::words_fast(32, 32); }
EMP_HD constexpr int path_qp_words(int n) { return PathRangeQp::words(n - 4 > 0 ? n - 4 : 0, n - 2 > 0 ? n - 2 : 0) + n + 2; }

// complete scalar path QP on caller storage `mem` (path_qp_words(n) doubles)
// gi_work != NULL: the dual active-set solver (gi_words(n - 4) doubles) instead of the interior point
EMP_HD int path_qp_solve_scalar(double* mem, const double* l_min, const double* l_max, int n, double l0, double dl0,
                                double ddl0, const PathQpParams& prm, double* out_l, double* out_dl, double* out_ddl,
                                int* iters, double* gi_work = nullptr) {
    *iters = 0;
    if (n < 4) return 2;
    PathRangeQp Q;
    double* cc = mem;
    Q.bind(mem + n + 2, n - 4, n - 2);
    int rc = path_qp_setup(Q, cc, l_min, l_max, n, l0, dl0, ddl0, prm);
    if (rc) return rc;
    if (Q.N > 0) {
        // start from the unconstrained minimiser P u = -q
        for (int m = 0; m < Q.N * 4; ++m) Q.M[m] = Q.P[m];
        if (!band_chol<3>(Q.M, Q.N)) return 2;
        for (int m = 0; m < Q.N; ++m) Q.u[m] = -Q.q[m];
        band_solve<3>(Q.M, Q.u, Q.N);
        rc = gi_work ? range_qp_gi_scalar(Q, gi_work) : Q.solve_scalar();
        *iters = Q.iters;
        if (rc) return rc;
        for (int m = 0; m < Q.N; ++m) cc[m + 3] = Q.u[m];
    } else {
        for (int t = 0; t < Q.ns; ++t)
            for (int f = 0; f < 2; ++f)
                if (Q.c[t * 2 + f] > Q.hi[t * 2 + f] + 1e-9 || Q.c[t * 2 + f] < Q.lo[t * 2 + f] - 1e-9) return 1;
    }
    path_qp_finish(cc, n, prm.ds, out_l, out_dl, out_ddl);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Smoothing QP set-up for ONE coordinate (ref smooth_reference_line, planning_utils.py:262-353)
// ---------------------------------------------------------------------------------------------
struct SmoothQpParams {
    double w_smooth, w_length, w_ref, thr;
};

EMP_HD void box_qp_forms(BoxRangeQp& Q) {
    Q.off0 = 0;
    Q.g[0][0] = 1.0;
    Q.eps_p = 1e-10;                                     // box of 0.2 m: same rule as the path QP, tighter scale
    Q.eps_mu = 1e-13;
    Q.eps_d_rel = 1e-10;
}

// Q bound to N = ns = m.  ref[i*stride] = reference coordinate.  u starts at the reference (strictly interior).
EMP_HD int box_qp_setup(BoxRangeQp& Q, const double* ref, int stride, int m, const SmoothQpParams& prm) {
    if (m < 2 || !(prm.thr > 0.0)) return 2;
    box_qp_forms(Q);
    for (int i = 0; i < m * 3; ++i) Q.P[i] = 0.0;
    for (int i = 0; i < m; ++i) Q.P[i * 3] += 2.0 * prm.w_ref;                     // H = 2 (ws D2'D2 + wl D1'D1 + wr I)
    const double d2[3] = {1.0, -2.0, 1.0};
    for (int r = 0; r + 2 < m; ++r)
        for (int p = 0; p < 3; ++p)
            for (int cidx = p; cidx < 3; ++cidx) Q.P[(r + p) * 3 + (cidx - p)] += 2.0 * prm.w_smooth * d2[p] * d2[cidx];
    const double d1[2] = {1.0, -1.0};
    for (int r = 0; r + 1 < m; ++r)
        for (int p = 0; p < 2; ++p)
            for (int cidx = p; cidx < 2; ++cidx) Q.P[(r + p) * 3 + (cidx - p)] += 2.0 * prm.w_length * d1[p] * d1[cidx];
    for (int i = 0; i < m; ++i) {
        const double r = ref[i * stride];
        Q.q[i] = -2.0 * prm.w_ref * r;                                             // ref :346
        Q.c[i] = 0.0;
        Q.lo[i] = r - prm.thr;                                                     // ref :308-311
        Q.hi[i] = r + prm.thr;
        Q.u[i] = r;
    }
    return 0;
}

}  // namespace emp
