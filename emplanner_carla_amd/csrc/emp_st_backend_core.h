// emp_st_backend_core.h - scalar arithmetic of the S-T speed planning back end:
// convex space, speed QP set-up, densification, path-speed merge.  Usable from HIP device code and from plain C++
// (tests/host_check).  oracle/st_backend.py states the same arithmetic in NumPy.
//
// Arithmetic contract: as in emp_core.h - written order, separately rounded binary64 operations,
// -ffp-contract=off.  The one place where the reference itself is not reproducible to the last bit is x ** 2 on
// NumPy scalars in increase_points (libm pow, not always x * x); the kernels square with a multiplication.
//
// "ref:" comments cite reference planner/speed_planning_test.py.
#pragma once

#include "emp_core.h"
#include "emp_qp_core.h"

namespace emp {
namespace stb {

constexpr int kDp = 16;        // DP columns, ref :318
constexpr int kQp = 17;        // QP stations incl. the planning start, ref :428
constexpr int kDense = 401;    // ref :541, :576

// status bits of the back-end entry points (include/emplanner.h EMP_STB_*)
constexpr int kStbRange = 2;       // interp1d bounds error (ValueError in the reference)
constexpr int kStbIndex = 4;       // IndexError in the reference
constexpr int kStbQpFailed = 8;    // speed QP infeasible / not converged
constexpr int kStbNoProfile = 64;  // the DP profile starts with NaN: nothing to work on (the reference's behaviour
                                   // on such input is an accident of negative slice indices)

struct Interp {
    double y;
    bool out_of_range;
};

// scipy.interpolate.interp1d(x, y)(x_new), linear, bounds_error=True, for ascending x of length n >= 2 given by
// accessors; the interval is searchsorted(x, x_new) clipped to [1, n-1].  ref :341, :357
template <class FX, class FY>
EMP_HD Interp interp1d(FX x, FY y, int n, double x_new) {
    Interp r{0.0, false};
    if (n < 1 || x_new < x(0) || x_new > x(n - 1)) {
        r.out_of_range = true;
        return r;
    }
    if (n < 2) {                    // scipy divides by zero here; flagged as out of range
        r.out_of_range = true;
        return r;
    }
    int idx = 0, hi = n;            // first index with x[idx] >= x_new  (searchsorted, side='left'): bisection
    while (idx < hi) {
        const int mid = (idx + hi) >> 1;
        if (x(mid) < x_new) idx = mid + 1;
        else hi = mid;
    }
    if (idx < 1) idx = 1;
    if (idx > n - 1) idx = n - 1;
    const double xlo = x(idx - 1), xhi = x(idx), ylo = y(idx - 1), yhi = y(idx);
    const double slope = (yhi - ylo) / (xhi - xlo);
    r.y = slope * (x_new - xlo) + ylo;
    return r;
}

// ref :361-382 - the DP column whose [t_j, t_j+1) holds t, else 0
EMP_HD int time_index(const double* dp_t, double t) {
    for (int j = 0; j < kDp - 1; ++j) {
        if (dp_t[0] > t) return j;
        if (dp_t[j] <= t && t < dp_t[j + 1]) return j;
    }
    return 0;
}

// ref :308-407 (generate_convex_space).  path_index2s / kappa: `path_len` entries as handed to the reference (may be
// zero padded); obstacle sets: n_slots entries, NaN = empty.  Outputs [16] each.  Returns the status bits.
EMP_HD int convex_space(const double* dp_s, const double* dp_t, const double* idx2s, const double* kappa, int path_len,
                        const double* s_in, const double* s_out, const double* t_in, const double* t_out, int n_slots,
                        double max_lateral_accel, double* s_lb, double* s_ub, double* sd_lb, double* sd_ub) {
    const double inf = __builtin_inf();
    for (int i = 0; i < kDp; ++i) {
        s_lb[i] = -inf;
        s_ub[i] = inf;
        sd_lb[i] = -inf;
        sd_ub[i] = inf;
    }
    if (dp_s[0] != dp_s[0]) return kStbNoProfile;
    int path_end = path_len;                                     // ref :326-330
    for (int k = 1; k < path_len; ++k) {
        if (idx2s[k] == 0.0 && idx2s[k - 1] != 0.0) {
            path_end = k - 1;
            break;
        }
        path_end = k;
    }
    int dp_end = kDp;                                            // ref :333-336
    for (int k = 0; k < kDp; ++k)
        if (dp_s[k] != dp_s[k]) {
            dp_end = k - 1;
            break;
        }
    for (int i = 0; i < kDp; ++i) {                              // ref :339-347
        if (dp_s[i] != dp_s[i]) break;
        const Interp kp = interp1d([&](int k) { return idx2s[k]; }, [&](int k) { return kappa[k]; }, path_end, dp_s[i]);
        if (kp.out_of_range) return kStbRange;
        sd_lb[i] = 0.0;
        sd_ub[i] = sqrt(max_lateral_accel / (fabs(kp.y) + 1e-10));
    }
    for (int i = 0; i < n_slots; ++i) {                          // ref :349-405
        if (s_in[i] != s_in[i]) continue;
        const double obs_t = (t_in[i] + t_out[i]) / 2.0;
        const double obs_s = (s_in[i] + s_out[i]) / 2.0;
        const double obs_speed = (s_out[i] - s_in[i]) / (t_out[i] - t_in[i]);
        // [0] + dp_speed_t[0:dp_end] against [0] + dp_speed_s[0:dp_end]  (the slice drops the last valid column)
        const Interp dp = interp1d([&](int k) { return k == 0 ? 0.0 : dp_t[k - 1]; },
                                   [&](int k) { return k == 0 ? 0.0 : dp_s[k - 1]; }, dp_end + 1, obs_t);
        if (dp.out_of_range) return kStbRange;
        int t_lb = time_index(dp_t, t_in[i]) - 2;
        int t_ub = time_index(dp_t, t_out[i]) + 2;
        if (t_lb < 3) t_lb = 3;                                  // ref :385
        if (t_ub > dp_end) t_ub = dp_end;                        // ref :386
        for (int m = t_lb; m <= t_ub; ++m) {
            if (m >= kDp) return kStbIndex;                      // s_ub[16] / dp_speed_t[16]
            const double line = s_in[i] + obs_speed * (dp_t[m] - t_in[i]);
            if (obs_s > dp.y) s_ub[m] = (line < s_ub[m]) ? line : s_ub[m];     // yield, ref :388-393  (min(a, b): b unless a < b)
            else s_lb[m] = (line > s_lb[m]) ? line : s_lb[m];                   // overtake, ref :394-399
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Speed QP (ref :410-511) as a banded range QP in cubic B-spline coefficients.
//
// The reference's variables are X = (s_i, s_dot_i, s_dot2_i) per time station i = 0..n-1 tied together by the
// continuity equations Aeq' X = 0 (:452-458): s_dot2 piecewise linear, s_dot and s continuous - i.e. s(t) is a C2
// cubic spline on the uniform knots i dt.  In the cubic B-spline basis
//     s_i = (c_{i-1} + 4 c_i + c_{i+1}) / 6,  s_dot_i = (c_{i+1} - c_{i-1}) / (2 dt),  s_dot2_i = (c_{i-1} - 2 c_i + c_{i+1}) / dt^2
// the equations hold identically, the pinned start (0, v0, a0) fixes c_{-1}, c_0, c_1, and the n-1 free
// coefficients c_2..c_n see a heptadiagonal Hessian.  Per station i >= 1 four forms on the window c_{i-1}..c_{i+2}:
// s_i, s_dot_i, s_dot2_i (bounds :472-479) and s_{i+1} - s_i >= 0 (:463-468).  The bounds are the INTENDED ones
// (separate lb / ub; the reference aliases them and never passes them, see oracle/st_backend.py).
// ---------------------------------------------------------------------------------------------
using SpeedRangeQp = RangeQp<3, 4, 4>;
constexpr double kSpeedBig = 1e4;          // stands in for an infinite bound: no s (m) or s_dot (m/s) of an 8 s horizon with
                                           // s_dot2 <= 4 comes near it (curvature bounds on straights reach 1.4e5 m/s)

struct SpeedQpParams {
    double w_s_dot2, w_v_ref, w_jerk, v_ref;
};

EMP_HD void speed_qp_forms(SpeedRangeQp& Q, double dt) {
    const double dt2 = dt * dt;
    Q.off0 = -2;
    Q.z0_rel = 1.0;
    const double gs[4] = {1.0 / 6.0, 4.0 / 6.0, 1.0 / 6.0, 0.0};
    const double gv[4] = {-1.0 / (2.0 * dt), 0.0, 1.0 / (2.0 * dt), 0.0};
    const double ga[4] = {1.0 / dt2, -2.0 / dt2, 1.0 / dt2, 0.0};
    const double gm[4] = {-1.0 / 6.0, -3.0 / 6.0, 3.0 / 6.0, 1.0 / 6.0};
    for (int p = 0; p < 4; ++p) {
        Q.g[0][p] = gs[p];
        Q.g[1][p] = gv[p];
        Q.g[2][p] = ga[p];
        Q.g[3][p] = gm[p];
    }
}

// ref :424-441 - number of valid DP columns; returns qp_size (>= 2) or a negative status
EMP_HD int speed_qp_size(const double* dp_s) {
    int dp_end = kDp;
    for (int i = 0; i < kDp; ++i)
        if (dp_s[i] != dp_s[i]) {
            dp_end = i - 1;
            break;
        }
    if (dp_end >= kDp) return -kStbIndex;           // dp_speed_s[16], ref :435
    if (dp_end < 0) return -kStbNoProfile;
    if (dp_end == 0) return -kStbQpFailed;          // dt = T / 0
    return dp_end + 1;
}

EMP_HD constexpr int speed_qp_words(int n) { return SpeedRangeQp::words(n - 1, n - 1) + n + 2; }

// Fills Q (bound to N = ns = n-1) and cc[n+2].  returns 0 ok, 1 infeasible by inspection.
EMP_HD int speed_qp_setup(SpeedRangeQp& Q, double* cc, int n, double dt, double v0, double a0, const double* s_lb,
                          const double* s_ub, const double* sd_lb, const double* sd_ub, const SpeedQpParams& prm) {
    const int N = n - 1;
    const double dt2 = dt * dt;
    speed_qp_forms(Q, dt);
    for (int j = 0; j < n + 2; ++j) cc[j] = 0.0;
    const double c0 = 0.0 - dt2 * a0 / 6.0;                                   // s_0 = 0, ref :482
    cc[1] = c0;
    cc[2] = c0 + dt2 * a0 / 2.0 + dt * v0;
    cc[0] = c0 + dt2 * a0 / 2.0 - dt * v0;
    auto clampb = [](double v) { return v > kSpeedBig ? kSpeedBig : (v < -kSpeedBig ? -kSpeedBig : v); };
    for (int i = 1; i < n; ++i) {                                             // bounds of station i: DP column i-1
        const int t = i - 1;
        double lo[4] = {clampb(s_lb[i - 1]), clampb(sd_lb[i - 1]), -6.0, 0.0};
        double hi[4] = {clampb(s_ub[i - 1]), clampb(sd_ub[i - 1]), 4.0, kSpeedBig};
        if (i == 1 && lo[0] < 0.0) lo[0] = 0.0;                              // s_0 - s_1 <= 0 with s_0 = 0
        if (i == n - 1) {                                                     // no station behind the last one
            lo[3] = -kSpeedBig;
            hi[3] = kSpeedBig;
        }
        for (int f = 0; f < 4; ++f) {
            if (lo[f] > hi[f] + 1e-9) return 1;
            double v = 0.0;
            for (int p = 0; p < 4; ++p)
                if (i + p < n + 2) v += Q.g[f][p] * cc[i + p];               // fixed coefficients only (free ones are 0)
            Q.c[t * 4 + f] = v;
            Q.lo[t * 4 + f] = lo[f];
            Q.hi[t * 4 + f] = hi[f];
        }
    }
    // objective (ref :489-500, H = 2 (...)): sum_i w_a s_dot2_i^2 + w_v (s_dot_i - v_ref)^2 + sum_i w_j (s_dot2_{i+1} - s_dot2_i)^2
    const double bv[3] = {-1.0 / (2.0 * dt), 0.0, 1.0 / (2.0 * dt)};
    const double ba[3] = {1.0 / dt2, -2.0 / dt2, 1.0 / dt2};
    const double jk[4] = {-1.0 / dt2, 3.0 / dt2, -3.0 / dt2, 1.0 / dt2};
    const double wa2 = 2.0 * prm.w_s_dot2, wv2 = 2.0 * prm.w_v_ref, wj2 = 2.0 * prm.w_jerk;
    for (int m = 0; m < N; ++m) {
        Q.q[m] = 0.0;
        Q.u[m] = 0.0;
        for (int d = 0; d < 4; ++d) Q.P[m * 4 + d] = 0.0;
    }
    auto add = [&](int jp, int jq, double val) {        // symmetric entry, coefficient (cc) indices jp <= jq
        const int fp = jp - 3, fq = jq - 3;
        const bool pf = fp >= 0 && fp < N, qf = fq >= 0 && fq < N;
        if (pf && qf) Q.P[fp * 4 + (fq - fp)] += val;
        else if (pf) Q.q[fp] += val * cc[jq];
        else if (qf) Q.q[fq] += val * cc[jp];
    };
    const double lin = -2.0 * prm.w_v_ref * prm.v_ref;                        // ref :499-500
    for (int i = 0; i < n; ++i) {
        for (int p = 0; p < 3; ++p)
            for (int r = p; r < 3; ++r) add(i + p, i + r, wa2 * ba[p] * ba[r] + wv2 * bv[p] * bv[r]);
        if (i + 1 < n)
            for (int p = 0; p < 4; ++p)
                for (int r = p; r < 4; ++r) add(i + p, i + r, wj2 * jk[p] * jk[r]);
        for (int p = 0; p < 3; ++p) {
            const int f = i + p - 3;
            if (f >= 0 && f < N) Q.q[f] += lin * bv[p];
        }
    }
    return 0;
}

// ref :506-510 - per-station outputs, NaN behind the last station
EMP_HD void speed_qp_finish(const double* cc, int n, double dt, double* qs, double* qv, double* qa, double* qt) {
    const double nan = __builtin_nan("");
    for (int i = 0; i < kQp; ++i) {
        if (i < n) {
            qs[i] = (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0;
            qv[i] = (cc[i + 2] - cc[i]) / (2.0 * dt);
            qa[i] = (cc[i] - 2.0 * cc[i + 1] + cc[i + 2]) / (dt * dt);
            qt[i] = (double)i * dt;
        } else {
            qs[i] = qv[i] = qa[i] = qt[i] = nan;
        }
    }
}

// complete scalar speed QP on caller storage `mem` (speed_qp_words(17) doubles); returns status bits
EMP_HD int speed_qp_solve_scalar(double* mem, const double* dp_s, const double* dp_t, double v0, double a0,
                                 const double* s_lb, const double* s_ub, const double* sd_lb, const double* sd_ub,
                                 const SpeedQpParams& prm, double* qs, double* qv, double* qa, double* qt, int* iters) {
    *iters = 0;
    const double nan = __builtin_nan("");
    for (int i = 0; i < kQp; ++i) qs[i] = qv[i] = qa[i] = qt[i] = nan;
    const int n = speed_qp_size(dp_s);
    if (n < 0) return -n;
    const double dt = dp_t[n - 1] / (double)(n - 1);                          // ref :437, :449
    SpeedRangeQp Q;
    double* cc = mem;
    Q.bind(mem + n + 2, n - 1, n - 1);
    if (speed_qp_setup(Q, cc, n, dt, v0, a0, s_lb, s_ub, sd_lb, sd_ub, prm)) return kStbQpFailed;
    for (int m = 0; m < Q.N * 4; ++m) Q.M[m] = Q.P[m];
    if (!band_chol<3>(Q.M, Q.N)) return kStbQpFailed;
    for (int m = 0; m < Q.N; ++m) Q.u[m] = -Q.q[m];
    band_solve<3>(Q.M, Q.u, Q.N);
    const int rc = Q.solve_scalar();
    *iters = Q.iters;
    if (rc) return kStbQpFailed;
    for (int m = 0; m < Q.N; ++m) cc[m + 3] = Q.u[m];
    speed_qp_finish(cc, n, dt, qs, qv, qa, qt);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// increase_points (ref :514-566)
// ---------------------------------------------------------------------------------------------
// index of the last valid entry of relative_time_init [17] (ref :535-539); >= kQp means "no NaN" (IndexError)
EMP_HD int dense_t_end(const double* rel) {
    for (int i = 0; i < kQp; ++i)
        if (rel[i] != rel[i]) return i - 1;
    return kQp;
}
// the interval the reference's inner loop finds for current_t (ref :553-556), -1 if none
EMP_HD int dense_match(const double* rel, int t_end, double current_t) {
    for (int j = 0; j < t_end - 1; ++j)
        if (rel[j] <= current_t && current_t < rel[j + 1]) return j;
    return -1;
}
// sample i on interval tmp (ref :557-563); x ** 2 as x * x
EMP_HD void dense_sample(const double* s0, const double* v0, const double* a0, const double* rel, int tmp, double current_t,
                         double* s, double* v, double* a) {
    const double x = current_t - rel[tmp];
    const double x2 = x * x;
    *s = ((s0[tmp] + v0[tmp] * x) + ((1.0 / 3.0) * a0[tmp]) * x2) + ((1.0 / 6.0) * a0[tmp + 1]) * x2;
    *v = (v0[tmp] + (0.5 * a0[tmp]) * x) + (0.5 * a0[tmp + 1]) * x;
    *a = a0[tmp] + ((a0[tmp + 1] - a0[tmp]) * x) / (rel[tmp + 1] - rel[tmp]);
}

// ---------------------------------------------------------------------------------------------
// path_speed_merge (ref :569-620): numpy.interp on the path arrays
// ---------------------------------------------------------------------------------------------
// numpy's binary search: j with xp[j] <= x < xp[j+1]; -1 below, n above the last knot
EMP_HD int np_interp_index(const double* xp, int n, double x) {
    if (x < xp[0]) return -1;
    if (x > xp[n - 1]) return n;
    int lo = 0, hi = n;                      // first index with xp[idx] > x, minus one
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo - 1;
}
EMP_HD double np_interp_at(const double* xp, const double* fp, int n, int j, double x) {
    if (x != x) return x;
    if (j == -1) return fp[0];
    if (j >= n - 1) return fp[n - 1];
    if (xp[j] == x) return fp[j];
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double r = slope * (x - xp[j]) + fp[j];
    if (r != r) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (r != r && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}

}  // namespace stb
}  // namespace emp
