// emp_core.h - scalar building blocks of the EM-Planner hot path, usable from HIP device code and
// (for the host-side logic checks in tests/) from plain C++.
//
// Arithmetic contract: every expression in the "DP" section is evaluated in the written order with
// separately rounded IEEE-754 binary64 operations; the translation unit MUST be compiled with
// -ffp-contract=off (build.py does).  oracle/exact.py states the same operation order in NumPy and
// the GPU results are compared with it bit for bit.
//
// "ref:" comments cite the reference implementation (paths relative to the reference tree).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define EMP_HD __host__ __device__ __forceinline__
#else
#define EMP_HD inline
#endif

// Development hooks (emp_qp_params.reserved read as the QP kernels' debug stage): compiled in only with -DEMP_DEV_HOOKS=1.
#ifndef EMP_DEV_HOOKS
#define EMP_DEV_HOOKS 0
#endif
// Wavefront priorities (s_setprio, 0..3) of the kernels that share the SIMDs when two batches are in flight.
#ifndef EMP_PRIO_BACK
#define EMP_PRIO_BACK 3      // path QP, Cartesian tail
#endif
#ifndef EMP_PRIO_CART
#define EMP_PRIO_CART EMP_PRIO_BACK   // Cartesian tail (what the sweep of the next batch runs beside)
#endif
#ifndef EMP_PRIO_FRONT
#define EMP_PRIO_FRONT 0     // projection, edge costs
#endif
#ifndef EMP_PRIO_SWEEP
#define EMP_PRIO_SWEEP 3     // min-plus sweep
#endif

namespace emp {

constexpr int kSamples = 10;          // ref: path_planning.py:486-493 (10 samples per lattice edge)
constexpr double kDanger2 = 16.0;     // ref: cal_obs_cost danger_dis=4  (path_planning.py:588,602)
constexpr double kSafe2 = 36.0;       //      safe_dis=6                 (:606)
constexpr double kSoftGain = 5000.0;  //      5000 / d^2                 (:608)
constexpr double kLanePenalty = 10000.0;  // ref: path_planning.py:317-318, 341-342

// ---------------------------------------------------------------------------------------------
// DP section (bit-exact contract)
// ---------------------------------------------------------------------------------------------

// ref: path_planning.py:326 / :478 - lateral offset of lattice row i
EMP_HD double lattice_l(int row, int i, double sample_l) {
    return ((double)(row + 1) / 2.0 - 1.0 - (double)i) * sample_l;
}

// lattice_l for a fractional row (the no-obstacle bypass yields (row+1)/2-1, ref :363/:370)
EMP_HD double lattice_l_f(int row, double r, double sample_l) {
    return ((double)(row + 1) / 2.0 - 1.0 - r) * sample_l;
}

struct Quintic {
    double a0, a1, a2, a3, a4, a5;  // l(t) = sum a_k t^k, t = s - s0
};

// Closed-form quintic through (l0, dl0, ddl0) at t=0 and (l1, 0, 0) at t=T.
// Same mathematics as ref planning_utils.py:671-703 (which inverts a 6x6 matrix in absolute s).
EMP_HD Quintic quintic_shifted(double l0, double dl0, double ddl0, double l1, double T) {
    Quintic q;
    const double h = l1 - l0;
    const double T2 = T * T;
    const double T3 = T2 * T;
    const double T4 = T3 * T;
    const double T5 = T4 * T;
    q.a0 = l0;
    q.a1 = dl0;
    q.a2 = 0.5 * ddl0;
    q.a3 = ((20.0 * h - (12.0 * dl0) * T) - (3.0 * ddl0) * T2) / (2.0 * T3);
    q.a4 = ((-30.0 * h + (16.0 * dl0) * T) + (3.0 * ddl0) * T2) / (2.0 * T4);
    q.a5 = ((12.0 * h - (6.0 * dl0) * T) - ddl0 * T2) / (2.0 * T5);
    return q;
}

EMP_HD double quintic_l(const Quintic& q, double t) {
    double p = q.a5;
    p = q.a4 + t * p;
    p = q.a3 + t * p;
    p = q.a2 + t * p;
    p = q.a1 + t * p;
    p = q.a0 + t * p;
    return p;
}

EMP_HD double quintic_dl(const Quintic& q, double t) {
    double p = 5.0 * q.a5;
    p = 4.0 * q.a4 + t * p;
    p = 3.0 * q.a3 + t * p;
    p = 2.0 * q.a2 + t * p;
    p = q.a1 + t * p;
    return p;
}

EMP_HD double quintic_ddl(const Quintic& q, double t) {
    double p = 20.0 * q.a5;
    p = 12.0 * q.a4 + t * p;
    p = 6.0 * q.a3 + t * p;
    p = 2.0 * q.a2 + t * p;
    return p;
}

// sample abscissa i of an edge: ref path_planning.py:493/:566  s = start_s + i * sample_s / 10
EMP_HD double sample_t(int i, double sample_s) { return ((double)i * sample_s) / 10.0; }

// The reference's third-derivative term is 6 c3 + 24 c4 s + 60 c5 (s * 2) with c_k the ABSOLUTE-s
// coefficients (quirk, path_planning.py:498/:571).  Rebuild c3, c4, c5 from the shifted ones.
struct JerkQuirk {
    double k0, k1, k2x2;  // 6 c3, 24 c4, 2 * (60 c5): doubling is exact, so k2x2 * s == (60 c5) * (s * 2) bit for bit
};
EMP_HD JerkQuirk jerk_quirk(const Quintic& q, double s0) {
    const double c5 = q.a5;
    const double c4 = q.a4 - (5.0 * q.a5) * s0;
    const double c3 = (q.a3 - (4.0 * q.a4) * s0) + ((10.0 * q.a5) * s0) * s0;
    JerkQuirk j;
    j.k0 = 6.0 * c3;
    j.k1 = 24.0 * c4;
    j.k2x2 = (60.0 * c5) * 2.0;
    return j;
}
// The sum of the squared quirk term over the kSamples samples s_n = s0 + t_n, in closed form.  The term is LINEAR in s,
//   6 c3 + 24 c4 s + 60 c5 (2 s) = A + K1 t_n   with  K1 = 24 c4 + 120 c5,  A = 6 c3 + K1 s0,
// so  sum_n (A + K1 t_n)^2 = kSamples A^2 + 2 A K1 T1 + K1^2 T2  with the lattice constants T1 = sum t_n, T2 = sum t_n^2:
// about twenty operations per edge where the sample loop (abscissa, two products, two sums, square, accumulate) took
// eighty - a quarter of the edge-cost kernel's instructions.  Same mathematics as the reference's loop (:492-499,
// :565-572), rounded differently in the last bits (the reference's own values carry ~1e-6 of noise from its 6x6 inverse).
// Operation order as written; oracle/exact.py states the same.
constexpr int kSampleMoments = 2;      // T1, T2 stored behind the kSamples sample offsets
EMP_HD void sample_moments(double sample_s, double* T1, double* T2) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kSamples; ++i) {
        const double t = sample_t(i, sample_s);
        a = a + t;
        b = b + t * t;
    }
    *T1 = a;
    *T2 = b;
}
EMP_HD double jerk_quirk_sum(const Quintic& q, double s0, double T1, double T2) {
    const JerkQuirk j = jerk_quirk(q, s0);
    const double K1 = j.k1 + j.k2x2;
    const double A = j.k0 + K1 * s0;
    return ((double)kSamples * (A * A) + (2.0 * A) * (K1 * T1)) + (K1 * K1) * T2;
}

// ref: cal_obs_cost (path_planning.py:588-609) driven by the caller's d^2 loop (:503-509 / :577-583):
// ordered scan of the 10 samples of ONE obstacle with the early break on the first hard hit.
// l_samples[i] are the edge's lateral samples, s0 its start abscissa.
EMP_HD double obstacle_cost(const double* l_samples, double s0, double sample_s, double obs_s, double obs_l,
                            double w_collision) {
    double c = 0.0;
    for (int i = 0; i < kSamples; ++i) {
        const double s = s0 + sample_t(i, sample_s);
        const double d_lon = obs_s - s;
        const double d_lat = obs_l - l_samples[i];
        const double d2 = d_lon * d_lon + d_lat * d_lat;
        if (d2 <= kDanger2) {
            c = c + w_collision;
            break;
        } else if (d2 < kSafe2) {  // d2 > 16 already known
            c = c + kSoftGain / d2;
        }
    }
    return c;
}

// Conservative reach test: an obstacle contributes exactly 0 unless some sample is closer than
// 6 m, which needs |obs_s - s_i| < 6 and |obs_l - l_i| < 6.  The 0.5 m margin dwarfs every rounding
// error of the bounds, so skipping on this test never changes a result.
EMP_HD bool obstacle_in_reach(double obs_s, double obs_l, double s_first, double s_last, double l_lo,
                              double l_hi) {
    return (obs_s > s_first - 6.5) && (obs_s < s_last + 6.5) && (obs_l > l_lo - 6.5) && (obs_l < l_hi + 6.5);
}

// Full cost of one lattice edge (generic form; ref: cal_start_cost :435-514, cal_neighbor_cost :517-585).
// The kernels use a tabulated form for the neighbour edges that performs the same operations.
EMP_HD double segment_cost(const Quintic& q, double s0, double sample_s, const double* obs_s,
                           const double* obs_l, int n_obs, double w_collision, double w0, double w1, double w2,
                           double w_ref) {
    double T1, T2;
    sample_moments(sample_s, &T1, &T2);
    const double S_d3 = jerk_quirk_sum(q, s0, T1, T2);
    double l_s[kSamples];
    double S_l = 0.0, S_dl = 0.0, S_ddl = 0.0;
    for (int i = 0; i < kSamples; ++i) {
        const double t = sample_t(i, sample_s);
        const double l = quintic_l(q, t);
        const double dl = quintic_dl(q, t);
        const double ddl = quintic_ddl(q, t);
        l_s[i] = l;
        S_l = S_l + l * l;
        S_dl = S_dl + dl * dl;
        S_ddl = S_ddl + ddl * ddl;
    }
    double coll = 0.0;
    for (int m = 0; m < n_obs; ++m) coll = coll + obstacle_cost(l_s, s0, sample_s, obs_s[m], obs_l[m], w_collision);
    const double smooth = (w0 * S_dl + w1 * S_ddl) + w2 * S_d3;
    return (smooth + coll) + w_ref * S_l;
}

// The quirked jerk sum of a NEIGHBOUR edge (dl0 = ddl0 = 0), factorised (round 5): the edge's shifted coefficients are h
// times the unit quintic's, h = l1 - l0, the reference's third-derivative term (path_planning.py:571) is linear in them, so
// its sum of squares over the samples is (h h) F(s0) with F the unit quintic's sum - evaluated once per (scene, column) by
// the kernels (emp_dp_kernels.h jerk_unit_sum: the same operations on tabulated u3, u4, u5, T1, T2).
EMP_HD double neighbour_jerk_factor(double sample_s, double s0) {
    double T1, T2;
    sample_moments(sample_s, &T1, &T2);
    const Quintic u = quintic_shifted(0.0, 0.0, 0.0, 1.0, sample_s);
    return jerk_quirk_sum(u, s0, T1, T2);
}
// Full cost of a neighbour edge from (s0, l0) to (s0 + sample_s, l1) in the generic (untabulated) form: what the tiled, fused
// and wide edge kernels compute from the pair table (ref: cal_neighbor_cost, path_planning.py:517-585).
EMP_HD double neighbour_cost(double l0, double l1, double s0, double sample_s, const double* obs_s, const double* obs_l,
                             int n_obs, double w_collision, double w0, double w1, double w2, double w_ref) {
    const Quintic q = quintic_shifted(l0, 0.0, 0.0, l1, sample_s);
    double l_s[kSamples];
    double S_l = 0.0, S_dl = 0.0, S_ddl = 0.0;
    for (int i = 0; i < kSamples; ++i) {
        const double t = sample_t(i, sample_s);
        const double l = quintic_l(q, t);
        const double dl = quintic_dl(q, t);
        const double ddl = quintic_ddl(q, t);
        l_s[i] = l;
        S_l = S_l + l * l;
        S_dl = S_dl + dl * dl;
        S_ddl = S_ddl + ddl * ddl;
    }
    double coll = 0.0;
    for (int m = 0; m < n_obs; ++m) coll = coll + obstacle_cost(l_s, s0, sample_s, obs_s[m], obs_l[m], w_collision);
    const double h = l1 - l0;
    const double smooth = (w0 * S_dl + w1 * S_ddl) + (w2 * (h * h)) * neighbour_jerk_factor(sample_s, s0);
    return (smooth + coll) + w_ref * S_l;
}

// number of samples numpy.arange(0, int(span), res) yields (ref: path_planning.py:405/:423)
EMP_HD int arange_count(double span, double res) {
    const double top = (double)(long long)span;  // int(): truncation toward zero
    if (!(top > 0.0) || !(res > 0.0)) return 0;
    return (int)ceil(top / res);
}

}  // namespace emp
