// emp_st_core.h - scalar arithmetic of the S-T speed DP (SURVEY.md section 8 row a-ST, BASELINE config 5),
// usable from HIP device code and from plain C++ (tests/host_check).
//
// Arithmetic contract: as in emp_core.h - written order, separately rounded binary64 operations,
// -ffp-contract=off.  oracle/st_speed.py (exact_*) states the same order in NumPy.  The only operation
// that is not correctly rounded is pow() in the 0.5..1.5 m band of the collision cost.
//
// "ref:" comments cite reference planner/speed_planning_test.py.
#pragma once

#include "emp_core.h"

namespace emp {
namespace st {

constexpr int kRows = 40;         // ref :114 (four np.arange pieces of ten samples each)
constexpr int kCols = 16;         // ref :116
constexpr int kStSamples = 5;     // ref :244
constexpr int kMaxObs = 64;       // obstacle slots per scene the kernels accept (prune mask is 64 bits)
constexpr double kPruneGap = 1.6; // > the 1.5 m reach of CalcCollisionCost, with room for rounding

// ref :114 - s_list[idx]; np.arange yields start + i*step, exact for these values
EMP_HD double s_list_at(int idx) {
    if (idx < 10) return 0.0 + (double)idx * 0.5;
    if (idx < 20) return 5.5 + (double)(idx - 10) * 1.0;
    if (idx < 30) return 16.0 + (double)(idx - 20) * 1.5;
    return 32.0 + (double)(idx - 30) * 2.5;
}
// ref :287-305 (CalcSTCoordinate) - row 0 carries the LARGEST s
EMP_HD double s_of_row(int row) { return s_list_at(kRows - row - 1); }
EMP_HD double t_of_col(int col) { return 0.5 + (double)col * 0.5; }

// ref :274-284 (CalcCollisionCost)
EMP_HD double collision_cost(double w, double d) {
    const double a = fabs(d);
    if (a < 0.5) return w;
    if (0.5 < a && a < 1.5) return pow(w, (0.5 - d) + 1.0);
    return 0.0;
}

// ref :258-269 - cost of one sample point (s, t) against one S-T obstacle segment
EMP_HD double point_cost(double w, double s, double t, double s_in, double t_in, double s_out, double t_out) {
    const double v1x = s_in - s, v1y = t_in - t;
    const double v2x = s_out - s, v2y = t_out - t;
    const double v3x = v2x - v1x, v3y = v2y - v1y;
    const double p = v1x * v3x + v1y * v3y;
    const double q = v2x * v3x + v2y * v3y;
    if ((p > 0.0 && q > 0.0) || (p < 0.0 && q < 0.0)) {
        const double d11 = v1x * v1x + v1y * v1y;
        const double d22 = v2x * v2x + v2y * v2y;
        const double m = d22 < d11 ? d22 : d11;
        if (m >= 2.25) return 0.0;  // sqrt is monotone and sqrt(2.25) == 1.5 exactly: d >= 1.5 costs nothing
        return collision_cost(w, sqrt(m));
    }
    const double cross = v1x * v3y - v1y * v3x;
    const double d33 = v3x * v3x + v3y * v3y;
    if (cross * cross > 2.2500001 * d33) return 0.0;  // d > 1.5 (1 + 2e-8): far outside rounding of the quotient
    return collision_cost(w, fabs(cross) / sqrt(d33));
}

// ref :234-271 (CalcObsCost).  Obstacles whose bounding box is farther than kPruneGap from the edge's
// sample span in s or in t contribute exactly 0 and are skipped; n_obs <= kMaxObs.
EMP_HD double obs_cost(double w, double s0, double t0, double s1, double t1, int n_obs, const double* s_in,
                       const double* s_out, const double* t_in, const double* t_out) {
    const double dt = (t1 - t0) / 4.0;
    const double k = (s1 - s0) / (t1 - t0);
    double ss[kStSamples], tt[kStSamples];
#pragma unroll
    for (int m = 0; m < kStSamples; ++m) {
        const double f = (double)(m - 1);  // ref :251-252: the first sample lies one step BEFORE the edge
        tt[m] = t0 + f * dt;
        ss[m] = s0 + (k * f) * dt;
    }
    const double s_lo = fmin(ss[0], ss[kStSamples - 1]), s_hi = fmax(ss[0], ss[kStSamples - 1]);
    const double t_lo = fmin(tt[0], tt[kStSamples - 1]), t_hi = fmax(tt[0], tt[kStSamples - 1]);
    uint64_t live = 0;
    for (int j = 0; j < n_obs; ++j) {
        if (isnan(s_in[j])) continue;  // ref :255
        const bool apart = fmin(s_in[j], s_out[j]) - s_hi >= kPruneGap || s_lo - fmax(s_in[j], s_out[j]) >= kPruneGap ||
                           fmin(t_in[j], t_out[j]) - t_hi >= kPruneGap || t_lo - fmax(t_in[j], t_out[j]) >= kPruneGap;
        if (!apart) live |= (uint64_t)1 << j;
    }
    double total = 0.0;
    if (live == 0) return total;
#pragma unroll
    for (int m = 0; m < kStSamples; ++m)
        for (int j = 0; j < n_obs; ++j)
            if ((live >> j) & 1) total = total + point_cost(w, ss[m], tt[m], s_in[j], t_in[j], s_out[j], t_out[j]);
    return total;
}

struct Weights {
    double v_ref, w_ref, w_acc, w_obs;  // ref :101-102 reference_speed, w_cost_ref_speed, w_cost_accel, w_cost_obs
};

// ref :217-226 - the state-dependent part of CalcDpCost (everything except the obstacle term)
EMP_HD void kinematic_cost(const Weights& w, double s0, double t0, double v0, double s1, double t1, double* acc,
                           double* ref) {
    const double v = (s1 - s0) / (t1 - t0);
    const double a = (v - v0) / (t1 - t0);
    const double e = v - w.v_ref;
    *ref = w.w_ref * (e * e);
    const double a2 = a * a;
    *acc = (4.0 > a && a > -6.0) ? w.w_acc * a2 : (100000.0 * w.w_acc) * a2;
}

// ref :191-231 (CalcDpCost) given the resolved start state
EMP_HD double edge_cost(const Weights& w, double s0, double t0, double v0, double s1, double t1, int n_obs,
                        const double* s_in, const double* s_out, const double* t_in, const double* t_out, double* obs_out) {
    double acc, ref;
    kinematic_cost(w, s0, t0, v0, s1, t1, &acc, &ref);
    const double obs = obs_cost(w.w_obs, s0, t0, s1, t1, n_obs, s_in, s_out, t_in, t_out);
    if (obs_out) *obs_out = obs;
    return (obs + acc) + ref;
}

// ref :38-98 (generate_st_graph) for one scene; arrays of n slots
EMP_HD void st_graph(int n, const double* obs_s, const double* obs_l, const double* obs_s_dot, const double* obs_l_dot,
                     double* s_in, double* s_out, double* t_in, double* t_out) {
    const double nan = NAN;
    bool alive = true;
    for (int i = 0; i < n; ++i) {
        s_in[i] = s_out[i] = t_in[i] = t_out[i] = nan;
        if (!alive) continue;
        if (isnan(obs_s[i])) {  // ref :51-52: the scan stops at the first empty slot
            alive = false;
            continue;
        }
        if (fabs(obs_l_dot[i]) < 0.3) continue;  // ref :53-66: slow lateral movers are ignored either way
        const double t_zero = -obs_l[i] / obs_l_dot[i];
        const double b1 = 2.0 / obs_l_dot[i] + t_zero;
        const double b2 = -2.0 / obs_l_dot[i] + t_zero;
        const double t_max = b1 > b2 ? b1 : b2;
        const double t_min = b1 > b2 ? b2 : b1;
        if (t_max < 1.0 || t_min > 8.0) continue;  // ref :79-83
        if (t_min < 0.0 && t_max > 0.0) {          // ref :84-90: already inside the +-2 m band
            s_in[i] = obs_s[i];
            t_in[i] = 0.0;
        } else {
            s_in[i] = obs_s[i] + obs_s_dot[i] * t_min;
            t_in[i] = t_min;
        }
        s_out[i] = obs_s[i] + obs_s_dot[i] * t_max;
        t_out[i] = t_max;
    }
}

// ref :158-172 - terminal node: right column top to bottom, then top row left to right, both with <=.
// cost is row-major [kRows][kCols].  Returns false when every candidate is NaN.
template <class CostAt>
EMP_HD bool terminal_node(CostAt cost, int* row, int* col) {
    double best = INFINITY;
    int r = -1, c = -1;
    for (int i = 0; i < kRows; ++i)
        if (cost(i, kCols - 1) <= best) {
            best = cost(i, kCols - 1);
            r = i;
            c = kCols - 1;
        }
    for (int j = 0; j < kCols; ++j)
        if (cost(0, j) <= best) {
            best = cost(0, j);
            r = 0;
            c = j;
        }
    *row = r;
    *col = c;
    return r >= 0;
}

}  // namespace st
}  // namespace emp
