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
constexpr double kReachGap = 1.51;  // the same for st::reach_interval: its few operations round at 1e-13 of the values

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

EMP_HD int ctz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)v) - 1;
#else
    return __builtin_ctzll(v);
#endif
}

// x / dt.  Every edge between two grid columns has dt == 0.5 (ref :116), where the quotient is the exact
// power-of-two scaling x * 2; only edges that start at the DP origin need a real division.
EMP_HD double div_dt(double x, double dt) { return dt == 0.5 ? x * 2.0 : x / dt; }

#if defined(__HIPCC__)
#define EMP_ST_COLD __host__ __device__ __noinline__
#else
#define EMP_ST_COLD __attribute__((noinline))
#endif

// w ** y for the collision cost's base (ref :281: w_cost_obs ** ((0.5 - min_dis) + 1)).  The base is fixed per call,
// so log2(w) is computed once on the host as a double-double and the kernels evaluate exp2(y * log2 w): the product
// exactly (fma), exp2 of its head, first-order correction for its tail - within 2 ulp of the true power, against
// < 1 ulp for libm's pow, at a fifth of the instructions of the general pow() expansion (which was 30 % of the speed
// DP kernel).  A base that is not a positive finite number falls back to pow() and its special cases.
struct PowBase {
    double w, lg_hi, lg_lo;
};
inline PowBase make_pow_base(double w) {        // host side (the library's launchers, tests/host_check)
    PowBase b{w, NAN, 0.0};
    if (w > 0.0 && w < INFINITY) {
        const long double L = log2l((long double)w);
        b.lg_hi = (double)L;
        b.lg_lo = (double)(L - (long double)b.lg_hi);
    }
    return b;
}
EMP_HD double pow_base(const PowBase& b, double y) {
    if (!(b.lg_hi == b.lg_hi)) return pow(b.w, y);
    const double p = y * b.lg_hi;
    const double tail = fma(y, b.lg_hi, -p) + y * b.lg_lo;
    const double r = exp2(p);
    return fma(r, 0.6931471805599453 * tail, r);
}

// pow_base for the compacted pair lists of the speed DP kernel: no call into the general pow() (whose expansion
// alone needs ~140 vector registers).  The exponent is in (0, 1) wherever the result is used (0.5 < d < 1.5), and
// there w ** y of a base that is 0, +inf or NaN is the base itself; a negative base (complex in the reference, which
// then fails on its next comparison) is refused by emp_speed_dp.
EMP_HD double pow_base_flat(const PowBase& b, double y) {
    const double p = y * b.lg_hi;
    const double tail = fma(y, b.lg_hi, -p) + y * b.lg_lo;
    const double r = exp2(p);
    const double v = fma(r, 0.6931471805599453 * tail, r);
    return b.lg_hi == b.lg_hi ? v : b.w;
}

// ref :274-284 (CalcCollisionCost)
EMP_HD double collision_cost(const PowBase& w, double d) {
    const double a = fabs(d);
    if (a < 0.5) return w.w;
    if (0.5 < a && a < 1.5) return pow_base(w, (0.5 - d) + 1.0);
    return 0.0;
}

// ref :258-269 - cost of one sample point (s, t) against one S-T obstacle segment: the reference's arithmetic
EMP_HD double point_cost(const PowBase& w, double s, double t, double s_in, double t_in, double s_out, double t_out) {
    const double v1x = s_in - s, v1y = t_in - t;
    const double v2x = s_out - s, v2y = t_out - t;
    const double v3x = v2x - v1x, v3y = v2y - v1y;
    const double p = v1x * v3x + v1y * v3y;
    const double q = v2x * v3x + v2y * v3y;
    double d;
    if ((p > 0.0 && q > 0.0) || (p < 0.0 && q < 0.0)) {  // the foot of the perpendicular misses the segment
        const double d11 = v1x * v1x + v1y * v1y;
        const double d22 = v2x * v2x + v2y * v2y;
        d = sqrt(d22 < d11 ? d22 : d11);  // min(dis1, dis2): sqrt is monotone, so min and sqrt commute
    } else {
        d = fabs(v1x * v3y - v1y * v3x) / sqrt(v3x * v3x + v3y * v3y);
    }
    return collision_cost(w, d);
}

// Out-of-line copy for the kernels: samples that are near an obstacle are rare per lane but not per
// wavefront, so they are gathered first and costed together (obs_cost below); one copy of the long pow()
// expansion per kernel.
EMP_ST_COLD static double point_cost_cold(PowBase w, double s, double t, double s_in, double t_in, double s_out,
                                          double t_out) {
    return point_cost(w, s, t, s_in, t_in, s_out, t_out);
}

// True only if point_cost is exactly 0, decided with the reference's own intermediate values and no sqrt /
// division: d >= 1.5 costs nothing (ref :281-282).  sqrt(2.25) == 1.5 exactly and sqrt is monotone; for the
// perpendicular distance the test leaves a 2e-8 relative margin, far outside the rounding of the quotient.
EMP_HD bool point_is_far(double s, double t, double s_in, double t_in, double s_out, double t_out) {
    const double v1x = s_in - s, v1y = t_in - t;
    const double v2x = s_out - s, v2y = t_out - t;
    const double v3x = v2x - v1x, v3y = v2y - v1y;
    const double p = v1x * v3x + v1y * v3y;
    const double q = v2x * v3x + v2y * v3y;
    const bool outside = (p > 0.0 && q > 0.0) || (p < 0.0 && q < 0.0);
    const double d11 = v1x * v1x + v1y * v1y;
    const double d22 = v2x * v2x + v2y * v2y;
    const double m = d22 < d11 ? d22 : d11;
    const double cross = v1x * v3y - v1y * v3x;
    const double d33 = v3x * v3x + v3y * v3y;
    return outside ? m >= 2.25 : cross * cross > 2.2500001 * d33;
}

// point_cost without branches (the speed DP kernel runs it on compacted lists of (sample, obstacle) pairs, where the
// lanes of a wavefront take both sides of every test): one sqrt and one division for every pair, the selects pick
// what point_cost computes on its side of the branch - the same operations on the same operands, bit for bit.
EMP_HD double point_cost_flat(const PowBase& w, double s, double t, double s_in, double t_in, double s_out, double t_out) {
    const double v1x = s_in - s, v1y = t_in - t;
    const double v2x = s_out - s, v2y = t_out - t;
    const double v3x = v2x - v1x, v3y = v2y - v1y;
    const double p = v1x * v3x + v1y * v3y;
    const double q = v2x * v3x + v2y * v3y;
    const bool outside = (p > 0.0 && q > 0.0) || (p < 0.0 && q < 0.0);
    const double d11 = v1x * v1x + v1y * v1y;
    const double d22 = v2x * v2x + v2y * v2y;
    const double d33 = v3x * v3x + v3y * v3y;
    const double cross = v1x * v3y - v1y * v3x;
    const double r = sqrt(outside ? (d22 < d11 ? d22 : d11) : d33);
    const double quot = fabs(cross) / r;
    const double d = outside ? r : quot;
    const double a = fabs(d);
    const double mid = pow_base_flat(w, (0.5 - d) + 1.0);
    return a < 0.5 ? w.w : ((0.5 < a && a < 1.5) ? mid : 0.0);
}

// Reach of one obstacle segment at a fixed time t: an interval (lo, hi) of s outside which point_cost is exactly 0.
// The points within kReachGap of the segment lie in the rectangle |n| < G, -G < l < len + G of the segment's own
// frame (n across, l along; G = kReachGap = 1.51 > the 1.5 reach of CalcCollisionCost, ref :281-282); at a fixed t
// both conditions are intervals of s.  The 0.01 margin is ten orders of magnitude above the rounding of these few
// operations.  A degenerate segment (NaN frame) or a non-finite bound keeps everything: (-inf, +inf).
// Returns false when the interval is empty.
EMP_HD bool reach_interval(double t, double s_in, double t_in, double ux, double uy, double len, double* lo, double* hi) {
    const double G = kReachGap, inf = INFINITY;
    const double a = (t - t_in) * ux, b = (t - t_in) * uy, top = len + G;
    double l1 = -inf, h1 = inf, l2 = -inf, h2 = inf;
    bool empty = false;
    if (!(ux == ux) || !(uy == uy) || !(top == top)) {
        *lo = -inf;
        *hi = inf;
        return true;
    }
    if (uy != 0.0) {                       // |a - x uy| < G,  x = s - s_in
        const double x1 = (a - G) / uy, x2 = (a + G) / uy;
        l1 = fmin(x1, x2);
        h1 = fmax(x1, x2);
    } else if (!(fabs(a) < G)) {
        empty = true;
    }
    if (ux != 0.0) {                       // -G < x ux + b < top
        const double x1 = (-G - b) / ux, x2 = (top - b) / ux;
        l2 = fmin(x1, x2);
        h2 = fmax(x1, x2);
    } else if (!(-G < b && b < top)) {
        empty = true;
    }
    const double l = s_in + fmax(l1, l2), h = s_in + fmin(h1, h2);
    if (!(l == l) || !(h == h)) {          // inf - inf and the like: keep everything
        *lo = -inf;
        *hi = inf;
        return true;
    }
    *lo = empty ? inf : l;
    *hi = empty ? -inf : h;
    return !empty && l < h;
}

// Obstacle segments of one scene plus, per segment, its unit direction (ux, uy) in the (s, t) plane and its
// length: only used to REJECT obstacles / samples that are provably farther than kPruneGap (distance to a
// segment >= distance to its line, and >= the overshoot along it), never to compute a cost.  A degenerate
// segment has NaN direction, every rejection test is then false and the exact path runs.
struct ObsSet {
    int n;
    const double *s_in, *s_out, *t_in, *t_out, *ux, *uy, *len;
};

EMP_HD void obs_frame(double s_in, double t_in, double s_out, double t_out, double* ux, double* uy, double* len) {
    const double dx = s_out - s_in, dy = t_out - t_in;
    const double L = sqrt(dx * dx + dy * dy);
    *len = L;
    *ux = dx / L;
    *uy = dy / L;
}

// ref :234-271 (CalcObsCost).  Obstacles / samples that are provably at least kPruneGap (> 1.5) away
// contribute exactly 0 and are skipped; everything else goes through the reference's arithmetic.
EMP_HD double obs_cost(const PowBase& w, double s0, double t0, double s1, double t1, const ObsSet& o) {
    const double dt = (t1 - t0) * 0.25;  // == (t1 - t0) / (n - 1), n = 5 (ref :244-246)
    const double k = div_dt(s1 - s0, t1 - t0);
    // ref :251-252: sample m sits at t0 + (m-1) dt, s0 + (k (m-1)) dt - the first one lies BEFORE the edge
    const double s_a = s0 + (k * -1.0) * dt, s_b = s0 + (k * 3.0) * dt;
    const double t_a = t0 + -1.0 * dt, t_b = t0 + 3.0 * dt;
    const double s_lo = fmin(s_a, s_b), s_hi = fmax(s_a, s_b);
    const double t_lo = fmin(t_a, t_b), t_hi = fmax(t_a, t_b);
    const double G = kPruneGap;
    // One pass over the obstacles decides, per sample, which of them can be within reach.  The five samples lie on
    // the straight edge, so their coordinates in a segment's frame are linear in the sample index: the two end
    // samples' frame coordinates (needed for the edge-level rejection anyway) give every sample's by interpolation.
    // The tests carry a 0.1 margin over the 1.5 reach, ten orders of magnitude above the rounding of that
    // interpolation.  cand[m] bit j: sample m is inside obstacle j's 1.6-wide band and not beyond its ends.
    uint64_t cand[kStSamples];
#pragma unroll
    for (int m = 0; m < kStSamples; ++m) cand[m] = 0;
    bool any = false;
    for (int j = 0; j < o.n; ++j) {
        const double si = o.s_in[j], so = o.s_out[j], ti = o.t_in[j], to = o.t_out[j];
        if (isnan(si)) continue;  // ref :255
        // axis-aligned boxes of the sample span and of the segment
        bool apart = fmin(si, so) - s_hi >= G || s_lo - fmax(si, so) >= G || fmin(ti, to) - t_hi >= G || t_lo - fmax(ti, to) >= G;
        // the same in the segment's own frame: both end samples on one side of the 1.6-wide band / beyond one end
        const double ux = o.ux[j], uy = o.uy[j], top = o.len[j] + G;
        const double ax = s_a - si, ay = t_a - ti, bx = s_b - si, by = t_b - ti;
        const double na = ay * ux - ax * uy, nb = by * ux - bx * uy;
        const double la = ax * ux + ay * uy, lb = bx * ux + by * uy;
        apart = apart || (na >= G && nb >= G) || (na <= -G && nb <= -G) || (la <= -G && lb <= -G) || (la >= top && lb >= top);
        if (apart) continue;
        const double dn = nb - na, dl = lb - la;
#pragma unroll
        for (int m = 0; m < kStSamples; ++m) {
            const double w = 0.25 * (double)m;
            const double nn = na + dn * w, ll = la + dl * w;
            if (!(fabs(nn) >= G || ll <= -G || ll >= top)) {       // a NaN frame (degenerate segment) keeps the pair
                cand[m] |= (uint64_t)1 << j;
                any = true;
            }
        }
    }
    double total = 0.0;
    if (!any) return total;
    // pass 1: which of the candidate (sample, obstacle) pairs are near?  The reference's own intermediates, no sqrt.
    uint64_t near[kStSamples];
#pragma unroll
    for (int m = 0; m < kStSamples; ++m) {
        const double f = (double)(m - 1);
        const double t = t0 + f * dt;
        const double s = s0 + (k * f) * dt;
        uint64_t hit = 0;
        for (uint64_t rest = cand[m]; rest; rest &= rest - 1) {
            const int j = ctz64(rest);
            if (!point_is_far(s, t, o.s_in[j], o.t_in[j], o.s_out[j], o.t_out[j])) hit |= (uint64_t)1 << j;
        }
        near[m] = hit;
    }
    // pass 2: the near pairs in the reference's order (sample outer, obstacle inner; ref :249-269).  Each
    // lane pops its next pair, so a wavefront runs the expensive path max-over-lanes(#near) times.
    for (;;) {
        int m = -1;
        uint64_t word = 0;
#pragma unroll
        for (int i = kStSamples - 1; i >= 0; --i)
            if (near[i]) {
                m = i;
                word = near[i];
            }
        if (m < 0) break;
        const int j = ctz64(word);
#pragma unroll
        for (int i = 0; i < kStSamples; ++i)
            if (i == m) near[i] = word & (word - 1);
        const double f = (double)(m - 1);
        total = total + point_cost_cold(w, s0 + (k * f) * dt, t0 + f * dt, o.s_in[j], o.t_in[j], o.s_out[j], o.t_out[j]);
    }
    return total;  // adding the exact zeros of the skipped pairs would not change any bit
}

struct Weights {
    double v_ref, w_ref, w_acc;  // ref :101-102 reference_speed, w_cost_ref_speed, w_cost_accel
    PowBase w_obs;               // w_cost_obs
};

// ref :217-226 - the state-dependent part of CalcDpCost (everything except the obstacle term)
EMP_HD void kinematic_cost(const Weights& w, double s0, double t0, double v0, double s1, double t1, double* acc,
                           double* ref) {
    const double v = div_dt(s1 - s0, t1 - t0);
    const double a = div_dt(v - v0, t1 - t0);
    const double e = v - w.v_ref;
    *ref = w.w_ref * (e * e);
    const double a2 = a * a;
    *acc = (4.0 > a && a > -6.0) ? w.w_acc * a2 : (100000.0 * w.w_acc) * a2;
}

// ref :191-231 (CalcDpCost) given the resolved start state
EMP_HD double edge_cost(const Weights& w, double s0, double t0, double v0, double s1, double t1, const ObsSet& o,
                        double* obs_out) {
    double acc, ref;
    kinematic_cost(w, s0, t0, v0, s1, t1, &acc, &ref);
    const double obs = obs_cost(w.w_obs, s0, t0, s1, t1, o);
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
