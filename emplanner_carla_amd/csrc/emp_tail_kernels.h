// emp_tail_kernels.h - everything of one planning cycle around the DP: Cartesian->Frenet projection,
// QP bounds, path QP, midpoint re-interleave, Frenet->Cartesian, smoothing QP, heading/curvature.
//
// Mapping: the kernels of emp_plan_cycle run one scene per wavefront (or per half wavefront) with the reference
// line, the path and the QP state in LDS / registers; the stand-alone per-function kernels of the utilities keep
// the simple one-scene-per-lane form.  "ref:" cites the reference (paths relative to the reference tree).
#pragma once

#include <hip/hip_runtime.h>

#include "emp_core.h"
#include "emp_frenet_core.h"
#include "emp_qp_core.h"
#include "emp_qp_wave.h"
#include "emp_qp_rows.h"
#include "emp_smooth_rows.h"

namespace emp {

// status bits (mirror include/emplanner.h)
constexpr int kStDpInfeasible = 1, kStSOutOfRange = 2, kStBoundIndex = 4, kStQpFailed = 8, kStSmoothFailed = 16,
              kStTruncated = 32;

struct QpDev {
    PathQpParams qp;
    double obs_length, obs_width;
    int decimate, midpoint, use_qp;
    int debug_stage;   // development only: cut the QP kernel short after stage N (0 = run everything)
};

// ---------------------------------------------------------------------------------------------
// ref: test_9.py:113-177 - s_map, obstacle (s, l), planning-start (s, l) and (l, dl/ds, d2l/ds2)
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Wave-parallel form of the cycle front (one wavefront per scene, one reference-line node per lane).
// The reference's nearest-node scan (planning_utils.py:383-402) is an ordered scan with an early exit after
// 50 consecutive non-improvements; in parallel that is: L_i = first index achieving min(d_0..d_i) (a prefix
// minimum that keeps the FIRST occurrence, because only a strict '<' counts as an improvement), stop at the first
// i with i - L_i >= 50, answer L_i there (or L_{P-1} if the scan never stops early).
// ---------------------------------------------------------------------------------------------
// Inclusive prefix minimum of (d, index) over the 64 lanes, the EARLIER lane winning ties, by DPP moves (no LDS
// traffic, no address arithmetic): Hillis-Steele inside each row of 16 lanes (row_shr:1, 2, 4, 8; a lane without a
// source keeps +inf), then the row totals travel on with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into
// rows 2 and 3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void prefix_min_step(double& d, int& li) {
    union { double f; int w[2]; } a, r, inf;
    a.f = d;
    inf.f = __builtin_inf();
    r.w[0] = __builtin_amdgcn_update_dpp(inf.w[0], a.w[0], CTRL, ROW_MASK, 0xF, false);
    r.w[1] = __builtin_amdgcn_update_dpp(inf.w[1], a.w[1], CTRL, ROW_MASK, 0xF, false);
    const int oi = __builtin_amdgcn_update_dpp(0, li, CTRL, ROW_MASK, 0xF, false);
    if (!(d < r.f)) {                  // the value from the earlier lanes wins unless mine is strictly smaller;
        d = r.f;                       // a lane without a source sees +inf and keeps its own unless that is +inf / NaN
        li = oi;
    }
}
__device__ __forceinline__ void prefix_min_first(double& d, int& li) {
    prefix_min_step<0x111, 0xF>(d, li);   // row_shr:1
    prefix_min_step<0x112, 0xF>(d, li);   // row_shr:2
    prefix_min_step<0x114, 0xF>(d, li);   // row_shr:4
    prefix_min_step<0x118, 0xF>(d, li);   // row_shr:8
    prefix_min_step<0x142, 0xA>(d, li);   // row_bcast:15 -> rows 1, 3
    prefix_min_step<0x143, 0xC>(d, li);   // row_bcast:31 -> rows 2, 3
}

// One step of a value-only inclusive prefix minimum (lanes without a source, or outside the row mask, see +inf)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double prefix_fmin_step(double v) {
    union { double f; int w[2]; } a, r, inf;
    a.f = v;
    inf.f = __builtin_inf();
    r.w[0] = __builtin_amdgcn_update_dpp(inf.w[0], a.w[0], CTRL, ROW_MASK, 0xF, false);
    r.w[1] = __builtin_amdgcn_update_dpp(inf.w[1], a.w[1], CTRL, ROW_MASK, 0xF, false);
    return __builtin_fmin(v, r.f);
}

// The scan for lines of at most 64 nodes (one node per lane; the cycle's 51- and 61-node lines), without carrying the index
// through the prefix: with pm_i = min(d_0..d_i), the first occurrence L_i of pm_i satisfies i - L_i >= limit exactly when
// pm_i == pm_{i-limit} (no strict improvement among the last `limit` nodes), so the stopping node i* is the first lane where
// that holds and the answer is the first lane j <= i* with d_j == pm_{i*} - or, if the scan never stops, the first lane that
// attains the overall minimum.  Half the instructions of the indexed prefix (round 3).
__device__ inline int match_scan_wave64(const double* lx, const double* ly, int P, double x, double y, int limit) {
    const int lane = threadIdx.x & 63;
    double d = __builtin_inf();
    if (lane < P) {
        const double dx = lx[lane] - x, dy = ly[lane] - y;
        d = sqrt(dx * dx + dy * dy);
    }
    double pm = d;
    pm = prefix_fmin_step<0x111, 0xF>(pm);   // row_shr:1
    pm = prefix_fmin_step<0x112, 0xF>(pm);   // row_shr:2
    pm = prefix_fmin_step<0x114, 0xF>(pm);   // row_shr:4
    pm = prefix_fmin_step<0x118, 0xF>(pm);   // row_shr:8
    pm = prefix_fmin_step<0x142, 0xA>(pm);   // row_bcast:15 -> rows 1, 3
    pm = prefix_fmin_step<0x143, 0xC>(pm);   // row_bcast:31 -> rows 2, 3
    const double back = __shfl(pm, lane >= limit ? lane - limit : 0, 64);
    const unsigned long long stop = __ballot(lane < P && lane >= limit && pm == back);
    const int last = stop ? __builtin_ffsll((long long)stop) - 1 : P - 1;          // the node at which the scan ends
    const double target = __shfl(pm, last, 64);
    const unsigned long long hit = __ballot(lane <= last && d == target);
    return hit ? __builtin_ffsll((long long)hit) - 1 : 0;                           // (no finite distance at all: node 0)
}

__device__ inline int match_scan_wave(const double* lx, const double* ly, int P, double x, double y, int limit) {
    if (P <= 64 && limit >= 1) return match_scan_wave64(lx, ly, P, x, y, limit);   // uniform
    const int lane = threadIdx.x & 63;
    double cd = __builtin_inf();     // carry: prefix minimum and its first index over the previous chunks
    int ci = 0;
    for (int base = 0; base < P; base += 64) {
        const int i = base + lane;
        double d = __builtin_inf();
        int li = i;
        if (i < P) {
            const double dx = lx[i] - x, dy = ly[i] - y;
            d = sqrt(dx * dx + dy * dy);
        }
        prefix_min_first(d, li);                           // inclusive prefix-min keeping the earlier index on ties
        if (!(d < cd)) {                                   // fold the carry (earlier) in
            d = cd;
            li = ci;
        }
        const unsigned long long stop = __ballot(i < P && i - li >= limit);
        if (stop) {
            const int first = __builtin_ffsll((long long)stop) - 1;
            return __shfl(li, first, 64);
        }
        const int last = min(63, P - 1 - base);
        cd = __shfl(d, last, 64);
        ci = __shfl(li, last, 64);
    }
    return ci;
}

// dynamic LDS (doubles): 7 * max_ref
__global__ __launch_bounds__(64) void frenet_project_wave_kernel(
    int B, int max_ref, int max_obs, const double* __restrict__ ref_line, const int* __restrict__ n_ref,
    const double* __restrict__ origin_xy, const double* __restrict__ start_xy, const double* __restrict__ start_v,
    const double* __restrict__ start_a, const double* __restrict__ obs_xy, const int* __restrict__ n_obs,
    double* __restrict__ s_map, double* __restrict__ obs_s, double* __restrict__ obs_l, double* __restrict__ begin_sl,
    double* __restrict__ start, int obs_cap, const double* __restrict__ dyn, int* __restrict__ n_obs_out) {
    __builtin_amdgcn_s_setprio(EMP_PRIO_FRONT);
    // obs_xy rows hold max_obs slots, obs_s / obs_l rows obs_cap >= max_obs (+3 when `dyn` is given: the virtual
    // obstacles of test_9.py:137-169 are appended behind the projected ones and n_obs_out gets the total)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    double* lx = lds;
    double* ly = lx + max_ref;
    double* lth = ly + max_ref;
    double* lk = lth + max_ref;
    double* sm = lk + max_ref;
    double* lcos = sm + max_ref;                        // cos / sin of every node's heading: each node one lane, once per
    double* lsin = lcos + max_ref;                      // scene - the ten projections below read them instead of calling
    const double* line = ref_line + (size_t)b * max_ref * 4;   // cos / sin again (half of this kernel's instructions were those calls)
    const int P = min(max(n_ref[b], 0), max_ref);       // clamped to the row's capacity
    // Every coordinate the ten match scans below start from is fetched HERE, next to the reference line: one round trip
    // to memory for the whole kernel.  Loaded where they are used - one obstacle per scan, each scan behind the previous
    // one - they cost a dependent trip each: a third of the kernel's 30 us, on the queue that is the step's critical path.
    const int k = n_obs ? min(max(n_obs[b], 0), max_obs) : 0;   // clamped to the row's capacity
    const double ox = origin_xy[2 * b], oy = origin_xy[2 * b + 1];
    const double px = start_xy[2 * b], py = start_xy[2 * b + 1];
    double obx = 0.0, oby = 0.0;                        // obstacle `lane` of the first 64 (the rest are loaded in the loop)
    if (lane < k) {
        obx = obs_xy[((size_t)b * max_obs + lane) * 2];
        oby = obs_xy[((size_t)b * max_obs + lane) * 2 + 1];
    }
    for (int i = lane; i < P; i += 64) {
        lx[i] = line[4 * i];
        ly[i] = line[4 * i + 1];
        const double th = line[4 * i + 2];
        lth[i] = th;
        lk[i] = line[4 * i + 3];
        lcos[i] = cos(th);
        lsin[i] = sin(th);
    }
    __syncthreads();
    // cumulative chord length (ref planning_utils.py:461-466).  The chords are computed one per lane, but the
    // running sum is formed strictly left to right like the reference's loop: the planning-start s inherits its
    // last bits, and with an integer sample_s those bits decide int(end_s - start_s), i.e. the number of
    // densified points (path_planning.py:405).
    double carry = 0.0;
    for (int base = 0; base < P; base += 64) {
        const int i = base + lane;
        double d = 0.0;
        if (i >= 1 && i < P) {
            const double dx = lx[i] - lx[i - 1], dy = ly[i] - ly[i - 1];
            d = sqrt(dx * dx + dy * dy);
        }
        // s_i = chord_i + s_{i-1}, one add per point in the reference's order.  As a lane-shift sweep (emp_qp_wave.h): every
        // lane adds its chord to its left neighbour's CURRENT sum at every step, lanes 0..k are final after step k - three
        // vector instructions per step where the loop over readlane'd chords took fourteen (a quarter of this kernel).
        const int cnt = min(64, P - base);
        if (lane == 0) d = (base >= 1) ? d + carry : 0.0;            // the chunk's first point continues the previous chunk
        double mine = d;
        for (int k = 1; k < cnt; ++k) mine = d + lane_up1(mine);     // lane 0: neighbour reads 0, d + 0.0 == d bit for bit
        carry = __shfl(mine, cnt - 1, 64);
        if (i < P) sm[i] = mine;
    }
    __syncthreads();
    auto node = [&](int i) { return Node{lx[i], ly[i], lth[i], lk[i]}; };
    // origin of the s axis (ref :457-471)
    const int m0 = match_scan_wave(lx, ly, P, ox, oy, 50);
    const double s0 = projection_s_cs(node(m0), lcos[m0], lsin[m0], sm[m0], ox, oy);
    __syncthreads();
    for (int i = lane; i < P; i += 64) {
        const double v = sm[i] - s0;
        sm[i] = v;
        s_map[(size_t)b * max_ref + i] = v;
    }
    __syncthreads();
    // obstacles (ref test_9.py:122): matches one after the other (each scan is wave-parallel), then one
    // obstacle per lane for the projection arithmetic; l uses the FIRST obstacle's match (quirk :413)
    int my_match = 0, first_match = 0;
    for (int j = 0; j < k; ++j) {
        if (j >= 64 && (j & 63) == 0 && (j & ~63) + lane < k) {       // the next 64 obstacles (rows of more than 64 slots)
            obx = obs_xy[((size_t)b * max_obs + (j & ~63) + lane) * 2];
            oby = obs_xy[((size_t)b * max_obs + (j & ~63) + lane) * 2 + 1];
        }
        const double x = __shfl(obx, j & 63, 64), y = __shfl(oby, j & 63, 64);
        const int mj = match_scan_wave(lx, ly, P, x, y, 50);
        if (j == 0) first_match = mj;
        if ((j & 63) == lane) my_match = mj;
        if ((j & 63) == 63 || j == k - 1) {                 // flush a full set of lanes
            const int jj = (j & ~63) + lane;
            if (jj <= j) {
                obs_s[(size_t)b * obs_cap + jj] = projection_s_cs(node(my_match), lcos[my_match], lsin[my_match], sm[my_match], obx, oby);
                obs_l[(size_t)b * obs_cap + jj] = lateral_offset(project_on_cs(node(first_match), lcos[first_match], lsin[first_match], obx, oby), obx, oby);
            }
        }
    }
    // planning start (ref test_9.py:134 and :172-177)
    const int ms = match_scan_wave(lx, ly, P, px, py, 50);
    if (lane == 0) {
        const Node proj = project_on_cs(node(ms), lcos[ms], lsin[ms], px, py);
        const double bs = projection_s_cs(node(ms), lcos[ms], lsin[ms], sm[ms], px, py);
        if (begin_sl) {
            begin_sl[2 * b] = bs;
            begin_sl[2 * b + 1] = lateral_offset(proj, px, py);
        }
        const FrenetState fs = frenet_state(proj, px, py, start_v[2 * b], start_v[2 * b + 1], start_a[2 * b], start_a[2 * b + 1]);
        start[4 * b + 0] = bs;
        start[4 * b + 1] = fs.l;
        start[4 * b + 2] = fs.dl_ds;
        start[4 * b + 3] = fs.ddl_ds;
        // ref test_9.py:137-169: the FIRST dynamic obstacle (distance Dis, speed V_obs) becomes three obstacles on
        // the centre line covering the stretch where it and the ego meet, unless they part beyond s = 80 m
        int total = k;
        if (dyn && !isnan(dyn[2 * b]) && k + 3 <= obs_cap) {
            const double Len_vehicle = 2.910, Len_obs = 3.0;
            const double Dis = dyn[2 * b], V_obs = dyn[2 * b + 1];
            const double vx = start_v[2 * b], vy = start_v[2 * b + 1];
            const double V_ego = sqrt(vx * vx + vy * vy);
            const double delta_v = V_ego - V_obs;
            const double meet_t = ((Dis - Len_vehicle / 2.0) - Len_obs / 2.0) / delta_v;
            const double delta_t = (Len_vehicle + Len_obs) / delta_v;
            const double leave_t = meet_t + delta_t;
            const double meet_s = ((bs + Dis) + V_obs * meet_t) - Len_obs / 2.0;
            const double leave_s = ((bs + Dis) + V_obs * leave_t) + Len_obs / 2.0;
            const double delta_s = leave_s - meet_s;
            const double obs_pos = meet_s + delta_s / 2.0;
            if (leave_s < 80.0) {
                double* os = obs_s + (size_t)b * obs_cap + k;
                double* ol = obs_l + (size_t)b * obs_cap + k;
                os[0] = meet_s - 10.0;
                os[1] = obs_pos;
                os[2] = leave_s;
                ol[0] = ol[1] = ol[2] = 0.0;
                total = k + 3;
            }
        }
        if (n_obs_out) n_obs_out[b] = total;
    }
}

// ref: match_projection_points (mode 0) / find_match_points (mode 1), one lane per scene, points in order
__global__ void match_points_kernel(int B, int max_ref, int max_pts, const double* __restrict__ ref_line,
                                    const int* __restrict__ n_ref, const double* __restrict__ xy,
                                    const int* __restrict__ n_pts, const int* __restrict__ is_first_run,
                                    const int* __restrict__ pre_match_index, int* __restrict__ match_index,
                                    double* __restrict__ proj, int windowed_api) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* line = ref_line + (size_t)b * max_ref * 4;
    const int P = min(max(n_ref[b], 0), max_ref);       // clamped to the row's capacity
    const int k = min(max(n_pts[b], 0), max_pts);       // a count beyond the row's capacity is clamped, never followed
    int m_first = 0;
    // The windowed search starts at pre_match_index: the reference indexes the path with it (planning_utils.py:123)
    // and raises IndexError past the end; an index outside [0, P) - or an empty path - yields match_index = -1 for
    // every point of the scene (the drop-in turns that into IndexError) and touches no node.
    const bool windowed = windowed_api && !is_first_run[b];
    if (P < 1 || (windowed && (pre_match_index[b] < 0 || pre_match_index[b] >= P))) {
        for (int j = 0; j < k; ++j) {
            match_index[(size_t)b * max_pts + j] = -1;
            double* o = proj + ((size_t)b * max_pts + j) * 4;
            o[0] = o[1] = o[2] = o[3] = __builtin_nan("");
        }
        return;
    }
    for (int j = 0; j < k; ++j) {
        const double x = xy[((size_t)b * max_pts + j) * 2], y = xy[((size_t)b * max_pts + j) * 2 + 1];
        int m;
        if (!windowed) {
            m = match_scan(line, P, x, y, 0, 1, 50);                               // ref planning_utils.py:72-92 / :383-402
        } else {
            const int st = pre_match_index[b];                                     // ref :123-167
            const Node pm = node_at(line, st);
            const double flag = dot2(x - pm.x, y - pm.y, cos(pm.theta), sin(pm.theta));   // ref :139 np.dot
            m = match_scan(line, P, x, y, st, flag > 0.0 ? 1 : -1, 5);
        }
        if (j == 0) m_first = m;
        match_index[(size_t)b * max_pts + j] = m;
        const Node pr = project_on(node_at(line, m_first), x, y);                  // quirk: first point's match (:103/:169/:413)
        double* o = proj + ((size_t)b * max_pts + j) * 4;
        o[0] = pr.x;
        o[1] = pr.y;
        o[2] = pr.theta;
        o[3] = pr.kappa;
    }
}

__global__ void heading_kappa_kernel(int B, int max_pts, const double* __restrict__ xy, const int* __restrict__ n_pts,
                                     double* __restrict__ theta, double* __restrict__ kappa) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int m = n_pts[b];
    if (m < 2) return;
    heading_kappa(xy + (size_t)b * max_pts * 2, 2, m, theta + (size_t)b * max_pts, 1, kappa + (size_t)b * max_pts, 1);
}

// ---------------------------------------------------------------------------------------------
// ref: cal_lmin_lmax, path_planning.py:222-273.  returns false where the reference raises IndexError.
// ---------------------------------------------------------------------------------------------
__device__ inline int argmin_abs(const double* s, int stride, int n, double target) {
    int best = 0;
    double bv = fabs(s[0] - target);
    for (int j = 1; j < n; ++j) {
        const double v = fabs(s[j * stride] - target);
        if (v < bv) {
            bv = v;
            best = j;
        }
    }
    return best;
}

// The three scans of one obstacle (ref path_planning.py:240, :241, :257) in ONE pass over the stations, four stations a step: the
// same comparisons in the same order per target as three argmin_abs calls (strict '<': the first minimum), but the stations are
// read once and four LDS reads are in flight at a time - the three separate loops were 3 n dependent LDS round trips of a wavefront
// that has its SIMD to itself (round 6: 4 us of the path-QP kernel's 30 us set-up).  n >= 1.
__device__ inline void argmin_abs3(const double* s, int n, double t0, double t1, double t2, int* i0, int* i1, int* i2) {
    const double f = s[0];
    double b0 = fabs(f - t0), b1 = fabs(f - t1), b2 = fabs(f - t2);
    int k0 = 0, k1 = 0, k2 = 0;
    for (int j0 = 1; j0 < n; j0 += 4) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = s[min(j0 + u, n - 1)];      // past the end: the last station again (never a strict improvement)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            const double a0 = fabs(v[u] - t0), a1 = fabs(v[u] - t1), a2 = fabs(v[u] - t2);
            if (a0 < b0) { b0 = a0; k0 = j; }
            if (a1 < b1) { b1 = a1; k1 = j; }
            if (a2 < b2) { b2 = a2; k2 = j; }
        }
    }
    *i0 = min(k0, n - 1);      // (k never exceeds n - 1: a repeated last station is not strictly below itself; the clamp costs nothing)
    *i1 = min(k1, n - 1);
    *i2 = min(k2, n - 1);
}

__device__ inline bool lmin_lmax(const double* dp_s, const double* dp_l, int stride, int n, const double* obs_s,
                                 const double* obs_l, int n_obs, double obs_length, double obs_width, double* l_min,
                                 double* l_max) {
    for (int j = 0; j < n; ++j) {
        l_min[j] = -10.0;                                                          // ref :233-234
        l_max[j] = 10.0;
    }
    for (int k = 0; k < n_obs; ++k) {
        const int lo = argmin_abs(dp_s, stride, n, obs_s[k] - obs_length / 2.0) + 2;   // ref :240
        const int hi = argmin_abs(dp_s, stride, n, obs_s[k] + obs_length / 2.0) + 2;   // ref :241
        const int centre = argmin_abs(dp_s, stride, n, obs_s[k]);                      // ref :257
        const bool below = dp_l[centre * stride] < obs_l[k];                           // ref :263
        for (int j = lo; j <= hi; ++j) {
            if (j >= n) return false;                                                  // IndexError in the reference
            if (below) l_max[j] = fmin(l_max[j], obs_l[k] - obs_width / 2.0);
            else l_min[j] = fmax(l_min[j], obs_l[k] + obs_width / 2.0);
        }
    }
    return true;
}

__global__ void lmin_lmax_kernel(int B, int max_pts, int max_obs, const double* __restrict__ dp_s,
                                 const double* __restrict__ dp_l, const int* __restrict__ n_pts,
                                 const double* __restrict__ obs_s, const double* __restrict__ obs_l,
                                 const int* __restrict__ n_obs, double obs_length, double obs_width,
                                 double* __restrict__ l_min, double* __restrict__ l_max, int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool ok = lmin_lmax(dp_s + (size_t)b * max_pts, dp_l + (size_t)b * max_pts, 1, n_pts[b],
                              obs_s + (size_t)b * max_obs, obs_l + (size_t)b * max_obs, min(max(n_obs[b], 0), max_obs), obs_length,
                              obs_width, l_min + (size_t)b * max_pts, l_max + (size_t)b * max_pts);
    status[b] = ok ? 0 : kStBoundIndex;
}

// ---------------------------------------------------------------------------------------------
// heading / curvature of m points held in LDS (px, py), one point per lane (ref planning_utils.py:185-228)
// th: LDS scratch [m].  out[i*ostride + 0..3] = x, y, theta, kappa.  Contains barriers: whole wavefront calls.
// ---------------------------------------------------------------------------------------------
__device__ inline void heading_kappa_wave(const double* px, const double* py, int m, double* th, double* out,
                                          int ostride) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < m; i += 64) {
        const int a = (i - 1 > 0) ? i - 1 : 0, b = (i < m - 2) ? i : m - 2;
        const double dx = ((px[a + 1] - px[a]) + (px[b + 1] - px[b])) / 2.0;
        const double dy = ((py[a + 1] - py[a]) + (py[b + 1] - py[b])) / 2.0;
        th[i] = atan2(dy, dx);
    }
    __syncthreads();
    for (int i = lane; i < m; i += 64) {
        const int a = (i - 1 > 0) ? i - 1 : 0, b = (i < m - 2) ? i : m - 2;
        const double dx = ((px[a + 1] - px[a]) + (px[b + 1] - px[b])) / 2.0;
        const double dy = ((py[a + 1] - py[a]) + (py[b + 1] - py[b])) / 2.0;
        const double dpre = th[a + 1] - th[a], daft = th[b + 1] - th[b];
        double* o = out + (size_t)i * ostride;
        o[0] = px[i];
        o[1] = py[i];
        o[2] = th[i];
        o[3] = sin((dpre + daft) / 2.0) / sqrt(dx * dx + dy * dy);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// ref: Quadratic_planning, path_planning.py:78-219 (stand-alone stage): one wavefront per scene
// dynamic LDS: path_qp_words(cap) doubles
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void path_qp_wave_kernel(int B, int max_pts, int cap, QpDev Q,
                                                          const double* __restrict__ l_min,
                                                          const double* __restrict__ l_max,
                                                          const int* __restrict__ n_pts,
                                                          const double* __restrict__ start_l3,
                                                          double* __restrict__ qp_l, double* __restrict__ qp_dl,
                                                          double* __restrict__ qp_ddl, int* __restrict__ iters,
                                                          int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x;
    const size_t o = (size_t)b * max_pts;
    const int n = n_pts[b];
    int it = 0;
    int rc = 2;
    const bool fits = n <= cap && n <= max_pts;
    rc = path_qp_group<64>(lds, l_min + o, l_max + o, n, start_l3[3 * b], start_l3[3 * b + 1], start_l3[3 * b + 2], Q.qp,
                           qp_l + o, qp_dl + o, qp_ddl + o, &it, fits);
    if ((threadIdx.x & 63) == 0) {
        if (iters) iters[b] = it;
        status[b] = rc ? kStQpFailed : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// ref: smooth_reference_line, planning_utils.py:262-361 (stand-alone stage): one wavefront per polyline,
// x on lanes 0-31, y on lanes 32-63, then heading / curvature one point per lane.
// dynamic LDS: 2 * BoxRangeQp::words(cap, cap) + cap doubles
// ---------------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(64) void smooth_wave_kernel(int B, int max_pts, int cap, SmoothQpParams sx,
                                                         SmoothQpParams sy, const double* __restrict__ xy,
                                                         const int* __restrict__ n_pts, double* __restrict__ out,
                                                         int* __restrict__ iters, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x;
    const int m = n_pts[b];
    int it = 0, rc = 2;
    double *px = nullptr, *py = nullptr;
    if (m <= cap && m <= max_pts) rc = smooth_pair_wave<WIDE>(lds, xy + (size_t)b * max_pts * 2, 2, m, sx, sy, &px, &py, &it);
    if (rc == 0) heading_kappa_wave(px, py, m, lds + 2 * BoxRangeQp::words(m, m), out + (size_t)b * max_pts * 4, 4);
    if ((threadIdx.x & 63) == 0) {
        if (iters) iters[b] = it;
        status[b] = rc ? kStSmoothFailed : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Front end of one planning cycle (ref test_9.py:99-110): find_match_points for the predicted location on the
// GLOBAL path (planning_utils.py:49-182: windowed from the previous match unless is_first_run) -> sampling
// (:231-259: 10 nodes back, 40 forward, shifted at either end of the path, 51 nodes) -> smooth_reference_line
// (:262-361) -> the reference line of the cycle.  One wavefront per scene.
// dynamic LDS: 2 * kRefLinePoints (gathered xy) + 2 * BoxRangeQp::words(51, 51) + 51 doubles
// ---------------------------------------------------------------------------------------------
constexpr int kRefLinePoints = 51;

__global__ __launch_bounds__(64) void reference_line_wave_kernel(int B, int max_global, SmoothQpParams sx, SmoothQpParams sy,
                                                                 const double* __restrict__ global_path,
                                                                 const int* __restrict__ n_global,
                                                                 const double* __restrict__ pred_xy,
                                                                 const int* __restrict__ is_first_run,
                                                                 const int* __restrict__ pre_match_index,
                                                                 double* __restrict__ ref_line, int* __restrict__ n_ref,
                                                                 int* __restrict__ match_index, int* __restrict__ iters,
                                                                 int* __restrict__ status, int n_ref_on_fail = 0) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const double* line = global_path + (size_t)b * max_global * 4;
    const int P = min(n_global[b], max_global);          // a count beyond the row's capacity is clamped, never followed
    double* gxy = lds;                                   // [51][2]
    double* qmem = gxy + 2 * kRefLinePoints;
    double* out = ref_line + (size_t)b * kRefLinePoints * 4;
    int m = 0, first = 0, fail = 0;
    if (lane == 0) {
        const double x = pred_xy[2 * b], y = pred_xy[2 * b + 1];
        const int st = pre_match_index[b];
        if (P < 1 || ((!is_first_run || !is_first_run[b]) && (st < 0 || st >= P))) {
            fail = kStSOutOfRange;                       // the reference raises IndexError
        } else if (is_first_run && is_first_run[b]) {
            m = match_scan(line, P, x, y, 0, 1, 50);     // ref :72-92
        } else {
            const Node pm = node_at(line, st);           // ref :123-167
            const double flag = dot2(x - pm.x, y - pm.y, cos(pm.theta), sin(pm.theta));   // ref :139 np.dot
            m = match_scan(line, P, x, y, st, flag > 0.0 ? 1 : -1, 5);
        }
        // sampling (ref :244-259): the arguments are overwritten with 10 back / 40 forward
        int back = 10, fwd = 40;
        if (m < back) {
            back = m;
            fwd = 50 - back;
        }
        if (P - m - 1 < fwd) {
            fwd = P - m - 1;
            back = 50 - fwd;
        }
        first = m - back;
        if (!fail && (first < 0 || m + fwd + 1 > P)) fail = kStSOutOfRange;   // path shorter than 51 nodes: the
                                                                              // reference's slice wraps around
    }
    m = __shfl(m, 0, 64);
    first = __shfl(first, 0, 64);
    fail = __shfl(fail, 0, 64);
    int it = 0;
    if (!fail) {
        if (lane < kRefLinePoints) {
            gxy[2 * lane] = line[4 * (first + lane)];
            gxy[2 * lane + 1] = line[4 * (first + lane) + 1];
        }
        __syncthreads();
        double *px = nullptr, *py = nullptr;
        const int rc = smooth_pair_wave<true>(qmem, gxy, 2, kRefLinePoints, sx, sy, &px, &py, &it);
        if (rc) fail = kStSmoothFailed;
        else heading_kappa_wave(px, py, kRefLinePoints, qmem + 2 * BoxRangeQp::words(kRefLinePoints, kRefLinePoints), out, 4);
    }
    if (fail)
        for (int i = lane; i < kRefLinePoints * 4; i += 64) out[i] = 0.0;
    if (lane == 0) {
        n_ref[b] = fail ? n_ref_on_fail : kRefLinePoints;       // (the fused cycle hands a failed line on as two zero nodes)
        match_index[b] = m;
        if (iters) iters[b] = it;
        status[b] = fail;
    }
}

// ---------------------------------------------------------------------------------------------
// ref: frenet_2_x_y_theta_kappa without its smoothing call, path_planning.py:29-46
// ---------------------------------------------------------------------------------------------
__device__ inline int frenet_path_to_xy(const double* line, const double* s_map, int P, double begin_s, double begin_l,
                                        const double* path_s, const double* path_l, int n, double* target_xy,
                                        int cap, bool* s_error, bool* trunc) {
    *s_error = false;
    *trunc = false;
    int idx = 0, m = 0;
    Node pr;
    if (!proj_point(line, s_map, P, begin_s, &idx, &pr)) {                        // ref :31
        *s_error = true;
        return 0;
    }
    target_xy[0] = pr.x + begin_l * (-sin(pr.theta));                             // ref :32-34
    target_xy[1] = pr.y + begin_l * cos(pr.theta);
    m = 1;
    for (int i = 0; i < n; ++i) {
        const double s = path_s[i];
        if (s > s_map[P - 1]) break;                                               // ref :40-41
        if (!proj_point(line, s_map, P, s, &idx, &pr)) {                           // ref :42 (IndexError)
            *s_error = true;
            return m;
        }
        if (m >= cap) {
            *trunc = true;
            break;
        }
        target_xy[2 * m] = pr.x + path_l[i] * (-sin(pr.theta));                    // ref :44-46
        target_xy[2 * m + 1] = pr.y + path_l[i] * cos(pr.theta);
        ++m;
    }
    return m;
}

__global__ void path_to_xy_kernel(int B, int max_ref, int max_pts, const double* __restrict__ ref_line,
                                  const double* __restrict__ s_map, const int* __restrict__ n_ref,
                                  const double* __restrict__ begin_sl, const double* __restrict__ path_s,
                                  const double* __restrict__ path_l, const int* __restrict__ n_pts,
                                  double* __restrict__ target_xy, int* __restrict__ n_out, int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    bool s_err, trunc;
    const int m = frenet_path_to_xy(ref_line + (size_t)b * max_ref * 4, s_map + (size_t)b * max_ref, n_ref[b],
                                    begin_sl[2 * b], begin_sl[2 * b + 1], path_s + (size_t)b * max_pts,
                                    path_l + (size_t)b * max_pts, n_pts[b], target_xy + (size_t)b * (max_pts + 1) * 2,
                                    max_pts + 1, &s_err, &trunc);
    n_out[b] = m;
    status[b] = (s_err ? kStSOutOfRange : 0) | (trunc ? kStTruncated : 0);
}

// ---------------------------------------------------------------------------------------------
// One cycle, middle part (ref test_9.py:187-210): decimate -> bounds -> path QP -> midpoints.
// One scene per GROUP of G lanes (G = 32: two scenes per wavefront when cap <= 34 stations, else G = 64).
// dynamic LDS (doubles), per group: 5*cap + 4*max_obs + path_qp_words(cap)  (G = 32: + path_qp_words_pair() instead)
// ---------------------------------------------------------------------------------------------
// R > 0: eight scenes per wavefront (G = 8), R stations per lane (emp_qp_rows.h; cap <= 8 R + 2 stations).
// doubles of LDS per group (scene)
template <int G, int R = 0>
__host__ __device__ constexpr int cycle_qp_group_words(int cap, int max_obs) {
    // Rows form: an ODD number of doubles.  The eight (four) groups of a wavefront read the same offsets of their own slices in
    // every LDS instruction; with a stride that is a multiple of 32 doubles they would all sit on the same banks (468 doubles at
    // R = 3: the station and Hessian arrays at 2-4 times their conflict-free cost, 14.5 % of the kernel's wavefront cycles in
    // SQ_LDS_BANK_CONFLICT, profiles/r06a_sq.csv) - an odd stride walks the groups across the banks.
    if (R > 0) return (path_qp_words_rows<(R > 0 ? G : 8), (R > 0 ? R : 3)>() + (4 * max_obs <= 4 * G * R ? 0 : 4 * max_obs)) | 1;
    return 5 * cap + 4 * max_obs + (G == 32 ? path_qp_words_pair() : path_qp_words(cap));
}
template <int G, int R = 0>
__device__ __forceinline__ void cycle_qp_body(int B, int max_pts, int max_obs, int cap, const QpDev& Q,
                                                           const double* __restrict__ dp_s,
                                                           const double* __restrict__ dp_l,
                                                           const int* __restrict__ dp_len,
                                                           const double* __restrict__ obs_s,
                                                           const double* __restrict__ obs_l,
                                                           const int* __restrict__ n_obs,
                                                           const double* __restrict__ start,
                                                           double* __restrict__ path_s, double* __restrict__ path_l,
                                                           int* __restrict__ path_len, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    __builtin_amdgcn_s_setprio(EMP_PRIO_BACK);     // wavefront priority of the back stage, see emp_context.h
    constexpr int GPW = 64 / G;                                   // groups (scenes) per wavefront
    const int lane = threadIdx.x & 63, grp = lane / G, gl = lane & (G - 1);
    const int b = blockIdx.x * GPW + grp;
    const bool present = b < B;
    static_assert(R == 0 || G == 8 || G == 16, "the rows solver runs on groups of 8 or 16 lanes");
    const int per_group = cycle_qp_group_words<G, R>(cap, max_obs);
    double* lds = lds_all + (size_t)grp * per_group;
    const size_t o = (size_t)(present ? b : 0) * max_pts;
    double* sd = lds;                 // decimated station s   [cap]  (rows form: see below)
    double *ld, *lmin, *lmax, *ql, *otab, *qmem;
    if constexpr (R > 0) {
        // Rows form (round 5): a path-QP wavefront's LDS - eight scenes - is what bounds how many edge-cost blocks of the NEXT
        // batch fit on its CU (measured: +1.2 us per staged step for every KB a QP wavefront holds, profiles/r05_edge/README.md 8).
        // The arrays that are dead once the QP is set up live inside solver arrays that are not written before its first
        // iteration: sd, ld, lmin, lmax in tmp | wgt (2 capS F + 4 F >= 4 cap: the station abscissae are re-read from device memory
        // for the midpoints at the end), the QP's result ql - written after the last iteration - in rhs | dua (2 capN >= cap), the
        // obstacle table in P.  Same values, same arithmetic; 5 cap + 4 max_obs doubles per scene less.
        constexpr int capN = G * R, capS = G * R, kCc = G * R + 4;
        qmem = lds;
        double* qbase = qmem + kCc;                       // PathRangeQp::bind_fast(qbase, capN, capS, ...): P q u rhs dua c lo hi tmp wgt
        ql = qbase + capN * 4 + capN * 2;                 // rhs | dua
        sd = qbase + capN * 4 + capN * 4 + capS * 2 * 3;  // tmp | wgt | slack: sd, ld, lmin, lmax (4 cap <= 2 capS F + 4 F)
        ld = sd + cap;
        lmin = ld + cap;
        lmax = lmin + cap;
        // the obstacle table is dead before the QP is set up: it lives where the Hessian band P will be written (4 capN doubles),
        // or behind the group's other arrays when the obstacle rows are wider than that
        otab = 4 * max_obs <= 4 * capN ? qbase : qmem + path_qp_words_rows<G, R>();
    } else {
        ld = sd + cap;                // decimated DP l        [cap]
        lmin = ld + cap;              // [cap]
        lmax = lmin + cap;            // [cap]
        ql = lmax + cap;              // QP result l           [cap]
        otab = ql + cap;              // per obstacle: lo, hi, below, bound   [4*max_obs]
        qmem = otab + 4 * max_obs;
    }
#ifdef EMP_QP_PROBE_EMPTY
    if (Q.debug_stage == 4) return;                 // development timing: the launch alone (tools/qp_phase_probe.py)
#endif
    const int ne = present ? dp_len[b] : 0;
    const int dec = Q.decimate > 0 ? Q.decimate : 1;
    const int n = (ne + dec - 1) / dec;                                            // len(x[::dec])
    const int st = present ? status[b] : 0;
    int fail = 0;                      // status bits this kernel adds; once set the group idles through the barriers
    if (present && (n > cap || n + (Q.midpoint ? 1 : 0) > max_pts || n < 1)) fail = kStTruncated;
    bool live = present && !fail;
    for (int i = gl; i < (live ? n : 0); i += G) {
        sd[i] = dp_s[o + (size_t)i * dec];
        ld[i] = dp_l[o + (size_t)i * dec];
    }
    __syncthreads();
#ifdef EMP_QP_PROBE_EMPTY
    if (Q.debug_stage == 5) return;                 // ... behind the loads of the DP path
#endif
    if (Q.use_qp) {
        // ---- cal_lmin_lmax (ref path_planning.py:222-273): one obstacle per lane finds its index range,
        // then one station per lane folds the (commutative) min / max over the obstacles covering it
        const int nob = live ? min(max(n_obs[b], 0), max_obs) : 0;
        bool bad = false;
        for (int k = gl; k < nob; k += G) {
            const double os = obs_s[(size_t)b * max_obs + k], ol = obs_l[(size_t)b * max_obs + k];
            int lo, hi, centre;                                                    // ref :240, :241, :257
            argmin_abs3(sd, n, os - Q.obs_length / 2.0, os + Q.obs_length / 2.0, os, &lo, &hi, &centre);
            lo += 2;
            hi += 2;
            const bool below = ld[centre] < ol;                                    // ref :263
            otab[4 * k + 0] = (double)lo;
            otab[4 * k + 1] = (double)hi;
            otab[4 * k + 2] = below ? 1.0 : 0.0;
            otab[4 * k + 3] = below ? ol - Q.obs_width / 2.0 : ol + Q.obs_width / 2.0;
            if (lo <= hi && hi >= n) bad = true;                                   // IndexError in the reference
        }
        if (group_any<G>(bad) && live) {
            fail = kStBoundIndex;
            live = false;
        }
        __syncthreads();
        {
            // obstacle by obstacle (its four table entries read ONCE, the same for every lane), each lane folding it into its own
            // stations gl, gl + G, ...: the same fmin / fmax in the same obstacle order per station as the station-outer loop this
            // replaces, with nob LDS round trips instead of (stations per lane) x nob
            constexpr int T = R > 0 ? R + 1 : (G == 64 ? 4 : 2);                  // stations a lane can own (cap <= G R + 2; 34 / 32; 255 / 64)
            const int nn = live ? n : 0;
            double a[T], c[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                a[t] = -10.0;                                                      // ref :233-234
                c[t] = 10.0;
            }
            for (int k = 0; k < nob; ++k) {
                const double klo = otab[4 * k], khi = otab[4 * k + 1], kbelow = otab[4 * k + 2], kb = otab[4 * k + 3];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const double j = (double)(gl + G * t);
                    if (j >= klo && j <= khi) {
                        if (kbelow != 0.0) c[t] = fmin(c[t], kb);
                        else a[t] = fmax(a[t], kb);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = gl + G * t;
                if (j < nn) {
                    lmin[j] = a[t];
                    lmax[j] = c[t];
                }
            }
            for (int j = gl + G * T; j < nn; j += G) {                            // (no launch gives a lane more stations than T: kept for safety)
                double aj = -10.0, cj = 10.0;
                for (int k = 0; k < nob; ++k) {
                    if ((double)j >= otab[4 * k] && (double)j <= otab[4 * k + 1]) {
                        if (otab[4 * k + 2] != 0.0) cj = fmin(cj, otab[4 * k + 3]);
                        else aj = fmax(aj, otab[4 * k + 3]);
                    }
                }
                lmin[j] = aj;
                lmax[j] = cj;
            }
        }
        __syncthreads();
        if (Q.debug_stage == 1) live = false;
        int it = 0;
        const int sb = present ? b : 0;
        int rc;
        if constexpr (R > 0)
            rc = path_qp_group_rows<G, R>(qmem, lmin, lmax, n, start[4 * sb + 1], start[4 * sb + 2], start[4 * sb + 3], Q.qp, ql, &it,
                                       live, Q.debug_stage);
        else
            rc = path_qp_group<G>(qmem, lmin, lmax, n, start[4 * sb + 1], start[4 * sb + 2], start[4 * sb + 3], Q.qp,
                                  ql, nullptr, nullptr, &it, live, Q.debug_stage);
        if (live && rc) {
            fail = kStQpFailed;
            live = false;
        }
    } else {
        for (int i = gl; i < (live ? n : 0); i += G) ql[i] = ld[i];
        __syncthreads();
    }
    if (present) {
        double* ps = path_s + o;
        double* pl = path_l + o;
        int plen = 0;
        // station abscissa i: from LDS, or - rows form, where the solver has overwritten it - from the DP path again (the same
        // number: sd[i] IS dp_s[i dec]).  The choice is made at COMPILE time (round 6): as a run-time select between a device and an
        // LDS address it compiled to one flat load behind a select of two pointers, and builds of this kernel that differed only
        // in where an unrelated value was loaded read that address wrong (an aperture violation, or midpoints of the wrong stations).
        auto sd_at = [&](int i) {
            if constexpr (R > 0) return dp_s[o + (size_t)i * dec];
            else return sd[i];
        };
        if (live) {
            if (Q.midpoint) {                                                      // ref test_9.py:204-210
                for (int i = gl; i <= n; i += G) {
                    if (i == 0) {
                        ps[0] = sd_at(0);
                        pl[0] = ql[0];
                    } else if (i == n) {
                        ps[n] = sd_at(n - 1);
                        pl[n] = ql[n - 1];
                    } else {
                        ps[i] = (sd_at(i) + sd_at(i - 1)) / 2.0;
                        pl[i] = (ql[i] + ql[i - 1]) / 2.0;
                    }
                }
                plen = n + 1;
            } else {
                for (int i = gl; i < n; i += G) {
                    ps[i] = sd_at(i);
                    pl[i] = ql[i];
                }
                plen = n;
            }
        }
        for (int i = plen + gl; i < max_pts; i += G) {                           // padding reads as 0
            ps[i] = 0.0;
            pl[i] = 0.0;
        }
        if (gl == 0) {
            path_len[b] = plen;
            status[b] = st | fail;
        }
    }
}

#define EMP_CYCLE_QP_PARAMS                                                                                         \
    int B, int max_pts, int max_obs, int cap, QpDev Q, const double *__restrict__ dp_s, const double *__restrict__ dp_l, \
        const int *__restrict__ dp_len, const double *__restrict__ obs_s, const double *__restrict__ obs_l,           \
        const int *__restrict__ n_obs, const double *__restrict__ start, double *__restrict__ path_s,                  \
        double *__restrict__ path_l, int *__restrict__ path_len, int *__restrict__ status
#define EMP_CYCLE_QP_ARGS B, max_pts, max_obs, cap, Q, dp_s, dp_l, dp_len, obs_s, obs_l, n_obs, start, path_s, path_l, path_len, status
template <int G>
__global__ __launch_bounds__(64) void cycle_qp_wave_kernel(EMP_CYCLE_QP_PARAMS) {
    cycle_qp_body<G>(EMP_CYCLE_QP_ARGS);
}
// 64 / GP scenes per wavefront on groups of GP lanes, R stations per lane (emp_qp_rows.h): a quarter (GP = 8) of the
// wavefronts of the two-per-wavefront kernel, each as long as before.
// (Round 5, measured and not kept: amdgpu_waves_per_eu(2, 2) - 256 registers and 224 bytes of scratch per lane instead of 256 + 74
// accumulation registers as spill space - so that two edge-cost wavefronts fit beside a path-QP wavefront on its SIMD instead of
// one: the kernel alone 137 -> 157 us, the staged step 0.241 -> 0.250 ms.)
template <int GP, int R>
__global__ __launch_bounds__(64) void cycle_qp_rows_kernel(EMP_CYCLE_QP_PARAMS) {
    cycle_qp_body<GP, R>(EMP_CYCLE_QP_ARGS);
}
#undef EMP_CYCLE_QP_PARAMS
#undef EMP_CYCLE_QP_ARGS

// monotone index walk of cal_proj_point from index 0 (ref path_planning.py:62-64: `while s_map[idx + 1] < s: idx += 1`);
// *off_end when it runs past the end.  s_map is a cumulative chord length, i.e. non-decreasing, so the walk's stopping
// index is found by bisection (6 LDS round trips instead of up to P).
__device__ inline int walk_from_zero(const double* sm, int P, double s, bool* off_end) {
    int lo = 0, hi = P - 1;                       // answer in [0, P-1]; P-1 = "no index stops the walk"
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sm[mid + 1] < s) lo = mid + 1;
        else hi = mid;
    }
    *off_end = lo + 1 >= P;
    return lo;
}

// ---------------------------------------------------------------------------------------------
// One cycle, last part (ref path_planning.py:15-49): Frenet->Cartesian one point per lane, x / y smoothing
// on the two half-waves, heading / curvature one point per lane.  One wavefront per scene.
// dynamic LDS (doubles): max_ref + 3*cap + 2 * BoxRangeQp::words(cap, cap)
// The reference walks the s_map index monotonically from the previous point (:42-43); with a non-decreasing
// s_map (what cal_s_map_fun produces) that equals a running maximum of independent walks from index 0.
// ---------------------------------------------------------------------------------------------
// WIDE == false (cap <= 32, the benchmark's 23-point trajectories): only the half-wave smoothing path is
// compiled, which fits four wavefronts per SIMD - all 4096 scenes of a batch are resident at once.
template <bool WIDE>
__device__ __forceinline__ void cycle_cartesian_body(
    int B, int max_ref, int max_pts, int cap, SmoothQpParams sx, SmoothQpParams sy, const double* __restrict__ ref_line,
    const double* __restrict__ s_map, const int* __restrict__ n_ref, const double* __restrict__ begin_sl,
    const double* __restrict__ path_s, const double* __restrict__ path_l, const int* __restrict__ path_len,
    double* __restrict__ traj, int* __restrict__ traj_len, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __builtin_amdgcn_s_setprio(EMP_PRIO_CART);          // as in the path QP kernel
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    double* sm = lds;                       // [max_ref]
    double* txy = sm + max_ref;             // [cap][2] interleaved x, y
    double* th = txy + 2 * cap;             // [cap]
    double* qmem = th + cap;
    const int st = status[b];
    double* out_rows = traj + (size_t)b * (max_pts + 1) * 4;
    auto body = [&]() -> int {          // returns the number of trajectory points (0 = none); wave-uniform control flow
    if (st & (kStQpFailed | kStBoundIndex | kStTruncated)) return 0;
    const double* line = ref_line + (size_t)b * max_ref * 4;
    const int P = min(max(n_ref[b], 0), max_ref);       // clamped to the row's capacity
    const int n = path_len[b];
    for (int i = lane; i < P; i += 64) sm[i] = s_map[(size_t)b * max_ref + i];
    __syncthreads();
    // planning start (ref :31-34)
    bool off = false;
    const double bs = begin_sl[2 * b], bl = begin_sl[2 * b + 1];
    const int idx0 = walk_from_zero(sm, P, bs, &off);
    if (off || P < 2) {
        if (lane == 0) status[b] = st | kStSOutOfRange;
        return 0;
    }
    if (lane == 0) {
        const Node m0 = node_at(line, idx0);
        const double ds = bs - sm[idx0];
        const double th0 = m0.theta + m0.kappa * ds;
        txy[0] = (m0.x + ds * cos(m0.theta)) + bl * (-sin(th0));
        txy[1] = (m0.y + ds * sin(m0.theta)) + bl * cos(th0);
    }
    // path points: count = leading points with s <= s_map[-1] (ref :40-41), index = running max of walks
    const double s_last = sm[P - 1];
    int carry = idx0, count = n;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool in = i < n;
        const double s = in ? path_s[(size_t)b * max_pts + i] : 0.0;
        const int first_bad = __builtin_ffsll((long long)__ballot(in && s > s_last));   // 1-based lane, 0 if none
        if (first_bad) count = min(count, base + first_bad - 1);
        bool o2 = false;
        int k = in ? walk_from_zero(sm, P, fmin(s, s_last), &o2) : 0;
        for (int d = 1; d < 64; d <<= 1) {                 // inclusive running maximum across lanes
            const int v = __shfl_up(k, d, 64);
            if (lane >= d) k = max(k, v);
        }
        k = max(k, carry);
        carry = __shfl(k, 63, 64);
        if (in && i < count && i + 1 < cap) {
            const Node mm = node_at(line, k);
            const double ds = s - sm[k];
            const double thp = mm.theta + mm.kappa * ds;
            const double l = path_l[(size_t)b * max_pts + i];
            txy[2 * (i + 1)] = (mm.x + ds * cos(mm.theta)) + l * (-sin(thp));       // ref :44-46
            txy[2 * (i + 1) + 1] = (mm.y + ds * sin(mm.theta)) + l * cos(thp);
        }
        if (first_bad) break;
    }
    const int m = count + 1;
    __syncthreads();
    if (m > cap || m > max_pts + 1) {
        if (lane == 0) status[b] = st | kStTruncated;
        return 0;
    }
    if (m < 2) {
        if (lane == 0) status[b] = st | kStSmoothFailed;
        return 0;
    }
    int it = 0;
    double *px = nullptr, *py = nullptr;
    const int rc = smooth_pair_wave<WIDE>(qmem, txy, 2, m, sx, sy, &px, &py, &it);
    if (rc) {
        if (lane == 0) status[b] = st | kStSmoothFailed;
        return 0;
    }
    heading_kappa_wave(px, py, m, th, out_rows, 4);
    return m;
    };
    const int m_out = body();
    for (int i = m_out * 4 + lane; i < (max_pts + 1) * 4; i += 64) out_rows[i] = 0.0;       // padding reads as 0
    if (lane == 0) traj_len[b] = m_out;
}

#define EMP_CARTESIAN_ARGS B, max_ref, max_pts, cap, sx, sy, ref_line, s_map, n_ref, begin_sl, path_s, path_l, path_len, traj, traj_len, status
__global__ __launch_bounds__(64) void cycle_cartesian_wave_kernel_wide(
    int B, int max_ref, int max_pts, int cap, SmoothQpParams sx, SmoothQpParams sy, const double* __restrict__ ref_line,
    const double* __restrict__ s_map, const int* __restrict__ n_ref, const double* __restrict__ begin_sl,
    const double* __restrict__ path_s, const double* __restrict__ path_l, const int* __restrict__ path_len,
    double* __restrict__ traj, int* __restrict__ traj_len, int* __restrict__ status) {
    cycle_cartesian_body<true>(EMP_CARTESIAN_ARGS);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void cycle_cartesian_wave_kernel_narrow(
    int B, int max_ref, int max_pts, int cap, SmoothQpParams sx, SmoothQpParams sy, const double* __restrict__ ref_line,
    const double* __restrict__ s_map, const int* __restrict__ n_ref, const double* __restrict__ begin_sl,
    const double* __restrict__ path_s, const double* __restrict__ path_l, const int* __restrict__ path_len,
    double* __restrict__ traj, int* __restrict__ traj_len, int* __restrict__ status) {
    cycle_cartesian_body<false>(EMP_CARTESIAN_ARGS);
}
#undef EMP_CARTESIAN_ARGS

// ---------------------------------------------------------------------------------------------
// The same last part of the cycle with FOUR scenes per wavefront (trajectories of at most 8 R points: R = 3 the
// benchmark's 23, R = 4 up to 32): 16 lanes per scene - Frenet -> Cartesian one point per lane in passes of 16, the x
// and the y smoothing problem on the scene's two 8-lane groups with R points per lane (emp_smooth_rows.h), heading /
// curvature one point per lane again.  A quarter of the wavefronts of cycle_cartesian_wave_kernel_narrow for the same
// work: what that kernel spends per scene on 23 of 64 lanes (trigonometry) and on sweeps for two problems (smoothing)
// is shared by four scenes here.  A problem whose active-set classification does not settle (none on any test or
// benchmark scene) is solved by the whole wavefront with the half-wave solvers of the narrow kernel, one scene at a time.
// With GP = 16 a scene takes 32 lanes and trajectories of up to 64 points fit (BASELINE configs[4]'s 63): two per wavefront.
// dynamic LDS (doubles): (32 / GP) * (max_ref + 5 * cap) + 2 * BoxRangeQp::words(cap, cap)
// ---------------------------------------------------------------------------------------------
template <int GP, int R>
__global__ __launch_bounds__(64) void cycle_cartesian_rows_kernel(
    int B, int max_ref, int max_pts, int cap, SmoothQpParams sx, SmoothQpParams sy, const double* __restrict__ ref_line,
    const double* __restrict__ s_map, const int* __restrict__ n_ref, const double* __restrict__ begin_sl,
    const double* __restrict__ path_s, const double* __restrict__ path_l, const int* __restrict__ path_len,
    double* __restrict__ traj, int* __restrict__ traj_len, int* __restrict__ status, int force_fallback) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __builtin_amdgcn_s_setprio(EMP_PRIO_CART);
    constexpr int SL = 2 * GP, SPW = 64 / SL;     // lanes per scene, scenes per wavefront
    const int lane = threadIdx.x & 63, sc = lane / SL, sl = lane & (SL - 1);
    const int b = blockIdx.x * SPW + sc;
    const bool present = b < B;
    const size_t bb = present ? (size_t)b : 0;
    const int per_scene = max_ref + 5 * cap;
    double* sm = lds + (size_t)sc * per_scene;   // [max_ref]
    double* txy = sm + max_ref;                  // [cap][2] interleaved x, y
    double* th = txy + 2 * cap;                  // [cap]
    double* px = th + cap;                       // [cap] smoothed x
    double* py = px + cap;                       // [cap] smoothed y
    double* qmem = lds + (size_t)SPW * per_scene;  // the fall-back solver's storage (whole wavefront)
    const int st = present ? status[bb] : 0;
    int add = 0;                                 // status bits this kernel adds (scene-uniform)
    bool alive = present && !(st & (kStQpFailed | kStBoundIndex | kStTruncated));
    const double* line = ref_line + bb * max_ref * 4;
    const int P = alive ? min(max(n_ref[bb], 0), max_ref) : 0;       // clamped to the row's capacity
    const int n = alive ? path_len[bb] : 0;
    for (int i = sl; i < P; i += SL) sm[i] = s_map[bb * max_ref + i];
    __syncthreads();
    // planning start (ref :31-34)
    bool off = false;
    const double bs = alive ? begin_sl[2 * bb] : 0.0, bl = alive ? begin_sl[2 * bb + 1] : 0.0;
    const int idx0 = walk_from_zero(sm, P, bs, &off);
    if (alive && (off || P < 2)) {
        add = kStSOutOfRange;
        alive = false;
    }
    if (alive && sl == 0) {
        const Node m0 = node_at(line, idx0);
        const double ds = bs - sm[idx0];
        const double th0 = m0.theta + m0.kappa * ds;
        txy[0] = (m0.x + ds * cos(m0.theta)) + bl * (-sin(th0));
        txy[1] = (m0.y + ds * sin(m0.theta)) + bl * cos(th0);
    }
    // path points: count = leading points with s <= s_map[-1] (ref :40-41), index = running max of walks (:42-43)
    const double s_last = alive ? sm[P - 1] : 0.0;
    int carry = idx0, count = alive ? n : 0;
    bool stop = !alive;
    int nmax = 0;
#pragma unroll
    for (int g = 0; g < SPW; ++g) nmax = max(nmax, __builtin_amdgcn_readlane(alive ? n : 0, SL * g));
    constexpr unsigned long long kSceneMask = (SL == 32) ? 0xffffffffull : 0xffffull;
    for (int base = 0; base < nmax; base += SL) {
        const int i = base + sl;
        const bool in = !stop && i < n;
        const double s = in ? path_s[bb * max_pts + i] : 0.0;
        const unsigned bad16 = (unsigned)((__ballot(in && s > s_last) >> (SL * sc)) & kSceneMask);
        const int first_bad = __builtin_ffs((int)bad16);                               // 1-based lane of the scene, 0 if none
        if (!stop && first_bad) count = min(count, base + first_bad - 1);
        bool o2 = false;
        int k = in ? walk_from_zero(sm, P, fmin(s, s_last), &o2) : 0;
        for (int d = 1; d < SL; d <<= 1) {                 // inclusive running maximum across the scene's lanes
            const int v = __shfl_up(k, d, SL);
            if (sl >= d) k = max(k, v);
        }
        k = max(k, carry);
        carry = __shfl(k, SL - 1, SL);
        if (in && i < count && i + 1 < cap) {
            const Node mm = node_at(line, k);
            const double ds = s - sm[k];
            const double thp = mm.theta + mm.kappa * ds;
            const double l = path_l[bb * max_pts + i];
            txy[2 * (i + 1)] = (mm.x + ds * cos(mm.theta)) + l * (-sin(thp));       // ref :44-46
            txy[2 * (i + 1) + 1] = (mm.y + ds * sin(mm.theta)) + l * cos(thp);
        }
        if (first_bad) stop = true;
    }
    int m = count + 1;
    __syncthreads();
    if (alive && (m > cap || m > max_pts + 1 || m > GP * R)) {
        add = kStTruncated;
        alive = false;
    }
    if (alive && m < 2) {
        add = kStSmoothFailed;
        alive = false;
    }
    // smoothing (ref planning_utils.py:262-361): x on the scene's first GP lanes, y on the other GP, R points per lane
    const int grp = sl / GP, gl = sl & (GP - 1);
    double ref[R], u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = gl * R + r;
        const double v = txy[2 * (j < cap ? j : 0) + grp];
        ref[r] = (alive && j < m) ? v : 0.0;
    }
    int it = 0;
    int rc = box_qp_active_set_rows<GP, R>(ref, alive ? m : 0, grp ? sy : sx, u, &it);
    if (force_fallback && rc == 0) rc = -1;          // test hook (EMP_SMOOTH_FORCE_FALLBACK=1): every scene takes the fall-back
    const unsigned long long unsettled = __ballot(alive && rc < 0), wrong = __ballot(alive && rc > 0);
    const bool need_fb = ((unsettled >> (SL * sc)) & kSceneMask) != 0ull;
    if (((wrong >> (SL * sc)) & kSceneMask) != 0ull) {
        add = kStSmoothFailed;
        alive = false;
    }
    if (alive && !need_fb) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = gl * R + r;
            if (j < m) (grp ? py : px)[j] = u[r];
        }
    }
    __syncthreads();
    if (unsettled != 0ull) {             // wave-uniform; never taken on the test and benchmark scenes
        for (int s2 = 0; s2 < SPW; ++s2) {
            if (((unsettled >> (SL * s2)) & kSceneMask) == 0ull) continue;
            const int m2 = __builtin_amdgcn_readlane(m, SL * s2);
            double* base2 = lds + (size_t)s2 * per_scene;
            double *qx = nullptr, *qy = nullptr;
            int it2 = 0;
            const int rc2 = smooth_pair_wave<(GP * R > 32)>(qmem, base2 + max_ref, 2, m2, sx, sy, &qx, &qy, &it2);
            if (rc2 == 0) {
                for (int i = lane; i < m2; i += 64) {
                    base2[max_ref + 3 * cap + i] = qx[i];
                    base2[max_ref + 4 * cap + i] = qy[i];
                }
            } else if (sc == s2) {
                add = kStSmoothFailed;
                alive = false;
            }
            __syncthreads();
        }
    }
    // heading / curvature (ref planning_utils.py:185-228), one point per lane
    double* out_rows = traj + bb * (max_pts + 1) * 4;
    const int mm = alive ? m : 0;
    for (int i = sl; i < mm; i += SL) {
        const int a = (i - 1 > 0) ? i - 1 : 0, c = (i < mm - 2) ? i : mm - 2;
        const double dx = ((px[a + 1] - px[a]) + (px[c + 1] - px[c])) / 2.0;
        const double dy = ((py[a + 1] - py[a]) + (py[c + 1] - py[c])) / 2.0;
        th[i] = atan2(dy, dx);
    }
    __syncthreads();
    for (int i = sl; i < mm; i += SL) {
        const int a = (i - 1 > 0) ? i - 1 : 0, c = (i < mm - 2) ? i : mm - 2;
        const double dx = ((px[a + 1] - px[a]) + (px[c + 1] - px[c])) / 2.0;
        const double dy = ((py[a + 1] - py[a]) + (py[c + 1] - py[c])) / 2.0;
        const double dpre = th[a + 1] - th[a], daft = th[c + 1] - th[c];
        double* o = out_rows + (size_t)i * 4;
        o[0] = px[i];
        o[1] = py[i];
        o[2] = th[i];
        o[3] = sin((dpre + daft) / 2.0) / sqrt(dx * dx + dy * dy);
    }
    if (present) {
        for (int i = mm * 4 + sl; i < (max_pts + 1) * 4; i += SL) out_rows[i] = 0.0;       // padding reads as 0
        if (sl == 0) {
            traj_len[bb] = mm;
            if (add) status[bb] = st | add;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone forms of the projection helpers (one lane per scene, points in order)
// ---------------------------------------------------------------------------------------------
// ref: cal_s_map_fun, planning_utils.py:448-472
__global__ void s_map_kernel(int B, int max_ref, const double* __restrict__ ref_line, const int* __restrict__ n_ref,
                             const double* __restrict__ origin_xy, double* __restrict__ s_map) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    s_map_build(ref_line + (size_t)b * max_ref * 4, n_ref[b], origin_xy[2 * b], origin_xy[2 * b + 1],
                s_map + (size_t)b * max_ref);
}

// ref: cal_s_l_fun, planning_utils.py:475-509 (mode 0) and cal_projection_s_fun, :429-445 (mode 1: the caller
// supplies the match indices and only s is produced)
__global__ void s_l_kernel(int B, int max_ref, int max_pts, const double* __restrict__ ref_line,
                           const double* __restrict__ s_map, const int* __restrict__ n_ref,
                           const double* __restrict__ xy, const int* __restrict__ n_pts,
                           const int* __restrict__ match_in, double* __restrict__ out_s, double* __restrict__ out_l) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* line = ref_line + (size_t)b * max_ref * 4;
    const double* sm = s_map + (size_t)b * max_ref;
    const int P = min(max(n_ref[b], 0), max_ref), k = n_pts[b];
    int m_first = 0;
    for (int j = 0; j < k; ++j) {
        const size_t o = (size_t)b * max_pts + j;
        const double x = xy[o * 2], y = xy[o * 2 + 1];
        const int m = match_in ? match_in[o] : match_scan(line, P, x, y, 0, 1, 50);
        if (j == 0) m_first = m;
        out_s[o] = projection_s(node_at(line, m), sm[m], x, y);
        if (out_l) out_l[o] = lateral_offset(project_on(node_at(line, m_first), x, y), x, y);
    }
}

// ref: cal_s_l_deri_fun, planning_utils.py:512-588: out [B][max_pts][7] = l, dl/dt, ds/dt, d2l/dt2, dl/ds, d2s/dt2, d2l/ds2
__global__ void s_l_deri_kernel(int B, int max_ref, int max_pts, const double* __restrict__ ref_line,
                                const int* __restrict__ n_ref, const double* __restrict__ xy,
                                const double* __restrict__ vxy, const double* __restrict__ axy,
                                const int* __restrict__ n_pts, const double* __restrict__ origin_xy,
                                double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* line = ref_line + (size_t)b * max_ref * 4;
    const int P = min(max(n_ref[b], 0), max_ref), k = n_pts[b];
    int m_first = 0;
    for (int j = 0; j < k; ++j) {
        const size_t o = (size_t)b * max_pts + j;
        const double x = xy[o * 2], y = xy[o * 2 + 1];
        const int m = match_scan(line, P, x, y, 0, 1, 50);
        if (j == 0) m_first = m;
        const Node proj = project_on(node_at(line, m_first), x, y);
        const FrenetState f = frenet_state(proj, origin_xy[2 * b], origin_xy[2 * b + 1], vxy[o * 2], vxy[o * 2 + 1],
                                           axy[o * 2], axy[o * 2 + 1]);
        double* r = out + o * 7;
        r[0] = f.l; r[1] = f.l_dot; r[2] = f.s_dot; r[3] = f.l_ddot; r[4] = f.dl_ds; r[5] = f.s_ddot; r[6] = f.ddl_ds;
    }
}

// ref: cal_proj_point, path_planning.py:52-75: one query per lane; out [n][4], idx_out [n]; status 2 = IndexError
__global__ void proj_point_kernel(int n, int max_ref, const double* __restrict__ ref_line,
                                  const double* __restrict__ s_map, const int* __restrict__ n_ref,
                                  const double* __restrict__ s, const int* __restrict__ pre_idx,
                                  double* __restrict__ out, int* __restrict__ idx_out, int* __restrict__ status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int idx = pre_idx[t];
    Node pr{0, 0, 0, 0};
    const bool ok = idx >= 0 && proj_point(ref_line + (size_t)t * max_ref * 4, s_map + (size_t)t * max_ref, n_ref[t], s[t], &idx, &pr);
    out[4 * t] = pr.x; out[4 * t + 1] = pr.y; out[4 * t + 2] = pr.theta; out[4 * t + 3] = pr.kappa;
    idx_out[t] = idx;
    status[t] = ok ? 0 : kStSOutOfRange;
}

// ref: trajectory_index2s, planning_utils.py:758-780: cumulative chord length until the first NaN x
__global__ void index2s_kernel(int B, int max_pts, const double* __restrict__ x, const double* __restrict__ y,
                               const int* __restrict__ n_pts, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * max_pts;
    double acc = 0.0;
    for (int i = 0; i < n_pts[b]; ++i) out[o + i] = 0.0;
    for (int i = 1; i < n_pts[b]; ++i) {
        if (x[o + i] != x[o + i]) break;
        const double dx = x[o + i] - x[o + i - 1], dy = y[o + i] - y[o + i - 1];
        acc += sqrt(dx * dx + dy * dy);
        out[o + i] = acc;
    }
}

// ref: CalcProjPoint (planning_utils.py:736-755) + Frenet2Cartesian (:706-733): per point, NaN s stops the scene.
// line [B][max_ref][4], index2s [B][max_ref]; sl [B][max_pts][4] = s, l, dl, ddl -> out [B][max_pts][4] (NaN-filled)
__global__ void frenet2cartesian_kernel(int B, int max_ref, int max_pts, const double* __restrict__ ref_line,
                                        const double* __restrict__ index2s, const int* __restrict__ n_ref,
                                        const double* __restrict__ sl, const int* __restrict__ n_pts,
                                        double* __restrict__ out, int* __restrict__ status, int proj_only) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* line = ref_line + (size_t)b * max_ref * 4;
    const double* sm = index2s + (size_t)b * max_ref;
    const int P = min(max(n_ref[b], 0), max_ref);       // clamped to the row's capacity
    const double qnan = __builtin_nan("");
    int st = 0;
    bool stopped = false;
    for (int j = 0; j < n_pts[b]; ++j) {
        const double* v = sl + ((size_t)b * max_pts + j) * 4;
        double* o = out + ((size_t)b * max_pts + j) * 4;
        o[0] = o[1] = o[2] = o[3] = qnan;
        if (stopped || v[0] != v[0]) {                      // ref :718-719 break at the first NaN s
            stopped = true;
            continue;
        }
        int idx = 1;                                         // ref :742-744 starts at 1, first s_map[idx] >= s
        while (idx < P && sm[idx] < v[0]) ++idx;
        if (idx >= P) {
            st = kStSOutOfRange;                             // IndexError in the reference
            stopped = true;
            continue;
        }
        const Node m = node_at(line, idx);
        const double ds = v[0] - sm[idx];
        const double px = m.x + ds * cos(m.theta), py = m.y + ds * sin(m.theta);
        const double ph = m.theta + ds * m.kappa, pk = m.kappa;
        if (proj_only) {
            o[0] = px; o[1] = py; o[2] = ph; o[3] = pk;
            continue;
        }
        const double l = v[1], dl = v[2], ddl = v[3];
        o[0] = px + l * (-sin(ph));
        o[1] = py + l * cos(ph);
        const double hd = ph + atan(dl / (1.0 - pk * l));                                   // ref :727
        const double dth = hd - ph;
        o[2] = hd;
        o[3] = ((ddl + pk * dl * tan(dth)) * (cos(dth) * cos(dth)) / (1.0 - pk * l) + pk) * cos(dth) / (1.0 - pk * l);
    }
    status[b] = st;
}

// ref: cal_dy_obs_deri, planning_utils.py:783-808: in [n][5] = l, vx, vy, heading, kappa -> out [n][3] = s_dot, l_dot, dl
__global__ void dy_obs_deri_kernel(int n, const double* __restrict__ in, double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double l = in[5 * t], vx = in[5 * t + 1], vy = in[5 * t + 2], hd = in[5 * t + 3], k = in[5 * t + 4];
    const double l_dot = dot2(vx, vy, -sin(hd), cos(hd));                 // np.dot, :799-800
    const double s_dot = dot2(vx, vy, cos(hd), sin(hd)) / (1.0 - k * l);
    out[3 * t] = s_dot;
    out[3 * t + 1] = l_dot;
    out[3 * t + 2] = (fabs(s_dot) < 1e-6) ? 0.0 : l_dot / s_dot;
}

// small utilities ---------------------------------------------------------------------------------
// ref: cal_quintic_coefficient, planning_utils.py:671-703 - returns ABSOLUTE-s coefficients c0..c5 like the
// reference, computed from the closed form in the shifted coordinate (binomial re-expansion about s0).
__global__ void quintic_kernel(int n, const double* __restrict__ bc, double* __restrict__ coeff) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double* v = bc + 8 * t;
    const double l0 = v[0], dl0 = v[1], ddl0 = v[2], l1 = v[3], dl1 = v[4], ddl1 = v[5], s0 = v[6], T = v[7] - v[6];
    const double h = l1 - l0, T2 = T * T, T3 = T2 * T, T4 = T3 * T, T5 = T4 * T;
    double a[6];
    a[0] = l0;
    a[1] = dl0;
    a[2] = 0.5 * ddl0;
    a[3] = (20.0 * h - (8.0 * dl1 + 12.0 * dl0) * T - (3.0 * ddl0 - ddl1) * T2) / (2.0 * T3);
    a[4] = (-30.0 * h + (14.0 * dl1 + 16.0 * dl0) * T + (3.0 * ddl0 - 2.0 * ddl1) * T2) / (2.0 * T4);
    a[5] = (12.0 * h - 6.0 * (dl1 + dl0) * T - (ddl0 - ddl1) * T2) / (2.0 * T5);
    // p(s) = sum a_k (s - s0)^k  ->  sum c_j s^j  by repeated synthetic "shift" (Horner form in (s - s0))
    double c[6] = {a[5], 0, 0, 0, 0, 0};
    int deg = 0;
    for (int k = 4; k >= 0; --k) {               // c(s) <- c(s) * (s - s0) + a_k
        ++deg;
        for (int j = deg; j >= 1; --j) c[j] = c[j - 1] - s0 * c[j];
        c[0] = a[k] - s0 * c[0];                 // (descending j reads the not-yet-updated c[j-1])
    }
    for (int j = 0; j < 6; ++j) coeff[6 * t + j] = c[j];
}

// ref: cal_obs_cost, path_planning.py:588-609
__global__ void obs_cost_kernel(int n, int samples, double w, double danger, double safe, const double* __restrict__ sq,
                                double* __restrict__ cost) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double d2a = danger * danger, d2b = safe * safe;
    double c = 0.0;
    for (int i = 0; i < samples; ++i) {               // the reference loops over whatever it is handed (:601)
        const double v = sq[(size_t)samples * t + i];
        if (v <= d2a) {
            c = c + w;
            break;
        } else if (d2a < v && v < d2b) {
            c = c + kSoftGain / v;
        }
    }
    cost[t] = c;
}

// ref: cal_start_cost (path_planning.py:435-514) / cal_neighbor_cost (:517-585) for FREE edges: edges [n][8] = start s, l, dl, ddl,
// span (end s - start s), end l, sample_s, (unused).  The quintic ends at `end s` (the reference's cal_quintic_coefficient call, :475 / :553) while
// the ten samples step by sample_s / 10 from the start (:492-493 / :565-566) - the two coincide on the lattice and need not
// for a caller of the drop-in functions.  Obstacles per edge: obs_s, obs_l [n][max_obs], n_obs [n].
__global__ void free_edge_cost_kernel(int n, int max_obs, const double* __restrict__ edges, const double* __restrict__ obs_s,
                                      const double* __restrict__ obs_l, const int* __restrict__ n_obs, double w_coll, double w0,
                                      double w1, double w2, double w_ref, double* __restrict__ cost) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double* e = edges + (size_t)t * 8;
    const Quintic q = quintic_shifted(e[1], e[2], e[3], e[5], e[4]);
    const int k = max_obs > 0 ? min(max(n_obs[t], 0), max_obs) : 0;
    cost[t] = segment_cost(q, e[0], e[6], obs_s + (size_t)t * max_obs, obs_l + (size_t)t * max_obs, k, w_coll, w0, w1, w2, w_ref);
}

}  // namespace emp
