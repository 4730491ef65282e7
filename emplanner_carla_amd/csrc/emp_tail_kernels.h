// emp_tail_kernels.h - everything of one planning cycle around the DP: Cartesian->Frenet projection,
// QP bounds, path QP, midpoint re-interleave, Frenet->Cartesian, smoothing QP, heading/curvature.
//
// Mapping: one scene per lane (the per-scene work is a chain of short sequential recurrences: banded
// Cholesky, monotone index walks, ordered scans with early exits).  State lives in private arrays.
// "ref:" cites the reference (paths relative to the reference tree).
#pragma once

#include <hip/hip_runtime.h>

#include "emp_core.h"
#include "emp_frenet_core.h"
#include "emp_qp_core.h"

namespace emp {

// status bits (mirror include/emplanner.h)
constexpr int kStDpInfeasible = 1, kStSOutOfRange = 2, kStBoundIndex = 4, kStQpFailed = 8, kStSmoothFailed = 16,
              kStTruncated = 32;

struct QpDev {
    PathQpParams qp;
    double obs_length, obs_width;
    int decimate, midpoint, use_qp;
};

// ---------------------------------------------------------------------------------------------
// ref: test_9.py:113-177 - s_map, obstacle (s, l), planning-start (s, l) and (l, dl/ds, d2l/ds2)
// ---------------------------------------------------------------------------------------------
__global__ void frenet_project_kernel(int B, int max_ref, int max_obs, const double* __restrict__ ref_line,
                                      const int* __restrict__ n_ref, const double* __restrict__ origin_xy,
                                      const double* __restrict__ start_xy, const double* __restrict__ start_v,
                                      const double* __restrict__ start_a, const double* __restrict__ obs_xy,
                                      const int* __restrict__ n_obs, double* __restrict__ s_map,
                                      double* __restrict__ obs_s, double* __restrict__ obs_l,
                                      double* __restrict__ begin_sl, double* __restrict__ start) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* line = ref_line + (size_t)b * max_ref * 4;
    const int P = n_ref[b];
    double* sm = s_map + (size_t)b * max_ref;
    s_map_build(line, P, origin_xy[2 * b], origin_xy[2 * b + 1], sm);            // ref :113
    // obstacles (ref :122): s from each point's own match, l from the projection on the FIRST point's match
    const int k = n_obs ? n_obs[b] : 0;
    int m_first = 0;
    for (int j = 0; j < k; ++j) {
        const double x = obs_xy[((size_t)b * max_obs + j) * 2], y = obs_xy[((size_t)b * max_obs + j) * 2 + 1];
        const int m = match_scan(line, P, x, y, 0, 1, 50);
        if (j == 0) m_first = m;
        obs_s[(size_t)b * max_obs + j] = projection_s(node_at(line, m), sm[m], x, y);
        obs_l[(size_t)b * max_obs + j] = lateral_offset(project_on(node_at(line, m_first), x, y), x, y);
    }
    // planning start (ref :134 and :172-177; single-point lists, so "first match" is its own)
    const double px = start_xy[2 * b], py = start_xy[2 * b + 1];
    const int m = match_scan(line, P, px, py, 0, 1, 50);
    const Node proj = project_on(node_at(line, m), px, py);
    const double bs = projection_s(node_at(line, m), sm[m], px, py);
    if (begin_sl) {
        begin_sl[2 * b] = bs;
        begin_sl[2 * b + 1] = lateral_offset(proj, px, py);
    }
    const FrenetState fs = frenet_state(proj, px, py, start_v[2 * b], start_v[2 * b + 1], start_a[2 * b], start_a[2 * b + 1]);
    start[4 * b + 0] = bs;
    start[4 * b + 1] = fs.l;
    start[4 * b + 2] = fs.dl_ds;
    start[4 * b + 3] = fs.ddl_ds;
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
    const int P = n_ref[b];
    const int k = n_pts[b];
    int m_first = 0;
    for (int j = 0; j < k; ++j) {
        const double x = xy[((size_t)b * max_pts + j) * 2], y = xy[((size_t)b * max_pts + j) * 2 + 1];
        int m;
        if (!windowed_api || is_first_run[b]) {
            m = match_scan(line, P, x, y, 0, 1, 50);                               // ref planning_utils.py:72-92 / :383-402
        } else {
            const int st = pre_match_index[b];                                     // ref :123-167
            const Node pm = node_at(line, st);
            const double flag = (x - pm.x) * cos(pm.theta) + (y - pm.y) * sin(pm.theta);
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
                              obs_s + (size_t)b * max_obs, obs_l + (size_t)b * max_obs, n_obs[b], obs_length,
                              obs_width, l_min + (size_t)b * max_pts, l_max + (size_t)b * max_pts);
    status[b] = ok ? 0 : kStBoundIndex;
}

// ---------------------------------------------------------------------------------------------
// ref: Quadratic_planning, path_planning.py:78-219 (stand-alone stage)
// ---------------------------------------------------------------------------------------------
template <int NMAX>
__global__ void path_qp_kernel(int B, int max_pts, QpDev Q, const double* __restrict__ l_min,
                               const double* __restrict__ l_max, const int* __restrict__ n_pts,
                               const double* __restrict__ start_l3, double* __restrict__ qp_l,
                               double* __restrict__ qp_dl, double* __restrict__ qp_ddl, int* __restrict__ iters,
                               int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    PathQp<NMAX> qp;
    const size_t o = (size_t)b * max_pts;
    const int rc = qp.solve(l_min + o, l_max + o, n_pts[b], start_l3[3 * b], start_l3[3 * b + 1], start_l3[3 * b + 2],
                            Q.qp, qp_l + o, qp_dl + o, qp_ddl + o);
    if (iters) iters[b] = qp.iters;
    status[b] = rc ? kStQpFailed : 0;
}

// ---------------------------------------------------------------------------------------------
// ref: smooth_reference_line, planning_utils.py:262-361 (stand-alone stage): two lanes per scene (x, y)
// for the QP, then lane 0 of the pair computes heading / curvature.
// ---------------------------------------------------------------------------------------------
template <int MMAX>
__global__ void smooth_kernel(int B, int max_pts, SmoothQpParams sx, SmoothQpParams sy, const double* __restrict__ xy,
                              const int* __restrict__ n_pts, double* __restrict__ out, int* __restrict__ iters,
                              int* __restrict__ status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t >> 1, c = t & 1;
    if (b >= B) return;
    const int m = n_pts[b];
    BoxQp<MMAX> qp;
    const int rc = qp.solve(xy + (size_t)b * max_pts * 2 + c, 2, m, c ? sy : sx);
    double* o = out + (size_t)b * max_pts * 4;
    if (rc == 0)
        for (int i = 0; i < m; ++i) o[4 * i + c] = qp.x[i];
    if (rc) atomicOr(&status[b], kStSmoothFailed);
    if (iters && c == 0) iters[b] = qp.iters;
}

// heading / kappa of the smoothed points, in place in out[b][i][0..3] (ref planning_utils.py:357-360)
__global__ void traj_heading_kernel(int B, int max_pts, const int* __restrict__ n_pts, double* __restrict__ out,
                                    const int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int m = n_pts[b];
    if (m < 2 || (status[b] & (kStSmoothFailed | kStQpFailed | kStBoundIndex | kStSOutOfRange))) return;
    double* o = out + (size_t)b * max_pts * 4;
    heading_kappa(o, 4, m, o + 2, 4, o + 3, 4);
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
// Inputs: densified DP path; outputs path_s/path_l (n+1 points, or n without the midpoint step).
// ---------------------------------------------------------------------------------------------
template <int NMAX>
__global__ void cycle_qp_kernel(int B, int max_pts, int max_obs, QpDev Q, const double* __restrict__ dp_s,
                                const double* __restrict__ dp_l, const int* __restrict__ dp_len,
                                const double* __restrict__ obs_s, const double* __restrict__ obs_l,
                                const int* __restrict__ n_obs, const double* __restrict__ start,
                                double* __restrict__ path_s, double* __restrict__ path_l, int* __restrict__ path_len,
                                int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * max_pts;
    const int ne = dp_len[b];
    const int dec = Q.decimate > 0 ? Q.decimate : 1;
    const int n = (ne + dec - 1) / dec;                                            // len(x[::dec])
    double* ps = path_s + o;
    double* pl = path_l + o;
    int st = status[b];
    path_len[b] = 0;
    if (n > NMAX || n + (Q.midpoint ? 1 : 0) > max_pts) {
        status[b] = st | kStTruncated;
        return;
    }
    double ql[NMAX], qdl[NMAX], qddl[NMAX];
    if (Q.use_qp) {
        double l_min[NMAX], l_max[NMAX];
        if (!lmin_lmax(dp_s + o, dp_l + o, dec, n, obs_s + (size_t)b * max_obs, obs_l + (size_t)b * max_obs,
                       n_obs[b], Q.obs_length, Q.obs_width, l_min, l_max)) {
            status[b] = st | kStBoundIndex;
            return;
        }
        PathQp<NMAX> qp;
        const int rc = qp.solve(l_min, l_max, n, start[4 * b + 1], start[4 * b + 2], start[4 * b + 3], Q.qp, ql, qdl, qddl);
        if (rc) {
            status[b] = st | kStQpFailed;
            return;
        }
    } else {
        for (int i = 0; i < n; ++i) ql[i] = dp_l[o + (size_t)i * dec];
    }
    if (Q.midpoint) {                                                              // ref test_9.py:204-210
        ps[0] = dp_s[o];
        pl[0] = ql[0];
        for (int i = 1; i < n; ++i) {
            ps[i] = (dp_s[o + (size_t)i * dec] + dp_s[o + (size_t)(i - 1) * dec]) / 2.0;
            pl[i] = (ql[i] + ql[i - 1]) / 2.0;
        }
        ps[n] = dp_s[o + (size_t)(n - 1) * dec];
        pl[n] = ql[n - 1];
        path_len[b] = n + 1;
    } else {
        for (int i = 0; i < n; ++i) {
            ps[i] = dp_s[o + (size_t)i * dec];
            pl[i] = ql[i];
        }
        path_len[b] = n;
    }
    status[b] = st;
}

// One cycle, last part (ref path_planning.py:15-49): Frenet->Cartesian, then x / y smoothing on two lanes.
template <int MMAX>
__global__ void cycle_cartesian_kernel(int B, int max_ref, int max_pts, SmoothQpParams sx, SmoothQpParams sy,
                                       const double* __restrict__ ref_line, const double* __restrict__ s_map,
                                       const int* __restrict__ n_ref, const double* __restrict__ begin_sl,
                                       const double* __restrict__ path_s, const double* __restrict__ path_l,
                                       const int* __restrict__ path_len, double* __restrict__ traj,
                                       int* __restrict__ traj_len, int* __restrict__ status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t >> 1, c = t & 1;
    if (b >= B) return;
    const int st = status[b];
    if (c == 0) traj_len[b] = 0;
    if (st & (kStQpFailed | kStBoundIndex | kStTruncated)) return;
    // both lanes of the pair redo the cheap Frenet->Cartesian walk and keep only their own coordinate
    double txy[2 * MMAX];
    bool s_err, trunc;
    const int cap = (max_pts + 1 < MMAX) ? max_pts + 1 : MMAX;
    const int m = frenet_path_to_xy(ref_line + (size_t)b * max_ref * 4, s_map + (size_t)b * max_ref, n_ref[b],
                                    begin_sl[2 * b], begin_sl[2 * b + 1], path_s + (size_t)b * max_pts,
                                    path_l + (size_t)b * max_pts, path_len[b], txy, cap, &s_err, &trunc);
    if (s_err || trunc || m < 2) {
        if (c == 0) atomicOr(&status[b], s_err ? kStSOutOfRange : (trunc ? kStTruncated : kStSmoothFailed));
        return;
    }
    BoxQp<MMAX> qp;
    const int rc = qp.solve(txy + c, 2, m, c ? sy : sx);
    if (rc) {
        atomicOr(&status[b], kStSmoothFailed);
        return;
    }
    double* o = traj + (size_t)b * (max_pts + 1) * 4;
    for (int i = 0; i < m; ++i) o[4 * i + c] = qp.x[i];
    if (c == 0) traj_len[b] = m;
}

__global__ void cycle_heading_kernel(int B, int max_pts, int* __restrict__ traj_len, double* __restrict__ traj,
                                     int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int m = traj_len[b];
    if (status[b] & kStSmoothFailed) {
        traj_len[b] = 0;   // one coordinate failed after the other lane had set the length
        return;
    }
    if (m < 2) return;
    double* o = traj + (size_t)b * (max_pts + 1) * 4;
    heading_kappa(o, 4, m, o + 2, 4, o + 3, 4);
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
    const int P = n_ref[b], k = n_pts[b];
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
    const int P = n_ref[b], k = n_pts[b];
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
    const int P = n_ref[b];
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
    const double l_dot = vx * (-sin(hd)) + vy * cos(hd);
    const double s_dot = (vx * cos(hd) + vy * sin(hd)) / (1.0 - k * l);
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
__global__ void obs_cost_kernel(int n, double w, double danger, double safe, const double* __restrict__ sq,
                                double* __restrict__ cost) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double d2a = danger * danger, d2b = safe * safe;
    double c = 0.0;
    for (int i = 0; i < kSamples; ++i) {
        const double v = sq[kSamples * t + i];
        if (v <= d2a) {
            c = c + w;
            break;
        } else if (d2a < v && v < d2b) {
            c = c + kSoftGain / v;
        }
    }
    cost[t] = c;
}

}  // namespace emp
