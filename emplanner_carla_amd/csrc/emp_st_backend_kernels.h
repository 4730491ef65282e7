// emp_st_backend_kernels.h - HIP kernels of the S-T speed planning back end (SURVEY.md section 8f row 2;
// reference planner/speed_planning_test.py:308-620).  Arithmetic in emp_st_backend_core.h.
//
//   convex_space_kernel   generate_convex_space (:308-407)  one scene per lane: 16 columns x a few obstacle slots of
//                         scalar, branchy work per scene
//   speed_qp_kernel       speed_QP (:410-511), the intended problem: one scene per group of 32 lanes, the banded
//                         range-QP interior point solver of the path QP (emp_qp_wave.h) on <= 16 B-spline coefficients
//   densify_kernel        increase_points (:514-566)         one wavefront per scene, 401 samples over the lanes
//   merge_kernel          path_speed_merge (:569-620)        one wavefront per scene, path arrays in LDS
#pragma once

#include <hip/hip_runtime.h>

#include "emp_qp_wave.h"
#include "emp_st_backend_core.h"

namespace emp {
namespace stb {

__global__ __launch_bounds__(64) void convex_space_kernel(
    int B, int n_slots, int max_path, double max_lateral_accel, const double* __restrict__ dp_s,
    const double* __restrict__ dp_t, const double* __restrict__ idx2s, const double* __restrict__ kappa,
    const int* __restrict__ path_len, const double* __restrict__ s_in, const double* __restrict__ s_out,
    const double* __restrict__ t_in, const double* __restrict__ t_out, double* __restrict__ s_lb,
    double* __restrict__ s_ub, double* __restrict__ sd_lb, double* __restrict__ sd_ub, int* __restrict__ status) {
    // the six 16-entry arrays of a lane are indexed at run time: LDS, not registers (a lane's block is 97 doubles
    // long, an odd stride, so the lanes of a wavefront spread over the banks)
    constexpr int kLaneWords = 6 * kDp + 1;
    __shared__ double lane_mem[64 * kLaneWords];
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double* mine = lane_mem + threadIdx.x * kLaneWords;
    double *lb = mine, *ub = mine + kDp, *vlb = mine + 2 * kDp, *vub = mine + 3 * kDp, *ds = mine + 4 * kDp, *dt = mine + 5 * kDp;
    for (int i = 0; i < kDp; ++i) {
        ds[i] = dp_s[(size_t)b * kDp + i];
        dt[i] = dp_t[(size_t)b * kDp + i];
    }
    int n = path_len[b];
    n = n < 0 ? 0 : (n > max_path ? max_path : n);
    const size_t po = (size_t)b * max_path, so = (size_t)b * n_slots;
    const int st = convex_space(ds, dt, idx2s + po, kappa + po, n, s_in + so, s_out + so, t_in + so, t_out + so, n_slots,
                                max_lateral_accel, lb, ub, vlb, vub);
    const double nan = __builtin_nan("");
    for (int i = 0; i < kDp; ++i) {                     // the reference raises: nothing is returned
        s_lb[(size_t)b * kDp + i] = st ? nan : lb[i];
        s_ub[(size_t)b * kDp + i] = st ? nan : ub[i];
        sd_lb[(size_t)b * kDp + i] = st ? nan : vlb[i];
        sd_ub[(size_t)b * kDp + i] = st ? nan : vub[i];
    }
    status[b] = st;
}

// dynamic LDS: (64 / G) groups x (speed_qp_words(kQp) + 1) doubles
template <int G>
__global__ __launch_bounds__(64) void speed_qp_kernel(int B, SpeedQpParams prm, const double* __restrict__ v0,
                                                      const double* __restrict__ a0, const double* __restrict__ dp_s,
                                                      const double* __restrict__ dp_t, const double* __restrict__ s_lb,
                                                      const double* __restrict__ s_ub, const double* __restrict__ sd_lb,
                                                      const double* __restrict__ sd_ub, double* __restrict__ qs,
                                                      double* __restrict__ qv, double* __restrict__ qa,
                                                      double* __restrict__ qt, int* __restrict__ iters,
                                                      int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    constexpr int GPW = 64 / G;
    const int lane = threadIdx.x & 63, grp = lane / G, gl = lane & (G - 1);
    const int b = blockIdx.x * GPW + grp;
    const bool present = b < B;
    double* lds = lds_all + (size_t)grp * (speed_qp_words(kQp) + 1);
    const size_t o16 = (size_t)(present ? b : 0) * kDp, o17 = (size_t)(present ? b : 0) * kQp;
    const int nq = present ? speed_qp_size(dp_s + o16) : -kStbQpFailed;
    bool live = nq >= 2;
    const int n = live ? nq : 2;
    const double dt = live ? dp_t[o16 + n - 1] / (double)(n - 1) : 1.0;       // ref :437, :449
    SpeedRangeQp Q;
    double* cc = lds;
    Q.bind(lds + kQp + 2, live ? n - 1 : 0, live ? n - 1 : 0);
    speed_qp_forms(Q, dt);
    int* flag = reinterpret_cast<int*>(lds + speed_qp_words(kQp));            // set-up result of lane 0
    if (gl == 0) {
        int rc0 = 0;
        if (live) rc0 = speed_qp_setup(Q, cc, n, dt, v0[b], a0[b], s_lb + o16, s_ub + o16, sd_lb + o16, sd_ub + o16, prm);
        *flag = rc0;
    }
    __syncthreads();
    int fail = live ? 0 : -nq;
    if (live && *flag) {
        fail = kStbQpFailed;
        live = false;
    }
    // ---- start from the unconstrained minimiser P u = -q
    {
        double fa[4], flow[4], frinv = 0.0;
#pragma unroll
        for (int d = 0; d < 4; ++d) fa[d] = (live && gl < Q.N) ? Q.P[gl * 4 + d] : 0.0;
        const bool okc = band_chol_group<G, 3>(fa, frinv, flow, Q.N, gl, live && Q.N > 0);
        double b0 = (live && gl < Q.N) ? -Q.q[gl] : 0.0;
        band_solve_group<G, 3>(fa, frinv, flow, b0, Q.N, gl);
        if (live && gl < Q.N) Q.u[gl] = b0;
        if (live && !okc) {
            fail = kStbQpFailed;
            live = false;
        }
    }
    __syncthreads();
    const int rs = range_qp_solve_wave_fast<G>(Q, gl, live && Q.N > 0, 1000);
    if (live && rs) {
        fail = kStbQpFailed;
        live = false;
    }
    for (int m = gl; m < (live ? Q.N : 0); m += G) cc[m + 3] = Q.u[m];
    __syncthreads();
    if (present) {
        const double nan = __builtin_nan("");
        for (int i = gl; i < kQp; i += G) {
            const bool in = live && i < n;
            qs[o17 + i] = in ? (cc[i] + 4.0 * cc[i + 1] + cc[i + 2]) / 6.0 : nan;
            qv[o17 + i] = in ? (cc[i + 2] - cc[i]) / (2.0 * dt) : nan;
            qa[o17 + i] = in ? (cc[i] - 2.0 * cc[i + 1] + cc[i + 2]) / (dt * dt) : nan;
            qt[o17 + i] = in ? (double)i * dt : nan;
        }
        if (gl == 0) {
            status[b] = fail;
            if (iters) iters[b] = live ? Q.iters : 0;
        }
    }
}

__global__ __launch_bounds__(64) void densify_kernel(int B, const double* __restrict__ qs, const double* __restrict__ qv,
                                                     const double* __restrict__ qa, const double* __restrict__ qt,
                                                     double* __restrict__ s, double* __restrict__ v,
                                                     double* __restrict__ a, double* __restrict__ t,
                                                     int* __restrict__ status) {
    __shared__ double ls[kQp], lv[kQp], la[kQp], lt[kQp];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    if (lane < kQp) {
        ls[lane] = qs[(size_t)b * kQp + lane];
        lv[lane] = qv[(size_t)b * kQp + lane];
        la[lane] = qa[(size_t)b * kQp + lane];
        lt[lane] = qt[(size_t)b * kQp + lane];
    }
    __syncthreads();
    const int t_end = dense_t_end(lt);
    const size_t o = (size_t)b * kDense;
    if (t_end >= kQp || t_end < 0) {                       // relative_time_init[17] / an empty profile
        const double nan = __builtin_nan("");
        for (int i = lane; i < kDense; i += 64) s[o + i] = v[o + i] = a[o + i] = t[o + i] = nan;
        if (lane == 0) status[b] = t_end < 0 ? kStbNoProfile : kStbIndex;
        return;
    }
    const double dt = lt[t_end] / (double)(kDense - 1);    // ref :540-542
    int carry = 0;                                         // ref :549 tmp = 0; it keeps its value when no interval matches
    for (int base = 0; base < kDense; base += 64) {
        const int i = base + lane;
        const double cur = (double)(i - 1) * dt;           // ref :551 - the first sample lies at -dt
        int m = (i < kDense) ? dense_match(lt, t_end, cur) : -1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {           // "last match so far" = running maximum (matches ascend with t)
            const int o2 = __shfl_up(m, off, 64);
            if (lane >= off) m = max(m, o2);
        }
        m = max(m, carry);
        carry = __shfl(m, 63, 64);
        if (i < kDense) {
            double ss, vv, aa;
            dense_sample(ls, lv, la, lt, m, cur, &ss, &vv, &aa);
            s[o + i] = ss;
            v[o + i] = vv;
            a[o + i] = aa;
            t[o + i] = cur;
        }
    }
    if (lane == 0) status[b] = 0;
}

// dynamic LDS: 5 * max_path doubles.  out [B][7][401]: x, y, heading, kappa, speed, accel, time
__global__ __launch_bounds__(64) void merge_kernel(int B, int max_path, const double* __restrict__ s,
                                                   const double* __restrict__ v, const double* __restrict__ a,
                                                   const double* __restrict__ t, const double* __restrict__ now,
                                                   const double* __restrict__ path_s, const double* __restrict__ x_init,
                                                   const double* __restrict__ y_init, const double* __restrict__ h_init,
                                                   const double* __restrict__ k_init, const int* __restrict__ n_init,
                                                   double* __restrict__ out, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    double* ps = lds;
    double* px = ps + max_path;
    double* py = px + max_path;
    double* ph = py + max_path;
    double* pk = ph + max_path;
    int width = n_init[b];
    width = width < 0 ? 0 : (width > max_path ? max_path : width);
    const size_t po = (size_t)b * max_path;
    for (int i = lane; i < width; i += 64) {
        ps[i] = path_s[po + i];
        px[i] = x_init[po + i];
        py[i] = y_init[po + i];
        ph[i] = h_init[po + i];
        pk[i] = k_init[po + i];
    }
    __syncthreads();
    // ref :584-587: the first NaN of trajectory_x_init, minus one (and the slices [:index] drop one more point)
    int first_nan = width;
    for (int base = 0; base < width; base += 64) {
        const int i = base + lane;
        const unsigned long long m = __ballot(i < width && px[i] != px[i]);
        if (m) {
            first_nan = base + __builtin_ffsll((long long)m) - 1;
            break;
        }
    }
    const int index = first_nan - 1;
    double* ob = out + (size_t)b * 7 * kDense;
    int st = 0;
    if (first_nan >= width) st = kStbIndex;               // no NaN: the reference's scan runs off the array
    else if (index < 0) st = kStbNoProfile;               // an empty trajectory
    else if (index == 0) st = kStbRange;                  // np.interp on empty arrays raises ValueError
    if (st) {
        const double nan = __builtin_nan("");
        for (int i = lane; i < 7 * kDense; i += 64) ob[i] = nan;
        if (lane == 0) status[b] = st;
        return;
    }
    const size_t o = (size_t)b * kDense;
    const double now_b = now[b];
    for (int i = lane; i < kDense; i += 64) {
        double rx, ry, rh, rk;
        if (i < kDense - 1) {
            const double x = s[o + i];
            const int j = (x != x) ? 0 : np_interp_index(ps, index, x);
            rx = np_interp_at(ps, px, index, j, x);
            ry = np_interp_at(ps, py, index, j, x);
            rh = np_interp_at(ps, ph, index, j, x);
            rk = np_interp_at(ps, pk, index, j, x);
        } else {                                           // ref :608-611: the arrays' LAST slots (NaN when padded)
            rx = px[width - 1];
            ry = py[width - 1];
            rh = ph[width - 1];
            rk = pk[width - 1];
        }
        ob[0 * kDense + i] = rx;
        ob[1 * kDense + i] = ry;
        ob[2 * kDense + i] = rh;
        ob[3 * kDense + i] = rk;
        ob[4 * kDense + i] = v[o + i];
        ob[5 * kDense + i] = a[o + i];
        ob[6 * kDense + i] = t[o + i] + now_b;
    }
    if (lane == 0) status[b] = 0;
}

}  // namespace stb
}  // namespace emp
