// emp_mpc_kernels.h - batched lateral MPC controller (the controller input side).
// ref: controller/controller.py class Lateral_MPC_controller, :65-337 - `_control` from explicit inputs (the
// reference reads x, y, yaw, velocities from a live carla.Vehicle in cal_vehicle_info, :90-113).
//
// Mapping: one vehicle per GROUP of 12 lanes - lane r owns control r of the condensed problem (N = 6 steps x P = 2
// controls per step, :72-73) - and five vehicles per wavefront.  The small per-vehicle algebra (4x4 model,
// bilinear discretisation, 50-point nearest-point window, powers of A_bar) is computed redundantly by the 12
// lanes of a group; lane r then builds row r of the dense 12 x 12 Hessian and the box QP (|u| <= 1, :300-304) is
// solved by the same Mehrotra interior point as the planner's QPs with a dense Cholesky whose pivot rows travel
// by ds_bpermute inside the group.
#pragma once

#include <hip/hip_runtime.h>

#include "emp_qp_wave.h"
#include "emp_tail_kernels.h"   // status bits

namespace emp {
namespace mpc {

constexpr int kN = 6, kP = 2, kNu = kN * kP;     // ref :72-73
constexpr int kGroupsPerWave = 5;
constexpr int kWindow = 50;                      // ref :204
constexpr double kTs = 0.1;                      // ref :159 and _control's cal_error_k_fun(ts=0.1), :333

struct Params {
    double a, b, Cf, Cr, m, Iz;                  // vehicle_para (ref test_9.py:316)
    double q[4], f[4], r;                        // Q, F diagonals and R (ref :321-328)
};

struct V4 { double v[4]; };
struct M4 { double a[4][4]; };

__device__ __forceinline__ V4 matvec(const M4& A, const V4& x) {
    V4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y.v[i] = ((A.a[i][0] * x.v[0] + A.a[i][1] * x.v[1]) + A.a[i][2] * x.v[2]) + A.a[i][3] * x.v[3];
    return y;
}

// inverse of a 4 x 4 matrix by Gauss-Jordan with partial pivoting (the reference calls np.linalg.inv, :161)
__device__ inline bool inverse4(const M4& A, M4* out) {
    double w[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[i][j] = A.a[i][j];
            w[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        double best = fabs(w[c][c]);
#pragma unroll
        for (int r = c + 1; r < 4; ++r)
            if (fabs(w[r][c]) > best) {
                best = fabs(w[r][c]);
                piv = r;
            }
        if (!(best > 0.0)) return false;
#pragma unroll
        for (int r = c + 1; r < 4; ++r)
            if (r == piv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double t = w[c][j];
                    w[c][j] = w[r][j];
                    w[r][j] = t;
                }
            }
        const double inv = 1.0 / w[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[c][j] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double m = w[r][c];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[r][j] -= m * w[c][j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out->a[i][j] = w[i][4 + j];
    return true;
}

// value of lane `k` of MY 12-lane group
__device__ __forceinline__ double grp_bcast(double v, int group_base, int k) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    const int src = (group_base + k) << 2;
    r.i[0] = __builtin_amdgcn_ds_bpermute(src, a.i[0]);
    r.i[1] = __builtin_amdgcn_ds_bpermute(src, a.i[1]);
    return r.d;
}
template <class Op>
__device__ __forceinline__ double grp_reduce(double v, int group_base, Op op) {
    double acc = grp_bcast(v, group_base, 0);
#pragma unroll
    for (int k = 1; k < kNu; ++k) acc = op(acc, grp_bcast(v, group_base, k));
    return acc;
}

// Dense box QP  min 1/2 u'Hu + f'u, -1 <= u <= 1  on one 12-lane group: lane r holds row r of H (h[]) and f_r.
// Same interior point as smooth_pair_lanes (emp_qp_wave.h): G = I, so a row IS its unknown.
// Returns 0 ok / 2 failed (group-uniform); u_out = this lane's control.
__device__ inline int box_qp_full12(const double (&h)[kNu], double f_r, int r, int gb, bool live, double* u_out, int* iters_out) {
    const double eps_p = 1e-10, eps_mu = 1e-13, eps_d_rel = 1e-10;
    const double lo = -1.0, hi = 1.0;
    double u = 0.0, su = 1.0, sl = 1.0, zu = 1.0, zl = 1.0;          // slacks of the centre of the box are already 1
    const double qscale = fmax(1.0, grp_reduce(fabs(f_r), gb, [](double a, double b) { return fmax(a, b); }));
    int state = live ? 1 : 0, iters = 0;
    bool acceptable = false;
    const int rows = 2 * kNu;
    while (__any(state == 1)) {
        const bool run = state == 1;
        const double rpu = u - hi + su, rpl = lo - u + sl;
        const double isu = fast_rcp(su), isl = fast_rcp(sl), izu = fast_rcp(zu), izl = fast_rcp(zl);
        const double wu = zu * isu, wl = zl * isl;
        double hu = 0.0;
#pragma unroll
        for (int c = 0; c < kNu; ++c) hu = __builtin_fma(h[c], grp_bcast(u, gb, c), hu);
        const double rd = (hu + f_r) + (zu - zl);
        const double rd_max = grp_reduce(fabs(rd), gb, [](double a, double b) { return fmax(a, b); });
        const double rp_max = grp_reduce(fmax(fabs(rpu), fabs(rpl)), gb, [](double a, double b) { return fmax(a, b); });
        const double zmax = grp_reduce(fmax(zu, zl), gb, [](double a, double b) { return fmax(a, b); });
        const double mu = grp_reduce(su * zu + sl * zl, gb, [](double a, double b) { return a + b; }) / (double)rows;
        if (run) {
            const double dscale = fmax(qscale, zmax);
            if (rd_max <= eps_d_rel * dscale && rp_max <= eps_p && mu <= eps_mu) state = 0;
            else if (!(mu == mu) || mu > 1e30 || (iters >= kQpStallIter && rp_max > kQpStallResidual)) state = 2;
            else if (iters >= kQpMaxIter) state = acceptable ? 0 : 2;
            if (rd_max <= 100.0 * eps_d_rel * dscale && rp_max <= 10.0 * eps_p && mu <= 1000.0 * eps_mu) acceptable = true;
        }
        const bool go = state == 1;
        // ---- dense Cholesky of M = H + diag(wu + wl): lane r keeps row r; after step k its entry k is L[r][k]
        // (r > k) and lane k's entries j > k are U[k][j] = L[j][k]
        double a[kNu], rinv = 1.0;
#pragma unroll
        for (int c = 0; c < kNu; ++c) a[c] = go ? h[c] : ((c == r) ? 1.0 : 0.0);
#pragma unroll
        for (int c = 0; c < kNu; ++c)
            if (c == r && go) a[c] += wu + wl;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < kNu; ++k) {
            double rowk[kNu];
#pragma unroll
            for (int j = k; j < kNu; ++j) rowk[j] = grp_bcast(a[j], gb, k);
            const double piv = rowk[k];
            if (!(piv > 0.0)) bad = true;
            const double rs = fast_rsqrt(piv > 0.0 ? piv : 1.0);
            const double lik = a[k] * rs;                       // column-k entry of my row, scaled (symmetry: = U[k][r])
#pragma unroll
            for (int j = k + 1; j < kNu; ++j) {
                const double ukj = rowk[j] * rs;
                if (r > k) a[j] = __builtin_fma(-lik, ukj, a[j]);
                else if (r == k) a[j] = ukj;
            }
            if (r >= k) a[k] = (r == k) ? piv * rs : lik;
            if (r == k) rinv = rs;
        }
        if (go && bad) state = acceptable ? 0 : 2;
        const bool go2 = state == 1;
        auto solve = [&](double b) {
#pragma unroll
            for (int k = 0; k < kNu; ++k) {                     // L y = b
                const double yk = grp_bcast(b * rinv, gb, k);
                if (r == k) b = yk;
                else if (r > k) b = __builtin_fma(-a[k], yk, b);
            }
#pragma unroll
            for (int k = kNu - 1; k >= 0; --k) {                // L' x = y
                const double xk = grp_bcast(b * rinv, gb, k);
                if (r == k) b = xk;
                else if (r < k) b = __builtin_fma(-a[k], xk, b);
            }
            return b;
        };
        const double dua = solve(go2 ? -rd - ((wu * rpu - zu) - (wl * rpl - zl)) : 0.0);
        const double dsua = -rpu - dua, dsla = -rpl + dua;
        const double dzua = -zu - wu * dsua, dzla = -zl - wl * dsla;
        double ratio = go2 ? fmax(fmax(-dsua * isu, -dsla * isl), fmax(-dzua * izu, -dzla * izl)) : 0.0;
        ratio = grp_reduce(ratio, gb, [](double x, double y) { return fmax(x, y); });
        const double a_aff = (ratio > 1.0) ? fast_rcp(ratio) : 1.0;
        double mu_aff = go2 ? (su + a_aff * dsua) * (zu + a_aff * dzua) + (sl + a_aff * dsla) * (zl + a_aff * dzla) : 0.0;
        mu_aff = grp_reduce(mu_aff, gb, [](double x, double y) { return x + y; }) / (double)rows;
        double sigma = (mu > 0.0) ? mu_aff * fast_rcp(mu) : 0.0;
        sigma = sigma * sigma * sigma;
        const double rcu = su * zu + dsua * dzua - sigma * mu, rcl = sl * zl + dsla * dzla - sigma * mu;
        const double du = solve(go2 ? -rd - ((zu * rpu - rcu) * isu - (zl * rpl - rcl) * isl) : 0.0);
        const double dsu = -rpu - du, dsl = -rpl + du;
        const double dzu = -(rcu + zu * dsu) * isu, dzl = -(rcl + zl * dsl) * isl;
        ratio = go2 ? fmax(fmax(-dsu * isu, -dsl * isl), fmax(-dzu * izu, -dzl * izl)) : 0.0;
        ratio = grp_reduce(ratio, gb, [](double x, double y) { return fmax(x, y); });
        const double tau = qp_step_fraction(mu);
        const double alpha = (ratio > tau) ? tau * fast_rcp(ratio) : 1.0;
        if (go2) {
            su += alpha * dsu;
            sl += alpha * dsl;
            zu += alpha * dzu;
            zl += alpha * dzl;
            u += alpha * du;
            ++iters;
        }
    }
    *u_out = u;
    *iters_out = iters;
    return state;
}

// ref: Lateral_MPC_controller._control (:313-337) for B vehicles; grid = ceil(B / 5), block = 64.
__global__ __launch_bounds__(64) void mpc_lateral_kernel(int B, int max_path, Params prm, const double* __restrict__ target_path,
                                                         const int* __restrict__ n_path, const double* __restrict__ state,
                                                         const double* __restrict__ vx, const int* __restrict__ min_index_in,
                                                         double* __restrict__ steer, double* __restrict__ u_out,
                                                         double* __restrict__ e_rr_out, double* __restrict__ k_r_out,
                                                         int* __restrict__ min_index_out, double* __restrict__ pre_pro,
                                                         double* __restrict__ H_out, double* __restrict__ f_out,
                                                         int* __restrict__ iters_out, int* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int grp = lane / kNu, r = lane - grp * kNu;
    const int gb = grp * kNu;
    const int b = blockIdx.x * kGroupsPerWave + grp;
    const bool live = grp < kGroupsPerWave && b < B;
    const int bb = live ? b : 0;
    // ---- vehicle state (what cal_vehicle_info provides, ref :90-113)
    double x = state[5 * bb], y = state[5 * bb + 1], fi = state[5 * bb + 2];
    const double Vy = state[5 * bb + 3], fi_dot = state[5 * bb + 4], Vx = vx[bb];
    // ---- continuous error model (ref :115-148)
    M4 A;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) A.a[i][j] = 0.0;
    A.a[0][1] = 1.0;
    A.a[1][1] = (prm.Cf + prm.Cr) / (prm.m * Vx);
    A.a[1][2] = -(prm.Cf + prm.Cr) / prm.m;
    A.a[1][3] = (prm.a * prm.Cf - prm.b * prm.Cr) / (prm.m * Vx);
    A.a[2][3] = 1.0;
    A.a[3][1] = (prm.a * prm.Cf - prm.b * prm.Cr) / (prm.Iz * Vx);
    A.a[3][2] = -(prm.a * prm.Cf - prm.b * prm.Cr) / prm.Iz;
    A.a[3][3] = (prm.a * prm.a * prm.Cf + prm.b * prm.b * prm.Cr) / (prm.Iz * Vx);
    const V4 Bc{{0.0, -prm.Cf / prm.m, 0.0, -prm.a * prm.Cf / prm.Iz}};
    const V4 Cc{{0.0, (prm.a * prm.Cf + prm.b * prm.Cr) / (prm.m * Vx) - Vx, 0.0,
                 (prm.a * prm.a * prm.Cf + prm.b * prm.b * prm.Cr) / (prm.Iz * Vx)}};
    // ---- prediction and tracking error (ref :170-251, ts = 0.1)
    {
        const double c = cos(fi), s = sin(fi);
        const double xn = x + Vx * kTs * c - Vy * kTs * s;
        const double yn = y + Vy * kTs * c + Vx * kTs * s;
        x = xn;
        y = yn;
        fi = fi + fi_dot * kTs;
    }
    const double* path = target_path + (size_t)bb * max_path * 4;
    const int np_ = n_path[bb];
    int idx = min_index_in[bb];
    bool bad_index = live && (np_ < 1 || idx < 0 || idx >= np_);   // the reference raises IndexError at :224
    if (bad_index) idx = 0;
    {
        double min_d = 10000.0;                                     // squared metres (ref :201): farther than 100 m
        const int first = idx, last = min(first + kWindow, np_);    // keeps the previous match
        for (int i = first; i < last; ++i) {
            const double dx = path[4 * i] - x, dy = path[4 * i + 1] - y;
            const double d = dx * dx + dy * dy;
            if (d < min_d) {
                min_d = d;
                idx = i;
            }
        }
    }
    const double px = path[4 * idx], py = path[4 * idx + 1], pth = path[4 * idx + 2], pk = path[4 * idx + 3];
    const double ct = cos(pth), st = sin(pth);
    const double dvx = x - px, dvy = y - py;
    const double e_d = -st * dvx + ct * dvy;
    const double e_s = ct * dvx + st * dvy;
    const double theta_r = pth + pk * e_s;
    const double cd = cos(fi - theta_r), sd = sin(fi - theta_r);
    const double e_d_dot = Vy * cd + Vx * sd;
    const double e_fi = sd;
    const double S_dot = (Vx * cd - Vy * sd) / (1.0 - pk * e_d);
    const double e_fi_dot = fi_dot - pk * S_dot;
    const V4 e_rr{{e_d, e_d_dot, e_fi, e_fi_dot}};
    // ---- bilinear discretisation (ref :159-165)
    M4 lhs, rhs, inv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double e = (i == j) ? 1.0 : 0.0;
            lhs.a[i][j] = e - (kTs * A.a[i][j]) / 2.0;
            rhs.a[i][j] = e + (kTs * A.a[i][j]) / 2.0;
        }
    const bool inv_ok = inverse4(lhs, &inv);
    M4 Ab;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            Ab.a[i][j] = ((inv.a[i][0] * rhs.a[0][j] + inv.a[i][1] * rhs.a[1][j]) + inv.a[i][2] * rhs.a[2][j]) + inv.a[i][3] * rhs.a[3][j];
    V4 Bb = matvec(inv, Bc), Cb = matvec(inv, Cc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Bb.v[i] = Bb.v[i] * kTs;
        Cb.v[i] = Cb.v[i] * kTs * pk * Vx;
    }
    // ---- condensed problem (ref :262-298): g_t = A_bar^t B_bar, free response w_i = A_bar^i e_rr + Cc_i
    V4 g[kN];
    g[0] = Bb;
#pragma unroll
    for (int t = 1; t < kN; ++t) g[t] = matvec(Ab, g[t - 1]);
    V4 w[kN + 1];
    {
        V4 me = e_rr, cc{{0.0, 0.0, 0.0, 0.0}};
        w[0] = me;
#pragma unroll
        for (int i = 1; i <= kN; ++i) {
            me = matvec(Ab, me);
            cc = matvec(Ab, cc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cc.v[q] += Cb.v[q];
                w[i].v[q] = cc.v[q] + me.v[q];
            }
        }
    }
    const int jr = r / kP;
    double h[kNu], f_r = 0.0;
#pragma unroll
    for (int c = 0; c < kNu; ++c) {
        const int jc = c / kP;
        double acc = 0.0;
#pragma unroll
        for (int i = 1; i <= kN; ++i) {
            // block row i of C holds A_bar^(i-1-j) B_bar in the P columns of step j < i (ref :268-273)
            if (i - 1 - jr < 0 || i - 1 - jc < 0) continue;
            const double* wt = (i == kN) ? prm.f : prm.q;
            const V4& ga = g[(i - 1 - jr) < 0 ? 0 : (i - 1 - jr)];
            const V4& gc = g[i - 1 - jc];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += ga.v[q] * wt[q] * gc.v[q];
        }
        if (c == r) acc += prm.r;
        h[c] = 2.0 * acc;
    }
#pragma unroll
    for (int i = 1; i <= kN; ++i) {
        if (i - 1 - jr < 0) continue;
        const double* wt = (i == kN) ? prm.f : prm.q;
        const V4& ga = g[(i - 1 - jr) < 0 ? 0 : (i - 1 - jr)];
#pragma unroll
        for (int q = 0; q < 4; ++q) f_r += ga.v[q] * wt[q] * w[i].v[q];
    }
    f_r = 2.0 * f_r;
    // ---- box QP (ref :300-311) and outputs
    const bool solvable = live && !bad_index && inv_ok;
    double u = 0.0;
    int it = 0;
    const int rc = box_qp_full12(h, f_r, r, gb, solvable, &u, &it);
    if (live) {
        if (H_out)
#pragma unroll
            for (int c = 0; c < kNu; ++c) H_out[((size_t)b * kNu + r) * kNu + c] = h[c];
        if (f_out) f_out[(size_t)b * kNu + r] = f_r;
        if (u_out) u_out[(size_t)b * kNu + r] = (solvable && rc == 0) ? u : 0.0;
        if (r == 0) {
            steer[b] = (solvable && rc == 0) ? u : 0.0;           // ref :311: res['x'][0]
            if (e_rr_out)
#pragma unroll
                for (int q = 0; q < 4; ++q) e_rr_out[4 * b + q] = e_rr.v[q];
            if (k_r_out) k_r_out[b] = pk;
            min_index_out[b] = idx;
            if (pre_pro) {
                pre_pro[4 * b] = x;
                pre_pro[4 * b + 1] = y;
                pre_pro[4 * b + 2] = px + e_s * ct;
                pre_pro[4 * b + 3] = py + e_s * st;
            }
            if (iters_out) iters_out[b] = it;
            status[b] = bad_index ? kStSOutOfRange : ((!inv_ok || rc != 0) ? kStQpFailed : 0);
        }
    }
}

}  // namespace mpc

// ---------------------------------------------------------------------------------------------
// Lateral LQR controller, ref controller/controller.py class Lateral_LQR_controller (:374-611): `_control` from
// explicit inputs.  One vehicle per lane: the 4 x 4 algebra lives in registers and the Riccati iteration
// (:470-481: until max|dP| < 0.1, at most 5000 sweeps - a slowly creeping vehicle needs all of them) is a private
// loop; lanes of a wavefront simply run as long as their slowest vehicle.  Products are associated as NumPy
// evaluates the reference's expression (left to right).
// ---------------------------------------------------------------------------------------------
namespace lqr {

using mpc::M4;
using mpc::V4;

__device__ __forceinline__ M4 matmul(const M4& A, const M4& B) {
    M4 C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            C.a[i][j] = ((A.a[i][0] * B.a[0][j] + A.a[i][1] * B.a[1][j]) + A.a[i][2] * B.a[2][j]) + A.a[i][3] * B.a[3][j];
    return C;
}

__global__ void lqr_lateral_kernel(int B, int max_path, mpc::Params prm, const double* __restrict__ target_path,
                                   const int* __restrict__ n_path, const double* __restrict__ state,
                                   const double* __restrict__ vx, const int* __restrict__ min_index_in,
                                   double* __restrict__ steer, double* __restrict__ K_out, double* __restrict__ e_rr_out,
                                   double* __restrict__ k_r_out, int* __restrict__ min_index_out,
                                   double* __restrict__ pre_pro, int* __restrict__ sweeps_out, int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x = state[5 * b], y = state[5 * b + 1], fi = state[5 * b + 2];
    const double Vy = state[5 * b + 3], fi_dot = state[5 * b + 4], Vx = vx[b];
    // ---- continuous model (ref :424-455): Vx + 0.0001 guards the divisions
    const double Vg = Vx + 0.0001;
    M4 A;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) A.a[i][j] = 0.0;
    A.a[0][1] = 1.0;
    A.a[1][1] = (prm.Cf + prm.Cr) / (prm.m * Vg);
    A.a[1][2] = -(prm.Cf + prm.Cr) / prm.m;
    A.a[1][3] = (prm.a * prm.Cf - prm.b * prm.Cr) / (prm.m * Vg);
    A.a[2][3] = 1.0;
    A.a[3][1] = (prm.a * prm.Cf - prm.b * prm.Cr) / (prm.Iz * Vg);
    A.a[3][2] = -(prm.a * prm.Cf - prm.b * prm.Cr) / prm.Iz;
    A.a[3][3] = (prm.a * prm.a * prm.Cf + prm.b * prm.b * prm.Cr) / (prm.Iz * Vg);
    const V4 Bc{{0.0, -prm.Cf / prm.m, 0.0, -prm.a * prm.Cf / prm.Iz}};
    // ---- discretisation and Riccati iteration (ref :466-481)
    M4 lhs, rhs, inv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double e = (i == j) ? 1.0 : 0.0;
            lhs.a[i][j] = e - (mpc::kTs * A.a[i][j]) / 2.0;
            rhs.a[i][j] = e + (mpc::kTs * A.a[i][j]) / 2.0;
        }
    const bool inv_ok = mpc::inverse4(lhs, &inv);
    const M4 Ad = matmul(inv, rhs);
    V4 Bd = mpc::matvec(inv, Bc);
#pragma unroll
    for (int i = 0; i < 4; ++i) Bd.v[i] = Bd.v[i] * mpc::kTs;
    M4 AT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) AT.a[i][j] = Ad.a[j][i];
    M4 P, Ppre;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P.a[i][j] = Ppre.a[i][j] = (i == j) ? prm.q[i] : 0.0;
    int sweeps = 0;
    for (int it = 0; it < 5000; ++it) {
        const M4 ATP = matmul(AT, P);
        const M4 ATPA = matmul(ATP, Ad);
        const V4 ATPB = mpc::matvec(ATP, Bd);
        V4 BTP;                                             // B' P (row vector)
#pragma unroll
        for (int j = 0; j < 4; ++j) BTP.v[j] = ((Bd.v[0] * P.a[0][j] + Bd.v[1] * P.a[1][j]) + Bd.v[2] * P.a[2][j]) + Bd.v[3] * P.a[3][j];
        V4 BTPA;
#pragma unroll
        for (int j = 0; j < 4; ++j) BTPA.v[j] = ((BTP.v[0] * Ad.a[0][j] + BTP.v[1] * Ad.a[1][j]) + BTP.v[2] * Ad.a[2][j]) + BTP.v[3] * Ad.a[3][j];
        const double BTPB = ((BTP.v[0] * Bd.v[0] + BTP.v[1] * Bd.v[1]) + BTP.v[2] * Bd.v[2]) + BTP.v[3] * Bd.v[3];
        const double g = 1.0 / (prm.r + BTPB);
        double delta = 0.0;
        M4 Pn;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Pn.a[i][j] = (ATPA.a[i][j] - (ATPB.v[i] * g) * BTPA.v[j]) + ((i == j) ? prm.q[i] : 0.0);
                delta = fmax(delta, fabs(Pn.a[i][j] - Ppre.a[i][j]));
            }
        P = Pn;
        sweeps = it + 1;
        if (delta < 0.1) break;
        Ppre = Pn;
    }
    V4 K;
    {
        V4 BTP;
#pragma unroll
        for (int j = 0; j < 4; ++j) BTP.v[j] = ((Bd.v[0] * P.a[0][j] + Bd.v[1] * P.a[1][j]) + Bd.v[2] * P.a[2][j]) + Bd.v[3] * P.a[3][j];
        const double BTPB = ((BTP.v[0] * Bd.v[0] + BTP.v[1] * Bd.v[1]) + BTP.v[2] * Bd.v[2]) + BTP.v[3] * Bd.v[3];
        const double g = 1.0 / (BTPB + prm.r);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            K.v[j] = g * (((BTP.v[0] * Ad.a[0][j] + BTP.v[1] * Ad.a[1][j]) + BTP.v[2] * Ad.a[2][j]) + BTP.v[3] * Ad.a[3][j]);
    }
    // ---- prediction and tracking error (ref :488-567, ts = 0.1): nearest point over the whole path
    {
        const double c = cos(fi), s = sin(fi);
        const double xn = x + Vx * mpc::kTs * c - Vy * mpc::kTs * s;
        const double yn = y + Vy * mpc::kTs * c + Vx * mpc::kTs * s;
        x = xn;
        y = yn;
        fi = fi + fi_dot * mpc::kTs;
    }
    const double* path = target_path + (size_t)b * max_path * 4;
    const int np_ = n_path[b];
    int idx = min_index_in[b];
    {
        double min_d = 10000.0;
        for (int i = 0; i < np_; ++i) {
            const double dx = path[4 * i] - x, dy = path[4 * i + 1] - y;
            const double d = dx * dx + dy * dy;
            if (d < min_d) {
                min_d = d;
                idx = i;
            }
        }
    }
    const bool bad_index = np_ < 1 || idx < 0 || idx >= np_;          // IndexError in the reference
    if (bad_index) idx = 0;
    const double px = path[4 * idx], py = path[4 * idx + 1], pth = path[4 * idx + 2], pk = path[4 * idx + 3];
    const double ct = cos(pth), st = sin(pth);
    const double dvx = x - px, dvy = y - py;
    const double e_d = -st * dvx + ct * dvy;
    const double e_s = ct * dvx + st * dvy;
    const double theta_r = pth + pk * e_s;
    const double cd = cos(fi - theta_r), sd = sin(fi - theta_r);
    const double e_d_dot = Vy * cd + Vx * sd;
    const double e_fi = sd;
    const double S_dot = (Vx * cd - Vy * sd) / (1.0 - pk * e_d);
    const double e_fi_dot = fi_dot - pk * S_dot;
    // ---- feed-forward (ref :569-583) and control (ref :606)
    const double K3 = K.v[2];
    double delta_f = pk * (prm.a + prm.b - prm.b * K3 -
                           (prm.b / prm.Cf + prm.a * K3 / prm.Cr - prm.a / prm.Cr) * (prm.m * Vx * Vx) / (prm.a + prm.b));
    delta_f = delta_f * 3.141592653589793 / 180.0;
    const double u = -(((K.v[0] * e_d + K.v[1] * e_d_dot) + K.v[2] * e_fi) + K.v[3] * e_fi_dot) + delta_f;
    const bool ok = inv_ok && !bad_index;
    steer[b] = ok ? u : 0.0;
    if (K_out)
#pragma unroll
        for (int j = 0; j < 4; ++j) K_out[4 * b + j] = K.v[j];
    if (e_rr_out) {
        e_rr_out[4 * b] = e_d;
        e_rr_out[4 * b + 1] = e_d_dot;
        e_rr_out[4 * b + 2] = e_fi;
        e_rr_out[4 * b + 3] = e_fi_dot;
    }
    if (k_r_out) k_r_out[b] = pk;
    min_index_out[b] = idx;
    if (pre_pro) {
        pre_pro[4 * b] = x;
        pre_pro[4 * b + 1] = y;
        pre_pro[4 * b + 2] = px + e_s * ct;
        pre_pro[4 * b + 3] = py + e_s * st;
    }
    if (sweeps_out) sweeps_out[b] = sweeps;
    status[b] = bad_index ? kStSOutOfRange : (inv_ok ? 0 : kStQpFailed);
}

}  // namespace lqr
}  // namespace emp
