// host_check.cpp - runs the scalar building blocks of csrc/*.h on the CPU for the `-m "not gpu"` suite.
// TEST TOOL ONLY: compiled with g++ into tests/host_check/_build/, loaded by tests/test_host_logic.py,
// never by the package.  It lets kernel *logic* (banded QP, Frenet helpers, edge arithmetic) be checked
// against the oracle without a GPU; GPU parity proper is tests/test_gpu_*.py through the C-ABI.
#include "../../emplanner_carla_amd/csrc/emp_core.h"
#include "../../emplanner_carla_amd/csrc/emp_frenet_core.h"
#include "../../emplanner_carla_amd/csrc/emp_qp_core.h"
#include "../../emplanner_carla_amd/csrc/emp_st_core.h"
#include "../../emplanner_carla_amd/csrc/emp_st_backend_core.h"

using namespace emp;

extern "C" {

double hc_segment_cost(double l0, double dl0, double ddl0, double l1, double s0, double sample_s, const double* obs_s,
                       const double* obs_l, int n_obs, double w_coll, double w0, double w1, double w2, double w_ref) {
    const Quintic q = quintic_shifted(l0, dl0, ddl0, l1, sample_s);
    return segment_cost(q, s0, sample_s, obs_s, obs_l, n_obs, w_coll, w0, w1, w2, w_ref);
}

double hc_neighbour_cost(double l0, double l1, double s0, double sample_s, const double* obs_s, const double* obs_l, int n_obs,
                         double w_coll, double w0, double w1, double w2, double w_ref) {
    return neighbour_cost(l0, l1, s0, sample_s, obs_s, obs_l, n_obs, w_coll, w0, w1, w2, w_ref);
}

int hc_path_qp(int n, const double* l_min, const double* l_max, double l0, double dl0, double ddl0, const double* prm8,
               double* out_l, double* out_dl, double* out_ddl, int* iters) {
    static double mem[path_qp_words(256)];
    if (n > 256) return 2;
    PathQpParams p{prm8[0], prm8[1], prm8[2], prm8[3], prm8[4], prm8[5], prm8[6], prm8[7]};
    return path_qp_solve_scalar(mem, l_min, l_max, n, l0, dl0, ddl0, p, out_l, out_dl, out_ddl, iters);
}

int hc_path_qp_gi(int n, const double* l_min, const double* l_max, double l0, double dl0, double ddl0, const double* prm8,
                  double* out_l, double* out_dl, double* out_ddl, int* iters) {
    static double mem[path_qp_words(256)];
    static double work[gi_words(256)];
    if (n > 256) return 2;
    PathQpParams p{prm8[0], prm8[1], prm8[2], prm8[3], prm8[4], prm8[5], prm8[6], prm8[7]};
    return path_qp_solve_scalar(mem, l_min, l_max, n, l0, dl0, ddl0, p, out_l, out_dl, out_ddl, iters, work);
}

int hc_box_qp(int m, const double* ref, int stride, double w_smooth, double w_length, double w_ref, double thr,
              double* out, int* iters) {
    static double mem[BoxRangeQp::words(256, 256)];
    *iters = 0;
    if (m > 256 || m < 2) return 2;
    BoxRangeQp Q;
    Q.bind(mem, m, m);
    SmoothQpParams p{w_smooth, w_length, w_ref, thr};
    int rc = box_qp_setup(Q, ref, stride, m, p);
    if (rc) return rc;
    rc = Q.solve_scalar();
    *iters = Q.iters;
    for (int i = 0; i < m; ++i) out[i] = Q.u[i];
    return rc;
}

void hc_heading_kappa(const double* xy, int m, double* theta, double* kappa) { heading_kappa(xy, 2, m, theta, 1, kappa, 1); }

void hc_s_map(const double* line, int n_ref, double ox, double oy, double* s_map) { s_map_build(line, n_ref, ox, oy, s_map); }

void hc_dot2(int n, const double* a, const double* b, double* out) {     // a, b: [n][2]
    for (int i = 0; i < n; ++i) out[i] = dot2(a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1]);
}

int hc_match(const double* line, int n_ref, double x, double y, int first, int step, int limit) {
    return match_scan(line, n_ref, x, y, first, step, limit);
}

// ---- S-T speed DP scalar pieces (emp_st_core.h) ---------------------------------------------------
double hc_st_edge_cost(const double* w4, const double* edge5, int n_obs, const double* s_in, const double* s_out,
                       const double* t_in, const double* t_out, double* obs) {
    const st::Weights w{w4[0], w4[1], w4[2], st::make_pow_base(w4[3])};
    double ux[st::kMaxObs], uy[st::kMaxObs], len[st::kMaxObs];
    for (int j = 0; j < n_obs; ++j) st::obs_frame(s_in[j], t_in[j], s_out[j], t_out[j], &ux[j], &uy[j], &len[j]);
    const st::ObsSet set{n_obs, s_in, s_out, t_in, t_out, ux, uy, len};
    return st::edge_cost(w, edge5[0], edge5[1], edge5[2], edge5[3], edge5[4], set, obs);
}

// speed DP kernel, round 3: the branch-free pair cost against the reference-order one, and whether the sample lies inside
// the segment's reach interval at its time (it must whenever the pair costs anything)
void hc_st_reach(int n, double w_obs, const double* seg4, const double* pt2, double* cost, double* cost_flat, int* inside,
                 double* lo_hi) {
    const st::PowBase w = st::make_pow_base(w_obs);
    for (int i = 0; i < n; ++i) {
        const double s_in = seg4[4 * i], t_in = seg4[4 * i + 1], s_out = seg4[4 * i + 2], t_out = seg4[4 * i + 3];
        const double s = pt2[2 * i], t = pt2[2 * i + 1];
        cost[i] = st::point_cost(w, s, t, s_in, t_in, s_out, t_out);
        cost_flat[i] = st::point_cost_flat(w, s, t, s_in, t_in, s_out, t_out);
        double ux, uy, len, lo, hi;
        st::obs_frame(s_in, t_in, s_out, t_out, &ux, &uy, &len);
        st::reach_interval(t, s_in, t_in, ux, uy, len, &lo, &hi);
        inside[i] = s > lo && s < hi;
        lo_hi[2 * i] = lo;
        lo_hi[2 * i + 1] = hi;
    }
}

void hc_st_graph(int n, const double* s, const double* l, const double* sd, const double* ld, double* s_in, double* s_out,
                 double* t_in, double* t_out) {
    st::st_graph(n, s, l, sd, ld, s_in, s_out, t_in, t_out);
}

void hc_st_grid(double* s_rows, double* t_cols) {
    for (int r = 0; r < st::kRows; ++r) s_rows[r] = st::s_of_row(r);
    for (int c = 0; c < st::kCols; ++c) t_cols[c] = st::t_of_col(c);
}

int hc_st_terminal(const double* cost, int* row, int* col) {
    return st::terminal_node([&](int r, int c) { return cost[r * st::kCols + c]; }, row, col) ? 1 : 0;
}


// ---- S-T speed planning back end (emp_st_backend_core.h) ------------------------------------
int hc_stb_convex_space(const double* dp_s, const double* dp_t, const double* idx2s, const double* kappa, int path_len,
                        const double* s_in, const double* s_out, const double* t_in, const double* t_out, int n_slots,
                        double max_lat, double* s_lb, double* s_ub, double* sd_lb, double* sd_ub) {
    return stb::convex_space(dp_s, dp_t, idx2s, kappa, path_len, s_in, s_out, t_in, t_out, n_slots, max_lat, s_lb, s_ub,
                             sd_lb, sd_ub);
}

int hc_stb_speed_qp(const double* dp_s, const double* dp_t, double v0, double a0, const double* s_lb, const double* s_ub,
                    const double* sd_lb, const double* sd_ub, const double* w4, double* qs, double* qv, double* qa,
                    double* qt, int* iters) {
    static double mem[stb::speed_qp_words(stb::kQp)];
    const stb::SpeedQpParams prm{w4[0], w4[1], w4[2], w4[3]};
    return stb::speed_qp_solve_scalar(mem, dp_s, dp_t, v0, a0, s_lb, s_ub, sd_lb, sd_ub, prm, qs, qv, qa, qt, iters);
}

// increase_points with the kernel's structure: match per sample, running maximum as the sticky interval
int hc_stb_increase_points(const double* qs, const double* qv, const double* qa, const double* qt, double* s, double* v,
                           double* a, double* t) {
    const int t_end = stb::dense_t_end(qt);
    if (t_end >= stb::kQp || t_end < 0) return t_end < 0 ? stb::kStbNoProfile : stb::kStbIndex;
    const double dt = qt[t_end] / (double)(stb::kDense - 1);
    int tmp = 0;
    for (int i = 0; i < stb::kDense; ++i) {
        const double cur = (double)(i - 1) * dt;
        const int m = stb::dense_match(qt, t_end, cur);
        if (m > tmp) tmp = m;
        stb::dense_sample(qs, qv, qa, qt, tmp, cur, &s[i], &v[i], &a[i]);
        t[i] = cur;
    }
    return 0;
}

double hc_stb_np_interp(const double* xp, const double* fp, int n, double x) {
    return stb::np_interp_at(xp, fp, n, (x != x) ? 0 : stb::np_interp_index(xp, n, x), x);
}

}  // extern "C"
