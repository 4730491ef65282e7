// sanitize_main.cpp - the host-side logic of the library (the scalar cores the kernels are built from: emp_core.h,
// emp_frenet_core.h, emp_qp_core.h, emp_st_core.h, emp_st_backend_core.h, reached through host_check.cpp's entry
// points) driven with deterministic random and hostile inputs under -fsanitize=address,undefined.  Test tool only:
// tests/test_host_logic.py builds and runs it; any sanitizer report makes it exit non-zero.
//
// Inputs: sizes at and beside every compile-time capacity (0, 1, 2, 3, 31..34, 63..65, 255, 256, 257), empty and
// crossed bounds (infeasible QPs), NaN / Inf coordinates, repeated points, obstacle counts of zero and of the slot
// capacity, S-T slots that are all NaN.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
double hc_segment_cost(double, double, double, double, double, double, const double*, const double*, int, double, double,
                       double, double, double);
int hc_path_qp(int, const double*, const double*, double, double, double, const double*, double*, double*, double*, int*);
int hc_box_qp(int, const double*, int, double, double, double, double, double*, int*);
void hc_heading_kappa(const double*, int, double*, double*);
void hc_s_map(const double*, int, double, double, double*);
int hc_match(const double*, int, double, double, int, int, int);
double hc_st_edge_cost(const double*, const double*, int, const double*, const double*, const double*, const double*, double*);
void hc_st_graph(int, const double*, const double*, const double*, const double*, double*, double*, double*, double*);
void hc_st_grid(double*, double*);
int hc_st_terminal(const double*, int*, int*);
int hc_stb_convex_space(const double*, const double*, const double*, const double*, int, const double*, const double*,
                        const double*, const double*, int, double, double*, double*, double*, double*);
int hc_stb_speed_qp(const double*, const double*, double, double, const double*, const double*, const double*, const double*,
                    const double*, double*, double*, double*, double*, int*);
int hc_stb_increase_points(const double*, const double*, const double*, const double*, double*, double*, double*, double*);
double hc_stb_np_interp(const double*, const double*, int, double);
}

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double u01() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (double)(g_state >> 11) / 9007199254740992.0;
}
static double uni(double a, double b) { return a + (b - a) * u01(); }
static double hostile(double v) {                                   // now and then a value no real scene holds
    const double r = u01();
    if (r < 0.01) return NAN;
    if (r < 0.02) return INFINITY;
    if (r < 0.03) return -INFINITY;
    if (r < 0.04) return 1e300;
    return v;
}

int main() {
    long calls = 0;
    double sink = 0.0;
    const int sizes[] = {0, 1, 2, 3, 4, 5, 21, 31, 32, 33, 34, 51, 63, 64, 65, 128, 255, 256, 257};
    const double prm8[8] = {2.0, 1000.0, 3000.0, 150.0, 250.0, 40.0, 3.0, 3.0};   // ds, w_l, w_ddl, w_dddl, w_centre, w_end, d, w
    for (int rep = 0; rep < 6; ++rep) {
        for (int n : sizes) {
            // ---- path QP: feasible corridors, crossed bounds, hostile values
            std::vector<double> lo(n + 1), hi(n + 1), ql(n + 1), qdl(n + 1), qddl(n + 1);
            for (int i = 0; i < n; ++i) {
                lo[i] = uni(-10.0, -1.0);
                hi[i] = uni(1.0, 10.0);
                if (rep == 1 && i % 7 == 3) std::swap(lo[i], hi[i]);                 // crossed: infeasible
                if (rep == 2) { lo[i] = hostile(lo[i]); hi[i] = hostile(hi[i]); }
            }
            int iters = 0;
            if (n >= 1) sink += hc_path_qp(n, lo.data(), hi.data(), uni(-0.5, 0.5), uni(-0.05, 0.05), uni(-0.01, 0.01), prm8,
                                           ql.data(), qdl.data(), qddl.data(), &iters);
            ++calls;
            // ---- box QP (reference-line / trajectory smoothing) on n points, stride 2
            std::vector<double> xy(2 * (n + 1)), out(n + 1), th(n + 1), kp(n + 1);
            for (int i = 0; i < n; ++i) {
                xy[2 * i] = 2.0 * i + uni(-0.3, 0.3);
                xy[2 * i + 1] = 8.0 * std::sin(i / 12.0) + uni(-0.3, 0.3);
                if (rep == 3 && i % 5 == 0 && i) { xy[2 * i] = xy[2 * i - 2]; xy[2 * i + 1] = xy[2 * i - 1]; }   // repeated point
                if (rep == 4) xy[2 * i] = hostile(xy[2 * i]);
            }
            sink += hc_box_qp(n, xy.data(), 2, 0.4, 0.3, 0.3, 0.2, out.data(), &iters);
            if (n >= 2) hc_heading_kappa(xy.data(), n, th.data(), kp.data());
            // ---- reference line helpers on an n-point line
            std::vector<double> line(4 * (n + 1)), smap(n + 1);
            for (int i = 0; i < n; ++i) {
                line[4 * i] = xy[2 * i];
                line[4 * i + 1] = xy[2 * i + 1];
                line[4 * i + 2] = uni(-3.1, 3.1);
                line[4 * i + 3] = uni(-0.01, 0.01);
            }
            if (n >= 1) {
                hc_s_map(line.data(), n, hostile(uni(0, 2.0 * n)), uni(-5, 5), smap.data());
                sink += hc_match(line.data(), n, uni(0, 2.0 * n), hostile(uni(-5, 5)), (int)(u01() * n), u01() < 0.5 ? 1 : -1, 5);
                sink += hc_match(line.data(), n, uni(0, 2.0 * n), uni(-5, 5), 0, 1, 50);
            }
            calls += 4;
        }
        // ---- lattice edge cost with 0..64 obstacles, hostile positions
        for (int k : {0, 1, 3, 8, 16, 64}) {
            std::vector<double> os(k + 1), ol(k + 1);
            for (int m = 0; m < k; ++m) { os[m] = hostile(uni(0, 100)); ol[m] = hostile(uni(-8, 8)); }
            sink += hc_segment_cost(uni(-6, 6), uni(-0.1, 0.1), uni(-0.02, 0.02), uni(-6, 6), uni(0, 100), uni(0.5, 15.0), os.data(),
                                    ol.data(), k, 1e12, 300, 1000, 5000, 20) > 0;
            ++calls;
        }
        // ---- S-T speed DP pieces and the speed planning back end: 16 slots, any number of them NaN
        double s[16], l[16], sd[16], ld[16], si[16], so[16], ti[16], to[16];
        const int present = rep == 0 ? 0 : (rep == 1 ? 16 : (int)(u01() * 17));
        for (int j = 0; j < 16; ++j) {
            const bool on = j < present;
            s[j] = on ? uni(5, 50) : NAN;
            l[j] = on ? uni(-6, 6) : NAN;
            sd[j] = on ? uni(0, 6) : NAN;
            ld[j] = on ? (u01() < 0.1 ? uni(-0.29, 0.29) : (l[j] > 0 ? -1 : 1) * uni(0.5, 2)) : NAN;
        }
        hc_st_graph(16, s, l, sd, ld, si, so, ti, to);
        const double w4[4] = {50.0, 4000.0, 100.0, 1e7};
        double rows[40], cols[16];
        hc_st_grid(rows, cols);
        for (int e = 0; e < 200; ++e) {
            const double edge5[5] = {uni(0, 60), uni(0, 8), uni(0, 30), uni(0, 60), uni(0, 8)};
            double obs = 0;
            sink += hc_st_edge_cost(w4, edge5, 16, si, so, ti, to, &obs) > 0;
            ++calls;
        }
        std::vector<double> cost(40 * 16);
        for (double& c : cost) c = rep == 5 ? INFINITY : hostile(uni(0, 1e6));
        int r = 0, c = 0;
        sink += hc_st_terminal(cost.data(), &r, &c);
        double dps[16], dpt[16], idx2s[401], kap[401], slb[16], sub[16], vlb[16], vub[16];
        for (int j = 0; j < 16; ++j) { dpt[j] = 0.5 * (j + 1); dps[j] = j < 3 + (int)(u01() * 13) ? 4.0 * (j + 1) : NAN; }
        const int plen = rep % 2 ? 401 : 120;
        for (int j = 0; j < 401; ++j) { idx2s[j] = j < plen ? 0.5 * j : NAN; kap[j] = j < plen ? uni(-0.05, 0.05) : NAN; }
        const int crc = hc_stb_convex_space(dps, dpt, idx2s, kap, plen, si, so, ti, to, 16, 0.3 * 9.8, slb, sub, vlb, vub);
        double qs[17], qv[17], qa[17], qt[17];
        const double wq[4] = {100.0, 10.0, 1.0, 50.0};
        int it = 0;
        if (crc == 0) sink += hc_stb_speed_qp(dps, dpt, uni(0, 15), uni(-1, 1), slb, sub, vlb, vub, wq, qs, qv, qa, qt, &it);
        for (int j = 0; j < 17; ++j) { qs[j] = 3.0 * j; qv[j] = 6.0; qa[j] = 0.0; qt[j] = j < 5 + (int)(u01() * 12) ? 0.5 * j : NAN; }
        double ds[401], dv[401], da[401], dt[401];
        sink += hc_stb_increase_points(qs, qv, qa, qt, ds, dv, da, dt);
        for (int q = 0; q < 50; ++q) sink += hc_stb_np_interp(idx2s, kap, plen, hostile(uni(-5, 220))) > 0;
        calls += 5;
    }
    std::printf("sanitize_main: %ld calls, checksum %d\n", calls, (int)std::fmod(std::fabs(sink), 1000.0));
    return 0;
}
