// emp_frenet_core.h - Cartesian <-> Frenet helpers of the reference's planning_utils.py, scalar form.
// Tolerance contract (not bit-exact): device sin/cos/atan2 differ from libm in the last ulp; tests
// compare at 1e-6 relative.  "ref:" cites the reference (paths relative to the reference tree).
#pragma once

#include "emp_core.h"

namespace emp {

struct Node {
    double x, y, theta, kappa;
};
EMP_HD Node node_at(const double* line, int i) { return Node{line[4 * i], line[4 * i + 1], line[4 * i + 2], line[4 * i + 3]}; }

// ref: match_projection_points scan, planning_utils.py:383-402 - strict minimum from index `first`,
// stepping `step` (+1/-1), stop after `limit` consecutive non-improvements.
EMP_HD int match_scan(const double* line, int n_ref, double x, double y, int first, int step, int limit) {
    int match = 0, worse = 0;
    double best = __builtin_inf();
    for (int i = first; i >= 0 && i < n_ref; i += step) {
        const double dx = line[4 * i] - x, dy = line[4 * i + 1] - y;
        const double d = sqrt(dx * dx + dy * dy);
        if (d < best) {
            best = d;
            match = i;
            worse = 0;
        } else if (++worse >= limit) {
            break;
        }
    }
    return match;
}

// numpy.dot of two 2-vectors, as the reference evaluates every projection (planning_utils.py:107, :173, :417, :443, :507,
// :546-578, :799-800).  On an x86 host numpy hands it to OpenBLAS' ddot kernel, which accumulates with fused
// multiply-adds: the result is fma(a1, b1, a0 * b0) - one rounding of the second product fewer than a0*b0 + a1*b1,
// different in the last bit a quarter of the time (checked against numpy 2.2 / OpenBLAS 0.3.29 on 200 000 random
// pairs: identical every time).  That bit decides ties the reference cannot see: a planning start that projects exactly
// onto a node of the reference line (the synthetic scenes put it there) lands on one side or the other of
// `s_map[idx + 1] < s` (path_planning.py:63) by the rounding of this sum.  Evaluating it the way the reference's host
// does took the full-cycle mismatches of a 2048-scene sweep (tools/parity_sweep.py) from two scenes, first trajectory point
// 0.45 mm off, to one - whose tie is decided by the last bit of cos / sin instead.
EMP_HD double dot2(double a0, double a1, double b0, double b1) { return __builtin_fma(a1, b1, a0 * b0); }

// ref: planning_utils.py:414-424 - projection on the tangent line of a matched node
EMP_HD Node project_on(const Node& m, double x, double y) {
    const double c = cos(m.theta), s = sin(m.theta);
    const double ds = dot2(x - m.x, y - m.y, c, s);
    return Node{m.x + ds * c, m.y + ds * s, m.theta + m.kappa * ds, m.kappa};
}

// the same with cos / sin of the node's heading supplied (the projection kernel evaluates them once per node and scene)
EMP_HD Node project_on_cs(const Node& m, double c, double s, double x, double y) {
    const double ds = dot2(x - m.x, y - m.y, c, s);
    return Node{m.x + ds * c, m.y + ds * s, m.theta + m.kappa * ds, m.kappa};
}
EMP_HD double projection_s_cs(const Node& m, double c, double s, double s_at_m, double x, double y) {
    return s_at_m + dot2(x - m.x, y - m.y, c, s);
}

// ref: cal_projection_s_fun, planning_utils.py:439-443
EMP_HD double projection_s(const Node& m, double s_at_m, double x, double y) {
    return s_at_m + dot2(x - m.x, y - m.y, cos(m.theta), sin(m.theta));
}

// ref: cal_s_l_fun tail, planning_utils.py:499-507
EMP_HD double lateral_offset(const Node& proj, double x, double y) {
    return dot2(x - proj.x, y - proj.y, -sin(proj.theta), cos(proj.theta));
}

// ref: cal_s_map_fun, planning_utils.py:448-472.  s_map has n_ref entries.
EMP_HD void s_map_build(const double* line, int n_ref, double ox, double oy, double* s_map) {
    const int m0 = match_scan(line, n_ref, ox, oy, 0, 1, 50);
    double acc = 0.0;
    s_map[0] = 0.0;
    for (int i = 1; i < n_ref; ++i) {
        const double dx = line[4 * i] - line[4 * (i - 1)], dy = line[4 * i + 1] - line[4 * (i - 1) + 1];
        acc = sqrt(dx * dx + dy * dy) + acc;
        s_map[i] = acc;
    }
    const double s0 = projection_s(node_at(line, m0), s_map[m0], ox, oy);
    for (int i = 0; i < n_ref; ++i) s_map[i] = s_map[i] - s0;
}

// ref: cal_s_l_deri_fun, planning_utils.py:538-586 for one point.  (px, py) is the position used for l
// (the reference passes origin_xy there, :542); proj is the projection of the point itself.
struct FrenetState {
    double l, l_dot, s_dot, l_ddot, dl_ds, s_ddot, ddl_ds;
};
EMP_HD FrenetState frenet_state(const Node& proj, double px, double py, double vx, double vy, double ax, double ay) {
    FrenetState o;
    const double c = cos(proj.theta), s = sin(proj.theta);
    const double k = proj.kappa;
    o.l = dot2(px - proj.x, py - proj.y, -s, c);
    o.l_dot = dot2(vx, vy, -s, c);
    o.s_dot = dot2(vx, vy, c, s) / (1.0 - k * o.l);
    o.l_ddot = dot2(ax, ay, -s, c) - k * (1.0 - k * o.l) * (o.s_dot * o.s_dot);
    o.dl_ds = (fabs(o.s_dot) < 1e-6) ? 0.0 : o.l_dot / o.s_dot;
    o.s_ddot = (dot2(ax, ay, c, s) + 2.0 * (o.s_dot * o.s_dot * k * o.dl_ds) + o.s_dot * o.s_dot * 0.0 * o.l) / (1.0 - k * o.l);
    o.ddl_ds = (fabs(o.s_dot) < 1e-6) ? 0.0 : (o.l_ddot - o.dl_ds * o.s_ddot) / (o.s_dot * o.s_dot);
    return o;
}

// ref: cal_proj_point, path_planning.py:52-75 - monotone walk; returns false where the reference would
// raise IndexError (s beyond the last s_map entry).
EMP_HD bool proj_point(const double* line, const double* s_map, int n_ref, double s, int* idx, Node* out) {
    int i = *idx;
    while (true) {
        if (i + 1 >= n_ref) return false;
        if (!(s_map[i + 1] < s)) break;
        ++i;
    }
    const Node m = node_at(line, i);
    const double ds = s - s_map[i];
    *out = Node{m.x + ds * cos(m.theta), m.y + ds * sin(m.theta), m.theta + m.kappa * ds, m.kappa};
    *idx = i;
    return true;
}

// ref: cal_heading_kappa, planning_utils.py:185-228.  xy interleaved [m][stride]; needs m >= 2.
EMP_HD void heading_kappa(const double* xy, int stride, int m, double* theta, int tstride, double* kappa, int kstride) {
    // theta first (node value = mean of adjacent forward differences, ends replicated)
    for (int i = 0; i < m; ++i) {
        const int a = (i - 1 > 0) ? i - 1 : 0, b = (i < m - 2) ? i : m - 2;
        const double dx = ((xy[(a + 1) * stride] - xy[a * stride]) + (xy[(b + 1) * stride] - xy[b * stride])) / 2.0;
        const double dy = ((xy[(a + 1) * stride + 1] - xy[a * stride + 1]) + (xy[(b + 1) * stride + 1] - xy[b * stride + 1])) / 2.0;
        theta[i * tstride] = atan2(dy, dx);
    }
    for (int i = 0; i < m; ++i) {
        const int a = (i - 1 > 0) ? i - 1 : 0, b = (i < m - 2) ? i : m - 2;
        const double dx = ((xy[(a + 1) * stride] - xy[a * stride]) + (xy[(b + 1) * stride] - xy[b * stride])) / 2.0;
        const double dy = ((xy[(a + 1) * stride + 1] - xy[a * stride + 1]) + (xy[(b + 1) * stride + 1] - xy[b * stride + 1])) / 2.0;
        const double dpre = theta[(a + 1) * tstride] - theta[a * tstride];
        const double daft = theta[(b + 1) * tstride] - theta[b * tstride];
        kappa[i * kstride] = sin((dpre + daft) / 2.0) / sqrt(dx * dx + dy * dy);
    }
}

}  // namespace emp
