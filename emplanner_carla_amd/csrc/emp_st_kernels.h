// emp_st_kernels.h - HIP kernels of the S-T speed DP (ref: planner/speed_planning_test.py:38-305).
//
// speed_dp_kernel: one workgroup of 320 lanes (5 wavefronts) per scene.
//   * The acceleration term of an edge depends on the DP state (the speed with which the best path enters
//     the source node, ref :216-218) - but only on the state of the PREVIOUS column, which is complete
//     when a column starts.  So all 40 x 40 candidates `cost(k, c-1) + edge(k -> j)` of a column are
//     computed in parallel (5 per lane) into an LDS table, including the expensive CalcObsCost (5 samples
//     x every S-T obstacle, ref :234-271); lanes 0..39 then take the first minimum over the source rows,
//     which is what the reference's ordered strict-< scan keeps (ref :138-152).
//   * Absent obstacles (NaN, ref :255) are squeezed out once, in order, when the scene is loaded.
//   * cost / s_dot / node tables live in LDS for the whole sweep (terminal search and backtrack read
//     them there) and are written to HBM once, coalesced.
// No HBM traffic besides the 4 x n_obs input doubles and the optional 3 x 640 table entries per scene:
// the kernel is FP64-VALU bound.
#pragma once

#include <hip/hip_runtime.h>

#include "emp_st_core.h"

namespace emp {

struct StDev {
    int B, max_obs;
    st::Weights w;
};

constexpr int kStBlock = 320;

inline size_t speed_dp_lds_bytes(int max_obs) {
    // obstacles (4 arrays + 3 frame arrays) | edge table | cost, s_dot tables | previous column (cost, s_dot) | node bytes
    return (7 * (size_t)max_obs + st::kRows * st::kRows + 2 * st::kRows * st::kCols + 2 * st::kRows) * sizeof(double) +
           st::kRows * st::kCols;
}

__global__ __launch_bounds__(kStBlock) void speed_dp_kernel(StDev d, const double* __restrict__ g_s_in,
                                                            const double* __restrict__ g_s_out,
                                                            const double* __restrict__ g_t_in,
                                                            const double* __restrict__ g_t_out,
                                                            const double* __restrict__ v_start, double* __restrict__ g_cost,
                                                            double* __restrict__ g_s_dot, int* __restrict__ g_node,
                                                            int* __restrict__ g_end, double* __restrict__ speed_s,
                                                            double* __restrict__ speed_t) {
    using namespace st;
    extern __shared__ double lds[];
    double* o_s_in = lds;
    double* o_s_out = o_s_in + d.max_obs;
    double* o_t_in = o_s_out + d.max_obs;
    double* o_t_out = o_t_in + d.max_obs;
    double* o_ux = o_t_out + d.max_obs;
    double* o_uy = o_ux + d.max_obs;
    double* o_len = o_uy + d.max_obs;
    double* tab = o_len + d.max_obs;            // [k][j] candidate cost(k, c-1) + edge (k, c-1) -> (j, c)
    double* t_cost = tab + kRows * kRows;       // [row][col]
    double* t_sdot = t_cost + kRows * kCols;    // [row][col]
    double* p_cost = t_sdot + kRows * kCols;    // previous column
    double* p_sdot = p_cost + kRows;
    unsigned char* t_node = reinterpret_cast<unsigned char*>(p_sdot + kRows);

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const size_t ob = (size_t)b * d.max_obs;
    __shared__ int n_live;
    if (tid < 64) {  // max_obs <= 64: one wavefront squeezes the present obstacles to the front, in order
        const bool has = tid < d.max_obs && !isnan(g_s_in[ob + (tid < d.max_obs ? tid : 0)]);
        const unsigned long long m = __ballot(has);
        if (has) {
            const int at = __popcll(m & (((unsigned long long)1 << tid) - 1));
            o_s_in[at] = g_s_in[ob + tid];
            o_s_out[at] = g_s_out[ob + tid];
            o_t_in[at] = g_t_in[ob + tid];
            o_t_out[at] = g_t_out[ob + tid];
            obs_frame(o_s_in[at], o_t_in[at], o_s_out[at], o_t_out[at], &o_ux[at], &o_uy[at], &o_len[at]);
        }
        if (tid == 0) n_live = __popcll(m);
    }
    const double v_origin = v_start[b];
    __syncthreads();
    const ObsSet obs{n_live, o_s_in, o_s_out, o_t_in, o_t_out, o_ux, o_uy, o_len};

    // ---- first column: every node is reached from the DP origin (0, 0) (ref :125-131) -------------
    if (tid < kRows) {
        const double s1 = s_of_row(tid), t1 = t_of_col(0);
        const double c = edge_cost(d.w, 0.0, 0.0, v_origin, s1, t1, obs, nullptr);
        const double v = s1 / t1;
        t_cost[tid * kCols] = c;
        t_sdot[tid * kCols] = v;
        t_node[tid * kCols] = 0;
        p_cost[tid] = c;
        p_sdot[tid] = v;
    }
    __syncthreads();

    for (int c = 1; c < kCols; ++c) {
        const double t1 = t_of_col(c), t_prev = t_of_col(c - 1);
        // ---- all 1600 candidates of this column (ref :138-145) ------------------------------------
        for (int e = tid; e < kRows * kRows; e += kStBlock) {
            const int k = e / kRows, j = e - k * kRows;
            const double s0 = k == 0 ? 0.0 : s_of_row(k);  // ref :208-212: source row 0 means "the origin"
            const double t0 = k == 0 ? 0.0 : t_prev;
            const double v0 = k == 0 ? v_origin : p_sdot[k];
            const double ec = edge_cost(d.w, s0, t0, v0, s_of_row(j), t1, obs, nullptr);
            tab[e] = ec + p_cost[k];
        }
        __syncthreads();
        // ---- first minimum over the source rows == the ordered strict-< scan from +inf (ref :145-152) --
        if (tid < kRows) {
            double best = INFINITY, best_v = 0.0;
            int best_k = 0;
            for (int k = 0; k < kRows; ++k) {
                const double cand = tab[k * kRows + tid];
                if (cand < best) {
                    best = cand;
                    best_k = k;
                }
            }
            // ref :148-150: the stored speed uses the real source node, even for k == 0
            if (best < INFINITY) best_v = (s_of_row(tid) - s_of_row(best_k)) * 2.0;  // / (t1 - t_prev), exactly 0.5
            else best_k = 0;
            t_cost[tid * kCols + c] = best;
            t_sdot[tid * kCols + c] = best_v;
            t_node[tid * kCols + c] = (unsigned char)best_k;
            p_cost[tid] = best;
            p_sdot[tid] = best_v;
        }
        __syncthreads();
    }

    // ---- tables out (optional), coalesced --------------------------------------------------------
    const size_t tb = (size_t)b * kRows * kCols;
    for (int i = tid; i < kRows * kCols; i += kStBlock) {
        if (g_cost) g_cost[tb + i] = t_cost[i];
        if (g_s_dot) g_s_dot[tb + i] = t_sdot[i];
        if (g_node) g_node[tb + i] = t_node[i];
    }
    // ---- terminal node and backtrack (ref :155-186; predecessor cast to int, s and t not aliased) --
    if (tid < kCols) {
        speed_s[(size_t)b * kCols + tid] = NAN;
        speed_t[(size_t)b * kCols + tid] = NAN;
    }
    __syncthreads();
    if (tid == 0) {
        int row, col;
        const bool ok = terminal_node([&](int r, int c) { return t_cost[r * kCols + c]; }, &row, &col);
        g_end[2 * b] = row;
        g_end[2 * b + 1] = col;
        if (ok) {
            for (;;) {
                speed_s[(size_t)b * kCols + col] = s_of_row(row);
                speed_t[(size_t)b * kCols + col] = t_of_col(col);
                if (col == 0) break;
                row = t_node[row * kCols + col];
                --col;
            }
        }
    }
}

// ref :38-98 - one scene per lane
__global__ void st_graph_kernel(int B, int n, const double* __restrict__ obs_s, const double* __restrict__ obs_l,
                                const double* __restrict__ obs_s_dot, const double* __restrict__ obs_l_dot,
                                double* __restrict__ s_in, double* __restrict__ s_out, double* __restrict__ t_in,
                                double* __restrict__ t_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t o = (size_t)b * n;
    st::st_graph(n, obs_s + o, obs_l + o, obs_s_dot + o, obs_l_dot + o, s_in + o, s_out + o, t_in + o, t_out + o);
}

// ref :191-271 - arbitrary edges (s0, t0, v0, s1, t1) against the scene's obstacles; one edge per lane,
// blockIdx.y = scene; the obstacle frames of the scene are built once per block in LDS (7 * max_obs doubles)
__global__ void st_edge_cost_kernel(StDev d, int n_edges, const double* __restrict__ edges,
                                    const double* __restrict__ s_in, const double* __restrict__ s_out,
                                    const double* __restrict__ t_in, const double* __restrict__ t_out,
                                    double* __restrict__ total, double* __restrict__ obs) {
    extern __shared__ double lds[];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const size_t o = (size_t)b * d.max_obs;
    double* f = lds;
    const int K = d.max_obs;
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        f[j] = s_in[o + j];
        f[K + j] = s_out[o + j];
        f[2 * K + j] = t_in[o + j];
        f[3 * K + j] = t_out[o + j];
        st::obs_frame(f[j], f[2 * K + j], f[K + j], f[3 * K + j], &f[4 * K + j], &f[5 * K + j], &f[6 * K + j]);
    }
    __syncthreads();
    if (e >= n_edges) return;
    const st::ObsSet set{K, f, f + K, f + 2 * K, f + 3 * K, f + 4 * K, f + 5 * K, f + 6 * K};
    const double* q = edges + ((size_t)b * n_edges + e) * 5;
    double oc;
    const double c = st::edge_cost(d.w, q[0], q[1], q[2], q[3], q[4], set, &oc);
    total[(size_t)b * n_edges + e] = c;
    if (obs) obs[(size_t)b * n_edges + e] = oc;
}

// ref :274-284
__global__ void st_collision_cost_kernel(int n, st::PowBase w, const double* __restrict__ dist, double* __restrict__ cost) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cost[i] = st::collision_cost(w, dist[i]);
}

}  // namespace emp
