// emp_st_kernels.h - HIP kernels of the S-T speed DP (ref: planner/speed_planning_test.py:38-305).
//
// speed_dp_kernel: one workgroup of 320 lanes (5 wavefronts) per scene.
//   * The acceleration term of an edge depends on the DP state (the speed with which the best path enters
//     the source node, ref :216-218) - but only on the state of the PREVIOUS column, which is complete
//     when a column starts.  So all 40 x 40 candidates `cost(k, c-1) + edge(k -> j)` of a column are
//     independent, and so is the expensive part of an edge, CalcObsCost (5 samples x every S-T obstacle,
//     ref :234-271), of everything the DP computes: see the kernel's own comment for how it is organised.
//   * Absent obstacles (NaN, ref :255) are squeezed out once, in order, when the scene is loaded.
//   * cost / s_dot of the previous column, the predecessor bytes and row 0's costs live in LDS (terminal search
//     and backtrack read them there); the optional tables go to HBM column by column.
// No HBM traffic besides the 4 x n_obs input doubles and the optional 3 x 640 table entries per scene:
// the kernel is FP64-VALU bound.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "emp_st_core.h"

namespace emp {

struct StDev {
    int B, max_obs;
    st::Weights w;
};

constexpr int kStBlock = 320;

// ---- speed_dp_kernel (round 3): near pairs found by interval tests, costed on compacted lists --------------------
//
// What an edge costs beyond six kinematic operations is CalcObsCost (ref :234-271): five samples against every
// S-T obstacle, a point-to-segment distance (sqrt, division) and a power for each pair.  All of it is independent
// of the DP state, and almost all of it is exactly 0 (the pair is 1.5 or more apart).  Two observations:
//   * the five sample TIMES of an edge depend only on the column and on whether the edge starts at a grid node
//     ("regular": t_prev + (m-1)/8) or at the DP origin (ref :208-212: (m-1) t1/4).  At a fixed time an obstacle
//     segment's reach is an interval of s (st::reach_interval), so ten intervals per obstacle and column - computed
//     by a few lanes while wavefront 0 takes the previous column's minima - replace the per-edge frame tests: a
//     pair is a candidate iff lo < s_m < hi.
//   * the candidate pairs (1.8 per edge on the benchmark scenes, but 4 for the busiest lane of a wavefront and 0 for
//     most) are written to a per-wavefront LDS list, costed 64 at a time by all lanes with the branch-free
//     st::point_cost_flat, and summed by their edge's lane in the reference's order (sample outer, obstacle inner):
//     the expensive code runs ceil(pairs / 64) times per wavefront instead of max-over-lanes(pairs) times.
// Mapping: lane = (kb, j) with j = tid % 40 the destination row and kb = tid / 40; the lane takes the source rows
// k = kb, kb + 8, .. kb + 32 in ascending order (interleaved, so that every wavefront sees source rows from the whole
// s range: obstacles are local in s, and contiguous bands left one wavefront with most of a column's pairs) and keeps
// the first minimum of its five candidates; lanes 0-39 then take the minimum over the eight partial results, the
// lowest k winning ties (= the reference's ordered strict-< scan, ref :138-152).
#ifndef EMP_ST_WAVES
#define EMP_ST_WAVES 5          // wavefronts per SIMD the register allocation aims for.  Round 6, per 4096 scenes alone: 4 (128 VGPRs) 2.6 ms,
                              // 5 (96) 2.02 ms, 6 (80) 2.02 ms and the same configs[4] step (round 3's kernel: 3.2 / 2.4 / 2.3 ms)
#endif
constexpr int kStListCap = 256;   // list entries per wavefront (a longer list is processed in windows)
// Development builds only (tools/st_phase_probe.py; results of a gutted kernel mean nothing, durations do): -DEMP_ST_PROBE=
//   1  no obstacle reaches any column (interval tests, node sums, pair lists all empty: the kinematic DP alone)
//   2  interval tests and node sums run, no pair is emitted or costed
//   3  pairs are emitted and summed, the cost of a window is not computed
//   5  the whole kernel, and speed_t [b][0..2] = passes of a wavefront with pairs, pairs, 64-pair cost rounds of the scene
#ifndef EMP_ST_PROBE
#define EMP_ST_PROBE 0
#endif
constexpr int kStWaves = kStBlock / 64;
constexpr int kStParts = kStBlock / st::kRows;   // 8 partial minima per destination row

__device__ __forceinline__ int st_wave_incl_sum(int v) {
    // inclusive prefix sum over the 64 lanes: Hillis-Steele inside the 16-lane rows, then the row totals travel
    // with row_bcast:15 / row_bcast:31 (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1 (bound_ctrl: a lane without a source reads 0 - the
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2  move folds into the add)
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
// LDS traffic between the lanes of ONE wavefront: DS operations execute in issue order; the fences keep the
// compiler from moving them across the hand-over.
__device__ __forceinline__ void st_wave_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long st_uniform64(unsigned long long v) {   // a wave-uniform value into scalar registers
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

struct StLds {   // carve-up of the dynamic LDS of speed_dp_kernel
    double *o_s_in, *o_s_out, *o_t_in, *o_t_out, *o_ux, *o_uy, *o_len;   // [max_obs] each, squeezed
    double* iv;       // [2][max_obs][5][2]  reach intervals (lo, hi) per class (0 = regular edges, 1 = edges from the origin), obstacle
                      //                     and sample: the eight bounds an edge tests against one obstacle are 80 contiguous bytes,
                      //                     fetched by one burst of LDS reads (sample-major they were four dependent round trips)
    double* node_c;   // [40][max_obs]     cost of the source NODES of a column against every obstacle (sample m = 1)
    double* t_tab;    // [10]              sample times of the two classes
    double* s_tab;    // [40]              s of the grid rows
    double* part_c;   // [8][40]           partial minima
    double* p_cost;   // [2][40]           previous / current column
    double* p_sdot;   // [2][40]
    double* row0;     // [16]              cost of row 0 in every column (terminal search)
    unsigned long long* colmask;   // [2][2] obstacles with a non-empty interval in the column: [column parity][class]
    double* list_s;   // [waves][cap + 1]  sample s of a pair, overwritten by the pair's cost
    uint32_t* list_c; // [waves][cap + 1]  obstacle | sample slot << 8
    unsigned char* part_k;  // [8][40]
    unsigned char* t_node;  // [40][16]
};
inline size_t speed_dp_lds_bytes(int max_obs) {
    return ((27 + st::kRows) * (size_t)max_obs + 10 + st::kRows + kStParts * st::kRows + 4 * st::kRows + st::kCols + 4 +
            kStWaves * (kStListCap + 1)) * sizeof(double) +
           kStWaves * (kStListCap + 1) * sizeof(uint32_t) + 4 + kStParts * st::kRows + st::kRows * st::kCols;
}
__device__ __forceinline__ StLds st_carve(double* lds, int max_obs) {
    StLds L;
    L.o_s_in = lds;
    L.o_s_out = L.o_s_in + max_obs;
    L.o_t_in = L.o_s_out + max_obs;
    L.o_t_out = L.o_t_in + max_obs;
    L.o_ux = L.o_t_out + max_obs;
    L.o_uy = L.o_ux + max_obs;
    L.o_len = L.o_uy + max_obs;
    L.iv = L.o_len + max_obs;
    L.node_c = L.iv + 20 * max_obs;
    L.t_tab = L.node_c + st::kRows * max_obs;
    L.s_tab = L.t_tab + 10;
    L.part_c = L.s_tab + st::kRows;
    L.p_cost = L.part_c + kStParts * st::kRows;
    L.p_sdot = L.p_cost + 2 * st::kRows;
    L.row0 = L.p_sdot + 2 * st::kRows;
    L.colmask = reinterpret_cast<unsigned long long*>(L.row0 + st::kCols);
    L.list_s = L.row0 + st::kCols + 4;
    L.list_c = reinterpret_cast<uint32_t*>(L.list_s + kStWaves * (kStListCap + 1));
    L.part_k = reinterpret_cast<unsigned char*>(L.list_c + kStWaves * (kStListCap + 1));
    L.t_node = L.part_k + kStParts * st::kRows;
    return L;
}

// The pairs of one list window, 64 at a time: entry p = (sample s, obstacle | sample slot << 8) becomes the pair's cost.
// obs = the seven squeezed obstacle arrays.
__device__ __forceinline__ void st_cost_window(const st::PowBase& w, double* list_s, const uint32_t* list_c, const double* t_tab,
                                               const double* obs, int max_obs, int n, int lane) {
#pragma unroll 1
    for (int p = lane; p < n; p += 64) {
        const uint32_t code = list_c[p];
        const int jj = code & 255u;
        list_s[p] = st::point_cost_flat(w, list_s[p], t_tab[code >> 8], obs[jj], obs[2 * max_obs + jj], obs[max_obs + jj],
                                        obs[3 * max_obs + jj]);
    }
}

// What waves 1-4 prepare for destination column c while wavefront 0 takes the previous column's minima (nothing here
// depends on the DP state), by `nthreads` lanes numbered `x0`:
//   * reach intervals of every obstacle at the ten sample times of the column, and the column's obstacle mask;
//   * the cost of the column's 40 source NODES against every obstacle.  Sample m = 1 of an edge is its source node
//     itself (ref :251-252 with i - 1 = 0: s0 + (k 0) dt = s0, t0 + 0 dt = t0) whatever the destination row: computed
//     here once per (node, obstacle) instead of once per edge, and added by the edges in the reference's place in the
//     order of additions (after the pairs of sample 0, obstacle by obstacle; the exact zeros change nothing).
__device__ __forceinline__ void st_column_setup(const StDev& d, const StLds& L, int n_live, int c, int x0, int nthreads) {
    using namespace st;
    const int MO = d.max_obs;
    const double t1 = t_of_col(c);
    for (int x = x0; x < 10 * n_live; x += nthreads) {
        const int slot = x / n_live, j = x - slot * n_live;
        const int cls = slot / kStSamples, m = slot - cls * kStSamples;
        const double t0 = cls ? 0.0 : t_of_col(c - 1);
        const double dt = (t1 - t0) * 0.25;
        const double t = t0 + (double)(m - 1) * dt;
        double lo, hi;
        bool some = reach_interval(t, L.o_s_in[j], L.o_t_in[j], L.o_ux[j], L.o_uy[j], L.o_len[j], &lo, &hi);
        if (c == 0 && cls == 0) some = false;   // the first column has no regular edges
        if (!some || m == 1) {                  // sample 1 goes through node_c (its obstacle still counts for the mask)
            lo = INFINITY;
            hi = -INFINITY;
        }
        L.iv[((cls * MO + j) * kStSamples + m) * 2] = lo;
        L.iv[((cls * MO + j) * kStSamples + m) * 2 + 1] = hi;
        if (some) atomicOr(&L.colmask[(c & 1) * 2 + cls], (unsigned long long)1 << j);
    }
    if (x0 < 10) {
        const int cls = x0 / kStSamples, m = x0 - cls * kStSamples;
        const double t0 = cls ? 0.0 : t_of_col(c > 0 ? c - 1 : 0);
        const double dt = (t1 - t0) * 0.25;
        L.t_tab[x0] = t0 + (double)(m - 1) * dt;      // the expression of st::obs_cost, ref :251
    }
    // nodes: x = obstacle * 40 + row, so that a wavefront covers one or two obstacles and skips those out of reach in time
    const double t_prev = c > 0 ? t_of_col(c - 1) : 0.0;
    for (int x = x0; x < kRows * n_live; x += nthreads) {
        const int j = x / kRows, k = x - j * kRows;
        const double s = k == 0 ? 0.0 : L.s_tab[k];     // ref :208-212: source row 0 means "the origin"
        const double t = k == 0 ? 0.0 : t_prev;
        const double ti = L.o_t_in[j], to = L.o_t_out[j];
        double cost = 0.0;
        // a point farther than the 1.5 reach from the segment's time span costs exactly 0 (NaN keeps the exact path)
        if (!(t < fmin(ti, to) - kPruneGap || t > fmax(ti, to) + kPruneGap))
            cost = point_cost_flat(d.w.w_obs, s, t, L.o_s_in[j], ti, L.o_s_out[j], to);
        L.node_c[k * MO + j] = cost;
    }
}

template <typename MaskT>
__global__ __launch_bounds__(kStBlock, EMP_ST_WAVES) void speed_dp_kernel(StDev d, const double* __restrict__ g_s_in,
                                                            const double* __restrict__ g_s_out,
                                                            const double* __restrict__ g_t_in,
                                                            const double* __restrict__ g_t_out,
                                                            const double* __restrict__ v_start, double* __restrict__ g_cost,
                                                            double* __restrict__ g_s_dot, int* __restrict__ g_node,
                                                            int* __restrict__ g_end, double* __restrict__ speed_s,
                                                            double* __restrict__ speed_t, const int* __restrict__ order) {
    using namespace st;
    extern __shared__ double lds[];
    const int MO = d.max_obs;
    const StLds L = st_carve(lds, MO);
    const int b = order ? order[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ob = (size_t)b * MO;
    __shared__ int n_live_s;
    __shared__ int probe_cnt[3];
    if (EMP_ST_PROBE == 5 && threadIdx.x < 3) probe_cnt[threadIdx.x] = 0;
    if (tid < 64) {  // max_obs <= 64: one wavefront squeezes the present obstacles to the front, in order (ref :255)
        const bool has = tid < MO && !isnan(g_s_in[ob + (tid < MO ? tid : 0)]);
        const unsigned long long m = __ballot(has);
        if (has) {
            const int at = __popcll(m & (((unsigned long long)1 << tid) - 1));
            L.o_s_in[at] = g_s_in[ob + tid];
            L.o_s_out[at] = g_s_out[ob + tid];
            L.o_t_in[at] = g_t_in[ob + tid];
            L.o_t_out[at] = g_t_out[ob + tid];
            obs_frame(L.o_s_in[at], L.o_t_in[at], L.o_s_out[at], L.o_t_out[at], &L.o_ux[at], &L.o_uy[at], &L.o_len[at]);
        }
        if (tid == 0) {
            n_live_s = __popcll(m);
            L.colmask[0] = L.colmask[1] = L.colmask[2] = L.colmask[3] = 0;
        }
    }
    const double v_origin = v_start[b];
    if (tid < kRows) {
        L.p_cost[kRows + tid] = 0.0;   // "previous column" of column 0: the origin, cost 0 (x + 0.0 == x)
        L.s_tab[tid] = s_of_row(tid);
    }
    __syncthreads();
    const int n_live = n_live_s;
    st_column_setup(d, L, n_live, 0, tid, kStBlock);

    const int j = tid % kRows, kb = tid / kRows;
    const double s1 = L.s_tab[j];
    double* my_s = L.list_s + wave * (kStListCap + 1);      // [cap] + one spare entry that takes the writes outside a window
    uint32_t* my_c = L.list_c + wave * (kStListCap + 1);
    const size_t tb = (size_t)b * kRows * kCols;

#pragma unroll 1
    for (int c = 0; c < kCols; ++c) {
        __syncthreads();   // [B] intervals / node costs of column c, and the previous column's cost / speed, are in place
        const int cur = c & 1, prev = cur ^ 1;
        const double t1 = t_of_col(c);
        // obstacles within reach of the column's regular edges, and of its edges from the origin (whose samples span
        // the whole time from 0: nearly every obstacle); only wavefront 0 has lanes of the second kind (k == 0)
        const unsigned long long mask_reg = st_uniform64(L.colmask[cur * 2]);
        const unsigned long long mask_all = mask_reg | st_uniform64(L.colmask[cur * 2 + 1]);
        if (tid == 0) L.colmask[prev * 2] = L.colmask[prev * 2 + 1] = 0;   // the next column's setup ORs into them after [C]
        double best = INFINITY;
        int best_k = 0;
        // One pass = one source row per lane (five per lane and column; only the origin in column 0).  GENERAL passes hold edges
        // from the DP origin (k == 0: the first pass of wavefront 0, where lanes 0-39 start at the origin and lanes 40-63 at row
        // 1; all of column 0) and select per lane; every other pass is REGULAR - every lane an edge between two grid columns, whose
        // t1 - t0 is 0.5 exactly: dt = 0.125, (s1 - s0) / 0.5 = (s1 - s0) * 2 (st::div_dt), one class of intervals, no selects.
        // Same operations on the same operands either way (round 6: the regular form is 30 instructions a pass shorter).
        auto edge_pass = [&](auto general_tag, const int i) {
            constexpr bool GENERAL = decltype(general_tag)::value;
            const int k = kb + kStParts * i;          // interleaved: every wavefront sees source rows from the whole s range
            const bool active = !GENERAL || c > 0 || tid < kRows;
            const bool from_origin = GENERAL && k == 0;   // ref :208-212 (and every edge of column 0, ref :125-131)
            unsigned long long colmask = GENERAL ? mask_all : mask_reg;
            if (EMP_ST_PROBE == 1) colmask &= (unsigned long long)(d.B < 0);   // opaque zero
            const double s0 = from_origin ? 0.0 : L.s_tab[k];
            const double t0 = from_origin ? 0.0 : t_of_col(c - 1);
            const double v0 = from_origin ? v_origin : L.p_sdot[prev * kRows + k];
            const double dt = GENERAL ? (t1 - t0) * 0.25 : 0.125;
            // (s1 - s0) / (t1 - t0): only edges from the origin divide
            const double ks = GENERAL ? div_dt(s1 - s0, t1 - t0) : (s1 - s0) * 2.0;
            const int slot0 = from_origin ? kStSamples : 0;
            auto s_m = [&](int m) { return s0 + (ks * (double)(m - 1)) * dt; };   // ref :252
            // ---- candidate pairs of samples 0, 2, 3, 4: lo < s_m < hi --------------------------------------
            MaskT mask[kStSamples];
#pragma unroll
            for (int m = 0; m < kStSamples; ++m) mask[m] = 0;
            const double* ivl = L.iv + (size_t)slot0 * MO * 2;
            const double sm0 = s_m(0), sm2 = s_m(2), sm3 = s_m(3), sm4 = s_m(4);
            const double* my_nodes = L.node_c + k * MO;
            MaskT node_nz = 0;                          // obstacles the source node has a non-zero cost against (sample 1)
            for (unsigned long long rest = colmask; rest; rest &= rest - 1) {
                const int jj = ctz64(rest);
                const double* q = ivl + jj * (2 * kStSamples);
                const double lo0 = q[0], hi0 = q[1], lo2 = q[4], hi2 = q[5], lo3 = q[6], hi3 = q[7], lo4 = q[8], hi4 = q[9];
                const double nc = my_nodes[jj];
                const MaskT bit = (MaskT)1 << jj;
                mask[0] |= ((sm0 > lo0) & (sm0 < hi0)) ? bit : (MaskT)0;   // '&': no short-circuit branch
                mask[2] |= ((sm2 > lo2) & (sm2 < hi2)) ? bit : (MaskT)0;
                mask[3] |= ((sm3 > lo3) & (sm3 < hi3)) ? bit : (MaskT)0;
                mask[4] |= ((sm4 > lo4) & (sm4 < hi4)) ? bit : (MaskT)0;
                node_nz |= (nc != 0.0) ? bit : (MaskT)0;                    // NaN counts; x + 0.0 == x for the sums of costs here
            }
            if (!active) mask[0] = mask[2] = mask[3] = mask[4] = 0;       // column 0: only the 40 edges from the origin exist
            const int cnt0 = sizeof(MaskT) == 8 ? __popcll((unsigned long long)mask[0]) : __popc((unsigned)mask[0]);
            int cnt = cnt0;
#pragma unroll
            for (int m = 2; m < kStSamples; ++m) cnt += sizeof(MaskT) == 8 ? __popcll((unsigned long long)mask[m]) : __popc((unsigned)mask[m]);
            const int incl = st_wave_incl_sum(cnt);
            int total = __builtin_amdgcn_readlane(incl, 63);
            if (EMP_ST_PROBE == 2) total *= (d.B < 0);
            if (EMP_ST_PROBE == 5 && lane == 0 && total > 0) {
                atomicAdd(&probe_cnt[0], 1);
                atomicAdd(&probe_cnt[1], total);
                atomicAdd(&probe_cnt[2], (total + 63) / 64);
            }
            const int off = incl - cnt;
            double obs = 0.0;
            bool nodes_done = false;
            auto add_nodes = [&]() {                  // sample 1: the source node against the column's obstacles, in order -
                for (MaskT rest = node_nz; rest; rest &= rest - 1)          // the exact zeros (nearly all of them) left out
                    obs = obs + my_nodes[sizeof(MaskT) == 8 ? ctz64((uint64_t)rest) : __ffs((unsigned)rest) - 1];
                nodes_done = true;
            };
#pragma unroll 1
            for (int base = 0; base < total; base += kStListCap) {
                // ---- emit this window's pairs, each lane its own in (sample, obstacle) order ----------------
                int idx = off - base;
#pragma unroll
                for (int m = 0; m < kStSamples; ++m) {
                    for (MaskT rest = mask[m]; rest; rest &= rest - 1) {
                        const int jj = sizeof(MaskT) == 8 ? ctz64((uint64_t)rest) : __ffs((unsigned)rest) - 1;
                        const int at = (unsigned)idx < (unsigned)kStListCap ? idx : kStListCap;   // outside the window: the spare entry
                        my_s[at] = s_m(m);
                        my_c[at] = (uint32_t)jj | (uint32_t)(slot0 + m) << 8;
                        ++idx;
                    }
                }
                st_wave_handover();
                // ---- cost them, 64 at a time -----------------------------------------------------------------
                st_cost_window(d.w.w_obs, my_s, my_c, L.t_tab, L.o_s_in, MO, EMP_ST_PROBE == 3 ? (d.B < 0) : min(kStListCap, total - base), lane);
                st_wave_handover();
                // ---- ordered sum of the lane's own pairs (ref :249-269: sample outer, obstacle inner) -------
                const int lo0 = off - base, mid = lo0 + cnt0, hi0 = lo0 + cnt;
                for (int q = max(lo0, 0); q < min(mid, kStListCap); ++q) obs = obs + my_s[q];          // sample 0
                if (!nodes_done && mid <= kStListCap) add_nodes();                                      // sample 1
                for (int q = max(mid, 0); q < min(hi0, kStListCap); ++q) obs = obs + my_s[q];          // samples 2-4
                st_wave_handover();
            }
            if (!nodes_done) add_nodes();
            double acc, ref;
            if (GENERAL) {
                kinematic_cost(d.w, s0, t0, v0, s1, t1, &acc, &ref);
            } else {                                  // st::kinematic_cost with t1 - t0 == 0.5: v = ks, a = (v - v0) * 2
                const double a = (ks - v0) * 2.0, e = ks - d.w.v_ref, a2 = a * a;
                ref = d.w.w_ref * (e * e);
                acc = (4.0 > a && a > -6.0) ? d.w.w_acc * a2 : (100000.0 * d.w.w_acc) * a2;
            }
            const double cand = ((obs + acc) + ref) + L.p_cost[prev * kRows + k];
            if (c == 0) best = cand;                  // column 0 stores the edge cost as it is
            else if (cand < best) {
                best = cand;
                best_k = k;
            }
        };
        int i_first = 0;
        if (c == 0 || wave == 0) {                    // (column 0: wavefronts 1-4 have no edge at all)
            if (wave == 0) edge_pass(std::true_type{}, 0);
            i_first = 1;
        }
        const int i_end = c == 0 ? 0 : kStSamples;
#pragma unroll 1
        for (int i = i_first; i < i_end; ++i) edge_pass(std::false_type{}, i);
        L.part_c[tid] = best;
        L.part_k[tid] = (unsigned char)best_k;
        __syncthreads();   // [C] partial minima complete; nobody reads this column's intervals any more
        if (tid < kRows) {
            double v = L.part_c[tid], best_v;
            int kk = 0;
            if (c == 0) {
                best_v = s1 / t1;                     // ref :129
            } else {
                v = INFINITY;
                for (int part = 0; part < kStParts; ++part) {
                    const double x = L.part_c[part * kRows + tid];
                    const int xk = L.part_k[part * kRows + tid];
                    if (x < v || (x == v && xk < kk)) {   // the partial minima interleave the source rows: lowest k wins a tie
                        v = x;
                        kk = xk;
                    }
                }
                // ref :148-150: the stored speed uses the real source node, even for k == 0
                if (v < INFINITY) best_v = (s1 - L.s_tab[kk]) * 2.0;   // / (t1 - t_prev), exactly 0.5
                else {
                    best_v = 0.0;
                    kk = 0;
                }
            }
            L.p_cost[cur * kRows + tid] = v;
            L.p_sdot[cur * kRows + tid] = best_v;
            L.t_node[tid * kCols + c] = (unsigned char)kk;
            if (tid == 0) L.row0[c] = v;
            if (g_cost) g_cost[tb + tid * kCols + c] = v;
            if (g_s_dot) g_s_dot[tb + tid * kCols + c] = best_v;
            if (g_node) g_node[tb + tid * kCols + c] = kk;
        } else if (tid >= 64 && c + 1 < kCols) {
            st_column_setup(d, L, n_live, c + 1, tid - 64, kStBlock - 64);
        }
    }
    __syncthreads();
    // ---- terminal node and backtrack (ref :155-186; predecessor cast to int, s and t not aliased) --
    if (tid < kCols) {
        speed_s[(size_t)b * kCols + tid] = NAN;
        speed_t[(size_t)b * kCols + tid] = NAN;
    }
    __syncthreads();
    if (tid == 0) {
        int row, col;
        const double* last = L.p_cost + ((kCols - 1) & 1) * kRows;
        const bool ok = terminal_node([&](int r, int cc) { return cc == kCols - 1 ? last[r] : L.row0[cc]; }, &row, &col);
        g_end[2 * b] = row;
        g_end[2 * b + 1] = col;
        if (ok) {
            for (;;) {
                speed_s[(size_t)b * kCols + col] = s_of_row(row);
                speed_t[(size_t)b * kCols + col] = t_of_col(col);
                if (col == 0) break;
                row = L.t_node[row * kCols + col];
                --col;
            }
        }
        if (EMP_ST_PROBE == 5)
            for (int x = 0; x < 3; ++x) speed_t[(size_t)b * kCols + x] = (double)probe_cnt[x];
    }
}

// ---- heaviest scenes first -----------------------------------------------------------------------------------
// A scene's cost grows with its number of S-T obstacles (0.6 ms per 4096 scenes without any, +0.7 ms per obstacle), a
// launch holds a few blocks per CU at a time, and the blocks start in index order: with the scenes in input order the
// last heavy ones run on an otherwise idle chip (measured: a quarter of the kernel's duration).  Two small kernels
// build a permutation with the scenes sorted by obstacle count, descending (a counting sort; ties in arrival order
// of the atomics, which changes nothing in any result), and block i of speed_dp_kernel takes scene order[i].
constexpr int kStKeys = st::kMaxObs + 1;
__global__ void st_count_kernel(int B, int max_obs, const double* __restrict__ s_in, unsigned char* __restrict__ key,
                                int* __restrict__ hist) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int j = 0; j < max_obs; ++j) n += !isnan(s_in[(size_t)b * max_obs + j]);
    key[b] = (unsigned char)n;
    atomicAdd(&hist[n], 1);
}
__global__ void st_scatter_kernel(int B, const unsigned char* __restrict__ key, const int* __restrict__ hist,
                                  int* __restrict__ cursor, int* __restrict__ order) {
    __shared__ int start[kStKeys];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = kStKeys - 1; k >= 0; --k) {
            start[k] = acc;
            acc += hist[k];
        }
    }
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int k = key[b];
    order[start[k] + atomicAdd(&cursor[k], 1)] = b;
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

// ref :23-35: tangential speed and acceleration of the planning start (one lane per scene)
__global__ void st_start_condition_kernel(int n, const double* __restrict__ vx, const double* __restrict__ vy, const double* __restrict__ ax,
                                          const double* __restrict__ ay, const double* __restrict__ heading, double* __restrict__ s_dot,
                                          double* __restrict__ s_dot2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double c = cos(heading[i]), s = sin(heading[i]);
    s_dot[i] = c * vx[i] + s * vy[i];
    s_dot2[i] = c * ax[i] + s * ay[i];
}

}  // namespace emp
