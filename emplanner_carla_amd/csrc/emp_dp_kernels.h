// emp_dp_kernels.h - HIP kernels of the S-L lattice DP for gfx950 (wave64).
//
// Scene tiling.  A lattice column has `row` nodes; S = 64 / row scenes are packed side by side
// in one wavefront (lane = s * row + i, i = destination row), so every lane of the min-plus sweep
// does useful work and every load of the edge tensor is one fully coalesced 512-byte row:
//
//     edge_tiled[tile][j-1][k][lane]      tile = scene / S, lane = (scene % S) * row + i,  64 lanes
//
// (k = source row in column j-1, j = 1..col-1).  Lanes >= S*row are padding.
//
// Kernels
//   dp_edge_kernel     edge costs (ref: cal_start_cost / cal_neighbor_cost, path_planning.py:435-585),
//                      FP64 VALU bound; writes the tiled (or canonical) tensor, 8 B per lane coalesced.
//   dp_sweep_kernel    min-plus sweep + argmin + backtrack (ref: path_planning.py:301-361), HBM bound:
//                      streams the tiled tensor once through a register ring of PD prefetched columns.
//   dp_enrich_wave_kernel   row indices -> densified (s, l) path (ref: path_planning.py:364-432).
#pragma once

#include <hip/hip_runtime.h>

#include "emp_core.h"

namespace emp {

struct DpDev {
    int row, col;
    int S;       // scenes per wavefront tile = 64 / row
    int tiles;   // ceil(B / S)
    int B;
    int max_obs;
    double sample_s, sample_l, res;
    double w_coll, w0, w1, w2, w_ref;
};

// Pair-table fields behind the kSamples lateral samples (round 5: the three quintic coefficients gave way to ONE jerk
// weight - see kF_JERK - which takes the table from 17 to 15 fields: 53 instead of 60 KB of LDS at 21 rows)
constexpr int kF_JERK = kSamples + 0;     // w2 (h h), h = l_cur - l_pre: the quirked jerk term of the edge is this times F(s0)
constexpr int kF_BASE = kSamples + 1;     // w0 sum dl^2 + w1 sum ddl^2
constexpr int kF_REF = kSamples + 2;      // w_ref sum l^2
constexpr int kF_LLO = kSamples + 3;      // min(l_pre, l_cur)
constexpr int kF_LHI = kSamples + 4;      // max(l_pre, l_cur)
constexpr int kTableFields = kSamples + 5;
// behind the fields: the kSamples sample offsets t_n, their two moments (emp_core.h sample_moments) and the UNIT quintic's
// a3, a4, a5 (the coefficients of the neighbour edge with h = 1)
constexpr int kUnitQuintic = 3;
constexpr int kTableTail = kSamples + kSampleMoments + kUnitQuintic;

// The quirked jerk sum of a NEIGHBOUR edge, factorised (round 5).  A neighbour edge starts with dl = ddl = 0, so its
// shifted coefficients are h times the unit quintic's (a3, a4, a5) = h (u3, u4, u5), h = l_cur - l_pre; the absolute-s
// coefficients c3, c4, c5 the reference's third-derivative term needs (path_planning.py:571) are linear in (a3, a4, a5), the
// term is linear in them, and its sum of squares over the ten samples is therefore h^2 times a function of the edge's start
// abscissa alone:   S_dddl(k -> i, s0) = (h h) F(s0),   F(s0) = jerk_quirk_sum(unit quintic, s0).
// F is evaluated once per (scene, column) - 28 vector instructions - instead of once per edge; per edge one multiplication
// by the tabulated w2 (h h) remains.  Same mathematics as jerk_quirk_sum on the edge's own coefficients (rounds 2-4), rounded
// differently in the last bits; oracle/exact.py states the same operation order and stays the bit-exact target.
__device__ __forceinline__ double jerk_unit_sum(const double* tail, double s0) {
    Quintic u;
    u.a0 = u.a1 = u.a2 = 0.0;
    u.a3 = tail[kSamples + kSampleMoments + 0];
    u.a4 = tail[kSamples + kSampleMoments + 1];
    u.a5 = tail[kSamples + kSampleMoments + 2];
    return jerk_quirk_sum(u, s0, tail[kSamples], tail[kSamples + 1]);
}

// ---------------------------------------------------------------------------------------------
// edge costs
// ---------------------------------------------------------------------------------------------
// Pair table: everything of a neighbour edge that depends neither on the scene nor on the column
// (dl0 = ddl0 = 0, T = sample_s: the lateral samples, sum l^2, sum dl^2, sum ddl^2, a3..a5 are functions of the
// row pair (k, i) only; the quirked jerk term and the obstacles are what see the absolute s).  It depends only
// on the lattice parameters, so it is built once per parameter set by this one-block kernel and kept in device
// memory: [kTableFields][row*row] doubles (pair index = k*row + i) followed by the kSamples sample offsets t_n,
// their two moments sum t_n, sum t_n^2 (emp_core.h sample_moments) and the unit quintic's a3, a4, a5 (kTableTail doubles).
__global__ __launch_bounds__(256) void dp_pair_table_kernel(DpDev P, double* __restrict__ tab) {
    const int row = P.row, rr = P.row * P.row;
    for (int p = threadIdx.x; p < rr; p += blockDim.x) {
        const int k = p / row, i = p - k * row;
        const double l_pre = lattice_l(row, k, P.sample_l);
        const double l_cur = lattice_l(row, i, P.sample_l);
        const Quintic q = quintic_shifted(l_pre, 0.0, 0.0, l_cur, P.sample_s);
        double S_l = 0.0, S_dl = 0.0, S_ddl = 0.0;
        for (int n = 0; n < kSamples; ++n) {
            const double t = sample_t(n, P.sample_s);
            const double l = quintic_l(q, t);
            const double dl = quintic_dl(q, t);
            const double ddl = quintic_ddl(q, t);
            tab[n * rr + p] = l;
            S_l = S_l + l * l;
            S_dl = S_dl + dl * dl;
            S_ddl = S_ddl + ddl * ddl;
        }
        const double h = l_cur - l_pre;                     // as quintic_shifted
        tab[kF_JERK * rr + p] = P.w2 * (h * h);
        tab[kF_BASE * rr + p] = P.w0 * S_dl + P.w1 * S_ddl;
        tab[kF_REF * rr + p] = P.w_ref * S_l;
        tab[kF_LLO * rr + p] = fmin(l_pre, l_cur);
        tab[kF_LHI * rr + p] = fmax(l_pre, l_cur);
    }
    if (threadIdx.x < kSamples) tab[kTableFields * rr + threadIdx.x] = sample_t(threadIdx.x, P.sample_s);
    if (threadIdx.x == 0) {
        sample_moments(P.sample_s, &tab[kTableFields * rr + kSamples], &tab[kTableFields * rr + kSamples + 1]);
        const Quintic u = quintic_shifted(0.0, 0.0, 0.0, 1.0, P.sample_s);
        tab[kTableFields * rr + kSamples + kSampleMoments + 0] = u.a3;
        tab[kTableFields * rr + kSamples + kSampleMoments + 1] = u.a4;
        tab[kTableFields * rr + kSamples + kSampleMoments + 2] = u.a5;
    }
}

// kSoftGain / d2 for 16 < d2 < 36, equal to the IEEE binary64 quotient except with probability ~2^-43 per operand (NOT a
// proof of correct rounding): the instruction sequence the compiler emits for an IEEE division (reciprocal seed, Newton
// steps, quotient, residual, final fma) without its range scaling and special-case fix-up, which do nothing for operands
// of this size, and (round 3) with ONE Newton step instead of two: six instructions instead of twelve, eight to nine
// times per obstacle scan.  One step takes the seed (2^-24 or better) to 2^-48 or better; the quotient q = 5000 r then
// carries that error, the residual 5000 - d2 q is exact to 2^-53 of itself (fma), and the correction r (5000 - d2 q) is
// wrong by q 2^-96 at most - so the final fma returns the NEIGHBOUR of the correctly rounded quotient only if the true
// quotient lies within 2^-96 q of a rounding boundary: one operand in 2^43, i.e. about one soft term in 10^4 batches of
// 4096 scenes (1e9 divisions each), and then by one unit in the last place (1.1e-16 relative against the 1e-6 bar; an
// argmin moves only if two path costs tie to that unit).  Evidence, not proof: tools/soft_quotient_test.hip checks 8.6e9
// operands of the interval (an even sweep and a hashed one) against the compiler's IEEE division bit for bit, as a test
// of the GPU suite; every edge tensor of the suite and the rows of a 262144-scene DP sweep
// (profiles/r03_final_parity_sweep_dp_262144.json) equal the oracle's.  Building with -DEMP_SOFT_NEWTON_STEPS=2 restores
// the compiler's own (proven) sequence at +4 % of the edge kernel.  With the shorter block the compiler would drop the
// wave-level skip branch around it (its threshold is twelve instructions) and execute all ten divisions of a scan under
// masks - 8 % slower than the long form; obstacle_scan_dense keeps the branch with an empty volatile asm.
#ifndef EMP_SOFT_NEWTON_STEPS
#define EMP_SOFT_NEWTON_STEPS 1
#endif
__device__ __forceinline__ double soft_cost_quotient(double d2) {
    double r = __builtin_amdgcn_rcp(d2);
    r = __builtin_fma(r, __builtin_fma(-d2, r, 1.0), r);
    if (EMP_SOFT_NEWTON_STEPS > 1) r = __builtin_fma(r, __builtin_fma(-d2, r, 1.0), r);
    const double q = kSoftGain * r;
    return __builtin_fma(__builtin_fma(-d2, q, kSoftGain), r, q);
}

// ref: cal_obs_cost (path_planning.py:588-609) for one obstacle in DENSE form: all ten squared distances first
// (straight-line code: the ten lateral samples are read from LDS back to back and the fifty operations overlap), then
// the ordered accumulation.  Same operations in the same order as emp_core.h obstacle_cost: the soft terms 5000 / d2
// of the samples before the first hard hit, ascending, then the collision weight once (nothing is added after the
// hit, so adding it behind the loop is the reference's order).  `l_n` points at the pair's first lateral sample,
// consecutive samples are `stride` doubles apart.  Against the nested form with the early break (one LDS read, one
// wait and two exec-mask regions per sample, each nested in the previous one) this issues 3 % more vector
// instructions and 45 % fewer scalar ones, and the kernel went from 175 to 157 us (profiles/r02_edge_variants.md).
#ifndef EMP_SCAN_GROUP
#define EMP_SCAN_GROUP 10          // samples whose distances are formed together (10: all of them first; 5: two halves - ten
#endif                             // registers fewer in flight, the same operations in the same order)
__device__ __forceinline__ double obstacle_scan_dense(double s0, const double* t_smp, const double* l_n, int stride,
                                                      double os, double ol, double w_coll) {
    double c = 0.0;
    bool alive = true;
#pragma unroll
    for (int g = 0; g < kSamples; g += EMP_SCAN_GROUP) {
        double d2[EMP_SCAN_GROUP];
#pragma unroll
        for (int n = 0; n < EMP_SCAN_GROUP; ++n) {
            const double d_lon = os - (s0 + t_smp[g + n]);        // the sample abscissa, ref :493/:566 (rebuilt, not kept: registers)
            const double d_lat = ol - l_n[(g + n) * stride];
            d2[n] = d_lon * d_lon + d_lat * d_lat;
        }
#pragma unroll
        for (int n = 0; n < EMP_SCAN_GROUP; ++n) {
            const bool near = alive & (d2[n] < kSafe2);
            const bool hard = near & (d2[n] <= kDanger2);
            if (near & !hard) {
#if EMP_SOFT_NEWTON_STEPS == 1
                asm volatile("");          // keeps the wave-level skip branch around the (now short) division block
#endif
                c = c + soft_cost_quotient(d2[n]);
            }
            alive = alive & !hard;
        }
    }
    if (!alive) c = c + w_coll;
    return c;
}

// Exact pruning of an (edge, obstacle) pair: the ten samples of a neighbour edge lie in the box [s0, s9] x [l_lo, l_hi]
// (the lateral profile l0 + h (10 u^3 - 15 u^4 + 6 u^5) is monotone between the two rows), so an obstacle whose distance
// to the box is 6.04 m or more (squared: 36.5, a margin that dwarfs every rounding error) is farther than safe_dis = 6 m
// from every sample and contributes exactly 0.
__device__ __forceinline__ bool obstacle_box_in_reach(double os, double ol, double s0, double s9, double l_lo, double l_hi) {
    const double dx = fmax(fmax(s0 - os, os - s9), 0.0);
    const double dy = fmax(fmax(l_lo - ol, ol - l_hi), 0.0);
    return dx * dx + dy * dy < 36.5;
}

// The `row` neighbour edges from column j-1 into row i of column j for one scene (ref: cal_neighbor_cost,
// path_planning.py:517-585), handed to `store(k, cost)` for k = 0..row-1 (k = source row).  What depends on (scene,
// column) only - the sample abscissae and which obstacles are within longitudinal reach - is set up once and reused
// for the source rows.  tab / t_smp: the pair table and the sample offsets in LDS; my_obs_s / my_obs_l: the scene's
// obstacles in LDS; ps = plan_start_s; nob = the scene's obstacle count (clamped to the row's capacity).
// ROW > 0: the lattice's row count at compile time (round 4: every offset into the pair table - (field) row^2 + k row + i, and the ten
// lateral samples of a scan row^2 doubles apart - is then an immediate of the LDS instruction instead of vector arithmetic in
// front of it); ROW == 0: any row count (the generic, fused and wide kernels).
// MASK: the type of the per-column reach mask - 64 bits, or 32 when the call's obstacle rows hold at most 32 slots (the first-set-bit
// walk over a 64-bit mask in vector registers costs twice the instructions; obstacles beyond the mask's width take the full test
// per edge either way).
// box_dx2 (the tiled edge kernel; else nullptr): this scene's slice [mask width] of the wavefront's LDS scratch for the longitudinal
// half of the box test - (max(s0 - os, os - s9, 0))^2 depends on the scene, the column and the obstacle only, so the scene's `row`
// lanes compute it once per column (lane i the obstacles i, i + row, ...) and the `row` source rows read it back: five of the
// twelve vector instructions of every box test.  DS operations of a wavefront execute in issue order: no barrier but the
// compiler's.  Same operands and operations as obstacle_box_in_reach.
template <int ROW = 0, typename MASK = unsigned long long, typename Store>
__device__ __forceinline__ void dp_edge_column(const DpDev& P, int j, int i, double ps, int nob, const double* tab,
                                               const double* t_smp, const double* my_obs_s, const double* my_obs_l,
                                               Store&& store, double* box_dx2 = nullptr) {
    constexpr int kMaskBits = (int)sizeof(MASK) * 8;
    const int row = ROW > 0 ? ROW : P.row, rr = row * row;
    const int nmask = min(nob, kMaskBits);
    const double s0 = ps + (double)j * P.sample_s;                  // ref :330 pre_node_s
    const double s9 = s0 + t_smp[kSamples - 1];
    const double F = jerk_unit_sum(t_smp, s0);                      // the column's jerk factor (see jerk_unit_sum)
    // longitudinal half of obstacle_in_reach (emp_core.h): same bounds, evaluated once per column
    MASK near_s = 0;
    for (int m = 0; m < nmask; ++m) {
        const double os = my_obs_s[m];
        if (os > s0 - 6.5 && os < s9 + 6.5) near_s |= (MASK)1 << m;
    }
    if (box_dx2) {
        for (int m = i; m < nmask; m += row) {
            const double os = my_obs_s[m];
            const double dx = fmax(fmax(s0 - os, os - s9), 0.0);
            box_dx2[m] = dx * dx;
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int k = 0; k < row; ++k) {
        const int p = k * row + i;
        const double smooth = tab[kF_BASE * rr + p] + tab[kF_JERK * rr + p] * F;
        const double l_lo = tab[kF_LLO * rr + p], l_hi = tab[kF_LHI * rr + p];
        double coll = 0.0;
        for (MASK rest = near_s; rest; rest &= rest - 1) {                 // ascending m, as the reference
            const int m = (kMaskBits == 32 ? __ffs((int)rest) : __ffsll((long long)rest)) - 1;
            const double ol = my_obs_l[m];
            if (box_dx2) {
                const double dy = fmax(fmax(l_lo - ol, ol - l_hi), 0.0);
                if (!(box_dx2[m] + dy * dy < 36.5)) continue;                   // obstacle_box_in_reach: contributes exactly 0
                coll = coll + obstacle_scan_dense(s0, t_smp, &tab[p], rr, my_obs_s[m], ol, P.w_coll);
                continue;
            }
            const double os = my_obs_s[m];
            if (!obstacle_box_in_reach(os, ol, s0, s9, l_lo, l_hi)) continue;   // contributes exactly 0
            coll = coll + obstacle_scan_dense(s0, t_smp, &tab[p], rr, os, ol, P.w_coll);
        }
        for (int m = kMaskBits; m < nob; ++m) {                           // beyond the mask: full test per edge
            const double os = my_obs_s[m], ol = my_obs_l[m];
            if (!obstacle_in_reach(os, ol, s0, s9, l_lo, l_hi)) continue;
            coll = coll + obstacle_scan_dense(s0, t_smp, &tab[p], rr, os, ol, P.w_coll);
        }
        store(k, (smooth + coll) + tab[kF_REF * rr + p]);
    }
    if (box_dx2) __builtin_amdgcn_wave_barrier();      // the next column's fill comes after every read of this one
}

// grid = (tiles, column chunks), block = 256.  Dynamic LDS: the pair table (copied from `pair_tab`),
// followed by the tile's obstacles [S][max_obs] x2 doubles and the sample offsets.
#ifndef EMP_EDGE_WAVES
#define EMP_EDGE_WAVES 5
#endif
template <bool TILED, int ROW = 0, typename MASK = unsigned long long>
// Five wavefronts per SIMD (at most 102 registers; 94 used, nothing spilled - the sample abscissae are rebuilt from
// s0 + t_n where they are needed instead of living in twenty registers): alone the kernel takes the same 158 us as with
// four, with a second batch's path-QP wavefronts on the SIMDs it gets a slot more (0.348 -> 0.340 ms per step).
// (Tried with several batches in flight: padding the allocation to 104 registers, so that four wavefronts leave room on
// a SIMD for a wavefront of the sweep or the Cartesian tail that a fifth edge wavefront cannot take.  The overlapped sweep
// got 15 % shorter, the edge kernel 3 % longer, and the step longer in every pipeline mode: not kept.)
__global__ __launch_bounds__(1024, EMP_EDGE_WAVES) void dp_edge_kernel(DpDev P, const double* __restrict__ pair_tab,
                                                      const double* __restrict__ obs_s,
                                                      const double* __restrict__ obs_l,
                                                      const int* __restrict__ n_obs,
                                                      const double* __restrict__ start,
                                                      double* __restrict__ start_cost,
                                                      double* __restrict__ edge, int cols_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __builtin_amdgcn_s_setprio(EMP_PRIO_FRONT);
    const int row = ROW > 0 ? ROW : P.row, rr = row * row;
    double* tab = lds;                                  // [kTableFields][rr], pair index = k*row + i
    double* t_obs_s = lds + kTableFields * rr;          // [S][max_obs]
    double* t_obs_l = t_obs_s + P.S * P.max_obs;
    double* t_smp = t_obs_l + P.S * P.max_obs;          // [kTableTail] sample offsets t_n, sum t_n, sum t_n^2, unit quintic
    // per wavefront: [S][mask width] longitudinal box terms of the column it works on (dp_edge_column: box_dx2)
    const int box_w = min(P.max_obs, (int)sizeof(MASK) * 8);
    double* box_all = t_smp + kTableTail;
    const int tile = blockIdx.x;
    const int tid = threadIdx.x;

    for (int x = tid; x < kTableFields * rr; x += blockDim.x) tab[x] = pair_tab[x];
    if (tid < kTableTail) t_smp[tid] = pair_tab[kTableFields * rr + tid];
    for (int x = tid; x < P.S * P.max_obs; x += blockDim.x) {
        const int s = x / P.max_obs, m = x - s * P.max_obs;
        const int b = tile * P.S + s;
        t_obs_s[x] = (b < P.B) ? obs_s[(size_t)b * P.max_obs + m] : 0.0;
        t_obs_l[x] = (b < P.B) ? obs_l[(size_t)b * P.max_obs + m] : 0.0;
    }
    __syncthreads();

    const int lanes_used = P.S * row;

    // ---- start edges (column 0): generic form, one thread per (scene, row); only chunk 0 does them
    if (blockIdx.y == 0 && start_cost != nullptr && tid < lanes_used) {
        const int s = tid / row, i = tid - s * row;
        const int b = tile * P.S + s;
        if (b < P.B) {
            const double ps = start[b * 4 + 0], pl = start[b * 4 + 1], pdl = start[b * 4 + 2],
                         pddl = start[b * 4 + 3];
            const Quintic q = quintic_shifted(pl, pdl, pddl, lattice_l(row, i, P.sample_l), P.sample_s);
            start_cost[(size_t)b * row + i] =
                segment_cost(q, ps, P.sample_s, t_obs_s + s * P.max_obs, t_obs_l + s * P.max_obs, min(max(n_obs[b], 0), P.max_obs),
                             P.w_coll, P.w0, P.w1, P.w2, P.w_ref);
        }
    }

    // ---- neighbour edges of this block's column chunk.  A thread keeps its lane (scene, row i); each
    // wavefront takes whole columns, so what depends on (scene, column) only - the sample abscissae and which
    // obstacles are within longitudinal reach - is set up once and reused for the `row` source rows.
    // Consecutive lanes store consecutive doubles of the tiled tensor.
    const int j_begin = 1 + blockIdx.y * cols_per_chunk;
    const int j_end = min(P.col, j_begin + cols_per_chunk);
    const int lane = tid & 63;
    const int s = lane / row, i = lane - s * row;
    const int b = tile * P.S + s;
    if (lane >= lanes_used || b >= P.B) return;
    const double ps = start[b * 4 + 0];
    const int nob = min(max(n_obs[b], 0), P.max_obs);      // a count beyond the row's capacity is clamped, never followed
    const double* my_obs_s = t_obs_s + s * P.max_obs;
    const double* my_obs_l = t_obs_l + s * P.max_obs;
    double* my_box = box_all + ((size_t)(tid >> 6) * P.S + s) * box_w;
    for (int j = j_begin + (tid >> 6); j < j_end; j += (int)(blockDim.x >> 6)) {
        // (the sample offsets t_n are the same for every lane: read through the kernel argument they come back in scalar
        // registers, where the LDS copy costs a vector move and an LDS read per pair of them in every scan)
        dp_edge_column<ROW, MASK>(P, j, i, ps, nob, tab, pair_tab + kTableFields * rr, my_obs_s, my_obs_l, [&](int k, double cost) {
            if (TILED) {
                edge[(((size_t)tile * (P.col - 1) + (j - 1)) * row + k) * 64 + lane] = cost;
            } else {
                edge[(size_t)b * (P.col - 1) * rr + (size_t)(j - 1) * rr + i * row + k] = cost;
            }
        }, my_box);
    }
}

// ---------------------------------------------------------------------------------------------
// edge costs, work-ring form (round 5; the default for the tiled lattices)
// ---------------------------------------------------------------------------------------------
// The lockstep kernel above walks, per (column, source row k), the obstacles within reach of the column with all lanes of
// the wavefront executing the SAME scan: a lane idles while its own box test failed (destination rows out of the
// obstacle's lateral reach) or while its scene has fewer obstacles near this column than a neighbour scene of the tile -
// 0.49 active lanes per scan on the benchmark batch (profiles/r04a_sq.csv), on the kernel that is 40 % of the GPU time.
//
// Here the dense part stays in lockstep (base cost, box tests: lane = (scene, destination row i), as above), but an edge
// with at least one obstacle in reach is not scanned in place: it is PUSHED as one entry {column j, source row k, owner
// lane, obstacle mask} into a per-wavefront LDS ring, and whenever a ring holds 64 entries the wavefront
// pops them, ONE ENTRY PER LANE, and every lane scans its own entry's obstacles in ascending order (the reference's
// order, path_planning.py:573-582) - the same operations on the same operands as dp_edge_column, so the tensor is
// bit-identical; only which lane computes an edge changes.  Two rings: edges with ONE obstacle in reach (a straight-line
// scan, every lane busy) and edges with several (a loop over the mask; lanes idle only while their entry has fewer
// obstacles than the round's maximum).  Priced on the CPU before it was built (tools/edge_ring_sim.py): 0.93 active lanes
// of a scan instead of 0.49 at 40 x 9 with 8 obstacles, 0.84 instead of 0.54 at 120 x 21 with 16.
// Edges without an obstacle in reach are stored from the dense part (coalesced, the other lanes masked); ring entries are
// stored by the lane that scanned them, 8 bytes each - the entries of a round are neighbours in (j, k, lane) order, so the
// stores of a round still fall into two or three 512-byte rows of the tiled tensor.
// Capacity: a ring never holds more than 63 left-over + 64 pushed entries = 127 < kRingSlots.
constexpr int kRingSlots = 128;

// Per wavefront, in LDS: code[2][kRingSlots] (32 bits an entry: column j << 17 | source row k << 12 | obstacle m << 6 | owner
// lane - the one-obstacle ring's entry IS its obstacle, it has no mask) and, for the several-obstacle ring only, one mask per
// slot as wide as the scene's obstacle count needs (1, 2, 4 or 8 bytes: ascending bit = ascending m).  The edge's smoothness
// term is recomputed by the lane that pops the entry.  LDS is what decides how many edge blocks sit beside the previous batch's
// path-QP wavefronts in the staged step (allocated in 1280-byte granules, 128 a CU): the 40 x 9 lattice's two-wavefront block
// is 14.4 KB = 12 granules, six of them fit beside two path-QP wavefronts (profiles/r05_edge/README.md 8).
EMP_HD constexpr int edge_ring_mask_bytes(int max_obs) { return max_obs <= 8 ? 1 : max_obs <= 16 ? 2 : max_obs <= 32 ? 4 : 8; }
EMP_HD constexpr int edge_ring_bytes(int max_obs) { return 2 * kRingSlots * 4 + kRingSlots * edge_ring_mask_bytes(max_obs); }
// the code's fields: 32 rows (5 bits of k), 64 obstacles, 64 lanes, columns below 2^15
constexpr int kRingMaxCol = 32767;

#ifndef EMP_EDGE_RING_BOUNDS
#define EMP_EDGE_RING_BOUNDS __launch_bounds__(1024, EMP_EDGE_WAVES)
#endif
template <bool TILED, int ROW = 0, typename MASK = unsigned>
__global__ EMP_EDGE_RING_BOUNDS void dp_edge_ring_kernel(DpDev P, const double* __restrict__ pair_tab,
                                                           const double* __restrict__ obs_s,
                                                           const double* __restrict__ obs_l,
                                                           const int* __restrict__ n_obs,
                                                           const double* __restrict__ start,
                                                           double* __restrict__ start_cost,
                                                           double* __restrict__ edge, int cols_per_chunk,
                                                           unsigned long long* __restrict__ clock_probe = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __builtin_amdgcn_s_setprio(EMP_PRIO_FRONT);
    // EMP_OPT_EDGE_CLOCK_PROBE (measurement; a scalar branch when off): the constant 100 MHz counter at the wavefront's first and
    // last instruction - how long a wavefront is resident and how many are resident at once, alone and inside the staged step
    const unsigned long long probe_r0 = clock_probe ? wall_clock64() : 0;
    const unsigned long long probe_c0 = clock_probe ? clock64() : 0;       // the shader clock: what the chip holds under FP64 load
    constexpr int kMaskBits = (int)sizeof(MASK) * 8;
    const int row = ROW > 0 ? ROW : P.row, rr = row * row;
    double* tab = lds;                                  // [kTableFields][rr], pair index = k*row + i
    double* t_obs_s = lds + kTableFields * rr;          // [S][max_obs]
    double* t_obs_l = t_obs_s + P.S * P.max_obs;
    double* t_ps = t_obs_l + P.S * P.max_obs;           // [S] plan_start_s of the tile's scenes (S <= 64)
    double* box_all = t_ps + ((P.S + 1) & ~1);          // per wavefront: [2][S][max_obs] lateral reach band of each obstacle in its column
    const int waves = (int)(blockDim.x >> 6);
    unsigned char* rings = reinterpret_cast<unsigned char*>(box_all + (size_t)waves * 2 * P.S * P.max_obs);
    const int mask_bytes = edge_ring_mask_bytes(P.max_obs);
    // behind the rings, per wavefront: the jerk factor F of each of its columns, [columns per wavefront][S] - computed once per
    // (scene, column) in the dense pass, read back by whichever lane pops an entry of that column
    const int cpw = (cols_per_chunk + waves - 1) / waves;
    double* f_all = reinterpret_cast<double*>(rings + (size_t)waves * edge_ring_bytes(P.max_obs));
    const double* t_smp = pair_tab + kTableFields * rr; // sample offsets through the kernel argument: scalar registers
    const int tile = blockIdx.x;
    const int tid = threadIdx.x;

    for (int x = tid; x < kTableFields * rr; x += blockDim.x) tab[x] = pair_tab[x];
    for (int x = tid; x < P.S * P.max_obs; x += blockDim.x) {
        const int sc = x / P.max_obs, m = x - sc * P.max_obs;
        const int bb = tile * P.S + sc;
        t_obs_s[x] = (bb < P.B) ? obs_s[(size_t)bb * P.max_obs + m] : 0.0;
        t_obs_l[x] = (bb < P.B) ? obs_l[(size_t)bb * P.max_obs + m] : 0.0;
    }
    if (tid < P.S) t_ps[tid] = (tile * P.S + tid < P.B) ? start[(size_t)(tile * P.S + tid) * 4 + 0] : 0.0;
    __syncthreads();

    const int lanes_used = P.S * row;
    // ---- start edges (column 0): generic form, one thread per (scene, row); only chunk 0 does them
    if (blockIdx.y == 0 && start_cost != nullptr && tid < lanes_used) {
        const int sc = tid / row, ii = tid - sc * row;
        const int bb = tile * P.S + sc;
        if (bb < P.B) {
            const double ps0 = start[bb * 4 + 0], pl = start[bb * 4 + 1], pdl = start[bb * 4 + 2], pddl = start[bb * 4 + 3];
            const Quintic q = quintic_shifted(pl, pdl, pddl, lattice_l(row, ii, P.sample_l), P.sample_s);
            start_cost[(size_t)bb * row + ii] =
                segment_cost(q, ps0, P.sample_s, t_obs_s + sc * P.max_obs, t_obs_l + sc * P.max_obs, min(max(n_obs[bb], 0), P.max_obs),
                             P.w_coll, P.w0, P.w1, P.w2, P.w_ref);
        }
    }

    const int j_begin = 1 + blockIdx.y * cols_per_chunk;
    const int j_end = min(P.col, j_begin + cols_per_chunk);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane / row, i = lane - s * row;
    const int b = tile * P.S + s;
    const bool live = (lane < lanes_used) && (b < P.B);      // dead lanes take no part in the dense pass, but pop entries
    const int sl = live ? s : 0;
    const double ps = t_ps[sl];
    const int nob = live ? min(min(max(n_obs[b], 0), P.max_obs), kMaskBits) : 0;
    const double* my_obs_s = t_obs_s + sl * P.max_obs;
    const double* my_obs_l = t_obs_l + sl * P.max_obs;
    double* my_band = box_all + ((size_t)wave * P.S + sl) * 2 * P.max_obs;   // [max_obs][2]: (lo, hi) of an obstacle side by side - one LDS
                                                                            // read for both (two arrays: two dependent round trips a test)
    unsigned* r_code = reinterpret_cast<unsigned*>(rings + (size_t)wave * edge_ring_bytes(P.max_obs));   // [2][kRingSlots]
    unsigned char* r_mask = reinterpret_cast<unsigned char*>(r_code + 2 * kRingSlots);                     // [kRingSlots] of mask_bytes
    auto put_mask = [&](int slot, MASK v) {
        if (mask_bytes == 1) r_mask[slot] = (unsigned char)v;
        else if (mask_bytes == 2) reinterpret_cast<unsigned short*>(r_mask)[slot] = (unsigned short)v;
        else if (mask_bytes == 4) reinterpret_cast<unsigned*>(r_mask)[slot] = (unsigned)v;
        else reinterpret_cast<unsigned long long*>(r_mask)[slot] = (unsigned long long)v;
    };
    auto get_mask = [&](int slot) -> MASK {
        if (mask_bytes == 1) return (MASK)r_mask[slot];
        if (mask_bytes == 2) return (MASK) reinterpret_cast<const unsigned short*>(r_mask)[slot];
        if (mask_bytes == 4) return (MASK) reinterpret_cast<const unsigned*>(r_mask)[slot];
        return (MASK) reinterpret_cast<const unsigned long long*>(r_mask)[slot];
    };
    double* f_tab = f_all + (size_t)wave * cpw * P.S;
    const float inv_waves = 1.0f / (float)waves;
    int head[2] = {0, 0}, cnt[2] = {0, 0};                   // wave-uniform ring state

    auto store_edge = [&](int j, int k, int owner, double cost) {
        if (TILED) {
            edge[(((size_t)tile * (P.col - 1) + (j - 1)) * row + k) * 64 + owner] = cost;
        } else {
            const int so = owner / row, io = owner - so * row;
            edge[(size_t)(tile * P.S + so) * (P.col - 1) * rr + (size_t)(j - 1) * rr + io * row + k] = cost;
        }
    };
    // pop min(cnt, 64) entries of ring c, one per lane, scan, store
    auto round = [&](int c) {
        const int n = min(cnt[c], 64);
        if (lane < n) {
            const int slot = (head[c] + lane) & (kRingSlots - 1);
            const unsigned code = r_code[c * kRingSlots + slot];
            const int owner = (int)(code & 63u), k = (int)((code >> 12) & 31u), j = (int)(code >> 17);
            const int so = owner / row, io = owner - so * row;
            const int p = k * row + io;
            const double s0 = t_ps[so] + (double)j * P.sample_s;            // ref :330 pre_node_s
            const int it = __float2int_rn((float)(j - j_begin - wave) * inv_waves);       // the wavefront's it-th column (exact: small multiples)
            const double smooth = tab[kF_BASE * rr + p] + tab[kF_JERK * rr + p] * f_tab[it * P.S + so];       // as the dense pass
            const double* o_s = t_obs_s + so * P.max_obs;
            const double* o_l = t_obs_l + so * P.max_obs;
            double coll = 0.0;
            if (c == 0) {                                                     // exactly one obstacle in reach
                const int m = (int)((code >> 6) & 63u);
                coll = coll + obstacle_scan_dense(s0, t_smp, &tab[p], rr, o_s[m], o_l[m], P.w_coll);
            } else {
                for (MASK rest = get_mask(slot); rest; rest &= rest - 1) {                              // ascending m, as the reference
                    const int m = (kMaskBits == 32 ? __ffs((int)rest) : __ffsll((long long)rest)) - 1;
                    coll = coll + obstacle_scan_dense(s0, t_smp, &tab[p], rr, o_s[m], o_l[m], P.w_coll);
                }
            }
            store_edge(j, k, owner, (smooth + coll) + tab[kF_REF * rr + p]);
        }
        head[c] = (head[c] + n) & (kRingSlots - 1);
        cnt[c] -= n;
        __builtin_amdgcn_wave_barrier();                   // the slots are free for the next pushes
    };

    for (int j = j_begin + wave; j < j_end; j += waves) {
        const double s0 = ps + (double)j * P.sample_s;                  // ref :330 pre_node_s
        const double s9 = s0 + t_smp[kSamples - 1];
        // The box test (obstacle_box_in_reach: dx^2 + dy^2 < 36.5 with dy = max(l_lo - ol, ol - l_hi, 0)) solved for the lateral
        // band once per (scene, column, obstacle): with r = sqrt(36.5 - dx^2) an edge's box [l_lo, l_hi] is in reach iff
        // l_hi > ol - r and l_lo < ol + r - two compares per (edge, obstacle) instead of seven operations.  Like the box test
        // itself this only PRUNES pairs that contribute exactly 0 (every sample 6.04 m or more away, against the 6 m where a
        // cost begins): the half metre of margin dwarfs the rounding of the square root, and a pair pruned by one form and not
        // the other contributes 0 either way, so the tensor does not depend on which form a kernel uses.
        // The scene's `row` lanes share the work - lane i the obstacles i, i + row, ... - and the longitudinal half of the reach
        // test falls out of it: an obstacle is in reach of the column iff its band exists (thr > 0: dx < 6.04).  Every lane of
        // the scene reads those verdicts out of ONE ballot per `row` obstacles (the scene's lanes are neighbours), where each
        // lane used to walk all the scene's obstacles itself (round 6: nob LDS reads and 4 nob instructions a column and lane;
        // its 6.5 m margin let through a few obstacles more, whose bands were empty all the same).
        MASK near_s = 0;
        const int my_first_lane = sl * row;
        for (int m0 = 0; m0 < P.max_obs && m0 < kMaskBits; m0 += row) {      // wave-uniform trip count
            const int m = m0 + i;
            bool in_reach = false;
            if (live && m < nob) {
                const double os = my_obs_s[m], ol = my_obs_l[m];
                const double dx = fmax(fmax(s0 - os, os - s9), 0.0);
                const double thr = 36.5 - dx * dx;
                const double r = sqrt(fmax(thr, 0.0));
                in_reach = thr > 0.0;
                my_band[2 * m] = in_reach ? ol - r : __builtin_inf();
                my_band[2 * m + 1] = in_reach ? ol + r : -__builtin_inf();
            }
            const unsigned long long verdicts = (__ballot(in_reach) >> my_first_lane) & ((1ull << row) - 1);   // obstacles m0 .. m0 + row - 1 of MY scene
            near_s |= (MASK)((MASK)verdicts << m0);
        }
        if (!live) near_s = 0;
        __builtin_amdgcn_wave_barrier();
        const double F = jerk_unit_sum(t_smp, s0);                      // the column's jerk factor
        if (live && i == 0) f_tab[((j - j_begin - wave) / waves) * P.S + s] = F;
        for (int k = 0; k < row; ++k) {
            const int p = k * row + i;
            MASK pass = 0;
            double smooth = 0.0;
            if (live) {
                smooth = tab[kF_BASE * rr + p] + tab[kF_JERK * rr + p] * F;
                const double l_lo = tab[kF_LLO * rr + p], l_hi = tab[kF_LHI * rr + p];
                for (MASK rest = near_s; rest; rest &= rest - 1) {
                    const int m = (kMaskBits == 32 ? __ffs((int)rest) : __ffsll((long long)rest)) - 1;
                    const double b_lo = my_band[2 * m], b_hi = my_band[2 * m + 1];
                    pass |= ((l_hi > b_lo) & (l_lo < b_hi)) ? (MASK)1 << m : (MASK)0;       // '&': no short-circuit branches
                }
                if (pass == 0) store_edge(j, k, lane, (smooth + 0.0) + tab[kF_REF * rr + p]);
            }
            const bool one = pass != 0 && (pass & (pass - 1)) == 0;
            const bool many = pass != 0 && !one;
            const unsigned long long b1 = __ballot(one), b2 = __ballot(many);
            if (b1 | b2) {
                const unsigned long long mine = one ? b1 : b2;
                const int c = one ? 0 : 1;
                const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0));
                if (pass != 0) {
                    const int slot = ((one ? head[0] + cnt[0] : head[1] + cnt[1]) + before) & (kRingSlots - 1);
                    const int m1 = (kMaskBits == 32 ? __ffs((int)pass) : __ffsll((long long)pass)) - 1;
                    r_code[c * kRingSlots + slot] = ((unsigned)j << 17) | ((unsigned)k << 12) | ((unsigned)(one ? m1 : 0) << 6) | (unsigned)lane;
                    if (!one) put_mask(slot, pass);
                }
                cnt[0] += __popcll(b1);
                cnt[1] += __popcll(b2);
                __builtin_amdgcn_wave_barrier();
                if (cnt[0] >= 64) round(0);
                if (cnt[1] >= 64) round(1);
            }
        }
        __builtin_amdgcn_wave_barrier();                   // the next column's box terms come after every read of this one
    }
    while (cnt[0] > 0) round(0);
    while (cnt[1] > 0) round(1);
    if (clock_probe && lane == 0) {
        unsigned long long* o = clock_probe + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * waves + wave) * 4;
        o[0] = probe_r0;
        o[1] = wall_clock64();
        o[2] = probe_c0;
        o[3] = clock64();
    }
}

// ---------------------------------------------------------------------------------------------
// min-plus sweep + backtrack
// ---------------------------------------------------------------------------------------------
// One wavefront per tile of S scenes; block = WPB wavefronts.  LDS: predecessor bytes [wave][col][64].
// ROW > 0: compile-time row count, register double buffer of PD columns (loads for the next group are
// in flight while the current group is reduced).  ROW == 0: generic fallback with a runtime row count.
// BT == false (the planning cycle, round 3): the backtrack is left to the densification kernel that follows - this kernel
// hands it the predecessor bytes (`pre_out` [tiles][col][64], the LDS table as it stands) and the terminal row (`term_out`
// [B]) instead of the rows.  The chain of col - 1 dependent LDS reads by one lane per scene was 1.6 us at the end of every
// wavefront of a 20-us launch that is otherwise a pure stream - on the queue that is the step's critical path; behind the
// densification kernel's own start-up on the back stage's queue it costs the step nothing.
template <int ROW, int PD, int WPB, bool NT = false, bool BT = true>
__global__ __launch_bounds__(64 * WPB) void dp_sweep_kernel(DpDev P, const double* __restrict__ start_cost,
                                                       const double* __restrict__ edge,
                                                       const int* __restrict__ n_obs,
                                                       double* __restrict__ rows_out,
                                                       double* __restrict__ min_cost_out,
                                                       int* __restrict__ status_out,
                                                       unsigned char* __restrict__ pre_out = nullptr,
                                                       int* __restrict__ term_out = nullptr,
                                                       unsigned long long* __restrict__ clock_probe = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pre_lds[];   // [WPB][64] doubles, then [WPB][col][64] bytes
    // EMP_OPT_SWEEP_CLOCK_PROBE (measurement; a scalar branch when off): the shader-clock counter and the constant 100 MHz
    // reference counter at the wavefront's first and last instruction - their ratio is the clock this wavefront ran at,
    // whatever runs beside it (a counter pass of rocprofv3 would serialise the kernels of the staged step).
    unsigned long long probe_c0 = 0, probe_r0 = 0;
    if (clock_probe) {
        probe_c0 = clock64();
        probe_r0 = wall_clock64();
    }
    auto probe_end = [&](int tile_, int lane_) {
        if (clock_probe && lane_ == 0) {
            unsigned long long* o = clock_probe + (size_t)tile_ * 4;
            o[0] = probe_c0;
            o[1] = clock64();
            o[2] = probe_r0;
            o[3] = wall_clock64();
        }
    };
    // One wavefront per tile, ~75 instructions per column between two waits for HBM: with two batches in flight it shares
    // its SIMD with the previous batch's path-QP / Cartesian wavefronts, which raise their priority to 3; at priority 0
    // every one of its short bursts queued behind them and the stream slowed from 20.5 to 22.5-25 us.
    __builtin_amdgcn_s_setprio(EMP_PRIO_SWEEP);
    const int row = ROW > 0 ? ROW : P.row;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* cost_lds = reinterpret_cast<double*>(pre_lds);
    const int tile = blockIdx.x * WPB + wave;
    if (tile >= P.tiles) return;                         // whole wavefront exits together
    unsigned char* pre = pre_lds + (size_t)WPB * 64 * sizeof(double) + (size_t)wave * P.col * 64;
    const int s = lane / row, i = lane - s * row;
    const int b = tile * P.S + s;
    const bool live = (lane < P.S * row) && (b < P.B);
    const int base = s * row;
    const bool left = i < (row >> 1);                    // ref :317 / :341 lane penalty rows
    const double INF = __builtin_inf();
    const double pen = left ? kLanePenalty : 0.0;

    double cost = INF;
    if (live) {
        cost = start_cost[(size_t)b * row + i];
        if (left) cost = cost + kLanePenalty;            // ref :318
    }
    const double* tile_edge = edge + (size_t)tile * (P.col - 1) * row * 64 + lane;

    // The previous column's cost front is exchanged through LDS: every lane stores its cost, then reads the
    // `row` costs of its own scene (same address across the scene's lanes = broadcast).  DS operations of one
    // wavefront execute in issue order, so the reads see the stores without a barrier; the reads carry
    // immediate offsets and are issued back to back (one LDS round trip per column instead of `row` cross-lane
    // permutes, each waited for).
    double* front = cost_lds + wave * 64;
    auto publish = [&]() {
        front[lane] = cost;
        __builtin_amdgcn_wave_barrier();
    };
    auto relax_one = [&](double e, int k, double& best, int& arg) {
        const double ck = front[base + k];
        double cand = ck + e;                             // ref :340
        if (left) cand = cand + kLanePenalty;             // ref :342
        if (cand < best) {                                // strict, k ascending: lowest k wins ties (:344)
            best = cand;
            arg = k;
        }
    };

    if constexpr (ROW > 0) {
        // Register ring of PD columns: column j's edges are loaded PD columns before they are reduced.  A shallow
        // ring wins (PD = 2 at 9 rows: 18 rows of 512 B in flight per wavefront): the launch is limited by reading
        // the tensor, not by load latency, and a deeper ring only lengthens the prologue and the register file.
        // Columns past the end re-read the last column (never used), which keeps the loads unconditional and the
        // code straight-line.
        const int last_col = P.col - 1;
        double ring[PD][ROW];
        auto load_col = [&](double (&dst)[ROW], int jcol) {
            if (last_col < 1) return;                    // one-column lattice (the drop-in cal_start_cost): no edges at all
            const int j = min(jcol, last_col);
            const double* src = tile_edge + (size_t)(j - 1) * ROW * 64;
#pragma unroll
            for (int k = 0; k < ROW; ++k) dst[k] = NT ? __builtin_nontemporal_load(&src[k * 64]) : src[k * 64];   // read once: nontemporal
        };
        // The predecessor of column j is extracted one column late: its 2 * ROW compare / select instructions
        // would otherwise sit, in program order, between the new cost and the next column's LDS exchange.  Issued
        // after the next column's front reads they fill that round trip instead.
        double held[ROW], held_best = INF;     // candidates of the previous column and their minimum
        int held_j = 0;                        // 0: nothing pending
#pragma unroll
        for (int k = 0; k < ROW; ++k) held[k] = INF;
        auto settle = [&]() {
            // the LOWEST k that attains the minimum: what the reference's strict '<' with k ascending keeps (:344).
            // ref :301/:304: cost starts at +inf and the predecessor at 1; NaN / +inf candidates leave it there.
            int arg = ROW - 1;
#pragma unroll
            for (int k = ROW - 2; k >= 0; --k) arg = (held[k] == held_best) ? k : arg;
            if (held_j > 0) pre[held_j * 64 + lane] = (unsigned char)((held_best < INF) ? arg : 1);
        };
        auto relax_col = [&](const double (&e)[ROW], int j) {
            publish();
            double cand[ROW];
#pragma unroll
            for (int k = 0; k < ROW; ++k) cand[k] = front[base + k];      // one LDS round trip for the whole front
            settle();                                                       // previous column's predecessor
#pragma unroll
            for (int k = 0; k < ROW; ++k) cand[k] = (cand[k] + e[k]) + pen;   // ref :340, :342; pen is 0.0 off the
                                                                            // penalty rows: x + 0.0 == x bit for bit
            // The new cost is the plain minimum: a v_min_f64 tree, the only thing the next column waits for.
            // fmin (IEEE minNum) skips NaN operands, like the reference's `<`, which never accepts a NaN candidate;
            // the last fmin against +inf covers the all-NaN case (the reference's cost then stays +inf).
            double tree[ROW];
#pragma unroll
            for (int k = 0; k < ROW; ++k) tree[k] = cand[k];
#pragma unroll
            for (int span = 1; span < ROW; span <<= 1) {
#pragma unroll
                for (int k = 0; k + span < ROW; k += 2 * span) tree[k] = __builtin_fmin(tree[k], tree[k + span]);
            }
            cost = __builtin_fmin(tree[0], INF);
#pragma unroll
            for (int k = 0; k < ROW; ++k) held[k] = cand[k];
            held_best = cost;
            held_j = j;
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) load_col(ring[d], 1 + d);
        for (int j0 = 1; j0 < P.col; j0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                if (j0 + d < P.col) relax_col(ring[d], j0 + d);
                load_col(ring[d], j0 + d + PD);
            }
        }
        settle();                                                           // the last column's predecessor
    } else {
        for (int j = 1; j < P.col; ++j) {
            double best = INF;
            int arg = 1;
            publish();
            for (int k = 0; k < row; ++k) relax_one(tile_edge[((size_t)(j - 1) * row + k) * 64], k, best, arg);
            cost = best;
            pre[j * 64 + lane] = (unsigned char)arg;
        }
    }

    // ---- terminal argmin (first minimum, ref :349) - every lane of a scene computes the same answer
    double best = INF;
    int arg = 0;
    bool first = true;
    publish();
    for (int k = 0; k < row; ++k) {
        const double ck = front[base + k];
        if (first || ck < best) {
            best = ck;
            arg = k;
            first = false;
        }
    }
    __builtin_amdgcn_wave_barrier();                      // pre[] bytes of all lanes are written
    // (The backtrack below is col - 1 dependent LDS reads by one lane per scene.  A segmented form - every lane follows
    // four quarter-length chains at once from its own row, then three hops pick the right ones and the scene's lanes share
    // the output columns - is bit-identical and was SLOWER: 21.7 against 20.6 us per launch at 4096 scenes, 40x9.  So was
    // leaving the rows as bytes in LDS and writing the tile's [S][col] block with 64 consecutive doubles per store: 21.5 us;
    // the scattered stores below are asynchronous and hide behind the chain.  A third form kept every lane's predecessor
    // bytes in registers and followed the chains with v_readlane, row indices in scalar registers, seven scenes interleaved,
    // fully unrolled: 26 us - every wave64 vector instruction of that chain costs four cycles, and there are 273 of them.
    // Without any backtrack the launch takes 19.0.)
    if constexpr (!BT) {
        if (live && i == 0) {
            const bool bypass = (n_obs != nullptr) && (n_obs[b] == 0);
            term_out[b] = arg;
            if (min_cost_out) min_cost_out[b] = bypass ? INF : best;
            status_out[b] = (!bypass && best > P.w_coll) ? 1 : 0;    // ref :351 (EMP_ST_DP_INFEASIBLE)
        }
        // the wavefront's predecessor table, 16 bytes per lane and round (column 0 is never read)
        const uint4* src = reinterpret_cast<const uint4*>(pre);
        uint4* dst = reinterpret_cast<uint4*>(pre_out + (size_t)tile * P.col * 64);
        for (int w = lane; w < P.col * 4; w += 64) dst[w] = src[w];
        probe_end(tile, lane);
        return;
    }
    if (live && i == 0) {
        const bool bypass = (n_obs != nullptr) && (n_obs[b] == 0);
        double* out = rows_out + (size_t)b * P.col;
        if (bypass) {                                     // ref :362-363 no obstacles: centre row, DP skipped
            const double centre = (double)(row + 1) / 2.0 - 1.0;
            for (int j = 0; j < P.col; ++j) out[j] = centre;
            if (min_cost_out) min_cost_out[b] = INF;
            status_out[b] = 0;
        } else {
            int idx = arg;
            out[P.col - 1] = (double)idx;
            for (int j = P.col - 1; j >= 1; --j) {        // ref :355-359
                idx = pre[j * 64 + base + idx];
                out[j - 1] = (double)idx;
            }
            if (min_cost_out) min_cost_out[b] = best;
            status_out[b] = (best > P.w_coll) ? 1 : 0;    // ref :351 (EMP_ST_DP_INFEASIBLE)
        }
    }
    probe_end(tile, lane);
}

// ---------------------------------------------------------------------------------------------
// lattices wider than 32 rows (ref path_planning.py:276-279 takes any `row`; its loops :301-346)
// ---------------------------------------------------------------------------------------------
// The tiled kernels above pack 64 / row scenes into a wavefront and keep the pair table (17 row^2 doubles) in LDS; beyond
// 32 rows neither holds (one scene no longer fits a wavefront past 64 rows, the table no longer fits the LDS past 34).
// Wide lattices take this generic pair instead - the same device functions on the same operands, so the results are
// bit-identical to what the tiled kernels would compute - with the edge tensor in the canonical layout
// [B][col-1][i][k] (k fastest): correctness for every `row` up to kMaxWideRow, not speed (the reference's own
// default is 12 rows; BASELINE's widest lattice has 21).
constexpr int kMaxWideRow = 1024;      // (round 5: predecessors of the wide sweep are 16-bit; until round 4 a byte, 256 rows.  The pair
                                       // table is 15 row^2 doubles - 126 MB at 1024 rows - and the tensor (col - 1) row^2 doubles per scene)

// grid = (B, max(col - 1, 1)), block = row rounded up to a wavefront (<= 256 threads): thread i costs the `row` edges
// from column j-1 into row i of column j; pair table, sample offsets and obstacles are read from device memory.
__global__ __launch_bounds__(256) void dp_edge_wide_kernel(DpDev P, const double* __restrict__ pair_tab,
                                                           const double* __restrict__ obs_s, const double* __restrict__ obs_l,
                                                           const int* __restrict__ n_obs, const double* __restrict__ start,
                                                           double* __restrict__ start_cost, double* __restrict__ edge) {
    const int row = P.row;
    const size_t rr = (size_t)P.row * P.row;
    const int b = blockIdx.x, j = 1 + blockIdx.y;
    const int nob = min(max(n_obs[b], 0), P.max_obs);
    const double* my_obs_s = obs_s + (size_t)b * P.max_obs;
    const double* my_obs_l = obs_l + (size_t)b * P.max_obs;
    for (int i = threadIdx.x; i < row; i += blockDim.x) {   // beyond 256 rows a thread takes several destination rows
        if (blockIdx.y == 0 && start_cost != nullptr) {      // start edges (column 0), ref :306-318
            const double ps = start[b * 4 + 0], pl = start[b * 4 + 1], pdl = start[b * 4 + 2], pddl = start[b * 4 + 3];
            const Quintic q = quintic_shifted(pl, pdl, pddl, lattice_l(row, i, P.sample_l), P.sample_s);
            start_cost[(size_t)b * row + i] = segment_cost(q, ps, P.sample_s, my_obs_s, my_obs_l, nob, P.w_coll, P.w0, P.w1, P.w2, P.w_ref);
        }
        if (j >= P.col) continue;
        const double* t_smp = pair_tab + (size_t)kTableFields * rr;
        double* out = edge + (size_t)b * (P.col - 1) * rr + (size_t)(j - 1) * rr + (size_t)i * row;
        dp_edge_column(P, j, i, start[b * 4 + 0], nob, pair_tab, t_smp, my_obs_s, my_obs_l, [&](int k, double cost) { out[k] = cost; });
    }
}

// One block per scene, thread i = destination row (rows beyond the block size are taken in strides); the cost front
// of the previous column lives in LDS (two buffers), predecessors in device memory `pre` [B][col][row] 16-bit.  The
// arithmetic of dp_sweep_kernel's generic path: cand = (front[k] + e) (+ 10000 on the penalty rows), strict '<' with k
// ascending from (+inf, predecessor 1), first-minimum terminal, bypass without obstacles (ref :301-363).
__global__ __launch_bounds__(256) void dp_sweep_wide_kernel(DpDev P, const double* __restrict__ start_cost,
                                                            const double* __restrict__ edge, const int* __restrict__ n_obs,
                                                            unsigned short* __restrict__ pre, double* __restrict__ rows_out,
                                                            double* __restrict__ min_cost_out, int* __restrict__ status_out) {
    extern __shared__ double front_lds[];                 // [2][row]
    const int row = P.row;
    const size_t rr = (size_t)P.row * P.row;
    const int b = blockIdx.x;
    const double INF = __builtin_inf();
    double* cur = front_lds;
    double* nxt = front_lds + row;
    for (int i = threadIdx.x; i < row; i += blockDim.x) {
        double c = start_cost[(size_t)b * row + i];
        if (i < (row >> 1)) c = c + kLanePenalty;         // ref :318
        cur[i] = c;
    }
    __syncthreads();
    unsigned short* my_pre = pre + (size_t)b * P.col * row;
    for (int j = 1; j < P.col; ++j) {
        const double* ej = edge + (size_t)b * (P.col - 1) * rr + (size_t)(j - 1) * rr;
        for (int i = threadIdx.x; i < row; i += blockDim.x) {
            const bool left = i < (row >> 1);
            double best = INF;
            int arg = 1;                                  // ref :304
            for (int k = 0; k < row; ++k) {
                double cand = cur[k] + ej[(size_t)i * row + k];   // ref :340
                if (left) cand = cand + kLanePenalty;             // ref :342
                if (cand < best) {                                // strict, k ascending (:344)
                    best = cand;
                    arg = k;
                }
            }
            nxt[i] = best;
            my_pre[(size_t)j * row + i] = (unsigned short)arg;
        }
        __syncthreads();
        double* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (threadIdx.x == 0) {
        const bool bypass = (n_obs != nullptr) && (n_obs[b] == 0);
        double* out = rows_out + (size_t)b * P.col;
        if (bypass) {                                     // ref :362-363 no obstacles: centre row, DP skipped
            const double centre = (double)(row + 1) / 2.0 - 1.0;
            for (int j = 0; j < P.col; ++j) out[j] = centre;
            if (min_cost_out) min_cost_out[b] = INF;
            status_out[b] = 0;
        } else {
            double best = INF;
            int arg = 0;
            bool first = true;
            for (int k = 0; k < row; ++k) {               // first minimum, ref :349
                const double ck = cur[k];
                if (first || ck < best) {
                    best = ck;
                    arg = k;
                    first = false;
                }
            }
            int idx = arg;
            out[P.col - 1] = (double)idx;
            for (int j = P.col - 1; j >= 1; --j) {        // ref :355-359
                idx = my_pre[(size_t)j * row + idx];
                out[j - 1] = (double)idx;
            }
            if (min_cost_out) min_cost_out[b] = best;
            status_out[b] = (best > P.w_coll) ? 1 : 0;    // ref :351 (EMP_ST_DP_INFEASIBLE)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused DP: edge costs staged in LDS and swept in place (EMP_DP_FUSED)
// ---------------------------------------------------------------------------------------------
// ref: DP_algorithm up to the backtrack, path_planning.py:301-361, as ONE kernel that never writes the edge tensor.
// Block = one tile of S scenes (the sweep's unit: lane = (scene, destination row)), four wavefronts.  The columns are
// taken NC at a time: all four wavefronts cost the chunk's columns into an LDS buffer [NC][row][64] (columns handed
// out by an LDS counter, so that wavefront 0, which arrives late, takes fewer), a barrier, then wavefront 0 runs
// the min-plus recurrence over those NC columns while the other three already cost the next chunk into the second
// buffer.  Same arithmetic as dp_edge_kernel + dp_sweep_kernel (dp_edge_column, the v_min tree with the
// lowest-k predecessor, first-minimum terminal), so rows, min_cost and status are bit-identical to the two-kernel
// path.  Trade: no 8 E bytes per scene written and read back, against one block per tile - 586 blocks at 4096
// scenes 40x9, two or three per CU instead of the edge kernel's sixteen wavefronts per CU (HISTORY.md section 3.2).
struct FusedLds {
    int off_smp, off_obs_s, off_obs_l, off_buf, off_front, off_pre, off_ctr, total;   // bytes
};
__host__ __device__ inline FusedLds fused_lds(int row, int col, int S, int max_obs, int nc) {
    FusedLds L;
    int o = kTableFields * row * row * 8;
    L.off_smp = o;   o += kTableTail * 8;
    L.off_obs_s = o; o += S * max_obs * 8;
    L.off_obs_l = o; o += S * max_obs * 8;
    L.off_buf = o;   o += 2 * nc * row * 64 * 8;
    L.off_front = o; o += 64 * 8;
    L.off_pre = o;   o += ((col * 64 + 7) / 8) * 8;
    L.off_ctr = o;   o += 8;
    L.total = o;
    return L;
}

template <int ROW>
__global__ __launch_bounds__(256, 4) void dp_fused_kernel(DpDev P, const double* __restrict__ pair_tab,
                                                          const double* __restrict__ obs_s,
                                                          const double* __restrict__ obs_l,
                                                          const int* __restrict__ n_obs,
                                                          const double* __restrict__ start,
                                                          double* __restrict__ rows_out,
                                                          double* __restrict__ min_cost_out,
                                                          int* __restrict__ status_out, int nc) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int row = ROW > 0 ? ROW : P.row, rr = row * row;
    const FusedLds L = fused_lds(row, P.col, P.S, P.max_obs, nc);
    unsigned char* base_b = reinterpret_cast<unsigned char*>(lds);
    double* tab = lds;
    double* t_smp = reinterpret_cast<double*>(base_b + L.off_smp);
    double* t_obs_s = reinterpret_cast<double*>(base_b + L.off_obs_s);
    double* t_obs_l = reinterpret_cast<double*>(base_b + L.off_obs_l);
    double* buf = reinterpret_cast<double*>(base_b + L.off_buf);                  // [2][nc][row][64]
    double* front = reinterpret_cast<double*>(base_b + L.off_front);              // [64] cost front of the previous column
    unsigned char* pre = base_b + L.off_pre;                                      // [col][64] predecessor rows
    int* ctr = reinterpret_cast<int*>(base_b + L.off_ctr);                        // [2] column counters, one per buffer
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int x = tid; x < kTableFields * rr; x += blockDim.x) tab[x] = pair_tab[x];
    if (tid < kTableTail) t_smp[tid] = pair_tab[kTableFields * rr + tid];
    for (int x = tid; x < P.S * P.max_obs; x += blockDim.x) {
        const int sc = x / P.max_obs, m = x - sc * P.max_obs;
        const int bb = tile * P.S + sc;
        t_obs_s[x] = (bb < P.B) ? obs_s[(size_t)bb * P.max_obs + m] : 0.0;
        t_obs_l[x] = (bb < P.B) ? obs_l[(size_t)bb * P.max_obs + m] : 0.0;
    }
    if (tid < 2) ctr[tid] = 0;
    __syncthreads();

    const int s = lane / row, i = lane - s * row;
    const int b = tile * P.S + s;
    const bool live = (lane < P.S * row) && (b < P.B);
    const int sl = live ? s : 0;
    const double ps = live ? start[b * 4 + 0] : 0.0;
    const int nob = live ? min(max(n_obs[b], 0), P.max_obs) : 0;   // a count beyond the row's capacity is clamped, never followed
    const double* my_obs_s = t_obs_s + sl * P.max_obs;
    const double* my_obs_l = t_obs_l + sl * P.max_obs;
    const int base = s * row;
    const bool left = i < (row >> 1);                    // ref :317 / :341 lane penalty rows
    const double INF = __builtin_inf();
    const double pen = left ? kLanePenalty : 0.0;

    // ---- sweep state, used by wavefront 0 only
    double cost = INF;
    constexpr int HR = ROW > 0 ? ROW : 1;
    double held[HR], held_best = INF;                    // candidates of the previous column and their minimum
    int held_j = 0;
#pragma unroll
    for (int k = 0; k < HR; ++k) held[k] = INF;
    auto publish = [&]() {
        front[lane] = cost;
        __builtin_amdgcn_wave_barrier();
    };
    auto settle = [&]() {                                // see dp_sweep_kernel: the lowest k that attains the minimum
        if constexpr (ROW > 0) {
            int arg = ROW - 1;
#pragma unroll
            for (int k = ROW - 2; k >= 0; --k) arg = (held[k] == held_best) ? k : arg;
            if (held_j > 0) pre[held_j * 64 + lane] = (unsigned char)((held_best < INF) ? arg : 1);
        }
    };
    auto relax_col = [&](const double* e /* [row][64] at this lane */, int j) {
        if constexpr (ROW > 0) {
            publish();
            double cand[ROW];
#pragma unroll
            for (int k = 0; k < ROW; ++k) cand[k] = front[base + k];
            settle();
#pragma unroll
            for (int k = 0; k < ROW; ++k) cand[k] = (cand[k] + e[k * 64]) + pen;      // ref :340, :342
            double tree[ROW];
#pragma unroll
            for (int k = 0; k < ROW; ++k) tree[k] = cand[k];
#pragma unroll
            for (int span = 1; span < ROW; span <<= 1) {
#pragma unroll
                for (int k = 0; k + span < ROW; k += 2 * span) tree[k] = __builtin_fmin(tree[k], tree[k + span]);
            }
            cost = __builtin_fmin(tree[0], INF);
#pragma unroll
            for (int k = 0; k < ROW; ++k) held[k] = cand[k];
            held_best = cost;
            held_j = j;
        } else {
            double best = INF;
            int arg = 1;                                 // ref :304 predecessor initialised to 1
            publish();
            for (int k = 0; k < row; ++k) {
                double cand = front[base + k] + e[k * 64];                           // ref :340
                if (left) cand = cand + kLanePenalty;                                // ref :342
                if (cand < best) {                                                   // strict, k ascending (:344)
                    best = cand;
                    arg = k;
                }
            }
            cost = best;
            pre[j * 64 + lane] = (unsigned char)arg;
        }
    };

    if (wave == 0 && live) {                             // ref :306-318 start column
        const Quintic q = quintic_shifted(start[b * 4 + 1], start[b * 4 + 2], start[b * 4 + 3], lattice_l(row, i, P.sample_l),
                                          P.sample_s);
        cost = segment_cost(q, ps, P.sample_s, my_obs_s, my_obs_l, nob, P.w_coll, P.w0, P.w1, P.w2, P.w_ref);
        if (left) cost = cost + kLanePenalty;
    }

    const int ncol = P.col - 1;
    const int nchunks = (ncol + nc - 1) / nc;
    for (int c = 0; c < nchunks; ++c) {
        const int par = c & 1;
        const int j0 = 1 + c * nc;
        const int cols = min(nc, P.col - j0);
        double* cbuf = buf + (size_t)par * nc * row * 64;
        if (tid == 0) ctr[par ^ 1] = 0;                  // the other buffer's counter is idle between its two barriers
        for (;;) {
            int jj = 0;
            if (lane == 0) jj = atomicAdd(&ctr[par], 1);
            jj = __builtin_amdgcn_readfirstlane(jj);
            if (jj >= cols) break;
            double* dst = cbuf + (size_t)jj * row * 64 + lane;
            if (live) dp_edge_column(P, j0 + jj, i, ps, nob, tab, t_smp, my_obs_s, my_obs_l, [&](int k, double v) { dst[k * 64] = v; });
        }
        __syncthreads();                                 // chunk c is complete in cbuf
        if (wave == 0)
            for (int jj = 0; jj < cols; ++jj) relax_col(cbuf + (size_t)jj * row * 64 + lane, j0 + jj);
    }
    if (wave != 0) return;
    settle();                                            // the last column's predecessor

    // ---- terminal argmin (first minimum, ref :349), backtrack (:355-359), outputs: as dp_sweep_kernel
    double best = INF;
    int arg = 0;
    bool first = true;
    publish();
    for (int k = 0; k < row; ++k) {
        const double ck = front[base + k];
        if (first || ck < best) {
            best = ck;
            arg = k;
            first = false;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (live && i == 0) {
        const bool bypass = n_obs[b] == 0;
        double* out = rows_out + (size_t)b * P.col;
        if (bypass) {                                     // ref :362-363 no obstacles: centre row, DP skipped
            const double centre = (double)(row + 1) / 2.0 - 1.0;
            for (int j = 0; j < P.col; ++j) out[j] = centre;
            if (min_cost_out) min_cost_out[b] = INF;
            status_out[b] = 0;
        } else {
            int idx = arg;
            out[P.col - 1] = (double)idx;
            for (int j = P.col - 1; j >= 1; --j) {
                idx = pre[j * 64 + base + idx];
                out[j - 1] = (double)idx;
            }
            if (min_cost_out) min_cost_out[b] = best;
            status_out[b] = (best > P.w_coll) ? 1 : 0;    // ref :351 (EMP_ST_DP_INFEASIBLE)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// densification (ref: path_planning.py:364-432)
// ---------------------------------------------------------------------------------------------
// One wavefront per scene and one lattice segment per lane: every segment's start state is known
// from the chosen rows alone (s0 = plan_start_s + c * sample_s, l0 = the previous node, dl0 = ddl0 = 0 after the
// first segment), so the quintics are independent and only the output offsets need a prefix sum over the
// per-segment sample counts (ref :405 / :423: len(arange(0, int(end_s - start_s), res))).
// pre / term != nullptr (the planning cycle): the sweep left the backtrack to this kernel (dp_sweep_kernel, BT == false) -
// the scene's predecessor bytes are gathered into LDS, lane 0 follows the chain (ref :355-361; no obstacles: the centre row,
// :362-363) and `rows_out` [B][col] receives the rows.  dynamic LDS then: col doubles + col * row bytes.
__global__ __launch_bounds__(64) void dp_enrich_wave_kernel(DpDev P, const double* __restrict__ rows,
                                                           const double* __restrict__ start, int max_pts,
                                                           double* __restrict__ path_s, double* __restrict__ path_l,
                                                           int* __restrict__ path_len, int* __restrict__ status,
                                                           int or_status, const unsigned char* __restrict__ pre = nullptr,
                                                           const int* __restrict__ term = nullptr,
                                                           const int* __restrict__ n_obs = nullptr,
                                                           double* __restrict__ rows_out = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double enrich_lds[];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const double ps = start[b * 4 + 0];
    double* os = path_s + (size_t)b * max_pts;
    double* ol = path_l + (size_t)b * max_pts;
    // The scene's rows are read from LDS in BOTH forms (round 6): as a run-time choice between the device array and the LDS copy
    // the reads compiled to flat loads behind a select of two address spaces - the pattern that perturbed builds of the path-QP
    // kernel read wrong (emp_tail_kernels.h, sd_at).
    double* lrows = enrich_lds;                                                       // [col]
    const double* my_rows = lrows;
    if (pre == nullptr) {
        for (int j = lane; j < P.col; j += 64) lrows[j] = rows[(size_t)b * P.col + j];
        __syncthreads();
    } else {
        unsigned char* bt = reinterpret_cast<unsigned char*>(lrows + P.col);          // [col][row]
        const int tile = b / P.S, base = (b - tile * P.S) * P.row;
        const unsigned char* tp = pre + (size_t)tile * P.col * 64 + base;
        for (int j = 1 + lane; j < P.col; j += 64)
            for (int r = 0; r < P.row; ++r) bt[j * P.row + r] = tp[j * 64 + r];
        __syncthreads();
        if (lane == 0) {
            if (n_obs != nullptr && n_obs[b] == 0) {                                   // ref :362-363
                const double centre = (double)(P.row + 1) / 2.0 - 1.0;
                for (int j = 0; j < P.col; ++j) lrows[j] = centre;
            } else {
                int idx = term[b];
                lrows[P.col - 1] = (double)idx;
                for (int j = P.col - 1; j >= 1; --j) {                                // ref :355-359
                    idx = bt[j * P.row + idx];
                    lrows[j - 1] = (double)idx;
                }
            }
        }
        __syncthreads();
        for (int j = lane; j < P.col; j += 64) rows_out[(size_t)b * P.col + j] = lrows[j];
    }
    int n_before = 0;
    bool trunc = false;
    double last_s = ps, last_l = start[b * 4 + 1];
    for (int c0 = 0; c0 < P.col; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < P.col;
        const int cc = in ? c : P.col - 1;
        const double s0 = (cc == 0) ? ps : ps + (double)cc * P.sample_s;              // previous segment's end_s (ref :369)
        const double end_s = ps + (double)(cc + 1) * P.sample_s;
        const double l0 = (cc == 0) ? start[b * 4 + 1] : lattice_l_f(P.row, my_rows[cc - 1], P.sample_l);
        const double dl0 = (cc == 0) ? start[b * 4 + 2] : 0.0, ddl0 = (cc == 0) ? start[b * 4 + 3] : 0.0;
        const double end_l = lattice_l_f(P.row, my_rows[cc], P.sample_l);            // ref :370
        const double span = end_s - s0;
        const int cnt = in ? arange_count(span, P.res) : 0;
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int off = n_before + incl - cnt;
        const Quintic q = quintic_shifted(l0, dl0, ddl0, end_l, span);
        for (int k = 0; k < cnt; ++k) {
            const double t = (double)k * P.res;
            const int n = off + k;
            if (n < max_pts) {
                os[n] = s0 + t;
                ol[n] = quintic_l(q, t);
            } else {
                trunc = true;
            }
        }
        n_before += __shfl(incl, 63, 64);
        const int owner = min(P.col - 1 - c0, 63);                                   // lane of the chunk's last real segment
        last_s = __shfl(end_s, owner, 64);
        last_l = __shfl(end_l, owner, 64);
    }
    int n = n_before;
    if (n < max_pts) {                                                               // ref :429-430
        if (lane == 0) {
            os[n] = last_s;
            ol[n] = last_l;
        }
        ++n;
    } else {
        trunc = true;
    }
    n = min(n, max_pts);
    for (int k = n + lane; k < max_pts; k += 64) {                                   // padding reads as 0
        os[k] = 0.0;
        ol[k] = 0.0;
    }
    const bool any_trunc = __any(trunc);
    if (lane == 0) {
        path_len[b] = n;
        if (or_status) {
            if (any_trunc) status[b] |= 32;                                          // EMP_ST_TRUNCATED
        } else {
            status[b] = any_trunc ? 32 : 0;
        }
    }
}

// ref: enrich_DP_s_l (path_planning.py:378-432) on caller-supplied node lists (any spacing), for the drop-in
// function of the same name: node_s, node_l [B][max_nodes], n_nodes [B].
__global__ void enrich_nodes_kernel(int B, int max_nodes, double res, const double* __restrict__ node_s,
                                    const double* __restrict__ node_l, const int* __restrict__ n_nodes,
                                    const double* __restrict__ start, int max_pts, double* __restrict__ path_s,
                                    double* __restrict__ path_l, int* __restrict__ path_len, int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s0 = start[b * 4 + 0], l0 = start[b * 4 + 1], dl0 = start[b * 4 + 2], ddl0 = start[b * 4 + 3];
    double* os = path_s + (size_t)b * max_pts;
    double* ol = path_l + (size_t)b * max_pts;
    int n = 0;
    bool trunc = false;
    double end_s = s0, end_l = l0;
    const int cols = n_nodes[b];
    for (int c = 0; c < cols; ++c) {
        end_s = node_s[(size_t)b * max_nodes + c];
        end_l = node_l[(size_t)b * max_nodes + c];
        const double span = end_s - s0;
        const int cnt = arange_count(span, res);
        const Quintic q = quintic_shifted(l0, dl0, ddl0, end_l, span);
        for (int k = 0; k < cnt; ++k) {
            const double t = (double)k * res;
            if (n < max_pts) {
                os[n] = s0 + t;
                ol[n] = quintic_l(q, t);
                ++n;
            } else {
                trunc = true;
            }
        }
        s0 = end_s;
        l0 = end_l;
        dl0 = 0.0;
        ddl0 = 0.0;
    }
    if (cols > 0) {
        if (n < max_pts) {
            os[n] = end_s;
            ol[n] = end_l;
            ++n;
        } else {
            trunc = true;
        }
    }
    path_len[b] = n;
    status[b] = trunc ? 32 : 0;
}

}  // namespace emp
