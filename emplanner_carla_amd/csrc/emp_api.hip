// emp_api.hip - C-ABI of the MI355X EM-Planner hot path (see include/emplanner.h).
// One translation unit: kernels are header-only templates, this file owns launches and staging.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "emp_context.h"
#ifndef EMP_QP_LDS_PAD
#define EMP_QP_LDS_PAD 0       // development: the same for a path-QP wavefront
#endif
#ifndef EMP_EDGE_LDS_PAD
#define EMP_EDGE_LDS_PAD 0     // development: extra (unused) dynamic LDS per edge-cost block, to measure what LDS room is worth
#endif
#include "emp_dp_kernels.h"
#include "emp_st_kernels.h"
#include "emp_st_backend_kernels.h"
#include "emp_tail_kernels.h"
#include "emp_mpc_kernels.h"

namespace emp {
thread_local std::string g_create_error;
constexpr int kMaxTiledRow = 32;       // the tiled kernels (scenes packed into wavefronts, pair table in LDS) up to here

static int make_dp_dev(emp_ctx* ctx, const emp_dp_params* p, int B, int max_obs, DpDev* d) {
    EMP_REQUIRE(ctx, p != nullptr, "dp params are NULL");
    EMP_REQUIRE(ctx, p->row >= 1 && p->row <= kMaxWideRow, "row must be in [1, 1024]");
    EMP_REQUIRE(ctx, p->col >= 1 && p->col <= 255 * 16, "col must be in [1, 4080]");
    EMP_REQUIRE(ctx, B >= 0, "negative batch");
    EMP_REQUIRE(ctx, max_obs >= 0 && max_obs <= 256, "max_obs must be in [0, 256]");
    EMP_REQUIRE(ctx, p->sample_s > 0 && p->sample_l > 0 && p->sampling_res > 0, "sample_s, sample_l, sampling_res must be > 0");
    d->row = p->row;
    d->col = p->col;
    d->S = p->row <= kMaxTiledRow ? 64 / p->row : 1;      // wider lattices: one scene per block, canonical edge tensor
    d->tiles = (B + d->S - 1) / d->S;
    d->B = B;
    d->max_obs = max_obs > 0 ? max_obs : 1;
    d->sample_s = p->sample_s;
    d->sample_l = p->sample_l;
    d->res = p->sampling_res;
    d->w_coll = p->w_collision;
    d->w0 = p->w_smooth[0];
    d->w1 = p->w_smooth[1];
    d->w2 = p->w_smooth[2];
    d->w_ref = p->w_ref;
    return EMP_OK;
}

static bool wide(const DpDev& d) { return d.row > kMaxTiledRow; }
// elements of the edge tensor the DP kernels exchange: tiled up to 32 rows, canonical [B][col-1][row][row] beyond
static size_t tiled_elems(const DpDev& d) {
    if (wide(d)) return (size_t)d.B * (size_t)(d.col - 1) * d.row * d.row;
    return (size_t)d.tiles * (size_t)(d.col - 1) * d.row * 64;
}

// ---- device-level stage launchers (all pointers are device memory; nothing synchronises) -----
// pair table of the lattice (emp_dp_kernels.h dp_pair_table_kernel): rebuilt only when the lattice parameters change,
// stream-ordered before its first use
static int dp_pair_table(emp_ctx* ctx, const DpDev& d, const double** out) {
    const double key[8] = {(double)d.row, d.sample_s, d.sample_l, d.w0, d.w1, d.w_ref, d.w2, 0.0};
    const size_t tab_bytes = ((size_t)kTableFields * d.row * d.row + kTableTail) * sizeof(double);
    emp_ctx::PairTable& pt = ctx->pair_tables[ctx->active_lane];
    emp_ctx::Buf& tb = pt.buf;
    if (tb.bytes < tab_bytes) {
        const int grc = grow_buffer(ctx, tb, tab_bytes);
        if (grc) return grc;
        pt.valid = false;
    }
    if (!pt.valid || memcmp(key, pt.key, sizeof(key)) != 0) {
        hipLaunchKernelGGL(dp_pair_table_kernel, dim3(1), dim3(256), 0, ctx->stream, d, (double*)tb.p);
        EMP_LAUNCH_CHECK(ctx);
        ++ctx->alloc_gen;                 // (a captured cycle graph does not carry this launch: EMP_OPT_CYCLE_GRAPH)
        memcpy(pt.key, key, sizeof(key));
        pt.valid = true;
    }
    *out = (const double*)tb.p;
    return EMP_OK;
}

static int dev_dp_edge(emp_ctx* ctx, const DpDev& d, const double* obs_s, const double* obs_l, const int* n_obs,
                       const double* start, double* start_cost, double* edge, bool tiled) {
    if (d.B == 0) return EMP_OK;
    if (wide(d)) {          // more than 32 rows: generic kernel, canonical tensor whatever `tiled` says (emp_dp_kernels.h)
        const double* pair_tab = nullptr;
        { const int prc = dp_pair_table(ctx, d, &pair_tab); if (prc) return prc; }
        EMP_REQUIRE(ctx, d.B <= 0x7fffffff && d.col - 1 <= 65535, "batch or lattice too large for the wide-row edge kernel's grid");
        KernelTimer t(ctx, "dp_edge");
        hipLaunchKernelGGL(dp_edge_wide_kernel, dim3(d.B, d.col > 1 ? d.col - 1 : 1), dim3(std::min(((d.row + 63) / 64) * 64, 256)), 0, ctx->stream,
                           d, pair_tab, obs_s, obs_l, n_obs, start, start_cost, edge);
        EMP_LAUNCH_CHECK(ctx);
        return EMP_OK;
    }
    // per block: pair table, the tile's obstacles, sample offsets; per wavefront: the longitudinal box terms of its column
    // ([S][mask width] doubles, emp_dp_kernels.h: box_dx2)
    // EMP_OPT_EDGE_FORM: 0 (default) the work-ring kernel (emp_dp_kernels.h dp_edge_ring_kernel: edges with obstacles in reach are
    // queued per wavefront and scanned one entry per lane), 1 the lockstep kernel of rounds 1-4 - bit-identical tensors.  The ring
    // form needs every obstacle of a row inside one 64-bit mask and the column index inside 15 bits; anything else is lockstep.
    const bool ring = ctx->opt[EMP_OPT_EDGE_FORM] == 0 && d.max_obs <= 64 && d.col <= kRingMaxCol;
    const bool m32 = d.max_obs <= 32;          // every obstacle of a row fits a 32-bit reach mask
    const size_t lds_fixed = (ring ? ((size_t)kTableFields * d.row * d.row + 2 * (size_t)d.S * d.max_obs + ((d.S + 1) & ~1)) * sizeof(double)
                                   : ((size_t)kTableFields * d.row * d.row + 2 * (size_t)d.S * d.max_obs + kTableTail) * sizeof(double)) + EMP_EDGE_LDS_PAD;
    const size_t lds_wave = ring ? 2 * (size_t)d.S * d.max_obs * sizeof(double) + (size_t)edge_ring_bytes(d.max_obs) + 4 * (size_t)d.S * sizeof(double)
                                 : (size_t)d.S * (d.max_obs <= 32 ? d.max_obs : (d.max_obs < 64 ? d.max_obs : 64)) * sizeof(double);
    size_t lds = lds_fixed + 2 * lds_wave;        // the block-size rule below prices a two-wavefront block; the launch its own
    EMP_REQUIRE(ctx, lds <= 160 * 1024, "lattice too wide for the LDS pair table");
    int ncol = d.col - 1;
    int chunks = 1;
    // Block size: as many wavefronts per block as it takes to fill a CU's twenty wavefront slots (five per SIMD) with the
    // blocks its LDS holds - and no more, because a block needs a free slot on as many SIMDs as it has wavefronts at the same
    // moment: beside the previous batch's path-QP wavefronts (staged pipeline) a two-wavefront block of the 40 x 9 lattice
    // (12 KB of LDS: 13 blocks per CU) finds room where a four-wavefront block does not (step 0.322 -> 0.289 ms), while the
    // 120 x 21 lattice's 60 KB table allows two blocks per CU, which therefore carry ten wavefronts each.
    // EMP_OPT_EDGE_BLOCK (emp_set_option) overrides it.
    const int eb_env = ctx->opt[EMP_OPT_EDGE_BLOCK];
    int wpb = 4;
    if (!ring) {
        const int blocks_per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
        wpb = (20 + blocks_per_cu - 1) / (blocks_per_cu > 0 ? blocks_per_cu : 1);
        if (wpb < 2) wpb = 2;
        if (wpb > 16) wpb = 16;
    } else {
        // the ring form's wavefronts carry ~3 KB of LDS each: the smallest block that puts sixteen wavefronts on a CU (or as many
        // as the LDS allows).  Small blocks matter in the staged step, where a block must find all its slots free at once beside
        // the previous batch's path-QP wavefronts: at 40 x 9 two-wavefront blocks give 0.241 ms per step, three 0.258, four 0.266,
        // eight 0.293 - although ALONE the kernel is fastest with four (profiles/r05_edge/README.md)
        // (LDS is allocated in 1280-byte granules, 128 of them a CU)
        auto blocks_per_cu = [](size_t l) { return (size_t)128 / ((l + 1279) / 1280); };
        int best = 0;
        for (int w = 2; w <= 16; ++w) {
            const size_t l = lds_fixed + (size_t)w * lds_wave;
            if (l > 160 * 1024) break;
            best = std::max(best, (int)std::min<size_t>(blocks_per_cu(l) * w, 20));
        }
        for (int w = 2; w <= 16; ++w) {
            const size_t l = lds_fixed + (size_t)w * lds_wave;
            if (l > 160 * 1024) break;
            if ((int)std::min<size_t>(blocks_per_cu(l) * w, 20) >= std::min(best, 16)) {
                wpb = w;
                break;
            }
        }
    }
    // (a tensor beyond the 256 MiB Infinity Cache - 32768 scenes of the 40 x 9 lattice - keeps four wavefronts per block: the
    // faster front stage otherwise leaves the sweep, which then streams from DRAM for 150 us, beside the previous batch's
    // path QP: 0.75 of the roofline against 0.61)
    if (tiled && tiled_elems(d) * sizeof(double) > ((size_t)256 << 20) && wpb < 4) wpb = 4;
    if (eb_env) wpb = eb_env / 64;
    // (the rule above prices a two-wavefront block; a wide table with wide obstacle rows - 32 rows, 254+ obstacle slots: one block
    // of sixteen wavefronts per CU - may not hold sixteen per-wavefront scratch areas: fewer wavefronts, not a refusal)
    while (wpb > 1 && lds_fixed + (size_t)wpb * lds_wave > 160 * 1024) --wpb;
    const int eb = wpb * 64;
    lds = lds_fixed + (size_t)wpb * lds_wave;
    EMP_REQUIRE(ctx, lds <= 160 * 1024, "edge-cost block too large for the LDS");
    // each of the block's wavefronts takes whole columns: two per wavefront, or one while that leaves the chip short of blocks
    // (a single scene: 29 us with one column per wavefront, 44 with two); chunk sizes multiples of the wavefront count
    if (ncol > 0) {
        int cpw = ((long long)d.tiles * ((ncol + 2 * wpb - 1) / (2 * wpb)) >= 1024) ? 2 : 1;
        // (ring form: a wavefront drains its rings once, at the end of its columns - four columns per wavefront while that
        // still leaves several thousand blocks: 120 x 21 at 4096 scenes 2.18 -> 2.01 ms)
        if (ring && (long long)d.tiles * ((ncol + 4 * wpb - 1) / (4 * wpb)) >= 4096) cpw = 4;
        chunks = (ncol + cpw * wpb - 1) / (cpw * wpb);
        if (chunks < 1) chunks = 1;
    }
    int cols_per_chunk = ncol > 0 ? (ncol + chunks - 1) / chunks : 1;
    cols_per_chunk = cols_per_chunk >= wpb ? (cols_per_chunk / wpb) * wpb : wpb;
    chunks = ncol > 0 ? (ncol + cols_per_chunk - 1) / cols_per_chunk : 1;
    if (ring) {       // the ring form's jerk-factor table: [columns per wavefront][S] per wavefront (priced above with four columns, the most the rule below gives a wavefront)
        const size_t cpw_final = (size_t)(cols_per_chunk + wpb - 1) / wpb;
        lds = lds_fixed + (size_t)wpb * (lds_wave - 4 * (size_t)d.S * sizeof(double) + cpw_final * d.S * sizeof(double));
        EMP_REQUIRE(ctx, lds <= 160 * 1024, "edge-cost block too large for the LDS");
    }
    dim3 grid(d.tiles, chunks), block(eb);
    const double* pair_tab = nullptr;
    { const int prc = dp_pair_table(ctx, d, &pair_tab); if (prc) return prc; }
    // the benchmark lattices' row counts are compiled in (emp_dp_kernels.h: dp_edge_column<ROW>), as in the sweep; any other takes
    // the generic instantiation - the same operations either way
    using M32 = unsigned int;
    using M64 = unsigned long long;
#define EMP_EDGE_PICK(K, T)                                                                                           \
    (d.row == 9 ? (m32 ? K<T, 9, M32> : K<T, 9, M64>)                                                                 \
     : d.row == 21 ? (m32 ? K<T, 21, M32> : K<T, 21, M64>)                                                            \
     : d.row == 12 ? (m32 ? K<T, 12, M32> : K<T, 12, M64>)                                                            \
     : d.row == 5 ? (m32 ? K<T, 5, M32> : K<T, 5, M64>)                                                               \
                  : (m32 ? K<T, 0, M32> : K<T, 0, M64>))
    auto kern_ring = tiled ? EMP_EDGE_PICK(dp_edge_ring_kernel, true) : EMP_EDGE_PICK(dp_edge_ring_kernel, false);
    auto kern = tiled ? EMP_EDGE_PICK(dp_edge_kernel, true) : EMP_EDGE_PICK(dp_edge_kernel, false);
#undef EMP_EDGE_PICK
    if (lds > 48 * 1024)
        EMP_HIP(ctx, hipFuncSetAttribute(ring ? (const void*)kern_ring : (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    // EMP_OPT_EDGE_AFTER_ENRICH (staged pipeline): the edge kernel starts behind the previous call's densification kernel,
    // so that the path QP that follows it on the back queue is dispatched BEFORE this kernel's sixteen-wavefront blocks
    // take the compute units (emp_plan_cycle)
    if (ctx->edge_wait) {
        EMP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->edge_wait, 0));
        ctx->edge_wait = nullptr;
    }
    if (ring && ctx->opt[EMP_OPT_EDGE_CLOCK_PROBE]) {          // measurement: two reference ticks per wavefront of this launch
        const size_t waves_total = (size_t)grid.x * grid.y * wpb;
        const int grc = grow_buffer(ctx, ctx->edge_probe, waves_total * 4 * sizeof(unsigned long long));
        if (grc) return grc;
        if (!ctx->edge_probe_done) EMP_HIP(ctx, hipEventCreateWithFlags(&ctx->edge_probe_done, hipEventDisableTiming));
        ctx->edge_probe_waves = (long)waves_total;
        {
            KernelTimer t(ctx, "dp_edge");
            hipLaunchKernelGGL(kern_ring, grid, block, lds, ctx->stream, d, pair_tab, obs_s, obs_l, n_obs, start, start_cost, edge,
                               cols_per_chunk, (unsigned long long*)ctx->edge_probe.p);
        }
        EMP_LAUNCH_CHECK(ctx);
        EMP_HIP(ctx, hipEventRecord(ctx->edge_probe_done, ctx->stream));
        return EMP_OK;
    }
    // EMP_OPT_LANE_EDGE_ORDER (lane mode): this edge kernel starts when the previous call's, on another lane, is done - the
    // lanes' FP64-bound kernels take turns instead of running two or three at a time, and the HBM-bound sweep behind each of them
    // has one of them beside it, not two (measured: include/emplanner.h).  Pure ordering: results do not depend on it.
    const int leo = ctx->opt[EMP_OPT_LANE_EDGE_ORDER];
    const bool ordered = ctx->pipe_mode >= 2 && ctx->active_lane >= 0 && (leo == 1 || (leo == 2 && d.B >= 8192));
    if (ordered && ctx->lane_edge_done) EMP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_edge_done, 0));
    KernelTimer t(ctx, "dp_edge");
    if (ring) {
        hipLaunchKernelGGL(kern_ring, grid, block, lds, ctx->stream, d, pair_tab, obs_s, obs_l, n_obs, start, start_cost, edge,
                           cols_per_chunk, (unsigned long long*)nullptr);
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, ctx->stream, d, pair_tab, obs_s, obs_l, n_obs, start, start_cost, edge,
                           cols_per_chunk);
    }
    EMP_LAUNCH_CHECK(ctx);
    if (ordered) {
        emp_ctx::Lane& ln = ctx->lanes[ctx->active_lane];
        if (!ln.ev_edge) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_edge, hipEventDisableTiming));
        ctx->lane_edge_done = ln.ev_edge;
        EMP_HIP(ctx, hipEventRecord(ctx->lane_edge_done, ctx->stream));
    }
    return EMP_OK;
}

static int dev_dp_sweep(emp_ctx* ctx, const DpDev& d, const double* start_cost, const double* edge,
                        const int* n_obs, double* rows, double* min_cost, int* status) {
    if (d.B == 0) return EMP_OK;
    if (wide(d)) {          // more than 32 rows: one block per scene, predecessors in device memory
        emp_ctx::Buf& pre = ctx->named["dp_wide_pre_" + std::to_string(ctx->active_lane)];
        const int grc = grow_buffer(ctx, pre, (size_t)d.B * d.col * d.row * sizeof(unsigned short));
        if (grc) return grc;
        KernelTimer t(ctx, "dp_sweep");
        hipLaunchKernelGGL(dp_sweep_wide_kernel, dim3(d.B), dim3(std::min(((d.row + 63) / 64) * 64, 256)), 2 * (size_t)d.row * sizeof(double),
                           ctx->stream, d, start_cost, edge, n_obs, (unsigned short*)pre.p, rows, min_cost, status);
        EMP_LAUNCH_CHECK(ctx);
        return EMP_OK;
    }
    // EMP_OPT_SWEEP_EXCLUSIVE (staged pipeline): the sweep starts once the previous call's back stage is done
    if (ctx->sweep_wait) {
        EMP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->sweep_wait, 0));
        ctx->sweep_wait = nullptr;
    }
    // EMP_OPT_SWEEP_CLOCK_PROBE: four ticks per wavefront (emp_dp_kernels.h), read by emp_sweep_clock_mhz
    unsigned long long* probe = nullptr;
    if (ctx->opt[EMP_OPT_SWEEP_CLOCK_PROBE]) {
        if (ctx->clock_probe_tiles != d.tiles) ctx->probe_launches = 0;       // another batch size: the ring starts over
        const int grc = grow_buffer(ctx, ctx->clock_probe, (size_t)emp_ctx::kProbeSlots * d.tiles * 4 * sizeof(unsigned long long));
        if (grc) return grc;
        probe = (unsigned long long*)ctx->clock_probe.p + (size_t)(ctx->probe_launches % emp_ctx::kProbeSlots) * d.tiles * 4;
        ctx->clock_probe_tiles = d.tiles;
        ctx->probe_launches++;
    }
    KernelTimer t(ctx, "dp_sweep", true);   // the roofline kernel: events stamped by the dispatch itself
#define EMP_SWEEP(R, PD, WPB) EMP_SWEEP_NT(R, PD, WPB, false)
#define EMP_SWEEP_NT(R, PD, WPB, NT)                                                                        \
    do {                                                                                                    \
        const size_t lds = (size_t)(WPB) * (d.col * 64 + 64 * sizeof(double));                              \
        EMP_REQUIRE(ctx, lds <= 160 * 1024, "too many columns for the predecessor table in LDS");           \
        const bool defer = (R) > 0 && ctx->bt_pre && ctx->bt_term;                                          \
        if (lds > 48 * 1024)                                                                                \
            EMP_HIP(ctx, hipFuncSetAttribute(defer ? (const void*)dp_sweep_kernel<R, PD, WPB, NT, false>    \
                                                   : (const void*)dp_sweep_kernel<R, PD, WPB, NT, true>,    \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipEvent_t stop_ev = t.stop ? t.stop : ctx->front_stop;                                             \
        if (ctx->capturing && defer) {      /* a stream capture records plain launches only */              \
            hipLaunchKernelGGL((dp_sweep_kernel<R, PD, WPB, NT, false>), dim3((d.tiles + (WPB) - 1) / (WPB)), dim3(64 * (WPB)), lds, \
                               ctx->stream, d, start_cost, edge, n_obs, rows, min_cost, status, ctx->bt_pre, ctx->bt_term, probe); \
            ctx->bt_deferred = true;                                                                        \
        } else if (ctx->capturing) {                                                                        \
            hipLaunchKernelGGL((dp_sweep_kernel<R, PD, WPB, NT>), dim3((d.tiles + (WPB) - 1) / (WPB)), dim3(64 * (WPB)), lds, \
                               ctx->stream, d, start_cost, edge, n_obs, rows, min_cost, status,             \
                               (unsigned char*)nullptr, (int*)nullptr, probe);                              \
        } else if (defer) {                                                                                 \
            hipExtLaunchKernelGGL((dp_sweep_kernel<R, PD, WPB, NT, false>), dim3((d.tiles + (WPB) - 1) / (WPB)), dim3(64 * (WPB)), lds, \
                                  ctx->stream, t.start, stop_ev, 0, d, start_cost, edge, n_obs, rows, min_cost, status, \
                                  ctx->bt_pre, ctx->bt_term, probe);                                        \
            ctx->bt_deferred = true;                                                                        \
        } else {                                                                                            \
            hipExtLaunchKernelGGL((dp_sweep_kernel<R, PD, WPB, NT>), dim3((d.tiles + (WPB) - 1) / (WPB)), dim3(64 * (WPB)), lds, \
                                  ctx->stream, t.start, stop_ev, 0, d, start_cost, edge, n_obs, rows, min_cost, status, \
                                  (unsigned char*)nullptr, (int*)nullptr, probe);                           \
        }                                                                                                   \
        ctx->front_attached = stop_ev;                                                                      \
    } while (0)
    // Ring depth PD (columns in flight per wavefront), measured at 4096 scenes: 2 is best for rows 5..12 (row 9:
    // 20.4 us against 23.2 at PD = 8, 21.4 at PD = 1), 3 for the 21-row lattice; nontemporal loads change nothing.
    // Load policy, measured on the 40x9 lattice (sweep alone, TB/s of algorithmic bytes; plain / nontemporal): 4096 scenes
    // (105 MB tensor) 5.45 / 4.74, 8192 (226 MB) 6.47 / 6.21, 12288 (331 MB) 5.04 / 6.45, 16384 4.47 / 6.40, 32768 (883 MB)
    // 4.52 / 6.03.  A tensor that fits the 256 MiB Infinity Cache is still there when the sweep follows the edge kernel
    // that wrote it, and plain loads hit it; a larger one streams from HBM, where plain loads also drag every line through
    // the cache hierarchy they will never hit again: nontemporal from 256 MiB on.
    const bool nt = tiled_elems(d) * sizeof(double) > ((size_t)256 << 20);
#define EMP_SWEEP_AUTO(R, PD)                      \
    do {                                           \
        if (nt) EMP_SWEEP_NT(R, PD, 1, true);      \
        else EMP_SWEEP_NT(R, PD, 1, false);        \
    } while (0)
    switch (d.row) {
        case 5: EMP_SWEEP_AUTO(5, 2); break;
        case 9:
            // (round 5: three columns in flight instead of two.  Alone the two are within noise of each other - 0.72-0.77 of the
            // peak either way; beside the previous batch's Cartesian tail, where the sweep runs since EMP_OPT_SWEEP_EXCLUSIVE
            // defaults to 0, the deeper ring holds 0.67-0.68 where the shallow one holds 0.65: six A/B pairs, tools/step_ab.sh)
            // (ring depths 4 and 8 and the other load policy were option values until round 6: HISTORY.md 3.2 has their numbers)
            if (nt) EMP_SWEEP_NT(9, 2, 1, true);         // from DRAM (32768 scenes) the shallow ring keeps 0.697 against 0.692
            else EMP_SWEEP_NT(9, 3, 1, false);
            break;
        case 12: EMP_SWEEP_AUTO(12, 2); break;
        case 21: EMP_SWEEP_AUTO(21, 3); break;
        default: EMP_SWEEP(0, 1, 1); break;
    }
#undef EMP_SWEEP_AUTO
#undef EMP_SWEEP
#undef EMP_SWEEP_NT
    EMP_LAUNCH_CHECK(ctx);
    if (probe) {
        if (!ctx->clock_probe_done) EMP_HIP(ctx, hipEventCreateWithFlags(&ctx->clock_probe_done, hipEventDisableTiming));
        EMP_HIP(ctx, hipEventRecord(ctx->clock_probe_done, ctx->stream));
    }
    return EMP_OK;
}

// A launch that signals ctx->attach_stop from its own dispatch when the caller offered one and the kernel carries no timing
// events (emp_context.h); else a plain launch.
template <typename K, typename... A>
static void launch_attaching(emp_ctx* ctx, bool timed, K kern, dim3 grid, dim3 block, size_t lds, A... args) {
    if (ctx->attach_stop && !timed) {
        hipExtLaunchKernelGGL(kern, grid, block, lds, ctx->stream, nullptr, ctx->attach_stop, 0, args...);
        ctx->stop_attached = true;
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, ctx->stream, args...);
    }
}

static int dev_dp_enrich(emp_ctx* ctx, const DpDev& d, const double* rows, const double* start, int max_pts,
                         double* path_s, double* path_l, int* path_len, int* status, int or_status,
                         const unsigned char* pre = nullptr, const int* term = nullptr, const int* n_obs = nullptr,
                         double* rows_out = nullptr) {
    if (d.B == 0) return EMP_OK;
    const size_t lds = (size_t)d.col * sizeof(double) + (pre ? (size_t)d.col * d.row : 0);      // the rows [col], + the predecessor bytes
    EMP_REQUIRE(ctx, lds <= 160 * 1024, "too many columns for the densification kernel's predecessor table in LDS");
    if (lds > 48 * 1024)
        EMP_HIP(ctx, hipFuncSetAttribute((const void*)dp_enrich_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer t(ctx, "dp_enrich");
    launch_attaching(ctx, t.stop != nullptr, dp_enrich_wave_kernel, dim3(d.B), dim3(64), lds, d, rows, start, max_pts, path_s, path_l,
                     path_len, status, or_status, pre, term, n_obs, rows_out);
    EMP_LAUNCH_CHECK(ctx);
    return EMP_OK;
}

// DP_algorithm up to the backtrack.  `edge_scratch` may be NULL: taken from the named scratch.
// The edge tensor (and the start costs behind it) of the current call.  One per stream that runs DP kernels (the main
// stream, and every lane in LANES mode): the sweep of one call may still read its tensor while the edge kernel of the
// next call, on another lane, writes its own.
static int dp_edge_tensor(emp_ctx* ctx, const DpDev& d, double** edge, double** start_cost) {
    const size_t need = (tiled_elems(d) + (size_t)d.B * d.row) * sizeof(double);
    emp_ctx::Buf& sc = ctx->named["dp_edge_tensor_" + std::to_string(ctx->active_lane)];
    const int grc = grow_buffer(ctx, sc, need);
    if (grc) return grc;
    *edge = (double*)sc.p;
    *start_cost = *edge + tiled_elems(d);
    return EMP_OK;
}

// EMP_DP_FUSED: one kernel, edge costs staged in LDS and swept in place (emp_dp_kernels.h dp_fused_kernel)
static int dev_dp_fused(emp_ctx* ctx, const DpDev& d, const double* obs_s, const double* obs_l, const int* n_obs,
                        const double* start, double* rows, double* min_cost, int* status) {
    if (d.B == 0) return EMP_OK;
    const double* pair_tab = nullptr;
    int rc = dp_pair_table(ctx, d, &pair_tab);
    if (rc) return rc;
    // columns per chunk: 4 (one per wavefront) while two buffers of them leave room for three blocks per CU, fewer on wide
    // lattices whose pair table fills the LDS
    int nc = 4;
    while (nc > 1 && fused_lds(d.row, d.col, d.S, d.max_obs, nc).total > 53 * 1024) nc /= 2;
    const size_t lds = (size_t)fused_lds(d.row, d.col, d.S, d.max_obs, nc).total;
    EMP_REQUIRE(ctx, lds <= 160 * 1024, "lattice too wide for the fused DP kernel's LDS working set");
    KernelTimer t(ctx, "dp_fused");
#define EMP_FUSED(R)                                                                                                  \
    do {                                                                                                              \
        if (lds > 48 * 1024)                                                                                          \
            EMP_HIP(ctx, hipFuncSetAttribute((const void*)dp_fused_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                             (int)lds));                                                              \
        hipLaunchKernelGGL(dp_fused_kernel<R>, dim3(d.tiles), dim3(256), lds, ctx->stream, d, pair_tab, obs_s, obs_l,   \
                           n_obs, start, rows, min_cost, status, nc);                                                 \
    } while (0)
    switch (d.row) {
        case 5: EMP_FUSED(5); break;
        case 9: EMP_FUSED(9); break;
        case 12: EMP_FUSED(12); break;
        case 21: EMP_FUSED(21); break;
        default: EMP_FUSED(0); break;
    }
#undef EMP_FUSED
    EMP_LAUNCH_CHECK(ctx);
    return EMP_OK;
}

// DP_algorithm up to the backtrack, in either form.
static int dev_dp_plan(emp_ctx* ctx, const DpDev& d, const double* obs_s, const double* obs_l, const int* n_obs,
                       const double* start, emp_dp_mode mode, double* rows, double* min_cost, int* status) {
    if (d.B == 0) return EMP_OK;
    // the single-kernel form lives on the tiled layout: lattices wider than 32 rows take the two-kernel form either way
    if (mode == EMP_DP_FUSED && !wide(d)) return dev_dp_fused(ctx, d, obs_s, obs_l, n_obs, start, rows, min_cost, status);
    double *edge, *start_cost;
    int rc = dp_edge_tensor(ctx, d, &edge, &start_cost);
    if (rc) return rc;
    if ((rc = dev_dp_edge(ctx, d, obs_s, obs_l, n_obs, start, start_cost, edge, true))) return rc;
    return dev_dp_sweep(ctx, d, start_cost, edge, n_obs, rows, min_cost, status);
}

}  // namespace emp

using namespace emp;

extern "C" {

int emp_abi_version(void) { return EMP_ABI_VERSION; }

void emp_dp_params_default(emp_dp_params* p) {
    // ref: path_planning.py:276-279
    p->row = 12;
    p->col = 6;
    p->sample_s = 15.0;
    p->sample_l = 1.5;
    p->sampling_res = 2.0;
    p->w_collision = 1e12;
    p->w_smooth[0] = 300.0;
    p->w_smooth[1] = 1000.0;
    p->w_smooth[2] = 5000.0;
    p->w_ref = 20.0;
}

void emp_qp_params_default(emp_qp_params* q) {
    // ref: path_planning.py:78-81, test_9.py:187-192
    q->ds = 2.0;
    q->w_l = 1000.0;
    q->w_dl = 10000.0;
    q->w_ddl = 3000.0;
    q->w_dddl = 150.0;
    q->w_centre = 250.0;
    q->w_end_l = q->w_end_dl = q->w_end_ddl = 40.0;
    q->host_d1 = q->host_d2 = q->host_w = 3.0;
    q->obs_length = q->obs_width = 5.0;
    q->decimate = 2;
    q->midpoint = 1;
    q->use_qp = 1;
    q->reserved = 0;
}

void emp_smooth_params_default(emp_smooth_params* s) {
    // ref: planning_utils.py:262-264
    s->w_smooth = 0.4;
    s->w_length = 0.3;
    s->w_ref = 0.3;
    s->x_thre = 0.2;
    s->y_thre = 0.2;
}

int emp_create(int device_id, emp_ctx** out) {
    if (!out) return fail(nullptr, EMP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, EMP_ERR_NO_DEVICE,
                    std::string("no HIP device visible: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (device_id < 0 || device_id >= n) return fail(nullptr, EMP_ERR_INVALID, "device_id out of range");
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, EMP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, EMP_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, EMP_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    emp_ctx* c = new emp_ctx();
    c->device = device_id;
    c->cu_count = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, EMP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = c;
    return EMP_OK;
}

void emp_destroy(emp_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)sync_all(ctx);
    for (auto& b : ctx->pool)
        if (b.p) (void)hipFree(b.p);
    for (auto& ln : ctx->lanes) {
        for (auto& b : ln.pool)
            if (b.p) (void)hipFree(b.p);
        if (ln.ev_in) (void)hipEventDestroy(ln.ev_in);
        if (ln.ev_done) (void)hipEventDestroy(ln.ev_done);
        if (ln.ev_front) (void)hipEventDestroy(ln.ev_front);
        if (ln.ev_edge) (void)hipEventDestroy(ln.ev_edge);
        if (ln.ev_tail) (void)hipEventDestroy(ln.ev_tail);
        if (ln.ev_qp) (void)hipEventDestroy(ln.ev_qp);
        if (ln.ev_enrich) (void)hipEventDestroy(ln.ev_enrich);
        if (ln.ev_host) (void)hipEventDestroy(ln.ev_host);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    if (ctx->cycle_graph) (void)hipGraphExecDestroy(ctx->cycle_graph);
    if (ctx->back_stream) (void)hipStreamDestroy(ctx->back_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    if (ctx->ev_h2d) (void)hipEventDestroy(ctx->ev_h2d);
    if (ctx->ev_host_last) (void)hipEventDestroy(ctx->ev_host_last);
    for (auto& hp : ctx->pinned) (void)hipHostFree(hp.p);
    if (ctx->clock_probe.p) (void)hipFree(ctx->clock_probe.p);
    if (ctx->clock_probe_done) (void)hipEventDestroy(ctx->clock_probe_done);
    if (ctx->edge_probe.p) (void)hipFree(ctx->edge_probe.p);
    if (ctx->edge_probe_done) (void)hipEventDestroy(ctx->edge_probe_done);
    if (ctx->sweep_marker) (void)hipEventDestroy(ctx->sweep_marker);
    for (auto& kv : ctx->pair_tables)
        if (kv.second.buf.p) (void)hipFree(kv.second.buf.p);
    for (auto& kv : ctx->named)
        if (kv.second.p) (void)hipFree(kv.second.p);
    if (ctx->arena_h_in) (void)hipHostFree(ctx->arena_h_in);
    if (ctx->arena_h_out) (void)hipHostFree(ctx->arena_h_out);
    if (ctx->arena_d_in) (void)hipFree(ctx->arena_d_in);
    if (ctx->arena_d_out) (void)hipFree(ctx->arena_d_out);
    for (auto& kv : ctx->events)
        for (auto& pr : kv.second.pairs) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* emp_last_error(const emp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int emp_synchronize(emp_ctx* ctx) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_HIP(ctx, (hipError_t)sync_all(ctx));
    return EMP_OK;
}

void* emp_stream(emp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

void* emp_result_stream(emp_ctx* ctx) {
    if (!ctx) return nullptr;
    return (void*)ctx->result_stream();
}

// one thread per record slot; consecutive threads write consecutive doubles
__global__ __launch_bounds__(256) void pack_records_kernel(int B, int col, int max_pts, int cap, const int* __restrict__ status,
                                                           const int* __restrict__ traj_len,
                                                           const int* __restrict__ path_len,
                                                           const double* __restrict__ dp_rows,
                                                           const double* __restrict__ path_s,
                                                           const double* __restrict__ path_l,
                                                           const double* __restrict__ traj, double* __restrict__ rec) {
    const int width = 3 + col + 2 * cap + 4 * (cap + 1);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * width) return;
    const int b = (int)(idx / width);
    int c = (int)(idx - (size_t)b * width);
    double v;
    if (c < 3) {
        v = (double)(c == 0 ? status[b] : c == 1 ? traj_len[b] : path_len[b]);
    } else if ((c -= 3) < col) {
        v = dp_rows[(size_t)b * col + c];
    } else if ((c -= col) < cap) {
        v = path_s[(size_t)b * max_pts + c];
    } else if ((c -= cap) < cap) {
        v = path_l[(size_t)b * max_pts + c];
    } else {
        v = traj[(size_t)b * (max_pts + 1) * 4 + (c - cap)];
    }
    rec[idx] = v;
}

int emp_pack_records(emp_ctx* ctx, int32_t B, int32_t col, int32_t max_pts, int32_t path_cap, const int32_t* status,
                     const int32_t* traj_len, const int32_t* path_len, const double* dp_rows, const double* path_s,
                     const double* path_l, const double* traj, double* rec, int on_result_stream, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && col >= 1 && max_pts >= 1 && path_cap >= 1 && path_cap <= max_pts, "bad sizes");
    EMP_REQUIRE(ctx, status && traj_len && path_len && dp_rows && path_s && path_l && traj && rec, "NULL array");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    const bool on_rs = on_result_stream && ctx->pipelined() && where == EMP_DEVICE;
    Stage st(ctx, where, on_rs);             // on the result stream the launch is ordered behind the cycle by the stream itself
    int rc;
    const int *d_st, *d_tl, *d_pl;
    const double *d_rows, *d_ps, *d_pll, *d_traj;
    double* d_rec;
    const size_t width = 3 + (size_t)col + 2 * (size_t)path_cap + 4 * ((size_t)path_cap + 1);
    if ((rc = st.in(status, (size_t)B, &d_st))) return rc;
    if ((rc = st.in(traj_len, (size_t)B, &d_tl))) return rc;
    if ((rc = st.in(path_len, (size_t)B, &d_pl))) return rc;
    if ((rc = st.in(dp_rows, (size_t)B * col, &d_rows))) return rc;
    if ((rc = st.in(path_s, (size_t)B * max_pts, &d_ps))) return rc;
    if ((rc = st.in(path_l, (size_t)B * max_pts, &d_pll))) return rc;
    if ((rc = st.in(traj, (size_t)B * (max_pts + 1) * 4, &d_traj))) return rc;
    // the kernel writes every slot of rec: no zero fill (it would be queued on ctx->stream, unordered with a launch on
    // the result stream)
    if ((rc = st.out(rec, (size_t)B * width, &d_rec, false))) return rc;
    if (B) {
        const size_t total = (size_t)B * width;
        hipStream_t target = on_rs ? ctx->result_stream() : ctx->stream, saved = ctx->stream;
        ctx->stream = target;                                  // (the timer's events belong on the stream that carries the launch)
        {
            KernelTimer t(ctx, "pack_records");
            hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, target, B, col, max_pts,
                               path_cap, d_st, d_tl, d_pl, d_rows, d_ps, d_pll, d_traj, d_rec);
        }
        ctx->stream = saved;
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

// trajectory-only record: status, traj_len, traj [cap + 1][4]
__global__ __launch_bounds__(256) void pack_trajectory_records_kernel(int B, int max_pts, int cap, const int* __restrict__ status,
                                                                      const int* __restrict__ traj_len,
                                                                      const double* __restrict__ traj,
                                                                      double* __restrict__ rec) {
    const int width = 2 + 4 * (cap + 1);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * width) return;
    const int b = (int)(idx / width);
    const int c = (int)(idx - (size_t)b * width);
    rec[idx] = c == 0 ? (double)status[b] : c == 1 ? (double)traj_len[b] : traj[(size_t)b * (max_pts + 1) * 4 + (c - 2)];
}

int emp_pack_trajectory_records(emp_ctx* ctx, int32_t B, int32_t max_pts, int32_t path_cap, const int32_t* status,
                                const int32_t* traj_len, const double* traj, double* rec, int on_result_stream,
                                emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_pts >= 1 && path_cap >= 1 && path_cap <= max_pts, "bad sizes");
    EMP_REQUIRE(ctx, status && traj_len && traj && rec, "NULL array");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    const bool on_rs = on_result_stream && ctx->pipelined() && where == EMP_DEVICE;
    Stage st(ctx, where, on_rs);
    int rc;
    const int *d_st, *d_tl;
    const double* d_traj;
    double* d_rec;
    const size_t width = 2 + 4 * ((size_t)path_cap + 1);
    if ((rc = st.in(status, (size_t)B, &d_st))) return rc;
    if ((rc = st.in(traj_len, (size_t)B, &d_tl))) return rc;
    if ((rc = st.in(traj, (size_t)B * (max_pts + 1) * 4, &d_traj))) return rc;
    if ((rc = st.out(rec, (size_t)B * width, &d_rec, false))) return rc;      // every slot is written by the kernel
    if (B) {
        const size_t total = (size_t)B * width;
        hipStream_t target = on_rs ? ctx->result_stream() : ctx->stream, saved = ctx->stream;
        ctx->stream = target;
        {
            KernelTimer t(ctx, "pack_records");
            hipLaunchKernelGGL(pack_trajectory_records_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, target, B, max_pts,
                               path_cap, d_st, d_tl, d_traj, d_rec);
        }
        ctx->stream = saved;
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_set_pipeline(emp_ctx* ctx, int mode) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, mode <= EMP_PIPELINE_MAX && mode >= EMP_PIPELINE_AUTO, "mode: EMP_PIPELINE_AUTO, 0, EMP_PIPELINE_STAGED or 2..EMP_PIPELINE_MAX lanes");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    EMP_HIP(ctx, (hipError_t)sync_all(ctx));
    if (mode == EMP_PIPELINE_AUTO) {
        // Three lanes are the fastest form - when every stream of the process has a hardware queue of its own.  The HIP runtime
        // maps all streams onto GPU_MAX_HW_QUEUES queues (default 4), read once when it initialised: three lane streams, the main
        // stream, the copy / d2h streams of the page-locked path if this context has them, and the streams the REST of the
        // process brings (EMP_OPT_FOREIGN_STREAMS: the caller's own, a gather stream, RCCL's) must fit - otherwise lanes share
        // queues, serialise and run slower than the staged form, which needs two.  (The one environment variable the library reads:
        // it is the only place the queue count can be had from.)
        const char* env = getenv("GPU_MAX_HW_QUEUES");
        int queues = env ? atoi(env) : 4;
        if (queues < 1) queues = 4;
        const int owned = 1 + (ctx->copy_stream ? 1 : 0) + (ctx->d2h_stream ? 1 : 0);
        const int foreign = ctx->opt[EMP_OPT_FOREIGN_STREAMS];
        mode = (queues >= 3 + owned + foreign) ? 3 : EMP_PIPELINE_STAGED;
        ctx->auto_queues = queues;
        ctx->auto_streams = owned + foreign;
    }
    const int m = mode;
    const int need = m == EMP_PIPELINE_STAGED ? emp_ctx::kStagedPools : m;
    // Streams the new mode does not use go away (everything was drained above): a stream holds a share of one of the process's
    // few hardware queues, and a staged pipeline set up while three lane streams of an earlier mode were still alive found its
    // front and back stage on one queue - 0.43 ms a step instead of 0.22 (bench.py's staged legs behind the three-lane headline)
    for (size_t i = 0; i < ctx->lanes.size(); ++i)
        if (ctx->lanes[i].stream && (m == EMP_PIPELINE_STAGED || (int)i >= m)) {
            EMP_HIP(ctx, hipStreamDestroy(ctx->lanes[i].stream));
            ctx->lanes[i].stream = nullptr;
        }
    if (m != EMP_PIPELINE_STAGED && ctx->back_stream) {
        EMP_HIP(ctx, hipStreamDestroy(ctx->back_stream));
        ctx->back_stream = nullptr;
    }
    if ((int)ctx->lanes.size() < need) ctx->lanes.resize(need);
    for (int i = 0; i < need; ++i) {
        emp_ctx::Lane& ln = ctx->lanes[i];
        if (m != EMP_PIPELINE_STAGED && !ln.stream) EMP_HIP(ctx, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        if (!ln.ev_in) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_in, hipEventDisableTiming));
        if (!ln.ev_front) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_front, hipEventDisableTiming));
        if (!ln.ev_tail) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_tail, hipEventDisableTiming));
        if (!ln.ev_done) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_done, hipEventDisableTiming));
        if (!ln.ev_qp) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_qp, hipEventDisableTiming));
        if (!ln.ev_enrich) EMP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_enrich, hipEventDisableTiming));
    }
    if (m == EMP_PIPELINE_STAGED && !ctx->back_stream) {
        // the back stage's kernels are short chains of dependent instructions on few wavefronts: their queue gets the
        // higher priority, so that they are dispatched (and, with s_setprio in the kernels, issued) ahead of the front
        // stage's bulk work they overlap with.  (Until round 6 an option confined this stream to a CU mask: never a gain.)
        int prio_low = 0, prio_high = 0;
        EMP_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        EMP_HIP(ctx, hipStreamCreateWithPriority(&ctx->back_stream, hipStreamNonBlocking, prio_high));
    }
    for (auto& ln : ctx->lanes) ln.done_valid = ln.qp_valid = ln.enrich_valid = false;       // everything was drained above
    ctx->lane_edge_done = nullptr;
    ctx->pipe_mode = m;
    ctx->lane = 0;
    return EMP_OK;
}

int emp_pipeline_form(emp_ctx* ctx, int32_t* hw_queues, int32_t* other_streams) {
    if (!ctx) return EMP_ERR_INVALID;
    if (hw_queues) *hw_queues = ctx->auto_queues;
    if (other_streams) *other_streams = ctx->auto_streams;
    return ctx->pipe_mode;
}

int emp_pipeline_depth(emp_ctx* ctx) {
    if (!ctx) return EMP_ERR_INVALID;
    return ctx->pipe_mode == 0 ? 1 : ctx->lanes_in_use();
}

int emp_set_fence(emp_ctx* ctx, int enabled) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    ctx->fence = enabled != 0;
    return EMP_OK;
}

int emp_set_option(emp_ctx* ctx, int32_t option, int32_t value) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, option >= 0 && option < EMP_OPT_COUNT, "unknown option");
    bool ok = false;
    switch (option) {
        case EMP_OPT_PATH_QP_FORM:
        case EMP_OPT_CARTESIAN_FORM:
        case EMP_OPT_SMOOTH_FORCE_FALLBACK:
        case EMP_OPT_EDGE_AFTER_ENRICH:
        case EMP_OPT_EDGE_FORM:
        case EMP_OPT_EDGE_CLOCK_PROBE:
        case EMP_OPT_CYCLE_GRAPH:
        case EMP_OPT_SWEEP_CLOCK_PROBE: ok = value == 0 || value == 1; break;
        case EMP_OPT_LANE_EDGE_ORDER:
        case EMP_OPT_SWEEP_EXCLUSIVE: ok = value >= 0 && value <= 2; break;
        case EMP_OPT_EDGE_BLOCK: ok = value == 0 || (value >= 64 && value <= 1024 && value % 64 == 0); break;
        case EMP_OPT_FOREIGN_STREAMS: ok = value >= 0 && value <= 64; break;
    }
    EMP_REQUIRE(ctx, ok, "option value out of range (include/emplanner.h, emp_option)");
    ctx->opt[option] = value;
    if (option == EMP_OPT_SWEEP_CLOCK_PROBE) ctx->probe_launches = 0;      // the statistics start over
    return EMP_OK;
}

int emp_get_option(emp_ctx* ctx, int32_t option, int32_t* value) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, option >= 0 && option < EMP_OPT_COUNT && value, "unknown option or NULL value");
    *value = ctx->opt[option];
    return EMP_OK;
}

double emp_sweep_clock_mhz(emp_ctx* ctx, double* mean_wave_us, double* max_wave_us) {
    if (!ctx || !ctx->clock_probe.p || !ctx->clock_probe_done || ctx->clock_probe_tiles <= 0 || ctx->probe_launches <= 0) return -1.0;
    if (hipEventSynchronize(ctx->clock_probe_done) != hipSuccess) return -1.0;
    const long slots = ctx->probe_launches < emp_ctx::kProbeSlots ? ctx->probe_launches : emp_ctx::kProbeSlots;
    const size_t waves = (size_t)slots * ctx->clock_probe_tiles;
    std::vector<unsigned long long> h(waves * 4);
    if (hipMemcpy(h.data(), ctx->clock_probe.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return -1.0;
    double ticks_c = 0.0, ticks_r = 0.0, longest = 0.0;
    for (size_t t = 0; t < waves; ++t) {
        const double dc = (double)(h[4 * t + 1] - h[4 * t + 0]), dr = (double)(h[4 * t + 3] - h[4 * t + 2]);
        ticks_c += dc;
        ticks_r += dr;
        if (dr > longest) longest = dr;
    }
    if (ticks_r <= 0.0) return -1.0;
    if (mean_wave_us) *mean_wave_us = ticks_r / (double)waves / 100.0;      // 100 MHz reference
    if (max_wave_us) *max_wave_us = longest / 100.0;
    return ticks_c / ticks_r * 100.0;
}

int emp_edge_probe(emp_ctx* ctx, double* mean_wave_us, double* span_us, double* mean_resident_waves, int32_t* waves) {
    if (!ctx || !ctx->edge_probe.p || !ctx->edge_probe_done || ctx->edge_probe_waves <= 0) return EMP_ERR_INVALID;
    if (hipEventSynchronize(ctx->edge_probe_done) != hipSuccess) return EMP_ERR_HIP;
    std::vector<unsigned long long> h((size_t)ctx->edge_probe_waves * 4);
    if (hipMemcpy(h.data(), ctx->edge_probe.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return EMP_ERR_HIP;
    unsigned long long first = ~0ull, last = 0;
    double sum = 0.0;
    for (long w = 0; w < ctx->edge_probe_waves; ++w) {
        first = std::min(first, h[4 * w]);
        last = std::max(last, h[4 * w + 1]);
        sum += (double)(h[4 * w + 1] - h[4 * w]);
    }
    const double span = (double)(last - first);
    if (mean_wave_us) *mean_wave_us = sum / (double)ctx->edge_probe_waves / 100.0;      // 100 MHz reference
    if (span_us) *span_us = span / 100.0;
    if (mean_resident_waves) *mean_resident_waves = span > 0 ? sum / span : 0.0;
    if (waves) *waves = (int32_t)ctx->edge_probe_waves;
    return EMP_OK;
}

double emp_edge_clock_mhz(emp_ctx* ctx) {
    if (!ctx || !ctx->edge_probe.p || !ctx->edge_probe_done || ctx->edge_probe_waves <= 0) return -1.0;
    if (hipEventSynchronize(ctx->edge_probe_done) != hipSuccess) return -1.0;
    std::vector<unsigned long long> h((size_t)ctx->edge_probe_waves * 4);
    if (hipMemcpy(h.data(), ctx->edge_probe.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    double ticks_c = 0.0, ticks_r = 0.0;
    for (long w = 0; w < ctx->edge_probe_waves; ++w) {
        ticks_r += (double)(h[4 * w + 1] - h[4 * w]);
        ticks_c += (double)(h[4 * w + 3] - h[4 * w + 2]);
    }
    return ticks_r > 0.0 ? ticks_c / ticks_r * 100.0 : -1.0;          // 100 MHz reference
}

int emp_sweep_probe_spans(emp_ctx* ctx, double* start_spread_us, double* first_start_to_last_end_us) {
    if (!ctx || !ctx->clock_probe.p || !ctx->clock_probe_done || ctx->clock_probe_tiles <= 0 || ctx->probe_launches <= 0) return EMP_ERR_INVALID;
    if (hipEventSynchronize(ctx->clock_probe_done) != hipSuccess) return EMP_ERR_HIP;
    const long slots = ctx->probe_launches < emp_ctx::kProbeSlots ? ctx->probe_launches : emp_ctx::kProbeSlots;
    const size_t tiles = (size_t)ctx->clock_probe_tiles;
    std::vector<unsigned long long> h((size_t)slots * tiles * 4);
    if (hipMemcpy(h.data(), ctx->clock_probe.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return EMP_ERR_HIP;
    double spread = 0.0, span = 0.0;
    for (long sl = 0; sl < slots; ++sl) {
        unsigned long long first = ~0ull, last_start = 0, last_end = 0;
        for (size_t t = 0; t < tiles; ++t) {
            const unsigned long long* o = &h[((size_t)sl * tiles + t) * 4];
            if (o[2] < first) first = o[2];
            if (o[2] > last_start) last_start = o[2];
            if (o[3] > last_end) last_end = o[3];
        }
        spread += (double)(last_start - first);
        span += (double)(last_end - first);
    }
    if (start_spread_us) *start_spread_us = spread / (double)slots / 100.0;
    if (first_start_to_last_end_us) *first_start_to_last_end_us = span / (double)slots / 100.0;
    return EMP_OK;
}

int emp_device_alloc(emp_ctx* ctx, uint64_t bytes, void** out) {
    EMP_REQUIRE(ctx, ctx && out, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    EMP_HIP(ctx, hipMalloc(out, bytes ? bytes : 8));
    return EMP_OK;
}
int emp_device_free(emp_ctx* ctx, void* ptr) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_HIP(ctx, (hipError_t)sync_all(ctx));
    EMP_HIP(ctx, hipFree(ptr));
    return EMP_OK;
}
int emp_copy_to_device(emp_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    EMP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EMP_OK;
}
int emp_copy_to_host(emp_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    EMP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EMP_OK;
}

int emp_set_timing(emp_ctx* ctx, int enabled) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    ctx->timing = enabled != 0;
    if (ctx->timing)
        for (auto& kv : ctx->events) kv.second.used = 0;   // restart the statistics
    return EMP_OK;
}

int emp_set_timing_filter(emp_ctx* ctx, const char* kernel) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    ctx->timing_filter = kernel ? kernel : "";
    return EMP_OK;
}

int emp_kernel_launches(emp_ctx* ctx, const char* kernel) {
    if (!ctx || !kernel) return 0;
    auto it = ctx->events.find(kernel);
    return it == ctx->events.end() ? 0 : (int)it->second.used;
}

double emp_kernel_ms(emp_ctx* ctx, const char* kernel) {
    if (!ctx || !kernel) return -1.0;
    auto it = ctx->events.find(kernel);
    if (it == ctx->events.end() || it->second.used == 0) return -1.0;
    double total = 0.0;
    for (size_t i = 0; i < it->second.used; ++i) {
        auto& pr = it->second.pairs[i];
        if (hipEventSynchronize(pr.second) != hipSuccess) return -1.0;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) != hipSuccess) return -1.0;
        total += (double)ms;
    }
    return total / (double)it->second.used;
}

int emp_kernel_samples(emp_ctx* ctx, const char* kernel, double* ms, int32_t cap) {
    if (!ctx || !kernel || (cap > 0 && !ms) || cap < 0) return -1;
    auto it = ctx->events.find(kernel);
    if (it == ctx->events.end()) return 0;
    const size_t n = it->second.used;
    for (size_t i = 0; i < n && i < (size_t)cap; ++i) {
        auto& pr = it->second.pairs[i];
        float v = 0.f;
        if (hipEventSynchronize(pr.second) != hipSuccess || hipEventElapsedTime(&v, pr.first, pr.second) != hipSuccess) return -1;
        ms[i] = (double)v;
    }
    return (int)n;
}

// ---- DP ----------------------------------------------------------------------------------
uint64_t emp_edge_tensor_elems(const emp_dp_params* p, int32_t B, emp_edge_layout layout) {
    if (!p || p->row < 1 || p->row > emp::kMaxWideRow || p->col < 1 || B < 0) return 0;
    if (layout == EMP_EDGE_CANONICAL || p->row > emp::kMaxTiledRow) return (uint64_t)B * (p->col - 1) * p->row * p->row;
    const int S = 64 / p->row;
    const uint64_t tiles = ((uint64_t)B + S - 1) / S;
    return tiles * (uint64_t)(p->col - 1) * p->row * 64;
}

int emp_dp_edge_costs(emp_ctx* ctx, const emp_dp_params* p, int32_t B, int32_t max_obs, const double* obs_s,
                      const double* obs_l, const int32_t* n_obs, const double* start, double* start_cost,
                      double* edge, emp_edge_layout layout, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    DpDev d;
    int rc = make_dp_dev(ctx, p, B, max_obs, &d);
    if (rc) return rc;
    EMP_REQUIRE(ctx, n_obs && start && edge, "n_obs, start and edge are required");
    EMP_REQUIRE(ctx, max_obs == 0 || (obs_s && obs_l), "obs_s / obs_l are required when max_obs > 0");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    const double *d_os, *d_ol, *d_start;
    const int* d_n;
    double *d_c0, *d_e;
    if ((rc = st.in(obs_s, (size_t)B * max_obs, &d_os))) return rc;
    if ((rc = st.in(obs_l, (size_t)B * max_obs, &d_ol))) return rc;
    if ((rc = st.in(n_obs, (size_t)B, &d_n))) return rc;
    if ((rc = st.in(start, (size_t)B * 4, &d_start))) return rc;
    if ((rc = st.out(start_cost, (size_t)B * d.row, &d_c0))) return rc;
    if ((rc = st.out(edge, (size_t)emp_edge_tensor_elems(p, B, layout), &d_e, layout == EMP_EDGE_TILED))) return rc;  // canonical: fully written
    if (max_obs == 0) {  // kernels index [b * max_obs + m] only for m < n_obs == 0
        double* dummy;
        if ((rc = st.tmp(1, &dummy))) return rc;
        d_os = d_ol = dummy;
    }
    if ((rc = dev_dp_edge(ctx, d, d_os, d_ol, d_n, d_start, d_c0, d_e, layout == EMP_EDGE_TILED))) return rc;
    return st.finish();
}

int emp_dp_sweep(emp_ctx* ctx, const emp_dp_params* p, int32_t B, const double* start_cost, const double* edge,
                 double* rows, double* min_cost, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    DpDev d;
    int rc = make_dp_dev(ctx, p, B, 0, &d);
    if (rc) return rc;
    EMP_REQUIRE(ctx, start_cost && edge && rows && status, "start_cost, edge, rows and status are required");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    const double *d_c0, *d_e;
    double *d_rows, *d_min;
    int* d_st;
    if ((rc = st.in(start_cost, (size_t)B * d.row, &d_c0))) return rc;
    if ((rc = st.in(edge, tiled_elems(d), &d_e))) return rc;
    if ((rc = st.out(rows, (size_t)B * d.col, &d_rows))) return rc;
    if ((rc = st.out(min_cost, (size_t)B, &d_min))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if ((rc = dev_dp_sweep(ctx, d, d_c0, d_e, nullptr, d_rows, d_min, d_st))) return rc;
    return st.finish();
}

int emp_dp_plan(emp_ctx* ctx, const emp_dp_params* p, int32_t B, int32_t max_obs, const double* obs_s,
                const double* obs_l, const int32_t* n_obs, const double* start, emp_dp_mode mode, double* rows,
                double* min_cost, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    DpDev d;
    int rc = make_dp_dev(ctx, p, B, max_obs, &d);
    if (rc) return rc;
    EMP_REQUIRE(ctx, n_obs && start && rows && status, "n_obs, start, rows and status are required");
    EMP_REQUIRE(ctx, max_obs == 0 || (obs_s && obs_l), "obs_s / obs_l are required when max_obs > 0");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    const double *d_os, *d_ol, *d_start;
    const int* d_n;
    double *d_rows, *d_min;
    int* d_st;
    if ((rc = st.in(obs_s, (size_t)B * max_obs, &d_os))) return rc;
    if ((rc = st.in(obs_l, (size_t)B * max_obs, &d_ol))) return rc;
    if ((rc = st.in(n_obs, (size_t)B, &d_n))) return rc;
    if ((rc = st.in(start, (size_t)B * 4, &d_start))) return rc;
    if ((rc = st.out(rows, (size_t)B * d.col, &d_rows))) return rc;
    if ((rc = st.out(min_cost, (size_t)B, &d_min))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (max_obs == 0) {
        double* dummy;
        if ((rc = st.tmp(1, &dummy))) return rc;
        d_os = d_ol = dummy;
    }
    if ((rc = dev_dp_plan(ctx, d, d_os, d_ol, d_n, d_start, mode, d_rows, d_min, d_st))) return rc;
    return st.finish();
}

int emp_dp_enrich(emp_ctx* ctx, const emp_dp_params* p, int32_t B, const double* rows, const double* start,
                  int32_t max_pts, double* path_s, double* path_l, int32_t* path_len, int32_t* status,
                  emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    DpDev d;
    int rc = make_dp_dev(ctx, p, B, 0, &d);
    if (rc) return rc;
    EMP_REQUIRE(ctx, rows && start && path_s && path_l && path_len && status, "NULL argument");
    EMP_REQUIRE(ctx, max_pts >= 1, "max_pts must be >= 1");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    const double *d_rows, *d_start;
    double *d_ps, *d_pl;
    int *d_len, *d_st;
    if ((rc = st.in(rows, (size_t)B * d.col, &d_rows))) return rc;
    if ((rc = st.in(start, (size_t)B * 4, &d_start))) return rc;
    if ((rc = st.out(path_s, (size_t)B * max_pts, &d_ps))) return rc;
    if ((rc = st.out(path_l, (size_t)B * max_pts, &d_pl))) return rc;
    if ((rc = st.out(path_len, (size_t)B, &d_len))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if ((rc = dev_dp_enrich(ctx, d, d_rows, d_start, max_pts, d_ps, d_pl, d_len, d_st, 0))) return rc;
    return st.finish();
}

}  // extern "C"


// =============================================================================================
// Frenet / QP stages and the whole cycle
// =============================================================================================
namespace emp {

// emp_qp_params.reserved: 0 in every build that ships.  A development build (-DEMP_DEV_HOOKS) reads it as the QP kernels'
// debug stage (cut the solve short / cap its iterations for timing experiments, tools/qp_sensitivity_probe.py); a product
// build refuses it, so that an uninitialised struct can never mask a failed solve.
static bool qp_reserved_ok(const emp_qp_params* q) { return EMP_DEV_HOOKS || q->reserved == 0; }

static QpDev make_qp_dev(const emp_qp_params* q) {
    QpDev d;
    d.qp = PathQpParams{q->ds, q->w_l, q->w_ddl, q->w_dddl, q->w_centre, q->host_d1, q->host_d2, q->host_w};
    d.obs_length = q->obs_length;
    d.obs_width = q->obs_width;
    d.decimate = q->decimate > 0 ? q->decimate : 1;
    d.midpoint = q->midpoint;
    d.use_qp = q->use_qp;
    d.debug_stage = EMP_DEV_HOOKS ? q->reserved : 0;
    return d;
}

static inline dim3 grid1(int n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

static int dev_project(emp_ctx* ctx, int B, int max_ref, int max_obs, const double* ref_line, const int* n_ref,
                       const double* origin_xy, const double* start_xy, const double* start_v, const double* start_a,
                       const double* obs_xy, const int* n_obs, double* s_map, double* obs_s, double* obs_l,
                       double* begin_sl, double* start, int obs_cap = -1, const double* dyn = nullptr,
                       int* n_obs_out = nullptr) {
    if (B == 0) return EMP_OK;
    const size_t lds = (size_t)7 * max_ref * sizeof(double);
    EMP_REQUIRE(ctx, lds <= 160 * 1024, "reference line too long for the LDS-resident projection kernel");
    if (lds > 48 * 1024)
        EMP_HIP(ctx, hipFuncSetAttribute((const void*)frenet_project_wave_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer t(ctx, "project");
    hipLaunchKernelGGL(frenet_project_wave_kernel, dim3(B), dim3(64), lds, ctx->stream, B, max_ref, max_obs, ref_line,
                       n_ref, origin_xy, start_xy, start_v, start_a, obs_xy, n_obs, s_map, obs_s, obs_l, begin_sl, start,
                       obs_cap < 0 ? max_obs : obs_cap, dyn, n_obs_out);
    EMP_LAUNCH_CHECK(ctx);
    return EMP_OK;
}

template <typename K>
static int set_lds(emp_ctx* ctx, K kernel, size_t bytes) {
    EMP_REQUIRE(ctx, bytes <= 160 * 1024, "problem too large for the LDS-resident QP solver");
    if (bytes > 48 * 1024)
        EMP_HIP(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return EMP_OK;
}

static int dev_cycle_qp(emp_ctx* ctx, int B, int max_pts, int max_obs, const QpDev& Q, const double* dp_s,
                        const double* dp_l, const int* dp_len, const double* obs_s, const double* obs_l,
                        const int* n_obs, const double* start, double* path_s, double* path_l, int* path_len,
                        int* status) {
    if (B == 0) return EMP_OK;
    const int cap = (max_pts + Q.decimate - 1) / Q.decimate;          // most stations a scene can have
    const size_t per_group = ((size_t)5 * cap + 4 * (size_t)max_obs + path_qp_words(cap)) * sizeof(double);
    int rc;
    KernelTimer t(ctx, "path_qp");
    // Eight scenes per wavefront (emp_qp_rows.h) up to 34 stations, four up to 66.  EMP_OPT_PATH_QP_FORM = 1 (emp_set_option)
    // selects the kernels of rounds 1-2 (two scenes per wavefront up to 34 stations, one beyond): an interior-point iteration
    // of that kernel is 2500 instructions against 3000, for the slower of two scenes instead of eight - the form for a caller
    // who plans a handful of scenes and counts microseconds.  The two associate their sums differently (~2e-9) and only the
    // rows solver restores its last acceptable iterate on a fallback exit, so the choice is the CALLER's and never follows
    // the batch size (round 3 switched below 1024 scenes: a 512-scene shard then differed from its slice of a 4096-scene call).
    const bool pair_form = ctx->opt[EMP_OPT_PATH_QP_FORM] == 1;
    if (cap <= 66 && !pair_form) {                                    // 8 (4) scenes per wavefront on groups of 8 (16) lanes
        const int gp = cap <= 34 ? 8 : 16;
        const size_t words = cap <= 26 ? cycle_qp_group_words<8, 3>(cap, max_obs) : cap <= 34 ? cycle_qp_group_words<8, 4>(cap, max_obs)
                                                                                                : cycle_qp_group_words<16, 4>(cap, max_obs);
        const size_t per_wave = (size_t)(64 / gp) * words * sizeof(double) + EMP_QP_LDS_PAD;
        auto kern = cap <= 26 ? cycle_qp_rows_kernel<8, 3> : cap <= 34 ? cycle_qp_rows_kernel<8, 4> : cycle_qp_rows_kernel<16, 4>;
        if ((rc = set_lds(ctx, kern, per_wave))) return rc;
        const int spw = 64 / gp;
        launch_attaching(ctx, t.stop != nullptr, kern, dim3((B + spw - 1) / spw), dim3(64), per_wave, B, max_pts,
                         max_obs, cap, Q, dp_s, dp_l, dp_len, obs_s, obs_l, n_obs, start, path_s, path_l, path_len, status);
    } else if (cap <= 34) {                                           // N, ns <= 32: two scenes per wavefront
        const size_t per_pair = 2 * ((size_t)5 * cap + 4 * (size_t)max_obs + path_qp_words_pair()) * sizeof(double);
        auto kern = cycle_qp_wave_kernel<32>;
        if ((rc = set_lds(ctx, kern, per_pair))) return rc;
        launch_attaching(ctx, t.stop != nullptr, kern, dim3((B + 1) / 2), dim3(64), per_pair, B, max_pts,
                         max_obs, cap, Q, dp_s, dp_l, dp_len, obs_s, obs_l, n_obs, start, path_s, path_l, path_len, status);
    } else {
        if ((rc = set_lds(ctx, cycle_qp_wave_kernel<64>, per_group))) return rc;
        launch_attaching(ctx, t.stop != nullptr, cycle_qp_wave_kernel<64>, dim3(B), dim3(64), per_group, B, max_pts, max_obs, cap, Q,
                         dp_s, dp_l, dp_len, obs_s, obs_l, n_obs, start, path_s, path_l, path_len, status);
    }
    EMP_LAUNCH_CHECK(ctx);
    return EMP_OK;
}

static int dev_cycle_cartesian(emp_ctx* ctx, int B, int max_ref, int max_pts, int path_cap, const emp_smooth_params* sp,
                               const double* ref_line, const double* s_map, const int* n_ref, const double* begin_sl,
                               const double* path_s, const double* path_l, const int* path_len, double* traj,
                               int* traj_len, int* status) {
    if (B == 0) return EMP_OK;
    const SmoothQpParams sx{sp->w_smooth, sp->w_length, sp->w_ref, sp->x_thre};
    const SmoothQpParams sy{sp->w_smooth, sp->w_length, sp->w_ref, sp->y_thre};
    const int cap = path_cap + 1;                                        // trajectory = planning start + path points
    const size_t lds = ((size_t)max_ref + 3 * (size_t)cap + 2 * (size_t)BoxRangeQp::words(cap, cap)) * sizeof(double);
    // Four scenes per wavefront up to 32 trajectory points, two up to 64 (emp_tail_kernels.h, cycle_cartesian_rows_kernel);
    // EMP_OPT_CARTESIAN_FORM = 1 (A/B runs) keeps the one-scene-per-wavefront kernels of rounds 1-2 (bit-identical results).
    const bool wave_form = ctx->opt[EMP_OPT_CARTESIAN_FORM] == 1;
    if (cap <= 64 && !wave_form) {
        const int spw = cap <= 32 ? 4 : 2;
        const size_t lds4 = ((size_t)spw * ((size_t)max_ref + 5 * (size_t)cap) + 2 * (size_t)BoxRangeQp::words(cap, cap)) * sizeof(double);
        auto k4 = cap <= 24 ? cycle_cartesian_rows_kernel<8, 3> : cap <= 32 ? cycle_cartesian_rows_kernel<8, 4> : cycle_cartesian_rows_kernel<16, 4>;
        int rc4 = set_lds(ctx, k4, lds4);
        if (rc4) return rc4;
        KernelTimer t4(ctx, "to_cartesian");
        const int force_fb = ctx->opt[EMP_OPT_SMOOTH_FORCE_FALLBACK] ? 1 : 0;                        // test hook
        hipLaunchKernelGGL(k4, dim3((B + spw - 1) / spw), dim3(64), lds4, ctx->stream, B, max_ref, max_pts, cap, sx, sy, ref_line, s_map,
                           n_ref, begin_sl, path_s, path_l, path_len, traj, traj_len, status, force_fb);
        EMP_LAUNCH_CHECK(ctx);
        return EMP_OK;
    }
    auto kern = cap > 32 ? cycle_cartesian_wave_kernel_wide : cycle_cartesian_wave_kernel_narrow;
    int rc = set_lds(ctx, kern, lds);
    if (rc) return rc;
    KernelTimer t(ctx, "to_cartesian");
    hipLaunchKernelGGL(kern, dim3(B), dim3(64), lds, ctx->stream, B, max_ref, max_pts, cap, sx, sy, ref_line, s_map, n_ref,
                       begin_sl, path_s, path_l, path_len, traj, traj_len, status);
    EMP_LAUNCH_CHECK(ctx);
    return EMP_OK;
}

}  // namespace emp

extern "C" {

int emp_frenet_project(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_obs, const double* ref_line,
                       const int32_t* n_ref, const double* origin_xy, const double* start_xy, const double* start_v,
                       const double* start_a, const double* obs_xy, const int32_t* n_obs, double* s_map, double* obs_s,
                       double* obs_l, double* begin_sl, double* start, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 2 && max_obs >= 0, "bad sizes");
    EMP_REQUIRE(ctx, ref_line && n_ref && origin_xy && start_xy && start_v && start_a && s_map && start, "NULL argument");
    EMP_REQUIRE(ctx, max_obs == 0 || (obs_xy && n_obs && obs_s && obs_l), "obstacle arrays required when max_obs > 0");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_o, *d_sxy, *d_v, *d_a, *d_oxy;
    const int *d_nr, *d_no;
    double *d_sm, *d_os, *d_ol, *d_bsl, *d_start;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(origin_xy, (size_t)B * 2, &d_o))) return rc;
    if ((rc = st.in(start_xy, (size_t)B * 2, &d_sxy))) return rc;
    if ((rc = st.in(start_v, (size_t)B * 2, &d_v))) return rc;
    if ((rc = st.in(start_a, (size_t)B * 2, &d_a))) return rc;
    if ((rc = st.in(obs_xy, (size_t)B * max_obs * 2, &d_oxy))) return rc;
    if ((rc = st.in(max_obs ? n_obs : nullptr, (size_t)B, &d_no))) return rc;
    if ((rc = st.out(s_map, (size_t)B * max_ref, &d_sm))) return rc;
    if ((rc = st.out(obs_s, (size_t)B * max_obs, &d_os))) return rc;
    if ((rc = st.out(obs_l, (size_t)B * max_obs, &d_ol))) return rc;
    if ((rc = st.out(begin_sl, (size_t)B * 2, &d_bsl))) return rc;
    if ((rc = st.out(start, (size_t)B * 4, &d_start))) return rc;
    if ((rc = dev_project(ctx, B, max_ref, max_obs, d_ref, d_nr, d_o, d_sxy, d_v, d_a, d_oxy, d_no, d_sm, d_os, d_ol,
                          d_bsl, d_start)))
        return rc;
    return st.finish();
}

static int match_common(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                        const int32_t* n_ref, const double* xy, const int32_t* n_pts, const int32_t* is_first_run,
                        const int32_t* pre_match_index, int32_t* match_index, double* proj, emp_mem where, int windowed) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 1 && max_pts >= 1, "bad sizes");
    EMP_REQUIRE(ctx, ref_line && n_ref && xy && n_pts && match_index && proj, "NULL argument");
    EMP_REQUIRE(ctx, !windowed || (is_first_run && pre_match_index), "is_first_run / pre_match_index required");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_xy;
    const int *d_nr, *d_np, *d_first, *d_pre;
    int* d_mi;
    double* d_pr;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(xy, (size_t)B * max_pts * 2, &d_xy))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(is_first_run, (size_t)B, &d_first))) return rc;
    if ((rc = st.in(pre_match_index, (size_t)B, &d_pre))) return rc;
    if ((rc = st.out(match_index, (size_t)B * max_pts, &d_mi))) return rc;
    if ((rc = st.out(proj, (size_t)B * max_pts * 4, &d_pr))) return rc;
    if (B) {
        KernelTimer t(ctx, "match");
        hipLaunchKernelGGL(match_points_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_ref, max_pts, d_ref, d_nr,
                           d_xy, d_np, d_first, d_pre, d_mi, d_pr, windowed);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_match_projection(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                         const int32_t* n_ref, const double* xy, const int32_t* n_pts, int32_t* match_index,
                         double* proj, emp_mem where) {
    return match_common(ctx, B, max_ref, max_pts, ref_line, n_ref, xy, n_pts, nullptr, nullptr, match_index, proj, where, 0);
}

int emp_find_match_points(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                          const int32_t* n_ref, const double* xy, const int32_t* n_pts, const int32_t* is_first_run,
                          const int32_t* pre_match_index, int32_t* match_index, double* proj, emp_mem where) {
    return match_common(ctx, B, max_ref, max_pts, ref_line, n_ref, xy, n_pts, is_first_run, pre_match_index,
                        match_index, proj, where, 1);
}

int emp_heading_kappa(emp_ctx* ctx, int32_t B, int32_t max_pts, const double* xy, const int32_t* n_pts, double* theta,
                      double* kappa, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_pts >= 2 && xy && n_pts && theta && kappa, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double* d_xy;
    const int* d_np;
    double *d_t, *d_k;
    if ((rc = st.in(xy, (size_t)B * max_pts * 2, &d_xy))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.out(theta, (size_t)B * max_pts, &d_t))) return rc;
    if ((rc = st.out(kappa, (size_t)B * max_pts, &d_k))) return rc;
    if (B) {
        KernelTimer t(ctx, "heading");
        hipLaunchKernelGGL(heading_kappa_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_pts, d_xy, d_np, d_t, d_k);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_lmin_lmax(emp_ctx* ctx, int32_t B, int32_t max_pts, int32_t max_obs, const double* dp_s, const double* dp_l,
                  const int32_t* n_pts, const double* obs_s, const double* obs_l, const int32_t* n_obs,
                  double obs_length, double obs_width, double* l_min, double* l_max, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_pts >= 1 && max_obs >= 0, "bad sizes");
    EMP_REQUIRE(ctx, dp_s && dp_l && n_pts && n_obs && l_min && l_max && status, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_s, *d_l, *d_os, *d_ol;
    const int *d_np, *d_no;
    double *d_lo, *d_hi;
    int* d_st;
    if ((rc = st.in(dp_s, (size_t)B * max_pts, &d_s))) return rc;
    if ((rc = st.in(dp_l, (size_t)B * max_pts, &d_l))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(obs_s, (size_t)B * max_obs, &d_os))) return rc;
    if ((rc = st.in(obs_l, (size_t)B * max_obs, &d_ol))) return rc;
    if ((rc = st.in(n_obs, (size_t)B, &d_no))) return rc;
    if ((rc = st.out(l_min, (size_t)B * max_pts, &d_lo))) return rc;
    if ((rc = st.out(l_max, (size_t)B * max_pts, &d_hi))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (B) {
        KernelTimer t(ctx, "lmin_lmax");
        hipLaunchKernelGGL(lmin_lmax_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_pts, max_obs, d_s, d_l, d_np,
                           d_os, d_ol, d_no, obs_length, obs_width, d_lo, d_hi, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_path_qp(emp_ctx* ctx, const emp_qp_params* q, int32_t B, int32_t max_pts, const double* l_min,
                const double* l_max, const int32_t* n_pts, const double* start_l3, double* qp_l, double* qp_dl,
                double* qp_ddl, int32_t* iters, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, q && B >= 0 && max_pts >= 1 && max_pts <= 256, "bad sizes (max_pts must be in [1, 256])");
    EMP_REQUIRE(ctx, qp_reserved_ok(q), "emp_qp_params.reserved must be 0 (start from emp_qp_params_default)");
    EMP_REQUIRE(ctx, l_min && l_max && n_pts && start_l3 && qp_l && qp_dl && qp_ddl && status, "NULL argument");
    EMP_REQUIRE(ctx, q->ds > 0, "ds must be > 0");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_lo, *d_hi, *d_s3;
    const int* d_np;
    double *d_l, *d_dl, *d_ddl;
    int *d_it, *d_st;
    if ((rc = st.in(l_min, (size_t)B * max_pts, &d_lo))) return rc;
    if ((rc = st.in(l_max, (size_t)B * max_pts, &d_hi))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(start_l3, (size_t)B * 3, &d_s3))) return rc;
    if ((rc = st.out(qp_l, (size_t)B * max_pts, &d_l))) return rc;
    if ((rc = st.out(qp_dl, (size_t)B * max_pts, &d_dl))) return rc;
    if ((rc = st.out(qp_ddl, (size_t)B * max_pts, &d_ddl))) return rc;
    if ((rc = st.out(iters, (size_t)B, &d_it))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (B) {
        const QpDev Q = make_qp_dev(q);
        const size_t lds = (size_t)path_qp_words(max_pts) * sizeof(double);
        if ((rc = set_lds(ctx, path_qp_wave_kernel, lds))) return rc;
        KernelTimer t(ctx, "path_qp");
        hipLaunchKernelGGL(path_qp_wave_kernel, dim3(B), dim3(64), lds, ctx->stream, B, max_pts, max_pts, Q, d_lo, d_hi,
                           d_np, d_s3, d_l, d_dl, d_ddl, d_it, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_smooth_line(emp_ctx* ctx, const emp_smooth_params* sp, int32_t B, int32_t max_pts, const double* xy,
                    const int32_t* n_pts, double* out, int32_t* iters, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, sp && B >= 0 && max_pts >= 2 && max_pts <= 256, "bad sizes (max_pts must be in [2, 256])");
    EMP_REQUIRE(ctx, xy && n_pts && out && status, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double* d_xy;
    const int* d_np;
    double* d_out;
    int *d_it, *d_st;
    if ((rc = st.in(xy, (size_t)B * max_pts * 2, &d_xy))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.out(out, (size_t)B * max_pts * 4, &d_out))) return rc;
    if ((rc = st.out(iters, (size_t)B, &d_it))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (B) {
        const SmoothQpParams sx{sp->w_smooth, sp->w_length, sp->w_ref, sp->x_thre};
        const SmoothQpParams sy{sp->w_smooth, sp->w_length, sp->w_ref, sp->y_thre};
        const size_t lds = (2 * (size_t)BoxRangeQp::words(max_pts, max_pts) + (size_t)max_pts) * sizeof(double);
        auto kern = max_pts > 32 ? smooth_wave_kernel<true> : smooth_wave_kernel<false>;
        if ((rc = set_lds(ctx, kern, lds))) return rc;
        KernelTimer t(ctx, "smooth");
        hipLaunchKernelGGL(kern, dim3(B), dim3(64), lds, ctx->stream, B, max_pts, max_pts, sx, sy, d_xy, d_np,
                           d_out, d_it, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_reference_line(emp_ctx* ctx, const emp_smooth_params* sp, int32_t B, int32_t max_global,
                       const double* global_path, const int32_t* n_global, const double* pred_xy,
                       const int32_t* is_first_run, const int32_t* pre_match_index, double* ref_line, int32_t* n_ref,
                       int32_t* match_index, int32_t* iters, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, sp && B >= 0 && max_global >= 1, "bad sizes");
    EMP_REQUIRE(ctx, global_path && n_global && pred_xy && pre_match_index && ref_line && n_ref && match_index && status,
                "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_g, *d_xy;
    const int *d_ng, *d_first = nullptr, *d_pre;
    double* d_ref;
    int *d_nr, *d_m, *d_it = nullptr, *d_st;
    if ((rc = st.in(global_path, (size_t)B * max_global * 4, &d_g))) return rc;
    if ((rc = st.in(n_global, (size_t)B, &d_ng))) return rc;
    if ((rc = st.in(pred_xy, (size_t)B * 2, &d_xy))) return rc;
    if (is_first_run && (rc = st.in(is_first_run, (size_t)B, &d_first))) return rc;
    if ((rc = st.in(pre_match_index, (size_t)B, &d_pre))) return rc;
    if ((rc = st.out(ref_line, (size_t)B * kRefLinePoints * 4, &d_ref, false))) return rc;
    if ((rc = st.out(n_ref, (size_t)B, &d_nr, false))) return rc;
    if ((rc = st.out(match_index, (size_t)B, &d_m, false))) return rc;
    if (iters && (rc = st.out(iters, (size_t)B, &d_it, false))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        const SmoothQpParams sx{sp->w_smooth, sp->w_length, sp->w_ref, sp->x_thre};
        const SmoothQpParams sy{sp->w_smooth, sp->w_length, sp->w_ref, sp->y_thre};
        const size_t lds = (2 * (size_t)kRefLinePoints + 2 * (size_t)BoxRangeQp::words(kRefLinePoints, kRefLinePoints) +
                            (size_t)kRefLinePoints) * sizeof(double);
        if ((rc = set_lds(ctx, reference_line_wave_kernel, lds))) return rc;
        KernelTimer t(ctx, "reference_line");
        hipLaunchKernelGGL(reference_line_wave_kernel, dim3(B), dim3(64), lds, ctx->stream, B, max_global, sx, sy, d_g, d_ng,
                           d_xy, d_first, d_pre, d_ref, d_nr, d_m, d_it, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_frenet_path_to_xy(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                          const double* s_map, const int32_t* n_ref, const double* begin_sl, const double* path_s,
                          const double* path_l, const int32_t* n_pts, double* target_xy, int32_t* n_out,
                          int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 2 && max_pts >= 1, "bad sizes");
    EMP_REQUIRE(ctx, ref_line && s_map && n_ref && begin_sl && path_s && path_l && n_pts && target_xy && n_out && status,
                "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_sm, *d_b, *d_ps, *d_pl;
    const int *d_nr, *d_np;
    double* d_t;
    int *d_no, *d_st;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(s_map, (size_t)B * max_ref, &d_sm))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(begin_sl, (size_t)B * 2, &d_b))) return rc;
    if ((rc = st.in(path_s, (size_t)B * max_pts, &d_ps))) return rc;
    if ((rc = st.in(path_l, (size_t)B * max_pts, &d_pl))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.out(target_xy, (size_t)B * (max_pts + 1) * 2, &d_t))) return rc;
    if ((rc = st.out(n_out, (size_t)B, &d_no))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (B) {
        KernelTimer t(ctx, "path_to_xy");
        hipLaunchKernelGGL(path_to_xy_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_ref, max_pts, d_ref, d_sm,
                           d_nr, d_b, d_ps, d_pl, d_np, d_t, d_no, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_plan_cycle(emp_ctx* ctx, const emp_dp_params* p, const emp_qp_params* q, const emp_smooth_params* sp, int32_t B,
                   int32_t max_ref, int32_t max_obs, int32_t max_pts, emp_dp_mode mode, const emp_cycle_io* io,
                   emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, p && q && sp && io, "NULL parameter struct");
    EMP_REQUIRE(ctx, qp_reserved_ok(q), "emp_qp_params.reserved must be 0 (start from emp_qp_params_default)");
    // with a dynamic obstacle per scene (test_9.py:137-169) up to three virtual obstacles join the projected ones
    const bool has_dyn = io->dyn_dis_speed != nullptr;
    const int obs_cap = max_obs + (has_dyn ? 3 : 0);
    DpDev d;
    int rc = make_dp_dev(ctx, p, B, obs_cap, &d);
    if (rc) return rc;
    EMP_REQUIRE(ctx, max_ref >= 2 && max_pts >= 2 && max_pts <= 255, "max_ref >= 2 and 2 <= max_pts <= 255 required");
    // the optional front end (ABI 11): find_match_points on the global path, sampling, smooth_reference_line in front of the cycle
    const bool front = io->global_path != nullptr;
    EMP_REQUIRE(ctx, (front || (io->ref_line && io->n_ref)) && io->origin_xy && io->start_xy && io->start_v && io->start_a,
                "cycle inputs missing");
    EMP_REQUIRE(ctx, !front || (io->n_global && io->pre_match_index && io->match_index && io->ref_status && io->max_global >= 1 &&
                                max_ref == kRefLinePoints),
                "front end: n_global, pre_match_index, match_index, ref_status, max_global >= 1 and max_ref == EMP_REF_LINE_POINTS");
    EMP_REQUIRE(ctx, max_obs == 0 || (io->obs_xy && io->n_obs), "obstacle inputs missing");
    EMP_REQUIRE(ctx, io->traj && io->traj_len && io->status, "traj, traj_len and status are required outputs");
    EMP_REQUIRE(ctx, q->ds > 0, "ds must be > 0");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    // Pipelined modes (device pointers only; emp_context.h).  LANES: the whole call runs on the next lane - its stream
    // stands in for ctx->stream and its pool for ctx->pool - behind whatever the caller has ordered on the main stream so
    // far.  STAGED: the call takes the pool of the next of two lanes once the back stage that used it last is done; its
    // front stage runs on the main stream.
    // EMP_HOST_PINNED: the caller's arrays are page-locked; the call is pipelined like a device-pointer call, its inputs
    // arrive over the copy stream and its outputs leave over the d2h stream (emp_context.h Stage: async_host)
    const bool pinned = where == EMP_HOST_PINNED;
    if (pinned && !ctx->copy_stream) {
        EMP_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        EMP_HIP(ctx, hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
        EMP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_h2d, hipEventDisableTiming));
        EMP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_host_last, hipEventDisableTiming));
    }
    const int pmode = ((where == EMP_DEVICE || pinned) && B > 0) ? ctx->pipe_mode : 0;
    const bool piped = pmode != 0, staged = pmode == EMP_PIPELINE_STAGED;
    struct LaneSwap {
        emp_ctx* c;
        emp_ctx::Lane* ln = nullptr;
        hipStream_t main_stream;
        LaneSwap(emp_ctx* c_, int mode) : c(c_), main_stream(c_->stream) {
            if (!mode) return;
            c->lane = (c->lane + 1) % c->lanes_in_use();
            ++c->cycle_calls;
            ln = &c->lanes[c->lane];                 // (ln->ticket changes below, after the host-side wait for the lane's previous call)
            std::swap(c->pool, ln->pool);
            if (mode != EMP_PIPELINE_STAGED) {
                c->active_lane = c->lane;
                c->stream = ln->stream;
            }
        }
        ~LaneSwap() {
            if (!ln) return;
            c->stream = main_stream;
            std::swap(c->pool, ln->pool);
            c->active_lane = -1;
        }
    } lane(ctx, pmode);
    if (staged) {
        // the pool's previous user is four calls back: a host-side wait (emp_context.h, kStagedPools)
        if (lane.ln->done_valid) EMP_HIP(ctx, hipEventSynchronize(lane.ln->ev_done));
        if (lane.ln->host_valid) EMP_HIP(ctx, hipEventSynchronize(lane.ln->ev_host));   // ... and its outputs have left the pool
    } else if (piped) {
        if (lane.ln->host_valid) EMP_HIP(ctx, hipEventSynchronize(lane.ln->ev_host));
        // The lane's previous occupant (call k - n) and whatever its caller queued behind it on the lane's stream (record
        // packing) must be done before anything ordered on the main stream from here on may touch memory they use: the
        // caller keeps a call's outputs alive only until this call is issued, and the NEXT call runs on another lane.
        if (lane.ln->done_valid) {
            EMP_HIP(ctx, hipEventRecord(lane.ln->ev_tail, ctx->stream));
            EMP_HIP(ctx, hipStreamWaitEvent(lane.main_stream, lane.ln->ev_tail, 0));
        }
        EMP_HIP(ctx, hipEventRecord(lane.ln->ev_in, lane.main_stream));
        EMP_HIP(ctx, hipStreamWaitEvent(ctx->stream, lane.ln->ev_in, 0));
    }
    if (piped) {
        // Only now - the previous occupant's outputs are in its caller's arrays - does the lane stop answering for the previous
        // ticket: emp_wait_ticket(T) on another thread either still finds the lane under T with its event valid and waits for
        // the same event, or finds no lane under T, which now MEANS that this thread has waited for T already (the advisor's
        // round-5 finding: the ticket used to change before the wait, and a waiter in that window returned at once).
        lane.ln->host_valid = false;
        lane.ln->ticket = ctx->cycle_calls;
        // A pinned call's inputs go into the lane's pool over the copy stream.  If the lane's previous occupant was an EMP_DEVICE
        // cycle there has been no host-side wait for it: the copy stream waits for its kernels instead.
        if (pinned && !staged && lane.ln->done_valid) EMP_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, lane.ln->ev_done, 0));
    }
    // EMP_OPT_CYCLE_GRAPH: one batch at a time on device pointers - the third consecutive call with one signature is captured,
    // the following ones are one hipGraphLaunch.  The signature is everything a launch argument is made of; the context's
    // allocation count says whether a temporary or a lattice table moved or changed since the capture.
    struct CaptureGuard {
        emp_ctx* c;
        bool active = false;
        ~CaptureGuard() {
            if (!active) return;            // (an early return inside the captured region: end the capture, keep nothing)
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(c->stream, &g);
            if (g) (void)hipGraphDestroy(g);
            c->capturing = false;
        }
    } capture{ctx};
    const bool graph_ok = ctx->opt[EMP_OPT_CYCLE_GRAPH] == 1 && pmode == 0 && where == EMP_DEVICE && B > 0 && !ctx->timing &&
                          !ctx->opt[EMP_OPT_SWEEP_CLOCK_PROBE] && !ctx->opt[EMP_OPT_EDGE_CLOCK_PROBE];
    if (graph_ok) {
        std::vector<unsigned long long> key;
        auto add = [&](const void* ptr, size_t bytes) {
            const unsigned char* b8 = (const unsigned char*)ptr;
            for (size_t o = 0; o < bytes; o += 8) {
                unsigned long long w = 0;
                memcpy(&w, b8 + o, std::min<size_t>(8, bytes - o));
                key.push_back(w);
            }
        };
        const long long sizes[5] = {B, max_ref, max_obs, max_pts, (long long)mode};
        add(sizes, sizeof(sizes));
        add(p, sizeof(*p));
        add(q, sizeof(*q));
        add(sp, sizeof(*sp));
        add(io, sizeof(*io));
        add(ctx->opt, sizeof(ctx->opt));
        if (ctx->cycle_graph && key == ctx->cycle_graph_key && ctx->alloc_gen == ctx->cycle_graph_gen) {
            EMP_HIP(ctx, hipGraphLaunch(ctx->cycle_graph, ctx->stream));
            ++ctx->cycle_graph_replays;
            return EMP_OK;
        }
        if (ctx->cycle_graph) {             // another signature, or the buffers moved: the graph is stale
            (void)hipGraphExecDestroy(ctx->cycle_graph);
            ctx->cycle_graph = nullptr;
        }
        ctx->cycle_seen = (key == ctx->cycle_seen_key) ? ctx->cycle_seen + 1 : 1;
        ctx->cycle_seen_key = key;
        if (ctx->cycle_seen >= 3) {         // the two calls before this one allocated and built whatever this signature needs
            ctx->cycle_graph_gen = ctx->alloc_gen;
            EMP_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
            capture.active = true;
            ctx->capturing = true;
        }
    } else if (ctx->cycle_graph) {
        (void)hipGraphExecDestroy(ctx->cycle_graph);
        ctx->cycle_graph = nullptr;
        ctx->cycle_seen = 0;
    }
    Stage st(ctx, where, piped, pinned);
    const double *d_ref = nullptr, *d_o, *d_sxy, *d_v, *d_a, *d_oxy, *d_glob = nullptr;
    const int *d_nr = nullptr, *d_no, *d_nglob = nullptr, *d_prem = nullptr;
    if (front) {
        if ((rc = st.in(io->global_path, (size_t)B * io->max_global * 4, &d_glob))) return rc;
        if ((rc = st.in(io->n_global, (size_t)B, &d_nglob))) return rc;
        if ((rc = st.in(io->pre_match_index, (size_t)B, &d_prem))) return rc;
    } else {
        if ((rc = st.in(io->ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
        if ((rc = st.in(io->n_ref, (size_t)B, &d_nr))) return rc;
    }
    if ((rc = st.in(io->origin_xy, (size_t)B * 2, &d_o))) return rc;
    if ((rc = st.in(io->start_xy, (size_t)B * 2, &d_sxy))) return rc;
    if ((rc = st.in(io->start_v, (size_t)B * 2, &d_v))) return rc;
    if ((rc = st.in(io->start_a, (size_t)B * 2, &d_a))) return rc;
    if ((rc = st.in(io->obs_xy, (size_t)B * max_obs * 2, &d_oxy))) return rc;
    if ((rc = st.in(io->n_obs, (size_t)B, &d_no))) return rc;
    const double* d_dyn = nullptr;
    if (has_dyn && (rc = st.in(io->dyn_dis_speed, (size_t)B * 2, &d_dyn))) return rc;
    if ((rc = st.inputs_ready())) return rc;
    // outputs (optional ones fall back to device temporaries); no memsets: every kernel of the cycle writes its
    // rows completely, padding included
    double *d_rows, *d_dps, *d_dpl, *d_ps, *d_pl, *d_traj;
    int *d_dplen, *d_plen, *d_tlen, *d_st;
    if ((rc = st.out(io->dp_rows, (size_t)B * d.col, &d_rows, false))) return rc;
    if (!d_rows && (rc = st.tmp((size_t)B * d.col, &d_rows, false))) return rc;
    if ((rc = st.out(io->dp_s, (size_t)B * max_pts, &d_dps, false))) return rc;
    if (!d_dps && (rc = st.tmp((size_t)B * max_pts, &d_dps, false))) return rc;
    if ((rc = st.out(io->dp_l, (size_t)B * max_pts, &d_dpl, false))) return rc;
    if (!d_dpl && (rc = st.tmp((size_t)B * max_pts, &d_dpl, false))) return rc;
    if ((rc = st.out(io->dp_len, (size_t)B, &d_dplen, false))) return rc;
    if (!d_dplen && (rc = st.tmp((size_t)B, &d_dplen, false))) return rc;
    if ((rc = st.out(io->path_s, (size_t)B * max_pts, &d_ps, false))) return rc;
    if (!d_ps && (rc = st.tmp((size_t)B * max_pts, &d_ps, false))) return rc;
    if ((rc = st.out(io->path_l, (size_t)B * max_pts, &d_pl, false))) return rc;
    if (!d_pl && (rc = st.tmp((size_t)B * max_pts, &d_pl, false))) return rc;
    if ((rc = st.out(io->path_len, (size_t)B, &d_plen, false))) return rc;
    if (!d_plen && (rc = st.tmp((size_t)B, &d_plen, false))) return rc;
    if ((rc = st.out(io->traj, (size_t)B * (max_pts + 1) * 4, &d_traj, false))) return rc;
    if ((rc = st.out(io->traj_len, (size_t)B, &d_tlen, false))) return rc;
    if ((rc = st.out(io->status, (size_t)B, &d_st, false))) return rc;
    int *d_match = nullptr, *d_rst = nullptr;
    if (front) {
        if ((rc = st.out(io->match_index, (size_t)B, &d_match, false))) return rc;
        if ((rc = st.out(io->ref_status, (size_t)B, &d_rst, false))) return rc;
    }
    if ((rc = st.outputs_ready())) return rc;
    // intermediates
    double *d_sm, *d_os, *d_ol, *d_bsl, *d_start;
    const int mo = obs_cap > 0 ? obs_cap : 1;
    if ((rc = st.tmp((size_t)B * max_ref, &d_sm))) return rc;
    if ((rc = st.tmp((size_t)B * mo, &d_os))) return rc;
    if ((rc = st.tmp((size_t)B * mo, &d_ol))) return rc;
    if ((rc = st.tmp((size_t)B * 2, &d_bsl))) return rc;
    if ((rc = st.tmp((size_t)B * 4, &d_start))) return rc;
    int* d_zero_nobs = nullptr;
    if (max_obs == 0) {
        if ((rc = st.tmp((size_t)B, &d_zero_nobs, true))) return rc;
        d_no = d_zero_nobs;
    }
    int* d_ntot = nullptr;
    if (has_dyn && (rc = st.tmp((size_t)B, &d_ntot, false))) return rc;
    double* d_ref_w = nullptr;
    int* d_nr_w = nullptr;
    if (front) {
        if ((rc = st.tmp((size_t)B * kRefLinePoints * 4, &d_ref_w))) return rc;
        if ((rc = st.tmp((size_t)B, &d_nr_w))) return rc;
    }
    if (B == 0) return st.finish();
    if (front) {          // ref test_9.py:99-110, one wavefront per scene; the predicted location is the planning start
        const SmoothQpParams sx{sp->w_smooth, sp->w_length, sp->w_ref, sp->x_thre};
        const SmoothQpParams sy{sp->w_smooth, sp->w_length, sp->w_ref, sp->y_thre};
        const size_t lds = (2 * (size_t)kRefLinePoints + 2 * (size_t)BoxRangeQp::words(kRefLinePoints, kRefLinePoints) +
                            (size_t)kRefLinePoints) * sizeof(double);
        if ((rc = set_lds(ctx, reference_line_wave_kernel, lds))) return rc;
        KernelTimer t(ctx, "reference_line");
        hipLaunchKernelGGL(reference_line_wave_kernel, dim3(B), dim3(64), lds, ctx->stream, B, (int)io->max_global, sx, sy, d_glob,
                           d_nglob, d_sxy, (const int*)nullptr, d_prem, d_ref_w, d_nr_w, d_match, (int*)nullptr, d_rst, 2);
        EMP_LAUNCH_CHECK(ctx);
        d_ref = d_ref_w;
        d_nr = d_nr_w;
    }
    if ((rc = dev_project(ctx, B, max_ref, max_obs, d_ref, d_nr, d_o, d_sxy, d_v, d_a, d_oxy, d_no, d_sm, d_os, d_ol,
                          d_bsl, d_start, mo, d_dyn, d_ntot)))
        return rc;
    if (has_dyn) d_no = d_ntot;                            // downstream stages see the projected + virtual obstacles
    // the sweep may leave the backtrack to the densification kernel (emp_dp_kernels.h, BT == false): two temporaries for it
    unsigned char* d_pre = nullptr;
    int* d_term = nullptr;
    if (mode == EMP_DP_TWO_KERNEL && !wide(d)) {
        if ((rc = st.tmp((size_t)d.tiles * d.col * 64, &d_pre, false))) return rc;
        if ((rc = st.tmp((size_t)B, &d_term, false))) return rc;
    }
    ctx->front_stop = staged ? lane.ln->ev_front : nullptr;
    ctx->front_attached = nullptr;
    ctx->sweep_wait = nullptr;
    if (staged && ctx->opt[EMP_OPT_SWEEP_EXCLUSIVE]) {       // the previous call's back stage: its lane is the one before ours
        emp_ctx::Lane& prev = ctx->lanes[(ctx->lane + ctx->lanes_in_use() - 1) % ctx->lanes_in_use()];
        if (ctx->opt[EMP_OPT_SWEEP_EXCLUSIVE] == 2) {        // ... only its densification and path QP: the sweep runs beside the Cartesian tail
            if (prev.qp_valid) ctx->sweep_wait = prev.ev_qp;
        } else if (prev.done_valid) {
            ctx->sweep_wait = prev.ev_done;
        }
    }
    ctx->edge_wait = nullptr;
    if (staged && ctx->opt[EMP_OPT_EDGE_AFTER_ENRICH] && mode == EMP_DP_TWO_KERNEL && !wide(d)) {
        emp_ctx::Lane& prev = ctx->lanes[(ctx->lane + ctx->lanes_in_use() - 1) % ctx->lanes_in_use()];
        if (prev.enrich_valid) ctx->edge_wait = prev.ev_enrich;
    }
    ctx->bt_pre = d_pre;
    ctx->bt_term = d_term;
    ctx->bt_deferred = false;
    rc = dev_dp_plan(ctx, d, d_os, d_ol, d_no, d_start, mode, d_rows, nullptr, d_st);
    ctx->front_stop = nullptr;
    ctx->sweep_wait = nullptr;
    ctx->edge_wait = nullptr;
    ctx->bt_pre = nullptr;
    ctx->bt_term = nullptr;
    const bool deferred = ctx->bt_deferred;
    ctx->bt_deferred = false;
    if (rc) return rc;
    const QpDev Q = make_qp_dev(q);
    // EMP_OPT_SWEEP_EXCLUSIVE: a marker behind the sweep on the front stream.  Measured, not understood: which of two
    // regimes the two queues settle in depends on it.  With it the kernels keep the durations of the overlapped step (edge
    // 198 us, path QP 140, Cartesian 70, sweep 19) and the step takes 0.264 ms (mode 2) / 0.277 (mode 1); without it the
    // edge kernel runs at its stand-alone 147 us, starves the path QP beside it (250 us) and the step takes 0.32 - 0.35 ms
    // (profiles/r04_sweep/README.md).  The clock probe's event did the same by accident, which is how this was found.
    if (staged && ctx->opt[EMP_OPT_SWEEP_EXCLUSIVE]) {
        if (!ctx->sweep_marker) EMP_HIP(ctx, hipEventCreateWithFlags(&ctx->sweep_marker, hipEventDisableTiming));
        EMP_HIP(ctx, hipEventRecord(ctx->sweep_marker, ctx->stream));
    }
    if (staged) {      // the back stage (short kernels that last as long as their slowest scene) goes to the back stream
        // behind the sweep's own completion event where the launch attached one (a marker packet behind the sweep costs the
        // front queue ~5 us per step), else behind an event recorded here
        hipEvent_t front_done = ctx->front_attached;
        if (!front_done) {
            EMP_HIP(ctx, hipEventRecord(lane.ln->ev_front, ctx->stream));
            front_done = lane.ln->ev_front;
        }
        EMP_HIP(ctx, hipStreamWaitEvent(ctx->back_stream, front_done, 0));
        ctx->stream = ctx->back_stream;        // ~LaneSwap puts the main stream back
    }
    // The two events the NEXT call's front stage waits for (densification done, path QP done) are signalled by the dispatches
    // themselves where possible: a marker packet behind each idled the back queue ~6 us, and with the edge kernel's round-4 diet
    // the back queue is what bounds the step.
    const bool want_enrich_ev = staged && ctx->opt[EMP_OPT_EDGE_AFTER_ENRICH] != 0;
    ctx->attach_stop = want_enrich_ev ? lane.ln->ev_enrich : nullptr;
    ctx->stop_attached = false;
    if ((rc = dev_dp_enrich(ctx, d, d_rows, d_start, max_pts, d_dps, d_dpl, d_dplen, d_st, 1, deferred ? d_pre : nullptr,
                            deferred ? d_term : nullptr, d_no, d_rows))) {
        ctx->attach_stop = nullptr;
        return rc;
    }
    ctx->attach_stop = nullptr;
    if (want_enrich_ev) {
        if (!ctx->stop_attached) EMP_HIP(ctx, hipEventRecord(lane.ln->ev_enrich, ctx->stream));
        lane.ln->enrich_valid = true;
    } else if (staged) {
        lane.ln->enrich_valid = false;
    }
    ctx->attach_stop = staged ? lane.ln->ev_qp : nullptr;
    ctx->stop_attached = false;
    rc = dev_cycle_qp(ctx, B, max_pts, mo, Q, d_dps, d_dpl, d_dplen, d_os, d_ol, d_no, d_start, d_ps, d_pl, d_plen, d_st);
    ctx->attach_stop = nullptr;
    if (rc) return rc;
    if (staged) {
        if (!ctx->stop_attached) EMP_HIP(ctx, hipEventRecord(lane.ln->ev_qp, ctx->stream));
        lane.ln->qp_valid = true;
    }
    const int path_cap = (max_pts + Q.decimate - 1) / Q.decimate + (Q.midpoint ? 1 : 0);
    if ((rc = dev_cycle_cartesian(ctx, B, max_ref, max_pts, path_cap, sp, d_ref, d_sm, d_nr, d_bsl, d_ps, d_pl, d_plen,
                                  d_traj, d_tlen, d_st)))
        return rc;
    if (piped) {
        EMP_HIP(ctx, hipEventRecord(lane.ln->ev_done, ctx->stream));
        lane.ln->done_valid = true;
    }
    if (st.async_host()) {
        // outputs go home on the d2h stream behind the cycle's last kernel; pipelined: nobody waits here (emp_wait_cycle,
        // emp_synchronize, or the call that takes this pool over); not pipelined: the pool is the next call's, so wait
        if (piped) {
            if (!lane.ln->ev_host) EMP_HIP(ctx, hipEventCreateWithFlags(&lane.ln->ev_host, hipEventDisableTiming));
            if ((rc = st.finish_async(lane.ln->ev_done, lane.ln->ev_host))) return rc;
            lane.ln->host_valid = true;
            return EMP_OK;
        }
        EMP_HIP(ctx, hipEventRecord(ctx->ev_h2d, ctx->stream));
        if ((rc = st.finish_async(ctx->ev_h2d, ctx->ev_host_last))) return rc;
        EMP_HIP(ctx, hipEventSynchronize(ctx->ev_host_last));
        return EMP_OK;
    }
    if (capture.active) {
        hipGraph_t g = nullptr;
        capture.active = false;
        ctx->capturing = false;
        EMP_HIP(ctx, hipStreamEndCapture(ctx->stream, &g));
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        EMP_HIP(ctx, ie);
        if (ctx->alloc_gen != ctx->cycle_graph_gen) {      // something was allocated or rebuilt inside the capture after all
            (void)hipGraphExecDestroy(exec);
            ctx->cycle_seen = 0;
            return emp::fail(ctx, EMP_ERR_HIP, "EMP_OPT_CYCLE_GRAPH: the context's buffers changed inside a capture");
        }
        ctx->cycle_graph = exec;
        ctx->cycle_graph_key = ctx->cycle_seen_key;
        EMP_HIP(ctx, hipGraphLaunch(exec, ctx->stream));       // the captured launches have not run yet: this is the call's work
    }
    return st.finish();
}

int64_t emp_cycle_graph_replays(emp_ctx* ctx) { return ctx ? (int64_t)ctx->cycle_graph_replays : -1; }

uint64_t emp_cycle_ticket(emp_ctx* ctx) { return ctx ? ctx->cycle_calls : 0; }

int emp_wait_cycle(emp_ctx* ctx, int32_t calls_back) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    if (!ctx->pipelined()) return EMP_OK;                       // a non-pipelined call returned with its outputs in place
    const int n = ctx->lanes_in_use();
    EMP_REQUIRE(ctx, calls_back >= 0 && calls_back < n, "calls_back beyond the pipeline depth");
    emp_ctx::Lane& ln = ctx->lanes[((ctx->lane - calls_back) % n + n) % n];
    if (ln.host_valid) EMP_HIP(ctx, hipEventSynchronize(ln.ev_host));
    return EMP_OK;
}

int emp_wait_ticket(emp_ctx* ctx, uint64_t ticket) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    // Reads two atomics per lane (emp_context.h Lane).  The lane's event exists since the call that got the ticket; a call
    // that takes the lane over waits - on the host - for that same event BEFORE it changes the lane's ticket, so a ticket that no
    // lane holds any more has been waited for already.  (A waiter that read the old ticket just before it changed waits for the
    // event as re-recorded by the new call: longer than needed, never shorter.)
    for (auto& ln : ctx->lanes)
        if (ln.ticket == ticket && ln.host_valid && ln.ev_host) {
            const hipError_t e = hipEventSynchronize(ln.ev_host);
            if (e != hipSuccess) return EMP_ERR_HIP;       // (no ctx->err write: another thread may be inside a call)
            break;
        }
    return EMP_OK;
}

int emp_host_alloc(emp_ctx* ctx, uint64_t bytes, void** out) {
    EMP_REQUIRE(ctx, ctx && out, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    EMP_HIP(ctx, hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault));
    ctx->pinned.push_back({p, (size_t)(bytes ? bytes : 8)});
    *out = p;
    return EMP_OK;
}
int emp_host_free(emp_ctx* ctx, void* ptr) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    auto it = std::find_if(ctx->pinned.begin(), ctx->pinned.end(), [&](const emp_ctx::Pinned& a) { return a.p == ptr; });
    EMP_REQUIRE(ctx, it != ctx->pinned.end(), "not an emp_host_alloc pointer of this context");
    EMP_HIP(ctx, (hipError_t)sync_all(ctx));
    EMP_HIP(ctx, hipHostFree(ptr));
    ctx->pinned.erase(it);
    return EMP_OK;
}

int emp_quintic_coefficients(emp_ctx* ctx, int32_t n, const double* bc, double* coeff, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && bc && coeff, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double* d_bc;
    double* d_c;
    if ((rc = st.in(bc, (size_t)n * 8, &d_bc))) return rc;
    if ((rc = st.out(coeff, (size_t)n * 6, &d_c))) return rc;
    if (n) {
        hipLaunchKernelGGL(quintic_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, d_bc, d_c);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_obs_cost_n(emp_ctx* ctx, int32_t n, int32_t samples, double w_collision, double danger_dis, double safe_dis,
                   const double* square_d, double* cost, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && samples >= 0 && cost && (square_d || (size_t)n * samples == 0), "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double* d_sq;
    double* d_c;
    if ((rc = st.in(square_d, (size_t)n * samples, &d_sq))) return rc;
    if ((rc = st.out(cost, (size_t)n, &d_c))) return rc;
    if (n) {
        hipLaunchKernelGGL(obs_cost_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, samples, w_collision, danger_dis, safe_dis,
                           d_sq, d_c);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_obs_cost(emp_ctx* ctx, int32_t n, double w_collision, double danger_dis, double safe_dis, const double* square_d,
                 double* cost, emp_mem where) {
    return emp_obs_cost_n(ctx, n, kSamples, w_collision, danger_dis, safe_dis, square_d, cost, where);
}

int emp_free_edge_costs(emp_ctx* ctx, int32_t n, int32_t max_obs, const double* edges, const double* obs_s, const double* obs_l,
                        const int32_t* n_obs, double w_collision, const double* w_smooth3, double w_ref, double* cost, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && max_obs >= 0 && edges && w_smooth3 && cost, "bad argument");
    EMP_REQUIRE(ctx, max_obs == 0 || (obs_s && obs_l && n_obs), "obs_s / obs_l / n_obs are required when max_obs > 0");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    const double w0 = w_smooth3[0], w1 = w_smooth3[1], w2 = w_smooth3[2];      // host memory, like the parameter structs
    Stage st(ctx, where);
    int rc;
    const double *d_e, *d_os, *d_ol;
    const int* d_n;
    double* d_c;
    if ((rc = st.in(edges, (size_t)n * 8, &d_e))) return rc;
    if ((rc = st.in(obs_s, (size_t)n * max_obs, &d_os))) return rc;
    if ((rc = st.in(obs_l, (size_t)n * max_obs, &d_ol))) return rc;
    if ((rc = st.in(n_obs, (size_t)n, &d_n))) return rc;
    if ((rc = st.out(cost, (size_t)n, &d_c))) return rc;
    if (n) {
        hipLaunchKernelGGL(free_edge_cost_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, max_obs, d_e, d_os, d_ol, d_n, w_collision,
                           w0, w1, w2, w_ref, d_c);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

}  // extern "C"

// ---- stand-alone projection helpers and the utilities beside the path ---------------------------
extern "C" {

int emp_s_map(emp_ctx* ctx, int32_t B, int32_t max_ref, const double* ref_line, const int32_t* n_ref,
              const double* origin_xy, double* s_map, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 1 && ref_line && n_ref && origin_xy && s_map, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_o;
    const int* d_nr;
    double* d_sm;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(origin_xy, (size_t)B * 2, &d_o))) return rc;
    if ((rc = st.out(s_map, (size_t)B * max_ref, &d_sm))) return rc;
    if (B) {
        hipLaunchKernelGGL(s_map_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_ref, d_ref, d_nr, d_o, d_sm);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_s_l(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line, const double* s_map,
            const int32_t* n_ref, const double* xy, const int32_t* n_pts, const int32_t* match_index, double* s,
            double* l, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 1 && max_pts >= 1 && ref_line && s_map && n_ref && xy && n_pts && s,
                "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_sm, *d_xy;
    const int *d_nr, *d_np, *d_mi;
    double *d_s, *d_l;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(s_map, (size_t)B * max_ref, &d_sm))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(xy, (size_t)B * max_pts * 2, &d_xy))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(match_index, (size_t)B * max_pts, &d_mi))) return rc;
    if ((rc = st.out(s, (size_t)B * max_pts, &d_s))) return rc;
    if ((rc = st.out(l, (size_t)B * max_pts, &d_l))) return rc;
    if (B) {
        hipLaunchKernelGGL(s_l_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_ref, max_pts, d_ref, d_sm, d_nr,
                           d_xy, d_np, d_mi, d_s, d_l);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_s_l_deri(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                 const int32_t* n_ref, const double* xy, const double* v_xy, const double* a_xy, const int32_t* n_pts,
                 const double* origin_xy, double* out, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 1 && max_pts >= 1 && ref_line && n_ref && xy && v_xy && a_xy && n_pts &&
                         origin_xy && out, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_xy, *d_v, *d_a, *d_o;
    const int *d_nr, *d_np;
    double* d_out;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(xy, (size_t)B * max_pts * 2, &d_xy))) return rc;
    if ((rc = st.in(v_xy, (size_t)B * max_pts * 2, &d_v))) return rc;
    if ((rc = st.in(a_xy, (size_t)B * max_pts * 2, &d_a))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(origin_xy, (size_t)B * 2, &d_o))) return rc;
    if ((rc = st.out(out, (size_t)B * max_pts * 7, &d_out))) return rc;
    if (B) {
        hipLaunchKernelGGL(s_l_deri_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_ref, max_pts, d_ref, d_nr,
                           d_xy, d_v, d_a, d_np, d_o, d_out);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_proj_point(emp_ctx* ctx, int32_t n, int32_t max_ref, const double* ref_line, const double* s_map,
                   const int32_t* n_ref, const double* s, const int32_t* pre_match_index, double* out, int32_t* index,
                   int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && max_ref >= 1 && ref_line && s_map && n_ref && s && pre_match_index && out && index &&
                         status, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_sm, *d_s;
    const int *d_nr, *d_pre;
    double* d_out;
    int *d_idx, *d_st;
    if ((rc = st.in(ref_line, (size_t)n * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(s_map, (size_t)n * max_ref, &d_sm))) return rc;
    if ((rc = st.in(n_ref, (size_t)n, &d_nr))) return rc;
    if ((rc = st.in(s, (size_t)n, &d_s))) return rc;
    if ((rc = st.in(pre_match_index, (size_t)n, &d_pre))) return rc;
    if ((rc = st.out(out, (size_t)n * 4, &d_out))) return rc;
    if ((rc = st.out(index, (size_t)n, &d_idx))) return rc;
    if ((rc = st.out(status, (size_t)n, &d_st))) return rc;
    if (n) {
        hipLaunchKernelGGL(proj_point_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, max_ref, d_ref, d_sm, d_nr, d_s,
                           d_pre, d_out, d_idx, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_trajectory_index2s(emp_ctx* ctx, int32_t B, int32_t max_pts, const double* x, const double* y,
                           const int32_t* n_pts, double* index2s, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_pts >= 1 && x && y && n_pts && index2s, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_x, *d_y;
    const int* d_np;
    double* d_o;
    if ((rc = st.in(x, (size_t)B * max_pts, &d_x))) return rc;
    if ((rc = st.in(y, (size_t)B * max_pts, &d_y))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.out(index2s, (size_t)B * max_pts, &d_o))) return rc;
    if (B) {
        hipLaunchKernelGGL(index2s_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_pts, d_x, d_y, d_np, d_o);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_frenet2cartesian(emp_ctx* ctx, int32_t B, int32_t max_ref, int32_t max_pts, const double* ref_line,
                         const double* index2s, const int32_t* n_ref, const double* sl, const int32_t* n_pts,
                         double* out, int32_t* status, int32_t proj_only, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_ref >= 2 && max_pts >= 1 && ref_line && index2s && n_ref && sl && n_pts && out &&
                         status, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ref, *d_i2s, *d_sl;
    const int *d_nr, *d_np;
    double* d_out;
    int* d_st;
    if ((rc = st.in(ref_line, (size_t)B * max_ref * 4, &d_ref))) return rc;
    if ((rc = st.in(index2s, (size_t)B * max_ref, &d_i2s))) return rc;
    if ((rc = st.in(n_ref, (size_t)B, &d_nr))) return rc;
    if ((rc = st.in(sl, (size_t)B * max_pts * 4, &d_sl))) return rc;
    if ((rc = st.in(n_pts, (size_t)B, &d_np))) return rc;
    if ((rc = st.out(out, (size_t)B * max_pts * 4, &d_out))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (B) {
        hipLaunchKernelGGL(frenet2cartesian_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_ref, max_pts, d_ref,
                           d_i2s, d_nr, d_sl, d_np, d_out, d_st, proj_only);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_dy_obs_deri(emp_ctx* ctx, int32_t n, const double* in, double* out, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && in && out, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double* d_in;
    double* d_out;
    if ((rc = st.in(in, (size_t)n * 5, &d_in))) return rc;
    if ((rc = st.out(out, (size_t)n * 3, &d_out))) return rc;
    if (n) {
        hipLaunchKernelGGL(dy_obs_deri_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, d_in, d_out);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

}  // extern "C"

extern "C" int emp_enrich_nodes(emp_ctx* ctx, int32_t B, int32_t max_nodes, double resolution, const double* node_s,
                                const double* node_l, const int32_t* n_nodes, const double* start, int32_t max_pts,
                                double* path_s, double* path_l, int32_t* path_len, int32_t* status, emp_mem where) {
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_nodes >= 1 && max_pts >= 1 && resolution > 0, "bad sizes");
    EMP_REQUIRE(ctx, node_s && node_l && n_nodes && start && path_s && path_l && path_len && status, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_ns, *d_nl, *d_start;
    const int* d_nn;
    double *d_ps, *d_pl;
    int *d_len, *d_st;
    if ((rc = st.in(node_s, (size_t)B * max_nodes, &d_ns))) return rc;
    if ((rc = st.in(node_l, (size_t)B * max_nodes, &d_nl))) return rc;
    if ((rc = st.in(n_nodes, (size_t)B, &d_nn))) return rc;
    if ((rc = st.in(start, (size_t)B * 4, &d_start))) return rc;
    if ((rc = st.out(path_s, (size_t)B * max_pts, &d_ps))) return rc;
    if ((rc = st.out(path_l, (size_t)B * max_pts, &d_pl))) return rc;
    if ((rc = st.out(path_len, (size_t)B, &d_len))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st))) return rc;
    if (B) {
        hipLaunchKernelGGL(enrich_nodes_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_nodes, resolution, d_ns,
                           d_nl, d_nn, d_start, max_pts, d_ps, d_pl, d_len, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

// ---- S-T speed DP (reference planner/speed_planning_test.py) ------------------------------------
extern "C" {

void emp_speed_dp_params_default(emp_speed_dp_params* p) {
    if (!p) return;
    p->reference_speed = 50.0;
    p->w_cost_ref_speed = 4000.0;
    p->w_cost_accel = 100.0;
    p->w_cost_obs = 10000000.0;
}

static emp::StDev make_st_dev(const emp_speed_dp_params* p, int B, int max_obs) {
    emp::StDev d;
    d.B = B;
    d.max_obs = max_obs;
    d.w.v_ref = p->reference_speed;
    d.w.w_ref = p->w_cost_ref_speed;
    d.w.w_acc = p->w_cost_accel;
    d.w.w_obs = emp::st::make_pow_base(p->w_cost_obs);
    return d;
}

int emp_st_graph(emp_ctx* ctx, int32_t B, int32_t max_obs, const double* obs_s, const double* obs_l,
                 const double* obs_s_dot, const double* obs_l_dot, double* s_in, double* s_out, double* t_in,
                 double* t_out, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_obs >= 1, "bad sizes");
    EMP_REQUIRE(ctx, obs_s && obs_l && obs_s_dot && obs_l_dot && s_in && s_out && t_in && t_out, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const size_t n = (size_t)B * max_obs;
    const double *d_s, *d_l, *d_sd, *d_ld;
    double *d_si, *d_so, *d_ti, *d_to;
    if ((rc = st.in(obs_s, n, &d_s))) return rc;
    if ((rc = st.in(obs_l, n, &d_l))) return rc;
    if ((rc = st.in(obs_s_dot, n, &d_sd))) return rc;
    if ((rc = st.in(obs_l_dot, n, &d_ld))) return rc;
    if ((rc = st.out(s_in, n, &d_si, false))) return rc;
    if ((rc = st.out(s_out, n, &d_so, false))) return rc;
    if ((rc = st.out(t_in, n, &d_ti, false))) return rc;
    if ((rc = st.out(t_out, n, &d_to, false))) return rc;
    if (B) {
        KernelTimer t(ctx, "st_graph");
        hipLaunchKernelGGL(st_graph_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_obs, d_s, d_l, d_sd, d_ld, d_si,
                           d_so, d_ti, d_to);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

int emp_speed_dp(emp_ctx* ctx, const emp_speed_dp_params* p, int32_t B, int32_t max_obs, const double* s_in,
                 const double* s_out, const double* t_in, const double* t_out, const double* plan_start_s_dot,
                 double* cost, double* s_dot, int32_t* node, int32_t* end_node, double* speed_s, double* speed_t,
                 emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, p != nullptr, "speed dp params are NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_obs >= 1 && max_obs <= st::kMaxObs, "bad sizes (max_obs must be in [1, 64])");
    EMP_REQUIRE(ctx, s_in && s_out && t_in && t_out && plan_start_s_dot && end_node && speed_s && speed_t, "NULL argument");
    // ref :281 w_cost_obs ** (1.5 - d): complex for a negative base, and the reference fails on its next comparison
    EMP_REQUIRE(ctx, !(p->w_cost_obs < 0.0), "w_cost_obs must not be negative");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const size_t n = (size_t)B * max_obs, nt = (size_t)B * st::kRows * st::kCols;
    const double *d_si, *d_so, *d_ti, *d_to, *d_v;
    double *d_c = nullptr, *d_sd = nullptr, *d_ss, *d_tt;
    int *d_n = nullptr, *d_e;
    if ((rc = stg.in(s_in, n, &d_si))) return rc;
    if ((rc = stg.in(s_out, n, &d_so))) return rc;
    if ((rc = stg.in(t_in, n, &d_ti))) return rc;
    if ((rc = stg.in(t_out, n, &d_to))) return rc;
    if ((rc = stg.in(plan_start_s_dot, (size_t)B, &d_v))) return rc;
    if (cost && (rc = stg.out(cost, nt, &d_c, false))) return rc;
    if (s_dot && (rc = stg.out(s_dot, nt, &d_sd, false))) return rc;
    if (node && (rc = stg.out(node, nt, &d_n, false))) return rc;
    if ((rc = stg.out(end_node, (size_t)B * 2, &d_e, false))) return rc;
    if ((rc = stg.out(speed_s, (size_t)B * st::kCols, &d_ss, false))) return rc;
    if ((rc = stg.out(speed_t, (size_t)B * st::kCols, &d_tt, false))) return rc;
    // heaviest scenes first (emp_st_kernels.h: st_count_kernel); pointless when every block is resident at once
    int* d_order = nullptr;
    if (B > 512) {
        unsigned char* d_key;
        int* d_hist;
        if ((rc = stg.tmp<int>((size_t)B, &d_order))) return rc;
        if ((rc = stg.tmp<unsigned char>((size_t)B, &d_key))) return rc;
        if ((rc = stg.tmp<int>(2 * (size_t)kStKeys, &d_hist, true))) return rc;
        KernelTimer t(ctx, "speed_dp_order");
        hipLaunchKernelGGL(st_count_kernel, grid1(B, 256), dim3(256), 0, ctx->stream, B, max_obs, d_si, d_key, d_hist);
        hipLaunchKernelGGL(st_scatter_kernel, grid1(B, 256), dim3(256), 0, ctx->stream, B, d_key, d_hist, d_hist + kStKeys, d_order);
        EMP_LAUNCH_CHECK(ctx);
    }
    if (B) {
        const StDev d = make_st_dev(p, B, max_obs);
        KernelTimer t(ctx, "speed_dp");
        if (max_obs <= 32)
            hipLaunchKernelGGL(speed_dp_kernel<uint32_t>, dim3(B), dim3(kStBlock), speed_dp_lds_bytes(max_obs), ctx->stream, d,
                               d_si, d_so, d_ti, d_to, d_v, d_c, d_sd, d_n, d_e, d_ss, d_tt, d_order);
        else
            hipLaunchKernelGGL(speed_dp_kernel<uint64_t>, dim3(B), dim3(kStBlock), speed_dp_lds_bytes(max_obs), ctx->stream, d,
                               d_si, d_so, d_ti, d_to, d_v, d_c, d_sd, d_n, d_e, d_ss, d_tt, d_order);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

int emp_st_edge_costs(emp_ctx* ctx, const emp_speed_dp_params* p, int32_t B, int32_t n_edges, int32_t max_obs,
                      const double* edges, const double* s_in, const double* s_out, const double* t_in,
                      const double* t_out, double* total, double* obs, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, p != nullptr, "speed dp params are NULL");
    EMP_REQUIRE(ctx, B >= 0 && B <= 65535 && n_edges >= 0 && max_obs >= 1 && max_obs <= st::kMaxObs,
                "bad sizes (B <= 65535, max_obs in [1, 64])");
    EMP_REQUIRE(ctx, edges && s_in && s_out && t_in && t_out && total, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const size_t n = (size_t)B * max_obs, ne = (size_t)B * n_edges;
    const double *d_e, *d_si, *d_so, *d_ti, *d_to;
    double *d_t, *d_o = nullptr;
    if ((rc = stg.in(edges, ne * 5, &d_e))) return rc;
    if ((rc = stg.in(s_in, n, &d_si))) return rc;
    if ((rc = stg.in(s_out, n, &d_so))) return rc;
    if ((rc = stg.in(t_in, n, &d_ti))) return rc;
    if ((rc = stg.in(t_out, n, &d_to))) return rc;
    if ((rc = stg.out(total, ne, &d_t, false))) return rc;
    if (obs && (rc = stg.out(obs, ne, &d_o, false))) return rc;
    if (ne) {
        const StDev d = make_st_dev(p, B, max_obs);
        dim3 grid((n_edges + 63) / 64, B);
        hipLaunchKernelGGL(st_edge_cost_kernel, grid, dim3(64), 7 * (size_t)max_obs * sizeof(double), ctx->stream, d, n_edges, d_e, d_si, d_so, d_ti, d_to, d_t,
                           d_o);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

int emp_st_collision_cost(emp_ctx* ctx, int32_t n, double w_cost_obs, const double* min_dis, double* cost, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && min_dis && cost, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const double* d_d;
    double* d_c;
    if ((rc = stg.in(min_dis, (size_t)n, &d_d))) return rc;
    if ((rc = stg.out(cost, (size_t)n, &d_c, false))) return rc;
    if (n) {
        hipLaunchKernelGGL(st_collision_cost_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, st::make_pow_base(w_cost_obs), d_d, d_c);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

int emp_speed_start_condition(emp_ctx* ctx, int32_t n, const double* vx, const double* vy, const double* ax, const double* ay,
                              const double* heading, double* s_dot, double* s_dot2, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, n >= 0 && vx && vy && ax && ay && heading && s_dot && s_dot2, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const double *d_vx, *d_vy, *d_ax, *d_ay, *d_h;
    double *d_s1, *d_s2;
    if ((rc = stg.in(vx, (size_t)n, &d_vx))) return rc;
    if ((rc = stg.in(vy, (size_t)n, &d_vy))) return rc;
    if ((rc = stg.in(ax, (size_t)n, &d_ax))) return rc;
    if ((rc = stg.in(ay, (size_t)n, &d_ay))) return rc;
    if ((rc = stg.in(heading, (size_t)n, &d_h))) return rc;
    if ((rc = stg.out(s_dot, (size_t)n, &d_s1, false))) return rc;
    if ((rc = stg.out(s_dot2, (size_t)n, &d_s2, false))) return rc;
    if (n) {
        hipLaunchKernelGGL(st_start_condition_kernel, grid1(n, 64), dim3(64), 0, ctx->stream, n, d_vx, d_vy, d_ax, d_ay, d_h, d_s1, d_s2);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

// ---- S-T speed planning back end (reference speed_planning_test.py:308-620) ---------------------
void emp_speed_qp_params_default(emp_speed_qp_params* p) {
    if (!p) return;
    // ref: speed_QP keyword defaults, speed_planning_test.py:410-411
    p->w_cost_s_dot2 = 10.0;
    p->w_cost_v_ref = 50.0;
    p->w_cost_jerk = 500.0;
    p->reference_speed = 50.0;
}

int emp_speed_convex_space(emp_ctx* ctx, int32_t B, int32_t n_slots, int32_t max_path, double max_lateral_accel,
                           const double* dp_speed_s, const double* dp_speed_t, const double* path_index2s,
                           const double* path_kappa, const int32_t* path_len, const double* s_in, const double* s_out,
                           const double* t_in, const double* t_out, double* s_lb, double* s_ub, double* s_dot_lb,
                           double* s_dot_ub, int32_t* status, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && n_slots >= 1 && max_path >= 1, "bad sizes");
    EMP_REQUIRE(ctx, dp_speed_s && dp_speed_t && path_index2s && path_kappa && path_len && s_in && s_out && t_in && t_out &&
                         s_lb && s_ub && s_dot_lb && s_dot_ub && status, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const double *d_ds, *d_dt, *d_i2s, *d_k, *d_si, *d_so, *d_ti, *d_to;
    const int* d_pl;
    double *d_lb, *d_ub, *d_vlb, *d_vub;
    int* d_st;
    if ((rc = stg.in(dp_speed_s, (size_t)B * stb::kDp, &d_ds))) return rc;
    if ((rc = stg.in(dp_speed_t, (size_t)B * stb::kDp, &d_dt))) return rc;
    if ((rc = stg.in(path_index2s, (size_t)B * max_path, &d_i2s))) return rc;
    if ((rc = stg.in(path_kappa, (size_t)B * max_path, &d_k))) return rc;
    if ((rc = stg.in(path_len, (size_t)B, &d_pl))) return rc;
    if ((rc = stg.in(s_in, (size_t)B * n_slots, &d_si))) return rc;
    if ((rc = stg.in(s_out, (size_t)B * n_slots, &d_so))) return rc;
    if ((rc = stg.in(t_in, (size_t)B * n_slots, &d_ti))) return rc;
    if ((rc = stg.in(t_out, (size_t)B * n_slots, &d_to))) return rc;
    if ((rc = stg.out(s_lb, (size_t)B * stb::kDp, &d_lb, false))) return rc;
    if ((rc = stg.out(s_ub, (size_t)B * stb::kDp, &d_ub, false))) return rc;
    if ((rc = stg.out(s_dot_lb, (size_t)B * stb::kDp, &d_vlb, false))) return rc;
    if ((rc = stg.out(s_dot_ub, (size_t)B * stb::kDp, &d_vub, false))) return rc;
    if ((rc = stg.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        KernelTimer t(ctx, "speed_convex_space");
        hipLaunchKernelGGL(stb::convex_space_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, n_slots, max_path,
                           max_lateral_accel, d_ds, d_dt, d_i2s, d_k, d_pl, d_si, d_so, d_ti, d_to, d_lb, d_ub, d_vlb, d_vub,
                           d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

int emp_speed_qp(emp_ctx* ctx, const emp_speed_qp_params* p, int32_t B, const double* plan_start_s_dot,
                 const double* plan_start_s_dot2, const double* dp_speed_s, const double* dp_speed_t, const double* s_lb,
                 const double* s_ub, const double* s_dot_lb, const double* s_dot_ub, double* qp_s, double* qp_s_dot,
                 double* qp_s_dot2, double* relative_time, int32_t* iters, int32_t* status, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, p != nullptr && B >= 0, "bad argument");
    EMP_REQUIRE(ctx, plan_start_s_dot && plan_start_s_dot2 && dp_speed_s && dp_speed_t && s_lb && s_ub && s_dot_lb && s_dot_ub &&
                         qp_s && qp_s_dot && qp_s_dot2 && relative_time && status, "NULL argument");
    EMP_REQUIRE(ctx, p->w_cost_s_dot2 > 0 && p->w_cost_v_ref > 0 && p->w_cost_jerk >= 0, "weights must be positive");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const double *d_v0, *d_a0, *d_ds, *d_dt, *d_lb, *d_ub, *d_vlb, *d_vub;
    double *d_qs, *d_qv, *d_qa, *d_qt;
    int *d_it, *d_st;
    if ((rc = stg.in(plan_start_s_dot, (size_t)B, &d_v0))) return rc;
    if ((rc = stg.in(plan_start_s_dot2, (size_t)B, &d_a0))) return rc;
    if ((rc = stg.in(dp_speed_s, (size_t)B * stb::kDp, &d_ds))) return rc;
    if ((rc = stg.in(dp_speed_t, (size_t)B * stb::kDp, &d_dt))) return rc;
    if ((rc = stg.in(s_lb, (size_t)B * stb::kDp, &d_lb))) return rc;
    if ((rc = stg.in(s_ub, (size_t)B * stb::kDp, &d_ub))) return rc;
    if ((rc = stg.in(s_dot_lb, (size_t)B * stb::kDp, &d_vlb))) return rc;
    if ((rc = stg.in(s_dot_ub, (size_t)B * stb::kDp, &d_vub))) return rc;
    if ((rc = stg.out(qp_s, (size_t)B * stb::kQp, &d_qs, false))) return rc;
    if ((rc = stg.out(qp_s_dot, (size_t)B * stb::kQp, &d_qv, false))) return rc;
    if ((rc = stg.out(qp_s_dot2, (size_t)B * stb::kQp, &d_qa, false))) return rc;
    if ((rc = stg.out(relative_time, (size_t)B * stb::kQp, &d_qt, false))) return rc;
    if ((rc = stg.out(iters, (size_t)B, &d_it, false))) return rc;
    if ((rc = stg.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        const stb::SpeedQpParams prm{p->w_cost_s_dot2, p->w_cost_v_ref, p->w_cost_jerk, p->reference_speed};
        const size_t lds = 2 * (size_t)(stb::speed_qp_words(stb::kQp) + 1) * sizeof(double);
        KernelTimer t(ctx, "speed_qp");
        hipLaunchKernelGGL(stb::speed_qp_kernel<32>, dim3((B + 1) / 2), dim3(64), lds, ctx->stream, B, prm, d_v0, d_a0, d_ds,
                           d_dt, d_lb, d_ub, d_vlb, d_vub, d_qs, d_qv, d_qa, d_qt, d_it, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

int emp_speed_increase_points(emp_ctx* ctx, int32_t B, const double* s_init, const double* s_dot_init,
                              const double* s_dot2_init, const double* relative_time_init, double* s, double* s_dot,
                              double* s_dot2, double* relative_time, int32_t* status, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && s_init && s_dot_init && s_dot2_init && relative_time_init && s && s_dot && s_dot2 &&
                         relative_time && status, "bad argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const double *d_qs, *d_qv, *d_qa, *d_qt;
    double *d_s, *d_v, *d_a, *d_t;
    int* d_st;
    if ((rc = stg.in(s_init, (size_t)B * stb::kQp, &d_qs))) return rc;
    if ((rc = stg.in(s_dot_init, (size_t)B * stb::kQp, &d_qv))) return rc;
    if ((rc = stg.in(s_dot2_init, (size_t)B * stb::kQp, &d_qa))) return rc;
    if ((rc = stg.in(relative_time_init, (size_t)B * stb::kQp, &d_qt))) return rc;
    if ((rc = stg.out(s, (size_t)B * stb::kDense, &d_s, false))) return rc;
    if ((rc = stg.out(s_dot, (size_t)B * stb::kDense, &d_v, false))) return rc;
    if ((rc = stg.out(s_dot2, (size_t)B * stb::kDense, &d_a, false))) return rc;
    if ((rc = stg.out(relative_time, (size_t)B * stb::kDense, &d_t, false))) return rc;
    if ((rc = stg.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        KernelTimer t(ctx, "speed_increase_points");
        hipLaunchKernelGGL(stb::densify_kernel, dim3(B), dim3(64), 0, ctx->stream, B, d_qs, d_qv, d_qa, d_qt, d_s, d_v, d_a, d_t,
                           d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

int emp_path_speed_merge(emp_ctx* ctx, int32_t B, int32_t max_path, const double* s, const double* s_dot,
                         const double* s_dot2, const double* relative_time, const double* current_time,
                         const double* path_s, const double* x_init, const double* y_init, const double* heading_init,
                         const double* kappa_init, const int32_t* n_init, double* trajectory, int32_t* status,
                         emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, B >= 0 && max_path >= 1, "bad sizes");
    EMP_REQUIRE(ctx, s && s_dot && s_dot2 && relative_time && current_time && path_s && x_init && y_init && heading_init &&
                         kappa_init && n_init && trajectory && status, "NULL argument");
    const size_t lds = (size_t)5 * max_path * sizeof(double);
    EMP_REQUIRE(ctx, lds <= 64 * 1024, "path too long for the LDS-resident merge kernel");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage stg(ctx, where);
    int rc;
    const double *d_s, *d_v, *d_a, *d_t, *d_now, *d_ps, *d_x, *d_y, *d_h, *d_k;
    const int* d_n;
    double* d_out;
    int* d_st;
    if ((rc = stg.in(s, (size_t)B * stb::kDense, &d_s))) return rc;
    if ((rc = stg.in(s_dot, (size_t)B * stb::kDense, &d_v))) return rc;
    if ((rc = stg.in(s_dot2, (size_t)B * stb::kDense, &d_a))) return rc;
    if ((rc = stg.in(relative_time, (size_t)B * stb::kDense, &d_t))) return rc;
    if ((rc = stg.in(current_time, (size_t)B, &d_now))) return rc;
    if ((rc = stg.in(path_s, (size_t)B * max_path, &d_ps))) return rc;
    if ((rc = stg.in(x_init, (size_t)B * max_path, &d_x))) return rc;
    if ((rc = stg.in(y_init, (size_t)B * max_path, &d_y))) return rc;
    if ((rc = stg.in(heading_init, (size_t)B * max_path, &d_h))) return rc;
    if ((rc = stg.in(kappa_init, (size_t)B * max_path, &d_k))) return rc;
    if ((rc = stg.in(n_init, (size_t)B, &d_n))) return rc;
    if ((rc = stg.out(trajectory, (size_t)B * 7 * stb::kDense, &d_out, false))) return rc;
    if ((rc = stg.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        if (lds > 48 * 1024)
            EMP_HIP(ctx, hipFuncSetAttribute((const void*)stb::merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KernelTimer t(ctx, "path_speed_merge");
        hipLaunchKernelGGL(stb::merge_kernel, dim3(B), dim3(64), lds, ctx->stream, B, max_path, d_s, d_v, d_a, d_t, d_now, d_ps,
                           d_x, d_y, d_h, d_k, d_n, d_out, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return stg.finish();
}

}  // extern "C"

// ---- lateral MPC controller (reference controller/controller.py:65-337) -------------------------
extern "C" {

void emp_mpc_params_default(emp_mpc_params* p) {
    if (!p) return;
    // the driver's tuple (1.015, 2.910 - 1.015, 1412, -148970, -82204, 1537) (ref test_9.py:316) as the controller
    // unpacks it: (a, b, Cf, Cr, m, Iz) = vehicle_para (ref controller.py:132)
    p->a = 1.015;
    p->b = 2.910 - 1.015;
    p->Cf = 1412.0;
    p->Cr = -148970.0;
    p->m = -82204.0;
    p->Iz = 1537.0;
    const double q[4] = {250.0, 1.0, 50.0, 1.0};
    for (int i = 0; i < 4; ++i) {
        p->q_diag[i] = q[i];
        p->f_diag[i] = 1.0;
    }
    p->r = 1.0;
}

int emp_mpc_lateral(emp_ctx* ctx, const emp_mpc_params* p, int32_t B, int32_t max_path, const double* target_path,
                    const int32_t* n_path, const double* state, const double* vx, const int32_t* min_index,
                    double* steer, double* u, double* e_rr, double* k_r, int32_t* min_index_out, double* pre_pro,
                    double* H, double* f, int32_t* iters, int32_t* status, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, p && B >= 0 && max_path >= 1, "bad sizes");
    EMP_REQUIRE(ctx, target_path && n_path && state && vx && min_index && steer && min_index_out && status, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_path, *d_state, *d_vx;
    const int *d_np, *d_mi;
    double *d_steer, *d_u = nullptr, *d_e = nullptr, *d_k = nullptr, *d_pp = nullptr, *d_H = nullptr, *d_f = nullptr;
    int *d_mo, *d_it = nullptr, *d_st;
    if ((rc = st.in(target_path, (size_t)B * max_path * 4, &d_path))) return rc;
    if ((rc = st.in(n_path, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(state, (size_t)B * 5, &d_state))) return rc;
    if ((rc = st.in(vx, (size_t)B, &d_vx))) return rc;
    if ((rc = st.in(min_index, (size_t)B, &d_mi))) return rc;
    if ((rc = st.out(steer, (size_t)B, &d_steer, false))) return rc;
    if (u && (rc = st.out(u, (size_t)B * mpc::kNu, &d_u, false))) return rc;
    if (e_rr && (rc = st.out(e_rr, (size_t)B * 4, &d_e, false))) return rc;
    if (k_r && (rc = st.out(k_r, (size_t)B, &d_k, false))) return rc;
    if ((rc = st.out(min_index_out, (size_t)B, &d_mo, false))) return rc;
    if (pre_pro && (rc = st.out(pre_pro, (size_t)B * 4, &d_pp, false))) return rc;
    if (H && (rc = st.out(H, (size_t)B * mpc::kNu * mpc::kNu, &d_H, false))) return rc;
    if (f && (rc = st.out(f, (size_t)B * mpc::kNu, &d_f, false))) return rc;
    if (iters && (rc = st.out(iters, (size_t)B, &d_it, false))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        mpc::Params prm;
        prm.a = p->a; prm.b = p->b; prm.Cf = p->Cf; prm.Cr = p->Cr; prm.m = p->m; prm.Iz = p->Iz;
        for (int i = 0; i < 4; ++i) {
            prm.q[i] = p->q_diag[i];
            prm.f[i] = p->f_diag[i];
        }
        prm.r = p->r;
        KernelTimer t(ctx, "mpc_lateral");
        hipLaunchKernelGGL(mpc::mpc_lateral_kernel, dim3((B + mpc::kGroupsPerWave - 1) / mpc::kGroupsPerWave), dim3(64), 0,
                           ctx->stream, B, max_path, prm, d_path, d_np, d_state, d_vx, d_mi, d_steer, d_u, d_e, d_k, d_mo, d_pp,
                           d_H, d_f, d_it, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

void emp_lqr_params_default(emp_mpc_params* p) {
    if (!p) return;
    emp_mpc_params_default(p);
    p->q_diag[0] = 200.0;                                    // ref controller.py:593-597
}

int emp_lqr_lateral(emp_ctx* ctx, const emp_mpc_params* p, int32_t B, int32_t max_path, const double* target_path,
                    const int32_t* n_path, const double* state, const double* vx, const int32_t* min_index,
                    double* steer, double* K, double* e_rr, double* k_r, int32_t* min_index_out, double* pre_pro,
                    int32_t* sweeps, int32_t* status, emp_mem where) {
    using namespace emp;
    EMP_REQUIRE(ctx, ctx != nullptr, "ctx is NULL");
    EMP_REQUIRE(ctx, p && B >= 0 && max_path >= 1, "bad sizes");
    EMP_REQUIRE(ctx, target_path && n_path && state && vx && min_index && steer && min_index_out && status, "NULL argument");
    EMP_HIP(ctx, hipSetDevice(ctx->device));
    Stage st(ctx, where);
    int rc;
    const double *d_path, *d_state, *d_vx;
    const int *d_np, *d_mi;
    double *d_steer, *d_K = nullptr, *d_e = nullptr, *d_k = nullptr, *d_pp = nullptr;
    int *d_mo, *d_sw = nullptr, *d_st;
    if ((rc = st.in(target_path, (size_t)B * max_path * 4, &d_path))) return rc;
    if ((rc = st.in(n_path, (size_t)B, &d_np))) return rc;
    if ((rc = st.in(state, (size_t)B * 5, &d_state))) return rc;
    if ((rc = st.in(vx, (size_t)B, &d_vx))) return rc;
    if ((rc = st.in(min_index, (size_t)B, &d_mi))) return rc;
    if ((rc = st.out(steer, (size_t)B, &d_steer, false))) return rc;
    if (K && (rc = st.out(K, (size_t)B * 4, &d_K, false))) return rc;
    if (e_rr && (rc = st.out(e_rr, (size_t)B * 4, &d_e, false))) return rc;
    if (k_r && (rc = st.out(k_r, (size_t)B, &d_k, false))) return rc;
    if ((rc = st.out(min_index_out, (size_t)B, &d_mo, false))) return rc;
    if (pre_pro && (rc = st.out(pre_pro, (size_t)B * 4, &d_pp, false))) return rc;
    if (sweeps && (rc = st.out(sweeps, (size_t)B, &d_sw, false))) return rc;
    if ((rc = st.out(status, (size_t)B, &d_st, false))) return rc;
    if (B) {
        mpc::Params prm;
        prm.a = p->a; prm.b = p->b; prm.Cf = p->Cf; prm.Cr = p->Cr; prm.m = p->m; prm.Iz = p->Iz;
        for (int i = 0; i < 4; ++i) {
            prm.q[i] = p->q_diag[i];
            prm.f[i] = p->f_diag[i];
        }
        prm.r = p->r;
        KernelTimer t(ctx, "lqr_lateral");
        hipLaunchKernelGGL(lqr::lqr_lateral_kernel, grid1(B, 64), dim3(64), 0, ctx->stream, B, max_path, prm, d_path, d_np,
                           d_state, d_vx, d_mi, d_steer, d_K, d_e, d_k, d_mo, d_pp, d_sw, d_st);
        EMP_LAUNCH_CHECK(ctx);
    }
    return st.finish();
}

}  // extern "C"
