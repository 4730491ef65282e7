// emp_context.h - context object behind the C-ABI: device, stream, scratch pool, host staging, timing.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/emplanner.h"

// an atomic that a std::vector element may hold (copyable: the copy is a plain load / store, made only while no other thread
// can be looking - emp_set_pipeline resizing the lanes)
template <typename T>
struct Shared {
    std::atomic<T> v;
    Shared(T x = T()) : v(x) {}
    Shared(const Shared& o) : v(o.v.load()) {}
    Shared& operator=(const Shared& o) { v.store(o.v.load()); return *this; }
    Shared& operator=(T x) { v.store(x, std::memory_order_release); return *this; }
    operator T() const { return v.load(std::memory_order_acquire); }
};

struct emp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // grow-only pool of device buffers, handed out in call order and recycled by the next call
    struct Buf {
        void* p = nullptr;
        size_t bytes = 0;
    };
    std::vector<Buf> pool;
    size_t cursor = 0;
    // persistent named scratch (survives across the staged buffers of one call)
    std::map<std::string, Buf> named;
    std::string timing_filter;   // non-empty: only this kernel name is bracketed by events
    // the pair table of the edge-cost kernel and the lattice parameters it was built for (emp_api.hip: dp_pair_table);
    // one per stream that runs edge kernels (key: active_lane), so that a rebuild never races a reader on another stream
    struct PairTable {
        Buf buf;
        double key[8] = {0};
        bool valid = false;
    };
    std::map<int, PairTable> pair_tables;
    // per-kernel timing
    bool timing = false;
    struct Ev {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;   // one pair per launch since timing was enabled
        size_t used = 0;
    };
    std::map<std::string, Ev> events;
    int cu_count = 0;
    // Several batches in flight (emp_set_pipeline), two forms.
    // STAGED (mode 1, two batches): a cycle is a front stage (projection, edge costs, sweep) on `stream` and a back stage
    // (densified DP path, path QP, Cartesian tail) on `back_stream`; the back stage of call k overlaps the front stage of
    // call k+1.  The two calls in flight alternate between lanes[0] and lanes[1], of which only the pool of temporaries
    // and the events are used.  The front stages are serial, so one edge tensor serves every call (it stays in the
    // Infinity Cache) and the sweep overlaps nothing but the tail of the back stage before it.
    // LANES (mode n >= 2, n batches): call k runs WHOLE on the stream of lanes[k mod n], with that lane's pool, edge
    // tensor and pair table, behind everything queued on `stream` when it was issued; the dispatcher overlaps the kernels
    // of n consecutive cycles wherever it finds room.
    // ev_in: recorded on `stream` by a LANES call, its lane waits for it.  ev_front: end of a STAGED front stage.
    // ev_tail: LANES, the lane stream's tail when the lane is taken again - the main stream waits for it, so that memory
    // the caller releases once call k + n is issued is not handed to a call on ANOTHER lane while call k, or a consumer
    // queued behind it on the lane, still runs (the lanes are ordered behind the main stream, not against each other).
    // ev_done: end of the lane's latest cycle - what the next user of the lane's pool and every other entry point wait for.
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t ev_in = nullptr, ev_front = nullptr, ev_done = nullptr, ev_tail = nullptr;
        hipEvent_t ev_host = nullptr;   // EMP_HOST_PINNED cycles: the lane's latest cycle's outputs have reached the caller's host arrays
        // `ticket` and `host_valid` are the two fields emp_wait_ticket reads from ANOTHER thread while a call is in progress:
        // atomics, and a call that takes the lane over changes them only AFTER its own host-side wait for the previous
        // occupant's outputs (emp_plan_cycle) - until then the lane still answers for the previous ticket
        Shared<bool> host_valid{false};
        Shared<uint64_t> ticket{0};     // emp_cycle_ticket of the lane's latest cycle
        hipEvent_t ev_qp = nullptr;     // STAGED: end of the cycle's path QP on the back stream (EMP_OPT_SWEEP_EXCLUSIVE = 2)
        hipEvent_t ev_enrich = nullptr; // STAGED: end of the cycle's densification kernel on the back stream (EMP_OPT_EDGE_AFTER_ENRICH)
        hipEvent_t ev_edge = nullptr;   // LANES: end of the cycle's edge-cost kernel (EMP_OPT_LANE_EDGE_ORDER)
        bool done_valid = false, qp_valid = false, enrich_valid = false;
        std::vector<Buf> pool;
    };
    std::vector<Lane> lanes;            // created on demand, kept until emp_destroy
    hipStream_t back_stream = nullptr;  // STAGED: the back stages (highest queue priority)
    // EMP_HOST_PINNED (emp_plan_cycle): inputs go host -> device on copy_stream while the previous call computes, outputs device ->
    // host on d2h_stream behind the cycle's last kernel - neither ever sits on a queue that carries kernels.  Created on first use.
    hipStream_t copy_stream = nullptr, d2h_stream = nullptr;
    hipEvent_t ev_h2d = nullptr;        // the latest call's inputs have arrived
    hipEvent_t ev_host_last = nullptr;  // non-pipelined pinned call: outputs have reached the host
    struct Pinned {
        void* p;
        size_t bytes;
    };
    std::vector<Pinned> pinned;         // emp_host_alloc allocations still alive (freed by emp_destroy)
    // Small EMP_HOST calls (round 6): ONE page-locked arena and one device arena per direction.  The arrays of a synchronous call
    // with host pointers are packed into the input arena by the host and cross PCIe as one copy; the outputs come back as one
    // copy and are unpacked by the host (Stage).  A copy command costs 4-14 us whatever its size and a call of the reference's
    // function surface has three to twenty small arrays: emp_lmin_lmax went from 100 to ~60 us (profiles/r06_call_cost_probe.txt).
    static constexpr size_t kArena = 256 * 1024, kArenaArray = 64 * 1024;
    char* arena_h_in = nullptr;         // page-locked, kArena bytes each
    char* arena_h_out = nullptr;
    char* arena_d_in = nullptr;         // device, kArena bytes each
    char* arena_d_out = nullptr;
    bool arena_failed = false;          // an allocation failed once: the per-array path from then on
    // STAGED: the event the front stage's LAST kernel (the sweep) is asked to signal when it completes (hipExtLaunchKernelGGL's
    // stop event: no marker packet behind the kernel), and the event that launch did attach - its own timing event when the
    // kernel is being timed, else front_stop, else nullptr (launchers that attach nothing: the caller records an event).
    hipEvent_t front_stop = nullptr, front_attached = nullptr;
    // The planning cycle leaves the DP backtrack to the densification kernel (dp_sweep_kernel, BT == false): the caller offers
    // the two buffers, the sweep's launcher sets bt_deferred when it used them (compiled row counts only).
    unsigned char* bt_pre = nullptr;
    int* bt_term = nullptr;
    bool bt_deferred = false;
    int pipe_mode = 0;                  // 0 off, 1 STAGED, n >= 2 LANES with n lanes
    int lane = 0;                       // lane of the latest pipelined cycle call
    uint64_t cycle_calls = 0;           // pipelined emp_plan_cycle calls issued so far (emp_cycle_ticket)
    int active_lane = -1;               // LANES: the lane whose stream and pool stand in for `stream` / `pool` right now
    bool fence = true;                  // emp_set_fence: other entry points wait for the cycles in flight
    // emp_set_option (include/emplanner.h): per-context tuning / A-B / test-hook values; the library reads no environment
    int32_t opt[EMP_OPT_COUNT] = {0, 0, 0, 0, 0, 0, /* EDGE_AFTER_ENRICH */ 1, /* LANE_EDGE_ORDER */ 2, 0, 0, 0, /* FOREIGN_STREAMS */ 1};
    int auto_queues = 0, auto_streams = 0;   // what the latest emp_set_pipeline(EMP_PIPELINE_AUTO) saw (emp_pipeline_form)
    // EMP_OPT_CYCLE_GRAPH: the launches of one emp_plan_cycle call as an executable graph, the call signature it belongs to, how
    // often that signature has been seen in a row, and the allocation count (grow_buffer) it was captured under
    hipGraphExec_t cycle_graph = nullptr;
    std::vector<unsigned long long> cycle_graph_key, cycle_seen_key;
    int cycle_seen = 0;
    long long cycle_graph_replays = 0;
    unsigned long long alloc_gen = 0, cycle_graph_gen = 0;
    bool capturing = false;             // launchers avoid what a stream capture cannot record (hipExtLaunchKernelGGL)
    hipEvent_t edge_wait = nullptr;     // EMP_OPT_EDGE_AFTER_ENRICH: what the next edge-cost launch waits for on its stream
    hipEvent_t lane_edge_done = nullptr; // EMP_OPT_LANE_EDGE_ORDER: recorded behind the latest edge-cost launch of a lane-mode call (a lane's ev_edge)
    // STAGED: an event the next densification / path-QP launch is asked to signal from its own dispatch (hipExtLaunchKernelGGL's
    // stop event) instead of a marker packet behind it - a marker idles the back queue ~6 us, twice per step; `stop_attached`
    // says whether the launcher did (it does not while the kernel carries timing events)
    hipEvent_t attach_stop = nullptr;
    bool stop_attached = false;
    hipEvent_t sweep_marker = nullptr;  // EMP_OPT_SWEEP_EXCLUSIVE: recorded on the front stream behind the sweep (emp_api.hip)
    // EMP_OPT_SWEEP_EXCLUSIVE: the event the next sweep launch waits for on its own stream (the previous call's back stage)
    hipEvent_t sweep_wait = nullptr;
    // EMP_OPT_SWEEP_CLOCK_PROBE: a ring of kProbeSlots launches x [tiles][4] ticks (shader-clock begin / end, reference
    // begin / end); probe_launches counts the launches recorded since the option was last switched on
    static constexpr int kProbeSlots = 32;
    Buf clock_probe;
    Buf edge_probe;                     // EMP_OPT_EDGE_CLOCK_PROBE: [wavefronts of the latest edge launch][2] reference ticks
    long edge_probe_waves = 0;
    hipEvent_t edge_probe_done = nullptr;
    int clock_probe_tiles = 0;
    long probe_launches = 0;
    hipEvent_t clock_probe_done = nullptr;
    bool pipelined() const { return pipe_mode != 0; }
    // STAGED rotates kStagedPools pools of temporaries although only two calls overlap on the GPU: call k reuses the pool
    // of call k - 4 and the HOST waits for that call's back stage (long finished unless the host runs more than four
    // calls ahead, which this also bounds) - no barrier packet on the queue of the front stages.  With two pools the
    // stream had to wait for call k - 2 on the GPU: a cross-queue dependency in front of every projection kernel, ~11 us
    // of the command processor's time per 0.29 ms step on the queue that is the step's critical path.
    static constexpr int kStagedPools = 4;
    int lanes_in_use() const { return pipe_mode == 1 ? kStagedPools : pipe_mode; }
    hipStream_t result_stream() const {
        return pipe_mode == 0 ? stream : pipe_mode == 1 ? back_stream : lanes[lane].stream;
    }
};

namespace emp {

extern thread_local std::string g_create_error;

inline int fail(emp_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define EMP_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError(); /* the runtime keeps the error for the next hipGetLastError(): without this, \
                                        the launch check of the NEXT call would report this call's failure */ \
            return emp::fail((ctx), e_ == hipErrorOutOfMemory ? EMP_ERR_NOMEM : EMP_ERR_HIP,       \
                             std::string(#call) + ": " + hipGetErrorString(e_));                   \
        }                                                                                          \
    } while (0)

#define EMP_REQUIRE(ctx, cond, msg)                                                               \
    do {                                                                                           \
        if (!(cond)) return emp::fail((ctx), EMP_ERR_INVALID, std::string(msg));                   \
    } while (0)

// Waits for everything queued on the context's streams (the main one and every lane).
inline int sync_all(emp_ctx* ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (auto& ln : ctx->lanes) {
        const hipError_t e2 = ln.stream ? hipStreamSynchronize(ln.stream) : hipSuccess;
        if (e == hipSuccess) e = e2;
    }
    for (hipStream_t st : {ctx->back_stream, ctx->copy_stream, ctx->d2h_stream}) {
        if (!st) continue;
        const hipError_t e2 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e2;
    }
    return (int)e;
}

// Grow-only device buffer with 25 % headroom.  A buffer is replaced only once nothing queued on any of the context's
// streams can still touch it (hipFree's own implicit synchronisation is not relied upon).
inline int grow_buffer(emp_ctx* ctx, emp_ctx::Buf& b, size_t bytes) {
    if (b.bytes >= bytes) return EMP_OK;
    if (b.p) {
        EMP_HIP(ctx, (hipError_t)sync_all(ctx));
        EMP_HIP(ctx, hipFree(b.p));
    }
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = bytes + bytes / 4;
    EMP_HIP(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    ++ctx->alloc_gen;
    return EMP_OK;
}

// device scratch from the per-call pool
inline int pool_get(emp_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 8;
    if (ctx->cursor == ctx->pool.size()) ctx->pool.push_back({});
    emp_ctx::Buf& b = ctx->pool[ctx->cursor++];
    const int rc = grow_buffer(ctx, b, bytes);
    if (rc) return rc;
    *out = b.p;
    return EMP_OK;
}

// Staging of one call's arguments.  For EMP_DEVICE pointers pass through; for EMP_HOST inputs are copied
// to pool buffers and outputs are copied back in finish().
class Stage {
  public:
    // in_cycle: the pipelined emp_plan_cycle (and a launch placed on its lane) orders itself; every OTHER call in
    // pipelined mode first lets the main stream wait for the cycles still in flight, so that it may consume a cycle's
    // outputs as before
    // async_host (emp_plan_cycle with EMP_HOST_PINNED): the caller's arrays are page-locked (emp_host_alloc) - inputs are copied
    // on ctx->copy_stream (inputs_ready() orders the compute stream behind them), outputs on ctx->d2h_stream behind `after`
    // (finish_async), and nothing blocks the host
    Stage(emp_ctx* c, emp_mem where, bool in_cycle = false, bool async_host = false)
        : ctx_(c), dev_(where == EMP_DEVICE), async_(async_host && where != EMP_DEVICE) {
        c->cursor = 0;
        arena_ = !dev_ && !async_ && where == EMP_HOST && !c->capturing;
        if (!in_cycle && c->pipelined() && c->fence)
            for (auto& ln : c->lanes)
                if (ln.done_valid) (void)hipStreamWaitEvent(c->stream, ln.ev_done, 0);
    }

    template <typename T>
    int in(const T* host, size_t n, const T** out) {
        if (host == nullptr) { *out = nullptr; return EMP_OK; }
        if (dev_) { *out = host; return EMP_OK; }
        if (async_) {            // deferred: inputs_ready() places all inputs of the call at once and fills *out then
            ins_.push_back({(void*)host, nullptr, n * sizeof(T), (void**)out});
            *out = reinterpret_cast<const T*>(kPending);
            return EMP_OK;
        }
        if (arena_ && arena_take(n * sizeof(T), &in_used_)) {      // packed: one copy for all small inputs (flush_inputs)
            const size_t off = in_used_ - arena_round(n * sizeof(T));
            if (n) memcpy(ctx_->arena_h_in + off, host, n * sizeof(T));
            *out = (const T*)(ctx_->arena_d_in + off);
            return EMP_OK;
        }
        void* d = nullptr;
        int rc = pool_get(ctx_, n * sizeof(T), &d);
        if (rc) return rc;
        if (n) EMP_HIP(ctx_, hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, ctx_->stream));
        *out = (const T*)d;
        return EMP_OK;
    }
    // The packed inputs gathered so far go to the device: ONE copy on the context's stream.  Called by every out() and tmp() -
    // each entry point declares its outputs and temporaries behind its inputs and in front of its first launch - and by finish(),
    // which refuses a call that launched with inputs still pending (a new entry point that breaks the order fails loudly).
    int flush_inputs() {
        if (in_used_ > in_sent_) {
            EMP_HIP(ctx_, hipMemcpyAsync(ctx_->arena_d_in + in_sent_, ctx_->arena_h_in + in_sent_, in_used_ - in_sent_,
                                         hipMemcpyHostToDevice, ctx_->stream));
            in_sent_ = in_used_;
        }
        return EMP_OK;
    }
    // async_host: every input of the call is known.  Arrays that lie side by side in host memory (a HostRing slot is ONE
    // page-locked block: api.py) get one device block at the same offsets and cross PCIe as ONE copy - a copy command costs
    // 10-20 us of host and engine time whatever its size, and a call has nine inputs; others are copied one by one.  Then the
    // compute stream waits for the copy stream.
    int inputs_ready() {
        if (!async_) return EMP_OK;
        int rc = place(ins_, true);
        if (rc) return rc;
        EMP_HIP(ctx_, hipEventRecord(ctx_->ev_h2d, ctx_->copy_stream));
        EMP_HIP(ctx_, hipStreamWaitEvent(ctx_->stream, ctx_->ev_h2d, 0));
        return EMP_OK;
    }
    // async_host: every output of the call is known - device buffers for them, one block where the host arrays are one block
    int outputs_ready() {
        if (!async_) return EMP_OK;
        return place(backs_, false);
    }
    // async_host: copy the outputs back on the d2h stream once `after` (the cycle's completion event) has fired, then signal `done`
    int finish_async(hipEvent_t after, hipEvent_t done) {
        EMP_HIP(ctx_, hipStreamWaitEvent(ctx_->d2h_stream, after, 0));
        if (block_out_.bytes) {
            EMP_HIP(ctx_, hipMemcpyAsync(block_out_.host, block_out_.dev, block_out_.bytes, hipMemcpyDeviceToHost, ctx_->d2h_stream));
        } else {
            for (auto& b : backs_)
                if (b.bytes) EMP_HIP(ctx_, hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, ctx_->d2h_stream));
        }
        EMP_HIP(ctx_, hipEventRecord(done, ctx_->d2h_stream));
        return EMP_OK;
    }
    bool async_host() const { return async_; }
    // Outputs are zero-filled on the context's stream before the kernels run, so padding beyond a scene's
    // length reads as 0 in both memory spaces (pass zero=false for arrays the kernels fully overwrite).
    template <typename T>
    int out(T* host, size_t n, T** outp, bool zero = true) {
        if (host == nullptr) { *outp = nullptr; return EMP_OK; }
        T* d = host;
        if (async_) {            // deferred: outputs_ready()
            backs_.push_back({host, nullptr, n * sizeof(T), (void**)outp});
            *outp = reinterpret_cast<T*>(kPending);
            return EMP_OK;
        }
        if (!dev_) {
            int rcf = flush_inputs();
            if (rcf) return rcf;
            if (arena_ && arena_take(n * sizeof(T), &out_used_)) {       // packed: one zero fill, one copy back (finish)
                const size_t off = out_used_ - arena_round(n * sizeof(T));
                if (!out_zeroed_) {      // the whole output arena once per call instead of a memset per array
                    EMP_HIP(ctx_, hipMemsetAsync(ctx_->arena_d_out, 0, emp_ctx::kArena, ctx_->stream));
                    out_zeroed_ = true;
                }
                arena_backs_.push_back({host, nullptr, n * sizeof(T), nullptr});
                arena_offs_.push_back(off);
                *outp = (T*)(ctx_->arena_d_out + off);
                return EMP_OK;
            }
            void* v = nullptr;
            int rc = pool_get(ctx_, n * sizeof(T), &v);
            if (rc) return rc;
            d = (T*)v;
            backs_.push_back({host, d, n * sizeof(T), nullptr});
        }
        if (zero && n) EMP_HIP(ctx_, hipMemsetAsync(d, 0, n * sizeof(T), ctx_->stream));
        *outp = d;
        return EMP_OK;
    }
    // device-only temporary
    template <typename T>
    int tmp(size_t n, T** outp, bool zero = false) {
        if (!dev_ && !async_) {
            int rcf = flush_inputs();
            if (rcf) return rcf;
        }
        void* v = nullptr;
        int rc = pool_get(ctx_, n * sizeof(T), &v);
        if (rc) return rc;
        if (zero && n) EMP_HIP(ctx_, hipMemsetAsync(v, 0, n * sizeof(T), ctx_->stream));
        *outp = (T*)v;
        return EMP_OK;
    }
    int finish() {
        if (in_used_ > in_sent_)      // (cannot happen with the entry points as they are: see flush_inputs)
            return emp::fail(ctx_, EMP_ERR_INVALID, "internal: packed inputs were never sent (an entry point launched before declaring its outputs)");
        for (auto& b : backs_)
            if (b.bytes) EMP_HIP(ctx_, hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, ctx_->stream));
        if (out_used_) EMP_HIP(ctx_, hipMemcpyAsync(ctx_->arena_h_out, ctx_->arena_d_out, out_used_, hipMemcpyDeviceToHost, ctx_->stream));
        if (!dev_) EMP_HIP(ctx_, hipStreamSynchronize(ctx_->stream));
        for (size_t i = 0; i < arena_backs_.size(); ++i)
            if (arena_backs_[i].bytes) memcpy(arena_backs_[i].host, ctx_->arena_h_out + arena_offs_[i], arena_backs_[i].bytes);
        return EMP_OK;
    }
    bool on_device() const { return dev_; }

  private:
    struct Back {
        void* host;
        void* dev;
        size_t bytes;
        void** slot;       // async_host: where the device pointer goes once it is known
    };
    static constexpr uintptr_t kPending = 8;     // non-null placeholder of a deferred pointer (never dereferenced)
    // device memory for a call's arrays (async_host).  ONE device block with the host offsets - and one PCIe copy per direction -
    // only where that provably touches nothing but the call's own arrays: all of them inside ONE emp_host_alloc allocation of
    // this context, with at most kBlockGap bytes (alignment padding) between neighbours.  Outputs moved as a block overwrite
    // that padding (include/emplanner.h, EMP_HOST_PINNED).  Everything else - arrays allocated one by one, a smaller batch at
    // the head of a bigger slot, foreign data carved between two outputs - is copied array by array.
    static constexpr size_t kBlockGap = 512;
    bool one_block(const std::vector<Back>& v, uintptr_t* lo_out, size_t* span_out) const {
        if (v.size() < 2) return false;
        std::vector<std::pair<uintptr_t, size_t>> a;
        for (auto& b : v)
            if (b.bytes) a.push_back({(uintptr_t)b.host, b.bytes});
        if (a.size() < 2) return false;
        std::sort(a.begin(), a.end());
        const uintptr_t lo = a.front().first;
        uintptr_t end = lo;
        for (auto& x : a) {
            if (x.first < end || x.first - end > kBlockGap) return false;      // overlapping, or more than padding in between
            end = x.first + x.second;
        }
        for (auto& al : ctx_->pinned)
            if (lo >= (uintptr_t)al.p && end <= (uintptr_t)al.p + al.bytes) {
                *lo_out = lo;
                *span_out = end - lo;
                return true;
            }
        return false;
    }
    int place(std::vector<Back>& v, bool inputs) {
        if (v.empty()) return EMP_OK;
        uintptr_t lo = 0;
        size_t span = 0;
        if (one_block(v, &lo, &span)) {
            void* d = nullptr;
            int rc = pool_get(ctx_, span, &d);
            if (rc) return rc;
            for (auto& b : v) {
                b.dev = b.bytes ? (char*)d + ((uintptr_t)b.host - lo) : d;
                *b.slot = b.dev;
            }
            if (inputs) {
                EMP_HIP(ctx_, hipMemcpyAsync(d, (void*)lo, span, hipMemcpyHostToDevice, ctx_->copy_stream));
            } else {
                block_out_ = {(void*)lo, d, span, nullptr};
            }
            return EMP_OK;
        }
        for (auto& b : v) {
            void* d = nullptr;
            int rc = pool_get(ctx_, b.bytes, &d);
            if (rc) return rc;
            b.dev = d;
            *b.slot = d;
            if (inputs && b.bytes) EMP_HIP(ctx_, hipMemcpyAsync(d, b.host, b.bytes, hipMemcpyHostToDevice, ctx_->copy_stream));
        }
        return EMP_OK;
    }
    static size_t arena_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    // room for `bytes` in an arena whose fill mark is *used?  Creates the four arenas on first use; false = take the per-array path
    // (an array of more than kArenaArray bytes - a host memcpy of that size costs more than the copy command it saves - or a full arena)
    bool arena_take(size_t bytes, size_t* used) {
        if (bytes > emp_ctx::kArenaArray || *used + arena_round(bytes) > emp_ctx::kArena || ctx_->arena_failed) return false;
        if (!ctx_->arena_h_in) {
            void *hi = nullptr, *ho = nullptr, *di = nullptr, *dout = nullptr;
            if (hipHostMalloc(&hi, emp_ctx::kArena, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc(&ho, emp_ctx::kArena, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(&di, emp_ctx::kArena) != hipSuccess || hipMalloc(&dout, emp_ctx::kArena) != hipSuccess) {
                (void)hipGetLastError();
                if (hi) (void)hipHostFree(hi);
                if (ho) (void)hipHostFree(ho);
                if (di) (void)hipFree(di);
                if (dout) (void)hipFree(dout);
                ctx_->arena_failed = true;
                return false;
            }
            ctx_->arena_h_in = (char*)hi;
            ctx_->arena_h_out = (char*)ho;
            ctx_->arena_d_in = (char*)di;
            ctx_->arena_d_out = (char*)dout;
        }
        *used += arena_round(bytes);
        return true;
    }
    emp_ctx* ctx_;
    bool dev_, async_, arena_ = false, out_zeroed_ = false;
    size_t in_used_ = 0, in_sent_ = 0, out_used_ = 0;
    std::vector<Back> arena_backs_;
    std::vector<size_t> arena_offs_;
    std::vector<Back> backs_, ins_;
    Back block_out_ = {nullptr, nullptr, 0, nullptr};
};

// RAII kernel timer: when ctx->timing is on, brackets a launch with a fresh HIP event pair on the context's
// stream.  emp_kernel_ms() later averages all pairs recorded since timing was (re-)enabled.
struct KernelTimer {
    emp_ctx* ctx;
    hipEvent_t start = nullptr, stop = nullptr;
    bool attached = false;   // the launcher hands start / stop to hipExtLaunchKernelGGL itself
    // attach = true: the events are NOT recorded on the stream here; the caller passes them to
    // hipExtLaunchKernelGGL, which stamps the kernel's own begin and end (no extra stream packets, and the
    // interval excludes the wait between the record and the kernel's start)
    KernelTimer(emp_ctx* c, const char* name, bool attach = false) : ctx(c), attached(attach) {
        if (!c->timing) return;
        if (!c->timing_filter.empty() && c->timing_filter != name) return;
        emp_ctx::Ev& e = c->events[name];
        if (e.used == e.pairs.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            e.pairs.push_back({a, b});
        }
        auto& pr = e.pairs[e.used++];
        start = pr.first;
        stop = pr.second;
        if (!attached) (void)hipEventRecord(start, c->stream);
    }
    ~KernelTimer() {
        if (stop && !attached) (void)hipEventRecord(stop, ctx->stream);
    }
};

#define EMP_LAUNCH_CHECK(ctx)                                                                      \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess)                                                                      \
            return emp::fail((ctx), EMP_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e_)); \
    } while (0)

}  // namespace emp
