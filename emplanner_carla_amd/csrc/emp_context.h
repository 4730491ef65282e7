// emp_context.h - context object behind the C-ABI: device, stream, scratch pool, host staging, timing.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/emplanner.h"

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
    // lattice parameters the "dp_pair_table" scratch was built for (emp_api.hip: dev_dp_edge)
    std::string timing_filter;   // non-empty: only this kernel name is bracketed by events
    double pair_table_key[8] = {0};
    bool pair_table_valid = false;
    // per-kernel timing
    bool timing = false;
    struct Ev {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;   // one pair per launch since timing was enabled
        size_t used = 0;
    };
    std::map<std::string, Ev> events;
    int cu_count = 0;
    // Two-stage pipelining of consecutive emp_plan_cycle calls (emp_set_pipeline): the back stage (path QP, Cartesian
    // tail) of call k runs on `stream2` while the front stage (projection, DP) of call k+1 already runs on `stream`.
    // Each parity has its own pool of temporaries; ev_back[parity] marks the end of the back stage that read them.
    bool pipeline = false;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_front = nullptr;
    hipEvent_t ev_back[2] = {nullptr, nullptr};
    bool ev_back_valid[2] = {false, false};
    int parity = 0;
    std::vector<Buf> cycle_pool[2];
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

// Grow-only device buffer with 25 % headroom.  A buffer is replaced only once nothing queued on either of the context's
// streams can still touch it: with two batches in flight the back stage of an earlier call may be reading the very
// temporaries a larger batch now outgrows, and hipFree's own implicit synchronisation is not relied upon.
inline int grow_buffer(emp_ctx* ctx, emp_ctx::Buf& b, size_t bytes) {
    if (b.bytes >= bytes) return EMP_OK;
    if (b.p) {
        EMP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->stream2) EMP_HIP(ctx, hipStreamSynchronize(ctx->stream2));
        EMP_HIP(ctx, hipFree(b.p));
    }
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = bytes + bytes / 4;
    EMP_HIP(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
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
    // in_cycle: the pipelined emp_plan_cycle orders itself; every OTHER call in pipelined mode first lets the main
    // stream wait for the back stage still in flight, so that it may consume a cycle's outputs as before
    Stage(emp_ctx* c, emp_mem where, bool in_cycle = false) : ctx_(c), dev_(where == EMP_DEVICE) {
        c->cursor = 0;
        if (!in_cycle && c->pipeline)
            for (int par = 0; par < 2; ++par)
                if (c->ev_back_valid[par]) (void)hipStreamWaitEvent(c->stream, c->ev_back[par], 0);
    }

    template <typename T>
    int in(const T* host, size_t n, const T** out) {
        if (host == nullptr) { *out = nullptr; return EMP_OK; }
        if (dev_) { *out = host; return EMP_OK; }
        void* d = nullptr;
        int rc = pool_get(ctx_, n * sizeof(T), &d);
        if (rc) return rc;
        if (n) EMP_HIP(ctx_, hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, ctx_->stream));
        *out = (const T*)d;
        return EMP_OK;
    }
    // Outputs are zero-filled on the context's stream before the kernels run, so padding beyond a scene's
    // length reads as 0 in both memory spaces (pass zero=false for arrays the kernels fully overwrite).
    template <typename T>
    int out(T* host, size_t n, T** outp, bool zero = true) {
        if (host == nullptr) { *outp = nullptr; return EMP_OK; }
        T* d = host;
        if (!dev_) {
            void* v = nullptr;
            int rc = pool_get(ctx_, n * sizeof(T), &v);
            if (rc) return rc;
            d = (T*)v;
            backs_.push_back({host, d, n * sizeof(T)});
        }
        if (zero && n) EMP_HIP(ctx_, hipMemsetAsync(d, 0, n * sizeof(T), ctx_->stream));
        *outp = d;
        return EMP_OK;
    }
    // device-only temporary
    template <typename T>
    int tmp(size_t n, T** outp, bool zero = false) {
        void* v = nullptr;
        int rc = pool_get(ctx_, n * sizeof(T), &v);
        if (rc) return rc;
        if (zero && n) EMP_HIP(ctx_, hipMemsetAsync(v, 0, n * sizeof(T), ctx_->stream));
        *outp = (T*)v;
        return EMP_OK;
    }
    int finish() {
        for (auto& b : backs_)
            if (b.bytes) EMP_HIP(ctx_, hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, ctx_->stream));
        if (!dev_) EMP_HIP(ctx_, hipStreamSynchronize(ctx_->stream));
        return EMP_OK;
    }
    bool on_device() const { return dev_; }

  private:
    struct Back {
        void* host;
        void* dev;
        size_t bytes;
    };
    emp_ctx* ctx_;
    bool dev_;
    std::vector<Back> backs_;
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
