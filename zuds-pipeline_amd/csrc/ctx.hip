// Context, error string, scratch pool and HIP-event timers of libzudsmi.
#include <cstdarg>

#include "zm_internal.h"

static thread_local char g_err[1024] = "";

void zm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* zm_last_error(void) { return g_err; }
extern "C" const char* zm_version(void) { return "zudsmi 0.1.0 (gfx950)"; }

int zm_ctx::get(const char* name, size_t bytes, void** out) {
    auto it = scratch.find(name);
    if (it != scratch.end() && it->second.second >= bytes) {
        *out = it->second.first;
        return 0;
    }
    if (it != scratch.end()) {
        // the old buffer may still be in use by enqueued work
        ZM_HIP(hipStreamSynchronize(stream));
        ZM_HIP(hipFree(it->second.first));
        scratch.erase(it);
    }
    void* p = nullptr;
    size_t alloc = bytes + (bytes >> 3) + 256;
    ZM_HIP(hipMalloc(&p, alloc));
    scratch[name] = {p, alloc};
    *out = p;
    return 0;
}

int zm_ctx::get_pinned(const char* name, size_t bytes, void** out) {
    auto it = pinned.find(name);
    if (it != pinned.end() && it->second.second >= bytes) {
        *out = it->second.first;
        return 0;
    }
    if (it != pinned.end()) {
        ZM_HIP(hipStreamSynchronize(stream));
        ZM_HIP(hipHostFree(it->second.first));
        pinned.erase(it);
    }
    void* p = nullptr;
    ZM_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    pinned[name] = {p, bytes};
    *out = p;
    return 0;
}

void zm_ctx::release_all() {
    for (auto& kv : scratch) (void)hipFree(kv.second.first);
    scratch.clear();
    for (auto& kv : pinned) (void)hipHostFree(kv.second.first);
    pinned.clear();
    for (auto& kv : timers)
        for (auto& p : kv.second.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    timers.clear();
    for (auto e : event_pool) (void)hipEventDestroy(e);
    event_pool.clear();
    for (auto e : sync_events) (void)hipEventDestroy(e);
    if (bk_stats_event) { (void)hipEventDestroy(bk_stats_event); bk_stats_event = nullptr; }
    if (hp_done) { (void)hipEventDestroy(hp_done); hp_done = nullptr; }
    if (hp_sig_h) { (void)hipHostFree(hp_sig_h); hp_sig_h = hp_sig_d = nullptr; }
    sync_events.clear();
}

extern "C" int zm_ctx_create(int device, zm_ctx** out) {
    ZM_CHECK(out != nullptr, "zm_ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        zm_set_error("zm_ctx_create: no HIP device available (%s)",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return 3;
    }
    ZM_CHECK(device >= 0 && device < ndev, "zm_ctx_create: device %d out of range [0,%d)",
             device, ndev);
    ZM_HIP(hipSetDevice(device));
    zm_ctx* c = new zm_ctx();
    c->device = device;
    ZM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    *out = c;
    return 0;
}

// A context that works on the caller's stream from the start and never creates one of its own (round 6): the chains of
// a pool, the pipelined step and the I/O ring bind every context to a stream they already hold - created and destroyed
// streams shift the runtime's mapping of streams onto hardware queues and with it which chains can run side by side.
extern "C" int zm_ctx_create_on_stream(int device, void* hip_stream, zm_ctx** out) {
    ZM_CHECK(out != nullptr && hip_stream != nullptr, "zm_ctx_create_on_stream: null argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        zm_set_error("zm_ctx_create_on_stream: no HIP device available (%s)",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return 3;
    }
    ZM_CHECK(device >= 0 && device < ndev, "zm_ctx_create_on_stream: device %d out of range [0,%d)", device, ndev);
    ZM_HIP(hipSetDevice(device));
    zm_ctx* c = new zm_ctx();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    *out = c;
    return 0;
}

extern "C" int zm_ctx_destroy(zm_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->release_all();
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

// Round 6: binding a context to the stream it is already bound to costs nothing, and a change of stream orders the
// new stream behind the old one with an EVENT (the context's scratch buffers are shared by whatever runs on either)
// instead of making the host wait for the old stream to drain.  Rounds 1 - 5 synchronised here: the device chains bind
// their engine before every call (several objects share one engine), so the host stood still until the coadd of a
// step had finished before it enqueued the subtraction behind it - 0.13 ms of idle GPU per step (tools/sub_timeline.py).
extern "C" int zm_ctx_set_stream(zm_ctx* ctx, void* hip_stream) {
    ZM_CHECK(ctx != nullptr, "zm_ctx_set_stream: ctx is NULL");
    ZM_HIP(hipSetDevice(ctx->device));
    if (ZM_DEVENV("ZM_SET_STREAM_SYNC")) ZM_HIP(hipStreamSynchronize(ctx->stream));      // (developer: rounds 1 - 5)
    if (hip_stream != nullptr && (hipStream_t)hip_stream == ctx->stream) return 0;
    if (hip_stream == nullptr && ctx->own_stream) return 0;
    hipStream_t old = ctx->stream, next = (hipStream_t)hip_stream;
    const bool old_owned = ctx->own_stream;
    if (next == nullptr) ZM_HIP(hipStreamCreateWithFlags(&next, hipStreamNonBlocking));
    if (old) {
        hipEvent_t* ev = nullptr;
        ZM_TRY(zm_get_sync_events(ctx, 11, &ev));
        ZM_HIP(hipEventRecord(ev[10], old));
        ZM_HIP(hipStreamWaitEvent(next, ev[10], 0));
    }
    if (old_owned && old) ZM_HIP(hipStreamDestroy(old));       // (its pending work completes; the handle is released)
    ctx->stream = next;
    ctx->own_stream = hip_stream == nullptr;
    return 0;
}

extern "C" int zm_ctx_set_share(zm_ctx* ctx, int nctx) {
    ZM_CHECK(ctx != nullptr, "zm_ctx_set_share: ctx is NULL");
    ZM_CHECK(nctx >= 1 && nctx <= 64, "zm_ctx_set_share: %d contexts (1 .. 64)", nctx);
    ctx->share = nctx;
    return 0;
}

extern "C" int zm_ctx_set_conventions(zm_ctx* ctx, int edge, int mask_resample) {
    ZM_CHECK(ctx != nullptr, "zm_ctx_set_conventions: ctx is NULL");
    ZM_CHECK(edge == ZM_EDGE_ZERO || edge == ZM_EDGE_TRUNCATE, "zm_ctx_set_conventions: unknown edge rule %d", edge);
    ZM_CHECK(mask_resample == ZM_MASKRES_OR || mask_resample == ZM_MASKRES_LANCZOS_ROUND,
             "zm_ctx_set_conventions: unknown mask rule %d", mask_resample);
    ctx->edge = edge;
    ctx->mask_resample = mask_resample;
    return 0;
}

extern "C" int zm_ctx_query(zm_ctx* ctx, const char* what, int64_t* out) {
    ZM_CHECK(ctx && what && out, "zm_ctx_query: null argument");
    if (!strcmp(what, "fused_form")) *out = ctx->ff_last_form;
    else if (!strcmp(what, "dev_build")) *out = ZM_DEV_BUILD;
    else if (!strcmp(what, "edge")) *out = ctx->edge;
    else if (!strcmp(what, "mask_resample")) *out = ctx->mask_resample;
    else ZM_CHECK(false, "zm_ctx_query: unknown item \"%s\"", what);
    return 0;
}

// The context's second stream, made on first use (round 6): a context that only subtracts never needs one, and every
// stream a process holds takes part in the runtime's mapping of streams onto its few hardware queues - the chains of a
// pool or of the pipelined step lose their concurrency when idle streams push two busy ones onto one queue.
hipStream_t zm_ctx_aux(zm_ctx* ctx) {
    if (!ctx->aux) {
        (void)hipSetDevice(ctx->device);
        if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess) ctx->aux = nullptr;
    }
    return ctx->aux;
}

int zm_get_sync_events(zm_ctx* ctx, int n, hipEvent_t** out) {
    while ((int)ctx->sync_events.size() < n) {
        hipEvent_t e = nullptr;
        ZM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_events.push_back(e);
    }
    *out = ctx->sync_events.data();
    return 0;
}

extern "C" int zm_ctx_synchronize(zm_ctx* ctx) {
    ZM_CHECK(ctx != nullptr, "zm_ctx_synchronize: ctx is NULL");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- timers ---------------------------------------------------------------
zm_scope_timer::zm_scope_timer(zm_ctx* c, const char* n) : ctx(c), name(n) {
    if (!ctx->timing) return;
    if (!ctx->timing_only.empty() && ctx->timing_only != n) return;
    auto take = [&]() -> hipEvent_t {
        if (!ctx->event_pool.empty()) {
            hipEvent_t e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    };
    a = take();
    b = take();
    (void)hipEventRecord(a, ctx->stream);
}

zm_scope_timer::~zm_scope_timer() {
    if (!ctx->timing || !a) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->timers[name].pending.push_back({a, b});
}

extern "C" int zm_timing_enable(zm_ctx* ctx, int on) {
    ZM_CHECK(ctx != nullptr, "zm_timing_enable: ctx is NULL");
    ctx->timing = on != 0;
    return 0;
}

static int drain(zm_ctx* ctx, zm_timer_slot& s) {
    for (auto& p : s.pending) {
        ZM_HIP(hipEventSynchronize(p.second));
        float ms = 0.f;
        ZM_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        s.total_ms += ms;
        s.launches += 1;
        ctx->event_pool.push_back(p.first);
        ctx->event_pool.push_back(p.second);
    }
    s.pending.clear();
    return 0;
}

extern "C" int zm_timing_filter(zm_ctx* ctx, const char* only_or_null) {
    ZM_CHECK(ctx != nullptr, "zm_timing_filter: ctx is NULL");
    ctx->timing_only = only_or_null ? only_or_null : "";
    return 0;
}

extern "C" int zm_timing_reset(zm_ctx* ctx) {
    ZM_CHECK(ctx != nullptr, "zm_timing_reset: ctx is NULL");
    for (auto& kv : ctx->timers) {
        ZM_TRY(drain(ctx, kv.second));
        kv.second.total_ms = 0.0;
        kv.second.launches = 0;
    }
    return 0;
}

extern "C" int zm_timing_read(zm_ctx* ctx, const char* kernel_name, double* total_ms,
                              int64_t* launches) {
    ZM_CHECK(ctx && kernel_name && total_ms && launches, "zm_timing_read: null argument");
    auto it = ctx->timers.find(kernel_name);
    if (it == ctx->timers.end()) {
        *total_ms = 0.0;
        *launches = 0;
        return 0;
    }
    ZM_TRY(drain(ctx, it->second));
    *total_ms = it->second.total_ms;
    *launches = it->second.launches;
    return 0;
}
