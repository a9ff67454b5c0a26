// Runtime half of libplaner_hip.so: contexts, the stream-ordered caching
// pool, host<->device copies, events, whole-forward hipGraph capture and the
// RCCL weight broadcast.  Replaces what numpy's allocator and Python's GC do
// for the reference (net.py:37-72) -- nothing here has a native counterpart
// in the reference tree.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "common.h"

static thread_local std::string g_err;

void pl_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" {

const char *pl_last_error(void) { return g_err.c_str(); }
int pl_version(void) { return 500; }

int pl_device_count(int *count) {
    PL_REQUIRE(count, PL_EINVAL, "pl_device_count: null out");
    PL_HIP(hipGetDeviceCount(count));
    return PL_OK;
}

int pl_ctx_create(int device, pl_ctx **out) {
    PL_REQUIRE(out, PL_EINVAL, "pl_ctx_create: null out");
    int n = 0;
    PL_HIP(hipGetDeviceCount(&n));
    PL_REQUIRE(device >= 0 && device < n, PL_EINVAL,
               "pl_ctx_create: device %d out of range (%d visible)", device, n);
    PL_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PL_HIP(hipGetDeviceProperties(&prop, device));
    pl_ctx *c = new pl_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->hbm_bytes = prop.totalGlobalMem;
    c->arch = prop.gcnArchName;
    { const char *at = getenv("PLANER_HIP_AUTOTUNE"); c->autotune = !(at && at[0] == '0'); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        pl_set_error("hipStreamCreate: %s", hipGetErrorString(e));
        return PL_EHIP;
    }
    *out = c;
    return PL_OK;
}

int pl_pool_trim(pl_ctx *ctx) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    CtxGuard g(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "pl_pool_trim during capture");
    PL_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->free_blocks) {
        ctx->reserved -= kv.first;
        ctx->block_size.erase(kv.second);
        (void)hipFree(kv.second);
    }
    ctx->free_blocks.clear();
    return PL_OK;
}

int pl_ctx_destroy(pl_ctx *ctx) {
    if (!ctx) return PL_OK;
    CtxGuard g(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    pl_stager_destroy(ctx);
    pl_comm_destroy(ctx);
    for (auto &kv : ctx->block_size) (void)hipFree(kv.first);
    if (ctx->sync_event) (void)hipEventDestroy((hipEvent_t)ctx->sync_event);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PL_OK;
}

int pl_ctx_info(pl_ctx *ctx, int *device, int *cu_count, size_t *hbm_bytes,
                char *arch_name, size_t arch_name_len) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    if (device) *device = ctx->device;
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    if (arch_name && arch_name_len) {
        strncpy(arch_name, ctx->arch.c_str(), arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return PL_OK;
}

int pl_ctx_pci_bus_id(pl_ctx *ctx, char *bus_id, size_t len) {
    PL_REQUIRE(ctx && bus_id && len >= 16, PL_EINVAL, "pl_ctx_pci_bus_id: need a buffer of 16 bytes or more");
    PL_HIP(hipDeviceGetPCIBusId(bus_id, (int)len, ctx->device));
    return PL_OK;
}

int pl_sync(pl_ctx *ctx) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    CtxGuard g(ctx);
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "pl_sync during capture");
    PL_HIP(hipStreamSynchronize(ctx->stream));
    return PL_OK;
}

// ---- pool ---------------------------------------------------------------
// Blocks are recycled by size: a forward pass asks for the same handful of
// activation sizes every time, so after the first pass nothing calls
// hipMalloc.  Reuse is stream-ordered (single stream), so a block may be
// handed out again as soon as the host has enqueued its last reader.
static inline size_t round_block(size_t b) {
    if (b == 0) b = 1;
    const size_t q = b < (1u << 20) ? 512 : (size_t)1 << 16;
    return (b + q - 1) / q * q;
}

static void *take_fit(std::multimap<size_t, void *> &fl, size_t sz) {
    auto it = fl.lower_bound(sz);
    if (it == fl.end() || it->first > sz + sz / 4 + 4096) return nullptr;
    void *p = it->second;
    fl.erase(it);
    return p;
}

int pl_alloc(pl_ctx *ctx, size_t bytes, void **out) {
    PL_REQUIRE(ctx && out, PL_EINVAL, "pl_alloc: null argument");
    CtxGuard g(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t sz = round_block(bytes);
    void *p = nullptr;
    // (experiment cap_noreuse=1: a block freed during this capture is not handed out again inside it, so no kernel of
    //  the graph writes lines that an earlier kernel's readers may still hold in another XCD's L2)
    if (ctx->capturing && !pl_experiment("cap_noreuse", 0)) p = take_fit(ctx->cap_free, sz);
    if (!p) p = take_fit(ctx->free_blocks, sz);
    if (!p) {
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            pl_set_error("hipMalloc(%zu): %s", sz, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? PL_ENOMEM : PL_EHIP;
        }
        ctx->block_size[p] = sz;
        ctx->reserved += sz;
    }
    ctx->in_use += ctx->block_size[p];
    ctx->live.insert(p);
    if (ctx->capturing) ctx->cap_blocks.insert(p);
    *out = p;
    return PL_OK;
}

int pl_free(pl_ctx *ctx, void *ptr) {
    if (!ptr) return PL_OK;
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->block_size.find(ptr);
    PL_REQUIRE(it != ctx->block_size.end(), PL_EINVAL, "pl_free: unknown pointer %p", ptr);
    if (!ctx->live.erase(ptr)) return PL_OK;        // already released
    ctx->in_use -= it->second;
    if (ctx->graph_owned.count(ptr)) return PL_OK;  // memory stays with its graph
    if (ctx->capturing && ctx->cap_blocks.count(ptr))
        ctx->cap_free.emplace(it->second, ptr);
    else
        ctx->free_blocks.emplace(it->second, ptr);
    return PL_OK;
}

// The pool block that holds `ptr` (any address inside it): its base and size.  planer_amd.export uses it to find the extent of
// the constants a recorded forward pass reads (weights, prepared filters, lookup tables).
int pl_pool_block(pl_ctx *ctx, const void *ptr, void **base, size_t *bytes) {
    PL_REQUIRE(ctx && ptr && base && bytes, PL_EINVAL, "pl_pool_block: null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &kv : ctx->block_size) {
        const char *b = (const char *)kv.first;
        if ((const char *)ptr >= b && (const char *)ptr < b + kv.second) {
            *base = kv.first;
            *bytes = kv.second;
            return PL_OK;
        }
    }
    pl_set_error("pl_pool_block: %p is not inside a block of this context's pool", ptr);
    return PL_EINVAL;
}

int pl_pool_stats(pl_ctx *ctx, size_t *bytes_reserved, size_t *bytes_in_use) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (bytes_reserved) *bytes_reserved = ctx->reserved;
    if (bytes_in_use) *bytes_in_use = ctx->in_use;
    return PL_OK;
}

// ---- copies -------------------------------------------------------------
int pl_h2d(pl_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
    PL_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), PL_EINVAL, "pl_h2d: null argument");
    if (!bytes) return PL_OK;
    CtxGuard g(ctx);
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "pl_h2d during capture");
    PL_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    PL_HIP(hipStreamSynchronize(ctx->stream));  // src may be pageable/reused
    return PL_OK;
}

int pl_d2h(pl_ctx *ctx, void *dst_host, const void *src, size_t bytes) {
    PL_REQUIRE(ctx && (bytes == 0 || (dst_host && src)), PL_EINVAL, "pl_d2h: null argument");
    if (!bytes) return PL_OK;
    CtxGuard g(ctx);
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "pl_d2h during capture");
    PL_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(hipStreamSynchronize(ctx->stream));
    return PL_OK;
}

int pl_d2d(pl_ctx *ctx, void *dst, const void *src, size_t bytes) {
    PL_REQUIRE(ctx && (bytes == 0 || (dst && src)), PL_EINVAL, "pl_d2d: null argument");
    if (!bytes) return PL_OK;
    CtxGuard g(ctx);
    PL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PL_OK;
}

int pl_memset(pl_ctx *ctx, void *dst, int byte, size_t bytes) {
    PL_REQUIRE(ctx && (bytes == 0 || dst), PL_EINVAL, "pl_memset: null argument");
    if (!bytes) return PL_OK;
    CtxGuard g(ctx);
    PL_HIP(hipMemsetAsync(dst, byte, bytes, ctx->stream));
    return PL_OK;
}

// ---- events -------------------------------------------------------------
int pl_event_create(pl_ctx *ctx, pl_event **out) {
    PL_REQUIRE(ctx && out, PL_EINVAL, "pl_event_create: null argument");
    CtxGuard g(ctx);
    pl_event *e = new pl_event();
    e->ctx = ctx;
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) {
        delete e;
        pl_set_error("hipEventCreate: %s", hipGetErrorString(r));
        return PL_EHIP;
    }
    *out = e;
    return PL_OK;
}

int pl_event_record(pl_ctx *ctx, pl_event *ev) {
    PL_REQUIRE(ctx && ev, PL_EINVAL, "pl_event_record: null argument");
    CtxGuard g(ctx);
    PL_HIP(hipEventRecord(ev->ev, ctx->stream));
    return PL_OK;
}

int pl_event_sync(pl_event *ev) {
    PL_REQUIRE(ev, PL_EINVAL, "pl_event_sync: null argument");
    CtxGuard g(ev->ctx);
    PL_HIP(hipEventSynchronize(ev->ev));
    return PL_OK;
}

int pl_event_elapsed_ms(pl_event *start, pl_event *stop, float *ms) {
    PL_REQUIRE(start && stop && ms, PL_EINVAL, "pl_event_elapsed_ms: null argument");
    CtxGuard g(stop->ctx);
    PL_HIP(hipEventSynchronize(stop->ev));
    PL_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return PL_OK;
}

int pl_event_destroy(pl_event *ev) {
    if (!ev) return PL_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return PL_OK;
}

// Fork/join between the streams of two contexts on one device: everything
// enqueued on `waiter` after this call runs after everything enqueued on
// `signal` before it.  Lets a forward pass fan sub-batches out over side
// streams (concurrent kernels fill CUs a lone small grid leaves idle).
int pl_stream_wait(pl_ctx *waiter, pl_ctx *signal) {
    PL_REQUIRE(waiter && signal, PL_EINVAL, "pl_stream_wait: null ctx");
    if (waiter == signal) return PL_OK;
    PL_REQUIRE(waiter->device == signal->device, PL_EINVAL, "pl_stream_wait: contexts on different devices");
    CtxGuard g(signal);
    if (!signal->sync_event) {
        hipEvent_t ev;
        PL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        signal->sync_event = ev;
    }
    PL_HIP(hipEventRecord((hipEvent_t)signal->sync_event, signal->stream));
    PL_HIP(hipStreamWaitEvent(waiter->stream, (hipEvent_t)signal->sync_event, 0));
    return PL_OK;
}

// Exchange the streams of two contexts of one device (both are drained first).  A context is a memory pool plus a stream; which
// hardware queue a stream sits on is decided by the runtime when the stream is created, and a pipeline's rate depends on it
// (DESIGN 4.7): the plan compiler tries assignments of its replicas to the process's side streams without re-capturing anything --
// graphs are launched on their context's CURRENT stream, events and pools are not tied to a stream.
int pl_ctx_swap_streams(pl_ctx *a, pl_ctx *b) {
    PL_REQUIRE(a && b, PL_EINVAL, "pl_ctx_swap_streams: null ctx");
    if (a == b) return PL_OK;
    PL_REQUIRE(a->device == b->device, PL_EINVAL, "pl_ctx_swap_streams: contexts on different devices");
    PL_REQUIRE(!a->capturing && !b->capturing, PL_EINVAL, "pl_ctx_swap_streams during capture");
    PL_REQUIRE(!a->comm && !b->comm, PL_EINVAL, "pl_ctx_swap_streams: a context that carries a communicator keeps its stream");
    CtxGuard g(a);
    PL_HIP(hipStreamSynchronize(a->stream));
    PL_HIP(hipStreamSynchronize(b->stream));
    std::swap(a->stream, b->stream);
    return PL_OK;
}

int pl_stream_wait_event(pl_ctx *waiter, pl_event *ev) {
    PL_REQUIRE(waiter && ev, PL_EINVAL, "pl_stream_wait_event: null argument");
    PL_REQUIRE(waiter->device == ev->ctx->device, PL_EINVAL, "pl_stream_wait_event: event of another device");
    CtxGuard g(waiter);
    PL_HIP(hipStreamWaitEvent(waiter->stream, ev->ev, 0));
    return PL_OK;
}

// ---- whole-forward capture ----------------------------------------------
int pl_capture_begin(pl_ctx *ctx) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    CtxGuard g(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "capture already in progress");
    PL_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    ctx->cap_blocks.clear();
    ctx->cap_free.clear();
    return PL_OK;
}

int pl_capture_end(pl_ctx *ctx, pl_graph **out) {
    PL_REQUIRE(ctx && out, PL_EINVAL, "pl_capture_end: null argument");
    CtxGuard g(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    PL_REQUIRE(ctx->capturing, PL_EINVAL, "no capture in progress");
    ctx->capturing = false;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        // give every block back; nothing was executed
        for (auto &kv : ctx->cap_free) ctx->free_blocks.emplace(kv.first, kv.second);
        ctx->cap_free.clear();
        ctx->cap_blocks.clear();
        if (graph) (void)hipGraphDestroy(graph);
        pl_set_error("graph capture failed: %s", hipGetErrorString(e));
        return PL_EHIP;
    }
    pl_graph *pg = new pl_graph();
    pg->ctx = ctx;
    pg->graph = graph;
    pg->exec = exec;
    // every block touched during capture now belongs to the graph: the
    // recorded kernels carry its address, so nobody else may be handed it.
    pg->blocks.assign(ctx->cap_blocks.begin(), ctx->cap_blocks.end());
    for (void *p : pg->blocks) ctx->graph_owned.insert(p);
    // blocks still held by the caller (outputs) stay accounted as in use
    // until the graph dies; those already freed were subtracted by pl_free.
    ctx->cap_free.clear();
    ctx->cap_blocks.clear();
    *out = pg;
    return PL_OK;
}

int pl_graph_launch(pl_graph *g) {
    PL_REQUIRE(g && g->exec, PL_EINVAL, "pl_graph_launch: null graph");
    CtxGuard guard(g->ctx);
    PL_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
    return PL_OK;
}

int pl_graph_destroy(pl_graph *g) {
    if (!g) return PL_OK;
    pl_ctx *ctx = g->ctx;
    CtxGuard guard(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (void *p : g->blocks) {
            ctx->graph_owned.erase(p);
            auto it = ctx->block_size.find(p);
            // blocks the caller still holds go back to the pool on their pl_free
            if (it != ctx->block_size.end() && !ctx->live.count(p)) ctx->free_blocks.emplace(it->second, p);
        }
    }
    delete g;
    return PL_OK;
}

// ---- RCCL ---------------------------------------------------------------
// librccl is dlopen'ed on first use so single-GPU processes never pay for it.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                              ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};

static RcclApi *rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
#define SYM(field, name) api.field = (decltype(api.field))dlsym(api.lib, name)
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(Broadcast, "ncclBroadcast");
        SYM(AllReduce, "ncclAllReduce");
        SYM(AllGather, "ncclAllGather");
        SYM(GetErrorString, "ncclGetErrorString");
        SYM(CommCount, "ncclCommCount");
        SYM(CommUserRank, "ncclCommUserRank");
#undef SYM
    });
    if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.Broadcast || !api.AllReduce ||
        !api.AllGather || !api.CommDestroy)
        return nullptr;
    return &api;
}

#define PL_RCCL(api, expr)                                                    \
    do {                                                                      \
        ncclResult_t r_ = (expr);                                             \
        if (r_ != ncclSuccess) {                                              \
            pl_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                         (api)->GetErrorString ? (api)->GetErrorString(r_) : "rccl error"); \
            return PL_ERCCL;                                                  \
        }                                                                     \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PL_UNIQUE_ID_BYTES, "unique id size");

int pl_comm_unique_id(void *id_out) {
    PL_REQUIRE(id_out, PL_EINVAL, "pl_comm_unique_id: null out");
    RcclApi *api = rccl();
    PL_REQUIRE(api, PL_ERCCL, "librccl could not be loaded: %s", dlerror());
    ncclUniqueId id;
    PL_RCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return PL_OK;
}

int pl_comm_init_rank(pl_ctx *ctx, int world, int rank, const void *id) {
    PL_REQUIRE(ctx && id, PL_EINVAL, "pl_comm_init_rank: null argument");
    PL_REQUIRE(world >= 1 && rank >= 0 && rank < world, PL_EINVAL, "bad world/rank %d/%d", rank, world);
    PL_REQUIRE(!ctx->comm, PL_EINVAL, "communicator already initialised");
    RcclApi *api = rccl();
    PL_REQUIRE(api, PL_ERCCL, "librccl could not be loaded");
    CtxGuard g(ctx);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    PL_RCCL(api, api->CommInitRank(&comm, world, uid, rank));
    ctx->comm = comm;
    ctx->world = world;
    ctx->rank = rank;
    return PL_OK;
}

int pl_comm_bcast(pl_ctx *ctx, void *buf, size_t bytes, int root) {
    PL_REQUIRE(ctx && ctx->comm, PL_EINVAL, "pl_comm_bcast: no communicator");
    PL_REQUIRE(buf || !bytes, PL_EINVAL, "pl_comm_bcast: null buffer");
    RcclApi *api = rccl();
    CtxGuard g(ctx);
    PL_RCCL(api, api->Broadcast(buf, buf, bytes, ncclUint8, root, (ncclComm_t)ctx->comm, ctx->stream));
    return PL_OK;
}

int pl_comm_allreduce_max_f32(pl_ctx *ctx, float *buf, size_t n) {
    PL_REQUIRE(ctx && ctx->comm && buf, PL_EINVAL, "pl_comm_allreduce_max_f32: bad argument");
    RcclApi *api = rccl();
    CtxGuard g(ctx);
    PL_RCCL(api, api->AllReduce(buf, buf, n, ncclFloat32, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    return PL_OK;
}

int pl_comm_allgather(pl_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank) {
    PL_REQUIRE(ctx && ctx->comm && send && recv, PL_EINVAL, "pl_comm_allgather: bad argument");
    RcclApi *api = rccl();
    CtxGuard g(ctx);
    PL_RCCL(api, api->AllGather(send, recv, bytes_per_rank, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    return PL_OK;
}

int pl_comm_info(pl_ctx *ctx, int *ranks, int *rank) {
    PL_REQUIRE(ctx && ctx->comm, PL_EINVAL, "pl_comm_info: no communicator");
    RcclApi *api = rccl();
    PL_REQUIRE(api && api->CommCount && api->CommUserRank, PL_ERCCL, "librccl has no ncclCommCount / ncclCommUserRank");
    if (ranks) PL_RCCL(api, api->CommCount((ncclComm_t)ctx->comm, ranks));
    if (rank) PL_RCCL(api, api->CommUserRank((ncclComm_t)ctx->comm, rank));
    return PL_OK;
}

int pl_comm_destroy(pl_ctx *ctx) {
    if (!ctx || !ctx->comm) return PL_OK;
    RcclApi *api = rccl();
    if (api) (void)api->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
    return PL_OK;
}

}  // extern "C"
