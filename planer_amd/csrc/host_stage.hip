// Host arrays in, host arrays out: a ring of pinned staging buffers, pinned tickets for the way back and a small pool of copy
// threads.
//
// The reference's call contract is host ndarray -> host ndarray (net.Net.__call__, net.py:94-101: np.asarray in, .get()
// out).  pl_h2d / pl_d2h honour it synchronously (the runtime's pageable hipMemcpy: 55 GB/s on this box, but the host and
// the stream wait for every byte).  The asynchronous halves live here:
//   * H2D (pl_h2d_staged): the caller's bytes are copied into a pinned slot of a ring, chunk by chunk, and each chunk is
//     enqueued as a DMA on the CONSUMER's own stream as soon as it is staged (the DMA of chunk k runs under the memcpy of
//     chunk k + 1; the stream's order is the dependency).  The call returns when the caller's array has been read -- it may be
//     overwritten at once -- and no stream but the consumer's ever waits for the copy: with a pipeline of replicas the DMA
//     of batch k + 1 runs under the kernels of the other replicas.
//   * D2H (pl_d2h_begin / pl_d2h_finish): a ticket = a pinned buffer + an event; the copy is enqueued on the producer's
//     stream the moment a result is submitted, the host only waits when it asks for the bytes.
//   * pl_host_alloc / pl_host_free: pinned memory for callers that build their batches in place (no staging copy at all).
// Nothing here has a counterpart in the reference tree (numpy owns its memory; cupy's pinned pool is the closest relative).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>

#include "common.h"

namespace {

// ---- copy threads ------------------------------------------------------------------------------------------------------
// One process-wide pool, started on first use.  A large memcpy is cut into page-aligned pieces; the caller copies one
// piece itself and waits for the rest.  PLANER_HIP_COPY_THREADS sets the worker count (0: the caller alone).
class CopyPool {
public:
    CopyPool() {
        int n = 0;
        const char *e = getenv("PLANER_HIP_COPY_THREADS");
        if (e && *e) n = atoi(e);
        else n = (int)std::min(4u, std::max(1u, std::thread::hardware_concurrency() / 2)) - 1;      // (a lone core copies 19 MB at 75 GB/s: more than three helpers only add wake-ups)
        n = std::max(0, std::min(n, 63));
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int workers() const { return (int)workers_.size(); }

    void copy(void *dst, const void *src, size_t n) {
        constexpr size_t MIN_PIECE = 1u << 20;
        const size_t parts = std::min<size_t>(workers_.size() + 1, std::max<size_t>(1, n / MIN_PIECE));
        if (parts <= 1) {
            memcpy(dst, src, n);
            return;
        }
        std::lock_guard<std::mutex> one(call_mu_);          // one striped copy at a time
        const size_t piece = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t off = piece; off < n; off += piece) {
                jobs_.push_back({(char *)dst + off, (const char *)src + off, std::min(piece, n - off)});
                ++pending_;
            }
        }
        cv_.notify_all();
        memcpy(dst, src, std::min(piece, n));
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    struct Job { char *d; const char *s; size_t n; };
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                j = jobs_.front();
                jobs_.pop_front();
            }
            memcpy(j.d, j.s, j.n);
            {
                std::lock_guard<std::mutex> lk(mu_);
                --pending_;
            }
            done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Job> jobs_;
    int pending_ = 0;
    bool stop_ = false;
};

CopyPool &copy_pool() {
    static CopyPool *pool = new CopyPool();      // never destroyed: worker threads must not be joined from a static destructor
    return *pool;                                // that may run after the interpreter has torn the process half down
}

struct Slot {
    void *host = nullptr;
    size_t cap = 0, bytes = 0;
    hipEvent_t done = nullptr;
    bool pending = false, busy = false;
};

struct Stager {
    // Two dedicated copy streams, only with PLANER_HIP_COPY_STREAMS=1.  Measured on MI355X / ROCm 7.2 (tools/h2d_probe.py,
    // tools/host_submit_probe.py, profiles/r06_host_path.md): every stream a process creates shifts the mapping of its later
    // streams onto the four hardware queues -- two extra ones cost the seven-replica pipeline 10 % -- and a cross-stream hand-off
    // (event on one stream, wait on another) costs the copy more than the ordering it buys: 0.89 ms per 19 MB batch against
    // 0.45 ms with the copy enqueued on the consumer's own stream.  The default therefore has NO stream of its own: staged
    // chunks go out on the consumer's stream, copies back on the producer's.
    hipStream_t in_stream = nullptr, out_stream = nullptr;
    hipEvent_t fence = nullptr;              // orders a copy stream behind a compute stream
    std::mutex mu;
    std::vector<Slot> in_slots;              // H2D ring
    int in_turn = 0;
    std::deque<Slot> tickets;                // D2H buffers in flight (a deque: growing it leaves the slots where they are)
};

constexpr int H2D_SLOTS = 4, MAX_TICKETS = 256;

size_t chunk_bytes() {
    static size_t c = [] {
        const char *e = getenv("PLANER_HIP_COPY_CHUNK_KB");
        long kb = e && *e ? atol(e) : 4096;
        return (size_t)std::max(256L, kb) << 10;
    }();
    return c;
}

bool copy_streams_on() {
    static const bool on = [] {
        const char *e = getenv("PLANER_HIP_COPY_STREAMS");
        return e && e[0] == '1';
    }();
    return on;
}

int stager_of(pl_ctx *ctx, Stager **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->stager) {
        Stager *s = new Stager();
        if (copy_streams_on()) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            const char *pe = getenv("PLANER_HIP_COPY_PRIO");
            const int prio = (pe && pe[0] == '1') ? hi : lo;
            hipError_t e = hipStreamCreateWithPriority(&s->in_stream, hipStreamNonBlocking, prio);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&s->out_stream, hipStreamNonBlocking, prio);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->fence, hipEventDisableTiming);
            if (e != hipSuccess) {
                pl_set_error("copy streams: %s", hipGetErrorString(e));
                if (s->in_stream) (void)hipStreamDestroy(s->in_stream);
                if (s->out_stream) (void)hipStreamDestroy(s->out_stream);
                delete s;
                return PL_EHIP;
            }
        }
        s->in_slots.resize(H2D_SLOTS);
        ctx->stager = s;
    }
    *out = (Stager *)ctx->stager;
    return PL_OK;
}

int slot_reserve(Slot &sl, size_t bytes) {
    if (!sl.done) PL_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (sl.cap < bytes) {
        if (sl.host) (void)hipHostFree(sl.host);
        sl.host = nullptr;
        sl.cap = 0;
        const size_t cap = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        hipError_t e = hipHostMalloc(&sl.host, cap, hipHostMallocDefault);
        if (e != hipSuccess) {
            pl_set_error("hipHostMalloc(%zu): %s", cap, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? PL_ENOMEM : PL_EHIP;
        }
        sl.cap = cap;
    }
    return PL_OK;
}

bool is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                 // a plain malloc'ed pointer is "invalid value" here: not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

}  // namespace

void pl_stager_destroy(pl_ctx *ctx) {
    Stager *s = (Stager *)ctx->stager;
    if (!s) return;
    ctx->stager = nullptr;
    if (s->in_stream) (void)hipStreamSynchronize(s->in_stream);
    if (s->out_stream) (void)hipStreamSynchronize(s->out_stream);
    for (Slot &sl : s->in_slots)
        if (sl.pending) (void)hipEventSynchronize(sl.done);
    for (Slot &sl : s->tickets)
        if (sl.busy) (void)hipEventSynchronize(sl.done);
    auto drop = [](Slot &sl) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.host) (void)hipHostFree(sl.host);
    };
    for (Slot &sl : s->in_slots) drop(sl);
    for (Slot &sl : s->tickets) drop(sl);
    if (s->fence) (void)hipEventDestroy(s->fence);
    if (s->in_stream) (void)hipStreamDestroy(s->in_stream);
    if (s->out_stream) (void)hipStreamDestroy(s->out_stream);
    delete s;
}

extern "C" {

int pl_host_alloc(size_t bytes, void **out) {
    PL_REQUIRE(out, PL_EINVAL, "pl_host_alloc: null out");
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        pl_set_error("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? PL_ENOMEM : PL_EHIP;
    }
    *out = p;
    return PL_OK;
}

int pl_host_free(void *p) {
    if (!p) return PL_OK;
    PL_HIP(hipHostFree(p));
    return PL_OK;
}

// Host bytes -> device NOW, on no stream of ours: the runtime's synchronous copy (it pins the caller's pages and drives the DMA
// engine itself; the host waits on the engine's signal).  Nothing is enqueued on a compute stream, so no hardware queue
// stalls behind the transfer -- a copy enqueued on a stream holds that stream's hardware queue, and whatever other stream
// shares it, for the whole 0.35 ms a 19 MB batch takes.  The caller guarantees that nothing on the device still uses dst
// (Net.submit: a per-replica input buffer, behind the event of the replica's previous feed).
int pl_h2d_direct(pl_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
    PL_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), PL_EINVAL, "pl_h2d_direct: null argument");
    if (!bytes) return PL_OK;
    CtxGuard g(ctx);
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "pl_h2d_direct during capture");
    PL_HIP(hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
    return PL_OK;
}

int pl_copy_threads(int *workers) {
    PL_REQUIRE(workers, PL_EINVAL, "pl_copy_threads: null out");
    *workers = copy_pool().workers();
    return PL_OK;
}

// Host bytes -> device, ordered in front of everything `consumer` (NULL: ctx itself) enqueues after this call, and behind
// everything ctx's own stream held when it was made (dst is a block of ctx's stream-ordered pool: its previous reader may
// still be queued there).  Returns once src_host has been read: the caller may overwrite it.
int pl_h2d_staged(pl_ctx *ctx, pl_ctx *consumer, void *dst, const void *src_host, size_t bytes) {
    PL_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), PL_EINVAL, "pl_h2d_staged: null argument");
    if (!bytes) return PL_OK;
    if (!consumer) consumer = ctx;
    PL_REQUIRE(consumer->device == ctx->device, PL_EINVAL, "pl_h2d_staged: consumer on another device");
    CtxGuard g(ctx);
    PL_REQUIRE(!ctx->capturing && !consumer->capturing, PL_EINVAL, "pl_h2d_staged during capture");
    Stager *s = nullptr;
    if (int r = stager_of(ctx, &s)) return r;
    std::lock_guard<std::mutex> lk(s->mu);
    static const bool trace = getenv("PLANER_HIP_COPY_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t0 = now();
    // which stream carries the DMA: the consumer's own (default) -- its order IS the dependency -- or the copy stream
    hipStream_t via = consumer->stream;
    if (s->in_stream) {
        via = s->in_stream;
        PL_HIP(hipEventRecord(s->fence, ctx->stream));
        PL_HIP(hipStreamWaitEvent(via, s->fence, 0));
    } else if (consumer != ctx) {
        // dst is a block of ctx's pool: its previous reader may still be queued on ctx's stream
        if (!ctx->sync_event) {
            hipEvent_t ev;
            PL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->sync_event = ev;
        }
        PL_HIP(hipEventRecord((hipEvent_t)ctx->sync_event, ctx->stream));
        PL_HIP(hipStreamWaitEvent(via, (hipEvent_t)ctx->sync_event, 0));
    }
    Slot &sl = s->in_slots[s->in_turn];
    s->in_turn = (s->in_turn + 1) % (int)s->in_slots.size();
    if (sl.pending) {
        PL_HIP(hipEventSynchronize(sl.done));
        sl.pending = false;
    }
    const auto t2 = now();
    if (is_pinned(src_host)) {
        // the caller's own pinned memory (pl_host_alloc): the DMA reads it in place, LATER -- no staging copy and no wait;
        // the caller keeps it unchanged until the consumer's stream has passed this point
        if (!sl.done) PL_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        PL_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, via));
        PL_HIP(hipEventRecord(sl.done, via));
        sl.pending = true;
        if (via != consumer->stream) PL_HIP(hipStreamWaitEvent(consumer->stream, sl.done, 0));
        return PL_OK;
    }
    if (int r = slot_reserve(sl, bytes)) return r;
    const size_t chunk = chunk_bytes();
    double t_cpy = 0, t_enq = 0;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        const auto a = now();
        copy_pool().copy((char *)sl.host + off, (const char *)src_host + off, n);
        const auto b = now();
        PL_HIP(hipMemcpyAsync((char *)dst + off, (char *)sl.host + off, n, hipMemcpyHostToDevice, via));
        t_cpy += us(a, b);
        t_enq += us(b, now());
    }
    const auto t3 = now();
    PL_HIP(hipEventRecord(sl.done, via));
    sl.pending = true;
    if (via != consumer->stream) PL_HIP(hipStreamWaitEvent(consumer->stream, sl.done, 0));
    if (trace)
        fprintf(stderr, "[h2d_staged] %zu B: order + slot wait %.0f us, memcpy %.0f, enqueue %.0f, tail %.0f\n", bytes, us(t0, t2), t_cpy,
                t_enq, us(t3, now()));
    return PL_OK;
}

// Device -> host in two halves.  begin: a pinned buffer is picked, the copy is enqueued on the out-going copy stream behind
// everything `producer` (NULL: ctx) has enqueued so far; the host does not wait.  finish: waits for that copy, moves the
// bytes to dst_host (NULL: drop them) and releases the buffer.  *ticket = -1 when every buffer is in flight (the caller
// then uses pl_d2h).  The device block must stay allocated until finish.
int pl_d2h_begin(pl_ctx *ctx, pl_ctx *producer, const void *src, size_t bytes, int *ticket) {
    PL_REQUIRE(ctx && ticket && (bytes == 0 || src), PL_EINVAL, "pl_d2h_begin: null argument");
    if (!producer) producer = ctx;
    PL_REQUIRE(producer->device == ctx->device, PL_EINVAL, "pl_d2h_begin: producer on another device");
    CtxGuard g(ctx);
    PL_REQUIRE(!ctx->capturing && !producer->capturing, PL_EINVAL, "pl_d2h_begin during capture");
    Stager *s = nullptr;
    if (int r = stager_of(ctx, &s)) return r;
    std::lock_guard<std::mutex> lk(s->mu);
    int t = -1;
    for (int i = 0; i < (int)s->tickets.size(); ++i)            // smallest free buffer that fits, else any free one
        if (!s->tickets[i].busy && (t < 0 || (s->tickets[i].cap >= bytes && (s->tickets[t].cap < bytes || s->tickets[i].cap < s->tickets[t].cap))))
            t = i;
    if (t < 0) {
        if ((int)s->tickets.size() >= MAX_TICKETS) {
            *ticket = -1;
            return PL_OK;
        }
        s->tickets.emplace_back();
        t = (int)s->tickets.size() - 1;
    }
    Slot &sl = s->tickets[t];
    if (int r = slot_reserve(sl, std::max<size_t>(bytes, 1))) return r;
    hipStream_t via = producer->stream;
    if (s->out_stream) {
        via = s->out_stream;
        PL_HIP(hipEventRecord(s->fence, producer->stream));
        PL_HIP(hipStreamWaitEvent(via, s->fence, 0));
    }
    if (bytes) PL_HIP(hipMemcpyAsync(sl.host, src, bytes, hipMemcpyDeviceToHost, via));
    PL_HIP(hipEventRecord(sl.done, via));
    sl.busy = true;
    sl.bytes = bytes;
    *ticket = t;
    return PL_OK;
}

int pl_d2h_finish(pl_ctx *ctx, int ticket, void *dst_host) {
    PL_REQUIRE(ctx && ctx->stager, PL_EINVAL, "pl_d2h_finish: no copy in flight on this context");
    CtxGuard g(ctx);
    Stager *s = (Stager *)ctx->stager;
    Slot *sl = nullptr;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        PL_REQUIRE(ticket >= 0 && ticket < (int)s->tickets.size() && s->tickets[ticket].busy, PL_EINVAL,
                   "pl_d2h_finish: ticket %d is not in flight", ticket);
        sl = &s->tickets[ticket];
    }
    hipError_t e = hipEventSynchronize(sl->done);
    if (e == hipSuccess && dst_host && sl->bytes) copy_pool().copy(dst_host, sl->host, sl->bytes);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->tickets[ticket].busy = false;
    }
    if (e != hipSuccess) {
        pl_set_error("pl_d2h_finish: %s", hipGetErrorString(e));
        return PL_EHIP;
    }
    return PL_OK;
}

}  // extern "C"

