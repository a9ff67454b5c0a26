// Host arrays in, host arrays out: pinned staging rings, copy streams and a small pool of copy threads.
//
// The reference's call contract is host ndarray -> host ndarray (net.Net.__call__, net.py:94-101: np.asarray in, .get()
// out).  A numpy array is pageable memory; hipMemcpyAsync from it is a synchronous, single-threaded bounce through the
// runtime's own staging buffer on the caller's compute stream.  Here instead:
//   * H2D: the caller's bytes are copied by several host threads into a pinned slot of a ring (in chunks), each chunk goes
//     to the device by DMA on a dedicated copy stream as soon as it is staged (the DMA of chunk k runs under the memcpy of
//     chunk k + 1), and the CONSUMER's stream waits for the ring slot's event -- the compute streams never carry a copy and
//     the call returns as soon as the caller's array has been read (it may be overwritten at once).
//   * D2H: a ticket = a pinned buffer + an event; the copy is enqueued on a second copy stream behind the producer's
//     stream the moment a result is submitted, the host only waits when it asks for the bytes.
//   * pl_host_alloc / pl_host_free: pinned memory for callers that build their batches in place (no staging copy at all).
// Nothing here has a counterpart in the reference tree (numpy owns its memory; cupy's pinned pool is the closest relative).
#include <algorithm>
#include <atomic>
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
        else n = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2)) - 1;
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
        constexpr size_t MIN_PIECE = 256u << 10;
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
    hipStream_t in_stream = nullptr, out_stream = nullptr;
    hipEvent_t fence = nullptr;              // orders a copy stream behind a compute stream
    std::mutex mu;
    std::vector<Slot> in_slots;              // H2D ring
    int in_turn = 0;
    std::deque<Slot> tickets;                // D2H buffers in flight (a deque: growing it leaves the slots where they are)
};

constexpr int H2D_SLOTS = 4, MAX_TICKETS = 256;
constexpr size_t STAGE_MIN = 128u << 10;     // below this a copy keeps the plain path

size_t chunk_bytes() {
    static size_t c = [] {
        const char *e = getenv("PLANER_HIP_COPY_CHUNK_KB");
        long kb = e && *e ? atol(e) : 4096;
        return (size_t)std::max(256L, kb) << 10;
    }();
    return c;
}

int stager_of(pl_ctx *ctx, Stager **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->stager) {
        Stager *s = new Stager();
        hipError_t e = hipStreamCreateWithFlags(&s->in_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->out_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->fence, hipEventDisableTiming);
        if (e != hipSuccess) {
            pl_set_error("copy streams: %s", hipGetErrorString(e));
            if (s->in_stream) (void)hipStreamDestroy(s->in_stream);
            if (s->out_stream) (void)hipStreamDestroy(s->out_stream);
            delete s;
            return PL_EHIP;
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

bool staging_on() {
    const char *e = getenv("PLANER_HIP_STAGED");
    return !(e && e[0] == '0');
}

}  // namespace

void pl_stager_destroy(pl_ctx *ctx) {
    Stager *s = (Stager *)ctx->stager;
    if (!s) return;
    ctx->stager = nullptr;
    (void)hipStreamSynchronize(s->in_stream);
    (void)hipStreamSynchronize(s->out_stream);
    auto drop = [](Slot &sl) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.host) (void)hipHostFree(sl.host);
    };
    for (Slot &sl : s->in_slots) drop(sl);
    for (Slot &sl : s->tickets) drop(sl);
    (void)hipEventDestroy(s->fence);
    (void)hipStreamDestroy(s->in_stream);
    (void)hipStreamDestroy(s->out_stream);
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
    PL_HIP(hipEventRecord(s->fence, ctx->stream));
    PL_HIP(hipStreamWaitEvent(s->in_stream, s->fence, 0));
    Slot &sl = s->in_slots[s->in_turn];
    s->in_turn = (s->in_turn + 1) % (int)s->in_slots.size();
    if (sl.pending) {
        PL_HIP(hipEventSynchronize(sl.done));
        sl.pending = false;
    }
    if (is_pinned(src_host)) {
        // the caller's own pinned memory: DMA straight out of it, and hold the call until it has been read
        if (!sl.done) PL_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        PL_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, s->in_stream));
        PL_HIP(hipEventRecord(sl.done, s->in_stream));
        PL_HIP(hipStreamWaitEvent(consumer->stream, sl.done, 0));
        PL_HIP(hipEventSynchronize(sl.done));
        return PL_OK;
    }
    if (int r = slot_reserve(sl, bytes)) return r;
    const size_t chunk = chunk_bytes();
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        copy_pool().copy((char *)sl.host + off, (const char *)src_host + off, n);
        PL_HIP(hipMemcpyAsync((char *)dst + off, (char *)sl.host + off, n, hipMemcpyHostToDevice, s->in_stream));
    }
    PL_HIP(hipEventRecord(sl.done, s->in_stream));
    sl.pending = true;
    PL_HIP(hipStreamWaitEvent(consumer->stream, sl.done, 0));
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
    PL_HIP(hipEventRecord(s->fence, producer->stream));
    PL_HIP(hipStreamWaitEvent(s->out_stream, s->fence, 0));
    if (bytes) PL_HIP(hipMemcpyAsync(sl.host, src, bytes, hipMemcpyDeviceToHost, s->out_stream));
    PL_HIP(hipEventRecord(sl.done, s->out_stream));
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

// (used by runtime.hip's pl_h2d / pl_d2h for large copies)
int pl_h2d_large(pl_ctx *ctx, void *dst, const void *src_host, size_t bytes, bool *done) {
    *done = false;
    if (bytes < STAGE_MIN || !staging_on()) return PL_OK;
    if (int r = pl_h2d_staged(ctx, ctx, dst, src_host, bytes)) return r;
    PL_HIP(hipStreamSynchronize(ctx->stream));       // pl_h2d's contract: the bytes are on the device when it returns
    *done = true;
    return PL_OK;
}

int pl_d2h_large(pl_ctx *ctx, void *dst_host, const void *src, size_t bytes, bool *done) {
    *done = false;
    if (bytes < STAGE_MIN || !staging_on() || is_pinned(dst_host)) return PL_OK;
    int t = -1;
    if (int r = pl_d2h_begin(ctx, ctx, src, bytes, &t)) return r;
    if (t < 0) return PL_OK;
    if (int r = pl_d2h_finish(ctx, t, dst_host)) return r;
    *done = true;
    return PL_OK;
}
