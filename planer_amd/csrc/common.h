// Shared internals of libplaner_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/planer_hip.h"

void pl_set_error(const char *fmt, ...);

#define PL_HIP(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            pl_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                         hipGetErrorString(e_));                              \
            return PL_EHIP;                                                   \
        }                                                                     \
    } while (0)

#define PL_REQUIRE(cond, code, ...)                                           \
    do {                                                                      \
        if (!(cond)) {                                                        \
            pl_set_error(__VA_ARGS__);                                        \
            return (code);                                                    \
        }                                                                     \
    } while (0)

// Launch-error check that is legal during stream capture.
#define PL_LAUNCH_CHECK()                                                     \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            pl_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,    \
                         hipGetErrorString(e_));                              \
            return PL_EHIP;                                                   \
        }                                                                     \
    } while (0)

struct pl_graph;

// One device, one stream, one caching pool.  All pool state is guarded by mu.
struct pl_ctx {
    int device = 0;
    int cu_count = 0;
    size_t hbm_bytes = 0;
    std::string arch;
    hipStream_t stream = nullptr;
    std::mutex mu;

    // caching pool: every block ever hipMalloc'ed by this context
    std::unordered_map<void *, size_t> block_size;
    std::multimap<size_t, void *> free_blocks;
    std::unordered_set<void *> live;         // blocks currently handed out
    size_t reserved = 0, in_use = 0;

    // capture state: blocks handed out while capturing belong to the graph
    bool capturing = false;
    std::unordered_set<void *> cap_blocks;
    std::multimap<size_t, void *> cap_free;
    std::unordered_set<void *> graph_owned;  // pl_free on these is a no-op

    // conv tuning overrides
    int conv_cfg = -1;
    int conv_split_k = 0;
    bool autotune = true;
    int conv_t1 = 0, conv_occ = 0;
    void *sync_event = nullptr;      // hipEvent_t used by pl_stream_wait
    std::string last_plan;           // how the last conv on this context was launched (pl_conv2d_last_plan)
    long long last_gemm[4] = {0, 0, 0, 0};   // executed GEMM extents of that conv: groups, padded rows, cols, K (pl_conv2d_last_extents)
    int xcd_cols_request = 0;        // columns of the next grouped GEMM that one XCD must own (conv_winograd.hip sets it around conv_launch)
    int tune_misses = 0;             // conv shapes this context had to time because no cached launch plan existed

    // RCCL (dlopen'ed on first use)
    void *comm = nullptr;
    int world = 1, rank = 0;

    // pinned staging rings + copy streams for host arrays (host_stage.hip; created on first use)
    void *stager = nullptr;
};

void pl_stager_destroy(pl_ctx *ctx);     // host_stage.hip

struct pl_graph {
    pl_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<void *> blocks;
};

struct pl_event {
    pl_ctx *ctx = nullptr;
    hipEvent_t ev = nullptr;
};

struct CtxGuard {  // make the context's device current for this call
    explicit CtxGuard(pl_ctx *c) { (void)hipSetDevice(c->device); }
};

// Measurement-only switches, all behind ONE environment variable: PLANER_HIP_EXPERIMENT="key=value,key=value".  Read per call
// (tests and tools flip it at run time); a key that is absent gives `dflt`.  Nothing the shipped plans depend on lives here.
static inline int pl_experiment(const char *key, int dflt) {
    const char *e = getenv("PLANER_HIP_EXPERIMENT");
    if (!e) return dflt;
    const size_t n = strlen(key);
    for (const char *p = e; *p;) {
        if (!strncmp(p, key, n) && p[n] == '=') return atoi(p + n + 1);
        const char *c = strchr(p, ',');
        if (!c) break;
        p = c + 1;
    }
    return dflt;
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
