// pl_plan_build / pl_plan_run / pl_plan_destroy: a compiled forward pass replayed WITHOUT Python.
//
// The reference interprets its flow layer by layer (net.Net.forward, net.py:37-72).  The plan compiler of this library
// (planer_amd/plan.py, net.py) turns a flow into a fused program -- conv epilogues, channel-quad layouts, Winograd stages and
// chains, paired convs, the stem + max-pool kernel -- and so far only its Python host could run the result.  A PLAN FILE
// (planer_amd/export.py) is that program flattened: the sequence of C-ABI calls one forward pass makes, with every pointer
// replaced by (arena offset | constant offset | input / output slot), the constants (weights and the prepared filters) and
// the arena size.  pl_plan_build uploads the constants, runs the sequence once (so launch plans are tuned or taken from the
// database) and captures it into a hipGraph; pl_plan_run copies the caller's inputs in, launches the graph and copies the
// outputs out.  A host that binds include/planer_hip.h directly -- C, Go through cgo, Rust through FFI -- gets the fused path
// with three calls (SURVEY 8(b) export list, 7 step 5: "executed by one C call").
//
// File layout (little endian), written by planer_amd.export.export_plan:
//   char magic[8] = "PLPLAN1\0"; u64 const_bytes; u64 arena_bytes; u32 n_in; u32 n_out; u32 n_calls; u32 reserved;
//   n_in + n_out tensor records { u64 arena_offset; u64 bytes; u32 dtype (0 f32, 1 i32, 2 i64, 3 u8); u32 ndim; u32 dims[8]; }
//   n_calls call records { u32 name_len; char name[name_len padded to 4]; u32 nargs; args... }
//     arg = u32 kind (0 int, 1 double, 2 null, 3 arena pointer, 4 constant pointer, 5 host bytes, 6 the context) + u32 pad + u64 payload
//           (kind 5: payload = length, followed by the bytes padded to 8)
//   const_bytes of constants.
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct PlArg {
    long long i;
    double d;
    void *p;
};
struct PlDispatch {
    const char *name;
    int nargs;
    int (*fn)(const PlArg *);
};
#include "plan_dispatch.inc"

struct PlanTensor {
    size_t offset, bytes;
    unsigned dtype, ndim, dims[8];
};
struct PlanCall {
    const PlDispatch *d;
    std::vector<PlArg> args;
};

}  // namespace

struct pl_plan {
    pl_ctx *ctx = nullptr;
    void *arena = nullptr, *consts = nullptr;
    size_t arena_bytes = 0, const_bytes = 0;
    std::vector<PlanTensor> inputs, outputs;
    std::vector<PlanCall> calls;
    std::vector<std::vector<char>> host_blobs;     // kind-5 arguments (shape / stride tables of the few ops that take them)
    pl_graph *graph = nullptr;
};

namespace {

struct Reader {
    const unsigned char *p, *end;
    bool ok = true;
    // (every bound is checked against what is LEFT, never by adding a length from the file to a pointer: a length near 2^64
    //  must not wrap past the end)
    size_t left() const { return (size_t)(end - p); }
    template <class T>
    T get() {
        T v{};
        if (!ok || sizeof(T) > left()) {
            ok = false;
            return v;
        }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const unsigned char *bytes(unsigned long long n) {
        if (!ok || n > left()) {
            ok = false;
            return nullptr;
        }
        const unsigned char *r = p;
        p += (size_t)n;
        return r;
    }
    // n bytes stored padded to a multiple of `align`
    const unsigned char *padded(unsigned long long n, unsigned align) {
        if (!ok || n > left()) {                  // (before rounding up: the rounding itself could wrap)
            ok = false;
            return nullptr;
        }
        return bytes((n + align - 1) / align * align);
    }
};

int run_calls(pl_plan *pl) {
    for (size_t k = 0; k < pl->calls.size(); ++k) {
        const PlanCall &c = pl->calls[k];
        int rc = c.d->fn(c.args.data());
        if (rc != PL_OK) {
            std::string inner = pl_last_error();
            pl_set_error("pl_plan: call %zu (%s) failed: %s", k, c.d->name, inner.c_str());
            return rc;
        }
    }
    return PL_OK;
}

}  // namespace

extern "C" {

int pl_plan_build(pl_ctx *ctx, const void *program, size_t program_bytes, pl_plan **out) {
    PL_REQUIRE(ctx && program && out, PL_EINVAL, "pl_plan_build: null argument");
    Reader r{(const unsigned char *)program, (const unsigned char *)program + program_bytes};
    const unsigned char *magic = r.bytes(8);
    PL_REQUIRE(magic && !memcmp(magic, "PLPLAN1\0", 8), PL_EINVAL, "pl_plan_build: not a plan file (bad magic)");
    const unsigned long long const_bytes = r.get<unsigned long long>(), arena_bytes = r.get<unsigned long long>();
    const unsigned n_in = r.get<unsigned>(), n_out = r.get<unsigned>(), n_calls = r.get<unsigned>();
    (void)r.get<unsigned>();
    PL_REQUIRE(r.ok && n_in <= 64 && n_out <= 64 && n_calls <= (1u << 20), PL_EINVAL, "pl_plan_build: truncated or implausible header");
    pl_plan *pl = new pl_plan();
    pl->ctx = ctx;
    pl->arena_bytes = arena_bytes;
    pl->const_bytes = const_bytes;
    auto fail = [&](int code) {
        pl_plan_destroy(pl);
        return code;
    };
    for (unsigned t = 0; t < n_in + n_out; ++t) {
        PlanTensor pt;
        pt.offset = r.get<unsigned long long>();
        pt.bytes = r.get<unsigned long long>();
        pt.dtype = r.get<unsigned>();
        pt.ndim = r.get<unsigned>();
        for (int d = 0; d < 8; ++d) pt.dims[d] = r.get<unsigned>();
        if (!r.ok || pt.ndim > 8 || pt.bytes > arena_bytes || pt.offset > arena_bytes - pt.bytes) {
            pl_set_error("pl_plan_build: bad tensor record %u", t);
            return fail(PL_EINVAL);
        }
        (t < n_in ? pl->inputs : pl->outputs).push_back(pt);
    }
    CtxGuard g(ctx);
    int rc = pl_alloc(ctx, arena_bytes ? arena_bytes : 1, &pl->arena);
    if (rc != PL_OK) return fail(rc);
    rc = pl_alloc(ctx, const_bytes ? const_bytes : 1, &pl->consts);
    if (rc != PL_OK) return fail(rc);
    pl->calls.reserve(n_calls);
    for (unsigned k = 0; k < n_calls; ++k) {
        const unsigned name_len = r.get<unsigned>();
        const unsigned char *nm = r.padded(name_len, 4);
        const unsigned nargs = r.get<unsigned>();
        if (!r.ok || name_len > 96 || nargs > 40) {
            pl_set_error("pl_plan_build: bad call record %u", k);
            return fail(PL_EINVAL);
        }
        const std::string name((const char *)nm, name_len);
        const PlDispatch *d = nullptr;
        for (int i = 0; i < kPlanDispatchCount; ++i)
            if (name == kPlanDispatch[i].name) d = &kPlanDispatch[i];
        if (!d || d->nargs != (int)nargs) {
            pl_set_error("pl_plan_build: call %u names '%s' with %u arguments: not a replayable entry point of this library", k, name.c_str(), nargs);
            return fail(PL_EUNSUPPORTED);
        }
        PlanCall c;
        c.d = d;
        c.args.resize(nargs);
        for (unsigned a = 0; a < nargs; ++a) {
            const unsigned kind = r.get<unsigned>();
            (void)r.get<unsigned>();
            PlArg v{0, 0.0, nullptr};
            if (kind == 0) {
                v.i = r.get<long long>();
            } else if (kind == 1) {
                v.d = r.get<double>();
            } else if (kind == 2) {
                (void)r.get<unsigned long long>();
            } else if (kind == 3 || kind == 4) {
                const unsigned long long off = r.get<unsigned long long>();
                if (off > (kind == 3 ? arena_bytes : const_bytes)) r.ok = false;
                v.p = (char *)(kind == 3 ? pl->arena : pl->consts) + off;
            } else if (kind == 5) {
                const unsigned long long len = r.get<unsigned long long>();
                const unsigned char *b = r.padded(len, 8);
                if (b) {
                    pl->host_blobs.emplace_back((const char *)b, (const char *)b + len);
                    v.p = pl->host_blobs.back().data();
                }
            } else if (kind == 6) {
                (void)r.get<unsigned long long>();
                v.p = ctx;
            } else {
                r.ok = false;
            }
            if (!r.ok) {
                pl_set_error("pl_plan_build: bad argument %u of call %u (%s)", a, k, name.c_str());
                return fail(PL_EINVAL);
            }
            c.args[a] = v;
        }
        pl->calls.push_back(std::move(c));
    }
    const unsigned char *cb = r.bytes(const_bytes);
    if (!r.ok || !cb) {
        pl_set_error("pl_plan_build: the file ends before its %llu bytes of constants", const_bytes);
        return fail(PL_EINVAL);
    }
    if (const_bytes) {
        rc = pl_h2d(ctx, pl->consts, cb, const_bytes);
        if (rc != PL_OK) return fail(rc);
    }
    rc = pl_memset(ctx, pl->arena, 0, arena_bytes ? arena_bytes : 1);
    if (rc != PL_OK) return fail(rc);
    // one eager pass: launch plans are taken from the tuning database or found now; the pool learns the temporaries' sizes
    rc = run_calls(pl);
    if (rc != PL_OK) return fail(rc);
    rc = pl_sync(ctx);
    if (rc != PL_OK) return fail(rc);
    rc = pl_capture_begin(ctx);
    if (rc != PL_OK) return fail(rc);
    rc = run_calls(pl);
    pl_graph *graph = nullptr;
    int rc2 = pl_capture_end(ctx, &graph);
    if (rc != PL_OK || rc2 != PL_OK) {
        if (graph) pl_graph_destroy(graph);
        return fail(rc != PL_OK ? rc : rc2);
    }
    pl->graph = graph;
    *out = pl;
    return PL_OK;
}

int pl_plan_info(pl_plan *plan, int *n_inputs, int *n_outputs, size_t *arena_bytes, size_t *const_bytes, int *n_calls) {
    PL_REQUIRE(plan, PL_EINVAL, "pl_plan_info: null plan");
    if (n_inputs) *n_inputs = (int)plan->inputs.size();
    if (n_outputs) *n_outputs = (int)plan->outputs.size();
    if (arena_bytes) *arena_bytes = plan->arena_bytes;
    if (const_bytes) *const_bytes = plan->const_bytes;
    if (n_calls) *n_calls = (int)plan->calls.size();
    return PL_OK;
}

int pl_plan_tensor(pl_plan *plan, int output, int index, void **device_ptr, size_t *bytes, int *dtype, int *ndim, int *dims8) {
    PL_REQUIRE(plan, PL_EINVAL, "pl_plan_tensor: null plan");
    const auto &v = output ? plan->outputs : plan->inputs;
    PL_REQUIRE(index >= 0 && index < (int)v.size(), PL_EINVAL, "pl_plan_tensor: %s %d out of range", output ? "output" : "input", index);
    const PlanTensor &t = v[index];
    if (device_ptr) *device_ptr = (char *)plan->arena + t.offset;
    if (bytes) *bytes = t.bytes;
    if (dtype) *dtype = (int)t.dtype;
    if (ndim) *ndim = (int)t.ndim;
    if (dims8)
        for (int d = 0; d < 8; ++d) dims8[d] = d < (int)t.ndim ? (int)t.dims[d] : 0;
    return PL_OK;
}

int pl_plan_run(pl_plan *plan, const void *const *inputs, void *const *outputs) {
    PL_REQUIRE(plan && plan->graph, PL_EINVAL, "pl_plan_run: null plan");
    pl_ctx *ctx = plan->ctx;
    CtxGuard g(ctx);
    for (size_t i = 0; i < plan->inputs.size(); ++i) {
        if (!inputs || !inputs[i]) continue;                       // the caller wrote the plan's own buffer (pl_plan_tensor)
        PL_HIP(hipMemcpyAsync((char *)plan->arena + plan->inputs[i].offset, inputs[i], plan->inputs[i].bytes, hipMemcpyDeviceToDevice,
                              ctx->stream));
    }
    int rc = pl_graph_launch(plan->graph);
    if (rc != PL_OK) return rc;
    for (size_t i = 0; i < plan->outputs.size(); ++i) {
        if (!outputs || !outputs[i]) continue;
        PL_HIP(hipMemcpyAsync(outputs[i], (char *)plan->arena + plan->outputs[i].offset, plan->outputs[i].bytes, hipMemcpyDeviceToDevice,
                              ctx->stream));
    }
    return PL_OK;
}

int pl_plan_destroy(pl_plan *plan) {
    if (!plan) return PL_OK;
    if (plan->graph) pl_graph_destroy(plan->graph);
    if (plan->arena) pl_free(plan->ctx, plan->arena);
    if (plan->consts) pl_free(plan->ctx, plan->consts);
    delete plan;
    return PL_OK;
}

}  // extern "C"
