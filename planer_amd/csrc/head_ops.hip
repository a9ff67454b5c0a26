// Operators that ONNX-exported detection heads and post-processing graphs add around the convolutional
// trunk (reference layer.py:155-157, 202-234, 253-258): comparisons, Where, Cast, Gather, Erf (the
// reference's 1025-entry lookup table) and InstanceNormalization.  All HBM-bound: one pass over the
// operands, grid-stride loops capped near 8 blocks per CU; the one reduction (instance norm) is a
// wave64 shuffle tree per (n, c) row.
#include "common.h"
#include "device_utils.h"

namespace {

constexpr int TPB = 256;

inline unsigned stream_grid(pl_ctx *ctx, size_t work_items) {
    size_t blocks = (work_items + TPB - 1) / TPB;
    size_t cap = (size_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    if (blocks > cap) blocks = cap;
    return blocks ? (unsigned)blocks : 1u;
}

// layer.Equal / Greater / GreaterOrEqual (layer.py:204, 228, 232): numpy comparison -> bool bytes
__global__ void __launch_bounds__(TPB) compare_kernel(const float *a, const float *b, unsigned char *y, size_t n, int op,
                                                      int a_one, int b_one) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        const float u = a[a_one ? 0 : i], v = b[b_one ? 0 : i];
        y[i] = op == 0 ? (u == v) : op == 1 ? (u > v) : (u >= v);
    }
}

// layer.Where (layer.py:206): np.where(mask, x1, x2); x1 / x2 may be single values
__global__ void __launch_bounds__(TPB) where_kernel(const unsigned char *m, const float *a, const float *b, float *y,
                                                    size_t n, int a_one, int b_one) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) y[i] = m[i] ? a[a_one ? 0 : i] : b[b_one ? 0 : i];
}

// layer.Cast (layer.py:200): x.astype(dtype) between float32 / int32 / int64 / bool (numpy truncates
// float -> int towards zero; anything non-zero is True)
template <class S>
__device__ __forceinline__ void cast_store(void *dst, size_t i, int dt, S v) {
    if (dt == 0) reinterpret_cast<float *>(dst)[i] = (float)v;
    else if (dt == 1) reinterpret_cast<int *>(dst)[i] = (int)v;
    else if (dt == 2) reinterpret_cast<long long *>(dst)[i] = (long long)v;
    else reinterpret_cast<unsigned char *>(dst)[i] = v != (S)0;
}
__global__ void __launch_bounds__(TPB) cast_kernel(const void *src, void *dst, size_t n, int st, int dt) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        if (st == 0) cast_store(dst, i, dt, reinterpret_cast<const float *>(src)[i]);
        else if (st == 1) cast_store(dst, i, dt, reinterpret_cast<const int *>(src)[i]);
        else if (st == 2) cast_store(dst, i, dt, reinterpret_cast<const long long *>(src)[i]);
        else cast_store(dst, i, dt, (int)reinterpret_cast<const unsigned char *>(src)[i]);
    }
}

// layer.Gather (layer.py:157): np.take(x, idx, axis) on x viewed as (outer, axis_len, inner)
__global__ void __launch_bounds__(TPB) gather_kernel(const float *x, const int *idx, float *y, size_t n, int axis_len,
                                                     int n_idx, FastDiv divInner, FastDiv divIdx) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {      // i = (o*n_idx + k)*inner + r
        unsigned q, r, o, k;
        divInner.divmod((unsigned)i, q, r);
        divIdx.divmod(q, o, k);
        int t = idx[k];
        t = t < 0 ? t + axis_len : t;
        y[i] = x[((size_t)o * axis_len + t) * divInner.d + r];
    }
}

// layer.Erf (layer.py:253-258): x is clamped into [-2, 2] IN PLACE by multiplications with masks,
// scaled to a table index by *256 and truncated like astype('int16'); the table holds erf(i/256 - 2)
__global__ void __launch_bounds__(TPB) erf_lut_kernel(float *x, const float *lut, float *y, size_t n) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        float v = __fsub_rn(x[i], 2.f);          // x -= 2
        v = __fmul_rn(v, v < 0.f ? 1.f : 0.f);   // x *= x<0
        v = __fadd_rn(v, 4.f);                   // x += 4
        v = __fmul_rn(v, v > 0.f ? 1.f : 0.f);   // x *= x>0
        v = __fmul_rn(v, 256.f);                 // x *= 256
        x[i] = v;
        y[i] = lut[(int)(short)(int)v];
    }
}

// layer.InstanceNormalization (layer.py:214-224), in place like the reference: one wave per (n, c) row.
// mean = mean(x); var = mean((x-mean)^2); d = (var+eps)**0.5; x = x*(s/d) + (bias - s*mean/d)
__global__ void __launch_bounds__(TPB) instancenorm_kernel(float *x, const float *s, const float *b, int rows, int C,
                                                           int inner, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = TPB / 64;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        float *p = x + (size_t)row * inner;
        float sum = 0.f;
        for (int i = lane; i < inner; i += 64) sum += p[i];
        for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)inner;
        float sq = 0.f;
        for (int i = lane; i < inner; i += 64) {
            const float d = p[i] - mean;
            sq += d * d;
        }
        for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o);
        const float dev = powf(sq / (float)inner + eps, 0.5f);
        const float sc = s[row % C], k = __fdiv_rn(sc, dev), off = __fsub_rn(b[row % C], __fdiv_rn(__fmul_rn(sc, mean), dev));
        for (int i = lane; i < inner; i += 64) p[i] = __fadd_rn(__fmul_rn(p[i], k), off);
    }
}

}  // namespace

extern "C" {

int pl_compare_f32(pl_ctx *ctx, const float *a, const float *b, unsigned char *y, size_t n, int op, int a_one, int b_one) {
    PL_REQUIRE(ctx && (n == 0 || (a && b && y)), PL_EINVAL, "pl_compare_f32: null argument");
    PL_REQUIRE(op >= 0 && op <= 2, PL_EINVAL, "pl_compare_f32: bad op %d", op);
    if (!n) return PL_OK;
    CtxGuard g(ctx);
    compare_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(a, b, y, n, op, a_one, b_one);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_where_f32(pl_ctx *ctx, const unsigned char *mask, const float *a, const float *b, float *y, size_t n, int a_one,
                 int b_one) {
    PL_REQUIRE(ctx && (n == 0 || (mask && a && b && y)), PL_EINVAL, "pl_where_f32: null argument");
    if (!n) return PL_OK;
    CtxGuard g(ctx);
    where_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(mask, a, b, y, n, a_one, b_one);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_cast(pl_ctx *ctx, const void *src, void *dst, size_t n, int src_type, int dst_type) {
    PL_REQUIRE(ctx && (n == 0 || (src && dst)), PL_EINVAL, "pl_cast: null argument");
    PL_REQUIRE(src_type >= 0 && src_type <= 3 && dst_type >= 0 && dst_type <= 3, PL_EUNSUPPORTED,
               "pl_cast: types are 0 float32, 1 int32, 2 int64, 3 bool");
    if (!n) return PL_OK;
    CtxGuard g(ctx);
    cast_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(src, dst, n, src_type, dst_type);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_gather_f32(pl_ctx *ctx, const float *x, const int *idx, float *y, int outer, int axis_len, int inner, int n_idx) {
    PL_REQUIRE(ctx && x && idx && y, PL_EINVAL, "pl_gather_f32: null argument");
    PL_REQUIRE(outer >= 0 && axis_len > 0 && inner > 0 && n_idx >= 0, PL_EINVAL, "pl_gather_f32: bad shape");
    const size_t n = (size_t)outer * n_idx * inner;
    if (!n) return PL_OK;
    PL_REQUIRE(n < (1ull << 32), PL_EUNSUPPORTED, "gather: tensor too large");
    CtxGuard g(ctx);
    gather_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(x, idx, y, n, axis_len, n_idx, FastDiv(inner), FastDiv(n_idx));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_erf_lut_f32(pl_ctx *ctx, float *x, const float *lut, float *y, size_t n) {
    PL_REQUIRE(ctx && (n == 0 || (x && lut && y)), PL_EINVAL, "pl_erf_lut_f32: null argument");
    if (!n) return PL_OK;
    CtxGuard g(ctx);
    erf_lut_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(x, lut, y, n);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_instancenorm_f32(pl_ctx *ctx, float *x, const float *scale, const float *bias, int rows, int C, int inner, double eps) {
    PL_REQUIRE(ctx && x && scale && bias, PL_EINVAL, "pl_instancenorm_f32: null argument");
    PL_REQUIRE(rows >= 0 && C > 0 && inner > 0, PL_EINVAL, "pl_instancenorm_f32: bad shape");
    if (!rows) return PL_OK;
    CtxGuard g(ctx);
    instancenorm_kernel<<<stream_grid(ctx, (size_t)rows * 64), TPB, 0, ctx->stream>>>(x, scale, bias, rows, C, inner, (float)eps);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

}  // extern "C"
