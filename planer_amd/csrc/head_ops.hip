// Operators that ONNX-exported detection heads and post-processing graphs add around the convolutional
// trunk (reference layer.py:36-42, 155-157, 202-239, 253-258): comparisons, Where, Cast, Gather, Erf (the
// reference's 1025-entry lookup table), InstanceNormalization, ScatterND, NonZero, TopK and the LSTM cell.
// All HBM-bound: one pass over the operands, grid-stride loops capped near 8 blocks per CU; the reductions
// (instance norm, the NonZero scan) are wave64 shuffle trees, TopK sorts a row in LDS.
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

// layer.Scatternd (layer.py:208-212): data[tuple(indices[0, i])] = updates[0, i] for i = 0, 1, ... in order.
// The host turns the index tuples into row offsets and drops all but the LAST write to a row (sequential
// semantics without a race); this kernel copies the surviving update rows.
__global__ void __launch_bounds__(TPB) scatter_rows_kernel(float *dst, const long long *dst_row, const float *src,
                                                           const int *src_row, size_t n, FastDiv divRow) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {      // i = j*row_len + r
        unsigned j, r;
        divRow.divmod((unsigned)i, j, r);
        dst[(size_t)dst_row[j] * divRow.d + r] = src[(size_t)src_row[j] * divRow.d + r];
    }
}

// layer.NonZero (layer.py:230): np.array(np.nonzero(x)) -- the coordinates of the non-zero elements in
// row-major order, one row per axis, int64.  Pass 1 counts per 2048-element block, one workgroup turns the
// counts into offsets, pass 2 writes each block's coordinates at its offset (order inside a block from a
// wave ballot + shuffle scan).
constexpr int NZ_PER_THREAD = 8, NZ_BLOCK = TPB * NZ_PER_THREAD;      // = PL_NONZERO_BLOCK
static_assert(NZ_BLOCK == PL_NONZERO_BLOCK, "include/planer_hip.h states the scratch size");
__device__ __forceinline__ bool nz_at(const void *x, size_t i, int type) {
    return type == 0 ? reinterpret_cast<const float *>(x)[i] != 0.f
         : type == 1 ? reinterpret_cast<const int *>(x)[i] != 0
         : type == 2 ? reinterpret_cast<const long long *>(x)[i] != 0 : reinterpret_cast<const unsigned char *>(x)[i] != 0;
}
__global__ void __launch_bounds__(TPB) nonzero_count_kernel(const void *x, size_t n, int type, long long *counts) {
    __shared__ int wsum[TPB / 64];
    const size_t base = (size_t)blockIdx.x * NZ_BLOCK + (size_t)threadIdx.x * NZ_PER_THREAD;
    int c = 0;
#pragma unroll
    for (int e = 0; e < NZ_PER_THREAD; ++e) c += (base + e < n) && nz_at(x, base + e, type);
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// counts[0..nb) -> exclusive offsets in place, counts[nb] = total (one workgroup)
__global__ void __launch_bounds__(1024) nonzero_scan_kernel(long long *counts, size_t nb) {
    __shared__ long long wtot[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t b0 = 0; b0 < nb; b0 += 1024) {
        const size_t i = b0 + threadIdx.x;
        const long long v = i < nb ? counts[i] : 0;
        long long inc = v;                                   // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        long long before = carry_s;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        if (i < nb) counts[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[nb] = carry_s;
}
struct NzShape {
    int ndim;
    long long dim[8];
};
__global__ void __launch_bounds__(TPB) nonzero_write_kernel(const void *x, size_t n, int type, const long long *offsets,
                                                            NzShape shp, long long *out, long long total) {
    __shared__ int wsum[TPB / 64];
    const size_t base = (size_t)blockIdx.x * NZ_BLOCK + (size_t)threadIdx.x * NZ_PER_THREAD;
    bool f[NZ_PER_THREAD];
    int c = 0;
#pragma unroll
    for (int e = 0; e < NZ_PER_THREAD; ++e) {
        f[e] = (base + e < n) && nz_at(x, base + e, type);
        c += f[e];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    long long pos = offsets[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
#pragma unroll
    for (int e = 0; e < NZ_PER_THREAD; ++e) {
        if (!f[e]) continue;
        size_t rem = base + e;
        for (int d = shp.ndim - 1; d >= 0; --d) {
            const size_t q = rem / (size_t)shp.dim[d];
            out[(size_t)d * total + pos] = (long long)(rem - q * (size_t)shp.dim[d]);
            rem = q;
        }
        ++pos;
    }
}

// layer.TopK (layer.py:234-239): idx = take(argsort(x, axis), arange(k) * -largest - (largest > 0), axis);
// values = take_along_axis(x, idx).  largest = 1 -> the k greatest in descending order; largest = 0 -> the
// reference's index list is k zeros: k copies of the smallest element (kept).  Ascending order as numpy sorts
// floats (NaN last); ties -- which numpy's unstable sort leaves unspecified -- go by ascending index.
// x is viewed as (outer, n, inner); one workgroup per (outer, inner) row.
__device__ __forceinline__ bool topk_before(float a, int ia, float b, int ib) {
    if (ia < 0) return false;                                // padding sorts after everything
    if (ib < 0) return true;
    const bool an = a != a, bn = b != b;
    if (an != bn) return bn;
    if (!an && a != b) return a < b;
    return ia < ib;
}
constexpr int TOPK_LDS_MAX = 16384;
__global__ void __launch_bounds__(TPB) topk_sort_kernel(const float *x, float *vals, long long *idx, int n, int npad, int inner,
                                                        int k, int largest) {
    extern __shared__ __attribute__((aligned(16))) float topk_smem[];
    float *key = topk_smem;
    int *id = reinterpret_cast<int *>(topk_smem + npad);
    const int o = blockIdx.x / inner, in = blockIdx.x - o * inner;
    const float *row = x + (size_t)o * n * inner + in;
    for (int i = threadIdx.x; i < npad; i += TPB) {
        key[i] = i < n ? row[(size_t)i * inner] : 0.f;
        id[i] = i < n ? i : -1;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1)
        for (int step = size >> 1; step > 0; step >>= 1) {
            for (int t = threadIdx.x; t < npad / 2; t += TPB) {
                const int lo = 2 * t - (t & (step - 1)), hi = lo + step;
                const bool up = (lo & size) == 0;               // ascending run
                const float a = key[lo], b = key[hi];
                const int ia = id[lo], ib = id[hi];
                if (topk_before(b, ib, a, ia) == up) {
                    key[lo] = b; key[hi] = a;
                    id[lo] = ib; id[hi] = ia;
                }
            }
            __syncthreads();
        }
    for (int j = threadIdx.x; j < k; j += TPB) {
        const int s = largest ? n - 1 - j : 0;
        const size_t at = ((size_t)o * k + j) * inner + in;
        vals[at] = key[s];
        idx[at] = id[s];
    }
}
// rows longer than the LDS sort holds: k rounds of "the greatest element below the one taken last"
__global__ void __launch_bounds__(TPB) topk_select_kernel(const float *x, float *vals, long long *idx, int n, int inner, int k,
                                                          int largest) {
    __shared__ float wk[TPB / 64];
    __shared__ int wi[TPB / 64];
    __shared__ float lastk;
    __shared__ int lasti;
    const int o = blockIdx.x / inner, in = blockIdx.x - o * inner;
    const float *row = x + (size_t)o * n * inner + in;
    const int rounds = largest ? k : 1;
    for (int j = 0; j < rounds; ++j) {
        float bk = 0.f;
        int bi = -1;                                           // -1: nothing yet
        const float lk = j ? lastk : 0.f;
        const int li = j ? lasti : -1;
        for (int i = threadIdx.x; i < n; i += TPB) {
            const float v = row[(size_t)i * inner];
            if (largest) {
                if (j && !topk_before(v, i, lk, li)) continue;  // must sort strictly before the last one taken
                if (bi < 0 || topk_before(bk, bi, v, i)) { bk = v; bi = i; }
            } else if (bi < 0 || topk_before(v, i, bk, bi)) { bk = v; bi = i; }
        }
        for (int off = 32; off; off >>= 1) {
            const float ok = __shfl_xor(bk, off);
            const int oi = __shfl_xor(bi, off);
            const bool take = oi >= 0 && (bi < 0 || (largest ? topk_before(bk, bi, ok, oi) : topk_before(ok, oi, bk, bi)));
            if (take) { bk = ok; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { wk[threadIdx.x >> 6] = bk; wi[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < TPB / 64; ++w) {
                const bool take = wi[w] >= 0 && (bi < 0 || (largest ? topk_before(bk, bi, wk[w], wi[w]) : topk_before(wk[w], wi[w], bk, bi)));
                if (take) { bk = wk[w]; bi = wi[w]; }
            }
            lastk = bk; lasti = bi;
            if (largest) {
                const size_t at = ((size_t)o * k + j) * inner + in;
                vals[at] = bk; idx[at] = bi;
            }
        }
        __syncthreads();
    }
    if (!largest)
        for (int j = threadIdx.x; j < k; j += TPB) {
            const size_t at = ((size_t)o * k + j) * inner + in;
            vals[at] = lastk; idx[at] = lasti;
        }
}

// One step of util.lstm (util.py:109-118) after the two GEMMs: gates = ((x_t W^T + h R^T) + b[:4H]) + b[4H:],
// split i, o, f, c (ONNX order); sigmoid(i), sigmoid(f), tanh(c); C = f*c_prev + i*c; h = sigmoid(o) * tanh(C).
// Products and sums are rounded one at a time, as numpy does them.
__device__ __forceinline__ float ref_sigmoid(float v) { return __fdiv_rn(1.f, __fadd_rn(expf(-v), 1.f)); }
__global__ void __launch_bounds__(TPB) lstm_cell_kernel(const float *gx, const float *gh, const float *b, const float *c_prev,
                                                        float *h, float *c, int N, int H) {
    const int total = N * H;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < total; i += gridDim.x * TPB) {
        const int n = i / H, j = i - n * H;
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = q * H + j;
            g[q] = __fadd_rn(__fadd_rn(__fadd_rn(gx[(size_t)n * 4 * H + col], gh[(size_t)n * 4 * H + col]), b[col]), b[4 * H + col]);
        }
        const float ig = ref_sigmoid(g[0]), fg = ref_sigmoid(g[2]), cg = tanhf(g[3]);
        const float C = __fadd_rn(__fmul_rn(fg, c_prev[i]), __fmul_rn(ig, cg));
        c[i] = C;
        h[i] = __fmul_rn(ref_sigmoid(g[1]), tanhf(C));
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

int pl_scatter_rows_f32(pl_ctx *ctx, float *dst, const long long *dst_row, const float *src, const int *src_row, int n_rows,
                        int row_len) {
    PL_REQUIRE(ctx && (n_rows == 0 || (dst && dst_row && src && src_row)), PL_EINVAL, "pl_scatter_rows_f32: null argument");
    PL_REQUIRE(n_rows >= 0 && row_len > 0, PL_EINVAL, "pl_scatter_rows_f32: bad shape");
    const size_t n = (size_t)n_rows * row_len;
    if (!n) return PL_OK;
    PL_REQUIRE(n < (1ull << 32), PL_EUNSUPPORTED, "scatter: update too large");
    CtxGuard g(ctx);
    scatter_rows_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(dst, dst_row, src, src_row, n, FastDiv(row_len));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_nonzero_count(pl_ctx *ctx, const void *x, size_t n, int elem_type, long long *scratch, long long *total) {
    PL_REQUIRE(ctx && total && (n == 0 || (x && scratch)), PL_EINVAL, "pl_nonzero_count: null argument");
    PL_REQUIRE(elem_type >= 0 && elem_type <= 3, PL_EUNSUPPORTED, "pl_nonzero_count: types are 0 float32, 1 int32, 2 int64, 3 bool");
    *total = 0;
    if (!n) return PL_OK;
    PL_REQUIRE(!ctx->capturing, PL_EINVAL, "pl_nonzero_count during capture");
    CtxGuard g(ctx);
    const size_t nb = (n + NZ_BLOCK - 1) / NZ_BLOCK;
    PL_REQUIRE(nb < (1ull << 31), PL_EUNSUPPORTED, "nonzero: tensor too large");
    nonzero_count_kernel<<<(unsigned)nb, TPB, 0, ctx->stream>>>(x, n, elem_type, scratch);
    PL_LAUNCH_CHECK();
    nonzero_scan_kernel<<<1, 1024, 0, ctx->stream>>>(scratch, nb);
    PL_LAUNCH_CHECK();
    PL_HIP(hipMemcpyAsync(total, scratch + nb, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(hipStreamSynchronize(ctx->stream));          // the output's shape depends on the data
    return PL_OK;
}

int pl_nonzero_write(pl_ctx *ctx, const void *x, size_t n, int elem_type, const long long *scratch, const long long *shape,
                     int ndim, long long *out, long long total) {
    PL_REQUIRE(ctx && (n == 0 || (x && scratch && shape)), PL_EINVAL, "pl_nonzero_write: null argument");
    PL_REQUIRE(elem_type >= 0 && elem_type <= 3, PL_EUNSUPPORTED, "pl_nonzero_write: types are 0 float32, 1 int32, 2 int64, 3 bool");
    PL_REQUIRE(ndim >= 1 && ndim <= 8, PL_EUNSUPPORTED, "nonzero: 1 to 8 dimensions");
    if (!n || total <= 0) return PL_OK;
    PL_REQUIRE(out, PL_EINVAL, "pl_nonzero_write: null output");
    NzShape shp;
    shp.ndim = ndim;
    for (int d = 0; d < ndim; ++d) shp.dim[d] = shape[d];
    CtxGuard g(ctx);
    const size_t nb = (n + NZ_BLOCK - 1) / NZ_BLOCK;
    nonzero_write_kernel<<<(unsigned)nb, TPB, 0, ctx->stream>>>(x, n, elem_type, scratch, shp, out, total);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_topk_f32(pl_ctx *ctx, const float *x, int outer, int n, int inner, int k, int largest, float *values, long long *indices) {
    PL_REQUIRE(ctx && x && values && indices, PL_EINVAL, "pl_topk_f32: null argument");
    PL_REQUIRE(outer >= 0 && n > 0 && inner > 0 && k >= 0, PL_EINVAL, "pl_topk_f32: bad shape");
    PL_REQUIRE(largest == 0 || largest == 1, PL_EUNSUPPORTED, "topk: largest must be 0 or 1");
    PL_REQUIRE(k <= n, PL_EINVAL, "topk: k = %d exceeds the axis length %d", k, n);
    const size_t rows = (size_t)outer * inner;
    if (!rows || !k) return PL_OK;
    PL_REQUIRE(rows < (1ull << 31), PL_EUNSUPPORTED, "topk: too many rows");
    CtxGuard g(ctx);
    if (n <= TOPK_LDS_MAX) {
        int npad = 2;
        while (npad < n) npad <<= 1;
        const int lds = npad * 8;
        if (lds > 48 * 1024)
            PL_HIP(hipFuncSetAttribute((const void *)topk_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        topk_sort_kernel<<<(unsigned)rows, TPB, lds, ctx->stream>>>(x, values, indices, n, npad, inner, k, largest);
    } else {
        topk_select_kernel<<<(unsigned)rows, TPB, 0, ctx->stream>>>(x, values, indices, n, inner, k, largest);
    }
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_lstm_cell_f32(pl_ctx *ctx, const float *gates_x, const float *gates_h, const float *bias, const float *c_prev, float *h,
                     float *c, int N, int H) {
    PL_REQUIRE(ctx && gates_x && gates_h && bias && c_prev && h && c, PL_EINVAL, "pl_lstm_cell_f32: null argument");
    PL_REQUIRE(N >= 0 && H > 0 && (size_t)N * H < (1ull << 31), PL_EINVAL, "pl_lstm_cell_f32: bad shape");
    if (!N) return PL_OK;
    CtxGuard g(ctx);
    lstm_cell_kernel<<<stream_grid(ctx, (size_t)N * H), TPB, 0, ctx->stream>>>(gates_x, gates_h, bias, c_prev, h, c, N, H);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

}  // extern "C"
