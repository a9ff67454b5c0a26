// HBM-bound kernels of planer's forward pass: BatchNorm, ReLU, LeakyReLU,
// Sigmoid, Add, Maxpool/AveragePool, nearest UpSample, Concat copies, global
// average pool and the split-K combine.  Each moves its algorithmic bytes
// once: 16-byte vector loads/stores, grid capped near 8 blocks per CU with a
// grid-stride loop, wave64 shuffles for the one reduction.
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

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- unary / binary elementwise -----------------------------------------
struct OpRelu {
    __device__ float operator()(float x) const { return relu_ref(x); }
};
struct OpLeaky {
    float a, b;
    __device__ float operator()(float x) const { return leaky_ref(x, a, b); }
};
struct OpSigmoid {
    // 1/(1+exp(-x)) with an accurate expf; exp overflow gives inf -> result 0
    __device__ float operator()(float x) const { return __fdiv_rn(1.f, __fadd_rn(expf(-x), 1.f)); }
};

template <class Op>
__global__ void __launch_bounds__(TPB) unary_vec4(const float4 *x, float4 *y, size_t n4, Op op) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += stride) {
        float4 v = x[i];
        v.x = op(v.x); v.y = op(v.y); v.z = op(v.z); v.w = op(v.w);
        y[i] = v;
    }
}

template <class Op>
__global__ void __launch_bounds__(TPB) unary_scalar(const float *x, float *y, size_t n, Op op) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) y[i] = op(x[i]);
}

template <class Op>
int launch_unary(pl_ctx *ctx, const float *x, float *y, size_t n, Op op) {
    PL_REQUIRE(ctx && (n == 0 || (x && y)), PL_EINVAL, "elementwise: null argument");
    if (!n) return PL_OK;
    CtxGuard g(ctx);
    size_t n4 = 0;
    if (aligned16(x) && aligned16(y)) {
        n4 = n / 4;
        if (n4) unary_vec4<<<stream_grid(ctx, n4), TPB, 0, ctx->stream>>>((const float4 *)x, (float4 *)y, n4, op);
    }
    size_t done = n4 * 4;
    if (done < n) unary_scalar<<<stream_grid(ctx, n - done), TPB, 0, ctx->stream>>>(x + done, y + done, n - done, op);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

__global__ void __launch_bounds__(TPB) add_vec4(const float4 *a, const float4 *b, float4 *y, size_t n4) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += stride) {
        float4 u = a[i], v = b[i];
        y[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

__global__ void __launch_bounds__(TPB) add_scalar(const float *a, const float *b, float *y, size_t n) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) y[i] = a[i] + b[i];
}

// ---- per-channel affine (BatchNorm) and channel-broadcast add -------------
// x viewed as (outer, C, inner); one block row per (outer, c) plane chunk so
// the channel lookup is uniform per block instead of a division per element.
template <bool HAS_SCALE>
__global__ void __launch_bounds__(TPB) affine_plane(const float *x, float *y, const float *scale,
                                                    const float *shift, int C, int inner, FastDiv divC) {
    // blockIdx.y = plane (outer*C + c), blockIdx.x strides over the plane
    const unsigned plane = blockIdx.y;
    const unsigned c = plane - divC.div(plane) * (unsigned)C;
    const float s = HAS_SCALE ? scale[c] : 1.f;
    const float t = shift ? shift[c] : 0.f;
    const float *xp = x + (size_t)plane * inner;
    float *yp = y + (size_t)plane * inner;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < inner; i += gridDim.x * TPB) {
        float v = xp[i];
        if (HAS_SCALE) v = __fmul_rn(v, s);
        yp[i] = __fadd_rn(v, t);
    }
}

// flat form for small planes (e.g. 7x7): 4 consecutive elements per thread
template <bool HAS_SCALE>
__global__ void __launch_bounds__(TPB) affine_flat(const float *x, float *y, const float *scale,
                                                   const float *shift, size_t n, int C, FastDiv divInner,
                                                   FastDiv divC) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        unsigned plane = divInner.div((unsigned)i);
        unsigned c = plane - divC.div(plane) * (unsigned)C;
        float v = x[i];
        if (HAS_SCALE) v = __fmul_rn(v, scale[c]);
        y[i] = __fadd_rn(v, shift ? shift[c] : 0.f);
    }
}

template <bool HAS_SCALE>
int launch_affine(pl_ctx *ctx, const float *x, float *y, const float *scale, const float *shift,
                  int outer, int C, int inner) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "affine: null argument");
    PL_REQUIRE(outer >= 0 && C > 0 && inner > 0, PL_EINVAL, "affine: bad shape");
    size_t n = (size_t)outer * C * inner;
    if (!n) return PL_OK;
    PL_REQUIRE(n < (1ull << 32), PL_EUNSUPPORTED, "affine: tensor too large");
    CtxGuard g(ctx);
    size_t planes = (size_t)outer * C;
    if (inner >= 1024 && planes <= 65535) {
        dim3 grid(cdiv(inner, TPB * 4) ? cdiv(inner, TPB * 4) : 1, (unsigned)planes);
        affine_plane<HAS_SCALE><<<grid, TPB, 0, ctx->stream>>>(x, y, scale, shift, C, inner, FastDiv(C));
    } else {
        affine_flat<HAS_SCALE><<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(x, y, scale, shift, n, C,
                                                                         FastDiv(inner), FastDiv(C));
    }
    PL_LAUNCH_CHECK();
    return PL_OK;
}

// ---- pooling ---------------------------------------------------------------
// util.pool (util.py:79-92): zero padding; max accumulator starts at -1e4.
template <int MODE>
__global__ void __launch_bounds__(TPB) pool2d_kernel(const float *x, float *y, unsigned total, int H, int W,
                                                     int Ho, int Wo, int kh, int kw, int sh, int sw, int pt,
                                                     int pl, FastDiv divWo, FastDiv divHo) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned row, ow, nc, oh;
        divWo.divmod(i, row, ow);
        divHo.divmod(row, nc, oh);
        const float *xp = x + (size_t)nc * H * W;
        float acc = MODE == 0 ? -1e4f : 0.f;
        int h0 = (int)oh * sh - pt, w0 = (int)ow * sw - pl;
        for (int r = 0; r < kh; ++r) {
            int hi = h0 + r;
            bool hok = (unsigned)hi < (unsigned)H;
            for (int q = 0; q < kw; ++q) {
                int wi = w0 + q;
                float v = (hok && (unsigned)wi < (unsigned)W) ? xp[(size_t)hi * W + wi] : 0.f;
                acc = MODE == 0 ? fmaxf(v, acc) : acc + v;
            }
        }
        if (MODE == 1) acc = __fdiv_rn(acc, (float)(kh * kw));
        y[i] = acc;
    }
}

// 3x3 / stride 2 / pad 1 on an even-sized map (ResNet's stem pool): one thread
// produces 4 neighbouring outputs from 3 rows x (two aligned float4 + one scalar),
// 9 load instructions instead of 36, one float4 store.  Same semantics as above:
// padding contributes 0, accumulator starts at -1e4 (util.py:88,95).
__global__ void __launch_bounds__(TPB) maxpool_k3s2p1_x4(const float *x, float *y, unsigned total4, int H, int W,
                                                         int Ho, int Wo4, FastDiv divWo4, FastDiv divHo) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total4; i += stride) {
        unsigned row, q, nc, oh;
        divWo4.divmod(i, row, q);
        divHo.divmod(row, nc, oh);
        const int wi0 = (int)q * 8;                         // first input column of the aligned pair
        const float *xp = x + (size_t)nc * H * W + wi0;
        float4 acc = make_float4(-1e4f, -1e4f, -1e4f, -1e4f);
        const bool top_pad = oh == 0;                       // row -1 is padding
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = (int)oh * 2 - 1 + r;
            if (hi < 0) continue;
            const float *rp = xp + (size_t)hi * W;
            const float4 a = *reinterpret_cast<const float4 *>(rp);
            const float4 b = *reinterpret_cast<const float4 *>(rp + 4);
            const float l = q ? rp[-1] : 0.f;               // column -1 is padding
            acc.x = fmaxf(acc.x, fmaxf(l, fmaxf(a.x, a.y)));
            acc.y = fmaxf(acc.y, fmaxf(a.y, fmaxf(a.z, a.w)));
            acc.z = fmaxf(acc.z, fmaxf(a.w, fmaxf(b.x, b.y)));
            acc.w = fmaxf(acc.w, fmaxf(b.y, fmaxf(b.z, b.w)));
        }
        if (top_pad) {
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        }
        *reinterpret_cast<float4 *>(y + (size_t)row * (Wo4 * 4) + q * 4) = acc;
    }
}

// ---- nearest upsample --------------------------------------------------------
__global__ void __launch_bounds__(TPB) upsample_kernel(const float *x, float *y, unsigned total, int H, int W,
                                                       int OH, int OW, FastDiv divOW, FastDiv divOH,
                                                       FastDiv divFh, FastDiv divFw) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned row, ow, nc, oh;
        divOW.divmod(i, row, ow);
        divOH.divmod(row, nc, oh);
        y[i] = x[((size_t)nc * H + divFh.div(oh)) * W + divFw.div(ow)];
    }
}

// ---- strided row copy (concat) -------------------------------------------------
__global__ void __launch_bounds__(TPB) copy2d_vec4(float4 *dst, size_t dst_pitch4, const float4 *src,
                                                   size_t src_pitch4, unsigned width4, unsigned total4,
                                                   FastDiv divW) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total4; i += stride) {
        unsigned r, c;
        divW.divmod(i, r, c);
        dst[(size_t)r * dst_pitch4 + c] = src[(size_t)r * src_pitch4 + c];
    }
}

__global__ void __launch_bounds__(TPB) copy2d_scalar(float *dst, size_t dst_pitch, const float *src,
                                                     size_t src_pitch, unsigned width, unsigned total,
                                                     FastDiv divW) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned r, c;
        divW.divmod(i, r, c);
        dst[(size_t)r * dst_pitch + c] = src[(size_t)r * src_pitch + c];
    }
}

// ---- global average pool: one wave64 per (n,c) row ------------------------------
__global__ void __launch_bounds__(TPB) gap_kernel(const float *x, float *y, int rows, int inner, float inv) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        const float *xp = x + (size_t)r * inner;
        float s = 0.f;
        for (int i = lane; i < inner; i += 64) s += xp[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) y[r] = s * inv;
    }
}

// ---- split-K combine + fused conv tail ---------------------------------------------
__global__ void __launch_bounds__(TPB) splitk_reduce_kernel(const float *ws, int splits, size_t slab, float *y,
                                                            unsigned total, int C, FastDiv divInner,
                                                            FastDiv divC, Epilogue ep) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        float v = ws[i];
        for (int z = 1; z < splits; ++z) v += ws[(size_t)z * slab + i];
        unsigned plane = divInner.div(i);
        unsigned c = plane - divC.div(plane) * (unsigned)C;
        y[i] = apply_epilogue(ep, v, (int)c, i);
    }
}

// ---- channel-quad (Q4) variants ---------------------------------------------------
// Same semantics as the kernels above on tensors stored [N][C/4][H][W][4] (include/planer_hip.h):
// every thread handles one pixel of one channel quad, i.e. one float4 per tap.  Padding lanes of
// the last quad stay zero: max(-1e4, 0, ...) = 0, 0-sums, and the affine kernel writes them as 0.
__device__ __forceinline__ float4 f4max(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

template <int MODE>
__global__ void __launch_bounds__(TPB) pool2d_q4_kernel(const float4 *x, float4 *y, unsigned total, int H, int W,
                                                        int Ho, int Wo, int kh, int kw, int sh, int sw, int pt,
                                                        int pl, FastDiv divWo, FastDiv divHo) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned row, ow, nc, oh;
        divWo.divmod(i, row, ow);
        divHo.divmod(row, nc, oh);
        const float4 *xp = x + (size_t)nc * H * W;
        const float init = MODE == 0 ? -1e4f : 0.f;
        float4 acc = make_float4(init, init, init, init);
        const int h0 = (int)oh * sh - pt, w0 = (int)ow * sw - pl;
        for (int r = 0; r < kh; ++r) {
            const int hi = h0 + r;
            const bool hok = (unsigned)hi < (unsigned)H;
            for (int q = 0; q < kw; ++q) {
                const int wi = w0 + q;
                const float4 v = (hok && (unsigned)wi < (unsigned)W) ? xp[(size_t)hi * W + wi] : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = MODE == 0 ? f4max(v, acc) : f4add(acc, v);
            }
        }
        if (MODE == 1) {
            const float d = (float)(kh * kw);
            acc = make_float4(__fdiv_rn(acc.x, d), __fdiv_rn(acc.y, d), __fdiv_rn(acc.z, d), __fdiv_rn(acc.w, d));
        }
        y[i] = acc;
    }
}

// 3x3 / stride 2 / pad 1 max pool (ResNet's stem pool) on Q4: a thread owns TWO vertically adjacent outputs and
// reads its 5x3 input window once (15 float4 instead of 18; an input row is fetched for 2 output rows instead of
// 1.5) while consecutive lanes stay consecutive output columns -- the generic kernel above re-read 23 % of its input
// from HBM (profiles/r01_hbm_traffic.md).  Same zero padding, -1e4 start and per-output accumulation order as
// pool2d_q4_kernel<0> (util.py:79-95), so results are bit-identical.
__global__ void __launch_bounds__(TPB) maxpool_q4_k3s2p1_2x1(const float4 *x, float4 *y, unsigned total, int H, int W,
                                                             int Ho, int Wo, int Hb, unsigned x_bytes, FastDiv divWo,
                                                             FastDiv divHb) {
    const unsigned stride = gridDim.x * TPB;
    // the zero padding comes from the buffer's range check (an out-of-window lane gets an out-of-range offset):
    // all 15 loads of a thread are in flight together, no branch per element
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(x), 0, x_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {      // i = (nc*Hb + a)*Wo + ow
        unsigned row, ow, nc, a;
        divWo.divmod(i, row, ow);
        divHb.divmod(row, nc, a);
        const int h0 = 4 * (int)a - 1, w0 = 2 * (int)ow - 1;
        const int base = (int)nc * H * W;                                            // in float4s; < 2^27 (host check)
        float4 v[5][3];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int hi = h0 + r;
            const bool hok = (unsigned)hi < (unsigned)H;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int wi = w0 + q;
                const bool ok = hok && (unsigned)wi < (unsigned)W;
                v[r][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (base + hi * W + wi) << 4 : OOB, 0, 0));
            }
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int oh = 2 * (int)a + dy;
            if (oh >= Ho) continue;
            float4 acc = make_float4(-1e4f, -1e4f, -1e4f, -1e4f);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc = f4max(v[2 * dy + r][q], acc);
            y[((size_t)nc * Ho + oh) * Wo + ow] = acc;
        }
    }
}

__global__ void __launch_bounds__(TPB) upsample_q4_kernel(const float4 *x, float4 *y, unsigned total, int H, int W,
                                                          int OH, int OW, FastDiv divOW, FastDiv divOH,
                                                          FastDiv divFh, FastDiv divFw) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned row, ow, nc, oh;
        divOW.divmod(i, row, ow);
        divOH.divmod(row, nc, oh);
        y[i] = x[((size_t)nc * H + divFh.div(oh)) * W + divFw.div(ow)];
    }
}

// layer.Concatenate (layer.py:90-91) of TWO channel-quad tensors along channels in one launch, the first one optionally
// nearest-upsampled by (fh, fw) on the way (layer.UpSample, layer.py:80-82 -> util.upsample_nearest): the route layers of
// a detection net -- upsample -> concat -- are one pass over the output instead of three kernels.
__global__ void __launch_bounds__(TPB) concat2_q4_kernel(const float4 *a, const float4 *b, float4 *y, unsigned total, int qa,
                                                         int qb, int H, int W, int Ha, int Wa, FastDiv divHW, FastDiv divQ,
                                                         FastDiv divW, FastDiv divFh, FastDiv divFw) {
    const unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {      // i = (n*(qa+qb) + q)*H*W + pix
        unsigned r, pix, n, q;
        divHW.divmod(i, r, pix);
        divQ.divmod(r, n, q);
        if ((int)q < qa) {
            unsigned oh, ow;
            divW.divmod(pix, oh, ow);
            y[i] = a[(((size_t)n * qa + q) * Ha + divFh.div(oh)) * Wa + divFw.div(ow)];
        } else {
            y[i] = b[((size_t)n * qb + (q - qa)) * (size_t)(H * W) + pix];
        }
    }
}

// global average pool: one wave64 per (n, channel quad); output is plain [N][C]
__global__ void __launch_bounds__(TPB) gap_q4_kernel(const float4 *x, float *y, int rows, int Cq, int C, int inner,
                                                     float inv) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        const float4 *xp = x + (size_t)r * inner;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = lane; i < inner; i += 64) s = f4add(s, xp[i]);
        for (int off = 32; off > 0; off >>= 1) {
            s.x += __shfl_down(s.x, off, 64); s.y += __shfl_down(s.y, off, 64);
            s.z += __shfl_down(s.z, off, 64); s.w += __shfl_down(s.w, off, 64);
        }
        if (lane == 0) {
            const int n = r / Cq, cq = r - n * Cq;
            float *yp = y + (size_t)n * C + cq * 4;
            const int left = C - cq * 4;
            yp[0] = s.x * inv;
            if (left > 1) yp[1] = s.y * inv;
            if (left > 2) yp[2] = s.z * inv;
            if (left > 3) yp[3] = s.w * inv;
        }
    }
}

// BatchNorm (layer.py:125-127) on a Q4 tensor: y = x*scale[c] + shift[c], two roundings
__global__ void __launch_bounds__(TPB) affine_q4_kernel(const float4 *x, float4 *y, const float *scale,
                                                        const float *shift, unsigned total, int C, int Cq,
                                                        FastDiv divInner, FastDiv divCq) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const unsigned plane = divInner.div(i);
        const int cq = (int)(plane - divCq.div(plane) * (unsigned)Cq);
        const float4 v = x[i];
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cq * 4 + e;
            r[e] = c < C ? __fadd_rn(__fmul_rn(r[e], scale[c]), shift[c]) : 0.f;
        }
        y[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// ---- second-wave operators -----------------------------------
struct OpMath {
    int op;       // 0 exp 1 log 2 tanh 3 sqrt 4 reciprocal 5 hardsigmoid 6 clip
    float p0, p1;
    __device__ float operator()(float x) const {
        switch (op) {
            case 0: return expf(x);
            case 1: return logf(x);
            case 2: return tanhf(x);
            case 3: return sqrtf(x);   // correctly rounded (hipcc default)
            case 4: return __fdiv_rn(1.f, x);
            case 5: return fmaxf(fminf(__fadd_rn(__fmul_rn(x, p0), p1), 1.f), 0.f);   // layer.py:66-69
            default: return fmaxf(fminf(x, p1), p0);                                   // layer.py:250-251
        }
    }
};

__device__ __forceinline__ float bin_op(int op, float a, float b) {
    switch (op) {
        case 0: return __fadd_rn(a, b);
        case 1: return __fsub_rn(a, b);
        case 2: return __fmul_rn(a, b);
        case 3: return __fdiv_rn(a, b);
        default: return powf(a, b);
    }
}

// y = a (op) b with each operand either full-size, one value per channel, or a single value
__global__ void __launch_bounds__(TPB) binary_kernel(const float *a, const float *b, float *y, size_t n, int C, int op,
                                                     int amode, int bmode, FastDiv divInner, FastDiv divC) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        unsigned c = 0;
        if (amode == 1 || bmode == 1) {
            const unsigned plane = divInner.div((unsigned)i);
            c = plane - divC.div(plane) * (unsigned)C;
        }
        const float av = amode == 0 ? a[i] : amode == 1 ? a[c] : a[0];
        const float bv = bmode == 0 ? b[i] : bmode == 1 ? b[c] : b[0];
        y[i] = bin_op(op, av, bv);
    }
}

// y = a (op) b under numpy broadcasting (layer.py:93-111): the result's index, axis by axis, addresses each operand
// through its own strides -- 0 on an axis the operand is broadcast over.
struct BcastArgs {
    int ndim;
    unsigned shape[6];
    long long sa[6], sb[6];
};

__global__ void __launch_bounds__(TPB) binary_bcast_kernel(const float *a, const float *b, float *y, size_t n, int op,
                                                           BcastArgs p) {
    size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        unsigned rem = (unsigned)i;
        long long ia = 0, ib = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const unsigned q = rem / p.shape[d], t = rem - q * p.shape[d];
            rem = q;
            ia += (long long)t * p.sa[d];
            ib += (long long)t * p.sb[d];
        }
        y[i] = bin_op(op, a[ia], b[ib]);
    }
}

// layer.UpSample / Resize, mode "linear", integer factors (util.py:121-153 upsample_blinear): the map is
// edge-replicated by one pixel, every 2 x 2 neighbourhood (i, j) of it yields an fh x fw block
//   lt*w[0][a][b] + rt*w[1][a][b] + lb*w[2][a][b] + rb*w[3][a][b]
// and the result is that field cropped by fh/2, fw/2.  The weights are the reference's float16 table, handed in
// by the host as floats.  With a factor of 1 on one axis only the other axis is interpolated (two terms).
struct UpLinArgs {
    int H, W, fh, fw, terms;   // terms: 4 both axes, 2 one axis
    float w[4 * 64];
};

__global__ void __launch_bounds__(TPB) upsample_linear_kernel(const float *x, float *y, unsigned total, UpLinArgs p,
                                                              FastDiv divOW, FastDiv divOH) {
    const unsigned stride = gridDim.x * TPB;
    const int kk = p.fh * p.fw;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned row, ox, plane, oy;
        divOW.divmod(i, row, ox);
        divOH.divmod(row, plane, oy);
        const float *xp = x + (size_t)plane * p.H * p.W;
        int r0 = (int)oy, r1 = (int)oy, a = 0, c0 = (int)ox, c1 = (int)ox, b = 0;
        if (p.fh > 1) {
            const int Y = (int)oy + p.fh / 2, q = Y / p.fh;
            a = Y - q * p.fh;
            r0 = max(q - 1, 0);
            r1 = min(q, p.H - 1);
        }
        if (p.fw > 1) {
            const int X = (int)ox + p.fw / 2, q = X / p.fw;
            b = X - q * p.fw;
            c0 = max(q - 1, 0);
            c1 = min(q, p.W - 1);
        }
        const float *w = p.w + a * p.fw + b;
        float v;
        if (p.terms == 4) {
            v = __fmul_rn(xp[(size_t)r0 * p.W + c0], w[0]);
            v = __fmaf_rn(xp[(size_t)r0 * p.W + c1], w[kk], v);
            v = __fmaf_rn(xp[(size_t)r1 * p.W + c0], w[2 * kk], v);
            v = __fmaf_rn(xp[(size_t)r1 * p.W + c1], w[3 * kk], v);
        } else if (p.fw > 1) {
            v = __fmul_rn(xp[(size_t)r0 * p.W + c0], w[0]);
            v = __fmaf_rn(xp[(size_t)r0 * p.W + c1], w[kk], v);
        } else {
            v = __fmul_rn(xp[(size_t)r0 * p.W + c0], w[0]);
            v = __fmaf_rn(xp[(size_t)r1 * p.W + c0], w[kk], v);
        }
        y[i] = v;
    }
}

// layer.UpSample / Resize, mode "linear", fractional factors (util.py:194-219 upsample_size) on (planes, H, W):
// columns first, a*(1-cs) + b*cs, then rows on those -- the reference's order of roundings.  Sample positions
// come from the host (float32 linspace, clip, floor).
__global__ void __launch_bounds__(TPB) resize_planes_kernel(const float *x, float *y, unsigned total, int H, int W,
                                                            const int *ra, const float *rs, const int *ca,
                                                            const float *cs, FastDiv divOW, FastDiv divOH) {
    const unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned row, ox, plane, oy;
        divOW.divmod(i, row, ox);
        divOH.divmod(row, plane, oy);
        const int r0 = ra[oy], c0 = ca[ox];
        const float fr = rs[oy], fc = cs[ox];
        const float gc = __fsub_rn(1.f, fc), gr = __fsub_rn(1.f, fr);
        const float *p0 = x + ((size_t)plane * H + r0) * W + c0, *p1 = p0 + W;
        const float top = __fadd_rn(__fmul_rn(p0[0], gc), __fmul_rn(p0[1], fc));
        const float bot = __fadd_rn(__fmul_rn(p1[0], gc), __fmul_rn(p1[1], fc));
        y[i] = __fadd_rn(__fmul_rn(top, gr), __fmul_rn(bot, fr));
    }
}

// softmax / logsoftmax over the last axis, one wave64 per row (layer.py:141-153):
// y = x - max; s = sum(exp(y)); out = exp(y - log s)  (or y - log s)
__global__ void __launch_bounds__(TPB) softmax_kernel(const float *x, float *y, int rows, int cols, int logmode) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        const float *xp = x + (size_t)r * cols;
        float *yp = y + (size_t)r * cols;
        float m = -INFINITY;
        for (int i = lane; i < cols; i += 64) m = fmaxf(m, xp[i]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float s = 0.f;
        for (int i = lane; i < cols; i += 64) s += expf(__fsub_rn(xp[i], m));
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float ls = logf(s);
        for (int i = lane; i < cols; i += 64) {
            const float t = __fsub_rn(__fsub_rn(xp[i], m), ls);
            yp[i] = logmode ? t : expf(t);
        }
    }
}

// reduce over the trailing `cols` elements of each row: 0 sum, 1 mean, 2 max, 3 min
__global__ void __launch_bounds__(TPB) reduce_rows_kernel(const float *x, float *y, int rows, int cols, int op) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        const float *xp = x + (size_t)r * cols;
        float v = op == 2 ? -INFINITY : op == 3 ? INFINITY : 0.f;
        for (int i = lane; i < cols; i += 64) {
            const float t = xp[i];
            v = op == 2 ? fmaxf(v, t) : op == 3 ? fminf(v, t) : v + t;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float t = __shfl_xor(v, off, 64);
            v = op == 2 ? fmaxf(v, t) : op == 3 ? fminf(v, t) : v + t;
        }
        if (lane == 0) y[r] = op == 1 ? __fdiv_rn(v, (float)cols) : v;
    }
}

// general permutation of up to 6 axes: out[idx_out] = in[idx_in]
struct PermArgs {
    int ndim;
    unsigned oshape[6];     // output shape
    unsigned istride[6];    // input stride (elements) of the axis that feeds output axis d
};
__global__ void __launch_bounds__(TPB) transpose_kernel(const float *x, float *y, unsigned total, PermArgs p) {
    unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned rem = i;
        size_t src = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const unsigned q = rem / p.oshape[d], r = rem - q * p.oshape[d];
            src += (size_t)r * p.istride[d];
            rem = q;
        }
        y[i] = x[src];
    }
}

// General strided map of up to 6 axes: output index o_d reads input index
//     t = o_d*step_d + start_d;  wrap_d = 1: t mod extent_d, 2: t clamped to the axis, 3 / 4: t mirrored at the borders without /
//     with the border sample (np.pad's 'wrap', 'edge', 'reflect', 'symmetric');  div_d > 1: needs t % div_d == 0, t /= div_d
// and takes `fill` whenever an axis lands outside [0, extent_d).  One kernel covers Slice (start /
// step, negative steps), constant Pad (negative start), Tile (wrap), Expand (stride 0), Split, the
// zero-stuffing of ConvTranspose2d (div = stride) and its filter flip + transpose (step -1,
// permuted strides).
struct MapArgs {
    int ndim;
    unsigned oshape[6];
    long long istride[6];
    int start[6], step[6], div[6], extent[6], wrap[6];
    float fill;
};
__global__ void __launch_bounds__(TPB) strided_map_kernel(const float *x, float *y, size_t total, MapArgs p) {
    const size_t stride = (size_t)gridDim.x * TPB;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        size_t rem = i;
        long long src = 0;
        bool ok = true;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const size_t q = rem / p.oshape[d];
            const int o = (int)(rem - q * p.oshape[d]);
            rem = q;
            int t = o * p.step[d] + p.start[d];
            const int n = p.extent[d];
            if (p.wrap[d] == 1) {                      // modulo (Tile, np.pad 'wrap')
                t %= n;
                if (t < 0) t += n;
            } else if (p.wrap[d] == 2) {               // clamp (np.pad 'edge')
                t = min(max(t, 0), n - 1);
            } else if (p.wrap[d] >= 3) {               // mirror: 3 without the border sample ('reflect'), 4 with it ('symmetric')
                const int period = p.wrap[d] == 3 ? max(2 * (n - 1), 1) : 2 * n;
                t %= period;
                if (t < 0) t += period;
                if (t >= n) t = (p.wrap[d] == 3 ? period : period - 1) - t;
            }
            if (p.div[d] > 1) {
                ok = ok && t % p.div[d] == 0;
                t /= p.div[d];
            }
            ok = ok && t >= 0 && t < p.extent[d];
            src += (long long)t * p.istride[d];
        }
        y[i] = ok ? x[src] : p.fill;
    }
}

// ---- tiled large-image inference (util.tile, util.py:291-348) -----------------------------------
// util.resize (util.py:253-269) on an H x W x C image: separable bilinear with the reference's
// order of roundings -- columns first: a*(1-cs) + b*cs, then rows on those.  The sample positions
// (ra, rs, ca, cs) are computed on the host exactly like the reference's float32 linspace.
__global__ void __launch_bounds__(TPB) resize_hwc_kernel(const float *x, float *y, unsigned total, int W, int C,
                                                         int OW, const int *ra, const float *rs, const int *ca,
                                                         const float *cs, FastDiv divC, FastDiv divOW) {
    const unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned pix, ch, oh, ow;
        divC.divmod(i, pix, ch);
        divOW.divmod(pix, oh, ow);
        const int r0 = ra[oh], c0 = ca[ow];
        const float fr = rs[oh], fc = cs[ow];
        const float gc = __fsub_rn(1.f, fc), gr = __fsub_rn(1.f, fr);
        const float *p0 = x + ((size_t)r0 * W + c0) * C + ch, *p1 = p0 + (size_t)W * C;
        const float top = __fadd_rn(__fmul_rn(p0[0], gc), __fmul_rn(p0[C], fc));
        const float bot = __fadd_rn(__fmul_rn(p1[0], gc), __fmul_rn(p1[C], fc));
        y[i] = __fadd_rn(__fmul_rn(top, gr), __fmul_rn(bot, fr));
    }
}

// One window's result into the blend buffers (util.py:333-343): weight = distance to the window
// border + 1, capped at m + 1;  buf += rst * weight, count += weight.  Launches for overlapping
// windows are ordered by the stream.
__global__ void __launch_bounds__(TPB) tile_accumulate_kernel(const float *rst, float *buf, float *count, unsigned total,
                                                              int h, int w, int C, int r0, int c0, int OW, int m,
                                                              FastDiv divC, FastDiv divW) {
    const unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        unsigned pix, ch, r, c;
        divC.divmod(i, pix, ch);
        divW.divmod(pix, r, c);
        const int dr = min((int)r, h - 1 - (int)r), dc = min((int)c, w - 1 - (int)c);
        const float wt = (float)(min(min(dr, dc), m) + 1);
        const size_t o = (size_t)(r0 + (int)r) * OW + (c0 + (int)c);
        buf[o * C + ch] = __fadd_rn(buf[o * C + ch], __fmul_rn(rst[i], wt));
        if (ch == 0) count[o] = __fadd_rn(count[o], wt);
    }
}

__global__ void __launch_bounds__(TPB) tile_normalise_kernel(float *buf, const float *count, unsigned total, FastDiv divC) {
    const unsigned stride = gridDim.x * TPB;
    for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < total; i += stride) buf[i] = __fdiv_rn(buf[i], count[divC.div(i)]);
}

}  // namespace

extern "C" {

int pl_resize_hwc_f32(pl_ctx *ctx, const float *x, float *y, int H, int W, int C, int OH, int OW, const int *ra,
                      const float *rs, const int *ca, const float *cs) {
    PL_REQUIRE(ctx && x && y && ra && rs && ca && cs, PL_EINVAL, "pl_resize_hwc_f32: null argument");
    PL_REQUIRE(H > 1 && W > 1 && C > 0 && OH > 0 && OW > 0, PL_EINVAL, "pl_resize_hwc_f32: bad shape (needs H, W >= 2)");
    const size_t total = (size_t)OH * OW * C;
    PL_REQUIRE(total < (1ull << 32) && (size_t)H * W * C < (1ull << 32), PL_EUNSUPPORTED, "resize: image too large");
    CtxGuard g(ctx);
    resize_hwc_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(x, y, (unsigned)total, W, C, OW, ra, rs, ca, cs,
                                                                     FastDiv(C), FastDiv(OW));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_tile_accumulate_f32(pl_ctx *ctx, const float *rst, float *buf, float *count, int h, int w, int C, int r0, int c0,
                           int OH, int OW, int margin) {
    PL_REQUIRE(ctx && rst && buf && count, PL_EINVAL, "pl_tile_accumulate_f32: null argument");
    PL_REQUIRE(h > 0 && w > 0 && C > 0 && r0 >= 0 && c0 >= 0 && r0 + h <= OH && c0 + w <= OW && margin >= 0, PL_EINVAL,
               "pl_tile_accumulate_f32: window outside the output");
    const size_t total = (size_t)h * w * C;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "tile: window too large");
    CtxGuard g(ctx);
    tile_accumulate_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(rst, buf, count, (unsigned)total, h, w, C, r0,
                                                                          c0, OW, margin, FastDiv(C), FastDiv(w));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_tile_normalise_f32(pl_ctx *ctx, float *buf, const float *count, int OH, int OW, int C) {
    PL_REQUIRE(ctx && buf && count && OH > 0 && OW > 0 && C > 0, PL_EINVAL, "pl_tile_normalise_f32: bad argument");
    const size_t total = (size_t)OH * OW * C;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "tile: image too large");
    CtxGuard g(ctx);
    tile_normalise_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(buf, count, (unsigned)total, FastDiv(C));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_strided_map_f32(pl_ctx *ctx, const float *x, float *y, int ndim, const int *out_shape,
                       const long long *in_stride, const int *start, const int *step, const int *div,
                       const int *extent, const int *wrap, double fill) {
    PL_REQUIRE(ctx && y && out_shape && in_stride && start && step && div && extent && wrap, PL_EINVAL,
               "pl_strided_map_f32: null argument");
    PL_REQUIRE(ndim >= 1 && ndim <= 6, PL_EUNSUPPORTED, "pl_strided_map_f32: 1..6 axes supported, got %d", ndim);
    MapArgs p;
    p.ndim = ndim;
    p.fill = (float)fill;
    size_t total = 1;
    for (int d = 0; d < ndim; ++d) {
        PL_REQUIRE(out_shape[d] >= 0 && extent[d] >= 0 && div[d] >= 1, PL_EINVAL, "pl_strided_map_f32: bad axis %d", d);
        PL_REQUIRE(wrap[d] >= 0 && wrap[d] <= 4, PL_EINVAL, "pl_strided_map_f32: boundary mode %d of axis %d (0 fill, 1 wrap, 2 edge, 3 reflect, 4 symmetric)", wrap[d], d);
        PL_REQUIRE(!wrap[d] || extent[d] > 0, PL_EINVAL, "pl_strided_map_f32: wrap on an empty axis");
        p.oshape[d] = (unsigned)out_shape[d];
        p.istride[d] = in_stride[d];
        p.start[d] = start[d]; p.step[d] = step[d]; p.div[d] = div[d]; p.extent[d] = extent[d]; p.wrap[d] = wrap[d];
        total *= (size_t)out_shape[d];
    }
    if (!total) return PL_OK;
    PL_REQUIRE(x, PL_EINVAL, "pl_strided_map_f32: null input");
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "pl_strided_map_f32: tensor too large");
    CtxGuard g(ctx);
    strided_map_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(x, y, total, p);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_unary_f32(pl_ctx *ctx, const float *x, float *y, size_t n, int op, double p0, double p1) {
    PL_REQUIRE(op >= 0 && op <= 6, PL_EINVAL, "pl_unary_f32: bad op %d", op);
    return launch_unary(ctx, x, y, n, OpMath{op, (float)p0, (float)p1});
}

int pl_binary_f32(pl_ctx *ctx, const float *a, const float *b, float *y, int outer, int C, int inner, int op,
                  int a_mode, int b_mode) {
    PL_REQUIRE(ctx && a && b && y, PL_EINVAL, "pl_binary_f32: null argument");
    PL_REQUIRE(op >= 0 && op <= 4 && a_mode >= 0 && a_mode <= 2 && b_mode >= 0 && b_mode <= 2, PL_EINVAL,
               "pl_binary_f32: bad op / broadcast mode");
    PL_REQUIRE(outer >= 0 && C > 0 && inner > 0, PL_EINVAL, "pl_binary_f32: bad shape");
    const size_t n = (size_t)outer * C * inner;
    if (!n) return PL_OK;
    PL_REQUIRE(n < (1ull << 32), PL_EUNSUPPORTED, "binary op: tensor too large");
    CtxGuard g(ctx);
    binary_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(a, b, y, n, C, op, a_mode, b_mode, FastDiv(inner),
                                                              FastDiv(C));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_binary_bcast_f32(pl_ctx *ctx, const float *a, const float *b, float *y, int ndim, const int *shape,
                        const long long *a_stride, const long long *b_stride, int op) {
    PL_REQUIRE(ctx && shape && a_stride && b_stride, PL_EINVAL, "pl_binary_bcast_f32: null argument");
    PL_REQUIRE(ndim >= 1 && ndim <= 6, PL_EUNSUPPORTED, "pl_binary_bcast_f32: 1..6 axes supported, got %d", ndim);
    PL_REQUIRE(op >= 0 && op <= 4, PL_EINVAL, "pl_binary_bcast_f32: bad op %d", op);
    BcastArgs p;
    p.ndim = ndim;
    size_t n = 1;
    for (int d = 0; d < ndim; ++d) {
        PL_REQUIRE(shape[d] >= 0 && a_stride[d] >= 0 && b_stride[d] >= 0, PL_EINVAL, "pl_binary_bcast_f32: bad axis %d", d);
        p.shape[d] = (unsigned)shape[d];
        p.sa[d] = a_stride[d];
        p.sb[d] = b_stride[d];
        n *= (size_t)shape[d];
    }
    if (!n) return PL_OK;
    PL_REQUIRE(a && b && y, PL_EINVAL, "pl_binary_bcast_f32: null tensor");
    PL_REQUIRE(n < (1ull << 32), PL_EUNSUPPORTED, "binary op: tensor too large");
    CtxGuard g(ctx);
    binary_bcast_kernel<<<stream_grid(ctx, n), TPB, 0, ctx->stream>>>(a, b, y, n, op, p);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_upsample_linear_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W, int fh, int fw,
                           const float *weights) {
    PL_REQUIRE(ctx && x && y && weights, PL_EINVAL, "pl_upsample_linear_f32: null argument");
    PL_REQUIRE(NC >= 0 && H > 0 && W > 0 && fh > 0 && fw > 0, PL_EINVAL, "pl_upsample_linear_f32: bad shape");
    PL_REQUIRE(fh * fw > 1, PL_EINVAL, "pl_upsample_linear_f32: factors 1 x 1 are the identity");
    PL_REQUIRE(fh * fw <= 64, PL_EUNSUPPORTED, "pl_upsample_linear_f32: fh * fw <= 64 supported, got %d", fh * fw);
    const size_t total = (size_t)NC * H * fh * W * fw;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "upsample: tensor too large");
    UpLinArgs p;
    p.H = H; p.W = W; p.fh = fh; p.fw = fw;
    p.terms = (fh > 1 && fw > 1) ? 4 : 2;
    for (int i = 0; i < 4 * 64; ++i) p.w[i] = i < p.terms * fh * fw ? weights[i] : 0.f;
    CtxGuard g(ctx);
    upsample_linear_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(x, y, (unsigned)total, p, FastDiv(W * fw),
                                                                         FastDiv(H * fh));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_resize_linear_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W, int OH, int OW, const int *ra,
                         const float *rs, const int *ca, const float *cs) {
    PL_REQUIRE(ctx && x && y && ra && rs && ca && cs, PL_EINVAL, "pl_resize_linear_f32: null argument");
    PL_REQUIRE(NC >= 0 && H > 1 && W > 1 && OH > 0 && OW > 0, PL_EINVAL, "pl_resize_linear_f32: bad shape (needs H, W >= 2)");
    const size_t total = (size_t)NC * OH * OW;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32) && (size_t)NC * H * W < (1ull << 32), PL_EUNSUPPORTED, "resize: tensor too large");
    CtxGuard g(ctx);
    resize_planes_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(x, y, (unsigned)total, H, W, ra, rs, ca, cs,
                                                                       FastDiv(OW), FastDiv(OH));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_softmax_f32(pl_ctx *ctx, const float *x, float *y, int rows, int cols, int log_softmax) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "pl_softmax_f32: null argument");
    PL_REQUIRE(rows >= 0 && cols > 0, PL_EINVAL, "pl_softmax_f32: bad shape");
    if (!rows) return PL_OK;
    CtxGuard g(ctx);
    softmax_kernel<<<stream_grid(ctx, (size_t)rows * 64), TPB, 0, ctx->stream>>>(x, y, rows, cols, log_softmax);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_reduce_f32(pl_ctx *ctx, const float *x, float *y, int rows, int cols, int op) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "pl_reduce_f32: null argument");
    PL_REQUIRE(rows >= 0 && cols > 0 && op >= 0 && op <= 3, PL_EINVAL, "pl_reduce_f32: bad argument");
    if (!rows) return PL_OK;
    CtxGuard g(ctx);
    reduce_rows_kernel<<<stream_grid(ctx, (size_t)rows * 64), TPB, 0, ctx->stream>>>(x, y, rows, cols, op);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_transpose_f32(pl_ctx *ctx, const float *x, float *y, int ndim, const int *shape, const int *perm) {
    PL_REQUIRE(ctx && x && y && shape && perm, PL_EINVAL, "pl_transpose_f32: null argument");
    PL_REQUIRE(ndim >= 1 && ndim <= 6, PL_EUNSUPPORTED, "pl_transpose_f32: 1..6 axes");
    size_t istr[6], total = 1;
    for (int d = ndim - 1; d >= 0; --d) {
        PL_REQUIRE(shape[d] >= 0 && perm[d] >= 0 && perm[d] < ndim, PL_EINVAL, "pl_transpose_f32: bad shape/perm");
        istr[d] = total;
        total *= (size_t)shape[d];
    }
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "transpose: tensor too large");
    PermArgs p;
    p.ndim = ndim;
    for (int d = 0; d < ndim; ++d) {
        p.oshape[d] = (unsigned)shape[perm[d]];
        p.istride[d] = (unsigned)istr[perm[d]];
    }
    CtxGuard g(ctx);
    transpose_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(x, y, (unsigned)total, p);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_relu_f32(pl_ctx *ctx, const float *x, float *y, size_t n) { return launch_unary(ctx, x, y, n, OpRelu{}); }

int pl_leakyrelu_f32(pl_ctx *ctx, const float *x, float *y, size_t n, double alpha) {
    // layer.py:49: a = array(alpha, f32), b = array(1-alpha, f32)
    return launch_unary(ctx, x, y, n, OpLeaky{(float)alpha, (float)(1.0 - alpha)});
}

int pl_sigmoid_f32(pl_ctx *ctx, const float *x, float *y, size_t n) { return launch_unary(ctx, x, y, n, OpSigmoid{}); }

int pl_add_f32(pl_ctx *ctx, const float *a, const float *b, float *y, size_t n) {
    PL_REQUIRE(ctx && (n == 0 || (a && b && y)), PL_EINVAL, "pl_add_f32: null argument");
    if (!n) return PL_OK;
    CtxGuard g(ctx);
    size_t n4 = 0;
    if (aligned16(a) && aligned16(b) && aligned16(y)) {
        n4 = n / 4;
        if (n4) add_vec4<<<stream_grid(ctx, n4), TPB, 0, ctx->stream>>>((const float4 *)a, (const float4 *)b, (float4 *)y, n4);
    }
    size_t done = n4 * 4;
    if (done < n) add_scalar<<<stream_grid(ctx, n - done), TPB, 0, ctx->stream>>>(a + done, b + done, y + done, n - done);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_scale_shift_f32(pl_ctx *ctx, const float *x, float *y, const float *scale, const float *shift,
                       int outer, int C, int inner) {
    PL_REQUIRE(scale, PL_EINVAL, "pl_scale_shift_f32: null scale");
    return launch_affine<true>(ctx, x, y, scale, shift, outer, C, inner);
}

int pl_add_channel_f32(pl_ctx *ctx, const float *a, const float *b_c, float *y, int outer, int C, int inner) {
    PL_REQUIRE(b_c, PL_EINVAL, "pl_add_channel_f32: null b");
    return launch_affine<false>(ctx, a, y, nullptr, b_c, outer, C, inner);
}

int pl_pool2d_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W, int kh, int kw, int sh, int sw,
                  int pt, int pl, int pb, int pr, int mode) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "pl_pool2d_f32: null argument");
    PL_REQUIRE(NC >= 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0, PL_EINVAL, "pl_pool2d_f32: bad shape");
    PL_REQUIRE(mode == 0 || mode == 1, PL_EINVAL, "pl_pool2d_f32: mode must be 0 (max) or 1 (avg)");
    // util.pad grows each side by pads[0]/pads[1] only (util.py:8)
    PL_REQUIRE(pt == pb && pl == pr, PL_EUNSUPPORTED, "asymmetric pads are undefined in the reference (util.py:8)");
    int Ho = (H + pt + pb - kh + sh) / sh, Wo = (W + pl + pr - kw + sw) / sw;  // util.py:84-85
    PL_REQUIRE(Ho > 0 && Wo > 0, PL_EINVAL, "pl_pool2d_f32: empty output");
    size_t total = (size_t)NC * Ho * Wo;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32) && (size_t)NC * H * W < (1ull << 32), PL_EUNSUPPORTED, "pool: tensor too large");
    CtxGuard g(ctx);
    if (mode == 0 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && pt == 1 && pl == 1 && H == 2 * Ho && W == 2 * Wo &&
        Wo % 4 == 0 && aligned16(x) && aligned16(y)) {
        const unsigned total4 = (unsigned)(total / 4);
        maxpool_k3s2p1_x4<<<stream_grid(ctx, total4), TPB, 0, ctx->stream>>>(x, y, total4, H, W, Ho, Wo / 4,
                                                                         FastDiv(Wo / 4), FastDiv(Ho));
        PL_LAUNCH_CHECK();
        return PL_OK;
    }
    unsigned grid = stream_grid(ctx, total);
    if (mode == 0)
        pool2d_kernel<0><<<grid, TPB, 0, ctx->stream>>>(x, y, (unsigned)total, H, W, Ho, Wo, kh, kw, sh, sw, pt, pl, FastDiv(Wo), FastDiv(Ho));
    else
        pool2d_kernel<1><<<grid, TPB, 0, ctx->stream>>>(x, y, (unsigned)total, H, W, Ho, Wo, kh, kw, sh, sw, pt, pl, FastDiv(Wo), FastDiv(Ho));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_upsample_nearest_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W, int fh, int fw) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "pl_upsample_nearest_f32: null argument");
    PL_REQUIRE(NC >= 0 && H > 0 && W > 0 && fh > 0 && fw > 0, PL_EINVAL, "pl_upsample_nearest_f32: bad shape");
    size_t total = (size_t)NC * H * fh * W * fw;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "upsample: tensor too large");
    CtxGuard g(ctx);
    upsample_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(x, y, (unsigned)total, H, W, H * fh, W * fw,
                                                                  FastDiv(W * fw), FastDiv(H * fh), FastDiv(fh), FastDiv(fw));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_copy2d_f32(pl_ctx *ctx, float *dst, size_t dst_pitch, const float *src, size_t src_pitch, size_t width,
                  size_t rows) {
    PL_REQUIRE(ctx && (width * rows == 0 || (dst && src)), PL_EINVAL, "pl_copy2d_f32: null argument");
    PL_REQUIRE(dst_pitch >= width && src_pitch >= width, PL_EINVAL, "pl_copy2d_f32: pitch < width");
    size_t total = width * rows;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "copy2d: tensor too large");
    CtxGuard g(ctx);
    if (aligned16(dst) && aligned16(src) && width % 4 == 0 && dst_pitch % 4 == 0 && src_pitch % 4 == 0) {
        unsigned w4 = (unsigned)(width / 4), t4 = (unsigned)(total / 4);
        copy2d_vec4<<<stream_grid(ctx, t4), TPB, 0, ctx->stream>>>((float4 *)dst, dst_pitch / 4, (const float4 *)src,
                                                                 src_pitch / 4, w4, t4, FastDiv(w4));
    } else {
        copy2d_scalar<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(dst, dst_pitch, src, src_pitch, (unsigned)width,
                                                                     (unsigned)total, FastDiv((unsigned)width));
    }
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_gap_f32(pl_ctx *ctx, const float *x, float *y, int rows, int inner) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "pl_gap_f32: null argument");
    PL_REQUIRE(rows >= 0 && inner > 0, PL_EINVAL, "pl_gap_f32: bad shape");
    if (!rows) return PL_OK;
    CtxGuard g(ctx);
    gap_kernel<<<stream_grid(ctx, (size_t)rows * 64), TPB, 0, ctx->stream>>>(x, y, rows, inner, 1.f / (float)inner);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_pool2d_q4_f32(pl_ctx *ctx, const float *xq, float *yq, int N, int C, int H, int W, int kh, int kw, int sh,
                     int sw, int pt, int pl, int pb, int pr, int mode) {
    PL_REQUIRE(ctx && xq && yq, PL_EINVAL, "pl_pool2d_q4_f32: null argument");
    PL_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0, PL_EINVAL,
               "pl_pool2d_q4_f32: bad shape");
    PL_REQUIRE(mode == 0 || mode == 1, PL_EINVAL, "pl_pool2d_q4_f32: mode must be 0 (max) or 1 (avg)");
    PL_REQUIRE(pt == pb && pl == pr, PL_EUNSUPPORTED, "asymmetric pads are undefined in the reference (util.py:8)");
    PL_REQUIRE(aligned16(xq) && aligned16(yq), PL_EINVAL, "pl_pool2d_q4_f32: Q4 tensors must be 16-byte aligned");
    const int Ho = (H + pt + pb - kh + sh) / sh, Wo = (W + pl + pr - kw + sw) / sw;  // util.py:84-85
    PL_REQUIRE(Ho > 0 && Wo > 0, PL_EINVAL, "pl_pool2d_q4_f32: empty output");
    const size_t NC = (size_t)N * ((C + 3) / 4), total = NC * Ho * Wo;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 30) && NC * H * W < (1ull << 30), PL_EUNSUPPORTED, "pool: tensor too large");
    CtxGuard g(ctx);
    if (mode == 0 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && pt == 1 && pl == 1 && !getenv("PLANER_HIP_POOL_GENERIC") &&
        (size_t)N * ((C + 3) / 4) * H * W < (1ull << 27)) {
        const int Hb = (Ho + 1) / 2;
        const size_t blocks = total / ((size_t)Ho * Wo) * Hb * Wo;
        maxpool_q4_k3s2p1_2x1<<<stream_grid(ctx, blocks), TPB, 0, ctx->stream>>>((const float4 *)xq, (float4 *)yq, (unsigned)blocks, H, W,
                                                                                Ho, Wo, Hb, (unsigned)((size_t)N * ((C + 3) / 4) * H * W * 16),
                                                                                FastDiv(Wo), FastDiv(Hb));
        PL_LAUNCH_CHECK();
        return PL_OK;
    }
    const unsigned grid = stream_grid(ctx, total);
    if (mode == 0)
        pool2d_q4_kernel<0><<<grid, TPB, 0, ctx->stream>>>((const float4 *)xq, (float4 *)yq, (unsigned)total, H, W, Ho, Wo,
                                                         kh, kw, sh, sw, pt, pl, FastDiv(Wo), FastDiv(Ho));
    else
        pool2d_q4_kernel<1><<<grid, TPB, 0, ctx->stream>>>((const float4 *)xq, (float4 *)yq, (unsigned)total, H, W, Ho, Wo,
                                                         kh, kw, sh, sw, pt, pl, FastDiv(Wo), FastDiv(Ho));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_upsample_nearest_q4_f32(pl_ctx *ctx, const float *xq, float *yq, int N, int C, int H, int W, int fh, int fw) {
    PL_REQUIRE(ctx && xq && yq, PL_EINVAL, "pl_upsample_nearest_q4_f32: null argument");
    PL_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && fh > 0 && fw > 0, PL_EINVAL, "pl_upsample_nearest_q4_f32: bad shape");
    PL_REQUIRE(aligned16(xq) && aligned16(yq), PL_EINVAL, "pl_upsample_nearest_q4_f32: Q4 tensors must be 16-byte aligned");
    const size_t total = (size_t)N * ((C + 3) / 4) * H * fh * W * fw;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 30), PL_EUNSUPPORTED, "upsample: tensor too large");
    CtxGuard g(ctx);
    upsample_q4_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(
        (const float4 *)xq, (float4 *)yq, (unsigned)total, H, W, H * fh, W * fw, FastDiv(W * fw), FastDiv(H * fh),
        FastDiv(fh), FastDiv(fw));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_concat2_q4_f32(pl_ctx *ctx, const float *aq, const float *bq, float *yq, int N, int Ca, int Cb, int H, int W, int fh,
                      int fw) {
    PL_REQUIRE(ctx && aq && bq && yq, PL_EINVAL, "pl_concat2_q4_f32: null argument");
    PL_REQUIRE(N >= 0 && Ca > 0 && Cb > 0 && H > 0 && W > 0 && fh > 0 && fw > 0 && Ca % 4 == 0 && Cb % 4 == 0 && H % fh == 0 &&
                   W % fw == 0, PL_EINVAL, "pl_concat2_q4_f32: bad shape (channel counts must be multiples of 4, H, W of the factors)");
    PL_REQUIRE(aligned16(aq) && aligned16(bq) && aligned16(yq), PL_EINVAL, "pl_concat2_q4_f32: Q4 tensors must be 16-byte aligned");
    const size_t total = (size_t)N * ((Ca + Cb) / 4) * H * W;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 30), PL_EUNSUPPORTED, "concat: tensor too large");
    CtxGuard g(ctx);
    concat2_q4_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(
        (const float4 *)aq, (const float4 *)bq, (float4 *)yq, (unsigned)total, Ca / 4, Cb / 4, H, W, H / fh, W / fw,
        FastDiv(H * W), FastDiv((Ca + Cb) / 4), FastDiv(W), FastDiv(fh), FastDiv(fw));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_gap_q4_f32(pl_ctx *ctx, const float *xq, float *y, int N, int C, int HW) {
    PL_REQUIRE(ctx && xq && y, PL_EINVAL, "pl_gap_q4_f32: null argument");
    PL_REQUIRE(N >= 0 && C > 0 && HW > 0, PL_EINVAL, "pl_gap_q4_f32: bad shape");
    PL_REQUIRE(aligned16(xq), PL_EINVAL, "pl_gap_q4_f32: Q4 tensor must be 16-byte aligned");
    const int Cq = (C + 3) / 4;
    const size_t rows = (size_t)N * Cq;
    if (!rows) return PL_OK;
    PL_REQUIRE(rows < (1ull << 31), PL_EUNSUPPORTED, "gap: tensor too large");
    CtxGuard g(ctx);
    gap_q4_kernel<<<stream_grid(ctx, rows * 64), TPB, 0, ctx->stream>>>((const float4 *)xq, y, (int)rows, Cq, C, HW,
                                                                     1.f / (float)HW);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_scale_shift_q4_f32(pl_ctx *ctx, const float *xq, float *yq, const float *scale, const float *shift, int N,
                          int C, int HW) {
    PL_REQUIRE(ctx && xq && yq && scale && shift, PL_EINVAL, "pl_scale_shift_q4_f32: null argument");
    PL_REQUIRE(N >= 0 && C > 0 && HW > 0, PL_EINVAL, "pl_scale_shift_q4_f32: bad shape");
    PL_REQUIRE(aligned16(xq) && aligned16(yq), PL_EINVAL, "pl_scale_shift_q4_f32: Q4 tensors must be 16-byte aligned");
    const int Cq = (C + 3) / 4;
    const size_t total = (size_t)N * Cq * HW;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 30), PL_EUNSUPPORTED, "scale_shift: tensor too large");
    CtxGuard g(ctx);
    affine_q4_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>((const float4 *)xq, (float4 *)yq, scale, shift,
                                                                   (unsigned)total, C, Cq, FastDiv(HW), FastDiv(Cq));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_splitk_reduce_f32(pl_ctx *ctx, const float *ws, int splits, float *y, int N, int C, int inner,
                         const float *bias, const float *scale, const float *shift, const float *res, int act,
                         double alpha) {
    PL_REQUIRE(ctx && ws && y && splits >= 1, PL_EINVAL, "pl_splitk_reduce_f32: bad argument");
    size_t total = (size_t)N * C * inner;
    if (!total) return PL_OK;
    PL_REQUIRE(total < (1ull << 32), PL_EUNSUPPORTED, "splitk reduce: tensor too large");
    CtxGuard g(ctx);
    Epilogue ep = make_epilogue(bias, scale, shift, res, act, alpha);
    splitk_reduce_kernel<<<stream_grid(ctx, total), TPB, 0, ctx->stream>>>(ws, splits, total, y, (unsigned)total, C,
                                                                        FastDiv(inner), FastDiv(C), ep);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

}  // extern "C"
