// Every Winograd form of the 3x3 / stride 1 / pad 1 convolution (layer.Conv2d layer.py:22-26 -> util.conv_for util.py:17-44, with
// the fused tail of layer.py:125-127, 93-95, 44-51), one translation unit:
//   * F(2x2,3x3) on NCHW tensors and on channel-quad tensors: filter / input / output transform kernels around 16 grouped
//     GEMMs on the implicit-GEMM kernels of conv_direct.hip (plhip::conv_launch);
//   * F(4x4,3x3) on channel-quad tensors, staged: transforms (register kernels, row-split variants for small maps, the LDS
//     kernel that chains the output transform of one conv into the input transform of the next, wino4_chain_kernel.h) around
//     36 grouped GEMMs (conv_launch, or the filter-stationary kernel of wino4_gemm_as_kernel.h for 128 channels);
//   * fused 1-D F(4,3) along W (conv_w1d_kernel.h) and the fully fused F(4x4,3x3) (conv_wf4_kernel.h): one kernel each.
#include "conv_shared.h"

namespace {
using plhip::conv_launch;
using plhip::ensure_lds_attr;

// =============================================================================
// Winograd F(2x2,3x3) for 3x3 / stride 1 / pad 1 / group 1 convolutions.
// Y = A^T [ (G g G^T) .* (B^T d B) ] A  turns each 2x2 output tile into 16
// element-wise products instead of 36 MACs (2.25x fewer multiplies); summed
// over input channels the 16 "frequencies" are 16 independent GEMMs
//   M[f] (Cout x T) = U[f] (Cout x Cin) . V[f] (Cin x T),   T = N*ceil(Ho/2)*ceil(Wo/2)
// which run as ONE grouped 1x1 convolution (group = 16) on the MFMA kernel above.
// The filter transform U is made once per model; the input transform writes
// V[f][cin][tile] and the output transform reads M[f][cout][tile], applies the
// fused tail and writes NCHW.  The transforms move 4x the activation bytes, so
// this wins where activations are small next to the arithmetic (14x14, 7x7
// maps); the plan compiler times it against the direct kernel per conv.
// fp32 error of F(2,3) is a few 1e-7 relative (no large transform constants).
// =============================================================================
struct WinoArgs {
    int N, C, H, W, Cout, Ho, Wo, th, tw, T;  // th,tw = tiles per image; T = N*th*tw
    FastDiv divT, divTw, divTh;
    Epilogue ep;
};

// U[f][co][c] = (G g G^T)[f],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void __launch_bounds__(256) wino_filter_kernel(const float *w, float *U, unsigned total) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;   // (co, c) pair
    if (i >= total) return;
    const float *g = w + (size_t)i * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0;
        t[1][j] = 0.5f * (g0 + g1 + g2);
        t[2][j] = 0.5f * (g0 - g1 + g2);
        t[3][j] = g2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = t[r][0], b = t[r][1], c = t[r][2];
        U[(size_t)(r * 4 + 0) * total + i] = a;
        U[(size_t)(r * 4 + 1) * total + i] = 0.5f * (a + b + c);
        U[(size_t)(r * 4 + 2) * total + i] = 0.5f * (a - b + c);
        U[(size_t)(r * 4 + 3) * total + i] = c;
    }
}

// V[f][c][t] = (B^T d B)[f],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
__global__ void __launch_bounds__(256) wino_input_kernel(const float *x, float *V, const WinoArgs p, unsigned total) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned c, t, n, r, ty, tx;
        p.divT.divmod(i, c, t);               // i = c*T + t : consecutive lanes = consecutive tiles
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const int h0 = (int)ty * 2 - 1, w0 = (int)tx * 2 - 1;
        const float *xp = x + ((size_t)n * p.C + c) * p.H * p.W;
        float d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int hi = h0 + a;
            const bool hok = (unsigned)hi < (unsigned)p.H;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int wi = w0 + b;
                d[a][b] = (hok && (unsigned)wi < (unsigned)p.W) ? xp[(size_t)hi * p.W + wi] : 0.f;
            }
        }
        float m[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            m[0][b] = d[0][b] - d[2][b];
            m[1][b] = d[1][b] + d[2][b];
            m[2][b] = d[2][b] - d[1][b];
            m[3][b] = d[1][b] - d[3][b];
        }
        const size_t plane = (size_t)p.C * p.T;
        float *vp = V + (size_t)c * p.T + t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            vp[(size_t)(a * 4 + 0) * plane] = m[a][0] - m[a][2];
            vp[(size_t)(a * 4 + 1) * plane] = m[a][1] + m[a][2];
            vp[(size_t)(a * 4 + 2) * plane] = m[a][2] - m[a][1];
            vp[(size_t)(a * 4 + 3) * plane] = m[a][1] - m[a][3];
        }
    }
}

// y = epilogue(A^T m A),  A^T = [[1,1,1,0],[0,1,-1,-1]]
__global__ void __launch_bounds__(256) wino_output_kernel(const float *M, float *y, const WinoArgs p, unsigned total) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned co, t, n, r, ty, tx;
        p.divT.divmod(i, co, t);              // i = co*T + t
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const size_t plane = (size_t)p.Cout * p.T;
        const float *mp = M + (size_t)co * p.T + t;
        float m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = mp[(size_t)(a * 4 + b) * plane];
        float s[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s[0][b] = m[0][b] + m[1][b] + m[2][b];
            s[1][b] = m[1][b] - m[2][b] - m[3][b];
        }
        const int ho = (int)ty * 2, wo = (int)tx * 2;
        const size_t obase = (((size_t)n * p.Cout + co) * p.Ho + ho) * p.Wo + wo;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (ho + a >= p.Ho) continue;
            const float y0 = s[a][0] + s[a][1] + s[a][2];
            const float y1 = s[a][1] - s[a][2] - s[a][3];
            const size_t idx = obase + (size_t)a * p.Wo;
            y[idx] = apply_epilogue(p.ep, y0, (int)co, idx);
            if (wo + 1 < p.Wo) y[idx + 1] = apply_epilogue(p.ep, y1, (int)co, idx + 1);
        }
    }
}

}  // namespace

int plhip::winograd_launch(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *U, int Cout,
                    const float *bias, float *y, const float *scale, const float *shift, const float *res, int act,
                    double alpha) {
    PL_REQUIRE(Cin % 16 == 0, PL_EINVAL, "winograd path needs Cin %% 16 == 0");
    WinoArgs p;
    p.N = N; p.C = Cin; p.H = H; p.W = W; p.Cout = Cout; p.Ho = H; p.Wo = W;
    p.th = (H + 1) / 2; p.tw = (W + 1) / 2; p.T = N * p.th * p.tw;
    const size_t vin = (size_t)16 * Cin * p.T, vout = (size_t)16 * Cout * p.T;
    PL_REQUIRE(vin < (1ull << 29) && vout < (1ull << 31) && (size_t)Cin * p.T < (1ull << 32) &&
                   (size_t)Cout * p.T < (1ull << 32), PL_EUNSUPPORTED, "winograd: tensor too large");
    p.divT = FastDiv(p.T); p.divTw = FastDiv(p.tw); p.divTh = FastDiv(p.th);
    p.ep = make_epilogue(bias, scale, shift, res, act, alpha);
    float *V = nullptr, *M = nullptr;
    int rc = pl_alloc(ctx, vin * sizeof(float), (void **)&V);
    if (rc != PL_OK) return rc;
    rc = pl_alloc(ctx, vout * sizeof(float), (void **)&M);
    if (rc != PL_OK) {
        pl_free(ctx, V);
        return rc;
    }
    const unsigned cap = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    const unsigned tin = (unsigned)((size_t)Cin * p.T), tout = (unsigned)((size_t)Cout * p.T);
    wino_input_kernel<<<std::min(cap, (tin + 255) / 256), 256, 0, ctx->stream>>>(x, V, p, tin);
    // 16 GEMMs as one grouped 1x1 conv: input (1, 16*Cin, 1, T), filters (16*Cout, Cin, 1, 1), group 16
    // (the tile axis is presented as an (N*th) x tw image: a 1x1 conv does not care, and it keeps both
    // extents under the kernel's 14-bit spatial limit)
    rc = conv_launch(ctx, V, 1, 16 * Cin, N * p.th, p.tw, U, 16 * Cout, 1, 1, nullptr, M, 1, 1, 1, 1, 0, 0, 0, 0, 16,
                     nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0, 1);
    if (rc == PL_OK) {
        wino_output_kernel<<<std::min(cap, (tout + 255) / 256), 256, 0, ctx->stream>>>(M, y, p, tout);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            pl_set_error("winograd transform launch: %s", hipGetErrorString(le));
            rc = PL_EHIP;
        }
    }
    pl_free(ctx, M);
    pl_free(ctx, V);
    ctx->last_plan = "wino2[" + ctx->last_plan + "]";
    return rc;
}

namespace {


// ---- Winograd F(2x2,3x3) on channel-quad tensors ------------------------------------------------
// Same algebra as above with every scalar replaced by the float4 of a channel quad: the input
// transform reads x Q4 and writes V as the Q4 tensor (1, 16*Cin, 1, T) = [16*Cin/4][T][4]; the 16
// GEMMs are ONE grouped (group = 16) 1x1 conv on conv_q4_kernel; the output transform reads
// M = [16*Cout/4][T][4], applies the fused tail and writes y Q4.  Needs Cin % 4 == 0 and
// Cout % 4 == 0 (a group may not split a quad).
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4sum(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Uq[f][q][co][4] (q = cin/4, zero padded to Qpad quads) = (G g G^T)[f]
__global__ void __launch_bounds__(256) wino_filter_q4_kernel(const float *w, float *Uq, unsigned total, int Cin,
                                                             int Cout, int Qpad) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;   // (co, c) pair
    if (i >= total) return;
    const int co = (int)(i / (unsigned)Cin), c = (int)(i - (unsigned)co * Cin);
    const float *g = w + (size_t)i * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0;
        t[1][j] = 0.5f * (g0 + g1 + g2);
        t[2][j] = 0.5f * (g0 - g1 + g2);
        t[3][j] = g2;
    }
    const size_t plane = (size_t)Qpad * Cout * 4;
    float *up = Uq + ((size_t)(c >> 2) * Cout + co) * 4 + (c & 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = t[r][0], b = t[r][1], cc = t[r][2];
        up[(size_t)(r * 4 + 0) * plane] = a;
        up[(size_t)(r * 4 + 1) * plane] = 0.5f * (a + b + cc);
        up[(size_t)(r * 4 + 2) * plane] = 0.5f * (a - b + cc);
        up[(size_t)(r * 4 + 3) * plane] = cc;
    }
}

__global__ void __launch_bounds__(256) wino_input_q4_kernel(const float4 *x, float4 *V, const WinoArgs p, int Cq,
                                                            unsigned total) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned cq, t, n, r, ty, tx;
        p.divT.divmod(i, cq, t);              // i = cq*T + t : consecutive lanes = consecutive tiles
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const int h0 = (int)ty * 2 - 1, w0 = (int)tx * 2 - 1;
        const float4 *xp = x + ((size_t)n * Cq + cq) * p.H * p.W;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int hi = h0 + a;
            const bool hok = (unsigned)hi < (unsigned)p.H;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int wi = w0 + b;
                d[a][b] = (hok && (unsigned)wi < (unsigned)p.W) ? xp[(size_t)hi * p.W + wi] : z;
            }
        }
        float4 m[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            m[0][b] = f4sub(d[0][b], d[2][b]);
            m[1][b] = f4sum(d[1][b], d[2][b]);
            m[2][b] = f4sub(d[2][b], d[1][b]);
            m[3][b] = f4sub(d[1][b], d[3][b]);
        }
        const size_t plane = (size_t)Cq * p.T;
        float4 *vp = V + (size_t)cq * p.T + t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            vp[(size_t)(a * 4 + 0) * plane] = f4sub(m[a][0], m[a][2]);
            vp[(size_t)(a * 4 + 1) * plane] = f4sum(m[a][1], m[a][2]);
            vp[(size_t)(a * 4 + 2) * plane] = f4sub(m[a][2], m[a][1]);
            vp[(size_t)(a * 4 + 3) * plane] = f4sub(m[a][1], m[a][3]);
        }
    }
}

__global__ void __launch_bounds__(256) wino_output_q4_kernel(const float4 *M, float4 *y, const WinoArgs p, int Coq,
                                                             unsigned total) {
    const unsigned stride = gridDim.x * 256;
    const float4 *res4 = reinterpret_cast<const float4 *>(p.ep.res);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned coq, t, n, r, ty, tx;
        p.divT.divmod(i, coq, t);             // i = coq*T + t
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const size_t plane = (size_t)Coq * p.T;
        const float4 *mp = M + (size_t)coq * p.T + t;
        float4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = mp[(size_t)(a * 4 + b) * plane];
        float4 s[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s[0][b] = f4sum(f4sum(m[0][b], m[1][b]), m[2][b]);
            s[1][b] = f4sub(f4sub(m[1][b], m[2][b]), m[3][b]);
        }
        float bs[4], sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_chan_params(p.ep, (int)coq * 4 + e, bs[e], sc[e], sh[e]);   // Cout % 4 == 0
        const float4 bias = make_float4(bs[0], bs[1], bs[2], bs[3]), scale = make_float4(sc[0], sc[1], sc[2], sc[3]);
        const float4 shift = make_float4(sh[0], sh[1], sh[2], sh[3]);
        const int ho = (int)ty * 2, wo = (int)tx * 2;
        const size_t obase = (((size_t)n * Coq + coq) * p.Ho + ho) * p.Wo + wo;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (ho + a >= p.Ho) continue;
            const float4 y0 = f4sum(f4sum(s[a][0], s[a][1]), s[a][2]);
            const float4 y1 = f4sub(f4sub(s[a][1], s[a][2]), s[a][3]);
            const size_t idx = obase + (size_t)a * p.Wo;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            y[idx] = apply_epilogue4(p.ep, bias, scale, shift, res4 ? res4[idx] : z, 4, y0);
            if (wo + 1 < p.Wo) y[idx + 1] = apply_epilogue4(p.ep, bias, scale, shift, res4 ? res4[idx + 1] : z, 4, y1);
        }
    }
}

int winograd_q4_launch(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *Uq, int Cout,
                       const float *bias, float *yq, const float *scale, const float *shift, const float *resq,
                       int act, double alpha) {
    WinoArgs p;
    p.N = N; p.C = Cin; p.H = H; p.W = W; p.Cout = Cout; p.Ho = H; p.Wo = W;
    p.th = (H + 1) / 2; p.tw = (W + 1) / 2; p.T = N * p.th * p.tw;
    const int Cq = Cin / 4, Coq = Cout / 4;
    const size_t vin = (size_t)16 * Cin * p.T, vout = (size_t)16 * Cout * p.T;
    PL_REQUIRE(vin < (1ull << 29) && vout < (1ull << 31) && (size_t)Cq * p.T < (1ull << 32) &&
                   (size_t)Coq * p.T < (1ull << 32), PL_EUNSUPPORTED, "winograd: tensor too large");
    p.divT = FastDiv(p.T); p.divTw = FastDiv(p.tw); p.divTh = FastDiv(p.th);
    p.ep = make_epilogue(bias, scale, shift, resq, act, alpha);
    float *V = nullptr, *M = nullptr;
    int rc = pl_alloc(ctx, vin * sizeof(float), (void **)&V);
    if (rc != PL_OK) return rc;
    rc = pl_alloc(ctx, vout * sizeof(float), (void **)&M);
    if (rc != PL_OK) {
        pl_free(ctx, V);
        return rc;
    }
    const unsigned cap = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    const unsigned tin = (unsigned)((size_t)Cq * p.T), tout = (unsigned)((size_t)Coq * p.T);
    wino_input_q4_kernel<<<std::min(cap, (tin + 255) / 256), 256, 0, ctx->stream>>>((const float4 *)xq, (float4 *)V, p, Cq, tin);
    rc = conv_launch(ctx, V, 1, 16 * Cin, N * p.th, p.tw, Uq, 16 * Cout, 1, 1, nullptr, M, 1, 1, 1, 1, 0, 0, 0, 0, 16,
                     nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0, 2);
    if (rc == PL_OK) {
        wino_output_q4_kernel<<<std::min(cap, (tout + 255) / 256), 256, 0, ctx->stream>>>((const float4 *)M, (float4 *)yq, p, Coq, tout);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            pl_set_error("winograd transform launch: %s", hipGetErrorString(le));
            rc = PL_EHIP;
        }
    }
    pl_free(ctx, M);
    pl_free(ctx, V);
    ctx->last_plan = "wino2[" + ctx->last_plan + "]";
    return rc;
}

// ---- Winograd F(4x4,3x3) on channel-quad tensors ----------------------------------------------------
// Same pipeline as F(2x2,3x3) above (filter transform once, input transform, ONE grouped 1x1 conv on
// conv_q4_kernel -- here 36 groups --, output transform with the fused tail) with 6x6 input tiles that
// yield 4x4 outputs: 36 products per 16 outputs = 4x fewer multiplies than the direct conv (2.25x for
// F(2,3)) and LESS transform traffic (V holds 36 values per 16 pixels instead of 16 per 4).  The
// transforms use the standard interpolation points 0, +-1, +-2, inf:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Larger constants than F(2,3): the fp32 error is a few 1e-6 of max|y| (measured in the tests, bar
// 1e-4).  Threads work on single floats of the Q4 arrays (thread = (channel quad, tile, lane)), so
// 36 values fit in registers; four neighbouring lanes form the 16-byte accesses.
__device__ __forceinline__ void w4_bt(const float (&d)[6], float (&o)[6]) {      // o = B^T d
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    o[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    o[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&o)[4]) {      // o = A^T m
    const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], t = m[3] - m[4];
    o[0] = m[0] + p + r;
    o[1] = q + 2.f * t;
    o[2] = p + 4.f * r;
    o[3] = q + 8.f * t + m[5];
}

// uq[f = 6a+b][q = cin/4][co][cin%4] = (G g G^T)[a][b], zero padded to Qpad k-quads
__global__ void __launch_bounds__(256) wino4_filter_q4_kernel(const float *w, float *Uq, unsigned total, int Cin, int Cout,
                                                              int Qpad) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;   // (co, c) pair
    if (i >= total) return;
    const int co = (int)(i / (unsigned)Cin), c = (int)(i - (unsigned)co * Cin);
    const float *g = w + (size_t)i * 9;
    float t[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0 * 0.25f;
        t[1][j] = -(g0 + g1 + g2) * (1.f / 6.f);
        t[2][j] = (-g0 + g1 - g2) * (1.f / 6.f);
        t[3][j] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[4][j] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[5][j] = g2;
    }
    const size_t plane = (size_t)Qpad * Cout * 4;
    float *up = Uq + ((size_t)(c >> 2) * Cout + co) * 4 + (c & 3);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const float g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
        up[(size_t)(a * 6 + 0) * plane] = g0 * 0.25f;
        up[(size_t)(a * 6 + 1) * plane] = -(g0 + g1 + g2) * (1.f / 6.f);
        up[(size_t)(a * 6 + 2) * plane] = (-g0 + g1 - g2) * (1.f / 6.f);
        up[(size_t)(a * 6 + 3) * plane] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        up[(size_t)(a * 6 + 4) * plane] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        up[(size_t)(a * 6 + 5) * plane] = g2;
    }
}

// V[f][cq][t][e] = (B^T d B)[f];  thread i = ((cq*T + t)*4 + e)
__global__ void __launch_bounds__(256) wino4_input_q4_kernel(const float *x, float *V, const WinoArgs p, int Cq,
                                                             unsigned total) {
    const unsigned stride = gridDim.x * 256;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (unsigned)p.N * (unsigned)Cq * (unsigned)(p.H * p.W) * 16u, 0x00020000);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const unsigned e = i & 3, it = i >> 2;
        unsigned cq, t, n, r, ty, tx;
        p.divT.divmod(it, cq, t);
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const int h0 = (int)ty * 4 - 1, w0 = (int)tx * 4 - 1;
        // the zero border comes from the buffer's range check: all 36 loads of a thread are in flight at once
        // (a predicated plain load costs a branch and a wait per element)
        const int xbase = (int)(((n * (unsigned)Cq + cq) * (unsigned)(p.H * p.W)) * 4 + e);
        float dd[6][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int wi = w0 + b;
            const bool wok = (unsigned)wi < (unsigned)p.W;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int hi = h0 + a;
                const bool ok = wok && (unsigned)hi < (unsigned)p.H;
                dd[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         xrsrc, ok ? (xbase + (hi * p.W + wi) * 4) << 2 : (int)0x80000000, 0, 0));
            }
        }
        float m[6][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {                     // columns first: m[.][b] = B^T d[.][b]
            float d[6], o[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) d[a] = dd[a][b];
            w4_bt(d, o);
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a][b] = o[a];
        }
        const size_t plane = (size_t)Cq * p.T * 4;
        float *vp = V + (size_t)it * 4 + e;
#pragma unroll
        for (int a = 0; a < 6; ++a) {                     // then rows: V[a][.] = B^T m[a][.]
            float o[6];
            w4_bt(m[a], o);
#pragma unroll
            for (int b = 0; b < 6; ++b) vp[(size_t)(a * 6 + b) * plane] = o[b];
        }
    }
}

// y = epilogue(A^T m A);  thread i = coq*T + t works on float4s (the 4 channels of a quad): 36 b128
// loads, column pass into 24 float4 registers, row pass, fused tail, one b128 store per output pixel
__device__ __forceinline__ void w4_at4(const float4 (&m)[6], float4 (&o)[4]) {
    const float4 p = f4sum(m[1], m[2]), q = f4sub(m[1], m[2]), r = f4sum(m[3], m[4]), t = f4sub(m[3], m[4]);
    o[0] = f4sum(f4sum(m[0], p), r);
    o[1] = make_float4(q.x + 2.f * t.x, q.y + 2.f * t.y, q.z + 2.f * t.z, q.w + 2.f * t.w);
    o[2] = make_float4(p.x + 4.f * r.x, p.y + 4.f * r.y, p.z + 4.f * r.z, p.w + 4.f * r.w);
    o[3] = make_float4(q.x + 8.f * t.x + m[5].x, q.y + 8.f * t.y + m[5].y, q.z + 8.f * t.z + m[5].z,
                       q.w + 8.f * t.w + m[5].w);
}

__global__ void __launch_bounds__(256) wino4_output_q4_kernel(const float4 *M, float4 *y, const WinoArgs p, int Coq,
                                                              unsigned total) {
    const unsigned stride = gridDim.x * 256;
    const unsigned out_bytes = (unsigned)p.N * (unsigned)Coq * (unsigned)(p.Ho * p.Wo) * 16u;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? out_bytes : 0u, 0x00020000);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned coq, t, n, r, ty, tx;
        p.divT.divmod(i, coq, t);
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const size_t plane = (size_t)Coq * p.T;
        const float4 *mp = M + i;
        float4 s[4][6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {                     // columns: s[.][b] = A^T m[.][b]
            float4 m[6], o[4];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = mp[(size_t)(a * 6 + b) * plane];
            w4_at4(m, o);
#pragma unroll
            for (int a = 0; a < 4; ++a) s[a][b] = o[a];
        }
        float bs[4], sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_chan_params(p.ep, (int)coq * 4 + e, bs[e], sc[e], sh[e]);   // Cout % 4 == 0
        const float4 bias = make_float4(bs[0], bs[1], bs[2], bs[3]), scale = make_float4(sc[0], sc[1], sc[2], sc[3]);
        const float4 shift = make_float4(sh[0], sh[1], sh[2], sh[3]);
        const int ho = (int)ty * 4, wo = (int)tx * 4;
        // residual and y go through range-checked buffer accesses: pixels past the map's edge get an out-of-range
        // offset (loads return 0, stores are dropped), so the 16 residual quads are all in flight before the first
        // is used and there is no branch per pixel
        int off[4][4];
        float4 rs[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const unsigned row = ((n * (unsigned)Coq + coq) * (unsigned)p.Ho + (unsigned)(ho + a)) * (unsigned)p.Wo + (unsigned)wo;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                off[a][b] = (ho + a < p.Ho && wo + b < p.Wo) ? (int)((row + b) << 4) : (int)0x80000000;
                rs[a][b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[a][b], 0, 0));
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 o[4];
            w4_at4(s[a], o);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 v = apply_epilogue4(p.ep, bias, scale, shift, rs[a][b], 4, o[b]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                       yrsrc, off[a][b], 0, 0);
            }
        }
    }
}

// ---- row-split variants for small maps (batch-1 detection nets: a 52x52x256 map is 43 workgroups of the kernels
//      above on 256 CUs, latency-bound).  blockIdx.y picks ONE row of the transformed tile: a thread then needs only
//      the operand rows with a non-zero coefficient in that row of B^T / A^T (the other loads are dead code), does
//      one row pass and 6 (input) or 4 (output) stores -- 6x / 4x the threads, the same arithmetic per element. ----
template <int A>
__device__ __forceinline__ float w4_bt_row(const float (&d)[6]) {
    if constexpr (A == 0) return 4.f * d[0] - 5.f * d[2] + d[4];
    else if constexpr (A == 1) return -4.f * (d[1] + d[2]) + d[3] + d[4];
    else if constexpr (A == 2) return 4.f * (d[1] - d[2]) - d[3] + d[4];
    else if constexpr (A == 3) return 2.f * (d[3] - d[1]) - d[2] + d[4];
    else if constexpr (A == 4) return 2.f * (d[1] - d[3]) - d[2] + d[4];
    else return 4.f * d[1] - 5.f * d[3] + d[5];
}
template <int A>
__device__ __forceinline__ void wino4_input_row(const float *x, float *V, const WinoArgs &p, int Cq, unsigned total,
                                                const __amdgpu_buffer_rsrc_t xrsrc) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const unsigned e = i & 3, it = i >> 2;
        unsigned cq, t, n, r, ty, tx;
        p.divT.divmod(it, cq, t);
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const int h0 = (int)ty * 4 - 1, w0 = (int)tx * 4 - 1;
        const int xbase = (int)(((n * (unsigned)Cq + cq) * (unsigned)(p.H * p.W)) * 4 + e);
        float m[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int wi = w0 + b;
            const bool wok = (unsigned)wi < (unsigned)p.W;
            float d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int hi = h0 + a;
                const bool ok = wok && (unsigned)hi < (unsigned)p.H;
                d[a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     xrsrc, ok ? (xbase + (hi * p.W + wi) * 4) << 2 : (int)0x80000000, 0, 0));
            }
            m[b] = w4_bt_row<A>(d);
        }
        float o[6];
        w4_bt(m, o);
        const size_t plane = (size_t)Cq * p.T * 4;
        float *vp = V + (size_t)it * 4 + e;
#pragma unroll
        for (int b = 0; b < 6; ++b) vp[(size_t)(A * 6 + b) * plane] = o[b];
    }
}
__global__ void __launch_bounds__(256) wino4_input_rows_q4_kernel(const float *x, float *V, const WinoArgs p, int Cq,
                                                                  unsigned total) {
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x), 0, (unsigned)p.N * (unsigned)Cq * (unsigned)(p.H * p.W) * 16u, 0x00020000);
    switch (blockIdx.y) {
    case 0: wino4_input_row<0>(x, V, p, Cq, total, xrsrc); break;
    case 1: wino4_input_row<1>(x, V, p, Cq, total, xrsrc); break;
    case 2: wino4_input_row<2>(x, V, p, Cq, total, xrsrc); break;
    case 3: wino4_input_row<3>(x, V, p, Cq, total, xrsrc); break;
    case 4: wino4_input_row<4>(x, V, p, Cq, total, xrsrc); break;
    default: wino4_input_row<5>(x, V, p, Cq, total, xrsrc); break;
    }
}

template <int A>
__device__ __forceinline__ float4 w4_at4_row(const float4 (&m)[6]) {
    if constexpr (A == 0) return f4sum(f4sum(m[0], f4sum(m[1], m[2])), f4sum(m[3], m[4]));
    else {
        const float4 q = f4sub(m[1], m[2]), t = f4sub(m[3], m[4]), pp = f4sum(m[1], m[2]), r = f4sum(m[3], m[4]);
        if constexpr (A == 1) return make_float4(q.x + 2.f * t.x, q.y + 2.f * t.y, q.z + 2.f * t.z, q.w + 2.f * t.w);
        else if constexpr (A == 2) return make_float4(pp.x + 4.f * r.x, pp.y + 4.f * r.y, pp.z + 4.f * r.z, pp.w + 4.f * r.w);
        else return make_float4(q.x + 8.f * t.x + m[5].x, q.y + 8.f * t.y + m[5].y, q.z + 8.f * t.z + m[5].z, q.w + 8.f * t.w + m[5].w);
    }
}
template <int A>
__device__ __forceinline__ void wino4_output_row(const float4 *M, const WinoArgs &p, int Coq, unsigned total,
                                                 const __amdgpu_buffer_rsrc_t yrsrc, const __amdgpu_buffer_rsrc_t rrsrc) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned coq, t, n, r, ty, tx;
        p.divT.divmod(i, coq, t);
        p.divTw.divmod(t, r, tx);
        p.divTh.divmod(r, n, ty);
        const size_t plane = (size_t)Coq * p.T;
        const float4 *mp = M + i;
        float4 s[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float4 m[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = mp[(size_t)(a * 6 + b) * plane];
            s[b] = w4_at4_row<A>(m);
        }
        float bs[4], sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_chan_params(p.ep, (int)coq * 4 + e, bs[e], sc[e], sh[e]);
        const float4 bias = make_float4(bs[0], bs[1], bs[2], bs[3]), scale = make_float4(sc[0], sc[1], sc[2], sc[3]);
        const float4 shift = make_float4(sh[0], sh[1], sh[2], sh[3]);
        const int ho = (int)ty * 4 + A, wo = (int)tx * 4;
        const unsigned row = ((n * (unsigned)Coq + coq) * (unsigned)p.Ho + (unsigned)ho) * (unsigned)p.Wo + (unsigned)wo;
        int off[4];
        float4 rs[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            off[b] = (ho < p.Ho && wo + b < p.Wo) ? (int)((row + b) << 4) : (int)0x80000000;
            rs[b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[b], 0, 0));
        }
        float4 o[4];
        w4_at4(s, o);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float4 v = apply_epilogue4(p.ep, bias, scale, shift, rs[b], 4, o[b]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                   yrsrc, off[b], 0, 0);
        }
    }
}
__global__ void __launch_bounds__(256) wino4_output_rows_q4_kernel(const float4 *M, float4 *y, const WinoArgs p, int Coq,
                                                                   unsigned total) {
    const unsigned out_bytes = (unsigned)p.N * (unsigned)Coq * (unsigned)(p.Ho * p.Wo) * 16u;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? out_bytes : 0u, 0x00020000);
    switch (blockIdx.y) {
    case 0: wino4_output_row<0>(M, p, Coq, total, yrsrc, rrsrc); break;
    case 1: wino4_output_row<1>(M, p, Coq, total, yrsrc, rrsrc); break;
    case 2: wino4_output_row<2>(M, p, Coq, total, yrsrc, rrsrc); break;
    default: wino4_output_row<3>(M, p, Coq, total, yrsrc, rrsrc); break;
    }
}

#include "wino4_chain_kernel.h"
#include "wino4_gemm_as_kernel.h"
#include "conv1x1_wino_in_kernel.h"
#include "wino43_kernels.h"

// ---- F(4x4,3x3) stage by stage.  winograd4_q4_launch below runs the three stages of ONE conv; the plan
//      compiler (planer_amd/plan.py chain_winograd) calls the stages itself so that consecutive
//      Winograd convs share a transform kernel. ----
int wino4_geometry(WinoArgs &p, int N, int C, int H, int W, int Cout) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.Cout = Cout; p.Ho = H; p.Wo = W;
    p.th = (H + 3) / 4; p.tw = (W + 3) / 4; p.T = N * p.th * p.tw;
    const size_t vin = (size_t)36 * C * p.T, vout = (size_t)36 * Cout * p.T;
    PL_REQUIRE(vin < (1ull << 29) && vout < (1ull << 31) && (size_t)C * p.T < (1ull << 32) &&
                   (size_t)Cout * p.T < (1ull << 32) && (size_t)N * C * H * W < (1ull << 29) &&
                   (size_t)N * Cout * H * W < (1ull << 29), PL_EUNSUPPORTED, "winograd F(4,3): tensor too large");
    p.divT = FastDiv(p.T); p.divTw = FastDiv(p.tw); p.divTh = FastDiv(p.th);
    return PL_OK;
}

// Workgroup shape of the LDS transform kernel for planes of th x tw tiles: channel quads per workgroup (0: the
// plane does not fit) -- the largest divisor of Cq that keeps the workgroup in `budget` bytes of LDS and the grid
// at one workgroup per CU or more, preferring full waves (G x tiles close to a multiple of 64).
size_t wino4_chain_lds(int G, int th, int tw, bool from_m) {
    const size_t plane = (size_t)(4 * th + 2) * 4 * (tw + 1);
    return ((size_t)G * plane + (from_m ? (size_t)36 * G * th * tw + 3 * (size_t)G : 0)) * 16;
}
int wino4_chain_pick_g(pl_ctx *ctx, int N, int Cq, int th, int tw, bool from_m) {
    static const char *g_env = getenv("PLANER_HIP_WINO_G");
    const size_t one_max = 96 * 1024, many_max = 80 * 1024;
    if (wino4_chain_lds(1, th, tw, from_m) > one_max) return 0;
    if (g_env && atoi(g_env) > 0 && Cq % atoi(g_env) == 0 && wino4_chain_lds(atoi(g_env), th, tw, from_m) <= 150 * 1024)
        return atoi(g_env);
    const int cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256, tiles = th * tw;
    int best = 1;
    double best_eff = (double)tiles / ((tiles + 63) / 64 * 64);
    for (int G = 2; G <= Cq; ++G) {
        if (Cq % G) continue;
        if (wino4_chain_lds(G, th, tw, from_m) > many_max || (long)N * (Cq / G) < cus) break;
        const double eff = (double)(G * tiles) / ((G * tiles + 63) / 64 * 64);
        if (eff >= best_eff) best = G, best_eff = eff;
    }
    return best;
}

// M (FROM_M) or x -> y and / or V through LDS (wino4_chain_kernel.h); `p` carries the tail for FROM_M
int wino4_chain_launch(pl_ctx *ctx, const float *M, const float *x, const WinoArgs &p, int C, float *y, float *V) {
    const bool from_m = M != nullptr;
    const int Cq = C / 4;
    const int G = wino4_chain_pick_g(ctx, p.N, Cq, p.th, p.tw, from_m);
    PL_REQUIRE(G > 0, PL_EUNSUPPORTED, "winograd F(4,3) LDS transforms: a %d x %d map does not fit the workgroup's LDS", p.H, p.W);
    const size_t src_bytes = from_m ? (size_t)36 * C * p.T * 4 : (size_t)p.N * C * p.H * p.W * 4;
    PL_REQUIRE(src_bytes < (1ull << 31), PL_EUNSUPPORTED, "winograd F(4,3) LDS transforms: tensor above 2 GiB");
    Wino4ChainArgs a;
    a.M = M; a.x = x; a.y = (float4 *)y; a.V = (float4 *)V;
    a.N = p.N; a.Cq = Cq; a.H = p.H; a.W = p.W; a.th = p.th; a.tw = p.tw; a.tiles = p.th * p.tw; a.T = p.T;
    a.G = G; a.gt = G * a.tiles; a.per = (a.gt + 63) / 64 * 64;
    a.R = 4 * p.th + 2; a.XP = 4 * p.tw + 2; a.S = p.tw + 1; a.plane = a.R * 4 * a.S;
    a.src_bytes = (unsigned)src_bytes;
    a.res_bytes = (from_m && p.ep.res) ? (unsigned)((size_t)p.N * C * p.H * p.W * 4) : 0u;
    a.divGt = FastDiv(a.gt); a.divTiles = FastDiv(a.tiles); a.divTw = FastDiv(a.tw); a.divPer = FastDiv(a.per);
    a.divPlane = FastDiv(a.plane); a.div4S = FastDiv(4 * a.S); a.divS = FastDiv(a.S);
    a.divHW = FastDiv(p.H * p.W); a.divW = FastDiv(p.W);
    a.ep = p.ep;
    a.ipx = (pl_experiment("xcd", 0) && p.N % 8 == 0) ? p.N / 8 : 0;      // images per XCD (0: plain (quad group, image) grid)
    static const char *bd_env = getenv("PLANER_HIP_WINO_BD");
    int bd = bd_env ? atoi(bd_env) : 384;
    bd = std::max(64, std::min(512, bd / 64 * 64));
    const size_t lds = wino4_chain_lds(G, p.th, p.tw, from_m);
    auto kern = from_m ? wino4_chain_kernel<true> : wino4_chain_kernel<false>;
    if (lds > 48 * 1024) {
        int rc = ensure_lds_attr((const void *)kern, 150 * 1024);
        if (rc != PL_OK) return rc;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(Cq / G), (unsigned)p.N), dim3((unsigned)bd), lds, ctx->stream, a);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

// PLANER_HIP_WINO_LDS: 0 = register transform kernels only (no chaining), 1 (default) = the LDS kernel for chained
// transforms, the register kernels for lone input / output transforms (measured at ResNet-18's layer2-4 shapes, batch
// 32: lone LDS transforms 12.1 / 14.9 us against 10.1 / 13.4 us), 2 = the LDS kernel for lone transforms too
int wino4_lds_mode() {
    const char *e = getenv("PLANER_HIP_WINO_LDS");       // read per call: tests switch it at run time
    return e ? atoi(e) : 1;
}
bool wino4_lds_enabled() { return wino4_lds_mode() != 0; }

int wino4_input_launch(pl_ctx *ctx, const float *xq, float *V, const WinoArgs &p, int lds_ok) {
    const int Cq = p.C / 4;
    if (lds_ok && wino4_lds_mode() >= 2 && wino4_chain_pick_g(ctx, p.N, Cq, p.th, p.tw, false) > 0)
        return wino4_chain_launch(ctx, nullptr, xq, p, p.C, nullptr, V);
    const unsigned cap = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    const unsigned tin = (unsigned)((size_t)p.C * p.T);
    // small maps: one transformed row per thread (see the row-split kernels); PLANER_HIP_WINO_ROWS=0/1 forces
    static const char *rows_env = getenv("PLANER_HIP_WINO_ROWS");
    const unsigned cus = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256);
    const bool in_rows = rows_env ? atoi(rows_env) != 0 : (tin + 255) / 256 < cus;
    if (in_rows)
        wino4_input_rows_q4_kernel<<<dim3(std::min(cap, (tin + 255) / 256), 6), 256, 0, ctx->stream>>>(xq, V, p, Cq, tin);
    else
        wino4_input_q4_kernel<<<std::min(cap, (tin + 255) / 256), 256, 0, ctx->stream>>>(xq, V, p, Cq, tin);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int wino4_gemm_launch(pl_ctx *ctx, const float *V, const float *Uq, float *M, const WinoArgs &p) {
    // 128 input channels: the filter-stationary kernel (wino4_gemm_as_kernel.h).  PLANER_HIP_WINO_GEMM_AS=0 / 1 forces.
    const char *as_env = getenv("PLANER_HIP_WINO_GEMM_AS");
    const size_t v_bytes = (size_t)36 * p.C * p.T * 4, m_bytes = (size_t)36 * p.Cout * p.T * 4;
    const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    if (p.C == 128 && p.Cout % 128 == 0 && v_bytes < (1ull << 31) && m_bytes < (1ull << 31) && ctx->conv_cfg < 0 &&
        (as_env ? atoi(as_env) != 0 : (long long)36 * (p.Cout / 128) * ((p.T + 31) / 32) >= 2LL * cus)) {
        Wino4GemmAsArgs a;
        a.U = Uq; a.V = V; a.M = M; a.Cout = p.Cout; a.T = p.T;
        const int nsub = (p.T + 31) / 32, fm = 36 * (p.Cout / 128);
        a.nt = std::max(1, (int)(((long long)fm * nsub + cus - 1) / cus));       // one round of workgroups
        a.wpf = (nsub + a.nt - 1) / a.nt;
        a.u_bytes = (unsigned)((size_t)36 * 32 * p.Cout * 16); a.v_bytes = (unsigned)v_bytes; a.m_bytes = (unsigned)m_bytes;
        hipLaunchKernelGGL(wino4_gemm_as_kernel, dim3((unsigned)(fm * a.wpf)), dim3(256), 0, ctx->stream, a);
        PL_LAUNCH_CHECK();
        ctx->last_plan = "wino4[as128x32 nt=" + std::to_string(a.nt) + " blocks=" + std::to_string(fm * a.wpf) + "]";
        ctx->last_gemm[0] = 36; ctx->last_gemm[1] = p.Cout; ctx->last_gemm[2] = (long long)nsub * 32; ctx->last_gemm[3] = 128;
        return PL_OK;
    }
    // experiment xcd=1: the tile columns of N/8 images per XCD, for every frequency (the chain kernel maps its workgroups the same
    // way, wino4_chain_launch): M and V are handed from kernel to kernel inside one XCD.  Honoured by run_plan when the launch
    // plan is one unsplit pass whose column tiles split evenly.
    ctx->xcd_cols_request = (pl_experiment("xcd", 0) && p.N % 8 == 0) ? (p.N / 8) * p.th * p.tw : 0;
    // experiment mixed_tiles=1 (TIMING ONLY, results are garbage): the GEMM shape mixed F(4,3) x F(3,3) tiles would give maps of
    // 14 or 7 pixels -- 121 frequency groups x a quarter of today's tile columns -- on scratch filters of the right size, to see what
    // removing the tile padding would be worth inside the pipelined run before building the transforms (tools/mixed_tile_probe.py)
    if (pl_experiment("mixed_tiles", 0) && (p.H == 14 || p.H == 7) && p.H == p.W) {
        static std::map<std::pair<int, size_t>, float *> scratch;
        const size_t elems = (size_t)121 * ((p.C / 4 + 7) / 8 * 8) * p.Cout * 4;
        float *&u = scratch[{ctx->device, elems}];
        if (!u && hipMalloc(&u, elems * sizeof(float)) == hipSuccess) (void)hipMemset(u, 0, elems * sizeof(float));
        if (u) {
            int rc = conv_launch(ctx, V, 1, 121 * p.C, p.N * p.th / 4, p.tw, u, 121 * p.Cout, 1, 1, nullptr, M, 1, 1, 1, 1, 0, 0, 0, 0, 121,
                                 nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0, 2);
            ctx->xcd_cols_request = 0;
            ctx->last_plan = "wino4-mixed-probe[" + ctx->last_plan + "]";
            return rc;
        }
    }
    int rc = conv_launch(ctx, V, 1, 36 * p.C, p.N * p.th, p.tw, Uq, 36 * p.Cout, 1, 1, nullptr, M, 1, 1, 1, 1, 0, 0, 0, 0, 36,
                         nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0, 2);
    ctx->xcd_cols_request = 0;
    ctx->last_plan = "wino4[" + ctx->last_plan + "]";
    return rc;
}

int wino4_output_launch(pl_ctx *ctx, const float *M, float *yq, const WinoArgs &p, int lds_ok) {
    const int Coq = p.Cout / 4;
    if (lds_ok && wino4_lds_mode() >= 2 && wino4_chain_pick_g(ctx, p.N, Coq, p.th, p.tw, true) > 0)
        return wino4_chain_launch(ctx, M, nullptr, p, p.Cout, yq, nullptr);
    const unsigned cap = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    const unsigned tout = (unsigned)((size_t)Coq * p.T);
    static const char *rows_env = getenv("PLANER_HIP_WINO_ROWS");
    const unsigned cus = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256);
    const bool out_rows = rows_env ? atoi(rows_env) != 0 : (tout + 255) / 256 < cus / 2;
    if (out_rows)
        wino4_output_rows_q4_kernel<<<dim3(std::min(cap, (tout + 255) / 256), 4), 256, 0, ctx->stream>>>(
            (const float4 *)M, (float4 *)yq, p, Coq, tout);
    else
        wino4_output_q4_kernel<<<std::min(cap, (tout + 255) / 256), 256, 0, ctx->stream>>>((const float4 *)M, (float4 *)yq, p,
                                                                                           Coq, tout);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

// lds_mode: 0 = the register transform kernels (round 2), 1 = the LDS transform kernel where the plane fits
int winograd4_q4_launch(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *Uq, int Cout,
                        const float *bias, float *yq, const float *scale, const float *shift, const float *resq,
                        int act, double alpha, int lds_mode) {
    WinoArgs p;
    int rc = wino4_geometry(p, N, Cin, H, W, Cout);
    if (rc != PL_OK) return rc;
    p.ep = make_epilogue(bias, scale, shift, resq, act, alpha);
    const size_t vin = (size_t)36 * Cin * p.T, vout = (size_t)36 * Cout * p.T;
    float *V = nullptr, *M = nullptr;
    rc = pl_alloc(ctx, vin * sizeof(float), (void **)&V);
    if (rc != PL_OK) return rc;
    rc = pl_alloc(ctx, vout * sizeof(float), (void **)&M);
    if (rc != PL_OK) {
        pl_free(ctx, V);
        return rc;
    }
    rc = wino4_input_launch(ctx, xq, V, p, lds_mode);
    if (rc == PL_OK) rc = wino4_gemm_launch(ctx, V, Uq, M, p);
    if (rc == PL_OK) rc = wino4_output_launch(ctx, M, yq, p, lds_mode);
    pl_free(ctx, M);
    pl_free(ctx, V);
    return rc;
}

// ---- mixed-tile Winograd (wino43_kernels.h): maps whose sides are 7, 14 or 21 ----
bool wino43_side_ok(int d) { return d == 7 || d == 14 || d == 21; }
int wino43_args(W43Args &a, int N, int C, int H, int W) {
    PL_REQUIRE(wino43_side_ok(H) && wino43_side_ok(W), PL_EUNSUPPORTED, "mixed-tile winograd: map sides must be 7, 14 or 21 (got %d x %d)", H, W);
    memset(&a, 0, sizeof a);
    a.N = N; a.Cq = C / 4; a.H = H; a.W = W;
    a.ar = H / 7; a.ac = W / 7; a.TC = a.ar * a.ac; a.T = N * a.TC;
    const size_t vb = (size_t)W43_GROUPS * C * a.T * 4, xb = (size_t)N * C * H * W * 4;
    PL_REQUIRE(vb < (1ull << 31) && xb < (1ull << 31), PL_EUNSUPPORTED, "mixed-tile winograd: tensor too large");
    a.x_bytes = (unsigned)xb;
    a.divCT = FastDiv((unsigned)(a.Cq * a.T)); a.divT = FastDiv((unsigned)a.T); a.divTC = FastDiv((unsigned)a.TC); a.divAc = FastDiv((unsigned)a.ac);
    return PL_OK;
}
int wino43_lds_launch(pl_ctx *ctx, const float *M, const float *x, float *yq, float *V, int N, int C, int H, int W, const Epilogue &ep);
int wino43_lds_mode();
int wino43_input_launch(pl_ctx *ctx, const float *xq, float *V, int N, int C, int H, int W) {
    if (wino43_lds_mode())
        return wino43_lds_launch(ctx, nullptr, xq, nullptr, V, N, C, H, W, make_epilogue(nullptr, nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0));
    W43Args a;
    int rc = wino43_args(a, N, C, H, W);
    if (rc != PL_OK) return rc;
    a.x = xq; a.V = V;
    a.ep = make_epilogue(nullptr, nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0);
    const unsigned total = 4u * (unsigned)a.Cq * (unsigned)a.T * 4u;
    const unsigned cap = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    wino43_input_q4_kernel<<<std::min(cap, (total + 255) / 256), 256, 0, ctx->stream>>>(a, total);
    PL_LAUNCH_CHECK();
    return PL_OK;
}
int wino43_gemm_launch(pl_ctx *ctx, const float *V, const float *Uq, float *M, int N, int Cin, int H, int W, int Cout) {
    PL_REQUIRE(wino43_side_ok(H) && wino43_side_ok(W), PL_EUNSUPPORTED, "mixed-tile winograd: map sides must be 7, 14 or 21");
    const int ar = H / 7, ac = W / 7;
    // 121 GEMMs of (Cout x Cin) . (Cin x N ar ac) as ONE grouped 1x1 conv; the tile axis is presented as an (N ar) x ac image
    int rc = conv_launch(ctx, V, 1, W43_GROUPS * Cin, N * ar, ac, Uq, W43_GROUPS * Cout, 1, 1, nullptr, M, 1, 1, 1, 1, 0, 0, 0, 0, W43_GROUPS,
                         nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0, 2);
    ctx->last_plan = "wino43[" + ctx->last_plan + "]";
    return rc;
}
// the LDS kernel (whole planes per workgroup): M -> y and / or V (from_m), or x -> V
int wino43_lds_launch(pl_ctx *ctx, const float *M, const float *x, float *yq, float *V, int N, int C, int H, int W, const Epilogue &ep) {
    W43LdsArgs q;
    int rc = wino43_args(q.a, N, C, H, W);
    if (rc != PL_OK) return rc;
    W43Args &a = q.a;
    a.M = M; a.x = x; a.y = yq; a.V = V; a.ep = ep;
    q.from_m = M != nullptr;
    q.pcells = ((H + 2) * (W + 2) + 14) / 16 * 16 + 1;
    // G quads per workgroup: the largest power of two that keeps the workgroup under 64 KB of LDS (two or more per CU), under
    // ~700 row items and the grid at one workgroup per CU or more.  PLANER_HIP_WINO43_G forces.
    static const char *g_env = getenv("PLANER_HIP_WINO43_G");
    const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    // planes + the frequency slab (products, then the half-transformed patches) + tail parameters (wino43_lds_kernel's layout)
    auto lds_of = [&](int g) { return ((size_t)g * q.pcells + (size_t)W43_GROUPS * g * a.TC + 3 * (size_t)g) * 16; };
    int G = 1;
    for (int g = 2; g <= a.Cq; g *= 2)
        if (a.Cq % g == 0 && lds_of(g) <= 64 * 1024 && 22 * g * a.TC <= 704 && (long)N * (a.Cq / g) >= cus) G = g;
    if (g_env && atoi(g_env) > 0 && a.Cq % atoi(g_env) == 0 && lds_of(atoi(g_env)) <= 150 * 1024) G = atoi(g_env);
    PL_REQUIRE(lds_of(G) <= 150 * 1024, PL_EUNSUPPORTED, "mixed-tile winograd (LDS transforms): a plane does not fit");
    a.G = G;
    q.gt = G * a.TC;
    q.divGt = FastDiv((unsigned)q.gt); q.div2Gt = FastDiv(2u * (unsigned)q.gt);
    q.divPcells = FastDiv((unsigned)q.pcells); q.divPitch = FastDiv((unsigned)(W + 2));
    q.divHW = FastDiv((unsigned)(H * W)); q.divW = FastDiv((unsigned)W);
    const size_t lds = lds_of(G);
    if (lds > 48 * 1024) {
        rc = ensure_lds_attr((const void *)wino43_lds_kernel, 150 * 1024);
        if (rc != PL_OK) return rc;
    }
    const int items = 22 * q.gt, bd = std::max(128, std::min(512, (items + 63) / 64 * 64));
    hipLaunchKernelGGL(wino43_lds_kernel, dim3((unsigned)(a.Cq / G), (unsigned)N), dim3((unsigned)bd), lds, ctx->stream, q);
    PL_LAUNCH_CHECK();
    return PL_OK;
}
// PLANER_HIP_WINO43_LDS: 0 = whole-tile register kernels for lone transforms (chains always take the LDS kernel), 1 = the LDS kernel
// for lone transforms too
int wino43_lds_mode() {
    const char *e = getenv("PLANER_HIP_WINO43_LDS");
    return e ? atoi(e) : 1;
}
int wino43_output_launch(pl_ctx *ctx, const float *M, float *yq, float *Vnext, int N, int C, int H, int W, const Epilogue &ep) {
    if (Vnext || wino43_lds_mode()) return wino43_lds_launch(ctx, M, nullptr, yq, Vnext, N, C, H, W, ep);
    W43Args a;
    int rc = wino43_args(a, N, C, H, W);
    if (rc != PL_OK) return rc;
    a.M = M; a.y = yq; a.ep = ep;
    const unsigned total = 4u * (unsigned)a.Cq * (unsigned)a.T;
    const unsigned cap = (unsigned)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    wino43_output_q4_kernel<<<std::min(cap, (total + 255) / 256), 256, 0, ctx->stream>>>(a, total);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

#include "conv_w1d_kernel.h"

int w1d_launch(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *uq, int Cout, const float *bias,
               float *yq, const float *scale, const float *shift, const float *resq, int act, double alpha) {
    ConvArgs a;
    memset(&a, 0, sizeof a);
    const int Tw = (W + 3) / 4;                              // 4 output pixels per tile
    a.x = xq; a.w = uq; a.y = yq;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = H; a.Wo = W;
    a.kh = 3; a.kw = 3; a.sh = a.sw = a.dh = a.dw = 1; a.pt = a.pl = 1;
    a.groups = 1; a.cin_g = Cin; a.cout_g = Cout;
    a.cqg = Cin / 4; a.Cq = Cin / 4; a.Coq = (Cout + 3) / 4;
    a.Qtot = 3 * a.cqg; a.Qpad = (a.Qtot + 7) / 8 * 8;
    a.K = a.Qtot * 4;
    a.cols = N * H * Tw;
    a.HoWo = H * W; a.HW = H * W;
    const size_t in_elems = (size_t)N * Cin * H * W, out_elems = (size_t)N * a.Coq * 4 * H * W;
    const size_t w_elems = (size_t)6 * a.Qpad * Cout * 4;
    // (output addressed through a 32-bit buffer offset in the F(4,3) kernels' epilogue: < 2 GiB)
    PL_REQUIRE(in_elems < (1ull << 29) && out_elems < (1ull << 29) && w_elems < (1ull << 29) &&
                   (size_t)N * H * Tw < (1ull << 31), PL_EUNSUPPORTED, "winograd-1d: tensor too large");
    a.x_bytes = (int)(in_elems * 4); a.w_bytes = (int)(w_elems * 4);
    a.divHoWo = FastDiv(H * Tw); a.divWo = FastDiv(Tw); a.divCpt = FastDiv(a.cqg);
    a.divKhw = FastDiv(1); a.divKw = FastDiv(1);
    a.mtiles = (Cout + W1d4Cfg::BM - 1) / W1d4Cfg::BM;
    a.ntiles = (a.cols + W1d4Cfg::BN - 1) / W1d4Cfg::BN;
    a.tiles = a.mtiles * a.ntiles;
    a.divMt = FastDiv(a.mtiles);
    a.tile_offset = 0; a.tile_count = a.tiles; a.splits = 1;
    a.ep = make_epilogue(bias, scale, shift, resq, act, alpha);
    {
        int rc = ensure_lds_attr((const void *)conv_w1d4_kernel, W1d4Cfg::LDS_BYTES);
        if (rc != PL_OK) return rc;
        hipLaunchKernelGGL(conv_w1d4_kernel, dim3((unsigned)a.tiles), dim3(256), W1d4Cfg::LDS_BYTES, ctx->stream, a);
    }
    PL_LAUNCH_CHECK();
    {
        char buf[96];
        snprintf(buf, sizeof buf, "w1d4 64x64x8 tiles=%d", a.tiles);
        ctx->last_plan = buf;
        // 6 frequency GEMMs of (Cout x 3 Cin) . (3 Cin x column tiles), 64 x 64 tiles, whole chunks
        const int bk = 8;
        ctx->last_gemm[0] = 6;
        ctx->last_gemm[1] = (long long)(Cout + 63) / 64 * 64;
        ctx->last_gemm[2] = (long long)(a.cols + 63) / 64 * 64;
        ctx->last_gemm[3] = (long long)(a.Qtot * 4 + bk - 1) / bk * bk;
    }
    return PL_OK;
}

#include "conv_wf4_kernel.h"
#ifndef WF4_HALF_DEFAULT
#define WF4_HALF_DEFAULT 1
#endif
#ifndef WF4_STAGGER
#define WF4_STAGGER true      // (probe builds: false = every wave multiplies first, the patch transform follows)
#endif

int wf4_launch(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *u, int Cout, const float *bias,
               float *yq, const float *scale, const float *shift, const float *resq, int act, double alpha) {
    Wf4Args a;
    memset(&a, 0, sizeof a);
    a.x = xq; a.u = u; a.y = yq;
    a.N = N; a.Cq = Cin / 4; a.Coq = Cout / 4; a.H = H; a.W = W;
    a.th = (H + 3) / 4; a.tw = (W + 3) / 4;
    a.nchunks = Cin / 4;
    // blocks of 32 tiles on eight waves, or (PLANER_HIP_EXPERIMENT=wf4_half=1; needs the filter-in-registers build) of 16 tiles on
    // four waves -- two workgroups per CU with barriers of their own
    const bool half = WF4_GLOBAL_A && pl_experiment("wf4_half", WF4_HALF_DEFAULT) != 0;
    const int TB = half ? 16 : 32;
    int BC = 1, BR = 1, lBC = 0, lBR = 0;
    while (BC < a.tw && BC < 16) BC *= 2, ++lBC;
    while (BR < a.th && BR * BC < TB) BR *= 2, ++lBR;
    // Fewer tile rows per block and more images instead, where that needs fewer 32-tile blocks: 7 tile rows (a 28-pixel map) are
    // two blocks of 4 rows per image (64 slots for 49 tiles) but seven blocks of 1 row x 4 images (56 slots per image quartet's
    // 49 x 4 / 4) -- 112 instead of 128 workgroups for ResNet-18's layer2 at batch 32.  Ties keep the taller block (less halo in
    // the patch).  PLANER_HIP_EXPERIMENT=wf4_br=<rows> forces.
    {
        auto blocks_of = [&](int br) { const int nb = TB / (br * BC); return (long long)((N + nb - 1) / nb) * ((a.th + br - 1) / br); };
        int best_br = BR, best_l = lBR;
        for (int br = BR / 2, l = lBR - 1; br >= 1; br /= 2, --l)
            if (blocks_of(br) < blocks_of(best_br) && (TB / (br * BC)) * (4 * br + 2) * 4 * (BC + 1) <= (half ? 512 : WF4_P_CELLS)) best_br = br, best_l = l;
        const int force = pl_experiment("wf4_br", 0);
        if (force > 0 && force <= BR && (force & (force - 1)) == 0 && (TB / (force * BC)) * (4 * force + 2) * 4 * (BC + 1) <= (half ? 512 : WF4_P_CELLS)) {
            best_br = force;
            best_l = 0;
            while ((1 << best_l) < force) ++best_l;
        }
        BR = best_br; lBR = best_l;
    }
    const int NB = TB / (BR * BC);
    a.lBR = lBR; a.lBC = lBC;
    a.R = 4 * BR + 2; a.S = BC + 1;
    a.rblocks = (a.th + BR - 1) / BR; a.cblocks = (a.tw + BC - 1) / BC; a.cout_blocks = (Cout + 63) / 64;
    // Packed blocks (conv_wf4_kernel<.., PACK>): one image per block, one column block, sc = BC - tw spare slot columns that divide
    // tw, whole row blocks, batch a multiple of G = tw / sc + 1 -- the spare slots of G - 1 images' blocks carry the G-th image
    // (ResNet-18's layer1 at batch 32: 196 instead of 224 workgroups).  PLANER_HIP_EXPERIMENT=wf4_pack=0 switches it off.
    {
        const int sc = BC - a.tw;
        if (NB == 1 && a.cblocks == 1 && sc > 0 && a.tw % sc == 0 && a.th % BR == 0 && (lBC == 4 || lBC == 3) &&
            N % (a.tw / sc + 1) == 0 && a.R * 4 * (BC + 2) <= (half ? 512 : WF4_P_CELLS) && pl_experiment("wf4_pack", 1)) {
            a.pack_sc = sc; a.pack_g = a.tw / sc + 1; a.pack_gc = a.tw / sc;
            a.S = BC + 2;
        }
    }
    a.cells = NB * a.R * 4 * a.S;
    const long long groups = a.pack_g ? (long long)(N / a.pack_g) * (a.pack_g - 1) : (N + NB - 1) / NB;
    const long long blocks = groups * a.rblocks * a.cblocks * a.cout_blocks;
    const size_t xb = (size_t)N * Cin * H * W * 4, yb = (size_t)N * Cout * H * W * 4;
    const size_t ub = (size_t)a.cout_blocks * a.nchunks * WF4_A_FLOATS * 4;
    PL_REQUIRE(a.cells <= (half ? 512 : WF4_P_CELLS) && blocks < (1ll << 31) && xb < (1ull << 31) && yb < (1ull << 31) && ub < (1ull << 31),
               PL_EUNSUPPORTED, "fused winograd F(4x4,3x3): tensor too large");
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb; a.u_bytes = (unsigned)ub;
    a.divPlane = FastDiv(a.R * 4 * a.S); a.div4S = FastDiv(4 * a.S); a.divS = FastDiv(a.S);
    a.divCoB = FastDiv(a.cout_blocks); a.divCb = FastDiv(a.cblocks); a.divRb = FastDiv(a.rblocks);
    a.ep = make_epilogue(bias, scale, shift, resq, act, alpha);
    // LDS-DMA operands, waves 4-7 transform before their MFMAs and waves 0-3 after (the variant that measured fastest: 48.5 us
    // against 51.4 us for register-staged operands with the waves in step, layer1 of ResNet-18 at batch 32); one instantiation
    // per block width, so that every patch read is base + immediate
    void (*kern)(const Wf4Args) = nullptr;
#if WF4_GLOBAL_A
    if (half) {
        switch (a.pack_g ? 10 + lBC : lBC) {
        case 14: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 4, true, true>; break;
        case 13: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 3, true, true>; break;
        case 4: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 4, false, true>; break;
        case 3: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 3, false, true>; break;
        case 2: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 2, false, true>; break;
        case 1: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 1, false, true>; break;
        default: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 0, false, true>; break;
        }
    } else
#endif
    switch (a.pack_g ? 10 + lBC : lBC) {
    case 14: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 4, true>; break;
    case 13: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 3, true>; break;
    case 4: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 4>; break;
    case 3: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 3>; break;
    case 2: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 2>; break;
    case 1: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 1>; break;
    default: kern = conv_wf4_kernel<true, false, WF4_STAGGER, 0>; break;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(half ? 256 : 512), 0, ctx->stream, a);      // LDS: static
    PL_LAUNCH_CHECK();
    char buf[96];
    snprintf(buf, sizeof buf, "wf4 64co x %dtiles (%dx%dx%d%s) blocks=%lld", TB, NB, BR, BC, a.pack_g ? " packed" : "", blocks);
    ctx->last_plan = buf;
    ctx->last_gemm[0] = 36; ctx->last_gemm[1] = (long long)a.cout_blocks * 64;
    ctx->last_gemm[2] = groups * a.rblocks * a.cblocks * TB; ctx->last_gemm[3] = Cin;
    return PL_OK;
}

}  // namespace

extern "C" {

#ifdef WF4_STAMP
// probe builds only: copies the fused kernel's time stamps out (tools/wf4_stamp.py)
int pl_debug_wf4_stamps(void *dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wf4_stamps), bytes) == hipSuccess ? PL_OK : PL_EHIP;
}
int pl_debug_wf4_step_stamps(void *dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wf4_step_stamps), bytes) == hipSuccess ? PL_OK : PL_EHIP;
}
#endif

int pl_conv2d_w1d4_q4_filter_elems(int Cout, int Cin, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin > 0, PL_EINVAL, "winograd-1d filter size: bad argument");
    *elems = (size_t)6 * (((size_t)3 * (Cin / 4) + 7) / 8 * 8) * Cout * 4;
    return PL_OK;
}

int pl_conv2d_prepare_w1d4_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_w1d4_q4_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin % 4 == 0, PL_EINVAL, "winograd-1d filters need Cin %% 4 == 0");
    PL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, PL_EINVAL, "pl_conv2d_prepare_w1d4_q4_f32: unaligned output");
    const int cqg = Cin / 4, q_tot = 3 * cqg, q_pad = (q_tot + 7) / 8 * 8;
    const size_t total = (size_t)q_pad * Cout;
    PL_REQUIRE(total * 16 < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    pack_filter_w1d4_kernel<<<blocks, 256, 0, ctx->stream>>>(w, out, (unsigned)total, Cout, Cin, cqg, q_tot, q_pad, FastDiv(Cout),
                                                            FastDiv(cqg));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_conv2d_w1d4_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *uq, int Cout,
                          const float *bias, float *yq, const float *scale, const float *shift, const float *resq,
                          int act, double alpha) {
    PL_REQUIRE(ctx && xq && uq && yq, PL_EINVAL, "pl_conv2d_w1d4_q4_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && Cin % 4 == 0, PL_EINVAL,
               "pl_conv2d_w1d4_q4_f32: bad shape (Cin must be a multiple of 4)");
    PL_REQUIRE(H < 16384 && W < 16384, PL_EUNSUPPORTED, "pl_conv2d_w1d4_q4_f32: spatial extent above 16383");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_conv2d_w1d4_q4_f32: bad activation code");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(uq) |
                 reinterpret_cast<uintptr_t>(resq)) & 15u) == 0, PL_EINVAL, "Q4 tensors must be 16-byte aligned");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return w1d_launch(ctx, xq, N, Cin, H, W, uq, Cout, bias, yq, scale, shift, resq, act, alpha);
}

int pl_conv2d_winograd4_q4_filter_elems(int Cout, int Cin, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin > 0, PL_EINVAL, "pl_conv2d_winograd4_q4_filter_elems: bad argument");
    *elems = (size_t)36 * (((size_t)Cin / 4 + 7) / 8 * 8) * Cout * 4;
    return PL_OK;
}

int pl_conv2d_prepare_winograd4_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_winograd4_q4_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL,
               "winograd F(4,3) Q4 filters need Cin %% 4 == 0 and Cout %% 4 == 0");
    const size_t pairs = (size_t)Cout * Cin;
    size_t elems = 0;
    pl_conv2d_winograd4_q4_filter_elems(Cout, Cin, &elems);
    PL_REQUIRE(elems < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    PL_HIP(hipMemsetAsync(out, 0, elems * sizeof(float), ctx->stream));       // k-quad padding
    wino4_filter_q4_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, ctx->stream>>>(w, out, (unsigned)pairs, Cin, Cout,
                                                                                    (Cin / 4 + 7) / 8 * 8);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_conv2d_winograd4_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *uq, int Cout,
                               const float *bias, float *yq, const float *scale, const float *shift,
                               const float *resq, int act, double alpha) {
    PL_REQUIRE(ctx && xq && uq && yq, PL_EINVAL, "pl_conv2d_winograd4_q4_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL,
               "pl_conv2d_winograd4_q4_f32: bad shape (Cin, Cout must be multiples of 4)");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_conv2d_winograd4_q4_f32: bad activation code");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(uq) |
                 reinterpret_cast<uintptr_t>(resq)) & 15u) == 0, PL_EINVAL, "Q4 tensors must be 16-byte aligned");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    // which transform kernels: the register ones (round 2) unless PLANER_HIP_WINO_MONO_LDS=1
    static const int mono_lds = getenv("PLANER_HIP_WINO_MONO_LDS") ? atoi(getenv("PLANER_HIP_WINO_MONO_LDS")) : 0;
    return winograd4_q4_launch(ctx, xq, N, Cin, H, W, uq, Cout, bias, yq, scale, shift, resq, act, alpha, mono_lds);
}

// ---- fully fused F(4x4,3x3): conv_wf4_kernel.h ----
int pl_conv2d_wf4_filter_elems(int Cout, int Cin, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin > 0 && Cin % 4 == 0, PL_EINVAL, "pl_conv2d_wf4_filter_elems: bad argument");
    *elems = (size_t)((Cout + 63) / 64) * (Cin / 4) * WF4_A_FLOATS;
    return PL_OK;
}

int pl_conv2d_prepare_wf4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_wf4_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL,
               "fused winograd F(4x4,3x3) filters need Cin %% 4 == 0 and Cout %% 4 == 0");
    size_t elems = 0;
    pl_conv2d_wf4_filter_elems(Cout, Cin, &elems);
    PL_REQUIRE(elems < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    PL_HIP(hipMemsetAsync(out, 0, elems * sizeof(float), ctx->stream));       // channel padding of the last 64-block
    const size_t pairs = (size_t)Cout * Cin;
    wf4_filter_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, ctx->stream>>>(w, out, (unsigned)pairs, Cin, Cout);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_conv2d_wf4_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *u, int Cout,
                         const float *bias, float *yq, const float *scale, const float *shift, const float *resq,
                         int act, double alpha) {
    PL_REQUIRE(ctx && xq && u && yq, PL_EINVAL, "pl_conv2d_wf4_q4_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL,
               "pl_conv2d_wf4_q4_f32: bad shape (Cin, Cout must be multiples of 4)");
    PL_REQUIRE(H < 16384 && W < 16384, PL_EUNSUPPORTED, "pl_conv2d_wf4_q4_f32: spatial extent above 16383");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_conv2d_wf4_q4_f32: bad activation code");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(u) |
                 reinterpret_cast<uintptr_t>(resq) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(scale) |
                 reinterpret_cast<uintptr_t>(shift)) & 15u) == 0, PL_EINVAL, "Q4 tensors and per-channel parameters must be 16-byte aligned");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return wf4_launch(ctx, xq, N, Cin, H, W, u, Cout, bias, yq, scale, shift, resq, act, alpha);
}

// ---- the F(4x4,3x3) pipeline stage by stage (V / M: [36][C/4][T][4], T = N * ceil(H/4) * ceil(W/4)) ----
int pl_wino4_elems(int N, int C, int H, int W, size_t *elems) {
    PL_REQUIRE(elems && N >= 0 && C > 0 && H > 0 && W > 0 && C % 4 == 0, PL_EINVAL, "pl_wino4_elems: bad argument");
    *elems = (size_t)36 * C * N * ((H + 3) / 4) * ((W + 3) / 4);
    return PL_OK;
}

int pl_wino4_chain_supported(pl_ctx *ctx, int N, int C, int H, int W, int *ok) {
    PL_REQUIRE(ok && C > 0 && H > 0 && W > 0 && C % 4 == 0, PL_EINVAL, "pl_wino4_chain_supported: bad argument");
    *ok = wino4_lds_enabled() && wino4_chain_pick_g(ctx, N, C / 4, (H + 3) / 4, (W + 3) / 4, true) > 0 &&
          (size_t)36 * C * N * ((H + 3) / 4) * ((W + 3) / 4) * 4 < (1ull << 31);
    return PL_OK;
}

static int wino4_stage_check(const char *fn, pl_ctx *ctx, int N, int C, int H, int W, const void *a, const void *b,
                             const void *c, const void *d) {
    PL_REQUIRE(ctx, PL_EINVAL, "%s: null context", fn);
    PL_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && C % 4 == 0, PL_EINVAL, "%s: bad shape (C must be a multiple of 4)", fn);
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                 reinterpret_cast<uintptr_t>(d)) & 15u) == 0, PL_EINVAL, "%s: Q4 tensors must be 16-byte aligned", fn);
    return PL_OK;
}

int pl_wino4_input_q4_f32(pl_ctx *ctx, const float *xq, int N, int C, int H, int W, float *V) {
    int rc = wino4_stage_check("pl_wino4_input_q4_f32", ctx, N, C, H, W, xq, V, nullptr, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(xq && V, PL_EINVAL, "pl_wino4_input_q4_f32: null pointer");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    WinoArgs p;
    rc = wino4_geometry(p, N, C, H, W, C);
    if (rc != PL_OK) return rc;
    p.ep = make_epilogue(nullptr, nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0);
    return wino4_input_launch(ctx, xq, V, p, 1);
}

int pl_wino4_gemm_q4_f32(pl_ctx *ctx, const float *V, int N, int Cin, int H, int W, const float *uq, int Cout, float *M) {
    int rc = wino4_stage_check("pl_wino4_gemm_q4_f32", ctx, N, Cin, H, W, V, uq, M, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(V && uq && M && Cout > 0 && Cout % 4 == 0, PL_EINVAL, "pl_wino4_gemm_q4_f32: bad argument");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    WinoArgs p;
    rc = wino4_geometry(p, N, Cin, H, W, Cout);
    if (rc != PL_OK) return rc;
    return wino4_gemm_launch(ctx, V, uq, M, p);
}

int pl_wino4_output_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias, const float *scale,
                           const float *shift, const float *resq, int act, double alpha, float *yq) {
    int rc = wino4_stage_check("pl_wino4_output_q4_f32", ctx, N, C, H, W, M, yq, resq, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(M && yq, PL_EINVAL, "pl_wino4_output_q4_f32: null pointer");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_wino4_output_q4_f32: bad activation code");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    WinoArgs p;
    rc = wino4_geometry(p, N, C, H, W, C);
    if (rc != PL_OK) return rc;
    p.ep = make_epilogue(bias, scale, shift, resq, act, alpha);
    return wino4_output_launch(ctx, M, yq, p, 1);
}

int pl_wino4_chain_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias, const float *scale,
                          const float *shift, const float *resq, int act, double alpha, float *yq, float *Vnext) {
    int rc = wino4_stage_check("pl_wino4_chain_q4_f32", ctx, N, C, H, W, M, yq, resq, Vnext);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(M && Vnext, PL_EINVAL, "pl_wino4_chain_q4_f32: null pointer");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_wino4_chain_q4_f32: bad activation code");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    WinoArgs p;
    rc = wino4_geometry(p, N, C, H, W, C);
    if (rc != PL_OK) return rc;
    p.ep = make_epilogue(bias, scale, shift, resq, act, alpha);
    return wino4_chain_launch(ctx, M, nullptr, p, C, yq, Vnext);
}

// 1x1 conv + fused tail + the Winograd input transform of the 3x3 conv that follows (conv1x1_wino_in_kernel.h): xq (N, Cin, H, W)
// -> V of the (N, Cout, H, W) activation, F(4x4,3x3) (wino = 4, [36][Cout/4][T][4]) or F(2x2,3x3) (wino = 2, [16][Cout/4][T][4])
int pl_conv1x1_wino_in_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *wq, int Cout,
                              const float *bias, const float *scale, const float *shift, int act, double alpha, int wino, float *V) {
    PL_REQUIRE(ctx && xq && wq && V, PL_EINVAL, "pl_conv1x1_wino_in_q4_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 4 == 0 && (wino == 4 || wino == 2), PL_EINVAL,
               "pl_conv1x1_wino_in_q4_f32: bad shape (Cout must be a multiple of 4, wino 2 or 4)");
    // F(2x2,3x3): the kernel template has the branch, nothing in the package ever asked for it and no test pins it (round-5
    // advisor) -- refused until a caller and a parity test exist, rather than shipped unverified
    PL_REQUIRE(wino == 4, PL_EUNSUPPORTED, "pl_conv1x1_wino_in_q4_f32: only wino = 4 (the F(4x4,3x3) domain) is supported");
    PL_REQUIRE(act >= 0 && act <= 2, PL_EINVAL, "pl_conv1x1_wino_in_q4_f32: bad activation code");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(V)) & 15u) == 0, PL_EINVAL,
               "Q4 tensors must be 16-byte aligned");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    C1WArgs a;
    memset(&a, 0, sizeof a);
    a.x = xq; a.w = wq; a.V = (float4 *)V;
    a.N = N; a.Cq = (Cin + 3) / 4; a.H = H; a.W = W; a.Cout = Cout;
    a.Qtot = a.Cq; a.Qpad = (a.Qtot + 7) / 8 * 8;
    const int ts = wino == 4 ? 4 : 2;
    a.th = (H + ts - 1) / ts; a.tw = (W + ts - 1) / ts; a.T = N * a.th * a.tw;
    a.rty = wino == 4 ? 1 : 2; a.rtx = wino == 4 ? 2 : 4;
    a.rh = (a.th + a.rty - 1) / a.rty; a.rw = (a.tw + a.rtx - 1) / a.rtx;
    a.mtiles = (Cout + 31) / 32;
    const size_t xb = (size_t)N * a.Cq * H * W * 16, wb = (size_t)a.Qpad * Cout * 16;
    const size_t vb = (size_t)(wino == 4 ? 36 : 16) * Cout * a.T * 4;
    const long long blocks = (long long)N * a.rh * a.rw * a.mtiles;
    PL_REQUIRE(xb < (1ull << 31) && wb < (1ull << 31) && vb < (1ull << 33) && blocks < (1ll << 31), PL_EUNSUPPORTED,
               "pl_conv1x1_wino_in_q4_f32: tensor too large");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.divMt = FastDiv(a.mtiles); a.divRw = FastDiv(a.rw); a.divRh = FastDiv(a.rh);
    a.ep = make_epilogue(bias, scale, shift, nullptr, act, alpha);
    auto kern = conv1x1_wino_in_kernel<4>;
    int rc = ensure_lds_attr((const void *)kern, C1W_LDS_FLOATS * 4);
    if (rc != PL_OK) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C1W_WAVES * 64), C1W_LDS_FLOATS * 4, ctx->stream, a);
    PL_LAUNCH_CHECK();
    char buf[96];
    snprintf(buf, sizeof buf, "conv1x1+wino%d-in 32co x 6x10px regions=%d blocks=%lld", wino, N * a.rh * a.rw, blocks);
    ctx->last_plan = buf;
    ctx->last_gemm[0] = 1; ctx->last_gemm[1] = (long long)a.mtiles * 32;
    ctx->last_gemm[2] = (long long)N * a.rh * a.rw * 64; ctx->last_gemm[3] = (long long)a.Qpad * 4;
    return PL_OK;
}

// ---- mixed-tile Winograd, F(4,3) x F(3,3) segments (wino43_kernels.h): maps of 7 / 14 / 21 pixels a side ----
int pl_wino43_supported(int H, int W, int *ok) {
    PL_REQUIRE(ok, PL_EINVAL, "pl_wino43_supported: null argument");
    *ok = wino43_side_ok(H) && wino43_side_ok(W);
    return PL_OK;
}
int pl_wino43_elems(int N, int C, int H, int W, size_t *elems) {
    PL_REQUIRE(elems && N >= 0 && C > 0 && C % 4 == 0 && wino43_side_ok(H) && wino43_side_ok(W), PL_EINVAL, "pl_wino43_elems: bad argument");
    *elems = (size_t)W43_GROUPS * C * N * (H / 7) * (W / 7);
    return PL_OK;
}
int pl_conv2d_winograd43_q4_filter_elems(int Cout, int Cin, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin > 0, PL_EINVAL, "pl_conv2d_winograd43_q4_filter_elems: bad argument");
    *elems = (size_t)W43_GROUPS * (((size_t)Cin / 4 + 7) / 8 * 8) * Cout * 4;
    return PL_OK;
}
int pl_conv2d_prepare_winograd43_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_winograd43_q4_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL, "mixed-tile winograd filters need Cin %% 4 == 0 and Cout %% 4 == 0");
    const size_t pairs = (size_t)Cout * Cin;
    size_t elems = 0;
    pl_conv2d_winograd43_q4_filter_elems(Cout, Cin, &elems);
    PL_REQUIRE(elems < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    PL_HIP(hipMemsetAsync(out, 0, elems * sizeof(float), ctx->stream));
    wino43_filter_q4_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, ctx->stream>>>(w, out, (unsigned)pairs, Cin, Cout, (Cin / 4 + 7) / 8 * 8);
    PL_LAUNCH_CHECK();
    return PL_OK;
}
static int wino43_check(const char *fn, pl_ctx *ctx, int N, int C, int H, int W, const void *a, const void *b, const void *c, const void *d) {
    int rc = wino4_stage_check(fn, ctx, N, C, H, W, a, b, c, d);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(wino43_side_ok(H) && wino43_side_ok(W), PL_EUNSUPPORTED, "%s: map sides must be 7, 14 or 21", fn);
    return PL_OK;
}
int pl_wino43_input_q4_f32(pl_ctx *ctx, const float *xq, int N, int C, int H, int W, float *V) {
    int rc = wino43_check("pl_wino43_input_q4_f32", ctx, N, C, H, W, xq, V, nullptr, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(xq && V, PL_EINVAL, "pl_wino43_input_q4_f32: null pointer");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return wino43_input_launch(ctx, xq, V, N, C, H, W);
}
int pl_wino43_gemm_q4_f32(pl_ctx *ctx, const float *V, int N, int Cin, int H, int W, const float *uq, int Cout, float *M) {
    int rc = wino43_check("pl_wino43_gemm_q4_f32", ctx, N, Cin, H, W, V, uq, M, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(V && uq && M && Cout > 0 && Cout % 4 == 0, PL_EINVAL, "pl_wino43_gemm_q4_f32: bad argument");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return wino43_gemm_launch(ctx, V, uq, M, N, Cin, H, W, Cout);
}
int pl_wino43_output_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias, const float *scale,
                            const float *shift, const float *resq, int act, double alpha, float *yq) {
    int rc = wino43_check("pl_wino43_output_q4_f32", ctx, N, C, H, W, M, yq, resq, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(M && yq, PL_EINVAL, "pl_wino43_output_q4_f32: null pointer");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_wino43_output_q4_f32: bad activation code");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return wino43_output_launch(ctx, M, yq, nullptr, N, C, H, W, make_epilogue(bias, scale, shift, resq, act, alpha));
}
int pl_wino43_chain_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias, const float *scale,
                           const float *shift, const float *resq, int act, double alpha, float *yq, float *Vnext) {
    int rc = wino43_check("pl_wino43_chain_q4_f32", ctx, N, C, H, W, M, yq, resq, Vnext);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(M && Vnext, PL_EINVAL, "pl_wino43_chain_q4_f32: null pointer");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_wino43_chain_q4_f32: bad activation code");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return wino43_output_launch(ctx, M, yq, Vnext, N, C, H, W, make_epilogue(bias, scale, shift, resq, act, alpha));
}
int pl_conv2d_winograd43_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *uq, int Cout,
                                const float *bias, float *yq, const float *scale, const float *shift, const float *resq,
                                int act, double alpha) {
    int rc = wino43_check("pl_conv2d_winograd43_q4_f32", ctx, N, Cin, H, W, xq, uq, yq, resq);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(xq && uq && yq && Cout > 0 && Cout % 4 == 0, PL_EINVAL, "pl_conv2d_winograd43_q4_f32: bad argument");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_conv2d_winograd43_q4_f32: bad activation code");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    size_t vin = 0, vout = 0;
    pl_wino43_elems(N, Cin, H, W, &vin);
    pl_wino43_elems(N, Cout, H, W, &vout);
    float *V = nullptr, *M = nullptr;
    rc = pl_alloc(ctx, vin * sizeof(float), (void **)&V);
    if (rc != PL_OK) return rc;
    rc = pl_alloc(ctx, vout * sizeof(float), (void **)&M);
    if (rc != PL_OK) {
        pl_free(ctx, V);
        return rc;
    }
    rc = wino43_input_launch(ctx, xq, V, N, Cin, H, W);
    if (rc == PL_OK) rc = wino43_gemm_launch(ctx, V, uq, M, N, Cin, H, W, Cout);
    if (rc == PL_OK) rc = wino43_output_launch(ctx, M, yq, nullptr, N, Cout, H, W, make_epilogue(bias, scale, shift, resq, act, alpha));
    pl_free(ctx, M);
    pl_free(ctx, V);
    return rc;
}

int pl_conv2d_winograd_q4_filter_elems(int Cout, int Cin, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin > 0, PL_EINVAL, "pl_conv2d_winograd_q4_filter_elems: bad argument");
    *elems = (size_t)16 * (((size_t)Cin / 4 + 7) / 8 * 8) * Cout * 4;
    return PL_OK;
}

int pl_conv2d_prepare_winograd_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_winograd_q4_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL,
               "winograd Q4 filters need Cin %% 4 == 0 and Cout %% 4 == 0");
    const size_t pairs = (size_t)Cout * Cin;
    size_t elems = 0;
    pl_conv2d_winograd_q4_filter_elems(Cout, Cin, &elems);
    PL_REQUIRE(elems < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    PL_HIP(hipMemsetAsync(out, 0, elems * sizeof(float), ctx->stream));       // k-quad padding
    wino_filter_q4_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, ctx->stream>>>(w, out, (unsigned)pairs, Cin, Cout,
                                                                                   (Cin / 4 + 7) / 8 * 8);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_conv2d_winograd_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *uq, int Cout,
                              const float *bias, float *yq, const float *scale, const float *shift,
                              const float *resq, int act, double alpha) {
    PL_REQUIRE(ctx && xq && uq && yq, PL_EINVAL, "pl_conv2d_winograd_q4_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0, PL_EINVAL,
               "pl_conv2d_winograd_q4_f32: bad shape (Cin, Cout must be multiples of 4)");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "pl_conv2d_winograd_q4_f32: bad activation code");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(uq) |
                 reinterpret_cast<uintptr_t>(resq)) & 15u) == 0, PL_EINVAL, "Q4 tensors must be 16-byte aligned");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return winograd_q4_launch(ctx, xq, N, Cin, H, W, uq, Cout, bias, yq, scale, shift, resq, act, alpha);
}

// Plans are stored by configuration NAME so a cache survives re-ordering of the table.
int pl_conv2d_prepare_winograd_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_winograd_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin % 16 == 0, PL_EINVAL, "winograd filters need Cin %% 16 == 0");
    const size_t pairs = (size_t)Cout * Cin;
    PL_REQUIRE(pairs * 16 < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    wino_filter_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, ctx->stream>>>(w, out, (unsigned)pairs);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

}  // extern "C"
