// What the convolution translation units share (conv_direct.hip: implicit-GEMM kernels, launch plans, autotuner;
// conv_winograd.hip: every Winograd variant): the kernel argument block, the XCD-aware tile mapping, the fused tail on
// channel quads, and the few host functions that cross the two units.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "device_utils.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float *x, *w;
    float *y;         // NCHW output (fused pass) or compact split-K slabs [split][tile][BM*BN]
    int N, Cin, H, W, Cout, Ho, Wo;
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int cin_g, cout_g, groups;
    int K;            // cin_g*kh*kw
    int cols;         // N*Ho*Wo
    int HoWo, HW;
    int mtiles, ntiles, tiles;  // per group
    int tile_offset;  // first (group-major) tile index this launch covers
    int tile_count;   // tiles this launch covers (slab stride of the split pass)
    int splits, k_per_split;    // k_per_split: K elements (generic) or BK-chunks (tap-major)
    int x_bytes, w_bytes;
    // channel-quad (Q4) layout only: input quads per group / in total, output quads in total,
    // k-quads per group (real / padded), and whether a BK chunk always sits inside one filter tap
    int cqg, Cq, Coq, Qtot, Qpad, uni;
    int y_bytes;      // Q4 output size in bytes when it is under 2 GiB (buffer-addressed tail), else 0
    int rp_rq;        // > 0: row-packed small-Cin input (x = padded NHWC, H/W = padded extents); quads per filter row
    int xcd_cols;     // > 0: workgroup b (XCD b % 8) takes column tiles [xcd * xcd_cols, +xcd_cols) of EVERY group: the XCD that
                      // wrote a column block of the grouped GEMM's input also reads it (Winograd hand-offs, wino4_gemm_launch)
    FastDiv divKhw, divKw, divHoWo, divWo, divMt, divCpt;
    Epilogue ep;
};
// blockIdx.x -> (group, m-tile, n-tile).  XCD-aware: the 8 XCDs (private L2s)
// each walk a contiguous range of tiles, M-tiles fastest, so workgroups that
// are co-resident on an XCD share input pixels and filter panels in its L2.
struct TileCoord {
    unsigned g, local;  // group, index of the tile inside this launch
    int m0, col0;
};

// (bid, nblk): this workgroup's index among the nblk workgroups of ITS conv -- the whole grid, except in the two-conv
// launch (conv_q4_pair_kernel)
template <int BM, int BN>
__device__ __forceinline__ TileCoord tile_coord(const ConvArgs &p, unsigned bid = blockIdx.x, unsigned nblk = gridDim.x) {
    const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    TileCoord tc;
    if (p.xcd_cols > 0) {
        // column-block ownership (one unsplit pass over all groups, ntiles = 8 * xcd_cols): idx -> (m-tile fastest, group,
        // local column tile); every group's tiles of a column block run on ONE XCD
        const unsigned mt = idx % (unsigned)p.mtiles, r2 = idx / (unsigned)p.mtiles;
        const unsigned g = r2 % (unsigned)p.groups, ntl = r2 / (unsigned)p.groups;
        const unsigned nt = xcd * (unsigned)p.xcd_cols + ntl;
        tc.g = g;
        tc.local = (g * (unsigned)p.ntiles + nt) * (unsigned)p.mtiles + mt;
        tc.m0 = (int)mt * BM;
        tc.col0 = (int)nt * BN;
        return tc;
    }
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any grid
    tc.local = bid;
    const unsigned gt = bid + (unsigned)p.tile_offset;
    tc.g = gt / (unsigned)p.tiles;
    const unsigned t = gt - tc.g * (unsigned)p.tiles;
    const unsigned nt = p.divMt.div(t);
    tc.m0 = (int)(t - nt * (unsigned)p.mtiles) * BM;
    tc.col0 = (int)nt * BN;
    return tc;
}

// Fused tail on 4 consecutive output channels (layer.py:125-127, 93-95, 44-51 applied in that
// order, each its own rounding).  `valid` < 4 marks the last quad of a channel count that is not a
// multiple of 4: its padding lanes are written as zeros.
__device__ __forceinline__ float4 apply_epilogue4(const Epilogue &e, float4 bias, float4 scale, float4 shift,
                                                  float4 res, int valid, float4 v) {
    float r[4] = {v.x, v.y, v.z, v.w};
    const float bs[4] = {bias.x, bias.y, bias.z, bias.w}, sc[4] = {scale.x, scale.y, scale.z, scale.w};
    const float sh[4] = {shift.x, shift.y, shift.z, shift.w}, rs[4] = {res.x, res.y, res.z, res.w};
    if (e.bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __fadd_rn(r[i], bs[i]);
    }
    if (e.scale) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __fmul_rn(r[i], sc[i]);
    }
    if (e.shift) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __fadd_rn(r[i], sh[i]);
    }
    if (e.res && !e.res_post) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __fadd_rn(r[i], rs[i]);
    }
    if (e.act == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = relu_ref(r[i]);
    } else if (e.act == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = leaky_ref(r[i], e.la, e.lb);
    }
    if (e.res && e.res_post) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __fadd_rn(r[i], rs[i]);
    }
    if (valid < 4) {
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i >= valid) r[i] = 0.f;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// The per-channel parameters of one output channel (clamped to the group's last channel so that
// rows past Cout read something harmless).
__device__ __forceinline__ void load_chan_params(const Epilogue &e, int c, float &bias, float &scale, float &shift) {
    bias = e.bias ? e.bias[c] : 0.f;
    scale = e.scale ? e.scale[c] : 1.f;
    shift = e.shift ? e.shift[c] : 0.f;
}

}  // namespace

// host functions that cross translation units (hidden: not part of the C ABI)
namespace plhip {
// conv_direct.hip -- any convolution on the implicit-GEMM kernels (layout 0 OIHW, 1 tap-major, 2 channel-quad, 3 = NCHW
// Winograd filters, handed on to winograd_launch, 6 row-packed stem) under its cached / autotuned launch plan
__attribute__((visibility("hidden"))) int conv_launch(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w,
                                                        int Cout, int kh, int kw, const float *bias, float *y, int sh, int sw, int dh,
                                                        int dw, int pt, int pl, int pb, int pr, int group, const float *scale,
                                                        const float *shift, const float *res, int act, double alpha, int layout);
__attribute__((visibility("hidden"))) int ensure_lds_attr(const void *kern, int bytes);
// conv_winograd.hip -- F(2x2,3x3) on NCHW tensors (ConvFused w_layout 3)
__attribute__((visibility("hidden"))) int winograd_launch(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *U,
                                                            int Cout, const float *bias, float *y, const float *scale,
                                                            const float *shift, const float *res, int act, double alpha);
}  // namespace plhip
