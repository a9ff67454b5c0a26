// Winograd F(4x4,3x3) transforms through LDS, chained across convolutions -- included by
// conv_winograd.hip inside its anonymous namespace (uses w4_at4 / w4_at4_row / w4_bt / w4_bt_row,
// apply_epilogue4, Epilogue, FastDiv).
//
// The unchained pipeline of one conv is  input transform -> 36 grouped GEMMs -> output transform
// (winograd4_q4_launch): three kernels, and the activation tensor between two Winograd convs
// travels  M --(A^T m A + tail)--> y --(B^T d B)--> V  through HBM / L2 twice although the second
// transform needs nothing but y.  Here ONE workgroup owns whole planes: G channel quads of one
// image.  It
//   P0  pulls its 36 x (G x tiles) products M and the residual planes into LDS by LDS-DMA
//       (buffer_load_dwordx4 ... lds: lane-linear, zero fill by the range check -- the residual
//       lands in a zero-bordered plane [row][x mod 4][x div 4], so every later access of a wave
//       is to consecutive 16-byte cells),
//   P1  output-transforms tile rows out of LDS (thread = (row a of a 4x4 output tile, quad, tile),
//       wave-uniform a), applies the conv's fused tail (layer.py:125-127, 93-95, 44-51) and
//       leaves y in the plane (pixels past the map's edge as zeros: they are the next conv's padding),
//   P2  input-transforms the NEXT conv's 6x6 tiles out of the plane (thread = (row a of the
//       transformed tile, quad, tile)) and writes V with 16-byte stores, consecutive lanes =
//       consecutive tiles,
//   P3  writes y itself -- only when something other than the next conv reads it.
// The same kernel serves a lone input transform (FROM_M = false: P0 pulls x into the plane, then
// P2) and a lone output transform (P0, P1, P3), so every global access of the transform family is
// a coalesced 16-byte one and all the arithmetic runs out of LDS.
// Arithmetic (operation order, roundings) is exactly that of wino4_input_q4_kernel /
// wino4_output_q4_kernel: chained and unchained plans agree bit for bit (tests/test_gpu_wino_chain.py).

struct Wino4ChainArgs {
    const float *M;        // [36][Cq][T][4] products of the producing conv        (FROM_M)
    const float *x;        // [N][Cq][H][W][4] activations of the consuming conv   (!FROM_M)
    float4 *y;             // [N][Cq][H][W][4] fused-tail output, or null: nobody else reads it
    float4 *V;             // [36][Cq][T][4] transformed input of the consuming conv, or null
    int N, Cq, H, W, th, tw, tiles, T;
    int G;                 // channel quads per workgroup (divides Cq)
    int gt;                // G * tiles
    int per;               // gt rounded up to 64: items of one transformed row (wave-uniform row index)
    int R, XP, S;          // plane rows (4 th + 2), plane columns (4 tw + 2), cells per x-phase (tw + 1)
    int plane;             // R * 4 * S cells of 16 bytes per channel quad
    int ipx;               // > 0: images per XCD -- workgroup b (XCD b % 8) takes image (b % 8) * ipx + (b / 8) % ipx, so that the
                           // images whose tile columns one XCD's GEMM workgroups produced are transformed on that XCD
    unsigned src_bytes, res_bytes;     // buffer sizes for the range check (M or x; residual)
    FastDiv divGt, divTiles, divTw, divPer, divPlane, div4S, divS, divHW, divW;
    Epilogue ep;
};

// The transforms on channel PAIRS: a float4 cell is two register pairs, and on pairs every operation is one packed
// instruction (v_pk_add / v_pk_mul) with no operand shuffling.  Expression for expression the formulas of w4_bt_row / w4_bt /
// w4_at4_row / w4_at4 (same operations, same order, no contraction): results are bit-identical to the register kernels'.
typedef float wc_v2 __attribute__((ext_vector_type(2)));
template <int A>
__device__ __forceinline__ wc_v2 wc_bt_row2(const wc_v2 (&d)[6]) {
    if constexpr (A == 0) return 4.f * d[0] - 5.f * d[2] + d[4];
    else if constexpr (A == 1) return -4.f * (d[1] + d[2]) + d[3] + d[4];
    else if constexpr (A == 2) return 4.f * (d[1] - d[2]) - d[3] + d[4];
    else if constexpr (A == 3) return 2.f * (d[3] - d[1]) - d[2] + d[4];
    else if constexpr (A == 4) return 2.f * (d[1] - d[3]) - d[2] + d[4];
    else return 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void wc_bt2(const wc_v2 (&d)[6], wc_v2 (&o)[6]) {
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    o[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    o[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <int A>
__device__ __forceinline__ wc_v2 wc_at_row2(const wc_v2 (&m)[6]) {
    if constexpr (A == 0) return (m[0] + (m[1] + m[2])) + (m[3] + m[4]);
    else {
        const wc_v2 q = m[1] - m[2], t = m[3] - m[4], pp = m[1] + m[2], r = m[3] + m[4];
        if constexpr (A == 1) return q + 2.f * t;
        else if constexpr (A == 2) return pp + 4.f * r;
        else return q + 8.f * t + m[5];
    }
}
__device__ __forceinline__ void wc_at2(const wc_v2 (&m)[6], wc_v2 (&o)[4]) {
    const wc_v2 pp = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], t = m[3] - m[4];
    o[0] = (m[0] + pp) + r;
    o[1] = q + 2.f * t;
    o[2] = pp + 4.f * r;
    o[3] = q + 8.f * t + m[5];
}
__device__ __forceinline__ wc_v2 wc_lo(float4 v) { return (wc_v2){v.x, v.y}; }
__device__ __forceinline__ wc_v2 wc_hi(float4 v) { return (wc_v2){v.z, v.w}; }
// the tail a residual net's convolutions carry (scale, shift, [residual,] ReLU; no bias) written straight: what apply_epilogue4
// does for that combination, without its per-element option selects
__device__ __forceinline__ wc_v2 wc_plain_tail(wc_v2 v, wc_v2 sc, wc_v2 sh, wc_v2 rs, bool has_res) {
    v = v * sc;
    v = v + sh;
    if (has_res) v = v + rs;
    const wc_v2 z = v * 0.f;
    return (wc_v2){v.x > 0.f ? v.x : z.x, v.y > 0.f ? v.y : z.y};
}

// P1: row A of the 4x4 output tile (quad cql, tile (ty, tx)) out of the LDS slab, fused tail, into the plane
template <int A, bool PLAIN>
__device__ __forceinline__ void wc_out_row(const Wino4ChainArgs &p, const float4 *slab, float4 *plane, const float4 *prm,
                                           unsigned r, unsigned cql, unsigned ty, unsigned tx) {
    wc_v2 slo[6], shi[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        wc_v2 mlo[6], mhi[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float4 m = slab[(unsigned)(k * 6 + b) * (unsigned)p.gt + r];
            mlo[k] = wc_lo(m);
            mhi[k] = wc_hi(m);
        }
        slo[b] = wc_at_row2<A>(mlo);
        shi[b] = wc_at_row2<A>(mhi);
    }
    wc_v2 olo[4], ohi[4];
    wc_at2(slo, olo);
    wc_at2(shi, ohi);
    const float4 bias = prm[cql], scale = prm[p.G + cql], shift = prm[2 * p.G + cql];
    const int h = (int)ty * 4 + A;
    float4 *row = plane + ((size_t)cql * p.R + h + 1) * 4 * p.S;
    const bool has_res = p.ep.res != nullptr;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int w = (int)tx * 4 + b, xp = w + 1;
        float4 *cell = row + (xp & 3) * p.S + (xp >> 2);
        const float4 rs = *cell;                  // the residual (P0 put it there) or zero
        float4 v;
        if constexpr (PLAIN) {
            const wc_v2 lo = wc_plain_tail(olo[b], wc_lo(scale), wc_lo(shift), wc_lo(rs), has_res);
            const wc_v2 hi = wc_plain_tail(ohi[b], wc_hi(scale), wc_hi(shift), wc_hi(rs), has_res);
            v = make_float4(lo.x, lo.y, hi.x, hi.y);
        } else {
            v = apply_epilogue4(p.ep, bias, scale, shift, rs, 4, make_float4(olo[b].x, olo[b].y, ohi[b].x, ohi[b].y));
        }
        const bool ok = h < p.H && w < p.W;       // past the map's edge: the next conv's zero padding
        *cell = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// P2: row A of the transformed 6x6 tile of the consuming conv out of the plane -> six 16-byte stores
template <int A>
__device__ __forceinline__ void wc_in_row(const Wino4ChainArgs &p, const float4 *plane, unsigned cql, unsigned ty,
                                          unsigned tx, size_t vbase, size_t vplane) {
    wc_v2 mlo[6], mhi[6];
    const float4 *rows = plane + ((size_t)cql * p.R + ty * 4) * 4 * p.S;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int xp = (int)tx * 4 + j;
        const float4 *col = rows + (xp & 3) * p.S + (xp >> 2);
        wc_v2 dlo[6], dhi[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float4 d = col[(size_t)k * 4 * p.S];
            dlo[k] = wc_lo(d);
            dhi[k] = wc_hi(d);
        }
        mlo[j] = wc_bt_row2<A>(dlo);
        mhi[j] = wc_bt_row2<A>(dhi);
    }
    wc_v2 olo[6], ohi[6];
    wc_bt2(mlo, olo);
    wc_bt2(mhi, ohi);
#pragma unroll
    for (int j = 0; j < 6; ++j) p.V[(size_t)(A * 6 + j) * vplane + vbase] = make_float4(olo[j].x, olo[j].y, ohi[j].x, ohi[j].y);
}

template <bool FROM_M>
__global__ void __launch_bounds__(512) wino4_chain_kernel(const Wino4ChainArgs p) {
    extern __shared__ float4 wc_lds[];
    const unsigned tid = threadIdx.x, bd = blockDim.x, lane = tid & 63u;
    unsigned n = blockIdx.y, cq0 = blockIdx.x * (unsigned)p.G;
    if (p.ipx > 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, xcd = b & 7u, idx = b >> 3;
        n = xcd * (unsigned)p.ipx + idx % (unsigned)p.ipx;
        cq0 = (idx / (unsigned)p.ipx) * (unsigned)p.G;
    }
    const unsigned gt = (unsigned)p.gt, HW = (unsigned)(p.H * p.W);
    float4 *plane = wc_lds;                                   // [G][R][4][S]
    float4 *slab = wc_lds + (size_t)p.G * p.plane;            // [36][G * tiles]
    float4 *prm = slab + (FROM_M ? 36u * gt : 0u);            // [3][G]: bias, scale, shift of the quads

    // ---- P0: LDS-DMA.  A wave writes 64 consecutive 16-byte cells; lanes past the end are masked off,
    //      lanes whose source does not exist (border, padding, no residual) read out of range = zero ----
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(FROM_M ? p.M : p.x), 0, p.src_bytes, 0x00020000);
    if (FROM_M) {
        for (unsigned i0 = tid - lane; i0 < 36u * gt; i0 += bd) {
            const unsigned i = i0 + lane;
            if (i < 36u * gt) {
                unsigned f, r, cql, tl;
                p.divGt.divmod(i, f, r);
                p.divTiles.divmod(r, cql, tl);
                const unsigned src = ((f * (unsigned)p.Cq + cq0 + cql) * (unsigned)p.T + n * (unsigned)p.tiles + tl) << 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (__attribute__((address_space(3))) float *)(slab + i0), 16,
                                                         (int)src, 0, 0, 0);
            }
        }
        if (tid < 3u * (unsigned)p.G) {
            const unsigned which = tid / (unsigned)p.G, cql = tid - which * (unsigned)p.G;
            const float *src = which == 0 ? p.ep.bias : which == 1 ? p.ep.scale : p.ep.shift;
            const float fill = which == 1 ? 1.f : 0.f;
            prm[tid] = src ? *reinterpret_cast<const float4 *>(src + (size_t)(cq0 + cql) * 4) : make_float4(fill, fill, fill, fill);
        }
    }
    {
        // the plane: the residual (FROM_M) or x itself, zero outside the map
        const __amdgpu_buffer_rsrc_t prsrc =
            FROM_M ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.res_bytes, 0x00020000) : srsrc;
        const unsigned cells = (unsigned)p.G * (unsigned)p.plane;
        for (unsigned c0 = tid - lane; c0 < cells; c0 += bd) {
            const unsigned c = c0 + lane;
            if (c < cells) {
                unsigned cql, rem, r_, rem2, m, s;
                p.divPlane.divmod(c, cql, rem);
                p.div4S.divmod(rem, r_, rem2);
                p.divS.divmod(rem2, m, s);
                const int h = (int)r_ - 1, w = (int)(4 * s + m) - 1;
                const bool in = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                const unsigned src = (((n * (unsigned)p.Cq + cq0 + cql) * (unsigned)p.H + (unsigned)h) * (unsigned)p.W + (unsigned)w) << 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(prsrc, (__attribute__((address_space(3))) float *)(plane + c0), 16,
                                                         in ? (int)src : (int)0x80000000, 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- P1: products -> y (in the plane) ----
    if (FROM_M) {
        const bool plain = !p.ep.bias && p.ep.scale && p.ep.shift && p.ep.act == 1 && !(p.ep.res && p.ep.res_post);
        for (unsigned j = tid; j < 4u * (unsigned)p.per; j += bd) {
            unsigned a, r;
            p.divPer.divmod(j, a, r);
            if (r < gt) {
                unsigned cql, tl, ty, tx;
                p.divTiles.divmod(r, cql, tl);
                p.divTw.divmod(tl, ty, tx);
                if (plain) {
                    switch (a) {
                    case 0: wc_out_row<0, true>(p, slab, plane, prm, r, cql, ty, tx); break;
                    case 1: wc_out_row<1, true>(p, slab, plane, prm, r, cql, ty, tx); break;
                    case 2: wc_out_row<2, true>(p, slab, plane, prm, r, cql, ty, tx); break;
                    default: wc_out_row<3, true>(p, slab, plane, prm, r, cql, ty, tx); break;
                    }
                } else {
                    switch (a) {
                    case 0: wc_out_row<0, false>(p, slab, plane, prm, r, cql, ty, tx); break;
                    case 1: wc_out_row<1, false>(p, slab, plane, prm, r, cql, ty, tx); break;
                    case 2: wc_out_row<2, false>(p, slab, plane, prm, r, cql, ty, tx); break;
                    default: wc_out_row<3, false>(p, slab, plane, prm, r, cql, ty, tx); break;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- P2: y (or x) -> V of the consuming conv ----
    if (p.V) {
        const size_t vplane = (size_t)p.Cq * p.T;
        for (unsigned j = tid; j < 6u * (unsigned)p.per; j += bd) {
            unsigned a, r;
            p.divPer.divmod(j, a, r);
            if (r < gt) {
                unsigned cql, tl, ty, tx;
                p.divTiles.divmod(r, cql, tl);
                p.divTw.divmod(tl, ty, tx);
                const size_t vbase = (size_t)(cq0 + cql) * p.T + (size_t)n * p.tiles + tl;
                switch (a) {
                case 0: wc_in_row<0>(p, plane, cql, ty, tx, vbase, vplane); break;
                case 1: wc_in_row<1>(p, plane, cql, ty, tx, vbase, vplane); break;
                case 2: wc_in_row<2>(p, plane, cql, ty, tx, vbase, vplane); break;
                case 3: wc_in_row<3>(p, plane, cql, ty, tx, vbase, vplane); break;
                case 4: wc_in_row<4>(p, plane, cql, ty, tx, vbase, vplane); break;
                default: wc_in_row<5>(p, plane, cql, ty, tx, vbase, vplane); break;
                }
            }
        }
    }

    // ---- P3: y to memory, pixel order (the G planes of this workgroup are one contiguous run) ----
    if (FROM_M && p.y) {
        float4 *yp = p.y + ((size_t)n * p.Cq + cq0) * HW;
        for (unsigned i = tid; i < (unsigned)p.G * HW; i += bd) {
            unsigned cql, px, h, w;
            p.divHW.divmod(i, cql, px);
            p.divW.divmod(px, h, w);
            const unsigned xp = w + 1;
            yp[i] = plane[((size_t)cql * p.R + h + 1) * 4 * p.S + (xp & 3) * p.S + (xp >> 2)];
        }
    }
}
