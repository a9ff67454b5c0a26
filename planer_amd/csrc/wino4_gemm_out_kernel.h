// Winograd F(4x4,3x3): the 36 per-frequency GEMMs AND the output transform + fused tail in one kernel, for maps of few tiles
// (batch-1 detection nets) -- included by conv_winograd.hip inside its anonymous namespace (uses w4_at4 / w4_at4_row,
// apply_epilogue4, load_chan_params, Epilogue, FastDiv).
//
// The staged pipeline runs a 3x3 conv (layer.Conv2d layer.py:22-26 -> util.conv_for util.py:17-44, + BatchNorm :125-127,
// LeakyReLU :48-51, Add :93-95) as input transform -> 36 grouped GEMMs -> output transform.  On a 52x52 or 26x26 map at batch 1
// each of the three is a latency chain of a few microseconds (~20 us per conv against ~7 for a 1x1), and M
// -- 2.25x the activation -- exists only to be read back at once.  Here ONE workgroup owns 16 output channels x 16 tiles for
// ALL 36 frequencies: 36 accumulator blocks of v_mfma_f32_16x16x4_f32 (a lane holds one channel quad of one tile: exactly what
// the output transform and the Q4 store want), so A^T m A needs nothing from another workgroup and M never leaves the chip.
//   * 12 waves, wave w = frequencies w, w + 12, w + 24 (three accumulator blocks).  Both operands come straight from global
//     memory as 16-byte loads -- a lane's four k-values of one (frequency, k-quad): U[f][q][cout] and V[f][q][tile], the layouts
//     the staged pipeline already has -- three K steps (3 x 4 k-quads) in flight per wave (first version: 8 waves x 9
//     frequencies x 2 K halves with two register sets = 256 registers and spills); nothing is staged in LDS and the K loop has no
//     barrier.  (No operand reuse across workgroups beyond L2: right for ~100-200 workgroups on an idle chip, wrong for
//     batch 32 -- the tiled per-frequency GEMM keeps those.)
//   * the accumulators go to LDS ([f][channel quad][tile] cells of 16 bytes, 36 KB); thread (row a of the 4x4 output tile,
//     channel quad, tile) applies A^T . A, the fused tail, and stores four consecutive pixels of one row.
// Summation order: k-quads in ascending order, one fmaf chain per product -- deterministic, not the tiled kernel's chunked order
// (parity with the oracle under the conv tolerance; tests/test_gpu_wino_gemm_out.py).
struct W4GOArgs {
    const float *V, *U;          // [36][Cq][T][4], [36][Qpad][Cout][4]
    float4 *y;                   // [N][Coq][H][W][4]
    int N, Cq, Qpad, Cout, Coq, T, H, W, th, tw;
    int mtiles;                  // ceil(Cout / 16)
    int steps;                   // K steps of 4 k-quads
    unsigned v_bytes, u_bytes, y_bytes;
    FastDiv divMt, divTw, divTh;
    Epilogue ep;
};

constexpr int W4GO_WAVES = 12, W4GO_FPW = 3, W4GO_DEPTH = 3;       // waves, frequencies per wave, K steps in flight
constexpr int W4GO_LDS_CELLS = 36 * 4 * 16;                        // float4 cells

typedef float w4go_f32x4 __attribute__((ext_vector_type(4)));

template <int A>
__device__ __forceinline__ void w4go_finish(const W4GOArgs &p, const float4 *S, int kk, int li, int m0, int t0) {
    const int t = t0 + li, coq = (m0 >> 2) + kk;
    float4 s[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float4 m[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) m[k] = S[((k * 6 + b) * 4 + kk) * 16 + li];
        s[b] = w4_at4_row<A>(m);
    }
    if (t >= p.T || coq >= p.Coq) return;
    unsigned n, r, ty, tx;
    p.divTw.divmod((unsigned)t, r, tx);
    p.divTh.divmod(r, n, ty);
    float bs[4], sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) load_chan_params(p.ep, coq * 4 + e, bs[e], sc[e], sh[e]);
    const float4 bias = make_float4(bs[0], bs[1], bs[2], bs[3]), scale = make_float4(sc[0], sc[1], sc[2], sc[3]);
    const float4 shift = make_float4(sh[0], sh[1], sh[2], sh[3]);
    const int ho = (int)ty * 4 + A, wo = (int)tx * 4;
    const unsigned row = ((n * (unsigned)p.Coq + (unsigned)coq) * (unsigned)p.H + (unsigned)ho) * (unsigned)p.W + (unsigned)wo;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? p.y_bytes : 0u, 0x00020000);
    int off[4];
    float4 rs[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        off[b] = (ho < p.H && wo + b < p.W) ? (int)((row + b) << 4) : (int)0x80000000;
        rs[b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[b], 0, 0));
    }
    float4 o[4];
    w4_at4(s, o);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float4 v = apply_epilogue4(p.ep, bias, scale, shift, rs[b], 4, o[b]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), yrsrc, off[b], 0, 0);
    }
}

__global__ void __launch_bounds__(W4GO_WAVES * 64) wino4_gemm_out_kernel(const W4GOArgs p) {
    __shared__ __attribute__((aligned(16))) float4 S[W4GO_LDS_CELLS];          // [f][channel quad][tile]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kk = lane >> 4;
    unsigned nt, mt;
    p.divMt.divmod(blockIdx.x, nt, mt);
    const int m0 = (int)mt * 16, t0 = (int)nt * 16;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.V), 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.U), 0, p.u_bytes, 0x00020000);
    // this lane's cell inside a (frequency, K step) block: k-quad kk, channel m0 + li / tile t0 + li -- the vector offset; the
    // frequency plane and the K step ride in the scalar offset
    const unsigned uplane = (unsigned)p.Qpad * (unsigned)p.Cout * 16u, vplane = (unsigned)p.Cq * (unsigned)p.T * 16u;
    const unsigned ustep = 4u * (unsigned)p.Cout * 16u, vstep = 4u * (unsigned)p.T * 16u;
    const int ulane = m0 + li < p.Cout ? (int)(((unsigned)kk * (unsigned)p.Cout + (unsigned)(m0 + li)) * 16u) : OOB;
    const int vlane = t0 + li < p.T ? (int)(((unsigned)kk * (unsigned)p.T + (unsigned)(t0 + li)) * 16u) : OOB;
    const int qlast = p.Cq - 1 - kk;                 // k-quad 4 s + kk exists in V while 4 s <= qlast (U is zero padded to Qpad >= 4 steps)

    w4go_f32x4 acc[W4GO_FPW];
#pragma unroll
    for (int i = 0; i < W4GO_FPW; ++i) acc[i] = (w4go_f32x4){0.f, 0.f, 0.f, 0.f};
    float4 a[W4GO_DEPTH][W4GO_FPW], b[W4GO_DEPTH][W4GO_FPW];
    auto fetch = [&](int s, auto settag) {
        constexpr int set = decltype(settag)::value;
        const bool live = s < p.steps && 4 * s <= qlast;
        const int uo = live ? ulane : OOB, vo = live ? vlane : OOB;
#pragma unroll
        for (int i = 0; i < W4GO_FPW; ++i) {
            const unsigned f = (unsigned)wave + 12u * (unsigned)i;
            a[set][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, uo, (int)(f * uplane + (unsigned)s * ustep), 0));
            b[set][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, vo, (int)(f * vplane + (unsigned)s * vstep), 0));
        }
    };
    auto mult = [&](auto settag) {
        constexpr int set = decltype(settag)::value;
#pragma unroll
        for (int i = 0; i < W4GO_FPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[set][i].x, b[set][i].x, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < W4GO_FPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[set][i].y, b[set][i].y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < W4GO_FPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[set][i].z, b[set][i].z, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < W4GO_FPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[set][i].w, b[set][i].w, acc[i], 0, 0, 0);
    };
    using std::integral_constant;
    fetch(0, integral_constant<int, 0>{});
    fetch(1, integral_constant<int, 1>{});
    fetch(2, integral_constant<int, 2>{});
    for (int s = 0; s < p.steps; s += 3) {          // (steps past the end: every offset out of range -> zeros, three spare MFMA groups at most)
        mult(integral_constant<int, 0>{});
        fetch(s + 3, integral_constant<int, 0>{});
        mult(integral_constant<int, 1>{});
        fetch(s + 4, integral_constant<int, 1>{});
        mult(integral_constant<int, 2>{});
        fetch(s + 5, integral_constant<int, 2>{});
    }
    // ---- accumulators -> LDS: cell [f][kk][li] = this lane's channel quad (rows 4 kk .. + 3 of the C block) of tile li ----
#pragma unroll
    for (int i = 0; i < W4GO_FPW; ++i) {
        const int f = wave + 12 * i;
        S[(f * 4 + kk) * 16 + li] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    __syncthreads();
    // ---- output transform + tail: waves 0..3 = row a of the 4x4 output tile, lane = (channel quad, tile) ----
    switch (wave) {
    case 0: w4go_finish<0>(p, S, kk, li, m0, t0); break;
    case 1: w4go_finish<1>(p, S, kk, li, m0, t0); break;
    case 2: w4go_finish<2>(p, S, kk, li, m0, t0); break;
    case 3: w4go_finish<3>(p, S, kk, li, m0, t0); break;
    default: break;
    }
}
