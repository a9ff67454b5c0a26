// Fused 1-D Winograd F(4,3) convolution along W on channel-quad tensors -- included by conv_winograd.hip inside its
// anonymous namespace (shares ConvArgs, tile_coord, the Q4 epilogue helpers).
//
// For 3x3 / stride 1 / pad 1 / dilation 1 / group 1 convs (util.conv_for, util.py:17-44).  The transform is applied along W
// only: for every filter row r and input channel c the six frequencies of a 6-pixel segment give 4 outputs -- SIX GEMMs
// M_f = U_f (Cout x 3 Cin) . V_f (3 Cin x tiles), 6 products per 4 outputs per 3 taps = 2x fewer MFMAs than the direct conv --
// that SHARE ONE GATHER; nothing but x, the packed filter and y touches HBM.  (Its F(2,3) sibling -- four GEMMs, 1.5x fewer
// MFMAs, conv_w1d_kernel, w_layout 5 -- lost to this kernel on every shape it was picked for and was removed in round 4:
// layer1 62 us against 54 us)
// K runs (filter row, input channel); the filter is packed once per model as
//     uq[f][q = r*Cin/4 + c/4][co][4]   (zero padded to a multiple of 8 k-quads).


// =====================================================================================================
// Fused 1-D Winograd F(4,3) along W: 6 frequencies, 4 outputs per tile, 6 products per 4 outputs per 3 taps
// = 2x fewer MFMAs than the direct conv (1.5x for F(2,3) above).  Same idea -- the six GEMMs share one
// gather, a wave keeps a 32x32 accumulator block per frequency (6 x 16 registers), the output transform
// is lane-local -- with the interpolation points 0, +-1, +-2, inf (matrices as in the 2-D F(4x4,3x3)
// pipeline).  Six planes per operand only fit LDS at BK = 8 (48 KB, double buffered), so a chunk is one
// k-quad pair = 24 MFMAs per wave; loads are split by role: waves 0-1 gather 6 pixels of their (tile,
// k-quad) and transform them, waves 2-3 fetch the 6 frequencies of their (filter row, k-quad).
// A workgroup owns 64 output channels x 64 tiles (256 pixels of one output row each).
struct W1d4Cfg {
    static constexpr int BM = 64, BN = 64, KG = 2, F = 6;
    static constexpr int A_PLANE = KG * BM * 4, B_PLANE = KG * BN * 4;
    static constexpr int A_ELEMS = F * A_PLANE, B_ELEMS = F * B_PLANE;
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4 + 3 * BM * 4;     // 48 KB + epilogue parameters
};

__device__ __forceinline__ float4 f4s(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4a(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4m(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
// s*a + b as ONE v_fma_f32 per lane: fp32 VALU work beside fp32 MFMAs is not free on gfx950 (both use the
// SIMD's FMA lanes, tools/ubench/pc_interference.hip), so the input transform is written in as few ops as it takes
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {
    return make_float4(__builtin_fmaf(s, a.x, b.x), __builtin_fmaf(s, a.y, b.y), __builtin_fmaf(s, a.z, b.z), __builtin_fmaf(s, a.w, b.w));
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_w1d4_kernel(const ConvArgs p) {
    using C = W1d4Cfg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * C::A_ELEMS;
    float *prm = smem + 2 * (C::A_ELEMS + C::B_ELEMS);       // [3][BM], published by the K loop's barriers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool b_role = wave < 2;                            // wave-uniform
    const int kq = wave & 1;                                 // k-quad of the chunk this thread stages

    const TileCoord tc = tile_coord<C::BM, C::BN>(p);
    const int m0 = tc.m0, col0 = tc.col0;
    const int nchunks = (p.Qtot + C::KG - 1) / C::KG;
    if (tid < C::BM) {
        float b, sc, sh;
        load_chan_params(p.ep, min(m0 + tid, p.Cout - 1), b, sc, sh);
        prm[tid] = b; prm[C::BM + tid] = sc; prm[2 * C::BM + tid] = sh;
    }

    const int j = col0 + lane;
    int ho = -(1 << 20), pixbase = 0, cokmask = 0;
    if (j < p.cols) {
        unsigned n, rem, h, t;
        p.divHoWo.divmod((unsigned)j, n, rem);               // cols = N * H * Tw, Tw = ceil(W / 4)
        p.divWo.divmod(rem, h, t);
        ho = (int)h;
        pixbase = ((int)n * p.Cq * p.H + ho) * p.W + 4 * (int)t - 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) cokmask |= ((unsigned)(4 * (int)t - 1 + k) < (unsigned)p.W) << k;
    }
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    int aoff[C::F];
#pragma unroll
    for (int i = 0; i < C::F; ++i)
        aoff[i] = m0 + lane < p.Cout ? (((i * p.Qpad + kq) * p.Cout + m0 + lane) << 4) : OOB;

    float4 st0[C::F], st1[C::F];                             // staging: 6 pixels (B role) or 6 frequencies (A role)
    auto load_chunk = [&](int c, float4 (&st)[C::F]) {
        if (b_role) {
            const int q = c * C::KG + kq;
            const int r = (int)p.divCpt.div((unsigned)q);
            const int cq = q - r * p.cqg;
            const int dy = q < p.Qtot ? r - 1 : (1 << 15);
            const bool rok = (unsigned)(ho + dy) < (unsigned)p.H;
            const int vrow = (int)((unsigned)(pixbase + (cq * p.H + dy) * p.W) << 4);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                st[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       xrsrc, rok && ((cokmask >> k) & 1) ? vrow + 16 * k : OOB, 0, 0));
        } else {
            const int ksoff = (c * C::KG * p.Cout) << 4;
#pragma unroll
            for (int i = 0; i < C::F; ++i)
                st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, aoff[i], ksoff, 0));
        }
    };
    auto store_chunk = [&](int buf, const float4 (&st)[C::F]) {
        if (b_role) {                                        // V = B^T d
            const float4 d0 = st[0], d1 = st[1], d2 = st[2], d3 = st[3], d4 = st[4], d5 = st[5];
            const float4 a = f4fma(-4.f, d2, d4), b = f4fma(-4.f, d1, d3);
            const float4 c2 = f4m(d4, d2), e1 = f4m(d3, d1);
            float4 *bp = reinterpret_cast<float4 *>(Bs + buf * C::B_ELEMS) + kq * C::BN + lane;
            bp[0 * (C::B_PLANE / 4)] = f4fma(4.f, d0, f4fma(-5.f, d2, d4));
            bp[1 * (C::B_PLANE / 4)] = f4a(a, b);
            bp[2 * (C::B_PLANE / 4)] = f4m(a, b);
            bp[3 * (C::B_PLANE / 4)] = f4fma(2.f, e1, c2);
            bp[4 * (C::B_PLANE / 4)] = f4fma(-2.f, e1, c2);
            bp[5 * (C::B_PLANE / 4)] = f4fma(4.f, d1, f4fma(-5.f, d3, d5));
        } else {
            float4 *ap = reinterpret_cast<float4 *>(As + buf * C::A_ELEMS) + kq * C::BM + lane;
#pragma unroll
            for (int i = 0; i < C::F; ++i) ap[i * (C::A_PLANE / 4)] = st[i];
        }
    };

    f32x16 acc[C::F];
#pragma unroll
    for (int f = 0; f < C::F; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (lhi * C::BM + wm * 32 + l31) * 4;
    const int b_off = (lhi * C::BN + wn * 32 + l31) * 4;
    const int last = nchunks - 1;

    auto step = [&](auto parity, int k) {
        constexpr int P = decltype(parity)::value;
        float4 af[C::F], bf[C::F];
        const float *Ab = As + P * C::A_ELEMS + a_off, *Bb = Bs + P * C::B_ELEMS + b_off;
#pragma unroll
        for (int f = 0; f < C::F; ++f) {
            af[f] = *reinterpret_cast<const float4 *>(Ab + f * C::A_PLANE);
            bf[f] = *reinterpret_cast<const float4 *>(Bb + f * C::B_PLANE);
        }
        if constexpr (P == 0) {
            store_chunk(1, st1);
            load_chunk(min(k + 2, last), st0);
        } else {
            store_chunk(0, st0);
            load_chunk(min(k + 2, last), st1);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int f = 0; f < C::F; ++f) {
                const float av = s4 == 0 ? af[f].x : s4 == 1 ? af[f].y : s4 == 2 ? af[f].z : af[f].w;
                const float bv = s4 == 0 ? bf[f].x : s4 == 1 ? bf[f].y : s4 == 2 ? bf[f].z : bf[f].w;
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[f], 0, 0, 0);
            }
        __syncthreads();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    load_chunk(0, st0);
    load_chunk(min(1, last), st1);
    store_chunk(0, st0);
    __syncthreads();
    int k = 0;
    for (; k + 1 <= last; k += 2) {
        step(P0{}, k);
        step(P1{}, k + 1);
    }
    if (k <= last) step(P0{}, k);

    // ---- epilogue: y = A^T m per lane (4 pixels of one output row), fused tail, b128 stores.  Branch-free:
    //      the residual is fetched and y is written through buffer descriptors whose range check drops what falls
    //      outside the tensor (a predicated plain load / store costs a branch and a wait per element), and all 16
    //      residual quads of the lane are in flight before the first is used -- one memory round trip per tile ----
    const int jc = min(col0 + wn * 32 + l31, p.cols - 1);
    const bool live = col0 + wn * 32 + l31 < p.cols;
    unsigned n, rem, h, t;
    p.divHoWo.divmod((unsigned)jc, n, rem);
    p.divWo.divmod(rem, h, t);
    const int wo = 4 * (int)t;
    const float4 *prm4 = reinterpret_cast<const float4 *>(prm);
    const unsigned out_bytes = (unsigned)p.N * (unsigned)p.Coq * (unsigned)(p.H * p.W) * 16u;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? out_bytes : 0u, 0x00020000);
    const unsigned obase = (n * (unsigned)p.Coq * (unsigned)p.H + h) * (unsigned)p.W + (unsigned)wo;
    int off[4][4];
    float4 rs[4][4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int Rt = wm * 32 + 8 * rq + 4 * lhi;
        const unsigned idx = obase + (unsigned)((m0 + Rt) >> 2) * (unsigned)(p.H * p.W);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            off[rq][b] = (live && m0 + Rt < p.Cout && wo + b < p.W) ? (int)((idx + b) << 4) : OOB;
            rs[rq][b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[rq][b], 0, 0));
        }
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int Rt = wm * 32 + 8 * rq + 4 * lhi;
        const float4 bias = prm4[Rt >> 2], scale = prm4[(C::BM + Rt) >> 2], shift = prm4[(2 * C::BM + Rt) >> 2];
        const int valid = p.Cout - (m0 + Rt);
        float o[4][4];                                       // [pixel][channel lane]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float m0_ = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
            const float ps = m1 + m2, qs = m1 - m2, rr = m3 + m4, tt = m3 - m4;
            o[0][e] = m0_ + ps + rr;
            o[1][e] = __builtin_fmaf(2.f, tt, qs);
            o[2][e] = __builtin_fmaf(4.f, rr, ps);
            o[3][e] = __builtin_fmaf(8.f, tt, qs) + m5;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float4 v = apply_epilogue4(p.ep, bias, scale, shift, rs[rq][b], valid, make_float4(o[b][0], o[b][1], o[b][2], o[b][3]));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                   yrsrc, off[rq][b], 0, 0);
        }
    }
}

// OIHW 3x3 -> uq[f][q = r*cqg + cin/4][co][4] with the 6 F(4,3) frequencies of each filter row
__global__ void __launch_bounds__(256) pack_filter_w1d4_kernel(const float *w, float *out, unsigned total, int Cout,
                                                               int Cin, int cqg, int Qtot, int Qpad, FastDiv divCo,
                                                               FastDiv divCqg) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {   // i = q*Cout + co
        unsigned q, co;
        divCo.divmod(i, q, co);
        float uu[6][4];
#pragma unroll
        for (int f = 0; f < 6; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) uu[f][e] = 0.f;
        if ((int)q < Qtot) {
            unsigned r, cq;
            divCqg.divmod(q, r, cq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (int)cq * 4 + e;
                if (c >= Cin) continue;
                const float *g = w + (((size_t)co * Cin + c) * 3 + r) * 3;
                const float g0 = g[0], g1 = g[1], g2 = g[2];
                uu[0][e] = g0 * 0.25f;
                uu[1][e] = -(g0 + g1 + g2) * (1.f / 6.f);
                uu[2][e] = (-g0 + g1 - g2) * (1.f / 6.f);
                uu[3][e] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
                uu[4][e] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
                uu[5][e] = g2;
            }
        }
        const size_t plane = (size_t)Qpad * Cout;
#pragma unroll
        for (int f = 0; f < 6; ++f)
            reinterpret_cast<float4 *>(out)[f * plane + i] = make_float4(uu[f][0], uu[f][1], uu[f][2], uu[f][3]);
    }
}
