// A 1x1 convolution with its fused tail (layer.Conv2d layer.py:22-26 + BatchNorm :125-127 + ReLU / LeakyReLU :44-51) whose
// ONLY reader is the input transform of a staged Winograd 3x3 convolution (a Darknet block: 1x1 C -> C/2, then 3x3 C/2 -> C),
// in one kernel: the workgroup computes its 1x1 outputs on the 6 x 10 pixel REGION two 6x6 Winograd patches (F(4x4,3x3)) or eight
// 4x4 patches (F(2x2,3x3)) cover -- halo included, so 1.9x the convolution's multiplies, on a layer that is 1/9 of its neighbour --
// parks the activated values in LDS (pixels outside the map as zeros: they are the 3x3 conv's padding) and applies B^T d B from
// there: V leaves the kernel, y never exists and the input-transform launch is gone.  Made for batch-1 detection nets, where
// every kernel is a ~5 us latency chain: -1 launch per 3x3 conv.  Included by conv_winograd.hip.
//
// GEMM part = conv_ks_kernel<KsCfg<1, 2>> (conv_ks_kernel.h): 32 output channels x 64 columns on 8 waves, each wave a slice
// of K with both operands as float4 straight from global memory, partial tiles summed through LDS in wave order
// (bit-reproducible).  Columns are region pixels (row-major 6 x 10, 60 of 64 used) instead of a run of the flattened map.
// Transform part = the arithmetic of wino4_input_rows_q4_kernel / wino_input_q4_kernel, expression for expression.

struct C1WArgs {
    const float *x, *w;          // Q4 input [N][Cq][H][W][4]; filter wq[q][Cout][4] (prepare_q4_weights, group 1, 1x1)
    float4 *V;                   // [36 or 16][Cout/4][T][4]
    int N, Cq, H, W, Cout;
    int Qtot, Qpad;              // k-quads (= Cq), padded to 8
    int th, tw, T;               // Winograd tiles per image and in total
    int rty, rtx;                // tiles per region along y / x (F4: 1 x 2, F2: 2 x 4)
    int rh, rw;                  // regions per image along y / x
    int mtiles;                  // ceil(Cout / 32)
    unsigned x_bytes, w_bytes;
    FastDiv divMt, divRw, divRh;
    Epilogue ep;
};

constexpr int C1W_RW = 10, C1W_RH = 6, C1W_WAVES = 8, C1W_REGS = 32;
constexpr int C1W_LDS_FLOATS = C1W_WAVES * C1W_REGS * 64 + 8 * 65 * 4;
constexpr int C1W_PS = 65;        // cells of 16 bytes per channel quad of the LDS plane (64 pixels + 1: the quads of a lane group on different banks)

template <int WINO>      // 4: F(4x4,3x3) patches of 6x6, 36 frequencies; 2: F(2x2,3x3) patches of 4x4, 16 frequencies
__global__ void __launch_bounds__(C1W_WAVES * 64) conv1x1_wino_in_kernel(const C1WArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // C1W_LDS_FLOATS
    float *part = smem;                                    // [wave][reg][lane]
    float4 *plane = reinterpret_cast<float4 *>(smem + C1W_WAVES * C1W_REGS * 64);      // [8 quads][C1W_PS]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned reg, mt, n, ry0, rx0, r2;
    p.divMt.divmod(blockIdx.x, reg, mt);
    p.divRw.divmod(reg, r2, rx0);
    p.divRh.divmod(r2, n, ry0);
    constexpr int TS = WINO == 4 ? 4 : 2;                  // output pixels per tile side
    const int h0 = (int)ry0 * p.rty * TS - 1, w0 = (int)rx0 * p.rtx * TS - 1;   // first region pixel (the patches' halo)
    const int m0 = (int)mt * 32;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    const int HW = p.H * p.W;
    // this lane's two region pixels (B operand)
    int pbase[2];
    bool pin[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int jl = 32 * b + l31, ry = (jl * 205) >> 11, rx = jl - ry * C1W_RW;       // jl / 10 for jl < 64
        const int h = h0 + ry, w = w0 + rx;
        pin[b] = jl < C1W_RH * C1W_RW && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        pbase[b] = ((int)n * p.Cq) * HW + h * p.W + w;                                   // in quads
    }
    const int per = ((p.Qpad + 2 * C1W_WAVES - 1) / (2 * C1W_WAVES)) * 2;
    const int q0 = wave * per, q1 = min(p.Qpad, q0 + per);
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // every request of a round (8 k-quads per wave) is in flight at once
    auto k_rounds = [&](auto untag) {
        constexpr int UN = decltype(untag)::value;
        for (int q = q0; q < q1; q += 2 * UN) {
            float4 a[UN], bv[UN][2];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int qq = q + 2 * u + lhi;
                const bool live = qq < q1;
                a[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      wrsrc, (live && m0 + l31 < p.Cout) ? ((qq * p.Cout + m0 + l31) << 4) : OOB, 0, 0));
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    bv[u][b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                              xrsrc, (live && qq < p.Qtot && pin[b]) ? (int)((unsigned)(pbase[b] + qq * HW) << 4) : OOB, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, bv[u][b].x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, bv[u][b].y, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, bv[u][b].z, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, bv[u][b].w, acc[b], 0, 0, 0);
                }
        }
    };
    // (eight quad pairs in flight -- one round trip for K = 512 -- costs 155 registers: one workgroup per CU instead of two,
    //  YOLO-v3 at batch 1 1.270 against 1.240 ms)
    k_rounds(std::integral_constant<int, 4>{});
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wave * C1W_REGS + b * 16 + r) * 64 + lane] = acc[b][r];
    __syncthreads();
    // wave k sums registers 4k .. 4k + 3 of the tile over the eight partial tiles, in wave order -- exactly register quad
    // k = b * 4 + rq of the C layout: 4 consecutive channels (row 8 rq + 4 lhi) of pixel 32 b + l31 -- and applies the fused tail
    {
        const int k = wave, b = k >> 2, rq = k & 3;
        const int row = 8 * rq + 4 * lhi, c0 = m0 + row;
        float sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = 4 * k + i;
            float s = part[rg * 64 + lane];
#pragma unroll
            for (int w = 1; w < C1W_WAVES; ++w) s = __fadd_rn(s, part[(w * C1W_REGS + rg) * 64 + lane]);
            sv[i] = s;
        }
        const float4 v = make_float4(sv[0], sv[1], sv[2], sv[3]);
        float bs[4], sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_chan_params(p.ep, min(c0 + e, p.Cout - 1), bs[e], sc[e], sh[e]);
        float4 o = apply_epilogue4(p.ep, make_float4(bs[0], bs[1], bs[2], bs[3]), make_float4(sc[0], sc[1], sc[2], sc[3]),
                                   make_float4(sh[0], sh[1], sh[2], sh[3]), make_float4(0.f, 0.f, 0.f, 0.f), p.Cout - c0, v);
        if (!pin[b] || c0 >= p.Cout) o = make_float4(0.f, 0.f, 0.f, 0.f);     // outside the map: the 3x3 conv's zero padding
        plane[(row >> 2) * C1W_PS + 32 * b + l31] = o;
    }
    __syncthreads();
    const int Coq = p.Cout >> 2;                           // Cout % 4 == 0 (checked by the host)
    const size_t vplane = (size_t)Coq * p.T;
    if constexpr (WINO == 4) {
        // item = (row A of the transformed 6x6 tile: wave-uniform, quad 0..7, tile 0..1): waves 0..5, 16 lanes each
        if (wave < 6 && lane < 16) {
            const int cql = lane & 7, tl = lane >> 3;
            const int ty = (int)ry0, tx = (int)rx0 * 2 + tl;
            const int cq = (m0 >> 2) + cql;
            if (tx < p.tw && cq < Coq) {
                const float4 *cell = plane + cql * C1W_PS + tl * 4;      // patch origin = region pixel (0, 4 tl)
                wc_v2 mlo[6], mhi[6];
                auto column = [&](auto atag, int j) {
                    constexpr int A = decltype(atag)::value;
                    wc_v2 dlo[6], dhi[6];
#pragma unroll
                    for (int kk = 0; kk < 6; ++kk) {
                        const float4 d = cell[kk * C1W_RW + j];
                        dlo[kk] = wc_lo(d);
                        dhi[kk] = wc_hi(d);
                    }
                    mlo[j] = wc_bt_row2<A>(dlo);
                    mhi[j] = wc_bt_row2<A>(dhi);
                };
                auto run = [&](auto atag) {
                    constexpr int A = decltype(atag)::value;
#pragma unroll
                    for (int j = 0; j < 6; ++j) column(atag, j);
                    wc_v2 olo[6], ohi[6];
                    wc_bt2(mlo, olo);
                    wc_bt2(mhi, ohi);
                    const size_t vbase = (size_t)cq * p.T + ((size_t)n * p.th + ty) * p.tw + tx;
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        p.V[(size_t)(A * 6 + j) * vplane + vbase] = make_float4(olo[j].x, olo[j].y, ohi[j].x, ohi[j].y);
                };
                switch (wave) {
                case 0: run(std::integral_constant<int, 0>{}); break;
                case 1: run(std::integral_constant<int, 1>{}); break;
                case 2: run(std::integral_constant<int, 2>{}); break;
                case 3: run(std::integral_constant<int, 3>{}); break;
                case 4: run(std::integral_constant<int, 4>{}); break;
                default: run(std::integral_constant<int, 5>{}); break;
                }
            }
        }
    } else {
        // F(2x2,3x3): item = (quad 0..7, tile 0..7 = 2 x 4), one wave; the arithmetic of wino_input_q4_kernel
        if (wave == 0) {
            const int cql = lane & 7, tl = lane >> 3, tyl = tl >> 2, txl = tl & 3;
            const int ty = (int)ry0 * 2 + tyl, tx = (int)rx0 * 4 + txl;
            const int cq = (m0 >> 2) + cql;
            if (ty < p.th && tx < p.tw && cq < Coq) {
                const float4 *cell = plane + cql * C1W_PS + (tyl * 2) * C1W_RW + txl * 2;
                float4 d[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) d[a][b] = cell[a * C1W_RW + b];
                float4 m[4][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    m[0][b] = f4sub(d[0][b], d[2][b]);
                    m[1][b] = f4sum(d[1][b], d[2][b]);
                    m[2][b] = f4sub(d[2][b], d[1][b]);
                    m[3][b] = f4sub(d[1][b], d[3][b]);
                }
                float4 *vp = p.V + (size_t)cq * p.T + ((size_t)n * p.th + ty) * p.tw + tx;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    vp[(size_t)(a * 4 + 0) * vplane] = f4sub(m[a][0], m[a][2]);
                    vp[(size_t)(a * 4 + 1) * vplane] = f4sum(m[a][1], m[a][2]);
                    vp[(size_t)(a * 4 + 2) * vplane] = f4sub(m[a][2], m[a][1]);
                    vp[(size_t)(a * 4 + 3) * vplane] = f4sub(m[a][1], m[a][3]);
                }
            }
        }
    }
}
