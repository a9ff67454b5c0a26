// Channel-quad convolution for SMALL problems -- included by conv_direct.hip inside its anonymous namespace.
//
// A batch-1 detection net is a chain of convs on 13x13 .. 52x52 maps: 169 .. 2704 output pixels.  With 64x64 tiles
// that is a handful of workgroups, so the launch plans split K across workgroups and add a reduce kernel (two
// dependent launches, partial tiles through HBM): 10-11 us for a 0.18 GFLOP 1x1 layer.  Here the K split happens
// INSIDE a workgroup: its waves each take a slice of K and read both operands straight from global memory as float4s
// of 4 consecutive k (nothing is shared between waves, so nothing is staged in LDS); the partial tiles meet in LDS,
// every wave sums its share of the accumulator registers over all waves in fixed order, then the fused tail is applied
// and channel quads are stored.  One launch, no slabs, bit-reproducible.  The autotuner times the variants like any
// other tile configuration: k32x32x8 (16 waves), k64x32x8 / k32x64x8 (8 waves), k64x64x8 (4 waves).
template <int TM_, int TN_>
struct KsCfg {
    static constexpr int TM = TM_, TN = TN_, BM = 32 * TM, BN = 32 * TN, BK = 8;
    static constexpr int WAVES = 16 / (TM * TN), REGS = 16 * TM * TN;
    static constexpr int RPW = REGS / WAVES;                                // accumulator registers a wave sums
    static constexpr int LDS_BYTES = (WAVES * REGS * 64 + REGS * 64) * 4;   // partial tiles + the summed tile
};

template <class KC>
__global__ void __launch_bounds__(KC::WAVES * 64) conv_ks_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *part = smem;                                 // [wave][reg][lane]
    float *total = smem + KC::WAVES * KC::REGS * 64;    // [reg][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const TileCoord tc = tile_coord<KC::BM, KC::BN>(p);
    const int m0 = tc.m0, col0 = tc.col0;
    const unsigned g = tc.g;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    // this lane's pixels (B operand), one per 32-column block
    int hbase[KC::TN], wbase[KC::TN], cbase[KC::TN];
#pragma unroll
    for (int b = 0; b < KC::TN; ++b) {
        const int j = col0 + 32 * b + l31;
        hbase[b] = -(1 << 20); wbase[b] = 0; cbase[b] = 0;
        if (j < p.cols) {
            unsigned n, pix, ho, wo;
            p.divHoWo.divmod((unsigned)j, n, pix);
            p.divWo.divmod(pix, ho, wo);
            hbase[b] = (int)ho * p.sh - p.pt;
            wbase[b] = (int)wo * p.sw - p.pl;
            cbase[b] = ((int)n * p.Cq + (int)g * p.cqg) * p.HW;              // in quads
        }
    }
    // wave w takes k-quads [q0, q1): an even count, so the half-waves pair up (lanes 0-31: quad q, 32-63: quad q+1)
    const int per = ((p.Qpad + 2 * KC::WAVES - 1) / (2 * KC::WAVES)) * 2;
    const int q0 = wave * per, q1 = min(p.Qpad, q0 + per);
    f32x16 acc[KC::TM][KC::TN];
#pragma unroll
    for (int t = 0; t < KC::TM; ++t)
#pragma unroll
        for (int b = 0; b < KC::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.f;
    constexpr int UN = KC::TM * KC::TN == 4 ? 2 : 4;                        // quad pairs in flight per lane
    for (int q = q0; q < q1; q += 2 * UN) {
        float4 a[UN][KC::TM], bv[UN][KC::TN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int qq = q + 2 * u + lhi;
            const bool live = qq < q1;
#pragma unroll
            for (int t = 0; t < KC::TM; ++t)
                a[u][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrsrc, (live && m0 + 32 * t + l31 < p.cout_g)
                                                                    ? ((((int)g * p.Qpad + qq) * p.cout_g + m0 + 32 * t + l31) << 4) : OOB, 0, 0));
            const unsigned tap = p.divCpt.div((unsigned)qq);                 // q = tap * cqg + cq
            const int cq = qq - (int)tap * p.cqg;
            unsigned ta, tb;
            p.divKw.divmod(tap, ta, tb);
#pragma unroll
            for (int b = 0; b < KC::TN; ++b) {
                const int hi = hbase[b] + (int)ta * p.dh, wi = wbase[b] + (int)tb * p.dw;
                const bool ok = live && qq < p.Qtot && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                bv[u][b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          xrsrc, ok ? (int)((unsigned)(cbase[b] + cq * p.HW + hi * p.W + wi) << 4) : OOB, 0, 0));
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < KC::TM; ++t)
#pragma unroll
                for (int b = 0; b < KC::TN; ++b) {
                    acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t].x, bv[u][b].x, acc[t][b], 0, 0, 0);
                    acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t].y, bv[u][b].y, acc[t][b], 0, 0, 0);
                    acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t].z, bv[u][b].z, acc[t][b], 0, 0, 0);
                    acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t].w, bv[u][b].w, acc[t][b], 0, 0, 0);
                }
    }
#pragma unroll
    for (int t = 0; t < KC::TM; ++t)
#pragma unroll
        for (int b = 0; b < KC::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(wave * KC::REGS + (t * KC::TN + b) * 16 + r) * 64 + lane] = acc[t][b][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KC::RPW; ++i) {                                      // this wave's registers, waves 0, 1, 2, ... in order
        const int reg = wave * KC::RPW + i;
        float s = part[reg * 64 + lane];
#pragma unroll
        for (int w = 1; w < KC::WAVES; ++w) s = __fadd_rn(s, part[(w * KC::REGS + reg) * 64 + lane]);
        total[reg * 64 + lane] = s;
    }
    __syncthreads();
    // C layout of a block: column = lane & 31 (pixel), rows 8*(r>>2) + 4*(lane>>5) + (r&3).  Register quad k of the tile
    // (k = (t*TN + b)*4 + rq) = 4 consecutive channels of one pixel; the quads are dealt round robin to the waves.
    for (int k = wave; k < 4 * KC::TM * KC::TN; k += KC::WAVES) {
        const int blk = k >> 2, rq = k & 3, t = blk / KC::TN, b = blk - t * KC::TN;
        const int j = col0 + 32 * b + l31, row = 32 * t + 8 * rq + 4 * lhi;
        if (j >= p.cols || m0 + row >= p.cout_g) continue;
        const float4 v = make_float4(total[(4 * k + 0) * 64 + lane], total[(4 * k + 1) * 64 + lane],
                                     total[(4 * k + 2) * 64 + lane], total[(4 * k + 3) * 64 + lane]);
        const int c0 = (int)g * p.cout_g + m0 + row, cend = (int)g * p.cout_g + p.cout_g;
        float bs[4], sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_chan_params(p.ep, min(c0 + e, cend - 1), bs[e], sc[e], sh[e]);
        unsigned n, pix;
        p.divHoWo.divmod((unsigned)j, n, pix);
        const size_t idx4 = ((size_t)n * p.Coq + (size_t)(c0 >> 2)) * p.HoWo + pix;
        float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.ep.res) rs = reinterpret_cast<const float4 *>(p.ep.res)[idx4];
        reinterpret_cast<float4 *>(p.y)[idx4] =
            apply_epilogue4(p.ep, make_float4(bs[0], bs[1], bs[2], bs[3]), make_float4(sc[0], sc[1], sc[2], sc[3]),
                            make_float4(sh[0], sh[1], sh[2], sh[3]), rs, cend - c0, v);
    }
}
