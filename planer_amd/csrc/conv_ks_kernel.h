// Channel-quad convolution for SMALL problems -- included by conv_igemm.hip inside its anonymous namespace.
//
// A batch-1 detection net is a chain of convs on 13x13 .. 52x52 maps: 169 .. 2704 output pixels.  With 64x64 tiles
// that is a handful of workgroups, so the launch plans split K across workgroups and add a reduce kernel (two
// dependent launches, partial tiles through HBM): 10-11 us for a 0.18 GFLOP 1x1 layer.  Here the K split happens
// INSIDE a workgroup: a 1024-thread workgroup owns a 32-channel x 32-pixel tile, each of its 16 waves takes a
// sixteenth of K and reads both operands straight from global memory as float4s of 4 consecutive k (nothing is shared
// between waves, so nothing is staged in LDS), and the 16 partial tiles meet in LDS: wave r sums accumulator register
// r of all sixteen in fixed order, then four waves apply the fused tail and store channel quads.  One launch, no
// slabs, bit-reproducible.  The autotuner times it like any other tile configuration ("k32x32x8").
struct KsCfg {
    static constexpr int BM = 32, BN = 32, BK = 8, WAVES = 16;
    static constexpr int LDS_BYTES = (WAVES * 16 * 64 + 16 * 64) * 4;       // partial tiles + the summed tile
};

__global__ void __launch_bounds__(1024) conv_ks_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *part = smem;                                 // [wave][reg][lane]
    float *total = smem + KsCfg::WAVES * 16 * 64;       // [reg][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const TileCoord tc = tile_coord<KsCfg::BM, KsCfg::BN>(p);
    const int m0 = tc.m0, col0 = tc.col0;
    const unsigned g = tc.g;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    // this lane's pixel (B operand) and filter row (A operand)
    const int j = col0 + l31;
    int hbase = -(1 << 20), wbase = 0, cbase = 0;
    if (j < p.cols) {
        unsigned n, pix, ho, wo;
        p.divHoWo.divmod((unsigned)j, n, pix);
        p.divWo.divmod(pix, ho, wo);
        hbase = (int)ho * p.sh - p.pt;
        wbase = (int)wo * p.sw - p.pl;
        cbase = ((int)n * p.Cq + (int)g * p.cqg) * p.HW;                    // in quads
    }
    const bool rok = m0 + l31 < p.cout_g;
    // wave w takes k-quads [q0, q1): an even count, so the half-waves pair up (lanes 0-31: quad q, 32-63: quad q+1)
    const int per = ((p.Qpad + 2 * KsCfg::WAVES - 1) / (2 * KsCfg::WAVES)) * 2;
    const int q0 = wave * per, q1 = min(p.Qpad, q0 + per);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int q = q0; q < q1; q += 8) {                                       // four quad pairs in flight per lane
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int qq = q + 2 * u + lhi;
            const bool live = qq < q1;
            a[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  wrsrc, (live && rok) ? ((((int)g * p.Qpad + qq) * p.cout_g + m0 + l31) << 4) : OOB, 0, 0));
            const unsigned tap = p.divCpt.div((unsigned)qq);                 // q = tap * cqg + cq
            const int cq = qq - (int)tap * p.cqg;
            unsigned ta, tb;
            p.divKw.divmod(tap, ta, tb);
            const int hi = hbase + (int)ta * p.dh, wi = wbase + (int)tb * p.dw;
            const bool ok = live && qq < p.Qtot && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            b[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  xrsrc, ok ? (int)((unsigned)(cbase + cq * p.HW + hi * p.W + wi) << 4) : OOB, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    {                                                                        // wave r sums register r, waves 0, 1, 2, ... in order
        float s = part[wave * 64 + lane];
#pragma unroll
        for (int w = 1; w < KsCfg::WAVES; ++w) s = __fadd_rn(s, part[(w * 16 + wave) * 64 + lane]);
        total[wave * 64 + lane] = s;
    }
    __syncthreads();
    if (wave < 4 && j < p.cols) {
        // C layout: column = lane & 31 (pixel), rows 8*(r>>2) + 4*(lane>>5) + (r&3): register quad `wave` = 4 consecutive channels
        const int row = 8 * wave + 4 * lhi;
        if (m0 + row < p.cout_g) {
            const float4 v = make_float4(total[(4 * wave + 0) * 64 + lane], total[(4 * wave + 1) * 64 + lane],
                                         total[(4 * wave + 2) * 64 + lane], total[(4 * wave + 3) * 64 + lane]);
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
}
