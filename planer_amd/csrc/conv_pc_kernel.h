// Persistent producer / consumer kernels -- included by conv_igemm.hip inside its anonymous
// namespace after conv_w1d_kernel.h (shares ConvArgs, the Q4 epilogue helpers, f4a/f4m/f4s).
//
// Why.  fp32 MFMA is slow relative to everything around it (v_mfma_f32_32x32x2_f32 holds a SIMD's
// matrix pipe for 64 cycles and needs 2 operand registers), so ONE wave per SIMD that does nothing
// but MFMAs saturates the pipe -- if nothing else is in its instruction stream.  The 256-thread
// kernels make every wave do everything (global loads, the Winograd input transform, LDS writes,
// fragment reads, MFMAs, the epilogue) and rely on a second workgroup per CU to cover one
// workgroup's non-MFMA phases; measured, the matrix pipe is busy 40-60 % of the time in them.
// Here a workgroup is 8 waves on one CU, two per SIMD:
//   waves 0-3  CONSUMERS: ds_read fragments of chunk g+1 into one register set while the MFMAs of
//              chunk g run from the other; at the end of a tile the lane-local output transform and
//              the fused tail (same code as conv_w1d4_kernel).  No global loads in the K loop.
//   waves 4-7  PRODUCERS: global loads two chunks ahead of the LDS write, the input transform
//              B^T d on the way (waves 4-5: pixels) or plain filter planes (waves 6-7), ds_write.
// and it is PERSISTENT: a workgroup walks tiles t = blockIdx.x, +gridDim.x, ... and the producers
// run ahead across tile boundaries, so a tile's prologue (first loads, LDS fill) hides under the
// previous tile's MFMAs and only the consumers' epilogue interrupts the MFMA stream.
// One s_barrier per chunk joins all 8 waves; LDS holds two stages.  At step g the producers write
// chunk g+2 into stage g%2 (its previous content, chunk g, went to registers during step g-1) while
// the consumers read chunk g+1 from stage (g+1)%2.

struct W1d4PcCfg {
    static constexpr int BM = 64, BN = 64, KG = 2, F = 6, THREADS = 512;
    static constexpr int A_PLANE = KG * BM * 4, B_PLANE = KG * BN * 4;            // floats per frequency plane
    static constexpr int A_ELEMS = F * A_PLANE, B_ELEMS = F * B_PLANE, STAGE = A_ELEMS + B_ELEMS;
    static constexpr int PRM = 3 * BM;                                             // bias, scale, shift per row
    static constexpr int LDS_BYTES = (2 * STAGE + 2 * PRM) * 4;                    // 48 KB + parameters
};

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_w1d4_pc_kernel(const ConvArgs p) {
    using C = W1d4PcCfg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *prm_base = smem + 2 * C::STAGE;                    // [2 (tile parity)][3][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = (int)gridDim.x;
    const int my_tiles = (p.tiles - (int)blockIdx.x + grid - 1) / grid;
    const int nchunks = p.Qtot / C::KG;                       // even (host checks Cin % 16 == 0)
    const int G = my_tiles * nchunks;

    if (wave >= 4) {
        // ===================================== producers =====================================
        const int kq = wave & 1;                              // k-quad of the chunk this wave stages
        constexpr int OOB = (int)0x80000000;
        int it = 0, kc = 0;                                   // (tile, chunk) the NEXT load fetches
        if (wave >= 6) {
            // ---- A role: the six filter-frequency planes go global -> LDS directly (LDS-DMA: no
            //      registers, no ds_write); a wave's 64 lanes = the tile's 64 filter rows of one k-quad
            const __amdgpu_buffer_rsrc_t wrsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
            int arow = OOB;
            auto set_tile = [&](int it_) {
                const int t = (int)blockIdx.x + it_ * grid;
                const bool live = t < p.tiles;
                const int nt = (int)p.divMt.div((unsigned)(live ? t : 0));
                const int m0 = ((live ? t : 0) - nt * p.mtiles) * C::BM;
                arow = (live && m0 + lane < p.Cout) ? ((kq * p.Cout + m0 + lane) << 4) : OOB;
                if (live && kq == 0) {                        // this tile's per-row epilogue parameters
                    float b, sc, sh;
                    load_chan_params(p.ep, min(m0 + lane, p.Cout - 1), b, sc, sh);
                    float *prm = prm_base + (it_ & 1) * C::PRM;
                    prm[lane] = b; prm[C::BM + lane] = sc; prm[2 * C::BM + lane] = sh;
                }
            };
            auto dma_chunk = [&](int stage) {
                const int ksoff = (kc * C::KG * p.Cout) << 4;
                __attribute__((address_space(3))) float *dst =
                    (__attribute__((address_space(3))) float *)(smem + stage * C::STAGE + kq * C::BM * 4);
#pragma unroll
                for (int i = 0; i < C::F; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, dst + i * C::A_PLANE, 16, arow, ksoff + ((i * p.Qpad * p.Cout) << 4), 0, 0);
                if (++kc == nchunks) {
                    kc = 0;
                    set_tile(++it);
                }
            };
            set_tile(0);
            dma_chunk(0);                                     // chunk 0
            dma_chunk(1);                                     // chunk 1
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            for (int g = 0; g < G; g += 2) {
                dma_chunk(0);               // step g: chunk g+2 -> stage 0
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                dma_chunk(1);               // step g+1: chunk g+3 -> stage 1
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            return;
        }
        // ---- B role: six pixels of this lane's (tile, k-quad), the input transform B^T d on the way
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
        int ho = -(1 << 20), pixbase = 0, cokmask = 0;        // this lane's tile column
        auto set_tile = [&](int it_) {
            const int t = (int)blockIdx.x + it_ * grid;
            const bool live = t < p.tiles;                    // past the end: loads fall out of range, nobody reads them
            const int nt = (int)p.divMt.div((unsigned)(live ? t : 0));
            const int j = nt * C::BN + lane;
            ho = -(1 << 20); pixbase = 0; cokmask = 0;
            if (live && j < p.cols) {
                unsigned n, rem, h, tw;
                p.divHoWo.divmod((unsigned)j, n, rem);        // cols = N * H * Tw, Tw = ceil(W / 4)
                p.divWo.divmod(rem, h, tw);
                ho = (int)h;
                pixbase = ((int)n * p.Cq * p.H + ho) * p.W + 4 * (int)tw - 1;
#pragma unroll
                for (int k = 0; k < 6; ++k) cokmask |= ((unsigned)(4 * (int)tw - 1 + k) < (unsigned)p.W) << k;
            }
        };
        auto load_chunk = [&](float4 (&st)[C::F]) {
            const int q = kc * C::KG + kq;
            const int r = (int)p.divCpt.div((unsigned)q);
            const int cq = q - r * p.cqg;
            const bool rok = (unsigned)(ho + r - 1) < (unsigned)p.H;
            const int vrow = (int)((unsigned)(pixbase + (cq * p.H + r - 1) * p.W) << 4);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                st[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       xrsrc, rok && ((cokmask >> k) & 1) ? vrow + 16 * k : OOB, 0, 0));
            if (++kc == nchunks) {
                kc = 0;
                set_tile(++it);
            }
        };
        auto store_chunk = [&](int stage, const float4 (&st)[C::F]) {   // V = B^T d, points 0, +-1, +-2, inf
            const float4 d0 = st[0], d1 = st[1], d2 = st[2], d3 = st[3], d4 = st[4], d5 = st[5];
            const float4 a = f4fma(-4.f, d2, d4), b = f4fma(-4.f, d1, d3);
            const float4 c2 = f4m(d4, d2), e1 = f4m(d3, d1);
            float4 *bp = reinterpret_cast<float4 *>(smem + stage * C::STAGE + C::A_ELEMS) + kq * C::BN + lane;
            bp[0 * (C::B_PLANE / 4)] = f4fma(4.f, d0, f4fma(-5.f, d2, d4));
            bp[1 * (C::B_PLANE / 4)] = f4a(a, b);
            bp[2 * (C::B_PLANE / 4)] = f4m(a, b);
            bp[3 * (C::B_PLANE / 4)] = f4fma(2.f, e1, c2);
            bp[4 * (C::B_PLANE / 4)] = f4fma(-2.f, e1, c2);
            bp[5 * (C::B_PLANE / 4)] = f4fma(4.f, d1, f4fma(-5.f, d3, d5));
        };
        float4 st0[C::F], st1[C::F];
        set_tile(0);
        load_chunk(st0);                                      // chunk 0
        load_chunk(st1);                                      // chunk 1
        store_chunk(0, st0);
        load_chunk(st0);                                      // chunk 2
        store_chunk(1, st1);
        load_chunk(st1);                                      // chunk 3
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // chunks 0 and 1 are published
        asm volatile("s_barrier" ::: "memory");               // the consumers hold chunk 0 in registers
        for (int g = 0; g < G; g += 2) {
            store_chunk(0, st0);                              // step g: chunk g+2 -> stage 0
            load_chunk(st0);                                  //         chunk g+4 requested
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            store_chunk(1, st1);                              // step g+1: chunk g+3 -> stage 1
            load_chunk(st1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return;
    }

    // ======================================= consumers =======================================
    __builtin_amdgcn_s_setprio(1);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (lhi * C::BM + wm * 32 + l31) * 4;      // k-quad lhi, row of the wave's 32-row block
    const int b_off = C::A_ELEMS + (lhi * C::BN + wn * 32 + l31) * 4;
    f32x16 acc[C::F];
    float4 fa0[C::F], fb0[C::F], fa1[C::F], fb1[C::F];
    auto read_frags = [&](int stage, float4 (&af)[C::F], float4 (&bf)[C::F]) {
        const float *base = smem + stage * C::STAGE;
#pragma unroll
        for (int f = 0; f < C::F; ++f) {
            af[f] = *reinterpret_cast<const float4 *>(base + a_off + f * C::A_PLANE);
            bf[f] = *reinterpret_cast<const float4 *>(base + b_off + f * C::B_PLANE);
        }
    };
    auto mma = [&](const float4 (&af)[C::F], const float4 (&bf)[C::F]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int f = 0; f < C::F; ++f) {
                const float av = s4 == 0 ? af[f].x : s4 == 1 ? af[f].y : s4 == 2 ? af[f].z : af[f].w;
                const float bv = s4 == 0 ? bf[f].x : s4 == 1 ? bf[f].y : s4 == 2 ? bf[f].z : bf[f].w;
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[f], 0, 0, 0);
            }
    };
    float4 *y4 = reinterpret_cast<float4 *>(p.y);
    const float4 *res4 = reinterpret_cast<const float4 *>(p.ep.res);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

    asm volatile("s_barrier" ::: "memory");                    // chunks 0 and 1 are in LDS
    read_frags(0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage 0 may be rewritten
    for (int it = 0; it < my_tiles; ++it) {
#pragma unroll
        for (int f = 0; f < C::F; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
        for (int kc = 0; kc < nchunks; kc += 2) {
            read_frags(1, fa1, fb1);                          // chunk g+1 while chunk g multiplies
            mma(fa0, fb0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            read_frags(0, fa0, fb0);                          // chunk g+2 (the next tile's first, at the end)
            mma(fa1, fb1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // ---- epilogue: y = A^T m per lane (4 pixels of one output row), fused tail, b128 stores ----
        const int t = (int)blockIdx.x + it * grid;
        const int nt = (int)p.divMt.div((unsigned)t);
        const int m0 = (t - nt * p.mtiles) * C::BM, col0 = nt * C::BN;
        const int jc = min(col0 + wn * 32 + l31, p.cols - 1);
        const bool live = col0 + wn * 32 + l31 < p.cols;
        unsigned n, rem, h, tw;
        p.divHoWo.divmod((unsigned)jc, n, rem);
        p.divWo.divmod(rem, h, tw);
        const int wo = 4 * (int)tw;
        const float4 *prm4 = reinterpret_cast<const float4 *>(prm_base + (it & 1) * C::PRM);
        const unsigned obase = (n * (unsigned)p.Coq * (unsigned)p.H + h) * (unsigned)p.W + (unsigned)wo;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int Rt = wm * 32 + 8 * rq + 4 * lhi;
            if (!live || m0 + Rt >= p.Cout) continue;
            const unsigned idx = obase + (unsigned)((m0 + Rt) >> 2) * (unsigned)(p.H * p.W);
            float4 rs[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) rs[b] = (res4 && wo + b < p.W) ? res4[idx + b] : z;
            const float4 bias = prm4[Rt >> 2], scale = prm4[(C::BM + Rt) >> 2], shift = prm4[(2 * C::BM + Rt) >> 2];
            const int valid = p.Cout - (m0 + Rt);
            float o[4][4];                                    // [pixel][channel lane]
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
            for (int b = 0; b < 4; ++b)
                if (wo + b < p.W)
                    y4[idx + b] = apply_epilogue4(p.ep, bias, scale, shift, rs[b], valid,
                                                  make_float4(o[b][0], o[b][1], o[b][2], o[b][3]));
        }
    }
}
