// Included by conv_igemm.hip inside its anonymous namespace right after conv_q4_kernel.h (shares
// ConvArgs, TileCoord, store_tile_q4).  See conv_pc_kernel.h for the producer / consumer idea.
// =====================================================================================================
// Persistent producer / consumer implicit-GEMM convolution on channel-quad tensors (the arithmetic of
// conv_q4_kernel: Q4 activations, k-quad-major filters, K = (kh, kw, cin), groups, strides, dilations).
//
// Measured on MI355X (tools/ubench/pc_interference.hip): beside an MFMA-only wave, a partner wave's
// ds_write / ds_read / integer VALU cost nothing, its fp32 VALU costs the MFMA stream ~4.4 cycles per
// wave instruction (fp32 MFMA and fp32 VALU share the SIMD's FMA lanes), and a CU pulls only ~50 B/ns
// from L2 -- 24 KB per 648 ns MFMA step already stretches the step by 15-20 %.  A 64x64 tile moves
// 16 FLOP per staged byte, right at that ridge; hence here
//   * 128x128 (or 64x256) tiles per workgroup: 32 (25.6) FLOP per staged byte;
//   * both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), issued by the four
//     PRODUCER waves three chunks ahead into a ring of three stages: no staging registers, no ds_write,
//     no fp32 VALU in the K loop -- the gather's address arithmetic is integer and scalar (a producer wave
//     owns ONE k-quad of every chunk, so its filter tap and channel quad are wave-uniform);
//   * the four CONSUMER waves (one per SIMD, 64x64 outputs = four 32x32 accumulators each) read the
//     fragments of chunk g+1 while the 32 MFMAs of chunk g run, and apply the fused tail at the end of
//     a tile (store_tile_q4 of conv_q4_kernel.h);
//   * it is persistent: the producers run ahead across tile boundaries, so short-K problems (the
//     Winograd-domain GEMMs: K = Cin) do not pay a pipeline fill per tile.
// Step g (one s_barrier each): producers issue chunk g+3 into stage g%3 and wait until chunk g+2 has
// landed (counted vmcnt: this step's own DMAs stay in flight); consumers read chunk g+1 from stage
// (g+1)%3 into one fragment set and multiply chunk g from the other.
template <int BM_, int BN_>
struct PcCfg {
    static constexpr int BM = BM_, BN = BN_, BK = 16, KG = 4, STAGES = 3, THREADS = 512;
    static constexpr int WM = BM / 64, WN = BN / 64;           // consumer grid: 64x64 outputs per wave
    static_assert(WM * WN == 4, "four consumer waves");
    static constexpr int RB = BM / 64, CB = BN / 64;           // 64-row / 64-column DMA pieces per k-quad
    static constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    static constexpr int PRM = 3 * BM;
    static constexpr int LDS_BYTES = (STAGES * STAGE + 3 * PRM) * 4;
    static constexpr int DMA_PER_STEP = RB + CB;               // per producer wave
};

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4; expcnt / lgkmcnt left at max)
template <int N>
__device__ __forceinline__ void pc_wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
    asm volatile("" ::: "memory");
}
#define PC_WAIT_VMCNT(n) pc_wait_vmcnt<(n)>()

// one LDS-DMA piece: 64 lanes x 16 (4) bytes from (rsrc, per-lane voff + scalar soff) to lds[0 .. 64*16 (4))
__device__ __forceinline__ void pc_dma16(__amdgpu_buffer_rsrc_t rsrc, float *lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) float *)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void pc_dma4(__amdgpu_buffer_rsrc_t rsrc, float *lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) float *)lds, 4, voff, soff, 0, 0);
}

template <class C>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_pc_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *prm_base = smem + C::STAGES * C::STAGE;            // [3 (tile % 3)][3][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = (int)gridDim.x;
    const int T = p.tiles * p.groups;
    const int my_tiles = (T - (int)blockIdx.x + grid - 1) / grid;
    const int nchunks = p.Qpad / C::KG;                       // even: Qpad is a multiple of 8 k-quads
    const int G = my_tiles * nchunks;

    if (wave >= 4) {
        // ===================================== producers =====================================
        const int kq = wave - 4;                              // this wave's k-quad of every chunk
        constexpr int OOB = (int)0x80000000;
        const __amdgpu_buffer_rsrc_t xrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
        int it = 0, kc = 0, stage = 0;                        // (tile, chunk) and ring slot of the NEXT issue
        int grp = 0, m0 = 0;                                  // of tile `it`
        int arow[C::RB];                                      // per-lane filter row byte offset (or OOB)
        int hbase[C::CB], wbase[C::CB], cbase[C::CB];         // per-lane pixel of each 64-column piece
        auto set_tile = [&](int it_) {
            const int t = (int)blockIdx.x + it_ * grid;
            const bool live = t < T;                          // past the end: every lane out of range
            const unsigned gt = (unsigned)(live ? t : 0);
            grp = (int)(gt / (unsigned)p.tiles);
            const unsigned tt = gt - (unsigned)grp * (unsigned)p.tiles;
            const unsigned nt = p.divMt.div(tt);
            m0 = (int)(tt - nt * (unsigned)p.mtiles) * C::BM;
            const int col0 = (int)nt * C::BN;
#pragma unroll
            for (int rb = 0; rb < C::RB; ++rb) {
                const int row = m0 + rb * 64 + lane;
                arow[rb] = (live && row < p.cout_g) ? (row << 4) : OOB;
            }
#pragma unroll
            for (int cb = 0; cb < C::CB; ++cb) {
                const int j = col0 + cb * 64 + lane;
                hbase[cb] = -(1 << 20); wbase[cb] = 0; cbase[cb] = 0;
                if (live && j < p.cols) {
                    unsigned n, pix, ho, wo;
                    p.divHoWo.divmod((unsigned)j, n, pix);
                    p.divWo.divmod(pix, ho, wo);
                    hbase[cb] = (int)ho * p.sh - p.pt;
                    wbase[cb] = (int)wo * p.sw - p.pl;
                    cbase[cb] = ((int)n * p.Cq + grp * p.cqg) * p.HW + hbase[cb] * p.W + wbase[cb];   // in quads
                }
            }
            if (live && kq < 3) {                             // this tile's per-row parameters, one array per wave
                const float *src = kq == 0 ? p.ep.bias : kq == 1 ? p.ep.scale : p.ep.shift;
                if (src) {
                    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(src), 0, p.Cout * 4, 0x00020000);
#pragma unroll
                    for (int rb = 0; rb < C::RB; ++rb) {
                        const int row = m0 + rb * 64 + lane;
                        pc_dma4(prsrc, prm_base + (it_ % 3) * C::PRM + kq * C::BM + rb * 64,
                                  row < p.cout_g ? (grp * p.cout_g + row) << 2 : OOB, 0);
                    }
                }
            }
        };
        // one step's DMAs of this wave: k-quad q = kc*4 + kq of tile `it` into ring slot `stage`
        auto issue = [&]() {
            const int q = kc * C::KG + kq;
            const bool qok = q < p.Qtot;                      // K padding: zero filters, any finite input
            const unsigned tap = p.divCpt.div((unsigned)(qok ? q : 0));
            const int cq = (qok ? q : 0) - (int)tap * p.cqg;
            unsigned a, b;
            p.divKw.divmod(tap, a, b);
            const int dy = (int)a * p.dh, dx = (int)b * p.dw;
            float *sbase = smem + stage * C::STAGE;
            const int asoff = ((grp * p.Qpad + q) * p.cout_g) << 4;            // scalar
#pragma unroll
            for (int rb = 0; rb < C::RB; ++rb)
                pc_dma16(wrsrc, sbase + (kq * C::BM + rb * 64) * 4, arow[rb], asoff);
            const int bsoff = (cq * p.HW) << 4;                                // scalar: channel-quad plane
#pragma unroll
            for (int cb = 0; cb < C::CB; ++cb) {
                const bool ok = qok && (unsigned)(hbase[cb] + dy) < (unsigned)p.H && (unsigned)(wbase[cb] + dx) < (unsigned)p.W;
                const int voff = ok ? (int)((unsigned)(cbase[cb] + dy * p.W + dx) << 4) : OOB;
                pc_dma16(xrsrc, sbase + C::A_ELEMS + (kq * C::BN + cb * 64) * 4, voff, bsoff);
            }
            stage = stage == C::STAGES - 1 ? 0 : stage + 1;
            if (++kc == nchunks) {
                kc = 0;
                set_tile(++it);
            }
        };
        set_tile(0);
        issue();                                              // chunk 0 -> stage 0
        issue();                                              // chunk 1 -> stage 1
        issue();                                              // chunk 2 -> stage 2
        PC_WAIT_VMCNT(2 * C::DMA_PER_STEP);                   // chunk 0 (and tile 0's parameters) landed
        asm volatile("s_barrier" ::: "memory");
        PC_WAIT_VMCNT(C::DMA_PER_STEP);                       // chunk 1 landed
        asm volatile("s_barrier" ::: "memory");               // the consumers hold chunk 0 in registers
        for (int g = 0; g < G; ++g) {
            issue();                                          // chunk g+3 -> stage g%3
            PC_WAIT_VMCNT(C::DMA_PER_STEP);                   // chunk g+2 landed; this step's stay in flight
            asm volatile("s_barrier" ::: "memory");
        }
        PC_WAIT_VMCNT(0);
        return;
    }

    // ======================================= consumers =======================================
    __builtin_amdgcn_s_setprio(1);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (lhi * C::BM + wm * 64 + l31) * 4;      // k-quad lhi (+2u), first 32-row block of the wave
    const int b_off = C::A_ELEMS + (lhi * C::BN + wn * 64 + l31) * 4;
    f32x16 acc[2][2];
    float4 fa0[2][2], fb0[2][2], fa1[2][2], fb1[2][2];        // [k-quad pair u][32-block]
    auto read_frags = [&](int stage, float4 (&af)[2][2], float4 (&bf)[2][2]) {
        const float *base = smem + stage * C::STAGE;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[u][i] = *reinterpret_cast<const float4 *>(base + a_off + (2 * u * C::BM + i * 32) * 4);
                bf[u][i] = *reinterpret_cast<const float4 *>(base + b_off + (2 * u * C::BN + i * 32) * 4);
            }
    };
    auto mma = [&](const float4 (&af)[2][2], const float4 (&bf)[2][2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float av = s4 == 0 ? af[u][a].x : s4 == 1 ? af[u][a].y : s4 == 2 ? af[u][a].z : af[u][a].w;
                        const float bv = s4 == 0 ? bf[u][b].x : s4 == 1 ? bf[u][b].y : s4 == 2 ? bf[u][b].z : bf[u][b].w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
    };
    int stage = 1;                                            // ring slot of the chunk read NEXT
    asm volatile("s_barrier" ::: "memory");                   // chunk 0 is in LDS
    read_frags(0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int it = 0; it < my_tiles; ++it) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int kc = 0; kc < nchunks; kc += 2) {
            read_frags(stage, fa1, fb1);                      // chunk g+1 while chunk g multiplies
            stage = stage == C::STAGES - 1 ? 0 : stage + 1;
            mma(fa0, fb0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            read_frags(stage, fa0, fb0);                      // chunk g+2 (the next tile's first, at the end)
            stage = stage == C::STAGES - 1 ? 0 : stage + 1;
            mma(fa1, fb1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // ---- fused tail + store (conv_q4_kernel.h) ----
        const unsigned gt = (unsigned)((int)blockIdx.x + it * grid);
        TileCoord tc;
        tc.g = gt / (unsigned)p.tiles;
        tc.local = gt;
        const unsigned tt = gt - tc.g * (unsigned)p.tiles;
        const unsigned nt = p.divMt.div(tt);
        tc.m0 = (int)(tt - nt * (unsigned)p.mtiles) * C::BM;
        tc.col0 = (int)nt * C::BN;
        store_tile_q4<C::BM, C::BN, 2, 2, 64, 64>(p, tc, acc, wm, wn, lane, prm_base + (it % 3) * C::PRM);
    }
}
