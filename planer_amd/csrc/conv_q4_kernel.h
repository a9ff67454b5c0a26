// Channel-quad ("Q4") implicit-GEMM convolution -- included by conv_direct.hip
// inside its anonymous namespace (shares ConvArgs, tile_coord, Plan, tuning).
//
// Why a second activation layout.  Measured on MI355X (tools/ubench/mfma_mix.hip,
// DESIGN.md section 4): with NCHW activations the 4 k-values a thread stages per
// (pixel, k-quad) live in 4 different channel planes, i.e. 4 scalar-dword buffer
// loads; 8 of those per thread per K-chunk cost the fp32 MFMA pipe 13-17 % of its
// throughput at every occupancy, while the same bytes fetched as b128 loads cost
// nothing.  So inside a compiled plan activations are kept as
//     Q4:  x[n][c/4][h][w][c%4]        (C padded to a multiple of 4 with zeros)
// and filters are prepared once as
//     wq[g][q][co][4],  q = tap*(Cin_g/4) + cin/4     (k-quad major, zero padded)
// Then one b128 load brings a thread its (pixel, 4 channels) and another its
// (filter row, 4 k), both perfectly coalesced (consecutive lanes = consecutive
// pixels / rows = consecutive 16 B); LDS tiles are [k-quad][row][4] with NO
// padding: ds_write_b128 / ds_read_b128 are conflict-free by construction, and
// the 32x32 MFMA accumulator (lane holds rows 8j+4*hi .. +3 of one column) maps
// to one b128 store per 4 output channels.  The reference's NCHW semantics
// (layer.py:22-26, util.py:17-44) are unchanged at the plan's boundary: graph
// inputs/outputs are converted by pl_nchw_to_q4_f32 / pl_q4_to_nchw_f32.
//
// The pipeline (2-deep register prefetch, rotated head/tail MFMA groups, one
// barrier per chunk) is the tap-major kernel's; K runs (kh, kw, cin).

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct QuadCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BK % 8 == 0, "tile alignment");
    static constexpr int KG = BK / 4;                         // k-quads per chunk
    static constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;   // [KG][rows][4]
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
    static_assert(BN <= THREADS && THREADS % BN == 0, "BN vs threads");
    static constexpr int KG_PER_PASS = THREADS / BN;
    static constexpr int B_PASSES = (KG + KG_PER_PASS - 1) / KG_PER_PASS;
    static constexpr bool B_ALL_ACTIVE = (KG % KG_PER_PASS == 0);
    static constexpr int A_VEC = BM * KG;
    static constexpr int A_PER_THREAD = (A_VEC + THREADS - 1) / THREADS;
    // Register budget.  launch_bounds(256) alone lets the compiler settle for ONE wave per SIMD
    // (it then hoists the whole epilogue's loads and burns 260+ registers).  The K loop needs the
    // accumulators, two staging sets and the fragment sets; pin the waves per SIMD that estimate
    // allows (3 for the 128x128x16 tile, 2 for 64-accumulator tiles at BK 32, 4-5 for small tiles).
    static constexpr int REGS_EST = 16 * TM * TN + 8 * (B_PASSES + A_PER_THREAD) + 4 * (BK / 8) * (TM + TN) + 36;
    static constexpr int MIN_WAVES = 512 / ((REGS_EST + 7) / 8 * 8) > 5 ? 5 : 512 / ((REGS_EST + 7) / 8 * 8);
};

// One wave's accumulators -> Q4 output (fused pass) or this (split, tile)'s slab, laid out
// [row/4][BN][4] so both sides move float4s.  Fused pass: `prm` points at the tile's per-row
// parameters in LDS ([3][BM]: bias, scale, shift -- fetched from HBM when the kernel started, so
// nothing here waits on them); every residual quad is requested before the first is needed.
template <int BM, int BN, int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void store_tile_q4(const ConvArgs &p, const TileCoord &tc, f32x16 (&acc)[TM][TN], int wm,
                                              int wn, int lane, const float *prm) {
    const int l31 = lane & 31, lhi = lane >> 5;
    if (p.splits > 1) {
        float4 *slab = reinterpret_cast<float4 *>(p.y + ((size_t)blockIdx.y * p.tile_count + tc.local) * (BM * BN));
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int R = wm * WTM + a * 32 + 8 * rq + 4 * lhi;
                    slab[(R >> 2) * BN + wn * WTN + b * 32 + l31] =
                        make_float4(acc[a][b][4 * rq], acc[a][b][4 * rq + 1], acc[a][b][4 * rq + 2], acc[a][b][4 * rq + 3]);
                }
        return;
    }
    float4 *y4 = reinterpret_cast<float4 *>(p.y);
    const float4 *res4 = reinterpret_cast<const float4 *>(p.ep.res);
    const float4 *prm4 = reinterpret_cast<const float4 *>(prm);
    if (p.y_bytes) {
        // Branch-free tail (outputs under 2 GiB): residual and y go through buffer descriptors whose range check
        // drops lanes outside the tensor -- a predicated plain load / store costs a branch and a wait per element.
        // All residual quads of a 32-row block are in flight before the first is used.
        constexpr int OOB = (int)0x80000000;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.ep.res), 0, p.ep.res ? (unsigned)p.y_bytes : 0u, 0x00020000);
        unsigned ob[TN];
        bool ck[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int jc = tc.col0 + wn * WTN + b * 32 + l31;
            ck[b] = jc < p.cols;
            unsigned n, pix;
            p.divHoWo.divmod((unsigned)(ck[b] ? jc : 0), n, pix);
            ob[b] = (n * (unsigned)p.Coq + (unsigned)(((int)tc.g * p.cout_g + tc.m0 + wm * WTM) >> 2)) * (unsigned)p.HoWo + pix;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            int off[4][TN];
            float4 rs[4][TN];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int Rl = a * 32 + 8 * rq + 4 * lhi;
                    const bool ok = ck[b] && tc.m0 + wm * WTM + Rl < p.cout_g;
                    off[rq][b] = ok ? (int)((ob[b] + (unsigned)(Rl >> 2) * (unsigned)p.HoWo) << 4) : OOB;
                    rs[rq][b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[rq][b], 0, 0));
                }
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int Rt = wm * WTM + a * 32 + 8 * rq + 4 * lhi;
                const float4 bias = prm4[Rt >> 2], scale = prm4[(BM + Rt) >> 2], shift = prm4[(2 * BM + Rt) >> 2];
                const int valid = p.cout_g - (tc.m0 + Rt);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const float4 v =
                        make_float4(acc[a][b][4 * rq], acc[a][b][4 * rq + 1], acc[a][b][4 * rq + 2], acc[a][b][4 * rq + 3]);
                    const float4 o = apply_epilogue4(p.ep, bias, scale, shift, rs[rq][b], valid, v);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), yrsrc, off[rq][b], 0, 0);
                }
            }
        }
        return;
    }
    // quad index of (first row of the wave tile, this lane's pixel); < 2^29 (checked by the host)
    unsigned obase[TN];
    bool cok[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int jc = tc.col0 + wn * WTN + b * 32 + l31;
        cok[b] = jc < p.cols;
        unsigned n, pix;
        p.divHoWo.divmod((unsigned)(cok[b] ? jc : 0), n, pix);
        obase[b] = (n * (unsigned)p.Coq + (unsigned)(((int)tc.g * p.cout_g + tc.m0 + wm * WTM) >> 2)) * (unsigned)p.HoWo + pix;
    }
    // one 32-row block at a time: its residual quads are all in flight before the first is used
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        unsigned idx[4][TN];
        bool ok[4][TN];
        float4 rs[4][TN];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int Rl = a * 32 + 8 * rq + 4 * lhi;                      // row inside the wave tile
                ok[rq][b] = cok[b] && tc.m0 + wm * WTM + Rl < p.cout_g;
                idx[rq][b] = obase[b] + (unsigned)(Rl >> 2) * (unsigned)p.HoWo;
                rs[rq][b] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (res4 && ok[rq][b]) rs[rq][b] = res4[idx[rq][b]];
            }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int Rt = wm * WTM + a * 32 + 8 * rq + 4 * lhi;               // row inside the tile
            const float4 bias = prm4[Rt >> 2], scale = prm4[(BM + Rt) >> 2], shift = prm4[(2 * BM + Rt) >> 2];
            const int valid = p.cout_g - (tc.m0 + Rt);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if (!ok[rq][b]) continue;
                const float4 v =
                    make_float4(acc[a][b][4 * rq], acc[a][b][4 * rq + 1], acc[a][b][4 * rq + 2], acc[a][b][4 * rq + 3]);
                y4[idx[rq][b]] = apply_epilogue4(p.ep, bias, scale, shift, rs[rq][b], valid, v);
            }
        }
    }
}

// Sum the split-K slabs ([row/4][BN][4]) of tiles [tile_offset, +tile_count) and
// write Q4 with the fused tail.  blockIdx.x = tile, blockIdx.y = band of row quads.
constexpr int REDUCE_Q4_QUADS = 8;   // row quads per block: BN * 8 float4 per 256 threads
template <int BM, int BN>
__global__ void __launch_bounds__(256) reduce_tiles_q4_kernel(const ConvArgs p, const float *slabs, float *y) {
    const unsigned local = blockIdx.x;
    const unsigned gt = local + (unsigned)p.tile_offset;
    const unsigned g = gt / (unsigned)p.tiles;
    const unsigned t = gt - g * (unsigned)p.tiles;
    const unsigned nt = p.divMt.div(t);
    const int m0 = (int)(t - nt * (unsigned)p.mtiles) * BM, col0 = (int)nt * BN;
    constexpr int RPP = 256 / BN > 0 ? 256 / BN : 1;        // row quads per pass (BN <= 256)
    const int cl = threadIdx.x % BN;
    const int jc = col0 + cl;
    if (jc >= p.cols) return;
    unsigned n, pix;
    p.divHoWo.divmod((unsigned)jc, n, pix);
    const size_t obase = (size_t)n * p.Coq * p.HoWo + pix;
    const size_t sstride = (size_t)p.tile_count * (BM * BN / 4);      // float4 units
    const float4 *sp = reinterpret_cast<const float4 *>(slabs) + (size_t)local * (BM * BN / 4) + cl;
    float4 *y4 = reinterpret_cast<float4 *>(y);
    const int cend = (int)g * p.cout_g + p.cout_g;
#pragma unroll
    for (int i = 0; i < REDUCE_Q4_QUADS / RPP; ++i) {
        const int rq = blockIdx.y * REDUCE_Q4_QUADS + i * RPP + threadIdx.x / BN;
        const int R = m0 + rq * 4;
        if (rq * 4 < BM && R < p.cout_g) {
            float4 v = sp[(size_t)rq * BN];
            for (int z = 1; z < p.splits; ++z) {
                const float4 w = sp[z * sstride + (size_t)rq * BN];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            const int c0 = (int)g * p.cout_g + R;
            const size_t idx4 = obase + (size_t)(c0 >> 2) * p.HoWo;
            float bs[4], sc[4], sh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) load_chan_params(p.ep, min(c0 + e, cend - 1), bs[e], sc[e], sh[e]);
            float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.ep.res) rs = reinterpret_cast<const float4 *>(p.ep.res)[idx4];
            y4[idx4] = apply_epilogue4(p.ep, make_float4(bs[0], bs[1], bs[2], bs[3]), make_float4(sc[0], sc[1], sc[2], sc[3]),
                                       make_float4(sh[0], sh[1], sh[2], sh[3]), rs, cend - c0, v);
        }
    }
}

template <class C>
__device__ __forceinline__ void conv_q4_body(const ConvArgs &p, unsigned bid = blockIdx.x, unsigned nblk = gridDim.x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                         // [2][KG][BM][4]
    float *Bs = smem + 2 * C::A_ELEMS;        // [2][KG][BN][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;

    const TileCoord tc = tile_coord<C::BM, C::BN>(p, bid, nblk);
    const unsigned g = tc.g;
    const int m0 = tc.m0, col0 = tc.col0;
    const int split = blockIdx.y;
    const int total_chunks = (p.Qtot + C::KG - 1) / C::KG;
    const int cbeg = split * p.k_per_split;                    // in chunks
    const int nchunks = min(total_chunks, cbeg + p.k_per_split) - cbeg;

    // per-row epilogue parameters: requested now, parked in LDS after the K loop
    float prm_b = 0.f, prm_sc = 1.f, prm_sh = 0.f;
    if (p.splits <= 1 && tid < C::BM)
        load_chan_params(p.ep, (int)g * p.cout_g + min(m0 + tid, p.cout_g - 1), prm_b, prm_sc, prm_sh);

    // ---- B: thread -> (pixel column jl, k-quad kg0 + pass*KG_PER_PASS) ----------
    // for BN >= 64 the k-quad index is wave-uniform: its tap / channel arithmetic runs on the SALU
    constexpr bool KG_UNIFORM = C::BN >= 64;
    constexpr int WPC = KG_UNIFORM ? C::BN / 64 : 1;           // waves per column block
    const int jl = KG_UNIFORM ? (wave % WPC) * 64 + lane : tid % C::BN;
    const int kg0 = KG_UNIFORM ? wave / WPC : tid / C::BN;
    const int j = col0 + jl;
    bool jok = j < p.cols && (C::B_ALL_ACTIVE || kg0 < C::KG);
    int hbase = -(1 << 20), wbase = 0, cbase = 0, j_n = 0;
    if (jok) {
        unsigned n, pix, ho, wo;
        p.divHoWo.divmod((unsigned)j, n, pix);
        p.divWo.divmod(pix, ho, wo);
        hbase = (int)ho * p.sh - p.pt;
        wbase = (int)wo * p.sw - p.pl;
        cbase = ((int)n * p.Cq + (int)g * p.cqg) * p.HW + hbase * p.W + wbase;     // in quads
        j_n = (int)n;
    }
    constexpr int OOB = (int)0x80000000;
    // row-packed input: byte offset of (n, ho*sh, wo*sw, channel 0) in the padded NHWC image
    // (hbase / wbase carry -pad; the padded image starts at -pad)
    const int rp_base = (p.rp_rq && jok) ? (((j_n * p.H + hbase + p.pt) * p.W + wbase + p.pl) * p.Cin) << 2 : OOB;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- A: float4 v -> (k-quad kq = v / BM, filter row v % BM); rows are contiguous in wq ----
    int aoff[C::A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD; ++i) {
        const int v = tid + i * C::THREADS;
        const int kq = v / C::BM, row = v % C::BM;
        const bool rok = (C::A_VEC % C::THREADS == 0 || v < C::A_VEC) && m0 + row < p.cout_g;
        aoff[i] = rok ? ((((int)g * p.Qpad + kq) * p.cout_g + m0 + row) << 4) : OOB;
    }

    float4 breg0[C::B_PASSES], breg1[C::B_PASSES];
    float4 areg0[C::A_PER_THREAD], areg1[C::A_PER_THREAD];

    // Where chunk c sits on the K axis.  Common case (p.uni: Cin_g/4 is a multiple of the chunk's
    // k-quads, so a chunk never straddles two filter taps): the tap position is carried as scalar
    // state and advanced by compare/select -- the two exact divisions per k-quad that the general
    // case needs are ~55 scalar instructions per chunk, which a lone wave per SIMD cannot hide
    // (it issues at most one instruction every 4 cycles; the matrix pipe drains after 64).
    // General case (small or odd Cin, e.g. the 3->4 channel stem): every k-quad derives its own
    // (tap, channel quad); past-the-end quads (K padding) are pushed out of bounds arithmetically.
    int st_c = 0, st_cq0, st_a, st_b;
    {
        const int q0 = cbeg * C::KG;
        const unsigned tap = p.divCpt.div((unsigned)q0);
        unsigned a, b;
        p.divKw.divmod(tap, a, b);
        st_cq0 = q0 - (int)tap * p.cqg;
        st_a = (int)a;
        st_b = (int)b;
    }
    auto load_chunk = [&](int c, float4 (&breg)[C::B_PASSES], float4 (&areg)[C::A_PER_THREAD]) {
        const int q0 = (cbeg + c) * C::KG;                      // first k-quad of the chunk (uniform)
        if (p.rp_rq) {
            // Row-packed small-Cin input (the 3-channel 7x7 stem): x is [N][H+2p][Wp][Cin] with the
            // zero border already in place, so a filter row's kw*Cin floats are contiguous and K runs
            // (filter row, quad of that row segment).  The thread's base offset never changes, the
            // k-quad offset is scalar, there is nothing to range-check but the column itself.
#pragma unroll
            for (int ps = 0; ps < C::B_PASSES; ++ps) {
                const int q = min(q0 + kg0 + ps * C::KG_PER_PASS, p.Qtot - 1);     // K padding: zero filter rows
                const int a = (int)p.divCpt.div((unsigned)q);
                const int off = (a * p.W * p.Cin + (q - a * p.rp_rq) * 4) << 2;     // p.W = padded row length
                if constexpr (KG_UNIFORM)
                    breg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, rp_base, off, 0));
                else
                    breg[ps] = __builtin_bit_cast(
                        float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, jok ? rp_base + off : OOB, 0, 0));
            }
        } else if (p.uni) {
            // chunks are requested in order, each at most one further than the last (c is clamped at the end)
            const int ncq = st_cq0 + (c - st_c) * C::KG;
            st_c = c;
            const bool wrap = ncq >= p.cqg;
            st_cq0 = wrap ? 0 : ncq;
            const int nb = st_b + (wrap ? 1 : 0);
            const bool wb = nb >= p.kw;
            st_b = wb ? 0 : nb;
            st_a += wb ? 1 : 0;
            const int dy = st_a * p.dh, dx = st_b * p.dw;
            const bool ok = (unsigned)(hbase + dy) < (unsigned)p.H && (unsigned)(wbase + dx) < (unsigned)p.W;
            const int voff = ok ? (int)((unsigned)(cbase + dy * p.W + dx) << 4) : OOB;
#pragma unroll
            for (int ps = 0; ps < C::B_PASSES; ++ps) {
                const int coff = ((st_cq0 + kg0 + ps * C::KG_PER_PASS) * p.HW) << 4;
                if constexpr (KG_UNIFORM)      // channel-plane offset is scalar: rides in the soffset operand
                    breg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, coff, 0));
                else
                    breg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? voff + coff : OOB, 0, 0));
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < C::B_PASSES; ++ps) {
                const int q = q0 + kg0 + ps * C::KG_PER_PASS;
                const unsigned tap = p.divCpt.div((unsigned)q);
                const int cq = q - (int)tap * p.cqg;
                unsigned a, b;
                p.divKw.divmod(tap, a, b);
                const int dy = q < p.Qtot ? (int)a * p.dh : (1 << 15), dx = (int)b * p.dw;   // |hbase| < 2^14
                const bool ok = (unsigned)(hbase + dy) < (unsigned)p.H && (unsigned)(wbase + dx) < (unsigned)p.W;
                const int voff = (int)((unsigned)(cbase + dy * p.W + dx) << 4);   // garbage when !ok, never used
                const int coff = (cq * p.HW) << 4;
                if constexpr (KG_UNIFORM)
                    breg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? voff : OOB, coff, 0));
                else
                    breg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? voff + coff : OOB, 0, 0));
            }
        }
        const int ksoff = (q0 * p.cout_g) << 4;                 // scalar: chunk start along q
#pragma unroll
        for (int i = 0; i < C::A_PER_THREAD; ++i)
            areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, aoff[i], ksoff, 0));
    };

    auto store_chunk = [&](int buf, const float4 (&breg)[C::B_PASSES], const float4 (&areg)[C::A_PER_THREAD]) {
        float *Ab = As + buf * C::A_ELEMS;
        float *Bb = Bs + buf * C::B_ELEMS;
#pragma unroll
        for (int ps = 0; ps < C::B_PASSES; ++ps) {
            const int kg = kg0 + ps * C::KG_PER_PASS;
            if (C::B_ALL_ACTIVE || kg < C::KG)
                *reinterpret_cast<float4 *>(Bb + (kg * C::BN + jl) * 4) = breg[ps];
        }
#pragma unroll
        for (int i = 0; i < C::A_PER_THREAD; ++i) {
            const int v = tid + i * C::THREADS;
            if (C::A_VEC % C::THREADS == 0 || v < C::A_VEC)
                *reinterpret_cast<float4 *>(Ab + v * 4) = areg[i];     // v = kq*BM + row
        }
    };

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment of MFMA step group u: k-quad 2u + (lane>>5), row (lane&31) of the 32-row block
    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (lhi * C::BM + wm * C::WTM + l31) * 4;
    const int b_off = (lhi * C::BN + wn * C::WTN + l31) * 4;
    constexpr int U = C::BK / 8;
    float4 fa0[U][C::TM], fb0[U][C::TN], fa1[U][C::TM], fb1[U][C::TN];

    auto read_frags = [&](int buf, float4 (&af)[U][C::TM], float4 (&bf)[U][C::TN]) {
        const float *Ab = As + buf * C::A_ELEMS + a_off;
        const float *Bb = Bs + buf * C::B_ELEMS + b_off;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int a = 0; a < C::TM; ++a)
                af[u][a] = *reinterpret_cast<const float4 *>(Ab + (2 * u * C::BM + a * 32) * 4);
#pragma unroll
            for (int b = 0; b < C::TN; ++b)
                bf[u][b] = *reinterpret_cast<const float4 *>(Bb + (2 * u * C::BN + b * 32) * 4);
        }
    };
    auto mma = [&](const float4 (&af)[U][C::TM], const float4 (&bf)[U][C::TN], int u0, int u1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < u0 || u >= u1) continue;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int a = 0; a < C::TM; ++a)
#pragma unroll
                    for (int b = 0; b < C::TN; ++b) {
                        const float av = s4 == 0 ? af[u][a].x : s4 == 1 ? af[u][a].y : s4 == 2 ? af[u][a].z : af[u][a].w;
                        const float bv = s4 == 0 ? bf[u][b].x : s4 == 1 ? bf[u][b].y : s4 == 2 ? bf[u][b].z : bf[u][b].w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
        }
    };

    if (nchunks > 0) {
        const int last = nchunks - 1;
        // Rotated pipeline step (see conv_tap_kernel): fragments of chunk k are read while the
        // TAIL group of chunk k-1 runs; chunk k+1 goes to LDS and chunk k+2 is requested from
        // HBM under the HEAD groups of chunk k.  Past-the-end loads are clamped duplicates.
        auto step = [&](auto parity, int k, bool with_tail) {
            constexpr int P = decltype(parity)::value;
            if constexpr (P == 0) {
                read_frags(0, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                if (with_tail) mma(fa1, fb1, U - 1, U);
                store_chunk(1, breg1, areg1);
                __builtin_amdgcn_sched_barrier(0);
                load_chunk(min(k + 2, last), breg0, areg0);
                mma(fa0, fb0, 0, U - 1);
            } else {
                read_frags(1, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                if (with_tail) mma(fa0, fb0, U - 1, U);
                store_chunk(0, breg0, areg0);
                __builtin_amdgcn_sched_barrier(0);
                load_chunk(min(k + 2, last), breg1, areg1);
                mma(fa1, fb1, 0, U - 1);
            }
            __syncthreads();
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        load_chunk(0, breg0, areg0);
        load_chunk(min(1, last), breg1, areg1);
        store_chunk(0, breg0, areg0);
        __syncthreads();
        step(P0{}, 0, false);
        int k = 1;
        for (; k + 1 <= last; k += 2) {
            step(P1{}, k, true);
            step(P0{}, k + 1, true);
        }
        if (k <= last) {
            step(P1{}, k, true);
            mma(fa1, fb1, U - 1, U);
        } else {
            mma(fa0, fb0, U - 1, U);
        }
    }
    // every wave is past its last LDS read (the final barrier sits after the last fragment reads)
    if (p.splits <= 1) {
        if (tid < C::BM) {
            smem[tid] = prm_b;
            smem[C::BM + tid] = prm_sc;
            smem[2 * C::BM + tid] = prm_sh;
        }
        __syncthreads();
    }
    store_tile_q4<C::BM, C::BN, C::TM, C::TN, C::WTM, C::WTN>(p, tc, acc, wm, wn, lane, smem);
}

template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C::MIN_WAVES)))
conv_q4_kernel(const ConvArgs p) {
    conv_q4_body<C>(p);
}

// Two convolutions that read the SAME input in one launch: workgroups [0, pa.tile_count) run conv a, the rest conv b (both
// unsplit).  Made for a stride-2 3x3 conv and the 1x1 stride-2 projection beside it (a ResNet block that changes resolution):
// the projection alone is 392-784 tiles of two K chunks -- launch-bound at 0.25 of the matrix peak -- and re-reads exactly the
// pixels its sibling gathers; here its tiles fill the tail of the sibling's grid and find those pixels in the L2.
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C::MIN_WAVES)))
conv_q4_pair_kernel(const ConvArgs pa, const ConvArgs pb) {
    if (blockIdx.x < (unsigned)pa.tile_count)
        conv_q4_body<C>(pa, blockIdx.x, (unsigned)pa.tile_count);
    else
        conv_q4_body<C>(pb, blockIdx.x - (unsigned)pa.tile_count, (unsigned)pb.tile_count);
}

// OIHW [g*cout_g + co][cin_g][tap]  ->  wq[g][q][co][4], q = tap*cqg + cin/4 (zero padded to Qpad
// k-quads per group and to 4 channels per quad)
__global__ void __launch_bounds__(256) pack_filter_q4_kernel(const float *w, float *out, unsigned total, int cout_g,
                                                             int cin_g, int khw, int cqg, int Qtot, int Qpad,
                                                             FastDiv divCo, FastDiv divQpad, FastDiv divCqg) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {   // i = (g*Qpad + q)*cout_g + co
        unsigned r, co, g, q;
        divCo.divmod(i, r, co);
        divQpad.divmod(r, g, q);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)q < Qtot) {
            unsigned tap, cq;
            divCqg.divmod(q, tap, cq);
            const float *src = w + ((size_t)(g * cout_g + co) * cin_g + cq * 4) * khw + tap;
            const int left = cin_g - (int)cq * 4;
            v.x = src[0];
            if (left > 1) v.y = src[khw];
            if (left > 2) v.z = src[2 * khw];
            if (left > 3) v.w = src[3 * khw];
        }
        reinterpret_cast<float4 *>(out)[i] = v;
    }
}

// NCHW -> Q4 (pad channels written as zeros) and back
__global__ void __launch_bounds__(256) nchw_to_q4_kernel(const float *x, float *y, unsigned total, int C, int Cq,
                                                         int HW, FastDiv divHW, FastDiv divCq) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {   // i = (n*Cq + cq)*HW + pix
        unsigned r, pix, n, cq;
        divHW.divmod(i, r, pix);
        divCq.divmod(r, n, cq);
        const float *src = x + ((size_t)n * C + cq * 4) * HW + pix;
        const int left = C - (int)cq * 4;
        float4 v = make_float4(src[0], 0.f, 0.f, 0.f);
        if (left > 1) v.y = src[HW];
        if (left > 2) v.z = src[2 * (size_t)HW];
        if (left > 3) v.w = src[3 * (size_t)HW];
        reinterpret_cast<float4 *>(y)[i] = v;
    }
}

__global__ void __launch_bounds__(256) q4_to_nchw_kernel(const float *x, float *y, unsigned total, int C, int Cq,
                                                         int HW, FastDiv divHW, FastDiv divCq) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned r, pix, n, cq;
        divHW.divmod(i, r, pix);
        divCq.divmod(r, n, cq);
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float *dst = y + ((size_t)n * C + cq * 4) * HW + pix;
        const int left = C - (int)cq * 4;
        dst[0] = v.x;
        if (left > 1) dst[HW] = v.y;
        if (left > 2) dst[2 * (size_t)HW] = v.z;
        if (left > 3) dst[3 * (size_t)HW] = v.w;
    }
}

// NCHW -> zero-padded NHWC rows for the row-packed gather: y[n][h + pt][w + pl][c], border = 0.
// One pass: a thread produces 4 consecutive floats of y (one b128 store), i.e. 4 (pixel, channel)
// elements gathered from up to 3 channel planes.  The four gathers are buffer loads whose offset is pushed out of
// range for border / slack elements (zero fill by the range check): a predicated plain load (`in ? x[i] : 0`) is a
// branch plus a wait per element, so the four went out one at a time (10.7 us for ResNet-18's batch against the
// ~6.5 us its 39 MB take at the copy rate).
__global__ void __launch_bounds__(256) nchw_to_rowpack_kernel(const float *x, float *y, unsigned total4, unsigned total,
                                                              int C, int H, int W, int Hp, int Wp, int pt, int pl,
                                                              unsigned x_bytes, FastDiv divC, FastDiv divWp, FastDiv divHp) {
    const unsigned stride = gridDim.x * 256;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, x_bytes, 0x00020000);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned f = i * 4 + e;                       // float index in y
            unsigned pix, c, r, wp, n, hp;
            divC.divmod(f, pix, c);
            divWp.divmod(pix, r, wp);
            divHp.divmod(r, n, hp);
            const int h = (int)hp - pt, w = (int)wp - pl;
            const bool in = f < total && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            const unsigned off = (((n * (unsigned)C + c) * (unsigned)H + (unsigned)h) * (unsigned)W + (unsigned)w) << 2;
            v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, in ? (int)off : (int)0x80000000, 0, 0));
        }
        reinterpret_cast<float4 *>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// OIHW -> wq[q = a*RQ + jq][co][4]: element e of quad jq is float 4*jq + e of filter row a's
// (kw, cin) segment, i.e. tap kw = f / Cin, channel f % Cin (zero beyond kw*Cin and beyond Qtot)
__global__ void __launch_bounds__(256) pack_filter_rowpack_kernel(const float *w, float *out, unsigned total, int Cout,
                                                                  int Cin, int kh, int kw, int RQ, int Qtot,
                                                                  FastDiv divCo, FastDiv divRQ) {
    const unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {      // i = q*Cout + co
        unsigned q, co;
        divCo.divmod(i, q, co);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((int)q < Qtot) {
            unsigned a, jq;
            divRQ.divmod(q, a, jq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = (int)jq * 4 + e;
                if (f < kw * Cin) v[e] = w[(((size_t)co * Cin + f % Cin) * kh + a) * kw + f / Cin];
            }
        }
        reinterpret_cast<float4 *>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
