// Conv2d / Dense / MatMul forward as one implicit-GEMM kernel on the fp32
// matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s).
//
// Reference semantics: layer.Conv2d (layer.py:22-26) = util.conv_for
// (util.py:17-44): out[Cout, N*Ho*Wo] = K.reshape(Cout, Cin/g*kh*kw) @ im2col(x)
// with the K axis ordered (cin, kh, kw).  The reference materialises im2col in a
// scratch array with 9 strided slab copies and calls sgemm; here the im2col
// matrix never exists: each workgroup gathers its (BK x BN) slice of it
// straight from the NCHW input into LDS (zero-filling the padding halo),
// stages the matching (BM x BK) slice of the OIHW weights (already row-major
// in K, so no re-layout), and feeds 32x32x2 MFMAs from LDS.
//
// Mapping for MI355X:
//  - GEMM columns (n, ho, wo) are flattened, so consecutive lanes read
//    consecutive wo -> coalesced NCHW loads; output stores are coalesced the
//    same way (y is NCHW, i.e. row-major [Cout][Ho*Wo] per image).
//  - 256 threads = 4 wave64; each wave owns a (WTM x WTN) block of 32x32 MFMA
//    tiles.  A fragment: lane l holds A[row=l&31][k=l>>5]; B fragment: lane l
//    holds B[k=l>>5][col=l&31]; both are single ds_read_b32 with conflict-free
//    addressing (A is stored k-major in LDS with a +2 row pad so the
//    transposing ds_writes of the float4 weight loads spread over all banks).
//  - fp32 MFMA is slow relative to the memory system (64 cycles per 32x32x2),
//    so address arithmetic of the gather (exact magic-number div/mod) hides
//    under the matrix pipe; global loads for chunk t+1 are issued before the
//    MFMAs of chunk t (register-staged double buffering, one barrier per chunk).
//  - blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs (own L2) walks a
//    contiguous range of tiles, M-tiles fastest, so co-resident workgroups on
//    an XCD share the same input pixels and weight panels in L2.
//  - small-spatial layers (ResNet layer3/4 at batch 32 have 6272 / 1568 GEMM
//    columns) use split-K over gridDim.y with a second pass that sums the
//    slabs and applies the fused tail.
#include "conv_shared.h"

namespace {

// Write one wave's accumulators.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Fused pass: NCHW + epilogue.  Split
// pass: raw partial sums into this (split, tile)'s compact slab.
template <int BM, int BN, int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void store_tile(const ConvArgs &p, const TileCoord &tc, f32x16 (&acc)[TM][TN], int wm,
                                           int wn, int lane) {
    const int l31 = lane & 31, lhi = lane >> 5;
    if (p.splits > 1) {
        float *slab = p.y + ((size_t)blockIdx.y * p.tile_count + tc.local) * (BM * BN);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    slab[row * BN + wn * WTN + b * 32 + l31] = acc[a][b][r];
                }
        return;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int jc = tc.col0 + wn * WTN + b * 32 + l31;
        if (jc >= p.cols) continue;
        unsigned n, pix;
        p.divHoWo.divmod((unsigned)jc, n, pix);
        const size_t obase = ((size_t)n * p.Cout + (size_t)tc.g * p.cout_g) * p.HoWo + pix;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tc.m0 + wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row < p.cout_g) {
                    const size_t idx = obase + (size_t)row * p.HoWo;
                    p.y[idx] = apply_epilogue(p.ep, acc[a][b][r], (int)tc.g * p.cout_g + row, idx);
                }
            }
        }
    }
}

// Combine the split-K slabs of the tiles [tile_offset, tile_offset+tile_count)
// and apply the fused tail.  blockIdx.x = tile, blockIdx.y = band of rows.
// VEC4: a thread owns 4 neighbouring columns (float4 slab reads, float4 NCHW
// stores; needs Ho*Wo % 4 == 0 so the 4 pixels share an image and stay aligned).
constexpr int REDUCE_ROWS = 16;
template <int BM, int BN, bool VEC4>
__global__ void __launch_bounds__(256) reduce_tiles_kernel(const ConvArgs p, const float *slabs, float *y) {
    const unsigned local = blockIdx.x;
    const unsigned gt = local + (unsigned)p.tile_offset;
    const unsigned g = gt / (unsigned)p.tiles;
    const unsigned t = gt - g * (unsigned)p.tiles;
    const unsigned nt = p.divMt.div(t);
    const int m0 = (int)(t - nt * (unsigned)p.mtiles) * BM, col0 = (int)nt * BN;
    constexpr int CW = VEC4 ? 4 : 1;                 // columns per thread
    constexpr int TPR = BN / CW;                     // threads per row
    constexpr int RPP = 256 / TPR;                   // rows per pass
    constexpr int ROWS = VEC4 ? (REDUCE_ROWS * 4 > BM ? BM : REDUCE_ROWS * 4) : REDUCE_ROWS;   // rows per block
    const int cl = (threadIdx.x % TPR) * CW, r0 = blockIdx.y * ROWS + threadIdx.x / TPR;
    const int jc = col0 + cl;
    if (jc >= p.cols) return;                        // cols % 4 == 0 in VEC4 mode: all or nothing
    unsigned n, pix;
    p.divHoWo.divmod((unsigned)jc, n, pix);
    const size_t obase = ((size_t)n * p.Cout + (size_t)g * p.cout_g) * p.HoWo + pix;
    const size_t sstride = (size_t)p.tile_count * (BM * BN);
    const float *sp = slabs + (size_t)local * (BM * BN) + cl;
#pragma unroll
    for (int i = 0; i < ROWS / RPP; ++i) {
        const int rl = r0 + i * RPP;
        const int row = m0 + rl;
        if (rl < BM && row < p.cout_g) {
            const size_t idx = obase + (size_t)row * p.HoWo;
            const int c = (int)g * p.cout_g + row;
            if (VEC4) {
                float4 v = *reinterpret_cast<const float4 *>(sp + rl * BN);
                for (int z = 1; z < p.splits; ++z) {
                    const float4 w = *reinterpret_cast<const float4 *>(sp + z * sstride + rl * BN);
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                v.x = apply_epilogue(p.ep, v.x, c, idx);
                v.y = apply_epilogue(p.ep, v.y, c, idx + 1);
                v.z = apply_epilogue(p.ep, v.z, c, idx + 2);
                v.w = apply_epilogue(p.ep, v.w, c, idx + 3);
                *reinterpret_cast<float4 *>(y + idx) = v;
            } else {
                float v = sp[rl * BN];
                for (int z = 1; z < p.splits; ++z) v += sp[z * sstride + rl * BN];
                y[idx] = apply_epilogue(p.ep, v, c, idx);
            }
        }
    }
}

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    static constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be 32-aligned");
    static constexpr int LDA = BM + 2;                  // k-major A tile, padded
    static constexpr int LDB = BN;
    static constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4 + 2 * BK * 8;  // + per-k lookup table
    // B gather: thread owns column tid%BN and rows tid/BN + i*ROWS_PER_PASS
    static_assert(THREADS % BN == 0 || BN % THREADS == 0, "BN vs threads");
    static constexpr int ROWS_PER_PASS = THREADS / BN;  // BN <= 256
    static constexpr int B_PER_THREAD = BK / ROWS_PER_PASS;
    // A stage: float4 along K
    static constexpr int A_VEC = BM * BK / 4;            // float4 per chunk
    static constexpr int A_PER_THREAD = (A_VEC + THREADS - 1) / THREADS;
    static constexpr int KQ = BK / 4;                    // float4 per A row
};

template <class C, bool AVEC>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                                   // [2][BK][LDA]
    float *Bs = smem + 2 * C::A_ELEMS;                  // [2][BK][LDB]
    int2 *Lut = reinterpret_cast<int2 *>(smem + 2 * (C::A_ELEMS + C::B_ELEMS));  // [2][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;

    const TileCoord tc = tile_coord<C::BM, C::BN>(p);
    const unsigned g = tc.g;
    const int m0 = tc.m0, col0 = tc.col0;
    const int split = blockIdx.y;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nchunks = (kend - kbeg + C::BK - 1) / C::BK;

    // ---- per-thread gather column (fixed for the whole K loop) -------------
    const int jl = tid % C::BN;
    const int krow0 = tid / C::BN;
    const int j = col0 + jl;
    const bool jok = j < p.cols;
    // invalid column: hbase so negative that every row test fails
    int hbase = -(1 << 20), wbase = 0, cbase = 0;
    if (jok) {
        unsigned n, pix, ho, wo;
        p.divHoWo.divmod((unsigned)j, n, pix);
        p.divWo.divmod(pix, ho, wo);
        hbase = (int)ho * p.sh - p.pt;
        wbase = (int)wo * p.sw - p.pl;
        cbase = (int)n * p.Cin * p.HW + (int)g * p.cin_g * p.HW + hbase * p.W + wbase;
    }

    // Staging loads are buffer loads through wave-uniform descriptors built
    // from kernel arguments: the hardware range check returns 0 for an
    // out-of-range offset, so padding halo, K tail, row tail and column tail
    // are all "offset = OOB" -- no per-lane branch, no select on the loaded
    // value, nothing that makes hipcc wait vmcnt(0) in the middle of the gather.
    constexpr int OOB = (int)0x80000000;  // >= num_records (tensors are < 2 GiB, checked on the host)
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    const int wbase_row = (int)g * p.cout_g + m0;       // filter row of tile row 0

    // Per-k lookup table, one chunk ahead: {element offset cin*HW + a*dh*W + b*dw,
    // (a*dh) | (b*dw) << 16}.  k >= kend gets a row delta that fails every range test.
    auto write_lut = [&](int chunk) {
        if (tid < C::BK) {
            const int k = kbeg + chunk * C::BK + tid;
            unsigned cin, r, a, b;
            p.divKhw.divmod((unsigned)k, cin, r);
            p.divKw.divmod(r, a, b);
            int2 e;
            e.x = (int)cin * p.HW + (int)a * p.dh * p.W + (int)b * p.dw;
            e.y = k < kend ? ((int)a * p.dh) | (((int)b * p.dw) << 16) : 0x7fff;
            Lut[(chunk & 1) * C::BK + tid] = e;
        }
    };

    float breg[C::B_PER_THREAD];
    float4 areg[C::A_PER_THREAD];

    auto load_chunk = [&](int chunk) {
        const int k0 = kbeg + chunk * C::BK;
        const int2 *lut = Lut + (chunk & 1) * C::BK;
        // B: im2col gather.  Table entries first (wave-uniform addresses: LDS
        // broadcast reads), pinned so hipcc cannot make them lazy/conditional.
        int2 e[C::B_PER_THREAD];
#pragma unroll
        for (int i = 0; i < C::B_PER_THREAD; ++i) e[i] = lut[krow0 + i * C::ROWS_PER_PASS];
#pragma unroll
        for (int i = 0; i < C::B_PER_THREAD; ++i) asm volatile("" : "+v"(e[i].x), "+v"(e[i].y));
#pragma unroll
        for (int i = 0; i < C::B_PER_THREAD; ++i) {
            const int hi = hbase + (e[i].y & 0xffff), wi = wbase + (e[i].y >> 16);
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int off = ok ? (cbase + e[i].x) << 2 : OOB;
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, off, 0, 0));
        }
        // A: weights, row-major [cout][K]
#pragma unroll
        for (int i = 0; i < C::A_PER_THREAD; ++i) {
            const int v = tid + i * C::THREADS;
            const int row = v / C::KQ, kq = v % C::KQ;
            const int k = k0 + kq * 4;
            const bool rok = (C::A_VEC % C::THREADS == 0 || v < C::A_VEC) && m0 + row < p.cout_g;
            const int eoff = (wbase_row + row) * p.K + k;                       // elements
            if (AVEC) {
                // K % 4 == 0 and kend % 4 == 0: a float4 is entirely inside or outside
                const int off = (rok && k < kend) ? eoff << 2 : OOB;
                areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
            } else {
                float tv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int off = (rok && k + e < kend) ? (eoff + e) << 2 : OOB;
                    tv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, off, 0, 0));
                }
                areg[i] = make_float4(tv[0], tv[1], tv[2], tv[3]);
            }
        }
    };

    auto store_chunk = [&](int buf) {
        float *Ab = As + buf * C::A_ELEMS;
        float *Bb = Bs + buf * C::B_ELEMS;
#pragma unroll
        for (int i = 0; i < C::B_PER_THREAD; ++i)
            Bb[(krow0 + i * C::ROWS_PER_PASS) * C::LDB + jl] = breg[i];
#pragma unroll
        for (int i = 0; i < C::A_PER_THREAD; ++i) {
            const int v = tid + i * C::THREADS;
            if (C::A_VEC % C::THREADS == 0 || v < C::A_VEC) {
                const int row = v / C::KQ, kq = v % C::KQ;
                float *dst = Ab + (kq * 4) * C::LDA + row;
                dst[0] = areg[i].x;
                dst[C::LDA] = areg[i].y;
                dst[2 * C::LDA] = areg[i].z;
                dst[3 * C::LDA] = areg[i].w;
            }
        }
    };

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = lhi * C::LDA + wm * C::WTM + l31;
    const int b_off = lhi * C::LDB + wn * C::WTN + l31;

    auto compute = [&](int buf) {
        const float *Ab = As + buf * C::A_ELEMS + a_off;
        const float *Bb = Bs + buf * C::B_ELEMS + b_off;
        // all fragments of the chunk first (one LDS round trip), then the MFMAs
        float af[C::BK / 2][C::TM], bf[C::BK / 2][C::TN];
#pragma unroll
        for (int kk = 0; kk < C::BK; kk += 2) {
#pragma unroll
            for (int a = 0; a < C::TM; ++a) af[kk / 2][a] = Ab[kk * C::LDA + a * 32];
#pragma unroll
            for (int b = 0; b < C::TN; ++b) bf[kk / 2][b] = Bb[kk * C::LDB + b * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < C::BK / 2; ++kk)
#pragma unroll
            for (int a = 0; a < C::TM; ++a)
#pragma unroll
                for (int b = 0; b < C::TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][a], bf[kk][b], acc[a][b], 0, 0, 0);
        // keep the staging stores (and their vmcnt waits) BEHIND the MFMAs
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue ------------------------------------------------------------
    write_lut(0);
    __syncthreads();
    load_chunk(0);
    write_lut(1);
    store_chunk(0);
    __syncthreads();

    // ---- main loop: straight-line body (loads for c+1 fly over the MFMAs of c) --
    for (int c = 0; c + 1 < nchunks; ++c) {
        load_chunk(c + 1);
        write_lut(c + 2);
        compute(c & 1);
        store_chunk((c + 1) & 1);
        __syncthreads();
    }
    if (nchunks > 0) compute((nchunks - 1) & 1);

    store_tile<C::BM, C::BN, C::TM, C::TN, C::WTM, C::WTN>(p, tc, acc, wm, wn, lane);
}

// =============================================================================
// Tap-major kernel: the fast path for convs whose weights were re-ordered once
// (pl_conv2d_prepare_weights_f32) from OIHW [co][cin][tap] to [co][tap][cin].
// The GEMM K axis then runs (kh, kw, cin), so a whole BK-chunk shares one filter
// tap: padding validity and the spatial offset are computed ONCE per chunk per
// thread, and the per-element part of the im2col address (cin * H*W) is wave
// uniform and rides in the buffer load's scalar offset -- zero VALU per
// gathered element.  LDS tiles are [row][k] (k contiguous, +4 pad), filled with
// ds_write_b128 and read as ds_read_b128 fragments: lane (i, hi) takes k =
// 8u+4hi .. 8u+4hi+3 and feeds MFMA step s with its s-th value, i.e. step s pairs
// k = 8u+s (lanes 0-31) with k = 8u+4+s (lanes 32-63) for A and B alike.
// Requires cin_g % BK == 0 (ResNet/YOLO bodies); everything else takes the
// generic kernel above.
// =============================================================================
template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct TapCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BK % 8 == 0, "tile alignment");
    static constexpr int LDK = BK + 4;                        // floats; LDK/4 odd -> conflict-free b128
    static_assert((LDK / 4) % 2 == 1, "LDK/4 must be odd");
    static constexpr int A_ELEMS = BM * LDK, B_ELEMS = BN * LDK;
    static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
    static constexpr int KG = BK / 4;                         // float4 groups along k
    static constexpr int KG_PER_PASS = THREADS / BN;          // k-groups covered by one pass of all threads
    static constexpr int B_PASSES = (KG + KG_PER_PASS - 1) / KG_PER_PASS;
    static constexpr bool B_ALL_ACTIVE = (KG % KG_PER_PASS == 0);
    static constexpr int A_VEC = BM * KG;
    static constexpr int A_PER_THREAD = (A_VEC + THREADS - 1) / THREADS;
};

template <class C>
__global__ void __launch_bounds__(256) conv_tap_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                         // [2][BM][LDK]
    float *Bs = smem + 2 * C::A_ELEMS;        // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;

    const TileCoord tc = tile_coord<C::BM, C::BN>(p);
    const unsigned g = tc.g;
    const int m0 = tc.m0, col0 = tc.col0;
    const int split = blockIdx.y;
    // K range of this split, in BK-chunks of the tap-major axis
    const int cpt = p.cin_g / C::BK;                           // chunks per tap
    const int total_chunks = p.kh * p.kw * cpt;
    const int cbeg = split * p.k_per_split;                    // k_per_split counts chunks here
    const int nchunks = min(total_chunks, cbeg + p.k_per_split) - cbeg;

    // ---- per-thread gather column ------------------------------------------
    const int jl = tid % C::BN;
    const int kg0 = tid / C::BN;
    const int j = col0 + jl;
    const bool jok = j < p.cols && (C::B_ALL_ACTIVE || kg0 < C::KG);
    int hbase = -(1 << 20), wbase = 0, cbase = 0;
    if (jok) {
        unsigned n, pix, ho, wo;
        p.divHoWo.divmod((unsigned)j, n, pix);
        p.divWo.divmod(pix, ho, wo);
        hbase = (int)ho * p.sh - p.pt;
        wbase = (int)wo * p.sw - p.pl;
        cbase = (int)n * p.Cin * p.HW + (int)g * p.cin_g * p.HW + hbase * p.W + wbase + kg0 * 4 * p.HW;
    }
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    // A: per-thread constant part of the weight offset (elements), OOB for rows past Cout
    int aoff[C::A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD; ++i) {
        const int v = tid + i * C::THREADS;
        const int row = v / C::KG, kq = v % C::KG;
        const bool rok = (C::A_VEC % C::THREADS == 0 || v < C::A_VEC) && m0 + row < p.cout_g;
        aoff[i] = rok ? (((int)g * p.cout_g + m0 + row) * p.K + kq * 4) << 2 : OOB;
    }

    // two staging register sets: loads run TWO chunks ahead of the MFMAs, so a
    // lone wave per SIMD (the batch-32 regime: 1-3 workgroups per CU) still
    // covers the ~1000-cycle L2 latency with its own matrix work
    float4 breg0[C::B_PASSES], breg1[C::B_PASSES];
    float4 areg0[C::A_PER_THREAD], areg1[C::A_PER_THREAD];

    auto load_chunk = [&](int c, float4 (&breg)[C::B_PASSES], float4 (&areg)[C::A_PER_THREAD]) {
        const int ci = cbeg + c;                                // wave-uniform
        const unsigned tap = p.divCpt.div((unsigned)ci);
        const int cin0 = (ci - (int)tap * cpt) * C::BK;
        unsigned a, b;
        p.divKw.divmod(tap, a, b);
        const int dy = (int)a * p.dh, dx = (int)b * p.dw;
        const bool ok = (unsigned)(hbase + dy) < (unsigned)p.H && (unsigned)(wbase + dx) < (unsigned)p.W;
        const int voff = ok ? (cbase + dy * p.W + dx) << 2 : OOB;
#pragma unroll
        for (int ps = 0; ps < C::B_PASSES; ++ps) {
            float tv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int soff = ((cin0 + ps * C::KG_PER_PASS * 4 + e) * p.HW) << 2;   // scalar
                tv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff, soff, 0));
            }
            breg[ps] = make_float4(tv[0], tv[1], tv[2], tv[3]);
        }
        const int ksoff = (ci * C::BK) << 2;                    // scalar: chunk start along K
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
                *reinterpret_cast<float4 *>(Bb + jl * C::LDK + kg * 4) = breg[ps];
        }
#pragma unroll
        for (int i = 0; i < C::A_PER_THREAD; ++i) {
            const int v = tid + i * C::THREADS;
            if (C::A_VEC % C::THREADS == 0 || v < C::A_VEC)
                *reinterpret_cast<float4 *>(Ab + (v / C::KG) * C::LDK + (v % C::KG) * 4) = areg[i];
        }
    };

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int a_off = (wm * C::WTM + l31) * C::LDK + 4 * lhi;
    const int b_off = (wn * C::WTN + l31) * C::LDK + 4 * lhi;

    // ---- rotated software pipeline -------------------------------------------
    // A chunk's MFMAs are split into a HEAD (k-groups 0..U-2) and a TAIL (group
    // U-1).  Step k, right after the barrier that publishes chunk k in LDS:
    //   1. issue ALL fragment reads of chunk k          (ds_read_b128, into set k&1)
    //   2. run the TAIL of chunk k-1 from the other set  -> covers the LDS latency
    //   3. ds_write chunk k+1 (its loads were issued a whole step ago) so the
    //      writes complete under the head MFMAs, long before the barrier
    //   4. issue the global loads of chunk k+2 (two chunks ahead), run the HEAD of chunk k
    //   5. barrier
    // so the matrix pipe only idles for the barrier itself.
    constexpr int U = C::BK / 8;
    float4 fa0[U][C::TM], fb0[U][C::TN], fa1[U][C::TM], fb1[U][C::TN];

    auto read_frags = [&](int buf, float4 (&af)[U][C::TM], float4 (&bf)[U][C::TN]) {
        const float *Ab = As + buf * C::A_ELEMS + a_off;
        const float *Bb = Bs + buf * C::B_ELEMS + b_off;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int a = 0; a < C::TM; ++a) af[u][a] = *reinterpret_cast<const float4 *>(Ab + a * 32 * C::LDK + 8 * u);
#pragma unroll
            for (int b = 0; b < C::TN; ++b) bf[u][b] = *reinterpret_cast<const float4 *>(Bb + b * 32 * C::LDK + 8 * u);
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
        // One step, parity known at compile time so every register set is statically named.
        // Past-the-end loads/stores are clamped duplicates of the last chunk (branch-free body).
        auto step = [&](auto parity, int k, bool with_tail) {
            constexpr int P = decltype(parity)::value;
            if constexpr (P == 0) {
                read_frags(0, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                if (with_tail) mma(fa1, fb1, U - 1, U);
                store_chunk(1, breg1, areg1);          // early: completes under the head MFMAs
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
    store_tile<C::BM, C::BN, C::TM, C::TN, C::WTM, C::WTN>(p, tc, acc, wm, wn, lane);
}

// OIHW [co][cin][tap] -> [co][tap][cin]
__global__ void __launch_bounds__(256) permute_weights_kernel(const float *w, float *out, unsigned total, int cin,
                                                              int khw, FastDiv divCin, FastDiv divKhw) {
    unsigned stride = gridDim.x * 256;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned r, ci, co, tap;          // i indexes the OUTPUT: ((co*khw)+tap)*cin + ci
        divCin.divmod(i, r, ci);
        divKhw.divmod(r, co, tap);
        out[i] = w[((size_t)co * cin + ci) * khw + tap];
    }
}

#include "conv_q4_kernel.h"
#include "conv_ks_kernel.h"
#include "conv_smallcin_kernel.h"
#include "conv_smallcin_valu_kernel.h"
#include "conv_stem_pool_kernel.h"

// ---- configurations ---------------------------------------------------------
typedef Cfg<128, 128, 16, 2, 2> C128x128;
typedef Cfg<64, 128, 16, 2, 2> C64x128;
typedef Cfg<128, 64, 16, 2, 2> C128x64;
typedef Cfg<64, 64, 16, 2, 2> C64x64;
typedef Cfg<32, 128, 16, 1, 4> C32x128;
typedef Cfg<32, 256, 16, 1, 4> C32x256;
typedef Cfg<64, 256, 16, 2, 2> C64x256;
typedef Cfg<128, 32, 16, 4, 1> C128x32;

typedef TapCfg<128, 128, 16, 2, 2> T128x128x16;
typedef TapCfg<128, 128, 32, 2, 2> T128x128x32;
typedef TapCfg<64, 128, 16, 2, 2> T64x128x16;
typedef TapCfg<64, 128, 32, 2, 2> T64x128x32;
typedef TapCfg<128, 64, 16, 2, 2> T128x64x16;
typedef TapCfg<128, 64, 32, 2, 2> T128x64x32;
typedef TapCfg<64, 64, 16, 2, 2> T64x64x16;
typedef TapCfg<64, 64, 32, 2, 2> T64x64x32;
typedef TapCfg<128, 32, 32, 4, 1> T128x32x32;
typedef TapCfg<32, 128, 32, 1, 4> T32x128x32;
typedef TapCfg<256, 64, 16, 4, 1> T256x64x16;
typedef TapCfg<64, 256, 16, 1, 4> T64x256x16;

typedef QuadCfg<128, 128, 16, 2, 2> Q128x128x16;
typedef QuadCfg<128, 128, 32, 2, 2> Q128x128x32;
typedef QuadCfg<64, 128, 16, 2, 2> Q64x128x16;
typedef QuadCfg<64, 128, 32, 2, 2> Q64x128x32;
typedef QuadCfg<128, 64, 16, 2, 2> Q128x64x16;
typedef QuadCfg<128, 64, 32, 2, 2> Q128x64x32;
typedef QuadCfg<64, 64, 16, 2, 2> Q64x64x16;
typedef QuadCfg<64, 64, 32, 2, 2> Q64x64x32;
typedef QuadCfg<128, 32, 32, 4, 1> Q128x32x32;
typedef QuadCfg<32, 128, 32, 1, 4> Q32x128x32;
typedef QuadCfg<256, 64, 16, 4, 1> Q256x64x16;
typedef QuadCfg<64, 256, 16, 1, 4> Q64x256x16;

struct CfgInfo {
    const char *name;
    // 0: generic (OIHW weights, any shape)   1: tap-major (prepared weights, cin_g % bk == 0)
    // 2: channel-quad activations + k-quad-major filters (conv_q4_kernel.h)
    int tap;
    int bm, bn, bk, lds;
    void (*vec)(const ConvArgs);
    void (*scl)(const ConvArgs);
    void (*reduce)(const ConvArgs, const float *, float *);
    void (*reduce4)(const ConvArgs, const float *, float *);
    bool ks;     // intra-workgroup K split (conv_ks_kernel.h): 1024 threads, 32x32 tile, no split-K plans
    void (*pair)(const ConvArgs, const ConvArgs);    // two convs on one input in one launch (conv_q4_pair_kernel), or null
};

#define CFG_ENTRY(T, nm)                                                                                 \
    {                                                                                                    \
        nm, 0, T::BM, T::BN, T::BK, T::LDS_BYTES, conv_igemm_kernel<T, true>, conv_igemm_kernel<T, false>, \
            reduce_tiles_kernel<T::BM, T::BN, false>, reduce_tiles_kernel<T::BM, T::BN, true>            \
    }
#define TAP_ENTRY(T, nm) \
    { nm, 1, T::BM, T::BN, T::BK, T::LDS_BYTES, conv_tap_kernel<T>, conv_tap_kernel<T>, \
      reduce_tiles_kernel<T::BM, T::BN, false>, reduce_tiles_kernel<T::BM, T::BN, true> }

#define Q4_ENTRY(T, nm) \
    { nm, 2, T::BM, T::BN, T::BK, T::LDS_BYTES, conv_q4_kernel<T>, conv_q4_kernel<T>, \
      reduce_tiles_q4_kernel<T::BM, T::BN>, reduce_tiles_q4_kernel<T::BM, T::BN> }
#define Q4P_ENTRY(T, nm) \
    { nm, 2, T::BM, T::BN, T::BK, T::LDS_BYTES, conv_q4_kernel<T>, conv_q4_kernel<T>, \
      reduce_tiles_q4_kernel<T::BM, T::BN>, reduce_tiles_q4_kernel<T::BM, T::BN>, false, conv_q4_pair_kernel<T> }

const CfgInfo kCfgs[] = {
    CFG_ENTRY(C128x128, "128x128"), CFG_ENTRY(C64x128, "64x128"), CFG_ENTRY(C128x64, "128x64"),
    CFG_ENTRY(C64x64, "64x64"),     CFG_ENTRY(C32x128, "32x128"), CFG_ENTRY(C32x256, "32x256"),
    CFG_ENTRY(C64x256, "64x256"),   CFG_ENTRY(C128x32, "128x32"),
    TAP_ENTRY(T128x128x16, "t128x128x16"), TAP_ENTRY(T128x128x32, "t128x128x32"),
    TAP_ENTRY(T64x128x16, "t64x128x16"),   TAP_ENTRY(T64x128x32, "t64x128x32"),
    TAP_ENTRY(T128x64x16, "t128x64x16"),   TAP_ENTRY(T128x64x32, "t128x64x32"),
    TAP_ENTRY(T64x64x16, "t64x64x16"),     TAP_ENTRY(T64x64x32, "t64x64x32"),
    TAP_ENTRY(T128x32x32, "t128x32x32"),   TAP_ENTRY(T32x128x32, "t32x128x32"),
    TAP_ENTRY(T256x64x16, "t256x64x16"),   TAP_ENTRY(T64x256x16, "t64x256x16"),
    Q4_ENTRY(Q128x128x16, "q128x128x16"),  Q4_ENTRY(Q128x128x32, "q128x128x32"),
    Q4P_ENTRY(Q64x128x16, "q64x128x16"),    Q4P_ENTRY(Q64x128x32, "q64x128x32"),
    Q4_ENTRY(Q128x64x16, "q128x64x16"),    Q4P_ENTRY(Q128x64x32, "q128x64x32"),
    Q4P_ENTRY(Q64x64x16, "q64x64x16"),      Q4P_ENTRY(Q64x64x32, "q64x64x32"),
    Q4P_ENTRY(Q128x32x32, "q128x32x32"),    Q4_ENTRY(Q32x128x32, "q32x128x32"),
    Q4_ENTRY(Q256x64x16, "q256x64x16"),    Q4_ENTRY(Q64x256x16, "q64x256x16"),
#define KS_ENTRY(TM, TN, nm) \
    { nm, 2, KsCfg<TM, TN>::BM, KsCfg<TM, TN>::BN, KsCfg<TM, TN>::BK, KsCfg<TM, TN>::LDS_BYTES, conv_ks_kernel<KsCfg<TM, TN> >, \
      conv_ks_kernel<KsCfg<TM, TN> >, nullptr, nullptr, true }
    KS_ENTRY(1, 1, "k32x32x8"), KS_ENTRY(2, 1, "k64x32x8"), KS_ENTRY(1, 2, "k32x64x8"), KS_ENTRY(2, 2, "k64x64x8"),
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr int kLdsPerCu = 160 * 1024;

// weight layouts: 0 OIHW (generic kernel), 1 tap-major, 2 Q4 (activations AND filter in quad form)
bool cfg_applies(const CfgInfo &ci, int layout, int cin_g, int q_pad = 1 << 20) {
    if (ci.ks && (layout != 2 || (getenv("PLANER_HIP_KS") && atoi(getenv("PLANER_HIP_KS")) == 0))) return false;
    if (layout == 6) layout = 2;            // row-packed input: the channel-quad kernel with another gather
    if (ci.tap != layout) return false;
    return ci.tap != 1 || cin_g % ci.bk == 0;
}

// How one conv is run.  Tiles [0, t1) take the data-parallel pass with the
// fused epilogue; tiles [t1, T) take a split-K pass (s2 slices of K, compact
// per-tile slabs) followed by the tile reduce.  `occ` > 0 pins the number of
// co-resident workgroups per CU (by padding the LDS request) so a pass of
// exactly occ*CUs tiles lands as occ tiles on EVERY CU -- at batch 32 the
// ResNet layers have only 200-800 tiles, and an uneven 3-vs-4 tiles per CU
// split costs more than anything inside the kernel.
struct Plan {
    int cfg, t1, s2, occ;
};

void (*kernel_of(const CfgInfo &ci, bool avec))(const ConvArgs) { return (ci.tap || avec) ? ci.vec : ci.scl; }

}  // namespace

int plhip::ensure_lds_attr(const void *kern, int bytes) {
    static std::mutex mu;
    static std::map<const void *, int> done;
    std::lock_guard<std::mutex> lk(mu);
    auto it = done.find(kern);
    if (it != done.end() && it->second >= bytes) return PL_OK;
    PL_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[kern] = bytes;
    return PL_OK;
}

namespace {
using plhip::ensure_lds_attr;

// K slices a request for `splits` really yields (slices are whole BK-chunks)
int effective_splits(const CfgInfo &ci, int K, int splits) {
    if (splits <= 1) return 1;
    const int chunks = (K + ci.bk - 1) / ci.bk;
    const int cps = (chunks + splits - 1) / splits;
    return (chunks + cps - 1) / cps;
}

int launch_pass(pl_ctx *ctx, ConvArgs a, const CfgInfo &ci, bool avec, int tile_offset, int tile_count, int splits,
                int occ, float *out, int *used_splits) {
    *used_splits = 0;
    if (tile_count <= 0) return PL_OK;
    if (ci.tap == 2) {
        const int kg = ci.bk / 4;
        const int total_chunks = (a.Qtot + kg - 1) / kg;
        const int cps = (total_chunks + splits - 1) / splits;
        splits = (total_chunks + cps - 1) / cps;
        a.k_per_split = cps;
        a.uni = a.rp_rq ? 0 : a.cqg % kg == 0;
        a.divCpt = FastDiv(a.rp_rq ? a.rp_rq : a.cqg);
    } else if (ci.tap) {
        const int total_chunks = a.K / ci.bk;
        const int cps = (total_chunks + splits - 1) / splits;
        splits = (total_chunks + cps - 1) / cps;
        a.k_per_split = cps;
        a.divCpt = FastDiv(a.cin_g / ci.bk);
    } else {
        const int kps = ((a.K + splits - 1) / splits + ci.bk - 1) / ci.bk * ci.bk;
        splits = (a.K + kps - 1) / kps;
        a.k_per_split = kps;
    }
    a.splits = splits;
    a.tile_offset = tile_offset;
    a.tile_count = tile_count;
    a.y = out;
    if (ci.ks) {
        if (splits != 1) {
            pl_set_error("conv: the K-split-inside-the-workgroup kernel takes no split-K plan");
            return PL_EINVAL;
        }
        int rc = ensure_lds_attr((const void *)ci.vec, ci.lds);
        if (rc != PL_OK) return rc;
        hipLaunchKernelGGL(ci.vec, dim3((unsigned)tile_count), dim3((unsigned)(16 * 32 * 32 / (ci.bm * ci.bn) * 64)), ci.lds, ctx->stream, a);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            pl_set_error("conv launch (%s): %s", ci.name, hipGetErrorString(le));
            return PL_EHIP;
        }
        *used_splits = 1;
        return PL_OK;
    }
    int lds = ci.lds;
    if (occ > 0) {
        const int want = (kLdsPerCu / occ) & ~255;          // occ blocks fit, occ+1 do not
        if (want > lds && kLdsPerCu / (occ + 1) < want) lds = want;
    }
    auto kern = kernel_of(ci, avec);
    if (lds > 48 * 1024) {
        int rc = ensure_lds_attr((const void *)kern, lds);
        if (rc != PL_OK) return rc;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tile_count, (unsigned)splits), dim3(256), lds, ctx->stream, a);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        pl_set_error("conv launch (%s): %s", ci.name, hipGetErrorString(le));
        return PL_EHIP;
    }
    *used_splits = splits;  // K slices actually used (K may be too short for the request)
    return PL_OK;
}

int run_plan(pl_ctx *ctx, const ConvArgs &a0, const Plan &pl, bool avec, float *y) {
    const CfgInfo &ci = kCfgs[pl.cfg];
    ConvArgs a = a0;
    a.mtiles = (a.cout_g + ci.bm - 1) / ci.bm;
    a.ntiles = (a.cols + ci.bn - 1) / ci.bn;
    a.tiles = a.mtiles * a.ntiles;
    a.divMt = FastDiv(a.mtiles);
    const int T = a.tiles * a.groups;
    const int s2 = effective_splits(ci, a.K, pl.s2);
    const int t1 = s2 <= 1 ? T : (pl.t1 < 0 ? 0 : (pl.t1 > T ? T : pl.t1));
    {
        char buf[96];
        snprintf(buf, sizeof buf, "%s tiles=%d dp=%d split=%d occ=%d", ci.name, T, t1, s2, pl.occ);
        ctx->last_plan = buf;
        // what the matrix cores execute: whole tiles and whole K chunks
        const int kq = ci.tap == 2 ? a.Qtot * 4 : a.K;
        ctx->last_gemm[0] = a.groups;
        ctx->last_gemm[1] = (long long)a.mtiles * ci.bm;
        ctx->last_gemm[2] = (long long)a.ntiles * ci.bn;
        ctx->last_gemm[3] = (long long)(kq + ci.bk - 1) / ci.bk * ci.bk;
    }
    // column-block ownership per XCD (set by the caller as a REQUEST = tile columns that must not straddle XCDs): honoured when
    // the plan is one unsplit pass of a channel-quad tile kernel and the column tiles split evenly over the 8 XCDs
    if (a.xcd_cols > 0) {
        const int want_cols = a.xcd_cols;            // columns per XCD
        a.xcd_cols = (t1 == T && ci.tap == 2 && !ci.ks && a.ntiles % 8 == 0 && (a.ntiles / 8) * ci.bn == want_cols) ? a.ntiles / 8 : 0;
        if (a.xcd_cols) ctx->last_plan += " xcdcols";
    }
    int used = 0;
    int rc = launch_pass(ctx, a, ci, avec, 0, t1, 1, pl.occ, y, &used);
    if (rc != PL_OK) return rc;
    const int tail = T - t1;
    if (tail <= 0) return PL_OK;
    float *ws = nullptr;
    const size_t slab_elems = (size_t)tail * ci.bm * ci.bn;
    rc = pl_alloc(ctx, (size_t)s2 * slab_elems * sizeof(float), (void **)&ws);
    if (rc != PL_OK) return rc;
    rc = launch_pass(ctx, a, ci, avec, t1, tail, s2, 0, ws, &used);
    if (rc == PL_OK && used != s2) {
        pl_set_error("conv: split bookkeeping mismatch (%d vs %d)", used, s2);
        rc = PL_EINVAL;
    } else if (rc == PL_OK) {
        ConvArgs r = a;
        r.splits = used;
        r.tile_offset = t1;
        r.tile_count = tail;
        const bool vec4 = ci.tap == 2 || (a.HoWo % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0 &&
                                          (!a.ep.res || (reinterpret_cast<uintptr_t>(a.ep.res) & 15u) == 0));
        const int rows = ci.tap == 2 ? REDUCE_Q4_QUADS * 4 : vec4 ? std::min(ci.bm, REDUCE_ROWS * 4) : REDUCE_ROWS;
        hipLaunchKernelGGL(vec4 ? ci.reduce4 : ci.reduce, dim3((unsigned)tail, (unsigned)((ci.bm + rows - 1) / rows)),
                           dim3(256), 0, ctx->stream, r, (const float *)ws, y);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            pl_set_error("conv tile reduce: %s", hipGetErrorString(le));
            rc = PL_EHIP;
        }
    }
    pl_free(ctx, ws);  // stream-ordered: safe to recycle after the enqueue
    return rc;
}

// Static choice (while capturing, or with autotune off).
Plan choose_plan(pl_ctx *ctx, int layout, const ConvArgs &a) {
    const double cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    double best = 1e300;
    Plan pl{-1, 0, 1, 0};
    for (int c = 0; c < kNumCfgs; ++c) {
        const CfgInfo &ci = kCfgs[c];
        if (ci.ks || !cfg_applies(ci, layout, a.cin_g)) continue;
        const double mt = (a.cout_g + ci.bm - 1) / ci.bm, nt = (a.cols + ci.bn - 1) / ci.bn;
        const double tiles = mt * nt * a.groups;
        const double kch = std::ceil((double)a.K / ci.bk);
        const double per_chunk = (ci.bk / 2) * (ci.bm / 32) * (ci.bn / 32) / 4.0 + 2.0;   // MFMA slots per wave
        const int maxocc = std::min(8, kLdsPerCu / ci.lds);
        for (int s = 1; s <= 8; ++s) {
            if (s > 1 && kch / s < 6) break;
            const double blocks = tiles * s;
            const double per_cu = std::ceil(blocks / cus);
            const double rounds = std::ceil(per_cu / maxocc);
            double cost = per_cu * (std::ceil(kch / s) * per_chunk + 40.0) + rounds * 20.0;
            if (s > 1) cost += 50.0 + tiles * ci.bm * ci.bn * (s + 1) / (cus * 256.0);
            if (cost < best) {
                best = cost;
                pl = Plan{c, s > 1 ? 0 : (int)tiles, s, 0};
            }
        }
    }
    return pl;
}

struct TuneKey {
    int v[18];
    bool operator<(const TuneKey &o) const { return memcmp(v, o.v, sizeof v) < 0; }
};
std::mutex g_tune_mu;
// keyed by DEVICE (not context): side-stream contexts of one GPU share what was tuned
std::map<std::pair<int, TuneKey>, Plan> g_tune;

int max_split_env() {
    const char *ms_env = getenv("PLANER_CONV_MAX_SPLIT");
    return ms_env ? atoi(ms_env) : 1 << 20;
}

float time_plan(pl_ctx *ctx, const ConvArgs &a, const Plan &pl, bool avec, float *y, hipEvent_t e0, hipEvent_t e1,
                int reps) {
    float best = 1e30f;
    for (int rep = 0; rep <= reps; ++rep) {
        (void)hipEventRecord(e0, ctx->stream);
        if (run_plan(ctx, a, pl, avec, y) != PL_OK) return 1e30f;
        (void)hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;   // first run warms caches / clocks
    }
    return best;
}

// First sight of a conv shape: measure candidate plans (MIOpen-find style) and
// remember the winner for this context.
Plan tune_plan(pl_ctx *ctx, int layout, const ConvArgs &a, bool avec, float *y, Plan fallback) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return fallback;
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fallback;
    }
    const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    const double work = (double)a.cout_g * a.cols * a.groups;
    struct Cand {
        float ms;
        Plan pl;
    };
    std::vector<Cand> stage1;
    const int splits[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16};
    for (int c = 0; c < kNumCfgs; ++c) {
        const CfgInfo &ci = kCfgs[c];
        if (!cfg_applies(ci, layout, a.cin_g, a.Qpad)) continue;
        const int T = ((a.cout_g + ci.bm - 1) / ci.bm) * ((a.cols + ci.bn - 1) / ci.bn) * a.groups;
        if ((double)T * ci.bm * ci.bn > 2.5 * work + 1e5) continue;               // mostly padding
        Plan dp{c, T, 1, 0};
        stage1.push_back({time_plan(ctx, a, dp, avec, y, e0, e1, 2), dp});
        if (ci.ks) continue;                                                       // intra-workgroup split: no split-K variants
        const int chunks = (a.K + ci.bk - 1) / ci.bk;
        const char *ms_env = getenv("PLANER_CONV_MAX_SPLIT");     // experiments: cap split-K
        const int max_split = ms_env ? atoi(ms_env) : 1 << 20;
        for (int s : splits) {
            if (s > max_split) break;
            if (chunks / s < 3 || (double)T * s > 8.0 * cus * 4 || T > 6 * cus) break;
            Plan sk{c, 0, s, 0};
            stage1.push_back({time_plan(ctx, a, sk, avec, y, e0, e1, 2), sk});
        }
    }
    if (stage1.empty()) {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return fallback;
    }
    std::sort(stage1.begin(), stage1.end(), [](const Cand &x, const Cand &y2) { return x.ms < y2.ms; });
    Cand best = stage1[0];
    // stage 2: for the most promising tile shapes, an exactly balanced
    // data-parallel prefix (occ tiles on every CU) plus a split-K tail
    std::vector<int> tried;
    for (size_t i = 0; i < stage1.size() && tried.size() < 4; ++i) {
        const int c = stage1[i].pl.cfg;
        if (std::find(tried.begin(), tried.end(), c) != tried.end()) continue;
        tried.push_back(c);
        const CfgInfo &ci = kCfgs[c];
        if (ci.ks) continue;
        const int T = ((a.cout_g + ci.bm - 1) / ci.bm) * ((a.cols + ci.bn - 1) / ci.bn) * a.groups;
        const int chunks = (a.K + ci.bk - 1) / ci.bk;
        const int maxocc = std::min(8, kLdsPerCu / ci.lds);
        for (int occ = 1; occ <= maxocc; ++occ) {
            const int wave = occ * cus;
            if (wave > T) break;
            const int t1 = T / wave * wave;
            if (T / wave > 1 && occ < maxocc / 2) continue;   // several rounds: prefer fuller CUs
            const int tail = T - t1;
            if (tail == 0) {
                Plan pl{c, T, 1, occ};
                float ms = time_plan(ctx, a, pl, avec, y, e0, e1, 2);
                if (ms < best.ms) best = {ms, pl};
                continue;
            }
            for (int s : splits) {
                if (s > max_split_env()) break;
                if (chunks / s < 2) break;
                if ((double)tail * s < 0.4 * cus && s < 16) continue;          // tail would leave CUs idle
                if ((double)tail * s > 6.0 * cus) break;
                Plan pl{c, t1, s, occ};
                float ms = time_plan(ctx, a, pl, avec, y, e0, e1, 2);
                if (ms < best.ms) best = {ms, pl};
            }
        }
    }
    // confirm the winner against the runner-up with more repetitions (noise)
    if (stage1.size() > 1) {
        float m0 = time_plan(ctx, a, best.pl, avec, y, e0, e1, 5);
        float m1 = time_plan(ctx, a, stage1[0].pl, avec, y, e0, e1, 5);
        if (m1 < m0) best = {m1, stage1[0].pl};
        else best.ms = m0;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("PLANER_CONV_TUNE_LOG"))
        fprintf(stderr, "[planer_hip] conv N%d C%d %dx%d -> %d k%dx%d s%d g%d layout %d: %s t1 %d split %d occ %d (%.3f ms, %.1f TFLOP/s)\n",
                a.N, a.Cin, a.H, a.W, a.Cout, a.kh, a.kw, a.sh, a.groups, layout, kCfgs[best.pl.cfg].name, best.pl.t1,
                best.pl.s2, best.pl.occ, best.ms, 2.0 * a.cout_g * a.groups * (double)a.cols * a.K / best.ms / 1e9);
    return best.pl;
}

// pixels per row of the zero-padded NHWC image of the row-packed path: the padded width, and room for
// the last window's quads to overrun (a filter row is read as whole quads: 4*RQ floats)
inline int rowpack_row_pixels(int W, int pl, int Wo, int sw, int kw, int Cin) {
    const int rq = (kw * Cin + 3) / 4;
    return std::max(W + 2 * pl, (Wo - 1) * sw + (4 * rq + Cin - 1) / Cin + 1);
}

}  // namespace

int plhip::conv_launch(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w, int Cout, int kh, int kw,
                       const float *bias, float *y, int sh, int sw, int dh, int dw, int pt, int pl, int pb, int pr,
                       int group, const float *scale, const float *shift, const float *res, int act, double alpha,
                       int layout) {
    PL_REQUIRE(ctx && x && w && y, PL_EINVAL, "conv2d: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && kh > 0 && kw > 0, PL_EINVAL, "conv2d: bad shape");
    PL_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && pt >= 0 && pl >= 0 && group > 0, PL_EINVAL, "conv2d: bad parameter");
    // util.pad only honours pads[0]/pads[1] (util.py:8): anything else is undefined there
    PL_REQUIRE(pt == pb && pl == pr, PL_EUNSUPPORTED, "asymmetric pads are undefined in the reference (util.py:8)");
    PL_REQUIRE(Cin % group == 0 && Cout % group == 0, PL_EUNSUPPORTED, "group must divide Cin and Cout");
    PL_REQUIRE(act >= 0 && (act & 15) <= 2 && (act & ~31) == 0, PL_EINVAL, "conv2d: bad activation code");
    PL_REQUIRE((layout >= 0 && layout <= 3) || layout == 6, PL_EINVAL, "conv2d: bad weight layout");
    if (layout == 3) {
        PL_REQUIRE(kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == 1 && dw == 1 && pt == 1 && pl == 1 && pb == 1 &&
                       pr == 1 && group == 1, PL_EINVAL, "winograd filters serve 3x3 / stride 1 / pad 1 / group 1 only");
        if (N == 0) return PL_OK;
        CtxGuard guard(ctx);
        return plhip::winograd_launch(ctx, x, N, Cin, H, W, w, Cout, bias, y, scale, shift, res, act, alpha);
    }
    const int Ho = (H + pt + pb - (kh - 1) * dh - 1 + sh) / sh;  // util.py:25
    const int Wo = (W + pl + pr - (kw - 1) * dw - 1 + sw) / sw;  // util.py:26
    PL_REQUIRE(Ho > 0 && Wo > 0, PL_EINVAL, "conv2d: empty output (%d x %d)", Ho, Wo);
    PL_REQUIRE(H + 2 * pt < 16384 && W + 2 * pl < 16384 && kh * dh < 16384 && kw * dw < 16384, PL_EUNSUPPORTED,
               "conv2d: spatial extent above 16383");
    if (N == 0) return PL_OK;
    size_t out_elems = (size_t)N * Cout * Ho * Wo, in_elems = (size_t)N * Cin * H * W;
    size_t w_elems = (size_t)Cout * (Cin / group) * kh * kw;
    const int cqg = (Cin / group + 3) / 4, q_tot = kh * kw * cqg, q_pad = (q_tot + 7) / 8 * 8;
    // layout 6: x is the zero-padded NHWC image [N][H+2pt][Wp][Cin] made by rowpack_launch below
    const int rp_rq = layout == 6 ? (kw * Cin + 3) / 4 : 0;
    const int rp_wp = layout == 6 ? rowpack_row_pixels(W, pl, Wo, sw, kw, Cin) : 0;
    if (layout == 6) {
        PL_REQUIRE(group == 1 && dh == 1 && dw == 1 && Cin < 4, PL_EUNSUPPORTED, "conv2d (row-packed): group 1, dilation 1, Cin < 4");
        in_elems = (size_t)N * (H + 2 * pt) * rp_wp * Cin + 8;
        out_elems = (size_t)N * ((Cout + 3) / 4) * 4 * Ho * Wo;
        w_elems = (size_t)((kh * rp_rq + 7) / 8 * 8) * Cout * 4;
    }
    if (layout == 2) {
        PL_REQUIRE(group == 1 || ((Cin / group) % 4 == 0 && (Cout / group) % 4 == 0), PL_EUNSUPPORTED,
                   "conv2d (Q4): grouped convs need Cin/group and Cout/group to be multiples of 4");
        in_elems = (size_t)N * ((Cin + 3) / 4) * 4 * H * W;
        out_elems = (size_t)N * ((Cout + 3) / 4) * 4 * Ho * Wo;
        w_elems = (size_t)group * q_pad * (Cout / group) * 4;
    }
    PL_REQUIRE(out_elems < (1ull << 31) && in_elems < (1ull << 29) && w_elems < (1ull << 29),
               PL_EUNSUPPORTED, "conv2d: input/filter above 2 GiB or output above 2^31 elements");
    PL_REQUIRE(layout != 1 || (Cin / group) % 16 == 0, PL_EINVAL, "conv2d: tap-major weights need Cin/group %% 16 == 0");
    CtxGuard guard(ctx);

    // 1..4 input channels, 3x3 / stride 1, plain NCHW conv (+bias): the store-stream kernel of
    // conv_smallcin_kernel.h (BASELINE config 2).  40 us on (8,3,224,224)->64 against 47 us for the generic kernel
    // below (0.34 against 0.29 of 8 TB/s) once a workgroup takes four tiles; on small outputs its staging pass makes it
    // slower than the generic kernel, so it takes over from ~3 tiles per CU upwards.  PLANER_HIP_SMALLCIN=0 / 1 forces.
    const char *sc_env = getenv("PLANER_HIP_SMALLCIN");
    const int sc_cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    const bool sc_big = (long long)N * ((Ho * Wo + SC_PIX - 1) / SC_PIX) * ((Cout + SC_CO - 1) / SC_CO) >= 3LL * sc_cus;
    // ... and, where output rows are whole pixel quads, the vector-ALU kernel of conv_smallcin_valu_kernel.h: no matrix
    // instruction, 16-byte stores of contiguous 1 KB runs straight from the accumulators.  PLANER_HIP_SMALLCIN_VALU=0 / 1
    // forces; PLANER_HIP_SCV_CPB = output channels per workgroup (8..64).
    {
        const char *scv_env = getenv("PLANER_HIP_SMALLCIN_VALU");
        const long long quads = (long long)Ho * Wo / 4;
        if (layout == 0 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == 1 && dw == 1 && group == 1 && Cin <= 4 && pt == pl &&
            pt <= 1 && !scale && !shift && !res && act == PL_ACT_NONE && ctx->conv_cfg < 0 && out_elems < (1ull << 29) && Wo % 4 == 0 &&
            (scv_env ? atoi(scv_env) != 0 : (sc_env ? atoi(sc_env) != 0 : sc_big))) {
            const char *cpb_env = getenv("PLANER_HIP_SCV_CPB");
            int cpb = cpb_env ? atoi(cpb_env) : 0;
            if (cpb <= 0) {
                // as many channels per workgroup as still leave 12 waves per CU -- one resident round at 3-4 waves per SIMD (the input
                // window is re-read per channel block; measured on config 2: 64 / 32 / 16 / 8 channels = 29.2 / 28.7 / 29.2 / 32.3 us)
                cpb = 64;
                while (cpb > 8 && (long long)N * ((quads + 255) / 256) * ((Cout + cpb - 1) / cpb) * 4 < 12LL * sc_cus) cpb /= 2;
            }
            cpb = std::max(8, std::min(SCV_MAXC, (cpb + 7) / 8 * 8));
            SmallCinValuArgs va;
            va.x = x; va.w = w; va.bias = bias; va.y = y;
            va.N = N; va.H = H; va.W = W; va.Cout = Cout; va.Ho = Ho; va.Wo = Wo;
            va.quads = (int)quads; va.cpb = cpb;
            va.x_bytes = (unsigned)(in_elems * 4); va.y_bytes = (unsigned)(out_elems * 4);
            va.divQw = FastDiv(Wo / 4);
            void (*kern)(const SmallCinValuArgs) = nullptr;
            switch (Cin * 2 + pt) {
            case 2: kern = conv_smallcin_valu_kernel<1, 0>; break;
            case 3: kern = conv_smallcin_valu_kernel<1, 1>; break;
            case 4: kern = conv_smallcin_valu_kernel<2, 0>; break;
            case 5: kern = conv_smallcin_valu_kernel<2, 1>; break;
            case 6: kern = conv_smallcin_valu_kernel<3, 0>; break;
            case 7: kern = conv_smallcin_valu_kernel<3, 1>; break;
            case 8: kern = conv_smallcin_valu_kernel<4, 0>; break;
            default: kern = conv_smallcin_valu_kernel<4, 1>; break;
            }
            // measurement builds of the 3-channel / pad-1 instantiation (BASELINE config 2): PLANER_HIP_EXPERIMENT=scv_knock=<mask>
            // (1 no input loads, 2 no FMAs, 4 no stores, 8 no per-tap filter reads; tools/scv_probe.py)
            if (Cin == 3 && pt == 1) {
                switch (pl_experiment("scv_knock", 0)) {
                case 1: kern = conv_smallcin_valu_kernel<3, 1, 1>; break;
                case 2: kern = conv_smallcin_valu_kernel<3, 1, 2>; break;
                case 3: kern = conv_smallcin_valu_kernel<3, 1, 3>; break;
                case 4: kern = conv_smallcin_valu_kernel<3, 1, 4>; break;
                case 5: kern = conv_smallcin_valu_kernel<3, 1, 5>; break;
                case 13: kern = conv_smallcin_valu_kernel<3, 1, 13>; break;
                default: break;
                }
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)((quads + 255) / 256), (unsigned)((Cout + cpb - 1) / cpb), (unsigned)N), dim3(256), 0,
                               ctx->stream, va);
            PL_LAUNCH_CHECK();
            ctx->last_plan = "smallcin3x3valu 4px x " + std::to_string(cpb) + "co";
            // (no MFMA: the extents describe the FMAs issued, padded channels included)
            ctx->last_gemm[0] = 1; ctx->last_gemm[1] = (long long)((Cout + cpb - 1) / cpb) * cpb;
            ctx->last_gemm[2] = (long long)N * ((quads + 255) / 256) * 1024; ctx->last_gemm[3] = Cin * 9;
            return PL_OK;
        }
    }
    if (layout == 0 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == 1 && dw == 1 && group == 1 && Cin <= 4 &&
        pt == pl && pt <= 1 && !scale && !shift && !res && act == PL_ACT_NONE && ctx->conv_cfg < 0 && out_elems < (1ull << 29) &&
        (sc_env ? atoi(sc_env) != 0 : sc_big)) {
        // 256-pixel tiles per workgroup: as many as keep ~1.5 workgroups per CU in flight and the staged rows in 64 KB
        const int cus = sc_cus;
        const int tiles_img = (Ho * Wo + SC_PIX - 1) / SC_PIX, co_blocks = (Cout + SC_CO - 1) / SC_CO;
        int tpw = std::max(1, std::min(8, (int)((long long)tiles_img * co_blocks * N / (cus * 3 / 2))));
        // 16-byte stores through a wave-private LDS exchange (conv_smallcin_nchw_kernel<true>): PLANER_HIP_SMALLCIN_WIDE=0 / 1
        const char *wide_env = getenv("PLANER_HIP_SMALLCIN_WIDE");        // read per call: tests switch it
        const bool wide = (Ho * Wo) % 4 == 0 && (wide_env ? atoi(wide_env) != 0 : true);
        auto lds_for = [&](int t) {
            const int rows_max = (SC_PIX * t + Wo - 1) / Wo + 3;
            const size_t base = ((size_t)SC_MAXK * SC_CO + SC_CO + 4 + (size_t)Cin * rows_max * (W + 2) + 3) / 4 * 4;
            return (base + (wide ? 4 * 256 : 0)) * sizeof(float);
        };
        while (tpw > 1 && lds_for(tpw) > 64 * 1024) --tpw;
        const size_t lds = lds_for(tpw);
        if (lds <= 64 * 1024) {
            SmallCinArgs sa;
            sa.x = x; sa.w = w; sa.bias = bias; sa.y = y;
            sa.N = N; sa.Cin = Cin; sa.H = H; sa.W = W; sa.Cout = Cout; sa.Ho = Ho; sa.Wo = Wo; sa.pad = pt;
            sa.x_bytes = (int)(in_elems * 4);
            sa.y_bytes = (int)(out_elems * 4);
            sa.HoWo = Ho * Wo; sa.Wp = W + 2; sa.K = Cin * 9; sa.steps = (sa.K + 1) / 2; sa.tpw = tpw;
            sa.divWo = FastDiv(Wo); sa.divK = FastDiv(sa.K);
            sa.xb_off = (int)(lds / sizeof(float)) - 4 * 256;
            void (*kern)(const SmallCinArgs) = wide ? conv_smallcin_nchw_kernel<true> : conv_smallcin_nchw_kernel<false>;
            int rc = ensure_lds_attr((const void *)kern, 64 * 1024);
            if (rc != PL_OK) return rc;
            hipLaunchKernelGGL(kern, dim3((unsigned)((tiles_img + tpw - 1) / tpw), (unsigned)co_blocks, (unsigned)N),
                               dim3(256), lds, ctx->stream, sa);
            PL_LAUNCH_CHECK();
            ctx->last_plan = std::string(wide ? "smallcin3x3w " : "smallcin3x3 ") + std::to_string(tpw) + "x256px x 64co";
            ctx->last_gemm[0] = 1; ctx->last_gemm[1] = (long long)co_blocks * SC_CO;
            ctx->last_gemm[2] = (long long)N * tiles_img * SC_PIX; ctx->last_gemm[3] = (long long)sa.steps * 2;
            return PL_OK;
        }
    }

    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.w = w; a.y = y;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = Ho; a.Wo = Wo;
    a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw; a.pt = pt; a.pl = pl;
    a.groups = group; a.cin_g = Cin / group; a.cout_g = Cout / group;
    a.K = layout == 2 ? q_tot * 4 : a.cin_g * kh * kw;
    a.cqg = cqg; a.Cq = (Cin + 3) / 4; a.Coq = (Cout + 3) / 4; a.Qtot = q_tot; a.Qpad = q_pad;
    if (layout == 6) {
        a.rp_rq = rp_rq;
        a.Qtot = kh * rp_rq; a.Qpad = (a.Qtot + 7) / 8 * 8; a.K = a.Qtot * 4;
        a.H = H + 2 * pt; a.W = rp_wp;          // padded extents: what the gather indexes
    }
    a.cols = N * Ho * Wo;
    a.HoWo = Ho * Wo; a.HW = H * W;
    a.y_bytes = ((layout == 2 || layout == 6) && (size_t)N * a.Coq * Ho * Wo * 16 < (1ull << 31)) ? (int)((size_t)N * a.Coq * Ho * Wo * 16) : 0;
    a.x_bytes = (int)(in_elems * 4); a.w_bytes = (int)(w_elems * 4);
    a.divKhw = FastDiv(kh * kw); a.divKw = FastDiv(kw);
    a.divHoWo = FastDiv(a.HoWo); a.divWo = FastDiv(Wo);
    a.divMt = FastDiv(1); a.divCpt = FastDiv(1);
    a.ep = make_epilogue(bias, scale, shift, res, act, alpha);
    a.xcd_cols = ctx->xcd_cols_request;
    const bool avec = (a.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15u) == 0);

    // forced configuration (tests / tuning tools): split > 1 means split-K over all tiles
    if (ctx->conv_cfg >= 0 && ctx->conv_cfg < kNumCfgs && cfg_applies(kCfgs[ctx->conv_cfg], layout, a.cin_g, a.Qpad)) {
        const int s = ctx->conv_split_k > 0 ? ctx->conv_split_k : 1;
        Plan pl{ctx->conv_cfg, s > 1 ? ctx->conv_t1 : (1 << 30), s, ctx->conv_occ};
        return run_plan(ctx, a, pl, avec, y);
    }
    TuneKey key = {{layout, N, Cin, H, W, Cout, kh, kw, sh, sw, dh, dw, pt, pl, group, res != nullptr,
                    (scale != nullptr) * 2 + (bias != nullptr), act}};
    Plan plan{-1, 0, 1, 0};
    bool have = false;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tune.find({ctx->device, key});
        if (it != g_tune.end()) {
            plan = it->second;
            have = true;
        }
    }
    if (!have) {
        plan = choose_plan(ctx, layout, a);
        PL_REQUIRE(plan.cfg >= 0, PL_EUNSUPPORTED, "conv2d: no kernel configuration applies");
        const bool tune = ctx->autotune && !ctx->capturing && res != y && x != y;
        if (tune) {
            plan = tune_plan(ctx, layout, a, avec, y, plan);
            ++ctx->tune_misses;
        }
        if (tune || !ctx->autotune) {
            std::lock_guard<std::mutex> lk(g_tune_mu);
            g_tune[{ctx->device, key}] = plan;
        }
    }
    return run_plan(ctx, a, plan, avec, y);
}

namespace {
using plhip::conv_launch;

// ---- two channel-quad convs on one input in one launch (conv_q4_pair_kernel) ----
struct PairConv {
    const float *wq, *bias, *scale, *shift;
    float *y;
    int Cout, kh, kw, sh, sw, pt, pl, act;
    double alpha;
};

// ConvArgs of an unsplit channel-quad conv (what conv_launch builds for layout 2, group 1, dilation 1)
int pair_conv_args(ConvArgs &a, const float *x, int N, int Cin, int H, int W, const PairConv &c) {
    const int Ho = (H + 2 * c.pt - (c.kh - 1) - 1 + c.sh) / c.sh, Wo = (W + 2 * c.pl - (c.kw - 1) - 1 + c.sw) / c.sw;
    PL_REQUIRE(Ho > 0 && Wo > 0, PL_EINVAL, "conv pair: empty output (%d x %d)", Ho, Wo);
    const int cqg = (Cin + 3) / 4, q_tot = c.kh * c.kw * cqg, q_pad = (q_tot + 7) / 8 * 8;
    const size_t in_elems = (size_t)N * cqg * 4 * H * W, out_elems = (size_t)N * ((c.Cout + 3) / 4) * 4 * Ho * Wo;
    const size_t w_elems = (size_t)q_pad * c.Cout * 4;
    PL_REQUIRE(out_elems < (1ull << 29) && in_elems < (1ull << 29) && w_elems < (1ull << 29), PL_EUNSUPPORTED,
               "conv pair: tensor above 2 GiB");
    memset(&a, 0, sizeof a);
    a.x = x; a.w = c.wq; a.y = c.y;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = c.Cout; a.Ho = Ho; a.Wo = Wo;
    a.kh = c.kh; a.kw = c.kw; a.sh = c.sh; a.sw = c.sw; a.dh = a.dw = 1; a.pt = c.pt; a.pl = c.pl;
    a.groups = 1; a.cin_g = Cin; a.cout_g = c.Cout;
    a.K = q_tot * 4;
    a.cqg = cqg; a.Cq = cqg; a.Coq = (c.Cout + 3) / 4; a.Qtot = q_tot; a.Qpad = q_pad;
    a.cols = N * Ho * Wo;
    a.HoWo = Ho * Wo; a.HW = H * W;
    a.y_bytes = (int)(out_elems * 4); a.x_bytes = (int)(in_elems * 4); a.w_bytes = (int)(w_elems * 4);
    a.divKhw = FastDiv(c.kh * c.kw); a.divKw = FastDiv(c.kw);
    a.divHoWo = FastDiv(a.HoWo); a.divWo = FastDiv(Wo);
    a.ep = make_epilogue(c.bias, c.scale, c.shift, nullptr, c.act, c.alpha);
    return PL_OK;
}

// tile the conv for configuration ci, unsplit, whole conv in this launch
void pair_tile(ConvArgs &a, const CfgInfo &ci) {
    a.mtiles = (a.cout_g + ci.bm - 1) / ci.bm;
    a.ntiles = (a.cols + ci.bn - 1) / ci.bn;
    a.tiles = a.mtiles * a.ntiles;
    a.divMt = FastDiv(a.mtiles);
    const int kg = ci.bk / 4;
    a.k_per_split = (a.Qtot + kg - 1) / kg;
    a.splits = 1; a.tile_offset = 0; a.tile_count = a.tiles;
    a.uni = a.cqg % kg == 0;
    a.divCpt = FastDiv(a.cqg);
}

int pair_run(pl_ctx *ctx, ConvArgs a, ConvArgs b, int cfg) {
    const CfgInfo &ci = kCfgs[cfg];
    pair_tile(a, ci);
    pair_tile(b, ci);
    if (ci.lds > 48 * 1024) {
        int rc = ensure_lds_attr((const void *)ci.pair, ci.lds);
        if (rc != PL_OK) return rc;
    }
    hipLaunchKernelGGL(ci.pair, dim3((unsigned)(a.tile_count + b.tile_count)), dim3(256), ci.lds, ctx->stream, a, b);
    PL_LAUNCH_CHECK();
    char buf[96];
    snprintf(buf, sizeof buf, "pair[%s tiles=%d+%d]", ci.name, a.tile_count, b.tile_count);
    ctx->last_plan = buf;
    // executed MACs of both convs (whole tiles, whole chunks) as one product
    auto macs = [&](const ConvArgs &c) {
        return (long long)c.mtiles * ci.bm * c.ntiles * ci.bn * ((c.Qtot * 4 + ci.bk - 1) / ci.bk * ci.bk);
    };
    ctx->last_gemm[0] = 1; ctx->last_gemm[1] = 1; ctx->last_gemm[2] = 1; ctx->last_gemm[3] = macs(a) + macs(b);
    return PL_OK;
}

int pair_launch(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const PairConv &ca, const PairConv &cb) {
    ConvArgs a, b;
    int rc = pair_conv_args(a, x, N, Cin, H, W, ca);
    if (rc != PL_OK) return rc;
    rc = pair_conv_args(b, x, N, Cin, H, W, cb);
    if (rc != PL_OK) return rc;
    // horizontal stride / pad ride in the high bits of the vertical ones (zero when equal: keys of square geometries are unchanged)
    TuneKey key = {{20, N, Cin, H, W, ca.Cout, ca.kh, ca.kw, ca.sh + 1024 * (ca.sw - ca.sh), ca.pt + 1024 * (ca.pl - ca.pt), cb.Cout, cb.kh,
                    cb.kw, cb.sh + 1024 * (cb.sw - cb.sh), cb.pt + 1024 * (cb.pl - cb.pt),
                    (ca.scale != nullptr) * 2 + (ca.bias != nullptr), (cb.scale != nullptr) * 2 + (cb.bias != nullptr), ca.act * 4 + cb.act}};
    int cfg = -1;
    if (ctx->conv_cfg >= 0 && ctx->conv_cfg < kNumCfgs && kCfgs[ctx->conv_cfg].pair) cfg = ctx->conv_cfg;      // forced (tests, sweeps)
    if (cfg < 0) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tune.find({ctx->device, key});
        if (it != g_tune.end()) cfg = it->second.cfg;
    }
    if (cfg < 0) {
        // first sight of the pair: time every configuration that has a pair kernel (never while capturing)
        int first = -1;
        float best = 1e30f;
        const bool tune = ctx->autotune && !ctx->capturing;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (tune && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return PL_EHIP;
        for (int c = 0; c < kNumCfgs; ++c) {
            if (!kCfgs[c].pair) continue;
            if (first < 0) first = c;
            if (!tune) continue;
            float ms_best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, ctx->stream);
                if (pair_run(ctx, a, b, c) != PL_OK) break;
                (void)hipEventRecord(e1, ctx->stream);
                if (hipEventSynchronize(e1) != hipSuccess) break;
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < ms_best) ms_best = ms;
            }
            if (ms_best < best) best = ms_best, cfg = c;
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (cfg < 0) cfg = first;
        PL_REQUIRE(cfg >= 0, PL_EUNSUPPORTED, "conv pair: no kernel configuration");
        if (tune) {
            ++ctx->tune_misses;
            std::lock_guard<std::mutex> lk(g_tune_mu);
            g_tune[{ctx->device, key}] = Plan{cfg, 0, 1, 0};
            if (getenv("PLANER_CONV_TUNE_LOG"))
                fprintf(stderr, "[planer_hip] conv pair N%d C%d %dx%d -> %d k%d s%d + %d k%d s%d: %s (%.3f ms)\n", N, Cin, H, W, ca.Cout,
                        ca.kh, ca.sh, cb.Cout, cb.kh, cb.sh, kCfgs[cfg].name, best);
        }
    }
    return pair_run(ctx, a, b, cfg);
}

}  // namespace

// ---- Dense on a small batch (layer.Dense, layer.py:15-18: y = x @ K^T + B with M = batch <= 64 rows) -------------
// As a 1x1 "convolution" the batch is the GEMM's column axis: 32 columns give 8 tiles, so the launch plan splits K
// 16 ways and adds a reduce kernel -- 13.6 us for ResNet-18's 33 MFLOP classifier.  Here a workgroup owns 32 output
// features x 32 batch rows; its four waves take a quarter of K each, reading both operands straight from global
// memory as float4s of 4 consecutive k (nothing is shared between waves, so nothing goes through LDS but the final
// sum of the four partial tiles).  32x32x2 MFMAs: lanes 0-31 carry k, lanes 32-63 carry k + 4.
__global__ void __launch_bounds__(256) dense_small_kernel(const float *x, const float *w, const float *bias, float *y,
                                                          int M, int K, int N, unsigned x_bytes, unsigned w_bytes) {
    __shared__ float part[3][32 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    const int kq = K / 4;                                  // K % 8 == 0 (host check): pairs of float4
    const int steps = kq / 2, per_wave = (steps + 3) / 4;
    const int s0 = wave * per_wave, s1 = min(steps, s0 + per_wave);
    const int xo = m0 + l31 < M ? ((m0 + l31) * K + 4 * lhi) << 2 : OOB;
    const int wo = n0 + l31 < N ? ((n0 + l31) * K + 4 * lhi) << 2 : OOB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = s0; s < s1; s += 4) {                     // four load pairs in flight per lane
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = s + u < s1;
            a[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? xo : OOB, (s + u) * 32, 0));
            b[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, ok ? wo : OOB, (s + u) * 32, 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
    }
    // C layout: column = lane & 31 (output feature), rows (batch) 8*(r>>2) + 4*(lane>>5) + (r&3)
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave - 1][(8 * (r >> 2) + 4 * lhi + (r & 3)) * 32 + l31] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int n = n0 + l31;
        const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * lhi + (r & 3);
            float v = acc[r];
            v = __fadd_rn(v, part[0][row * 32 + l31]);        // fixed order: waves 0, 1, 2, 3
            v = __fadd_rn(v, part[1][row * 32 + l31]);
            v = __fadd_rn(v, part[2][row * 32 + l31]);
            if (bias) v = __fadd_rn(v, bv);
            if (m0 + row < M && n < N) y[(size_t)(m0 + row) * N + n] = v;
        }
    }
}

extern "C" {

int pl_conv2d_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w, int Cout, int kh,
                  int kw, const float *bias, float *y, int sh, int sw, int dh, int dw, int pt, int pl, int pb,
                  int pr, int group) {
    return conv_launch(ctx, x, N, Cin, H, W, w, Cout, kh, kw, bias, y, sh, sw, dh, dw, pt, pl, pb, pr, group,
                       nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0, 0);
}

int pl_conv2d_fused_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w, int Cout, int kh,
                        int kw, const float *bias, float *y, int sh, int sw, int dh, int dw, int pt, int pl, int pb,
                        int pr, int group, const float *scale, const float *shift, const float *res, int act,
                        double alpha, int w_layout) {
    return conv_launch(ctx, x, N, Cin, H, W, w, Cout, kh, kw, bias, y, sh, sw, dh, dw, pt, pl, pb, pr, group, scale,
                       shift, res, act, alpha, w_layout);
}

int pl_conv2d_prepare_weights_f32(pl_ctx *ctx, const float *w, int Cout, int Cin_g, int kh, int kw, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_weights_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin_g > 0 && kh > 0 && kw > 0, PL_EINVAL, "pl_conv2d_prepare_weights_f32: bad shape");
    const size_t total = (size_t)Cout * Cin_g * kh * kw;
    PL_REQUIRE(total < (1ull << 31), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    permute_weights_kernel<<<blocks, 256, 0, ctx->stream>>>(w, out, (unsigned)total, Cin_g, kh * kw, FastDiv(Cin_g),
                                                           FastDiv(kh * kw));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_conv2d_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *wq, int Cout, int kh,
                     int kw, const float *bias, float *yq, int sh, int sw, int dh, int dw, int pt, int pl, int pb,
                     int pr, int group, const float *scale, const float *shift, const float *resq, int act,
                     double alpha) {
    PL_REQUIRE(!xq || !yq || ((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) |
                               reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(resq)) & 15u) == 0,
               PL_EINVAL, "pl_conv2d_q4_f32: Q4 tensors must be 16-byte aligned");
    return conv_launch(ctx, xq, N, Cin, H, W, wq, Cout, kh, kw, bias, yq, sh, sw, dh, dw, pt, pl, pb, pr, group,
                       scale, shift, resq, act, alpha, 2);
}

int pl_conv2d_rowpack_filter_elems(int Cout, int Cin, int kh, int kw, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, PL_EINVAL, "pl_conv2d_rowpack_filter_elems: bad argument");
    *elems = (size_t)(((size_t)kh * ((kw * Cin + 3) / 4) + 7) / 8 * 8) * Cout * 4;
    return PL_OK;
}

int pl_conv2d_prepare_rowpack_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, int kh, int kw, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_rowpack_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin > 0 && Cin < 4 && kh > 0 && kw > 0, PL_EINVAL, "row-packed filters are for Cin < 4");
    PL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, PL_EINVAL, "pl_conv2d_prepare_rowpack_f32: unaligned output");
    const int rq = (kw * Cin + 3) / 4, q_tot = kh * rq, q_pad = (q_tot + 7) / 8 * 8;
    const size_t total = (size_t)q_pad * Cout;
    PL_REQUIRE(total * 16 < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    pack_filter_rowpack_kernel<<<blocks, 256, 0, ctx->stream>>>(w, out, (unsigned)total, Cout, Cin, kh, kw, rq, q_tot,
                                                               FastDiv(Cout), FastDiv(rq));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

// geometry of the row-packed (zero-padded NHWC) image of an input: rows, pixels per row, floats incl. the slack the last
// window's quads may overrun
static void rowpack_geometry(int N, int Cin, int H, int W, int kw, int sw, int pt, int pl, int &Hp, int &Wp, size_t &pelems) {
    const int Wo = (W + 2 * pl - kw + sw) / sw;
    Hp = H + 2 * pt;
    Wp = rowpack_row_pixels(W, pl, Wo, sw, kw, Cin);
    pelems = ((size_t)N * Hp * Wp * Cin + 8 + 3) / 4 * 4;
}

static int rowpack_pack(pl_ctx *ctx, const float *x, float *xp, int N, int Cin, int H, int W, int kw, int sw, int pt, int pl) {
    int Hp, Wp;
    size_t pelems;
    rowpack_geometry(N, Cin, H, W, kw, sw, pt, pl, Hp, Wp, pelems);
    PL_REQUIRE(pelems < (1ull << 29) && (size_t)N * Cin * H * W < (1ull << 29), PL_EUNSUPPORTED, "row-packed input above 2 GiB");
    const unsigned total = (unsigned)((size_t)N * Hp * Wp * Cin), total4 = (unsigned)((pelems + 3) / 4);   // incl. the slack
    nchw_to_rowpack_kernel<<<std::min<unsigned>((total4 + 255) / 256, 256 * 16), 256, 0, ctx->stream>>>(
        x, xp, total4, total, Cin, H, W, Hp, Wp, pt, pl, (unsigned)((size_t)N * Cin * H * W * 4), FastDiv(Cin), FastDiv(Wp), FastDiv(Hp));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

static int rowpack_check(pl_ctx *ctx, const void *x, int N, int Cin, int H, int W, const float *wq, int Cout, int kh, int kw,
                         float *yq, int sh, int sw, int pt, int pl, const float *resq) {
    PL_REQUIRE(ctx && x && wq && yq, PL_EINVAL, "pl_conv2d_rowpack_q4_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && Cin < 4 && H > 0 && W > 0 && Cout > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 &&
                   pt >= 0 && pl >= 0, PL_EINVAL, "pl_conv2d_rowpack_q4_f32: bad shape (Cin must be 1..3)");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(resq)) & 15u) == 0,
               PL_EINVAL, "Q4 tensors must be 16-byte aligned");
    const int Ho = (H + 2 * pt - kh + sh) / sh, Wo = (W + 2 * pl - kw + sw) / sw;
    PL_REQUIRE(N == 0 || (Ho > 0 && Wo > 0), PL_EINVAL, "pl_conv2d_rowpack_q4_f32: empty output");
    return PL_OK;
}

static int rowpack_conv(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *wq, int Cout, int kh,
                        int kw, const float *bias, float *yq, int sh, int sw, int pt, int pl, const float *scale,
                        const float *shift, const float *resq, int act, double alpha) {
    int rc = rowpack_check(ctx, x, N, Cin, H, W, wq, Cout, kh, kw, yq, sh, sw, pt, pl, resq);
    if (rc != PL_OK) return rc;
    if (N == 0) return PL_OK;
    int Hp, Wp;
    size_t pelems;
    rowpack_geometry(N, Cin, H, W, kw, sw, pt, pl, Hp, Wp, pelems);
    CtxGuard guard(ctx);
    float *xp = nullptr;
    rc = pl_alloc(ctx, pelems * sizeof(float), (void **)&xp);
    if (rc != PL_OK) return rc;
    rc = rowpack_pack(ctx, x, xp, N, Cin, H, W, kw, sw, pt, pl);
    if (rc == PL_OK)
        rc = conv_launch(ctx, xp, N, Cin, H, W, wq, Cout, kh, kw, bias, yq, sh, sw, 1, 1, pt, pl, pt, pl, 1, scale, shift, resq,
                         act, alpha, 6);
    pl_free(ctx, xp);            // stream-ordered
    return rc;
}

int pl_rowpack_input_elems(int N, int Cin, int H, int W, int kw, int sw, int pt, int pl, size_t *elems) {
    PL_REQUIRE(elems && N >= 0 && Cin > 0 && Cin < 4 && H > 0 && W > 0 && kw > 0 && sw > 0 && pt >= 0 && pl >= 0, PL_EINVAL,
               "pl_rowpack_input_elems: bad argument");
    int Hp, Wp;
    rowpack_geometry(N, Cin, H, W, kw, sw, pt, pl, Hp, Wp, *elems);
    return PL_OK;
}

int pl_rowpack_input_f32(pl_ctx *ctx, const float *x, float *xp, int N, int Cin, int H, int W, int kw, int sw, int pt, int pl) {
    PL_REQUIRE(ctx && x && xp, PL_EINVAL, "pl_rowpack_input_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && Cin < 4 && H > 0 && W > 0 && kw > 0 && sw > 0 && pt >= 0 && pl >= 0, PL_EINVAL,
               "pl_rowpack_input_f32: bad shape (Cin must be 1..3)");
    PL_REQUIRE((reinterpret_cast<uintptr_t>(xp) & 15u) == 0, PL_EINVAL, "pl_rowpack_input_f32: unaligned output");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return rowpack_pack(ctx, x, xp, N, Cin, H, W, kw, sw, pt, pl);
}

int pl_conv2d_rowpacked_q4_f32(pl_ctx *ctx, const float *xp, int N, int Cin, int H, int W, const float *wq, int Cout, int kh,
                               int kw, const float *bias, float *yq, int sh, int sw, int pt, int pl, const float *scale,
                               const float *shift, const float *resq, int act, double alpha) {
    int rc = rowpack_check(ctx, xp, N, Cin, H, W, wq, Cout, kh, kw, yq, sh, sw, pt, pl, resq);
    if (rc != PL_OK) return rc;
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    return conv_launch(ctx, xp, N, Cin, H, W, wq, Cout, kh, kw, bias, yq, sh, sw, 1, 1, pt, pl, pt, pl, 1, scale, shift, resq, act,
                       alpha, 6);
}

// The row-packed stem conv + maxpool(3x3 / stride 2 / pad 1) in one kernel (conv_stem_pool_kernel.h).  The kernel is built for
// the stem of an ImageNet-style net: 3 input channels, 7x7 / stride 2 / pad 3; any height, any width (maps wider than 112 conv
// columns are cut into column chunks).
// -> column chunks per strip, pooled columns per chunk, 16-column blocks per workgroup (0 chunks: not supported)
static void stem_pool_chunks(int Wo, int &chunks, int &pq, int &nb) {
    const int Wq = (Wo + 1) / 2;
    chunks = 0;
    for (int c = 1; c <= 64; ++c) {
        pq = (Wq + c - 1) / c;
        const int cols = c == 1 ? Wo : 2 * pq + 2;                 // chunk c > 0 starts one window early
        nb = (cols + 15) / 16;
        if (nb <= SP_NB_MAX) {
            chunks = c;
            return;
        }
    }
}
static bool stem_pool_shape_ok(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int pt, int pl) {
    const int Wo = (W + 2 * pl - kw + sw) / sw, Ho = (H + 2 * pt - kh + sh) / sh;
    int chunks = 0, pq = 0, nb = 0;
    if (Wo >= 1) stem_pool_chunks(Wo, chunks, pq, nb);
    return Cin == 3 && kh == SP_KH && kw == 7 && sh == 2 && sw == 2 && pt == 3 && pl == 3 && Wo >= 1 && chunks > 0 && Ho >= 2 && Cout > 0 &&
           Cout % 4 == 0;
}

int pl_conv2d_rowpacked_pool_supported(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int pt, int pl, int *ok) {
    PL_REQUIRE(ok, PL_EINVAL, "pl_conv2d_rowpacked_pool_supported: null argument");
    *ok = stem_pool_shape_ok(Cin, H, W, Cout, kh, kw, sh, sw, pt, pl) ? 1 : 0;
    return PL_OK;
}

int pl_conv2d_rowpacked_pool_q4_f32(pl_ctx *ctx, const float *xp, int N, int Cin, int H, int W, const float *wq, int Cout, int kh,
                                    int kw, const float *bias, float *yq, int sh, int sw, int pt, int pl, const float *scale,
                                    const float *shift, int act, double alpha) {
    int rc = rowpack_check(ctx, xp, N, Cin, H, W, wq, Cout, kh, kw, yq, sh, sw, pt, pl, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(stem_pool_shape_ok(Cin, H, W, Cout, kh, kw, sh, sw, pt, pl), PL_EUNSUPPORTED,
               "conv + maxpool (row-packed stem): 3 channels, 7x7 / stride 2 / pad 3, Cout %% 4 == 0");
    PL_REQUIRE(act >= 0 && act <= 2, PL_EINVAL, "pl_conv2d_rowpacked_pool_q4_f32: bad activation code");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u) == 0,
               PL_EINVAL, "pl_conv2d_rowpacked_pool_q4_f32: bias / scale / shift are read as 16-byte quads");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    int Hp, Wp;
    size_t pelems;
    rowpack_geometry(N, Cin, H, W, kw, sw, pt, pl, Hp, Wp, pelems);
    StemPoolArgs a;
    memset(&a, 0, sizeof a);
    a.xp = xp; a.wq = wq; a.y = yq;
    a.N = N; a.Hp = Hp; a.rowf = Wp * Cin;
    a.Ho = (H + 2 * pt - kh + sh) / sh; a.Wo = (W + 2 * pl - kw + sw) / sw;
    a.Hq = (a.Ho + 1) / 2; a.Wq = (a.Wo + 1) / 2;            // (Ho + 2 - 3 + 2) // 2, util.py:84-85
    a.Cout = Cout; a.Coq = Cout / 4;
    a.prows = SP_PROWS; a.steps = a.prows + 1;
    a.strips = (a.Hq + a.prows - 1) / a.prows;
    a.cout_blocks = (Cout + 63) / 64;
    const int q_pad = (kh * ((kw * Cin + 3) / 4) + 7) / 8 * 8;
    const size_t yb = (size_t)N * a.Coq * a.Hq * a.Wq * 16;
    int nb = 0;
    stem_pool_chunks(a.Wo, a.chunks, a.pq, nb);
    PL_REQUIRE(a.chunks > 0 && pelems < (1ull << 29) && yb < (1ull << 31) && 4 * SP_GROUPS <= q_pad, PL_EUNSUPPORTED,
               "conv + maxpool (row-packed stem): tensor too large");
    a.x_bytes = (unsigned)(pelems * 4); a.w_bytes = (unsigned)((size_t)q_pad * Cout * 16); a.y_bytes = (unsigned)yb;
    a.ep = make_epilogue(bias, scale, shift, nullptr, act, alpha);
    const long long blocks = (long long)N * a.strips * a.chunks * a.cout_blocks;
    PL_REQUIRE(blocks < (1ll << 31), PL_EUNSUPPORTED, "conv + maxpool (row-packed stem): grid too large");
    void (*kern)(const StemPoolArgs) = nullptr;
    switch (nb) {
    case 1: kern = conv_stem_pool_kernel<1, false>; break;
    case 2: kern = conv_stem_pool_kernel<2, false>; break;
    case 3: kern = conv_stem_pool_kernel<3, false>; break;
    case 4: kern = conv_stem_pool_kernel<4, false>; break;
    case 5: kern = conv_stem_pool_kernel<5, false>; break;
    case 6: kern = conv_stem_pool_kernel<6, false>; break;
    default: kern = conv_stem_pool_kernel<7, false>; break;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), 0, ctx->stream, a);
    PL_LAUNCH_CHECK();
    char buf[112];
    snprintf(buf, sizeof buf, "stem+maxpool 64co x 2 rows x %dpx, strips=%d chunks=%d blocks=%lld", 16 * nb, a.strips, a.chunks, blocks);
    ctx->last_plan = buf;
    // executed MFMA work: 16 conv rows per strip, 16 nb columns per chunk, K = 176
    ctx->last_gemm[0] = 1; ctx->last_gemm[1] = (long long)a.cout_blocks * 64;
    ctx->last_gemm[2] = (long long)N * a.strips * a.chunks * 2 * a.steps * 16 * nb; ctx->last_gemm[3] = 16 * SP_GROUPS;
    return PL_OK;
}

// The same kernel reading the NCHW input itself (conv_stem_pool_kernel<NB, true>): no row-packed copy of the batch.  Needs whole
// 16-byte cells per input row: W % 4 == 0 and a 16-byte aligned tensor.
int pl_conv2d_stem_pool_nchw_supported(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int pt, int pl, int *ok) {
    PL_REQUIRE(ok, PL_EINVAL, "pl_conv2d_stem_pool_nchw_supported: null argument");
    *ok = (stem_pool_shape_ok(Cin, H, W, Cout, kh, kw, sh, sw, pt, pl) && W % 4 == 0) ? 1 : 0;
    return PL_OK;
}

int pl_conv2d_stem_nchw_filter_elems(int Cout, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0, PL_EINVAL, "pl_conv2d_stem_nchw_filter_elems: bad argument");
    *elems = (size_t)((4 * SP_GROUPS + 7) / 8 * 8) * Cout * 4;
    return PL_OK;
}

int pl_conv2d_prepare_stem_nchw_f32(pl_ctx *ctx, const float *w, int Cout, float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_stem_nchw_f32: null pointer");
    PL_REQUIRE(Cout > 0, PL_EINVAL, "pl_conv2d_prepare_stem_nchw_f32: bad shape");
    PL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, PL_EINVAL, "pl_conv2d_prepare_stem_nchw_f32: unaligned output");
    const int q_pad = (4 * SP_GROUPS + 7) / 8 * 8;
    const size_t total = (size_t)q_pad * Cout;
    PL_REQUIRE(total * 16 < (1ull << 29), PL_EUNSUPPORTED, "filter too large");
    CtxGuard g(ctx);
    pack_filter_stem_nchw_kernel<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(w, reinterpret_cast<float4 *>(out), (unsigned)total, Cout);
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_conv2d_stem_pool_nchw_q4_f32(pl_ctx *ctx, const float *x, int N, int H, int W, const float *wq, int Cout, const float *bias,
                                    float *yq, const float *scale, const float *shift, int act, double alpha, int strip_rows) {
    const int Cin = 3, kh = SP_KH, kw = 7, sh = 2, sw = 2, pt = 3, pl = 3;
    int rc = rowpack_check(ctx, x, N, Cin, H, W, wq, Cout, kh, kw, yq, sh, sw, pt, pl, nullptr);
    if (rc != PL_OK) return rc;
    PL_REQUIRE(stem_pool_shape_ok(Cin, H, W, Cout, kh, kw, sh, sw, pt, pl) && W % 4 == 0, PL_EUNSUPPORTED,
               "conv + maxpool (NCHW stem): 3 channels, 7x7 / stride 2 / pad 3, W %% 4 == 0, Cout %% 4 == 0");
    PL_REQUIRE(act >= 0 && act <= 2, PL_EINVAL, "pl_conv2d_stem_pool_nchw_q4_f32: bad activation code");
    PL_REQUIRE(strip_rows == 0 || strip_rows == 7 || strip_rows == 14, PL_EINVAL, "pl_conv2d_stem_pool_nchw_q4_f32: strip_rows is 0 (= 7), 7 or 14");
    PL_REQUIRE(((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
                 reinterpret_cast<uintptr_t>(x)) & 15u) == 0,
               PL_EINVAL, "pl_conv2d_stem_pool_nchw_q4_f32: x / bias / scale / shift are read as 16-byte quads");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    StemPoolArgs a;
    memset(&a, 0, sizeof a);
    a.xp = x; a.wq = wq; a.y = yq;
    a.N = N; a.H = H; a.W = W;
    a.Ho = (H + 2 * pt - kh + sh) / sh; a.Wo = (W + 2 * pl - kw + sw) / sw;
    a.Hq = (a.Ho + 1) / 2; a.Wq = (a.Wo + 1) / 2;            // (Ho + 2 - 3 + 2) // 2, util.py:84-85
    a.Cout = Cout; a.Coq = Cout / 4;
    a.prows = strip_rows == 14 ? 14 : SP_PROWS;      // (14: half the workgroups, 15 steps instead of 2 x 8)
    a.steps = a.prows + 1;
    a.strips = (a.Hq + a.prows - 1) / a.prows;
    a.cout_blocks = (Cout + 63) / 64;
    const int q_pad = (4 * SP_GROUPS + 7) / 8 * 8;
    const size_t xb = (size_t)N * Cin * H * W * 4, yb = (size_t)N * a.Coq * a.Hq * a.Wq * 16;
    int nb = 0;
    stem_pool_chunks(a.Wo, a.chunks, a.pq, nb);
    PL_REQUIRE(a.chunks > 0 && xb < (1ull << 31) && yb < (1ull << 31), PL_EUNSUPPORTED, "conv + maxpool (NCHW stem): tensor too large");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)((size_t)q_pad * Cout * 16); a.y_bytes = (unsigned)yb;
    a.ep = make_epilogue(bias, scale, shift, nullptr, act, alpha);
    const long long blocks = (long long)N * a.strips * a.chunks * a.cout_blocks;
    PL_REQUIRE(blocks < (1ll << 31), PL_EUNSUPPORTED, "conv + maxpool (NCHW stem): grid too large");
    void (*kern)(const StemPoolArgs) = nullptr;
    switch (nb) {
    case 1: kern = conv_stem_pool_kernel<1, true>; break;
    case 2: kern = conv_stem_pool_kernel<2, true>; break;
    case 3: kern = conv_stem_pool_kernel<3, true>; break;
    case 4: kern = conv_stem_pool_kernel<4, true>; break;
    case 5: kern = conv_stem_pool_kernel<5, true>; break;
    case 6: kern = conv_stem_pool_kernel<6, true>; break;
    default: kern = conv_stem_pool_kernel<7, true>; break;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), 0, ctx->stream, a);
    PL_LAUNCH_CHECK();
    char buf[112];
    snprintf(buf, sizeof buf, "stem+maxpool(nchw) 64co x 2 rows x %dpx, strips=%d of %d rows, chunks=%d blocks=%lld", 16 * nb, a.strips, a.prows, a.chunks, blocks);
    ctx->last_plan = buf;
    ctx->last_gemm[0] = 1; ctx->last_gemm[1] = (long long)a.cout_blocks * 64;
    ctx->last_gemm[2] = (long long)N * a.strips * a.chunks * 2 * a.steps * 16 * nb; ctx->last_gemm[3] = 16 * SP_GROUPS;
    return PL_OK;
}

int pl_conv2d_rowpack_q4_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *wq, int Cout, int kh,
                             int kw, const float *bias, float *yq, int sh, int sw, int pt, int pl, const float *scale,
                             const float *shift, const float *resq, int act, double alpha) {
    return rowpack_conv(ctx, x, N, Cin, H, W, wq, Cout, kh, kw, bias, yq, sh, sw, pt, pl, scale, shift, resq, act, alpha);
}

int pl_conv2d_q4_pair_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                          const float *wq_a, int Cout_a, int kh_a, int kw_a, int sh_a, int sw_a, int pt_a, int pl_a,
                          const float *bias_a, const float *scale_a, const float *shift_a, int act_a, double alpha_a, float *yq_a,
                          const float *wq_b, int Cout_b, int kh_b, int kw_b, int sh_b, int sw_b, int pt_b, int pl_b,
                          const float *bias_b, const float *scale_b, const float *shift_b, int act_b, double alpha_b, float *yq_b) {
    PL_REQUIRE(ctx && xq && wq_a && wq_b && yq_a && yq_b, PL_EINVAL, "pl_conv2d_q4_pair_f32: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout_a > 0 && Cout_b > 0 && kh_a > 0 && kw_a > 0 && kh_b > 0 && kw_b > 0 &&
                   sh_a > 0 && sw_a > 0 && sh_b > 0 && sw_b > 0 && pt_a >= 0 && pl_a >= 0 && pt_b >= 0 && pl_b >= 0,
               PL_EINVAL, "pl_conv2d_q4_pair_f32: bad shape");
    PL_REQUIRE(H + 2 * std::max(pt_a, pt_b) < 16384 && W + 2 * std::max(pl_a, pl_b) < 16384, PL_EUNSUPPORTED,
               "pl_conv2d_q4_pair_f32: spatial extent above 16383");
    PL_REQUIRE(act_a >= 0 && act_a <= 2 && act_b >= 0 && act_b <= 2, PL_EINVAL, "pl_conv2d_q4_pair_f32: bad activation code");
    PL_REQUIRE(yq_a != yq_b, PL_EINVAL, "pl_conv2d_q4_pair_f32: the two outputs must differ");
    if (N == 0) return PL_OK;
    CtxGuard guard(ctx);
    const PairConv ca{wq_a, bias_a, scale_a, shift_a, yq_a, Cout_a, kh_a, kw_a, sh_a, sw_a, pt_a, pl_a, act_a, alpha_a};
    const PairConv cb{wq_b, bias_b, scale_b, shift_b, yq_b, Cout_b, kh_b, kw_b, sh_b, sw_b, pt_b, pl_b, act_b, alpha_b};
    return pair_launch(ctx, xq, N, Cin, H, W, ca, cb);
}

int pl_conv2d_q4_filter_elems(int Cout, int Cin_g, int kh, int kw, int group, size_t *elems) {
    PL_REQUIRE(elems && Cout > 0 && Cin_g > 0 && kh > 0 && kw > 0 && group > 0 && Cout % group == 0, PL_EINVAL,
               "pl_conv2d_q4_filter_elems: bad argument");
    const size_t q_pad = ((size_t)kh * kw * ((Cin_g + 3) / 4) + 7) / 8 * 8;
    *elems = (size_t)group * q_pad * (Cout / group) * 4;
    return PL_OK;
}

int pl_conv2d_prepare_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin_g, int kh, int kw, int group,
                             float *out) {
    PL_REQUIRE(ctx && w && out, PL_EINVAL, "pl_conv2d_prepare_q4_f32: null pointer");
    PL_REQUIRE(Cout > 0 && Cin_g > 0 && kh > 0 && kw > 0 && group > 0 && Cout % group == 0, PL_EINVAL,
               "pl_conv2d_prepare_q4_f32: bad shape");
    PL_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15u) == 0, PL_EINVAL, "pl_conv2d_prepare_q4_f32: unaligned output");
    const int cout_g = Cout / group, cqg = (Cin_g + 3) / 4, q_tot = kh * kw * cqg, q_pad = (q_tot + 7) / 8 * 8;
    const size_t total = (size_t)group * q_pad * cout_g;       // float4s
    PL_REQUIRE(total * 4 < (1ull << 29) && (size_t)Cout * Cin_g * kh * kw < (1ull << 31), PL_EUNSUPPORTED,
               "filter too large");
    CtxGuard g(ctx);
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    pack_filter_q4_kernel<<<blocks, 256, 0, ctx->stream>>>(w, out, (unsigned)total, cout_g, Cin_g, kh * kw, cqg, q_tot,
                                                          q_pad, FastDiv(cout_g), FastDiv(q_pad), FastDiv(cqg));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

static int q4_convert(pl_ctx *ctx, const float *x, float *y, int N, int C, int HW, bool to_q4) {
    PL_REQUIRE(ctx && x && y, PL_EINVAL, "q4 layout conversion: null pointer");
    PL_REQUIRE(N >= 0 && C > 0 && HW > 0, PL_EINVAL, "q4 layout conversion: bad shape");
    const int Cq = (C + 3) / 4;
    const size_t total = (size_t)N * Cq * HW;
    PL_REQUIRE(total < (1ull << 29), PL_EUNSUPPORTED, "q4 layout conversion: tensor above 2 GiB");
    PL_REQUIRE((reinterpret_cast<uintptr_t>(to_q4 ? y : x) & 15u) == 0, PL_EINVAL,
               "q4 layout conversion: Q4 tensor must be 16-byte aligned");
    if (total == 0) return PL_OK;
    CtxGuard g(ctx);
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    if (to_q4)
        nchw_to_q4_kernel<<<blocks, 256, 0, ctx->stream>>>(x, y, (unsigned)total, C, Cq, HW, FastDiv(HW), FastDiv(Cq));
    else
        q4_to_nchw_kernel<<<blocks, 256, 0, ctx->stream>>>(x, y, (unsigned)total, C, Cq, HW, FastDiv(HW), FastDiv(Cq));
    PL_LAUNCH_CHECK();
    return PL_OK;
}

int pl_nchw_to_q4_f32(pl_ctx *ctx, const float *x, float *yq, int N, int C, int HW) {
    return q4_convert(ctx, x, yq, N, C, HW, true);
}

int pl_q4_to_nchw_f32(pl_ctx *ctx, const float *xq, float *y, int N, int C, int HW) {
    return q4_convert(ctx, xq, y, N, C, HW, false);
}

int pl_tune_cache_save(pl_ctx *ctx, const char *path) {
    PL_REQUIRE(ctx && path, PL_EINVAL, "pl_tune_cache_save: null argument");
    FILE *f = fopen(path, "w");
    PL_REQUIRE(f, PL_EINVAL, "pl_tune_cache_save: cannot open %s", path);
    std::lock_guard<std::mutex> lk(g_tune_mu);
    for (auto &kv : g_tune) {
        if (kv.first.first != ctx->device) continue;
        for (int v : kv.first.second.v) fprintf(f, "%d ", v);
        fprintf(f, "%s %d %d %d\n", kCfgs[kv.second.cfg].name, kv.second.t1, kv.second.s2, kv.second.occ);
    }
    fclose(f);
    return PL_OK;
}

int pl_tune_cache_load(pl_ctx *ctx, const char *path, int *entries) {
    PL_REQUIRE(ctx && path, PL_EINVAL, "pl_tune_cache_load: null argument");
    if (entries) *entries = 0;
    FILE *f = fopen(path, "r");
    if (!f) return PL_OK;   // no cache yet
    std::lock_guard<std::mutex> lk(g_tune_mu);
    for (;;) {
        TuneKey key;
        bool ok = true;
        for (int &v : key.v) ok = ok && fscanf(f, "%d", &v) == 1;
        char name[64];
        Plan pl{-1, 0, 1, 0};
        if (!ok || fscanf(f, "%63s %d %d %d", name, &pl.t1, &pl.s2, &pl.occ) != 4) break;
        for (int c = 0; c < kNumCfgs; ++c)
            if (!strcmp(kCfgs[c].name, name)) pl.cfg = c;
        if (key.v[0] == 20) {                    // a conv pair (pl_conv2d_q4_pair_f32): any configuration with a pair kernel
            if (pl.cfg < 0 || !kCfgs[pl.cfg].pair) continue;
        } else if (pl.cfg < 0 || !cfg_applies(kCfgs[pl.cfg], key.v[0], key.v[2] / (key.v[14] > 0 ? key.v[14] : 1))) continue;
        g_tune[{ctx->device, key}] = pl;
        if (entries) ++*entries;
    }
    fclose(f);
    return PL_OK;
}

int pl_tune_stats(pl_ctx *ctx, int *entries, int *misses) {
    PL_REQUIRE(ctx, PL_EINVAL, "pl_tune_stats: null context");
    if (entries) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        int n = 0;
        for (auto &kv : g_tune) n += kv.first.first == ctx->device;
        *entries = n;
    }
    if (misses) *misses = ctx->tune_misses;
    return PL_OK;
}

int pl_conv2d_last_extents(pl_ctx *ctx, long long *ext4) {
    PL_REQUIRE(ctx && ext4, PL_EINVAL, "pl_conv2d_last_extents: null argument");
    for (int i = 0; i < 4; ++i) ext4[i] = ctx->last_gemm[i];
    return PL_OK;
}

int pl_set_autotune(pl_ctx *ctx, int enabled) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    ctx->autotune = enabled != 0;
    return PL_OK;
}

int pl_conv2d_set_config(pl_ctx *ctx, int cfg, int split_k) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    PL_REQUIRE(cfg < kNumCfgs, PL_EINVAL, "config %d out of range (%d)", cfg, kNumCfgs);
    ctx->conv_cfg = cfg;
    ctx->conv_split_k = split_k;
    ctx->conv_t1 = 0;
    ctx->conv_occ = 0;
    return PL_OK;
}

int pl_conv2d_set_plan(pl_ctx *ctx, int cfg, int dp_tiles, int split_k, int occupancy) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    PL_REQUIRE(cfg < kNumCfgs, PL_EINVAL, "config %d out of range (%d)", cfg, kNumCfgs);
    ctx->conv_cfg = cfg;
    ctx->conv_split_k = split_k;
    ctx->conv_t1 = dp_tiles;
    ctx->conv_occ = occupancy;
    return PL_OK;
}

int pl_conv2d_num_configs(void) { return kNumCfgs; }

int pl_conv2d_last_plan(pl_ctx *ctx, char *buf, size_t len) {
    PL_REQUIRE(ctx && buf && len, PL_EINVAL, "pl_conv2d_last_plan: bad argument");
    snprintf(buf, len, "%s", ctx->last_plan.c_str());
    return PL_OK;
}

int pl_conv2d_config_name(int cfg, char *buf, size_t len) {
    PL_REQUIRE(cfg >= 0 && cfg < kNumCfgs && buf && len, PL_EINVAL, "pl_conv2d_config_name: bad argument");
    snprintf(buf, len, "%s", kCfgs[cfg].name);
    return PL_OK;
}

// Dense / MatMul are 1x1 convolutions over degenerate images, so they run on
// the same MFMA kernel:
//   trans_b=1 (layer.Dense, layer.py:15-18): y[m][n] = sum_k a[m][k]*b[n][k] + bias[n]
//       -> "filters" = b [N][K], "input" = a as (M, K, 1, 1), output (M, N, 1, 1)
//   trans_b=0 (layer.MatMul, layer.py:20):  y[m][n] = sum_k a[m][k]*b[k][n]
//       -> "filters" = a [M][K], "input" = b as (1, K, 1, N), output (1, M, 1, N)
int pl_gemm_f32(pl_ctx *ctx, const float *a, int M, int K, const float *b, int N, int trans_b, const float *bias,
                float *y) {
    PL_REQUIRE(ctx && a && b && y, PL_EINVAL, "pl_gemm_f32: null pointer");
    PL_REQUIRE(M >= 0 && N >= 0 && K > 0, PL_EINVAL, "pl_gemm_f32: bad shape");
    if (M == 0 || N == 0) return PL_OK;
    static const bool small_off = getenv("PLANER_HIP_DENSE_SMALL") && atoi(getenv("PLANER_HIP_DENSE_SMALL")) == 0;
    if (trans_b && M <= 64 && N >= 32 && K % 8 == 0 && K >= 64 && (size_t)M * K < (1ull << 29) && (size_t)N * K < (1ull << 29) &&
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) == 0 && ctx->conv_cfg < 0 && !small_off) {
        CtxGuard guard(ctx);
        dense_small_kernel<<<dim3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32)), 256, 0, ctx->stream>>>(
            a, b, bias, y, M, K, N, (unsigned)((size_t)M * K * 4), (unsigned)((size_t)N * K * 4));
        PL_LAUNCH_CHECK();
        ctx->last_plan = "dense32x32 tiles=" + std::to_string(((N + 31) / 32) * ((M + 31) / 32)) + " kwaves=4";
        ctx->last_gemm[0] = 1; ctx->last_gemm[1] = (long long)(N + 31) / 32 * 32;
        ctx->last_gemm[2] = (long long)(M + 31) / 32 * 32; ctx->last_gemm[3] = (long long)(K + 7) / 8 * 8;
        return PL_OK;
    }
    if (trans_b)
        return conv_launch(ctx, a, M, K, 1, 1, b, N, 1, 1, bias, y, 1, 1, 1, 1, 0, 0, 0, 0, 1, nullptr, nullptr,
                           nullptr, PL_ACT_NONE, 0.0, 0);
    PL_REQUIRE(!bias, PL_EUNSUPPORTED, "pl_gemm_f32: bias needs trans_b=1");
    return conv_launch(ctx, b, 1, K, 1, N, a, M, 1, 1, nullptr, y, 1, 1, 1, 1, 0, 0, 0, 0, 1, nullptr, nullptr,
                       nullptr, PL_ACT_NONE, 0.0, 0);
}

}  // extern "C"
