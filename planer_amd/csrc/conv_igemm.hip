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
#include <cmath>

#include "common.h"
#include "device_utils.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float *x, *w;
    float *y;  // output, or split-K workspace [splits][N*Cout*HoWo]
    int N, Cin, H, W, Cout, Ho, Wo;
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int cin_g, cout_g, groups;
    int K;            // cin_g*kh*kw
    int cols;         // N*Ho*Wo
    int HoWo, HW;
    int mtiles, ntiles, tiles;  // per group
    int splits, k_per_split;
    size_t slab;      // N*Cout*HoWo
    int x_bytes, w_bytes;
    FastDiv divKhw, divKw, divHoWo, divWo, divMt;
    Epilogue ep;
};

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

    // ---- XCD-aware tile mapping (bijective for any grid size) -------------
    const unsigned nblk = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // bid -> (group, ntile, mtile); mtile fastest
    const unsigned g = bid / (unsigned)p.tiles;
    const unsigned t = bid - g * (unsigned)p.tiles;
    const unsigned nt = p.divMt.div(t);
    const unsigned mt = t - nt * (unsigned)p.mtiles;
    const int m0 = mt * C::BM;          // row offset inside the group
    const int col0 = nt * C::BN;
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

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31,
    //      row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float *yout = p.y + (size_t)split * p.slab;
    const bool fuse = p.splits == 1;
#pragma unroll
    for (int b = 0; b < C::TN; ++b) {
        const int jc = col0 + wn * C::WTN + b * 32 + l31;
        if (jc >= p.cols) continue;
        unsigned n, pix;
        p.divHoWo.divmod((unsigned)jc, n, pix);
        const size_t obase = ((size_t)n * p.Cout + (size_t)g * p.cout_g) * p.HoWo + pix;
#pragma unroll
        for (int a = 0; a < C::TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * C::WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row < p.cout_g) {
                    const size_t idx = obase + (size_t)row * p.HoWo;
                    float v = acc[a][b][r];
                    if (fuse) v = apply_epilogue(p.ep, v, (int)g * p.cout_g + row, idx);
                    yout[idx] = v;
                }
            }
        }
    }
}

// ---- configurations ---------------------------------------------------------
typedef Cfg<128, 128, 16, 2, 2> C128x128;
typedef Cfg<64, 128, 16, 2, 2> C64x128;
typedef Cfg<128, 64, 16, 2, 2> C128x64;
typedef Cfg<64, 64, 16, 2, 2> C64x64;
typedef Cfg<32, 128, 16, 1, 4> C32x128;
typedef Cfg<32, 256, 16, 1, 4> C32x256;
typedef Cfg<64, 256, 16, 2, 2> C64x256;
typedef Cfg<128, 32, 16, 4, 1> C128x32;

struct CfgInfo {
    const char *name;
    int bm, bn, bk, lds;
    void (*vec)(const ConvArgs);
    void (*scl)(const ConvArgs);
};

#define CFG_ENTRY(T, nm) \
    { nm, T::BM, T::BN, T::BK, T::LDS_BYTES, conv_igemm_kernel<T, true>, conv_igemm_kernel<T, false> }

const CfgInfo kCfgs[] = {
    CFG_ENTRY(C128x128, "128x128"), CFG_ENTRY(C64x128, "64x128"), CFG_ENTRY(C128x64, "128x64"),
    CFG_ENTRY(C64x64, "64x64"),     CFG_ENTRY(C32x128, "32x128"), CFG_ENTRY(C32x256, "32x256"),
    CFG_ENTRY(C64x256, "64x256"),   CFG_ENTRY(C128x32, "128x32"),
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// Pick tile + split-K so the grid covers the 256 CUs a few times over while
// wasting as little padded work as possible.
void choose_config(pl_ctx *ctx, int cout_g, int cols, int K, int groups, int &cfg, int &splits) {
    const double cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    double best = 1e300;
    cfg = 0;
    splits = 1;
    for (int c = 0; c < kNumCfgs; ++c) {
        const CfgInfo &ci = kCfgs[c];
        const double mt = (cout_g + ci.bm - 1) / ci.bm, nt = (cols + ci.bn - 1) / ci.bn;
        const double tiles = mt * nt * groups;
        for (int s = 1; s <= 16; s *= 2) {
            if (s > 1 && K / s < 8 * ci.bk) break;  // keep >= 8 chunks per split
            const double blocks = tiles * s;
            // padded MFMA work per block, in units of 32x32x2 MFMA issue slots (64 cycles)
            const double kchunks = (double)((K + s - 1) / s + ci.bk - 1) / ci.bk;
            const double mfma_per_wave = kchunks * (ci.bk / 2) * (ci.bm / 32) * (ci.bn / 32) / 4.0;
            // fixed per-chunk overhead (barrier + staging not hidden) and per-block prologue/epilogue
            const double block_cost = mfma_per_wave + kchunks * 3.0 + 40.0 + (ci.bm / 32) * (ci.bn / 32) * 2.0;
            // co-resident workgroups per CU (LDS/regs allow at least 2; big tiles: 2, small: 4)
            const double per_cu = (ci.bm * ci.bn >= 128 * 128) ? 2.0 : (ci.bm * ci.bn >= 64 * 128 ? 3.0 : 4.0);
            const double slots = cus * per_cu;
            const double rounds = std::ceil(blocks / slots);
            // each SIMD hosts per_cu waves; throughput-limited by MFMA pipe
            double cost = rounds * per_cu * block_cost;
            if (s > 1) cost += 60.0 + (double)cout_g * cols * groups * (s + 1) / (cus * 4 * 64.0) ;  // reduce pass
            if (cost < best) {
                best = cost;
                cfg = c;
                splits = s;
            }
        }
    }
}

int conv_launch(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w, int Cout, int kh, int kw,
                const float *bias, float *y, int sh, int sw, int dh, int dw, int pt, int pl, int pb, int pr,
                int group, const float *scale, const float *shift, const float *res, int act, double alpha) {
    PL_REQUIRE(ctx && x && w && y, PL_EINVAL, "conv2d: null pointer");
    PL_REQUIRE(N >= 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && kh > 0 && kw > 0, PL_EINVAL, "conv2d: bad shape");
    PL_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && pt >= 0 && pl >= 0 && group > 0, PL_EINVAL, "conv2d: bad parameter");
    // util.pad only honours pads[0]/pads[1] (util.py:8): anything else is undefined there
    PL_REQUIRE(pt == pb && pl == pr, PL_EUNSUPPORTED, "asymmetric pads are undefined in the reference (util.py:8)");
    PL_REQUIRE(Cin % group == 0 && Cout % group == 0, PL_EUNSUPPORTED, "group must divide Cin and Cout");
    PL_REQUIRE(act >= 0 && act <= 2, PL_EINVAL, "conv2d: bad activation code");
    const int Ho = (H + pt + pb - (kh - 1) * dh - 1 + sh) / sh;  // util.py:25
    const int Wo = (W + pl + pr - (kw - 1) * dw - 1 + sw) / sw;  // util.py:26
    PL_REQUIRE(Ho > 0 && Wo > 0, PL_EINVAL, "conv2d: empty output (%d x %d)", Ho, Wo);
    PL_REQUIRE(H + 2 * pt < 16384 && W + 2 * pl < 16384 && kh * dh < 16384 && kw * dw < 16384, PL_EUNSUPPORTED,
               "conv2d: spatial extent above 16383");
    if (N == 0) return PL_OK;
    const size_t out_elems = (size_t)N * Cout * Ho * Wo, in_elems = (size_t)N * Cin * H * W;
    const size_t w_elems = (size_t)Cout * (Cin / group) * kh * kw;
    PL_REQUIRE(out_elems < (1ull << 31) && in_elems <= (1ull << 29) && w_elems <= (1ull << 29),
               PL_EUNSUPPORTED, "conv2d: input/filter above 2 GiB or output above 2^31 elements");
    CtxGuard guard(ctx);

    ConvArgs a;
    a.x = x; a.w = w; a.y = y;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.Ho = Ho; a.Wo = Wo;
    a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw; a.pt = pt; a.pl = pl;
    a.groups = group; a.cin_g = Cin / group; a.cout_g = Cout / group;
    a.K = a.cin_g * kh * kw;
    a.cols = N * Ho * Wo;
    a.HoWo = Ho * Wo; a.HW = H * W;
    a.slab = out_elems;
    a.x_bytes = (int)(in_elems * 4); a.w_bytes = (int)(w_elems * 4);
    a.divKhw = FastDiv(kh * kw); a.divKw = FastDiv(kw);
    a.divHoWo = FastDiv(a.HoWo); a.divWo = FastDiv(Wo);
    a.ep = Epilogue{bias, scale, shift, res, act, (float)alpha, (float)(1.0 - alpha)};

    int cfg, splits;
    choose_config(ctx, a.cout_g, a.cols, a.K, group, cfg, splits);
    if (ctx->conv_cfg >= 0 && ctx->conv_cfg < kNumCfgs) cfg = ctx->conv_cfg;
    if (ctx->conv_split_k > 0) splits = ctx->conv_split_k;
    const CfgInfo &ci = kCfgs[cfg];
    // split boundaries on BK multiples so every chunk start keeps float4 alignment
    int kps = ((a.K + splits - 1) / splits + ci.bk - 1) / ci.bk * ci.bk;
    splits = (a.K + kps - 1) / kps;
    a.splits = splits; a.k_per_split = kps;
    a.mtiles = (a.cout_g + ci.bm - 1) / ci.bm;
    a.ntiles = (a.cols + ci.bn - 1) / ci.bn;
    a.tiles = a.mtiles * a.ntiles;
    a.divMt = FastDiv(a.mtiles);

    float *ws = nullptr;
    if (splits > 1) {
        int rc = pl_alloc(ctx, (size_t)splits * out_elems * sizeof(float), (void **)&ws);
        if (rc != PL_OK) return rc;
        a.y = ws;
    }
    const bool avec = (a.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15u) == 0);
    dim3 grid((unsigned)(a.tiles * group), (unsigned)splits);
    auto kern = avec ? ci.vec : ci.scl;
    hipLaunchKernelGGL(kern, grid, dim3(256), ci.lds, ctx->stream, a);
    hipError_t le = hipGetLastError();
    int rc = PL_OK;
    if (le != hipSuccess) {
        pl_set_error("conv_igemm launch (%s): %s", ci.name, hipGetErrorString(le));
        rc = PL_EHIP;
    }
    if (splits > 1) {
        if (rc == PL_OK)
            rc = pl_splitk_reduce_f32(ctx, ws, splits, y, N, Cout, a.HoWo, bias, scale, shift, res, act, alpha);
        pl_free(ctx, ws);  // stream-ordered: safe to recycle after the enqueue
    }
    return rc;
}

}  // namespace

extern "C" {

int pl_conv2d_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w, int Cout, int kh,
                  int kw, const float *bias, float *y, int sh, int sw, int dh, int dw, int pt, int pl, int pb,
                  int pr, int group) {
    return conv_launch(ctx, x, N, Cin, H, W, w, Cout, kh, kw, bias, y, sh, sw, dh, dw, pt, pl, pb, pr, group,
                       nullptr, nullptr, nullptr, PL_ACT_NONE, 0.0);
}

int pl_conv2d_fused_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W, const float *w, int Cout, int kh,
                        int kw, const float *bias, float *y, int sh, int sw, int dh, int dw, int pt, int pl, int pb,
                        int pr, int group, const float *scale, const float *shift, const float *res, int act,
                        double alpha) {
    return conv_launch(ctx, x, N, Cin, H, W, w, Cout, kh, kw, bias, y, sh, sw, dh, dw, pt, pl, pb, pr, group, scale,
                       shift, res, act, alpha);
}

int pl_conv2d_set_config(pl_ctx *ctx, int cfg, int split_k) {
    PL_REQUIRE(ctx, PL_EINVAL, "null ctx");
    PL_REQUIRE(cfg < kNumCfgs, PL_EINVAL, "config %d out of range (%d)", cfg, kNumCfgs);
    ctx->conv_cfg = cfg;
    ctx->conv_split_k = split_k;
    return PL_OK;
}

int pl_conv2d_num_configs(void) { return kNumCfgs; }

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
    if (trans_b)
        return conv_launch(ctx, a, M, K, 1, 1, b, N, 1, 1, bias, y, 1, 1, 1, 1, 0, 0, 0, 0, 1, nullptr, nullptr,
                           nullptr, PL_ACT_NONE, 0.0);
    PL_REQUIRE(!bias, PL_EUNSUPPORTED, "pl_gemm_f32: bias needs trans_b=1");
    return conv_launch(ctx, b, 1, K, 1, N, a, M, 1, 1, nullptr, y, 1, 1, 1, 1, 0, 0, 0, 0, 1, nullptr, nullptr,
                       nullptr, PL_ACT_NONE, 0.0);
}

}  // extern "C"
