// Fully fused Winograd F(4x4,3x3) convolution on channel-quad tensors -- included by conv_winograd.hip inside its
// anonymous namespace (uses Epilogue, FastDiv, apply_epilogue4).
//
// For 3x3 / stride 1 / pad 1 / group 1 convs (util.conv_for, util.py:17-44, with the fused tail of layer.py:125-127,
// 93-95, 44-51).  The staged pipeline (input transform -> 36 grouped GEMMs -> output transform) needs 4x fewer
// multiplies than the direct conv but moves V and M -- 2.25x the activation each, written and read -- through
// memory; the fused 1-D kernel (conv_w1d4_kernel) moves nothing extra but only halves the multiplies.  Here ONE
// workgroup carries a block of 32 tiles x 64 output channels through all 36 frequencies:
//   * 8 waves, two per SIMD, 256 registers each (amdgpu_waves_per_eu(2,2)).  Wave (wm, wn) owns output channels
//     [16 wm, 16 wm + 16) x tiles [16 wn, 16 wn + 16) as 36 accumulator blocks of v_mfma_f32_16x16x4_f32 (144 registers):
//     lane (i = lane % 16, rg = lane / 16) ends up holding, for ITS tile i and ITS channel quad 4 rg .. 4 rg + 3, all 36
//     frequencies -- so the output transform A^T m A is lane-local; the fused tail and the stores go through a wave-private
//     LDS exchange that turns (tile, quad) lanes into (tile, pixel) lanes: one store covers a contiguous 1 KB run.  M never
//     exists.  (The first version had 4 waves x 288 accumulators: 67 us against 48 us -- the compiler shuffled through AGPRs.)
//   * Round 6, what ships: BLOCKS OF 16 TILES ON FOUR WAVES (template parameter HALF; PLANER_HIP_EXPERIMENT=wf4_half=0 brings the
//     32-tile / eight-wave block described below back).  A workgroup then holds half the registers of a CU and 34.8 KB of LDS,
//     so two of them -- or one and another kernel's workgroups -- share a CU with barriers of their own: one's prologue, store
//     tail and barrier waits run under the other's K loop.  Wave wm owns output channels [16 wm, 16 wm + 16) x the 16 tiles; the
//     patch transform of a step is three TWO-ROW wave-items (wf4_transform_2rows: lane = row of a pair x tile x channel pair) on
//     waves 1-3.  Layer2 conv of ResNet-18 at batch 32: 43.7 against 62.1 us alone (224 instead of 112 workgroups), bench.py
//     58.1-58.4 k against 55.9-56.2 k img/s (profiles/r06_ab_wf4_half_blocks.txt).
//   * The filter never touches LDS (round 6, WF4_GLOBAL_A): it is laid out [cout block][chunk][16-channel block][group of 4
//     frequencies][lane][4], so the fragment a wave needs for one group of four MFMAs is ONE 16-byte load per lane over a
//     contiguous 1 KB, requested five groups ahead into a ring of six register slots (profiles/r06_wf4_stalls.md: 39.2 / 60.2
//     against 40.3 / 62.6 us per layer1 / layer2 conv, 110 KB less LDS traffic per K step, 69.6 instead of 143 KB of LDS).
//     (WF4_GLOBAL_A = 0, the form of rounds 3-5: the filter slice A[4 cout blocks][4 k][16][36] of a chunk, 36.9 KB, staged in
//     LDS by LDS-DMA and read back as fragments -- what the next item still describes.)
//   * K runs over input channel quads (one quad = one chunk = one MFMA K step of 4).  Per chunk the workgroup
//     holds in LDS: the filter slice A[4 cout blocks][4 k][16][36] (36.9 KB, by LDS-DMA straight from a filter laid out in
//     exactly that order), the input patch P of the block's tiles -- (4 BR + 2) x (4 BC + 2) pixels per image,
//     halo shared between neighbouring tiles, stored [row][x mod 4][x div 4] as 16-byte cells (by LDS-DMA, zero fill by the
//     range check) -- and the transformed patch V[2 wn][4 k][16 tiles][36] (18.4 KB) that six wave-items compute from P
//     (one row of B^T d B for all 32 tiles x a channel pair per lane, packed arithmetic): V never leaves the CU.
//   * One barrier per chunk.  In iteration c the waves request chunk c+1's filter slice and chunk c+2's patch (LDS-DMA,
//     one request per MFMA group), waves 4-7 transform P[(c+1)&1] into V[(c+1)&1] BEFORE their 36 MFMAs of chunk c and waves
//     0-1 after theirs (rows 4-5), so that the two waves of a SIMD alternate on its matrix pipe; every buffer written in an
//     interval was last read in the previous one.
//   * Fragment reads are conflict free by layout: a lane's 36 frequencies of one (k, row) are 144 consecutive bytes.
//   * What a K step costs (profiles/r04_wf4_knockout.md): 1.00 us of MFMAs + 0.36 fragment reads + 0.38 transform + 0.13 both
//     LDS-DMA streams = 1.61 us; fp32 MFMAs do not overlap with the other vector instructions of their SIMD.
// Executed MFMA work = 36/16 of a GEMM per output pixel = 4x fewer multiplies than the direct conv (tile and
// channel padding aside); HBM traffic = x + y (+ residual) + the filter.

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wf4Args {
    const float *x;        // [N][Cq][H][W][4]
    const float *u;        // [cout block of 64][chunk = cin/4][36][4 blocks of 16][4 k][16]   (pl_conv2d_prepare_wf4_f32)
    float *y;              // [N][Coq][H][W][4]
    int N, Cq, Coq, H, W, th, tw;
    int nchunks;           // Cin / 4
    int lBR, lBC;          // log2(tile rows), log2(tile columns) of a block; images per block NB = 32 >> (lBR + lBC)
    int R, S;              // patch rows 4 BR + 2; 16-byte cells per x phase BC + 1
    int cells;             // NB * R * 4 * S cells of one patch
    int rblocks, cblocks, cout_blocks;
    // PACK: the block's spare slot columns (BC - tw of them, one image per block, one column block) carry tiles of a DONOR image --
    // of every G = tw / sc + 1 images the last one is cut into (th / BR) x (tw / sc) groups of BR x sc tiles, one per block of the
    // other G - 1: no padding slots (a 56-pixel map: 196 real tiles in 7 blocks of 32 per image -> 49 blocks per 8 images instead of 56)
    int pack_sc, pack_g, pack_gc;      // spare columns, images per group, donor groups per tile row (tw / sc); 0: off
    unsigned x_bytes, u_bytes, y_bytes;
    FastDiv divPlane, div4S, divS, divCoB, divCb, divRb;
    Epilogue ep;
};

// probe builds only (tools/wf4_knock.sh): bit 0 no filter LDS-DMA, 1 no patch LDS-DMA, 2 no patch transform, 3 no MFMAs,
// 4 no fragment reads, 5 no output rows -- results are garbage, the time tells what a K step's parts cost
#ifndef WF4_KNOCK
#define WF4_KNOCK 0
#endif
// 1: the filter fragments go global -> REGISTERS (16-byte loads that a wave issues as one contiguous 1 KB run, a rolling
// prefetch WF4_GA_AHEAD MFMA groups deep) and never touch LDS; 0: the filter slice of a K step is staged in LDS by LDS-DMA and
// read back as fragments (two readers per value).  The filter layout differs (pl_conv2d_prepare_wf4_f32 follows the same macro).
#ifndef WF4_GLOBAL_A
#define WF4_GLOBAL_A 1
#endif
#ifndef WF4_HALF_TRIM
#define WF4_HALF_TRIM 1      // (probe builds: 0 keeps the 32-tile block's LDS allocation for the 16-tile block)
#endif
#ifndef WF4_GA_AHEAD
#define WF4_GA_AHEAD 5
#endif
#ifndef WF4_GA_SLOTS
#define WF4_GA_SLOTS 6
#endif
// probe builds only (-DWF4_STAMP, tools/wf4_stamp.py): s_memtime of wave 0 / wave 4 of every block at kernel entry, after chunk 0
// has landed, after the prologue, after the K loop and after the last output row
#ifdef WF4_STAMP
__device__ unsigned long long wf4_stamps[4096 * 2 * 8];
#define WF4_MARK(i)                                                                                          \
    do {                                                                                                     \
        if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 4096)                                       \
            wf4_stamps[(blockIdx.x * 2 + (wave >> 2)) * 8 + (i)] = __builtin_amdgcn_s_memtime();             \
    } while (0)
// ... and inside K steps 6 and 7 of every block, per wave: step entry, transform_first done, MFMAs done, transform_last done,
// barrier passed
__device__ unsigned long long wf4_step_stamps[512 * 8 * 2 * 8];
#define WF4_STEP_MARK(c, i)                                                                                  \
    do {                                                                                                     \
        if (((c) == 6 || (c) == 7) && lane == 0 && blockIdx.x < 512)                                          \
            wf4_step_stamps[((blockIdx.x * 8 + wave) * 2 + ((c) - 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define WF4_MARK(i)
#define WF4_STEP_MARK(c, i)
#endif
constexpr int WF4_A_FLOATS = 36 * 256, WF4_V_FLOATS = 2 * 36 * 64, WF4_P_PASSES = 2;      // patch passes of 512 cells
constexpr int WF4_P_CELLS = 1024, WF4_P_FLOATS = WF4_P_CELLS * 4;
constexpr int WF4_LDS_BYTES = (2 * WF4_A_FLOATS + 2 * WF4_V_FLOATS + 2 * WF4_P_FLOATS) * 4;

__device__ __forceinline__ void wf4_bt(const float (&d)[6], float (&o)[6]) {      // o = B^T d, with fused multiply-adds
    const float s = d[4] - d[2], t = d[3] - d[1];
    o[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
    o[1] = __builtin_fmaf(-4.f, d[1] + d[2], d[3] + d[4]);
    o[2] = __builtin_fmaf(4.f, d[1] - d[2], d[4] - d[3]);
    o[3] = __builtin_fmaf(2.f, t, s);
    o[4] = __builtin_fmaf(-2.f, t, s);
    o[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
}
template <int A>
__device__ __forceinline__ float wf4_bt_row(const float (&d)[6]) {
    if constexpr (A == 0) return __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
    else if constexpr (A == 1) return __builtin_fmaf(-4.f, d[1] + d[2], d[3] + d[4]);
    else if constexpr (A == 2) return __builtin_fmaf(4.f, d[1] - d[2], d[4] - d[3]);
    else if constexpr (A == 3) return __builtin_fmaf(2.f, d[3] - d[1], d[4] - d[2]);
    else if constexpr (A == 4) return __builtin_fmaf(-2.f, d[3] - d[1], d[4] - d[2]);
    else return __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
}

// one transform wave-item: row A of B^T d B for (tile, channel), lanes = 16 tiles x 4 channels (tile fastest): 6 columns x
// (the rows of d with a non-zero coefficient) out of the channel-planar patch, 6 consecutive values into
// V[half][ch][i][6 A .. 6 A + 5].  pbase: float index of the tile's first patch cell in its channel plane; vbase: float index
// of V[half][ch][i][0]; rs = cells per patch row (4 S), ps = cells per x phase (S).
template <int A, int rs, int ps, int cst>
__device__ __forceinline__ void wf4_transform_row(const float *P, float *V, int pbase, int vbase) {
    // all the patch values this row needs are requested before the first is used (the rows of d with a zero coefficient in
    // row A of B^T are never read: the compiler drops those loads)
    float d[6][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * cst;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            constexpr bool used[6][6] = {{1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0},
                                         {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0, 1, 0, 1}};
            d[b][k] = used[A][k] ? col[k * rs] : 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    float m[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) m[b] = wf4_bt_row<A>(d[b]);
    float o[6];
    wf4_bt(m, o);
    float2 *dst = reinterpret_cast<float2 *>(V + vbase + A * 6);       // vbase is a multiple of 36, 6 A is even: 8-byte aligned
    dst[0] = make_float2(o[0], o[1]);
    dst[1] = make_float2(o[2], o[3]);
    dst[2] = make_float2(o[4], o[5]);
}

// The same row for TWO channels per lane (16-byte-cell patch layout): every value is a channel pair out of one 8-byte LDS read
// and every operation a packed one -- half the vector instructions of the one-channel form and no operand shuffling.  Lanes =
// 32 tiles x 2 channel pairs (pair fastest: consecutive lanes read consecutive 8 bytes).  Same formulas, same roundings.
typedef float wf4_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wf4_v2 wf4_fma2(float a, wf4_v2 b, wf4_v2 c) { return __builtin_elementwise_fma((wf4_v2){a, a}, b, c); }
__device__ __forceinline__ void wf4_bt2(const wf4_v2 (&d)[6], wf4_v2 (&o)[6]) {
    const wf4_v2 s = d[4] - d[2], t = d[3] - d[1];
    o[0] = wf4_fma2(4.f, d[0], wf4_fma2(-5.f, d[2], d[4]));
    o[1] = wf4_fma2(-4.f, d[1] + d[2], d[3] + d[4]);
    o[2] = wf4_fma2(4.f, d[1] - d[2], d[4] - d[3]);
    o[3] = wf4_fma2(2.f, t, s);
    o[4] = wf4_fma2(-2.f, t, s);
    o[5] = wf4_fma2(4.f, d[1], wf4_fma2(-5.f, d[3], d[5]));
}
template <int A>
__device__ __forceinline__ wf4_v2 wf4_bt_row2(const wf4_v2 (&d)[6]) {
    if constexpr (A == 0) return wf4_fma2(4.f, d[0], wf4_fma2(-5.f, d[2], d[4]));
    else if constexpr (A == 1) return wf4_fma2(-4.f, d[1] + d[2], d[3] + d[4]);
    else if constexpr (A == 2) return wf4_fma2(4.f, d[1] - d[2], d[4] - d[3]);
    else if constexpr (A == 3) return wf4_fma2(2.f, d[3] - d[1], d[4] - d[2]);
    else if constexpr (A == 4) return wf4_fma2(-2.f, d[3] - d[1], d[4] - d[2]);
    else return wf4_fma2(4.f, d[1], wf4_fma2(-5.f, d[3], d[5]));
}
// pbase: float index of the pair's first value in the tile's first patch cell; vbase: float index of V[half][even channel][i][0]
// (the odd channel's row is 16 x 36 floats further on); rs = floats per patch row, ps = floats per x phase
template <int A, int rs, int ps>
__device__ __forceinline__ void wf4_transform_row2(const float *P, float *V, int pbase, int vbase) {
    wf4_v2 d[6][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            constexpr bool used[6][6] = {{1, 0, 1, 0, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0},
                                         {0, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0, 1, 0, 1}};
            d[b][k] = used[A][k] ? *reinterpret_cast<const wf4_v2 *>(col + k * rs) : (wf4_v2){0.f, 0.f};
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    wf4_v2 m[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) m[b] = wf4_bt_row2<A>(d[b]);
    wf4_v2 o[6];
    wf4_bt2(m, o);
    float *v0 = V + vbase + A * 6, *v1 = v0 + 16 * 36;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        v0[j] = o[j].x;
        v1[j] = o[j].y;
    }
}

// TWO rows of B^T d B per wave-item, for blocks of 16 tiles (HALF): lane = [row select r][tile, 4 bits][channel pair].  The rows of a
// pair share their loads up to a per-lane base (KIND 0: rows 0 / 5 read patch rows r, r + 2, r + 4) or read the same four patch
// rows with per-lane coefficients (KIND 1: rows 1 / 2, KIND 2: rows 3 / 4 -- each is c1 d1 + c2 d2 + c3 d3 + d4): three wave-items
// transform a 16-tile block where the one-row form would need six half-empty ones.  pbase: as for wf4_transform_row2, plus r rows
// for KIND 0; vbase: float index of V[even channel][tile][6 A(r)].
template <int KIND, int rs, int ps>
__device__ __forceinline__ void wf4_transform_2rows(const float *P, float *V, int pbase, int vbase, int r) {
    constexpr int NR = KIND == 0 ? 3 : 4;
    wf4_v2 d[6][NR];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const float *col = P + pbase + (b & 3) * ps + (b >> 2) * 4;
#pragma unroll
        for (int k = 0; k < NR; ++k)
            d[b][k] = *reinterpret_cast<const wf4_v2 *>(col + (KIND == 0 ? 2 * k : k + 1) * rs);
    }
    __builtin_amdgcn_sched_barrier(0);
    wf4_v2 m[6];
    if constexpr (KIND == 0) {
#pragma unroll
        for (int b = 0; b < 6; ++b) m[b] = wf4_fma2(4.f, d[b][0], wf4_fma2(-5.f, d[b][1], d[b][2]));
    } else {
        // rows 1 / 2: (-4, -4, 1) / (4, -4, -1);  rows 3 / 4: (-2, -1, 2) / (2, -1, -2)
        const float c1 = KIND == 1 ? (r ? 4.f : -4.f) : (r ? 2.f : -2.f);
        const float c2 = KIND == 1 ? -4.f : -1.f;
        const float c3 = KIND == 1 ? (r ? -1.f : 1.f) : (r ? -2.f : 2.f);
#pragma unroll
        for (int b = 0; b < 6; ++b) m[b] = wf4_fma2(c1, d[b][0], wf4_fma2(c2, d[b][1], wf4_fma2(c3, d[b][2], d[b][3])));
    }
    wf4_v2 o[6];
    wf4_bt2(m, o);
    float *v0 = V + vbase, *v1 = v0 + 16 * 36;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        v0[j] = o[j].x;
        v1[j] = o[j].y;
    }
}

// A^T m A on channel PAIRS (the accumulator's registers 0-1 and 2-3 are natural pairs): packed arithmetic throughout, the
// operations and their order are those of w4_at4_row / w4_at4.
template <int A>
__device__ __forceinline__ wf4_v2 wf4_at_row2(const wf4_v2 (&m)[6]) {
    if constexpr (A == 0) return (m[0] + (m[1] + m[2])) + (m[3] + m[4]);
    else {
        const wf4_v2 q = m[1] - m[2], t = m[3] - m[4], pp = m[1] + m[2], r = m[3] + m[4];
        if constexpr (A == 1) return q + 2.f * t;
        else if constexpr (A == 2) return pp + 4.f * r;
        else return (q + 8.f * t) + m[5];
    }
}
__device__ __forceinline__ void wf4_at2(const wf4_v2 (&m)[6], wf4_v2 (&o)[4]) {
    const wf4_v2 pp = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], t = m[3] - m[4];
    o[0] = (m[0] + pp) + r;
    o[1] = q + 2.f * t;
    o[2] = pp + 4.f * r;
    o[3] = (q + 8.f * t) + m[5];
}
// Coalesced form of the row for the tail the convolutions of a residual net carry (per-channel scale and shift, [residual,]
// ReLU, no bias).  In the MFMA's C layout a lane owns one tile x one channel quad, so a 16-byte
// store instruction scatters 64 pieces at 64-byte strides over four channel planes -- the end-of-kernel store burst of all
// workgroups ran at ~2 TB/s.  Here the row goes through a wave-private LDS buffer [b][quad][tile] (68-cell rows: the padding
// makes the read-back conflict-free) and comes back as lane = (tile, pixel b of the tile's row): one instruction then writes
// (and, for the residual, reads) 16 tiles x 4 pixels = one contiguous 1 KB run of a channel plane.  Scale and shift are applied
// before the exchange (a lane knows its quad's parameters there), residual and ReLU after it: the same operations in the same
// order on every value.
// PLAIN: that tail written straight (packed operations, no per-element option selects); otherwise the general fused tail in
// the same two halves -- bias / scale / shift before the exchange, residual / activation / late residual after it.
// second half of an output row: the six column sums s[b] (pairs lo / hi of the lane's channel quad) -> the row's four pixels,
// tail, exchange, stores
template <bool RES, bool PLAIN>
__device__ __forceinline__ void wf4_row_finish(const Wf4Args &p, const wf4_v2 (&slo)[6], const wf4_v2 (&shi)[6], int cqc, float4 scale,
                                               float4 shift, float4 *xb, int wr_cell, int rd_cell,
                                               const __amdgpu_buffer_rsrc_t yrsrc, const __amdgpu_buffer_rsrc_t rrsrc,
                                               const int (&off)[4]) {
    float4 rs[4];
    if constexpr (RES) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[q], 0, 0));
    }
    wf4_v2 olo[4], ohi[4];
    wf4_at2(slo, olo);
    wf4_at2(shi, ohi);
    if constexpr (PLAIN) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            olo[b] = olo[b] * (wf4_v2){scale.x, scale.y} + (wf4_v2){shift.x, shift.y};
            ohi[b] = ohi[b] * (wf4_v2){scale.z, scale.w} + (wf4_v2){shift.z, shift.w};
        }
    }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the general form fetches its quad's parameters here, row by row: twelve more registers held across the whole epilogue
    //  would spill beside the accumulators)
    float4 gbias = z4, gscale = z4, gshift = z4;
    if constexpr (!PLAIN) {
        if (p.ep.bias) gbias = reinterpret_cast<const float4 *>(p.ep.bias)[cqc];
        if (p.ep.scale) gscale = reinterpret_cast<const float4 *>(p.ep.scale)[cqc];
        if (p.ep.shift) gshift = reinterpret_cast<const float4 *>(p.ep.shift)[cqc];
    }
    Epilogue affine = p.ep, rest = p.ep;          // the general tail's two halves (apply_epilogue4 skips what is null / 0)
    affine.res = nullptr; affine.act = 0;
    rest.bias = rest.scale = rest.shift = nullptr;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float4 o = make_float4(olo[b].x, olo[b].y, ohi[b].x, ohi[b].y);
        if constexpr (!PLAIN) o = apply_epilogue4(affine, gbias, gscale, gshift, z4, 4, o);
        xb[b * 68 + wr_cell] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = xb[rd_cell + q * 16];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // the next row overwrites the buffer
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 o;
        if constexpr (PLAIN) {
            wf4_v2 lo = (wf4_v2){v[q].x, v[q].y}, hi = (wf4_v2){v[q].z, v[q].w};
            if constexpr (RES) {
                lo = lo + (wf4_v2){rs[q].x, rs[q].y};
                hi = hi + (wf4_v2){rs[q].z, rs[q].w};
            }
            const wf4_v2 zl = lo * 0.f, zh = hi * 0.f;     // relu_ref: x > 0 ? x : x * 0
            o = make_float4(lo.x > 0.f ? lo.x : zl.x, lo.y > 0.f ? lo.y : zl.y, hi.x > 0.f ? hi.x : zh.x, hi.y > 0.f ? hi.y : zh.y);
        } else {
            o = apply_epilogue4(rest, z4, z4, z4, RES ? rs[q] : z4, 4, v[q]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o),
                                               yrsrc, off[q], 0, 0);
    }
}

template <int A, bool RES, bool PLAIN>
__device__ __forceinline__ void wf4_output_row_coalesced(const Wf4Args &p, const f32x4 (&acc)[36], int cqc, float4 scale,
                                                         float4 shift, float4 *xb, int wr_cell, int rd_cell,
                                                         const __amdgpu_buffer_rsrc_t yrsrc, const __amdgpu_buffer_rsrc_t rrsrc,
                                                         const int (&off)[4]) {
    wf4_v2 slo[6], shi[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        wf4_v2 m[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) m[a] = __builtin_shufflevector(acc[a * 6 + b], acc[a * 6 + b], 0, 1);
        slo[b] = wf4_at_row2<A>(m);
#pragma unroll
        for (int a = 0; a < 6; ++a) m[a] = __builtin_shufflevector(acc[a * 6 + b], acc[a * 6 + b], 2, 3);
        shi[b] = wf4_at_row2<A>(m);
    }
    wf4_row_finish<RES, PLAIN>(p, slo, shi, cqc, scale, shift, xb, wr_cell, rd_cell, yrsrc, rrsrc, off);
}

// Variants (compile-time; the launcher picks one): DMA_A -- the filter slice goes global -> LDS by LDS-DMA instead of through
// registers; PLANAR -- the patch is stored channel-planar (transform lanes = 16 tiles x 4 channels, tile fastest: conflict-free
// patch reads AND 2-way instead of 4-way conflicts on the V writes) instead of as 16-byte cells (lanes channel fastest);
// STAGGER -- waves 4-7 transform before their MFMAs and waves 0-3 after, so the two waves of a SIMD alternate on its matrix pipe.
// HALF (round 6, needs WF4_GLOBAL_A): a workgroup of FOUR waves carries 16 tiles x 64 output channels -- half the registers and
// a quarter of the LDS of a CU, so two workgroups share it with barriers of their own: one's prologue, tail and barrier waits
// run under the other's K loop.  Three two-row transform items (waves 1-3) per K step.
template <bool DMA_A, bool PLANAR, bool STAGGER, int LBC, bool PACK = false, bool HALF = false>
__device__ __forceinline__ void conv_wf4_body(const Wf4Args &p) {
    constexpr int NT = HALF ? 256 : 512, LT = HALF ? 4 : 5;          // threads, log2(tiles) of a block
    static_assert(!HALF || (WF4_GLOBAL_A && DMA_A && !PLANAR), "half blocks: filter in registers, 16-byte-cell patch");
    // the LDS-DMA requests of a step go out one per MFMA group (measured: 40.9 -> 39.8 us per layer1 conv against all seven
    // in a row at the head of the step)
    constexpr bool SPREAD = DMA_A && !PLANAR && STAGGER;
    constexpr int S = (1 << LBC) + 1 + (PACK ? 1 : 0);      // 16-byte cells per x phase: compile time, so every patch read is base + immediate
    static_assert(!PACK || !PLANAR, "packed blocks: 16-byte-cell patch layout only");
    // six separate LDS objects (not one dynamic array): the compiler orders LDS-DMA against later LDS accesses object by
    // object, so a DMA into A1 / P0 does not hold up the reads of A0 / P1 / V0 and the writes of V1
#if WF4_GLOBAL_A
    float *const As0 = nullptr, *const As1 = nullptr;          // (no filter slice in LDS: 73.7 KB fewer)
#else
    __shared__ __attribute__((aligned(16))) float As0[WF4_A_FLOATS], As1[WF4_A_FLOATS];      // [4 cb][4 k][16 i][36 f]
#endif
    // (HALF: one tile half of V, 512 patch cells -- 34.8 KB a workgroup, so that what else runs on the CU keeps its LDS)
    __shared__ __attribute__((aligned(16))) float Vs0[WF4_V_FLOATS / (HALF && WF4_HALF_TRIM ? 2 : 1)], Vs1[WF4_V_FLOATS / (HALF && WF4_HALF_TRIM ? 2 : 1)];      // [2 wn][4 k][16 i][36 f]
    __shared__ __attribute__((aligned(16))) float Ps0[WF4_P_FLOATS / (HALF && WF4_HALF_TRIM ? 2 : 1)], Ps1[WF4_P_FLOATS / (HALF && WF4_HALF_TRIM ? 2 : 1)];      // [cells][4]  or  [4 ch][cells]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = HALF ? wave : wave >> 1, wn = HALF ? 0 : wave & 1;       // 16 output channels x 16 tiles per wave
    const int li = lane & 15, lk = lane >> 4;      // MFMA: operand row / column i, k index (A, B) or row group (C)

    // ---- which block: cout block fastest (blocks that share input pixels are neighbours), then columns, rows, images ----
    unsigned t1, coutblk, t2, colblk, ngrp, rowblk;
    p.divCoB.divmod(blockIdx.x, t1, coutblk);
    p.divCb.divmod(t1, t2, colblk);
    p.divRb.divmod(t2, ngrp, rowblk);
    const int BRm = (1 << p.lBR) - 1, BCm = (1 << LBC) - 1, lT = p.lBR + LBC;
    int n0 = (int)ngrp << (LT - lT);
    const int ty0 = (int)rowblk << p.lBR, tx0 = (int)colblk << LBC;
    // PACK: ngrp counts the RECEIVING images (G - 1 of every G); the donor's tile group of this block: (dgr, dgc)
    int dn = 0, dgr = 0, dgc = 0;
    if constexpr (PACK) {
        const int grp = (int)ngrp / (p.pack_g - 1), i = (int)ngrp - grp * (p.pack_g - 1);
        n0 = grp * p.pack_g + i;
        dn = grp * p.pack_g + p.pack_g - 1;
        const int id = i * p.rblocks + (int)rowblk;
        dgr = id / p.pack_gc;
        dgc = id - dgr * p.pack_gc;
    }
    const int HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, p.u_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((address_space(3))) float lds_float;

    // ---- patch cells this thread fetches every chunk (16 bytes = the 4 channels of a pixel; the channel quad rides in the
    //      scalar offset) ----
    int pvoff[WF4_P_PASSES];
#pragma unroll
    for (int ps = 0; ps < WF4_P_PASSES; ++ps) {
        const unsigned ci = (unsigned)(ps * NT + tid);
        pvoff[ps] = OOB;
        if (ci < (unsigned)p.cells) {
            unsigned nb, rem, r_, rem2, m, s;
            p.divPlane.divmod(ci, nb, rem);
            p.div4S.divmod(rem, r_, rem2);
            p.divS.divmod(rem2, m, s);
            int n = n0 + (int)nb, h = 4 * ty0 + (int)r_ - 1, w = 4 * tx0 + (int)(4 * s + m) - 1;
            if constexpr (PACK) {
                // plane columns from 4 (tw + 1) on hold the donor group's patch (one cell right of where the spare slots would
                // read by themselves: the main image's last tile column shares its two halo columns with nobody)
                const int pc = (int)(4 * s + m) - 4 * (p.tw + 1);
                if (pc >= 0) {
                    n = dn;
                    h = (4 * dgr << p.lBR) + (int)r_ - 1;
                    w = 4 * p.pack_sc * dgc + pc - 1;
                    if (pc > 4 * p.pack_sc + 1) n = p.N;           // (past the donor patch: nothing)
                }
            }
            if (n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
                pvoff[ps] = (int)(((unsigned)(n * p.Cq) * (unsigned)HW + (unsigned)(h * p.W + w)) << 4);
        }
    }
    float4 preg[WF4_P_PASSES], areg[5];
    constexpr bool DMA_P = DMA_A && !PLANAR;            // 16-byte cells land lane-linear: LDS-DMA can write them directly
    auto load_p = [&](int c, int buf) {
        const int soff = (c * HW) << 4;
        float *Pb = buf ? Ps1 : Ps0;
#pragma unroll
        for (int ps = 0; ps < WF4_P_PASSES; ++ps)
            if (ps * NT < p.cells) {
                if constexpr (DMA_P) {
                    if (ps * NT + tid < p.cells)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_float *)(Pb + (ps * NT + tid - lane) * 4), 16, pvoff[ps], soff, 0, 0);
                } else {
                    preg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, pvoff[ps], soff, 0));
                }
            }
    };
    auto store_p = [&](int buf) {
        if constexpr (DMA_P) return;
        float *Pb = buf ? Ps1 : Ps0;
#pragma unroll
        for (int ps = 0; ps < WF4_P_PASSES; ++ps)
            if (ps * NT < p.cells && ps * NT + tid < p.cells) {
                if constexpr (PLANAR) {
                    float *d = Pb + ps * NT + tid;
                    d[0] = preg[ps].x;
                    d[p.cells] = preg[ps].y;
                    d[2 * p.cells] = preg[ps].z;
                    d[3 * p.cells] = preg[ps].w;
                } else {
                    *reinterpret_cast<float4 *>(Pb + (ps * NT + tid) * 4) = preg[ps];
                }
            }
    };
    // filter slice: 2304 x 16 bytes = four passes of 512 threads and one of 256 (waves 0-3)
    auto load_a = [&](int c, int buf) {
        const int soff = (int)(((unsigned)coutblk * (unsigned)p.nchunks + (unsigned)c) * (unsigned)(WF4_A_FLOATS * 4));
        float *Ab = buf ? As1 : As0;
#pragma unroll
        for (int it = 0; it < 5; ++it)
            if (it < 4 || wave < 4) {
                if constexpr (DMA_A)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_float *)(Ab + (it * 512 + tid - lane) * 4), 16, (it * 512 + tid) << 4, soff, 0, 0);
                else
                    areg[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, (it * 512 + tid) << 4, soff, 0));
            }
    };
    // one piece (1 KB per wave) of the next filter slice / of the patch after next: the K step issues them BETWEEN its MFMA
    // groups, so that the ~100 issue cycles an LDS-DMA instruction costs a wave are spent while its SIMD partner's MFMAs
    // keep the matrix pipe busy (all seven in a row before the first MFMA left the pipe idle at the head of every step)
    auto load_a_piece = [&](int c, int buf, int it) {
        const int soff = (int)(((unsigned)coutblk * (unsigned)p.nchunks + (unsigned)c) * (unsigned)(WF4_A_FLOATS * 4));
        float *Ab = buf ? As1 : As0;
        if constexpr (WF4_KNOCK & 1) return;
        if (it < 4 || wave < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_float *)(Ab + (it * 512 + tid - lane) * 4), 16, (it * 512 + tid) << 4, soff, 0, 0);
    };
    auto load_p_piece = [&](int c, int buf, int ps) {
        const int soff = (c * HW) << 4;
        float *Pb = buf ? Ps1 : Ps0;
        if constexpr (WF4_KNOCK & 2) return;
        if (ps * NT < p.cells && ps * NT + tid < p.cells)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_float *)(Pb + (ps * NT + tid - lane) * 4), 16, pvoff[ps], soff, 0, 0);
    };
    auto store_a = [&](int buf) {
        if constexpr (!DMA_A) {
            float *Ab = buf ? As1 : As0;
#pragma unroll
            for (int it = 0; it < 5; ++it)
                if (it < 4 || wave < 4) *reinterpret_cast<float4 *>(Ab + (it * 512 + tid) * 4) = areg[it];
        }
    };

    // ---- transform: 12 wave-items = 6 rows of B^T d B x 2 halves of the block's tiles; a wave-item's 64 lanes are 16 tiles x
    //      4 channels.  Waves 4-7 take rows 0-3 (both halves), waves 0-3 row 4 or 5 of one half ----
    const int ti = PLANAR ? (lane & 15) : (lane >> 2), tch = PLANAR ? (lane >> 4) : (lane & 3);
    constexpr int rs = PLANAR ? 4 * S : 16 * S, psz = PLANAR ? S : 4 * S, cst = PLANAR ? 1 : 4;
    auto item_bases = [&](int half, int &pbase, int &vbase) {
        const int tj = half * 16 + ti;
        const int t_nb = tj >> lT, t_r = (tj >> LBC) & BRm;
        int t_c = tj & BCm;
        if (PACK && t_c >= p.tw) t_c += 1;                       // spare slots read the donor patch, one cell further right
        pbase = (PLANAR ? tch * p.cells : tch) + (((t_nb * p.R + 4 * t_r) * 4) * S + t_c) * cst;
        vbase = (((half * 4 + tch) * 16) + ti) * 36;
    };
    int pb0, vb0, pb1, vb1;
    item_bases(0, pb0, vb0);
    item_bases(1, pb1, vb1);
    // 16-byte-cell layout: a lane takes a channel PAIR of one of the block's 32 tiles (wf4_transform_row2), so a whole row of
    // B^T d B is ONE wave-item: six per step -- waves 4-7 rows 0-3, waves 0-1 rows 4-5
    int pb2 = 0, vb2 = 0;
    const int hr = lane >> 5;                                    // HALF: which row of its pair this lane transforms
    if constexpr (HALF) {
        const int tj = (lane >> 1) & 15, cp = lane & 1;
        const int t_nb = tj >> lT, t_r = (tj >> LBC) & BRm;
        int t_c = tj & BCm;
        if (PACK && t_c >= p.tw) t_c += 1;
        pb2 = 2 * cp + (((t_nb * p.R + 4 * t_r) * 4) * S + t_c) * 4;
        vb2 = ((2 * cp * 16) + tj) * 36;
    } else if constexpr (!PLANAR) {
        const int tj = lane >> 1, cp = lane & 1;
        const int t_nb = tj >> lT, t_r = (tj >> LBC) & BRm;
        int t_c = tj & BCm;
        if (PACK && t_c >= p.tw) t_c += 1;                       // spare slots read the donor patch, one cell further right
        pb2 = 2 * cp + (((t_nb * p.R + 4 * t_r) * 4) * S + t_c) * 4;
        vb2 = ((((tj >> 4) * 4 + 2 * cp) * 16) + (tj & 15)) * 36;
    }
    auto transform_first = [&](int pbuf, int vbuf) {           // waves 4-7: a whole row (both halves)
        if constexpr (WF4_KNOCK & 4) return;
        const float *Pb = pbuf ? Ps1 : Ps0;
        float *Vb = vbuf ? Vs1 : Vs0;
        if constexpr (HALF) {                                     // waves 1-3: rows 0 / 5, 1 / 2, 3 / 4
            if (wave == 1) wf4_transform_2rows<0, rs, psz>(Pb, Vb, pb2 + hr * rs, vb2 + hr * 30, hr);
            else if (wave == 2) wf4_transform_2rows<1, rs, psz>(Pb, Vb, pb2, vb2 + 6 + hr * 6, hr);
            else if (wave == 3) wf4_transform_2rows<2, rs, psz>(Pb, Vb, pb2, vb2 + 18 + hr * 6, hr);
        } else if constexpr (!PLANAR) {
            if (wave == 4) wf4_transform_row2<0, rs, psz>(Pb, Vb, pb2, vb2);
            else if (wave == 5) wf4_transform_row2<1, rs, psz>(Pb, Vb, pb2, vb2);
            else if (wave == 6) wf4_transform_row2<2, rs, psz>(Pb, Vb, pb2, vb2);
            else if (wave == 7) wf4_transform_row2<3, rs, psz>(Pb, Vb, pb2, vb2);
        } else if (wave == 4) {
            wf4_transform_row<0, rs, psz, cst>(Pb, Vb, pb0, vb0);
            wf4_transform_row<0, rs, psz, cst>(Pb, Vb, pb1, vb1);
        } else if (wave == 5) {
            wf4_transform_row<1, rs, psz, cst>(Pb, Vb, pb0, vb0);
            wf4_transform_row<1, rs, psz, cst>(Pb, Vb, pb1, vb1);
        } else if (wave == 6) {
            wf4_transform_row<2, rs, psz, cst>(Pb, Vb, pb0, vb0);
            wf4_transform_row<2, rs, psz, cst>(Pb, Vb, pb1, vb1);
        } else if (wave == 7) {
            wf4_transform_row<3, rs, psz, cst>(Pb, Vb, pb0, vb0);
            wf4_transform_row<3, rs, psz, cst>(Pb, Vb, pb1, vb1);
        }
    };
    const int pbw = (wave & 1) ? pb1 : pb0, vbw = (wave & 1) ? vb1 : vb0;
    auto transform_last = [&](int pbuf, int vbuf) {            // waves 0-3: row 4 or 5 of one half
        if constexpr (WF4_KNOCK & 4 || HALF) return;
        const float *Pb = pbuf ? Ps1 : Ps0;
        float *Vb = vbuf ? Vs1 : Vs0;
        if constexpr (!PLANAR) {
            if (wave == 0) wf4_transform_row2<4, rs, psz>(Pb, Vb, pb2, vb2);
            else if (wave == 1) wf4_transform_row2<5, rs, psz>(Pb, Vb, pb2, vb2);
        } else {
            if (wave < 2) wf4_transform_row<4, rs, psz, cst>(Pb, Vb, pbw, vbw);
            else if (wave < 4) wf4_transform_row<5, rs, psz, cst>(Pb, Vb, pbw, vbw);
        }
    };

    f32x4 acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = ((wm * 4 + lk) * 16 + li) * 36;      // this lane's 36 frequencies of A[k = lk][channel 16 wm + li]
    const int b_off = ((wn * 4 + lk) * 16 + li) * 36;      // ... of V[k = lk][tile 16 wn + li]
    // spread: the LDS-DMA requests of chunk c + 1's filter slice (into A[buf ^ 1]) and chunk c + 2's patch (into P[buf]) go
    // out one per MFMA group
#if WF4_GLOBAL_A
    // filter fragments: u[cout block][chunk][wm][g = 4 frequencies][lane][4] -- group g of chunk c for this wave is ONE load
    // instruction over a contiguous 1 KB; a ring of AHEAD + 1 register slots, the load for group G + AHEAD goes out when group
    // G's MFMAs do (G counts groups across K steps: 9 per step, two steps unrolled -> 18 = 3 rings of 6)
    constexpr int GA_D = WF4_GA_AHEAD, GA_SLOTS = WF4_GA_SLOTS;
    static_assert(18 % GA_SLOTS == 0, "two K steps (18 groups) must be whole turns of the ring");
    static_assert(GA_D >= 1 && GA_D < GA_SLOTS, "prefetch depth must leave one slot for the group in use");
    float4 ga[GA_SLOTS];
    const int ga_voff = lane << 4;
    auto ga_load = [&](auto slot, int c, int g) {           // (past the last chunk: the range check returns zeros, nothing uses them)
        const int soff = (int)((((unsigned)coutblk * (unsigned)p.nchunks + (unsigned)c) * 36u + (unsigned)(wm * 9 + g)) << 10);
        ga[decltype(slot)::value] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, ga_voff, soff, 0));
    };
    auto mma_ga = [&](auto parity, int c, bool more2) {
        constexpr int par = decltype(parity)::value;
        const float4 *Vp = reinterpret_cast<const float4 *>((par ? Vs1 : Vs0) + b_off);
        float4 fb[3];
        fb[0] = Vp[0];
        fb[1] = Vp[1];
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if (g + 2 < 9) fb[(g + 2) % 3] = Vp[g + 2];
            {
                const int gn = g + GA_D;                    // the group AHEAD: same chunk or the next one
                auto issue = [&](auto sl) { ga_load(sl, gn < 9 ? c : c + 1, gn < 9 ? gn : gn - 9); };
                switch ((9 * par + g + GA_D) % GA_SLOTS) {
                case 0: issue(std::integral_constant<int, 0>{}); break;
                case 1: issue(std::integral_constant<int, 1>{}); break;
                case 2: issue(std::integral_constant<int, 2>{}); break;
                case 3: issue(std::integral_constant<int, 3>{}); break;
                case 4: issue(std::integral_constant<int, 4>{}); break;
                case 5: issue(std::integral_constant<int, 5 % GA_SLOTS>{}); break;
                case 6: issue(std::integral_constant<int, 6 % GA_SLOTS>{}); break;
                case 7: issue(std::integral_constant<int, 7 % GA_SLOTS>{}); break;
                default: issue(std::integral_constant<int, 8 % GA_SLOTS>{}); break;
                }
            }
            const float4 a = ga[(9 * par + g) % GA_SLOTS], b = fb[g % 3];
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[4 * g + 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g < WF4_P_PASSES) {
                if (more2) load_p_piece(c + 2, par, g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#endif
    auto mma = [&](int buf, int c, bool more, bool more2, auto spread) {
        const float4 *Ap = reinterpret_cast<const float4 *>((buf ? As1 : As0) + a_off);
        const float4 *Vp = reinterpret_cast<const float4 *>((buf ? Vs1 : Vs0) + b_off);
        // three rotating fragment slots, requested two groups (8 MFMAs) ahead; the scheduling barriers keep that distance
        float4 fa[3], fb[3];
        constexpr bool NOFRAG = (WF4_KNOCK & 16) != 0, NOMMA = (WF4_KNOCK & 8) != 0;
        if constexpr (NOFRAG) {
            fa[0] = fa[1] = fa[2] = make_float4(1.f, 2.f, 3.f, (float)c);
            fb[0] = fb[1] = fb[2] = make_float4(1.f, 2.f, 3.f, (float)lane);
        } else {
            fa[0] = Ap[0]; fb[0] = Vp[0];
            fa[1] = Ap[1]; fb[1] = Vp[1];
        }
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if (!NOFRAG && g + 2 < 9) {
                fa[(g + 2) % 3] = Ap[g + 2];
                fb[(g + 2) % 3] = Vp[g + 2];
            }
            const float4 a = fa[g % 3], b = fb[g % 3];
            if constexpr (NOMMA) {
                acc[4 * g + 0].x += a.x * b.x; acc[4 * g + 1].x += a.y * b.y; acc[4 * g + 2].x += a.z * b.z; acc[4 * g + 3].x += a.w * b.w;
            } else {
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[4 * g + 3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(spread)::value && DMA_P) {
                if (g < 5) {
                    if (more) load_a_piece(c + 1, buf ^ 1, g);
                } else if (g < 5 + WF4_P_PASSES) {
                    if (more2) load_p_piece(c + 2, buf, g - 5);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: chunk 0 in LDS and transformed, chunk 1's patch in LDS ----
    WF4_MARK(0);
    load_p(0, 0);
#if WF4_GLOBAL_A
    {
        auto first = [&](auto sl) { ga_load(sl, 0, decltype(sl)::value); };      // groups 0 .. AHEAD - 1 of chunk 0
        first(std::integral_constant<int, 0>{});
        if (GA_D > 1) first(std::integral_constant<int, 1>{});
        if (GA_D > 2) first(std::integral_constant<int, 2>{});
        if (GA_D > 3) first(std::integral_constant<int, 3>{});
        if (GA_D > 4) first(std::integral_constant<int, 4>{});
        if (GA_D > 5) first(std::integral_constant<int, 5 % GA_SLOTS>{});
        if (GA_D > 6) first(std::integral_constant<int, 6 % GA_SLOTS>{});
        if (GA_D > 7) first(std::integral_constant<int, 7 % GA_SLOTS>{});
    }
#else
    load_a(0, 0);
#endif
    store_p(0);
    store_a(0);
    __syncthreads();
    WF4_MARK(1);
    if (p.nchunks > 1) load_p(1, 1);
    transform_first(0, 0);
    transform_last(0, 0);
    if (p.nchunks > 1) store_p(1);
    __syncthreads();
    WF4_MARK(2);
    // (cache policy of the LDS-DMA requests, measured: non-temporal patch / filter / both 41.6 / 41.3 / 42.5 us against 39.0 us for
    //  the default policy -- neighbouring workgroups share halo pixels and every workgroup the filter -- sc0 39.2 us)
    // one K step; the buffer parity is a compile-time constant, so the compiler can tell the LDS-DMA destinations
    // (A[nxt], P[cur]) from what the step reads and writes (A[cur], V[cur], P[nxt], V[nxt]) and lets the DMA fly
    auto kstep = [&](auto parity, int c) {
        constexpr int cur = decltype(parity)::value, nxt = cur ^ 1;
        const bool more = c + 1 < p.nchunks, more2 = c + 2 < p.nchunks;
        if (STAGGER) {
            // A[nxt] was last read by the MFMAs of chunk c - 1, P[cur] by the transform of chunk c: free for the whole step.
            // Waves 4-7 transform BEFORE they request their share of the next operands (the patch they read landed a step
            // ago; an LDS-DMA instruction holds a wave's issue slot ~100 cycles, seven of them would delay the transform
            // their MFMAs wait for): 38.2 -> 37.6 us
            WF4_STEP_MARK(c, 0);
            if (more) transform_first(nxt, nxt);
            WF4_STEP_MARK(c, 1);
#if WF4_GLOBAL_A
            mma_ga(parity, c, more2);
#else
            if constexpr (SPREAD) {
                mma(cur, c, more, more2, std::true_type{});
            } else {
                if (more) load_a(c + 1, nxt);
                if (more2) load_p(c + 2, cur);
                mma(cur, c, more, more2, std::false_type{});
            }
#endif
            WF4_STEP_MARK(c, 2);
            if (more) transform_last(nxt, nxt);
            WF4_STEP_MARK(c, 3);
        } else {
#if WF4_GLOBAL_A
            mma_ga(parity, c, more2);
#else
            if (more) load_a(c + 1, nxt);
            if (more2) load_p(c + 2, cur);
            mma(cur, c, more, more2, std::false_type{});
#endif
            if (more) {
                transform_first(nxt, nxt);
                transform_last(nxt, nxt);
            }
        }
        if (more) store_a(nxt);
        if (more2) store_p(cur);
        __syncthreads();
        WF4_STEP_MARK(c, 4);
    };
    // static priority for the waves that multiply first (0-3; their SIMD partners 4-7 open every step with the patch
    // transform): 39.2 -> 38.5-38.8 us per layer1 conv; the other half at priority 1 instead: 41.3 us
    if (!HALF && wave < 4) __builtin_amdgcn_s_setprio(1);
    for (int c = 0; c < p.nchunks; c += 2) {
        kstep(std::integral_constant<int, 0>{}, c);
        if (c + 1 < p.nchunks) kstep(std::integral_constant<int, 1>{}, c + 1);
    }

    // ---- lane-local output transform; fused tail and stores through the wave's exchange buffer ----
    WF4_MARK(3);
    const int coq = (int)coutblk * 16 + wm * 4 + lk;
    const int cqc = min(coq, p.Coq - 1);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? p.y_bytes : 0u, 0x00020000);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const bool plain = !p.ep.bias && p.ep.scale && p.ep.shift && p.ep.act == 1 && !(p.ep.res && p.ep.res_post);
    const float4 scale = plain ? reinterpret_cast<const float4 *>(p.ep.scale)[cqc] : one4;      // the straight-line tail's
    const float4 shift = plain ? reinterpret_cast<const float4 *>(p.ep.shift)[cqc] : z4;        // parameters, held in registers
    // the tail is the same for the whole launch: scalar branches pick the straight-line form where it applies and whether a
    // residual is fetched
#if WF4_GLOBAL_A
    // the K loop is over: V (and, for half blocks, P) is free -- 4.25 KB of exchange buffer per wave
    float4 *xb = HALF ? reinterpret_cast<float4 *>(wave == 0 ? Vs0 : wave == 1 ? Vs1 : wave == 2 ? Ps0 : Ps1)
                      : reinterpret_cast<float4 *>(wave < 4 ? Vs0 : Vs1) + (wave & 3) * (4 * 68);
#else
    float4 *xb = reinterpret_cast<float4 *>(As0) + wave * (4 * 68);      // the K loop is over: A0 is free (8 x 4.25 KB)
#endif
    const int te = lane >> 2, be = lane & 3;
    const int oj2 = wn * 16 + te;
    int n2 = n0 + (oj2 >> lT), ty2 = ty0 + ((oj2 >> LBC) & BRm), tx2 = tx0 + (oj2 & BCm);
    if constexpr (PACK) {
        if ((oj2 & BCm) >= p.tw) {                               // a spare slot: the donor image's tile
            n2 = dn;
            ty2 = (dgr << p.lBR) + ((oj2 >> LBC) & BRm);
            tx2 = p.pack_sc * dgc + (oj2 & BCm) - p.tw;
        }
    }
    const int x2 = tx2 * 4 + be, cq0 = (int)coutblk * 16 + wm * 4;
    const bool ok2 = n2 < p.N && ty2 < p.th && tx2 < p.tw && x2 < p.W;
    auto row = [&](auto first, auto res, auto pl) {
        constexpr int A = decltype(first)::value;
        const int yy = ty2 * 4 + A;
        int off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            off[q] = (ok2 && yy < p.H && cq0 + q < p.Coq)
                         ? (int)((((unsigned)(n2 * p.Coq + cq0 + q) * (unsigned)p.H + (unsigned)yy) * (unsigned)p.W + (unsigned)x2) << 4) : OOB;
        wf4_output_row_coalesced<A, decltype(res)::value, decltype(pl)::value>(p, acc, cqc, scale, shift, xb, lk * 16 + li,
                                                                              be * 68 + te, yrsrc, rrsrc, off);
    };
    auto rows = [&](auto res, auto pl) {
        row(std::integral_constant<int, 0>{}, res, pl);
        row(std::integral_constant<int, 1>{}, res, pl);
        row(std::integral_constant<int, 2>{}, res, pl);
        row(std::integral_constant<int, 3>{}, res, pl);
    };
    if (plain) {
        if (p.ep.res) rows(std::true_type{}, std::true_type{});
        else rows(std::false_type{}, std::true_type{});
    } else {
        if (p.ep.res) rows(std::true_type{}, std::false_type{});
        else rows(std::false_type{}, std::false_type{});
    }
    WF4_MARK(4);
#ifdef WF4_STAMP
    __builtin_amdgcn_s_waitcnt(0);             // (probe: when have this wave's stores left?)
    WF4_MARK(5);
#endif
}

template <bool DMA_A, bool PLANAR, bool STAGGER, int LBC, bool PACK = false, bool HALF = false>
__global__ void __launch_bounds__(HALF ? 256 : 512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_wf4_kernel(const Wf4Args p) {
    conv_wf4_body<DMA_A, PLANAR, STAGGER, LBC, PACK, HALF>(p);
}

// filter: OIHW 3x3 -> u[cout block][chunk][cb][kk][i][f] = (G g G^T)[f] of channel (64 blk + 16 cb + i, 4 chunk + kk); zero beyond Cout
__global__ void __launch_bounds__(256) wf4_filter_kernel(const float *w, float *u, unsigned total, int Cin, int Cout) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;   // (co, c) pair
    if (i >= total) return;
    const int co = (int)(i / (unsigned)Cin), c = (int)(i - (unsigned)co * Cin);
    const float *g = w + (size_t)i * 9;
    float t[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0 * 0.25f;
        t[1][j] = -(g0 + g1 + g2) * (1.f / 6.f);
        t[2][j] = (-g0 + g1 - g2) * (1.f / 6.f);
        t[3][j] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[4][j] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        t[5][j] = g2;
    }
    const int nchunks = Cin / 4;
    float f[36];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const float g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
        f[a * 6 + 0] = g0 * 0.25f;
        f[a * 6 + 1] = -(g0 + g1 + g2) * (1.f / 6.f);
        f[a * 6 + 2] = (-g0 + g1 - g2) * (1.f / 6.f);
        f[a * 6 + 3] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        f[a * 6 + 4] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        f[a * 6 + 5] = g2;
    }
    float *blk = u + ((size_t)(co >> 6) * nchunks + (c >> 2)) * WF4_A_FLOATS;
#if WF4_GLOBAL_A
    // [wm = 16-channel block][g = 4 frequencies][lane = k * 16 + channel][4]: what one wave loads for one MFMA group is contiguous
    float *up = blk + (((co >> 4) & 3) * 9 * 64 + (c & 3) * 16 + (co & 15)) * 4;
#pragma unroll
    for (int q = 0; q < 36; ++q) up[(q >> 2) * 256 + (q & 3)] = f[q];
#else
    float *up = blk + (((((co >> 4) & 3) * 4 + (c & 3)) * 16) + (co & 15)) * 36;
#pragma unroll
    for (int q = 0; q < 36; ++q) up[q] = f[q];
#endif
}
