// 3x3 / stride 1 convolution on 1..4 input channels, NCHW in and out -- included by conv_direct.hip inside
// its anonymous namespace.  BASELINE config 2 (Conv2d 3->64 on (8,3,224,224), reference util.conv_for,
// util.py:17-44) is this shape: K = 27, 1.39 GFLOP against 107.6 MB of traffic of which 102.8 MB are
// the output -- HBM-WRITE-bound (floor 13.4 us at 8 TB/s, ~17 us at the achievable 6.3 TB/s), the
// matrix cores need 8.8 us.  So the kernel is built around the store stream:
//   * a workgroup owns tpw x 256 consecutive output pixels of one image x 64 output channels; the filter
//     (k-major [K][64]) and the few input rows those pixels touch (zero border included) are staged in
//     LDS once -- x is read from HBM once (plus a two-row halo per workgroup), nothing is re-read;
//   * M = output channels, N = pixels on v_mfma_f32_32x32x2_f32: a lane's accumulator rows are output
//     channels and its column is ONE pixel, consecutive lanes = consecutive pixels, so every store
//     instruction writes two full 128-byte lines of two NCHW channel planes (no partial lines, no
//     transposition through LDS);
//   * K order (cin, kh, kw) = K.reshape(Cout, -1) exactly as the reference's sgemm sees it; the im2col
//     value of (pixel, k) is one ds_read_b32 at pixel base + a per-k offset kept in registers.
struct SmallCinArgs {
    const float *x, *w, *bias;
    float *y;
    int N, Cin, H, W, Cout, Ho, Wo, pad;
    int x_bytes, y_bytes;
    int HoWo, Wp, K, steps;          // Wp = W + 2: every staged row carries one zero column at each end
    int tpw;                         // 256-pixel tiles per workgroup (filter and input rows are staged once for all of them)
    int xb_off;                      // WIDE: float offset of the waves' exchange buffers (4 x 256 floats, 16-byte aligned) in LDS
    FastDiv divWo, divK;
};

constexpr int SC_PIX = 256, SC_CO = 64, SC_MAXK = 36;

// (threads 0..255 of the workgroup call this)
__device__ __forceinline__ void sc_stage(const SmallCinArgs &p, float *As, float *Bs, float *Ps, int tid, int n, int co0, int r0,
                                         int rows) {
    // ---- stage the filter (As[k][co] = w[co0+co][k]), the bias and the input rows (a thread = one column,
    //      coalesced along W; zero outside the image).  Every global load of the first batch -- 9 filter
    //      values, the bias, up to 24 (channel, row) lines -- is issued before the first LDS write, so a
    //      workgroup pays ONE memory round trip before its MFMAs start ----
    {
        constexpr int LB = 24;
        constexpr int OOB = (int)0x80000000;
        // zero fill by the buffers' range check: a predicated plain load makes hipcc wait for each element in turn
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.Cout * p.K * 4, 0x00020000);
        const int lines = p.Cin * rows;
        float fv[SC_MAXK * SC_CO / 256], xv[LB];
#pragma unroll
        for (int j = 0; j < SC_MAXK * SC_CO / 256; ++j) {          // flat, coalesced read of the 64 filters' K values
            const int i = tid + j * 256;
            fv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, i < SC_CO * p.K ? (co0 * p.K + i) << 2 : OOB, 0, 0));
        }
        const float bv = (tid < SC_CO && p.bias && co0 + tid < p.Cout) ? p.bias[co0 + tid] : 0.f;
        for (int l0 = 0; l0 < lines; l0 += LB)
            for (int cc = tid; cc < p.Wp || (l0 == 0 && cc == tid); cc += 256) {
                const int w = cc - p.pad;
                int c = l0 / rows, rr = l0 - c * rows;              // uniform
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    const int h = r0 - p.pad + rr;
                    const bool ok = l0 + j < lines && cc < p.Wp && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                    xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                          xrsrc, ok ? (((n * p.Cin + c) * p.H + h) * p.W + w) << 2 : OOB, 0, 0));
                    if (++rr == rows) { rr = 0; ++c; }
                }
                if (l0 == 0 && cc == tid) {                         // first batch: the filter rides along
#pragma unroll
                    for (int j = 0; j < SC_MAXK * SC_CO / 256; ++j) {       // element i = (co, k) -> As[k][co]
                        const int i = tid + j * 256;
                        unsigned co, k;
                        p.divK.divmod((unsigned)i, co, k);
                        if (i < SC_CO * p.K) As[k * SC_CO + co] = fv[j];
                        else if (i < SC_CO * 2 * p.steps) As[i] = 0.f;          // K padding rows
                    }
                    if (tid < SC_CO) Bs[tid] = bv;
                    if (tid < 4) Bs[SC_CO + tid] = 0.f;
                }
                if (cc < p.Wp) {
#pragma unroll
                    for (int j = 0; j < LB; ++j)
                        if (l0 + j < lines) Ps[(l0 + j) * p.Wp + cc] = xv[j];
                }
            }
    }
}

// WIDE: the output leaves through 16-byte stores.  In the MFMA's C layout a lane holds ONE pixel of 16 channels, so the plain
// form stores 4 bytes per lane (a wave instruction = two 128-byte lines); here four accumulator registers at a time (8 channels
// x 32 pixels) cross a wave-private 1 KB LDS buffer -- no workgroup barrier -- and come back as lane = (channel, pixel quad): a
// wave instruction then writes eight full 128-byte lines, a quarter of the store instructions.  Needs Ho*Wo % 4 == 0.
template <bool WIDE>
__global__ void __launch_bounds__(256) conv_smallcin_nchw_kernel(const SmallCinArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                         // [steps*2][64]   filter, k-major, zero padded
    float *Bs = smem + SC_MAXK * SC_CO;       // [64] bias (zeros without one)
    float *Ps = Bs + SC_CO + 4;               // [Cin][rows][Wp] input rows with zero borders; Ps[-1] = 0 (K padding reads it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n = blockIdx.z, co0 = blockIdx.y * SC_CO, pw0 = blockIdx.x * SC_PIX * p.tpw;
    const int p0 = pw0;
    const int plast = min(pw0 + SC_PIX * p.tpw, p.HoWo) - 1;
    const int r0 = (int)p.divWo.div((unsigned)p0), r1 = (int)p.divWo.div((unsigned)plast);
    const int rows = r1 - r0 + 3;             // input rows r0-pad .. r1-pad+2
    sc_stage(p, As, Bs, Ps, tid, n, co0, r0, rows);
    // per-k offsets into the staged rows for this lane's k = 2s + lhi
    int koff[SC_MAXK / 2];
#pragma unroll
    for (int s = 0; s < SC_MAXK / 2; ++s) {
        const int k = 2 * s + lhi;
        const int c = k / 9, t = k - 9 * c, dy = t / 3, dx = t - 3 * dy;
        koff[s] = k < p.K ? (c * rows + dy) * p.Wp + dx : -1;
    }
    __syncthreads();
    // ---- tiles, software-pipelined: the stores of tile t-1 (64 per lane) are issued in 16 groups of four
    //      between the MFMA steps of tile t, out of the other accumulator set -- the write stream (3.3 us per tile
    //      per CU at 5 TB/s) and the matrix pipe (1.5 us) overlap inside every wave instead of adding up
    //      chip-wide when all workgroups reach the same phase together ----
    // the lane's 32 bias values live in registers: an LDS read per store group would put an lgkmcnt wait -- which
    // is in-order, so it also waits for the operand reads of the MFMA steps ahead -- in front of every group
    float bias_r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) bias_r[i] = Bs[(i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2) + 4 * lhi];
    struct TileAt {
        int pbase[2];
        int yoff[2];                 // byte offset of (channel co0, this lane's pixel) in y, out of range when masked
        int yq[2];                   // WIDE: byte offset of (channel co0 + lane / 8, pixel quad lane % 8 of pixel block b)
        bool live;
    };
    constexpr int YOOB = (int)0x80000000;
    // y goes through a buffer descriptor: a store of a masked lane gets an out-of-range offset and is dropped --
    // no branch per store (a predicated plain store costs a branch each and serialises the tail)
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    auto locate = [&](int t) {
        TileAt ta;
        const int pt0 = pw0 + t * SC_PIX;
        ta.live = t < p.tpw && pt0 < p.HoWo;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int pix = pt0 + wave * 64 + b * 32 + l31;
            const bool ok = ta.live && pix < p.HoWo;
            unsigned ho, wo;
            p.divWo.divmod((unsigned)(ok ? pix : p0), ho, wo);
            ta.pbase[b] = ((int)ho - r0) * p.Wp + (int)wo;   // staged (row ho-r0, column wo) = image (ho-pad, wo-pad)
            ta.yoff[b] = ok ? (int)((((unsigned)(n * p.Cout + co0)) * (unsigned)p.HoWo + (unsigned)pix) << 2) : YOOB;
            const int pq = pt0 + wave * 64 + b * 32 + 4 * (lane & 7);
            ta.yq[b] = (WIDE && ta.live && pq < p.HoWo)
                           ? (int)((((unsigned)(n * p.Cout + co0 + (lane >> 3))) * (unsigned)p.HoWo + (unsigned)pq) << 2) : YOOB;
        }
        return ta;
    };
    auto zero = [&](f32x16 (&acc)[2][2]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    auto mma_step = [&](f32x16 (&acc)[2][2], const TileAt &ta, int s2) {
        const float a0 = As[(2 * s2 + lhi) * SC_CO + l31], a1 = As[(2 * s2 + lhi) * SC_CO + 32 + l31];
        const float b0 = Ps[koff[s2] >= 0 ? ta.pbase[0] + koff[s2] : -1];      // K padding reads the zero slot
        const float b1 = Ps[koff[s2] >= 0 ? ta.pbase[1] + koff[s2] : -1];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    };
    // bias + store, group g of 16: accumulator rows 2g, 2g+1 of both 32-channel blocks... lanes = consecutive pixels of
    // one channel plane -> full 128-byte lines
    auto store_group = [&](const f32x16 (&acc)[2][2], const TileAt &ta, int g) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = 2 * g + q, a = idx >> 4, r = idx & 15;
            const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const bool cok = co0 + co < p.Cout;
            const float bv = bias_r[idx];
            const int plane = (co * p.HoWo) << 2;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float v = acc[a][b][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p.bias ? __fadd_rn(v, bv) : v), yrsrc,
                                                      (cok && ta.yoff[b] != YOOB) ? ta.yoff[b] + plane : YOOB, 0, 0);
            }
        }
    };
    // WIDE: group g = (channel block a, register quad j, pixel block b): registers 4j..4j+3 of acc[a][b] = channels
    // 32a + 8j + (0..3) + 4 lhi x pixels 32b + l31
    float *xb = smem + p.xb_off + wave * 256;
    float bias_w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias_w[i] = WIDE ? Bs[(i >> 2) * 32 + 8 * (i & 3) + (lane >> 3)] : 0.f;
    auto exchange_wide = [&](const f32x16 (&acc)[2][2], int g) {
        const int a = g >> 3, j = (g >> 1) & 3, b = g & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) xb[(q + 4 * lhi) * 32 + l31] = acc[a][b][4 * j + q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return *reinterpret_cast<const float4 *>(xb + lane * 4);
    };
    auto store_wide = [&](float4 v, const TileAt &ta, int g) {
        const int a = g >> 3, j = (g >> 1) & 3, b = g & 1;
        const int co = a * 32 + 8 * j + (lane >> 3);
        if (p.bias) {
            const float bv = bias_w[a * 4 + j];
            v = make_float4(__fadd_rn(v.x, bv), __fadd_rn(v.y, bv), __fadd_rn(v.z, bv), __fadd_rn(v.w, bv));
        }
        const bool cok = co0 + co < p.Cout;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), yrsrc,
                                               (cok && ta.yq[b] != YOOB) ? ta.yq[b] + (((a * 32 + 8 * j) * p.HoWo) << 2) : YOOB, 0, 0);
    };
    f32x16 acc0[2][2], acc1[2][2];
    TileAt cur = locate(0), prev = cur;
    prev.live = false;
    auto run_tile = [&](f32x16 (&acc)[2][2], const f32x16 (&old)[2][2]) {       // tile `cur` into acc, tile `prev` out of old
        zero(acc);
        float4 pend = make_float4(0.f, 0.f, 0.f, 0.f);        // WIDE: group s2 is exchanged now and stored one step later
#pragma unroll
        for (int s2 = 0; s2 < SC_MAXK / 2; ++s2) {
            if (s2 < p.steps && cur.live) mma_step(acc, cur, s2);
            if constexpr (WIDE) {
                if (s2 >= 1 && s2 <= 16 && prev.live) store_wide(pend, prev, s2 - 1);
                if (s2 < 16 && prev.live) pend = exchange_wide(old, s2);
            } else {
                if (s2 < 16 && prev.live) store_group(old, prev, s2);
            }
        }
    };
    for (int t = 0; t <= p.tpw; t += 2) {
        run_tile(acc0, acc1);
        prev = cur; cur = locate(t + 1);
        run_tile(acc1, acc0);
        prev = cur; cur = locate(t + 2);
        if (!prev.live && !cur.live) break;
    }
}
