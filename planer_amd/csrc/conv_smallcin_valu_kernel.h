// 3x3 / stride 1 convolution on 1..4 input channels, NCHW in and out, on the VECTOR ALUs -- included by conv_direct.hip
// inside its anonymous namespace.  BASELINE config 2 (Conv2d 3->64 on (8,3,224,224); reference layer.Conv2d
// layer.py:22-26 -> util.conv_for util.py:17-44) has K = Cin*kh*kw = 27: 1.39 GFLOP against 102.8 MB of output, i.e.
// HBM-WRITE-bound (13.4 us at 8 TB/s) with 8.8 us of arithmetic at the fp32 peak -- and the fp32 VALU peak of gfx950 IS
// the fp32 MFMA peak (64 FMA lanes per SIMD either way).  The MFMA kernel of conv_smallcin_kernel.h stalls on its own
// store stream (MFMAs and a dense store stream issued by the same wave do not overlap); here there is
// no matrix instruction, no LDS exchange and no barrier in the loop:
//   * a lane owns FOUR consecutive output pixels of one row (one 16-byte piece of an NCHW plane) and walks the output
//     channels eight at a time: 32 accumulators, 864 v_fma_f32 per pass, then eight buffer_store_dwordx4 -- consecutive lanes
//     are consecutive pixel quads, so every store instruction writes one contiguous 1 KB run of a channel plane
//     (eight full 128-byte lines, no transposition);
//   * the lane's input window (Cin x 3 rows x 6 columns, zero padding by the buffer range check) is loaded ONCE into
//     registers and reused for every output channel the workgroup covers: x is read once per channel block;
//   * the filter of the workgroup's channel block sits in LDS k-major ([k][channel], 27 x 64 floats) and is read as
//     wave-uniform ds_read_b128 broadcasts (two per tap per pass): no per-lane filter traffic;
//   * nothing but occupancy overlaps arithmetic and stores: 111 registers = 4 waves per SIMD, each wave alternates
//     432 v_pk_fma_f32 with 8 KB of stores, so the store queues of a CU never drain while some wave is multiplying.
// Measured on config 2 (MI355X): 28.5 us per launch (3.77 TB/s of algorithmic bytes = 0.47 of 8 TB/s) against 37.4 us for the
// MFMA store-stream kernel; 256-thread workgroups x 32 channels is the best grid (128 / 64 threads: 29.8 / 32.6 us; 64 / 16
// channels: 29.2 / 29.3 us); non-temporal stores change nothing (29.1 us).
// K order (cin, kh, kw) = K.reshape(Cout, -1) as the reference's sgemm sees it, one fmaf chain per output, bias added last
// (layer.py:26).
struct SmallCinValuArgs {
    const float *x, *w, *bias;
    float *y;
    int N, H, W, Cout, Ho, Wo;
    int quads;                     // Ho * Wo / 4 pixel quads per plane
    int cpb;                       // output channels per workgroup (multiple of 8, <= 64)
    unsigned x_bytes, y_bytes;
    FastDiv divQw;                 // by Wo / 4
};

constexpr int SCV_MAXC = 64;
typedef float scv_v2 __attribute__((ext_vector_type(2)));

// acc += splat(x.lo or x.hi) * w on a channel pair: ONE v_pk_fma_f32, the input value picked by op_sel (hipcc builds the splat
// with two v_mov per value and keeps 2 x 54 registers of them alive otherwise)
__device__ __forceinline__ void scv_fma_lo(scv_v2 &acc, scv_v2 x, scv_v2 w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(w));
}
__device__ __forceinline__ void scv_fma_hi(scv_v2 &acc, scv_v2 x, scv_v2 w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(w));
}

// KNOCK (measurement builds only, PLANER_HIP_EXPERIMENT=scv_knock=<mask>, the 3-channel / pad-1 instantiation): bit 0 no input
// loads, bit 1 no FMAs, bit 2 no stores, bit 3 no per-tap filter reads.  Config 2 on MI355X (tools/scv_probe.py): everything
// 27.2 us; stores only 18.5-19.2 us (5.6-5.8 TB/s: the store pattern is not what bounds the kernel); no stores 23.1; FMAs alone
// (no loads, no stores, no filter reads) 17-19 us where 432 v_pk_fma_f32 x 16 passes on the busiest SIMD need 13 us.
template <int CIN, int PAD, int KNOCK = 0>
__global__ void __launch_bounds__(256) conv_smallcin_valu_kernel(const SmallCinValuArgs p) {
    constexpr int K = CIN * 9;
    __shared__ __attribute__((aligned(16))) float Ws[K * SCV_MAXC + SCV_MAXC];        // [k][channel], then the bias
    const int tid = threadIdx.x;
    const int n = blockIdx.z, co0 = blockIdx.y * p.cpb;
    // ---- filter block -> LDS, transposed to k-major; channels beyond Cout are zero.  Range-checked buffer loads, all of a
    //      thread's requests in flight before the first LDS write (a predicated plain load `c ? w[i] : 0` is a branch and a wait per
    //      element: seven dependent round trips at the head of a 25 us kernel) ----
    //      (the LDS writes sit behind the requests for the input window below, so both round trips overlap)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (unsigned)p.Cout * K * 4u, 0x00020000);
    constexpr int WPER = (K * SCV_MAXC + 255) / 256;
    float wreg[WPER];
#pragma unroll
    for (int t = 0; t < WPER; ++t) {
        const int i = tid + t * 256, co = i / K;         // flat read of the block's filters (coalesced), scattered LDS write
        wreg[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                wrsrc, (i < K * SCV_MAXC && co < p.cpb) ? (co0 * K + i) << 2 : (int)0x80000000, 0, 0));
    }
    const float bvl = (tid < SCV_MAXC && p.bias && tid < p.cpb && co0 + tid < p.Cout) ? p.bias[co0 + tid] : 0.f;

    // ---- this lane's pixel quad and its input window ----
    const unsigned j = blockIdx.x * 256u + tid;
    const bool live = j < (unsigned)p.quads;
    unsigned yy, q;
    p.divQw.divmod(live ? j : 0u, yy, q);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    scv_v2 xv[CIN][3][3];                              // six columns as three register pairs
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = (int)yy + r - PAD;
            const bool rok = live && (unsigned)h < (unsigned)p.H;
            const int base = ((n * CIN + c) * p.H + h) * p.W + 4 * (int)q;          // column 4q of that row
            // columns 4q - PAD .. 4q - PAD + 5: one 16-byte load (always inside the row) and two single values
            if constexpr (KNOCK & 1) {
                xv[c][r][0] = (scv_v2){(float)tid, 1.f}; xv[c][r][1] = (scv_v2){2.f, (float)c}; xv[c][r][2] = (scv_v2){(float)r, 3.f};
                continue;
            }
            const float4 m = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, rok ? base << 2 : OOB, 0, 0));
            if constexpr (PAD == 1) {
                const float lo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (rok && q > 0) ? (base - 1) << 2 : OOB, 0, 0));
                const float hi = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (rok && 4 * (int)q + 4 < p.W) ? (base + 4) << 2 : OOB, 0, 0));
                xv[c][r][0] = (scv_v2){lo, m.x}; xv[c][r][1] = (scv_v2){m.y, m.z}; xv[c][r][2] = (scv_v2){m.w, hi};
            } else {
                const float h0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, rok ? (base + 4) << 2 : OOB, 0, 0));
                const float h1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, rok ? (base + 5) << 2 : OOB, 0, 0));
                xv[c][r][0] = (scv_v2){m.x, m.y}; xv[c][r][1] = (scv_v2){m.z, m.w}; xv[c][r][2] = (scv_v2){h0, h1};
            }
        }
#pragma unroll
    for (int t = 0; t < WPER; ++t) {
        const int i = tid + t * 256, co = i / K;
        if (i < K * SCV_MAXC) Ws[(i - co * K) * SCV_MAXC + co] = wreg[t];
    }
    if (tid < SCV_MAXC) Ws[K * SCV_MAXC + tid] = bvl;
    __syncthreads();

    // ---- passes of eight output channels ----
    const int plane = p.quads * 16;                                     // bytes of one output plane
    int yoff = live ? (int)(((unsigned)(n * p.Cout + co0) * (unsigned)p.quads + j) << 4) : OOB;
    for (int g = 0; g < p.cpb; g += 8) {
        // accumulators as channel PAIRS (a filter pair is a register pair straight out of the ds_read_b128; the pixel's input
        // value feeds both halves of a v_pk_fma_f32 through op_sel): no operand shuffling
        scv_v2 acc[4][4];                                // [pixel][channel pair]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) acc[i][cp] = (scv_v2){0.f, 0.f};
        // the filter values of tap k + 1 are requested before the FMAs of tap k (two ds_read_b128 broadcasts per tap)
        float4 w0 = *reinterpret_cast<const float4 *>(Ws + g), w1 = *reinterpret_cast<const float4 *>(Ws + g + 4);
#pragma unroll
        for (int k = 0; k < ((KNOCK & 2) ? 1 : K); ++k) {
            const int c = k / 9, r = (k % 9) / 3, dx = k % 3;
            const scv_v2 wv[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
            if (k + 1 < K && !(KNOCK & 8)) {
                w0 = *reinterpret_cast<const float4 *>(Ws + (k + 1) * SCV_MAXC + g);
                w1 = *reinterpret_cast<const float4 *>(Ws + (k + 1) * SCV_MAXC + g + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) {
                    if ((i + dx) & 1) scv_fma_hi(acc[i][cp], xv[c][r][(i + dx) >> 1], wv[cp]);
                    else scv_fma_lo(acc[i][cp], xv[c][r][(i + dx) >> 1], wv[cp]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 b0 = *reinterpret_cast<const float4 *>(Ws + K * SCV_MAXC + g);
        const float4 b1 = *reinterpret_cast<const float4 *>(Ws + K * SCV_MAXC + g + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const float4 o = make_float4(acc[0][cc >> 1][cc & 1] + bv[cc], acc[1][cc >> 1][cc & 1] + bv[cc],
                                         acc[2][cc >> 1][cc & 1] + bv[cc], acc[3][cc >> 1][cc & 1] + bv[cc]);
            // the channel plane rides in the VECTOR offset (scalar offset 0).  With it in the scalar offset hipcc places no wait
            // state between a buffer_store_dwordx4 and a VALU write of its data registers (its hazard model exempts stores
            // whose soffset is a register) -- measured on gfx950: lanes 12-15 of every 16 then stored the NEXT value of the
            // second data register (Cin = 1 instantiation, where the allocator reused the registers at once)
            const bool st = (KNOCK & 4) ? (o.x == 12345.678f && live) : (live && co0 + g + cc < p.Cout);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o), yrsrc,
                                                   st ? yoff + cc * plane : OOB, 0, 0);
        }
        if (live) yoff += 8 * plane;
    }
}

