// The 36 grouped GEMMs of a staged F(4x4,3x3) convolution with 128 input channels, filter-stationary -- included by
// conv_winograd.hip inside its anonymous namespace.
//
// ResNet-18's layer2 (128 -> 128 channels on 28x28) is M = 128, K = 128, N = 1568 per frequency: the tiled kernel's best
// plan there is 128 x 32 tiles, which stage (128 + 32) x 32 x 4 bytes per 128 x 32 x 32 x 2 FLOP = 12.8 FLOP per byte --
// under the ~16 a CU can pull from L2 -- and re-fetch the frequency's 64 KB filter for every tile
// (0.46 of the MFMA peak at batch 32, 0.62 saturated).  Here a workgroup keeps the filter block of ONE frequency
// (128 rows x 128 k = 64 KB) in LDS for its whole life and walks `nt` column sub-tiles of 32: only V streams
// (16 KB per 128 x 32 x 128 x 2 FLOP = 64 FLOP per byte), double-buffered by LDS-DMA (separate LDS objects, so the
// transfer of sub-tile t + 1 is not waited for by the reads of sub-tile t).  One wave = 32 rows x 32 columns, two
// accumulators (even / odd k-quad pairs) so consecutive MFMAs do not wait for each other; M leaves the accumulators as
// 16-byte stores of 32 consecutive columns (512-byte runs), fire and forget under the next sub-tile's MFMAs.
//   Uq [36][32 q][Cout][4]   V [36][32 q][T][4]   M [36][Cout/4][T][4]
struct Wino4GemmAsArgs {
    const float *U, *V;
    float *M;
    int Cout, T;              // Cout % 128 == 0; T = N * tiles
    int nt;                   // 32-column sub-tiles per workgroup
    int wpf;                  // workgroups per (frequency, 128-row block)
    unsigned u_bytes, v_bytes, m_bytes;
};

__global__ void __launch_bounds__(256) wino4_gemm_as_kernel(const Wino4GemmAsArgs p) {
    __shared__ __attribute__((aligned(16))) float As[32 * 128 * 4];                 // [q][row][4]
    __shared__ __attribute__((aligned(16))) float Bs0[32 * 32 * 4], Bs1[32 * 32 * 4];   // [q][column][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int mblocks = p.Cout >> 7;
    const int chunk = (int)blockIdx.x % p.wpf, fm = (int)blockIdx.x / p.wpf;
    const int f = fm / mblocks, m0 = (fm - f * mblocks) << 7;
    const int st0 = chunk * p.nt, nsub = (p.T + 31) >> 5;
    const int st1 = min(st0 + p.nt, nsub);
    typedef __attribute__((address_space(3))) float lds_float;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.U), 0, p.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.V), 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(p.M, 0, p.m_bytes, 0x00020000);

    // the filter block: 4096 cells of 16 bytes = 16 passes of 256 threads; cell v = (q = v / 128, row = v % 128)
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int v = it * 256 + tid, q = v >> 7, row = v & 127;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_float *)(As + (v - lane) * 4), 16,
                                                 (int)((((unsigned)(f * 32 + q) * (unsigned)p.Cout + (unsigned)(m0 + row))) << 4), 0, 0, 0);
    }
    // a V sub-tile: 1024 cells = 4 passes; cell v = (q = v / 32, column = v % 32)
    auto load_b = [&](int st, float *Bb) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int v = it * 256 + tid, q = v >> 5, col = (st << 5) + (v & 31);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_float *)(Bb + (v - lane) * 4), 16,
                                                     col < p.T ? (int)(((unsigned)(f * 32 + q) * (unsigned)p.T + (unsigned)col) << 4) : OOB, 0, 0, 0);
        }
    };
    if (st0 < st1) load_b(st0, Bs0);
    __syncthreads();

    const float *Ab = As + (lhi * 128 + wave * 32 + l31) * 4;       // fragment of k-quad pair u: quad 2u + lhi, this lane's row
    const int b_off = (lhi * 32 + l31) * 4;
    const unsigned mrow = ((unsigned)(f * (p.Cout >> 2)) + (unsigned)((m0 >> 2) + 8 * wave + lhi)) * (unsigned)p.T;
    auto sub_tile = [&](auto parity, int st) {
        constexpr int cur = decltype(parity)::value;
        const float *Bb = (cur ? Bs1 : Bs0) + b_off;
        if (st + 1 < st1) load_b(st + 1, cur ? Bs0 : Bs1);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
        // 16 k-quad pairs; fragments two pairs ahead
        float4 fa[3], fb[3];
        fa[0] = *reinterpret_cast<const float4 *>(Ab);
        fb[0] = *reinterpret_cast<const float4 *>(Bb);
        fa[1] = *reinterpret_cast<const float4 *>(Ab + 2 * 128 * 4);
        fb[1] = *reinterpret_cast<const float4 *>(Bb + 2 * 32 * 4);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (u + 2 < 16) {
                fa[(u + 2) % 3] = *reinterpret_cast<const float4 *>(Ab + 2 * (u + 2) * 128 * 4);
                fb[(u + 2) % 3] = *reinterpret_cast<const float4 *>(Bb + 2 * (u + 2) * 32 * 4);
            }
            const float4 a = fa[u % 3], b = fb[u % 3];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc1, 0, 0, 0);
        }
        // everyone is done reading this sub-tile and the next one's V has landed (the barrier's vmcnt wait); the stores
        // below are only waited for at the NEXT barrier, a whole sub-tile of MFMAs later
        __syncthreads();
        const int col = (st << 5) + l31;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 v = make_float4(acc0[4 * rq] + acc1[4 * rq], acc0[4 * rq + 1] + acc1[4 * rq + 1],
                                         acc0[4 * rq + 2] + acc1[4 * rq + 2], acc0[4 * rq + 3] + acc1[4 * rq + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), mrsrc,
                                                   col < p.T ? (int)((mrow + (unsigned)(2 * rq) * (unsigned)p.T + (unsigned)col) << 4) : OOB, 0, 0);
        }
    };
    for (int st = st0; st < st1; st += 2) {
        sub_tile(std::integral_constant<int, 0>{}, st);
        if (st + 1 < st1) sub_tile(std::integral_constant<int, 1>{}, st + 1);
    }
}
