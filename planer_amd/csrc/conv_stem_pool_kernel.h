// Row-packed stem convolution + max-pool(3x3 / stride 2 / pad 1) in ONE kernel that never writes the full-resolution tensor --
// included by conv_direct.hip inside its anonymous namespace (uses Epilogue, apply_epilogue4).
//
// ResNet's stem (layer.Conv2d layer.py:22-26 -> util.conv_for util.py:17-44 on 3 input channels, 7x7 / stride 2 / pad 3, then
// BatchNorm + ReLU, then layer.Maxpool layer.py:71-72 -> util.pool util.py:79-95) writes 102.8 MB that the pool reads straight
// back (107 MB): 20 us of a forward go into a kernel that does no arithmetic, and two earlier fusions (a 15 x 17 pixel patch per
// 64 x 256 tile, then 64 x 128) lost to the unfused pair because every workgroup recomputed its patch's halo at one workgroup
// per CU.  Here a PERSISTENT workgroup (8 waves, one per CU: 32 images x 8 strips = 256 workgroups at batch 32) owns one
// image strip of 7 pooled rows and marches down its conv rows two at a time:
//   * the filter (44 k-quads x 64 channels x 16 B = 45 KB; K ordered (filter row, quad of the row's kw*Cin floats) exactly as
//     pl_conv2d_prepare_rowpack_f32 packs it) enters LDS once per workgroup and stays;
//   * per step the 9 packed input rows under a conv-row pair arrive by LDS-DMA into one of two row buffers (25 KB, requested
//     while the previous pair is multiplied) -- whole contiguous rows, no gather at staging time; the im2col gather happens at
//     the fragment read: pixel px of a conv row reads its 4 k-values at float 6 px + 4 jq of an input row (8-byte aligned
//     ds_read_b64 pairs);
//   * v_mfma_f32_16x16x4_f32: wave (mb, r) owns 16 output channels x the 112 pixels of conv row r of the pair = 7 accumulator
//     blocks (28 registers); a filter fragment is reused by the 7 blocks, 28 MFMAs per 15 LDS reads;
//   * the tail (bias / scale / shift / activation) runs on the accumulators -- a lane holds one pixel x one channel quad --
//     the horizontal 3-max comes from the neighbouring lanes (wave shuffles), and only the 56 pooled columns of a conv row
//     go to LDS (a ring of 4 rows, 14 KB each); the vertical 3-max and the 16-byte stores of ONE pooled row per step follow
//     the barrier.  One strip = 16 conv rows for 7 pooled rows: the row above the strip and one spare (14 % more MFMA work
//     than the conv alone) against 210 MB of traffic and a kernel launch.
// Padding semantics are the reference's: conv pixels outside the map count as 0 in a window (util.py:82), the running maximum
// starts at -1e4 (util.py:88,95).  max is exact, so the result equals conv kernel + pool kernel wherever the conv values agree
// (the K summation order here is one fmaf chain per output in this kernel's k order: equal to the tiled kernel's within
// rounding, tested against the oracle with the conv tolerance).
// Width: a workgroup covers NB blocks of 16 conv columns (template parameter, 1..7: registers and LDS are sized by it) of ONE
// column chunk of its strip.  A map up to 112 conv columns wide is one chunk starting at column 0 (the left neighbour of the
// first window is the pool's zero padding); wider maps are cut into chunks of `pq` pooled columns, chunk c > 0 starting one
// window early (conv column 2 c pq - 2, an even column, so window centres stay on even lanes) and not storing that first
// window -- two recomputed conv columns per chunk border.  Columns past the map's edge count as zero padding (masked before
// the horizontal maximum).  ResNet's 224-pixel stem is NB = 7, one chunk.
//
// NCHW = true: the workgroup reads the reference's NCHW tensor itself -- no row-packed copy of the input exists (that copy was a
// kernel of its own: 40 MB moved to re-lay a 19 MB batch).  The row buffer then holds one row per (input row, channel):
// [4 zeros | the row's pixels from column 2 x0 - 4 on], whole 16-byte cells of the NCHW row landing by LDS-DMA where the
// packed rows did (W % 4 == 0 keeps every cell inside or outside its row; cells outside arrive as zeros through the range
// check: they are the conv's padding).  K is ordered for it: a k-quad = four consecutive taps kw = 4 half - 1 .. 4 half + 2
// of one (filter row, channel) -- tap -1 is a zero filter value, so the seven taps of a row take two quads that start on
// 16-byte cells; conv column l reads its quad at float 2 l + 4 half of the LDS row (8-byte aligned pairs, as before).  The 42
// real quads + 2 zero quads are grouped so that the two quads a half-wave reads together (lanes 0-31: kk = 0, 1) sit two LDS
// rows apart: a row is 32 NB + 16 floats, two rows = 32 banks (mod 64) -- conflict-free like the packed layout.  Filter packed
// by pl_conv2d_prepare_stem_nchw_f32 in that order; everything behind the fragment reads is the same code.
struct StemPoolArgs {
    const float *xp;       // row-packed image [N][Hp][rowf] (zero border included); NCHW: the input tensor [N][3][H][W]
    const float *wq;       // [Qpad][Cout][4] row-packed filter
    float *y;              // pooled Q4 tensor [N][Coq][Hq][Wq][4]
    int N, Hp, rowf;       // packed rows per image, floats per packed row
    int H, W;              // NCHW: the input's height and width
    int Ho, Wo, Hq, Wq, Cout, Coq;
    int strips;            // strips of `prows` pooled rows per image
    int prows, steps;      // pooled rows per strip (7; 14: half the workgroups, 15 steps instead of 2 x 8), conv-row pairs per strip = prows + 1
    int cout_blocks;
    int chunks, pq;        // column chunks per strip, pooled columns per chunk
    unsigned x_bytes, w_bytes, y_bytes;
    Epilogue ep;
};

constexpr int SP_NB_MAX = 7;                     // 16-pixel blocks per conv row a workgroup can take (registers, LDS)
constexpr int SP_KH = 7, SP_RQ = 6;              // filter rows, k-quads per filter row (kw * Cin = 21 floats -> 6 quads)
constexpr int SP_GROUPS = (SP_KH * SP_RQ + 3) / 4;      // 11 groups of 4 k-quads (42 real + 2 zero quads)
constexpr int SP_XROWS = 9;                      // input rows under a conv-row pair: 2 r .. 2 r + 8
constexpr int SP_W_FLOATS = 4 * SP_GROUPS * 64 * 4;             // 44 quads x 64 channels x 4
constexpr int SP_PROWS = 7;                      // pooled rows per strip
constexpr int sp_xrow(int nb) { return 96 * nb + 32; }          // floats per staged input row: 6 per conv column + the last window's quads
constexpr int sp_xrow_nchw(int nb) { return 32 * nb + 16; }     // NCHW: floats per staged (input row, channel): 2 per conv column + 4 left + the last window's quad
// NCHW k order: group u reads (filter row, channel) pairs rho = 3 fr + c two apart -- kk = 0 / 2: rho_a (taps half 0 / 1), kk = 1 / 3:
// rho_b = rho_a + 2; group 10 pairs rho 20 with a zero-filter quad that reads row 18
__host__ __device__ constexpr int sp_nchw_rho(int u, int kk) { return u < 10 ? 4 * (u >> 1) + (u & 1) + 2 * (kk & 1) : ((kk & 1) ? 18 : 20); }
__host__ __device__ constexpr bool sp_nchw_real(int u, int kk) { return u < 10 || !(kk & 1); }

typedef float sp_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 sp_max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
template <int SP_NB, bool NCHW>
__global__ void __launch_bounds__(512) conv_stem_pool_kernel(const StemPoolArgs p) {
    constexpr int SP_XROW = NCHW ? sp_xrow_nchw(SP_NB) : sp_xrow(SP_NB), SP_X_FLOATS = (NCHW ? 3 : 1) * SP_XROWS * SP_XROW;
    constexpr int SP_PXF = NCHW ? 32 : 96;           // floats of a staged row per 16-column pixel block
    constexpr int SP_PC = 8 * SP_NB;                 // pooled columns a workgroup's conv columns hold
    constexpr int SP_HP_CELLS = 16 * SP_PC;          // one horizontally pooled conv row: 16 channel quads x SP_PC columns
    // separate LDS objects: an LDS-DMA into one row buffer must not hold up the fragment reads of the other (the compiler
    // orders LDS-DMA against later LDS accesses object by object)
    __shared__ __attribute__((aligned(16))) float Wf[SP_W_FLOATS];
    __shared__ __attribute__((aligned(16))) float X0[SP_X_FLOATS], X1[SP_X_FLOATS];
    __shared__ __attribute__((aligned(16))) float4 HP[4 * SP_HP_CELLS];
    typedef __attribute__((address_space(3))) float lds_float;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave & 3, rsel = wave >> 2;              // 16-channel block, conv row of the pair
    const int li = lane & 15, kk = lane >> 4;

    unsigned t1, t2, cob, n, strip, chunk;
    cob = blockIdx.x % (unsigned)p.cout_blocks;
    t1 = blockIdx.x / (unsigned)p.cout_blocks;
    chunk = t1 % (unsigned)p.chunks;
    t2 = t1 / (unsigned)p.chunks;
    n = t2 / (unsigned)p.strips;
    strip = t2 - n * (unsigned)p.strips;
    const int q0 = (int)chunk * p.pq;                       // first pooled column this workgroup stores
    const int x0 = chunk ? 2 * q0 - 2 : 0;                  // its first conv column; local pooled column l is global x0 / 2 + l
    const int lskip = chunk ? 1 : 0, qend = min(q0 + p.pq, p.Wq);
    const int p0 = (int)strip * p.prows;                    // first pooled row of the strip
    const int c0 = 2 * p0 - 1;                              // first conv row: the one above the first window's centre
    const int co0 = (int)cob * 64;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.xp), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wq), 0, p.w_bytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    // ---- the filter block: [44 quads][64 channels] cells of 16 bytes, 44 pieces of 64 cells ----
    for (int pc = wave; pc < 4 * SP_GROUPS; pc += 8) {     // piece pc = k-quad pc (64 channels = 64 cells)
        const int co = co0 + lane;
        const int off = co < p.Cout ? (int)(((unsigned)pc * (unsigned)p.Cout + (unsigned)co) << 4) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_float *)(Wf + pc * 256), 16, off, 0, 0, 0);
    }
    // ---- input rows of step t into buffer (t & 1): rows 2 (c0 + 2 t) .. + 8 of the packed image, 176 cells each ----
    auto load_rows = [&](int t, auto parity) {
        float *Xb = decltype(parity)::value ? X1 : X0;
        const int hr0 = 2 * (c0 + 2 * t);
        if constexpr (NCHW) {
            // 27 rows (input row r, channel c) of SP_XROW / 4 cells; cell j = input columns 2 x0 - 4 + 4 j .. + 3 of row hr0 - 3 + r
            constexpr int CPR = SP_XROW / 4, CELLS = 3 * SP_XROWS * CPR;
            for (int pc = wave; pc * 64 < CELLS; pc += 8) {
                const int idx = pc * 64 + lane;
                const int rr = idx / CPR, j = idx - rr * CPR;
                const int r = rr / 3, c = rr - 3 * r;
                const int h = hr0 - 3 + r, xin = 2 * x0 - 4 + 4 * j;
                const bool ok = idx < CELLS && (unsigned)h < (unsigned)p.H && xin >= 0 && xin + 3 < p.W;
                const int off = ok ? (int)(((((unsigned)n * 3u + (unsigned)c) * (unsigned)p.H + (unsigned)h) * (unsigned)p.W + (unsigned)xin) << 2) : OOB;
                if (idx < CELLS) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_float *)(Xb + pc * 256), 16, off, 0, 0, 0);
            }
        } else {
            constexpr int CELLS = SP_XROWS * (SP_XROW / 4);                    // 1584
            for (int pc = wave; pc * 64 < CELLS; pc += 8) {
                const int idx = pc * 64 + lane;
                const int r = idx / (SP_XROW / 4), c = idx - r * (SP_XROW / 4);
                const int hr = hr0 + r;
                const bool ok = idx < CELLS && (unsigned)hr < (unsigned)p.Hp && 6 * x0 + c * 4 < p.rowf + 3;
                const int off = ok ? (int)((((unsigned)n * (unsigned)p.Hp + (unsigned)hr) * (unsigned)p.rowf + (unsigned)(6 * x0) + 4u * (unsigned)c) << 2) : OOB;
                if (idx < CELLS) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_float *)(Xb + pc * 256), 16, off, 0, 0, 0);
            }
        }
    };
    load_rows(0, std::false_type{});

    // per-lane fragment offsets: group u, this lane's k-quad q = 4 u + kk -> (filter row, quad of the row); the two padding
    // quads (q = 42, 43: zero filter values) read the data of quads 0, 1
    int boff[SP_GROUPS];
#pragma unroll
    for (int u = 0; u < SP_GROUPS; ++u) {
        if constexpr (NCHW) {
            const int rho = (kk & 1) ? sp_nchw_rho(u, 1) : sp_nchw_rho(u, 0);
            boff[u] = (6 * rsel + rho) * SP_XROW + 4 * (kk >> 1) + 2 * li;       // + 32 nb per pixel block
        } else {
            int q = 4 * u + kk;
            if (q >= SP_KH * SP_RQ) q -= SP_KH * SP_RQ;
            const int fr = q / SP_RQ, jq = q - fr * SP_RQ;
            boff[u] = (2 * rsel + fr) * SP_XROW + 6 * li + 4 * jq;          // + 96 nb per pixel block
        }
    }
    const int aoff = (kk * 64 + mb * 16 + li) * 4;                          // + 1024 u: quad 4 u + kk, channel 16 mb + li

    // tail parameters of this lane's channel quad (a lane of the 16x16 C layout holds rows 4 (lane / 16) .. + 3 of its column)
    const int cq = (int)cob * 16 + mb * 4 + kk;
    const int cqc = min(cq, p.Coq - 1);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bias = p.ep.bias ? reinterpret_cast<const float4 *>(p.ep.bias)[cqc] : z4;
    const float4 scale = p.ep.scale ? reinterpret_cast<const float4 *>(p.ep.scale)[cqc] : one4;
    const float4 shift = p.ep.shift ? reinterpret_cast<const float4 *>(p.ep.shift)[cqc] : z4;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const bool plain = !p.ep.bias && p.ep.scale && p.ep.shift && p.ep.act == 1;
    __syncthreads();                                                         // filter and rows of step 0 have landed

    // Two accumulator sets: step t multiplies into set (t & 1) while the TAIL of step t - 1 (other set) and the pooling of row
    // t - 2 ride between its MFMA groups -- one pixel block's tail per group (a few dozen VALU / DPP instructions and one LDS
    // write in the shadow of 28 MFMAs), an LDS-only barrier in mid-loop, the pooled row's loads, maxima and stores after it.
    // Run as its own phase the tail cost 2.5 us per step with the matrix pipe idle (20 us of 102).
    sp_f32x4 accA[SP_NB], accB[SP_NB];
    float4 vprev = z4;                                           // tail output of the previous pixel block (its lane 15 is a neighbour)
    const float4 lo4 = make_float4(-1e4f, -1e4f, -1e4f, -1e4f);

    // tail of pixel block nb of the step whose conv row (of the strip) is k: fused tail, horizontal 3-max, pooled columns -> LDS
    auto tail_block = [&](const sp_f32x4 (&acc)[SP_NB], int nb, int k) {
        const bool row_in = (unsigned)(c0 + k) < (unsigned)p.Ho;         // rows outside the map are the pool's zero padding
        float4 v;
        if (plain) {
            // the tail a ResNet stem carries (scale, shift, ReLU; no bias) written straight (apply_epilogue4's run-time options
            // compile to per-element selects): multiply and add on register pairs (v_pk_mul_f32 / v_pk_add_f32, separate
            // roundings as in the reference's two passes), and the ReLU moved BEHIND the pooling -- x -> x (x > 0) is monotonic and
            // maps the zero padding to itself, so max and ReLU commute (value for value; a window of negatives gives -0.0 either
            // way) -- where it runs on a quarter of the values.  fp32 VALU work beside fp32 MFMAs is paid in matrix time.
            typedef float sp_v2 __attribute__((ext_vector_type(2)));
            const sp_v2 lo = (sp_v2){acc[nb][0], acc[nb][1]} * (sp_v2){scale.x, scale.y} + (sp_v2){shift.x, shift.y};
            const sp_v2 hi = (sp_v2){acc[nb][2], acc[nb][3]} * (sp_v2){scale.z, scale.w} + (sp_v2){shift.z, shift.w};
            v = make_float4(lo.x, lo.y, hi.x, hi.y);
        } else {
            v = apply_epilogue4(p.ep, bias, scale, shift, z4, 4, make_float4(acc[nb][0], acc[nb][1], acc[nb][2], acc[nb][3]));
        }
        if (!row_in || x0 + 16 * nb + li >= p.Wo) v = z4;               // (columns past the map's edge: zero padding too)
        // window of pooled column q = 8 nb + li / 2 (even lanes): pixels 16 nb + li - 1, li, li + 1.  Neighbours by DPP within
        // the 16-lane row (one VALU move each; as ds_bpermute -- an LDS instruction -- the 84 shuffles of a step cost 6 us per
        // launch): lane li - 1 by row_shr:1, whose lane 0 keeps `old` = lane 15 of the previous pixel block (row_ror:1 of it)
        // or the zero padding; lane li + 1 by row_shl:1.
        float4 m;
        float *mo = reinterpret_cast<float *>(&m);
        const float *cv = reinterpret_cast<const float *>(&v);
        const float *pv = reinterpret_cast<const float *>(&vprev);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = __builtin_bit_cast(int, cv[e]);
            const int prev15 = nb > 0 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, pv[e]), 0x121, 0xf, 0xf, false) : 0;   // row_ror:1
            const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(prev15, c, 0x111, 0xf, 0xf, false));              // row_shr:1
            const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, c, 0x101, 0xf, 0xf, false));                  // row_shl:1
            mo[e] = fmaxf(fmaxf(left, cv[e]), right);
        }
        vprev = v;
        float4 *hp = HP + (k & 3) * SP_HP_CELLS + (mb * 4 + kk) * SP_PC;
        if ((li & 1) == 0) hp[8 * nb + (li >> 1)] = m;
    };
    // half of pooled row j (cells i = tid + 512 * half): rows 2 j, 2 j + 1, 2 j + 2 of the ring, running maximum from -1e4
    auto pool_half = [&](int j, int half) {
        const int prow = p0 + j, i = tid + 512 * half;
        if (i < SP_HP_CELLS && prow < p.Hq) {
            const float4 *r0 = HP + ((2 * j) & 3) * SP_HP_CELLS, *r1 = HP + ((2 * j + 1) & 3) * SP_HP_CELLS,
                         *r2 = HP + ((2 * j + 2) & 3) * SP_HP_CELLS;
            const int c = i / SP_PC, ql = i - c * SP_PC, q = (x0 >> 1) + ql;
            float4 m = sp_max4(sp_max4(sp_max4(lo4, r0[i]), r1[i]), r2[i]);
            if (plain) m = make_float4(relu_ref(m.x), relu_ref(m.y), relu_ref(m.z), relu_ref(m.w));
            const int cqo = (int)cob * 16 + c;
            const int off = (cqo < p.Coq && ql >= lskip && q < qend)
                                ? (int)(((((unsigned)n * (unsigned)p.Coq + (unsigned)cqo) * (unsigned)p.Hq + (unsigned)prow) * (unsigned)p.Wq + (unsigned)q) << 4)
                                : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, m), yrsrc, off, 0, 0);
        }
    };
    auto lds_barrier = [&]() {          // orders LDS traffic only: __syncthreads() also waits for the wave's global stores (vmcnt)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    auto step = [&](int t, auto parity) {
        constexpr int cur = decltype(parity)::value;
        const float *Xb = cur ? X1 : X0;
        sp_f32x4 (&acc)[SP_NB] = cur ? accB : accA;
        const sp_f32x4 (&old)[SP_NB] = cur ? accA : accB;
        if (t + 1 < p.steps) load_rows(t + 1, std::integral_constant<bool, !cur>{});
#pragma unroll
        for (int nb = 0; nb < SP_NB; ++nb) acc[nb] = (sp_f32x4){0.f, 0.f, 0.f, 0.f};
        // fragments one group ahead, in two register sets that alternate (no copies: a copy is a v_mov, and fp32 VALU work
        // beside fp32 MFMAs costs matrix time); within a group the MFMAs run element by element over the 7 pixel blocks, so
        // two MFMAs on one accumulator are 7 issues apart (back to back they would wait out the 40-cycle dependent latency)
        float4 fa[2];
        float2 fb0[2][SP_NB], fb1[2][SP_NB];
        auto fetch = [&](int u, int set) {
            fa[set] = *reinterpret_cast<const float4 *>(Wf + aoff + 1024 * u);
#pragma unroll
            for (int nb = 0; nb < SP_NB; ++nb) {
                const float *src = Xb + boff[u] + SP_PXF * nb;
                fb0[set][nb] = *reinterpret_cast<const float2 *>(src);
                fb1[set][nb] = *reinterpret_cast<const float2 *>(src + 2);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int u = 0; u < SP_GROUPS; ++u) {
            const int set = u & 1;
            if (u + 1 < SP_GROUPS) fetch(u + 1, set ^ 1);
#pragma unroll
            for (int nb = 0; nb < SP_NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set].x, fb0[set][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < SP_NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set].y, fb0[set][nb].y, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < SP_NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set].z, fb1[set][nb].x, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < SP_NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set].w, fb1[set][nb].y, acc[nb], 0, 0, 0);
            // what rides along: the previous step's tail (two pixel blocks per group, groups 0-3), the hand-over barrier, the
            // pooled row (groups 4, 5) -- early enough for its stores to be complete when the step's closing barrier asks
            // (vmcnt counts stores: issued in groups 8-9 they cost ~1 us of exposed latency per step)
            if (t >= 1) {
                if (2 * u < SP_NB) tail_block(old, 2 * u, 2 * (t - 1) + rsel);
                if (2 * u + 1 < SP_NB) tail_block(old, 2 * u + 1, 2 * (t - 1) + rsel);
                if (u == 3) lds_barrier();                     // rows 2 t - 2, 2 t - 1 of the ring are complete
                if (t >= 2 && (u == 4 || u == 5)) pool_half(t - 2, u - 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next rows have landed (vmcnt), this step's row buffer and the ring rows read by the pooling are free again
        __syncthreads();
    };
    for (int t = 0; t < p.steps; t += 2) {
        step(t, std::false_type{});
        if (t + 1 < p.steps) step(t + 1, std::true_type{});
    }
    // drain: tail of the last step (its accumulator set by parity), last pooled row
    auto drain = [&](const sp_f32x4 (&old)[SP_NB]) {
#pragma unroll
        for (int nb = 0; nb < SP_NB; ++nb) tail_block(old, nb, 2 * (p.steps - 1) + rsel);
        lds_barrier();
        pool_half(p.prows - 1, 0);
        pool_half(p.prows - 1, 1);
    };
    if ((p.steps - 1) & 1) drain(accB);
    else drain(accA);
}

// OIHW stem filter [Cout][3][7][7] -> [k-quad q][Cout][4] in the NCHW kernel's k order: quad q = 4 u + kk holds taps
// kw = 4 (kk >> 1) - 1 .. + 3 of (filter row, channel) rho = sp_nchw_rho(u, kk); tap -1, the padding quad of group 10 and
// quads 44 .. 47 are zeros.
__global__ void __launch_bounds__(256) pack_filter_stem_nchw_kernel(const float *w, float4 *out, unsigned total, int Cout) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i / (unsigned)Cout), co = (int)(i - (unsigned)q * (unsigned)Cout);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int u = q >> 2, kk = q & 3;
    if (u < SP_GROUPS && sp_nchw_real(u, kk)) {
        const int rho = sp_nchw_rho(u, kk), fr = rho / 3, c = rho - 3 * fr, half = kk >> 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kw = 4 * half + j - 1;
            if (kw >= 0 && kw < 7) v[j] = w[((co * 3 + c) * SP_KH + fr) * 7 + kw];
        }
    }
    out[i] = make_float4(v[0], v[1], v[2], v[3]);
}
