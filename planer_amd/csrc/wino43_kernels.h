// Winograd with MIXED tiles for maps whose side is 7, 14 or 21 -- included by conv_winograd.hip inside its anonymous namespace.
//
// F(4x4,3x3) cuts a map into 4x4 output tiles, so a 14x14 map is computed as 16x16 and a 7x7 map as 8x8: 31 % of the
// multiplies of ResNet-18's layer3 / layer4 convolutions (reference: layer.Conv2d layer.py:22-26 -> util.conv_for
// util.py:17-44) fall on tiles or tile parts outside the map.  Here a side of 7a pixels is cut into a segments of 4 and a
// segments of 3 (14 = 4 + 4 + 3 + 3, 7 = 4 + 3): a segment of 4 is F(4,3) (6 frequencies, points 0, +-1, +-2, inf -- the
// transforms of the F(4x4,3x3) path), a segment of 3 is F(3,3) (5 frequencies, points 0, +-1, 2, inf):
//   B5^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
//   G5   = [1/2 0 0; -1/2 -1/2 -1/2; -1/6 1/6 -1/6; 1/6 1/3 2/3; 0 0 1]
//   A5^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]
// A tile is (row segment) x (column segment): four tile CLASSES (6x6, 6x5, 5x6, 5x5 frequencies) with a*a tiles each per image
// -- equally many, so the 36 + 30 + 30 + 25 = 121 per-frequency GEMMs all have N*a*a columns and run as ONE grouped 1x1
// convolution on conv_q4_kernel, exactly like the 36 of the F(4x4) path.  (6+6+5+5)^2 / 24^2 = 0.84 of that path's multiplies,
// no padding tile, 3.4x the filter bytes.  V / M are [121][C/4][N*a*a][4]; frequency group = base(class) + i * nb + j.
// F(3,3)'s constants are smaller than F(4,3)'s: the error stays that of the F(4x4) path (tests: <= 3e-5 of max|y|).
typedef float w43_f4 __attribute__((ext_vector_type(4)));

// row A of B^T d and of A^T m for a segment with NA frequencies; the whole-vector forms are DEFINED through the rows, so that
// kernels that transform one row per thread (the LDS kernel) and kernels that transform a whole tile per thread agree bit for bit
template <int NA, int A, class T>
__device__ __forceinline__ T w43_bt_row(const T (&d)[6]) {
    if constexpr (NA == 6) {
        if constexpr (A == 0) return 4.f * d[0] - 5.f * d[2] + d[4];
        else if constexpr (A == 1) return -4.f * (d[1] + d[2]) + d[3] + d[4];
        else if constexpr (A == 2) return 4.f * (d[1] - d[2]) - d[3] + d[4];
        else if constexpr (A == 3) return 2.f * (d[3] - d[1]) - d[2] + d[4];
        else if constexpr (A == 4) return 2.f * (d[1] - d[3]) - d[2] + d[4];
        else return 4.f * d[1] - 5.f * d[3] + d[5];
    } else {
        if constexpr (A == 0) return 2.f * (d[0] - d[2]) - d[1] + d[3];
        else if constexpr (A == 1) return d[3] - d[2] - 2.f * d[1];
        else if constexpr (A == 2) return 2.f * d[1] - 3.f * d[2] + d[3];
        else if constexpr (A == 3) return d[3] - d[1];
        else return 2.f * (d[1] - d[3]) - d[2] + d[4];
    }
}
template <int NA, int A, class T>
__device__ __forceinline__ T w43_at_row(const T (&m)[6]) {
    if constexpr (NA == 6) {
        if constexpr (A == 0) return m[0] + (m[1] + m[2]) + (m[3] + m[4]);
        else if constexpr (A == 1) return (m[1] - m[2]) + 2.f * (m[3] - m[4]);
        else if constexpr (A == 2) return (m[1] + m[2]) + 4.f * (m[3] + m[4]);
        else return (m[1] - m[2]) + 8.f * (m[3] - m[4]) + m[5];
    } else {
        if constexpr (A == 0) return m[0] + (m[1] + m[2]) + m[3];
        else if constexpr (A == 1) return (m[1] - m[2]) + 2.f * m[3];
        else return (m[1] + m[2]) + 4.f * m[3] + m[4];
    }
}
template <int NA, class T>
__device__ __forceinline__ void w43_bt(const T (&d)[6], T (&o)[6]) {          // o = B^T d (the first NA entries)
    o[0] = w43_bt_row<NA, 0>(d);
    o[1] = w43_bt_row<NA, 1>(d);
    o[2] = w43_bt_row<NA, 2>(d);
    o[3] = w43_bt_row<NA, 3>(d);
    o[4] = w43_bt_row<NA, 4>(d);
    if constexpr (NA == 6) o[5] = w43_bt_row<6, 5>(d);
    else o[5] = d[5];
}
template <int NA, class T>
__device__ __forceinline__ void w43_at(const T (&m)[6], T (&o)[4]) {          // o = A^T m: 4 (NA = 6) or 3 (NA = 5) outputs
    o[0] = w43_at_row<NA, 0>(m);
    o[1] = w43_at_row<NA, 1>(m);
    o[2] = w43_at_row<NA, 2>(m);
    if constexpr (NA == 6) o[3] = w43_at_row<6, 3>(m);
    else o[3] = o[2];
}
__device__ __forceinline__ void w43_g(int na, float g0, float g1, float g2, float (&o)[6]) {        // o = G g
    if (na == 6) {
        o[0] = g0 * 0.25f;
        o[1] = -(g0 + g1 + g2) * (1.f / 6.f);
        o[2] = (-g0 + g1 - g2) * (1.f / 6.f);
        o[3] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        o[4] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        o[5] = g2;
    } else {
        o[0] = g0 * 0.5f;
        o[1] = -(g0 + g1 + g2) * 0.5f;
        o[2] = (-g0 + g1 - g2) * (1.f / 6.f);
        o[3] = g0 * (1.f / 6.f) + g1 * (1.f / 3.f) + g2 * (2.f / 3.f);
        o[4] = g2;
        o[5] = 0.f;
    }
}
__host__ __device__ __forceinline__ int w43_fbase(int rc, int cc) { return rc ? (cc ? 96 : 66) : (cc ? 36 : 0); }
constexpr int W43_GROUPS = 121;

struct W43Args {
    const float *x;        // Q4 activation [N][Cq][H][W][4] (input transform) / residual comes in ep.res
    const float *M;        // [121][Cq][T][4]
    float *V;              // [121][Cq][T][4]
    float *y;              // Q4 output, or null (chain: nobody else reads it)
    int N, Cq, H, W;
    int ar, ac, TC, T;     // tiles per class per image along y / x, their product, N * TC
    int G;                 // chain kernel: channel quads per workgroup
    unsigned x_bytes;      // bytes of an (N, 4 Cq, H, W) Q4 tensor
    FastDiv divCT, divT, divTC, divAc;
    Epilogue ep;
};

// uq[f][q = cin/4][co][cin%4] for the 121 frequency groups, zero padded to Qpad k-quads
__global__ void __launch_bounds__(256) wino43_filter_q4_kernel(const float *w, float *Uq, unsigned total, int Cin, int Cout, int Qpad) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;   // (co, c) pair
    if (i >= total) return;
    const int co = (int)(i / (unsigned)Cin), c = (int)(i - (unsigned)co * Cin);
    const float *g = w + (size_t)i * 9;
    const size_t plane = (size_t)Qpad * Cout * 4;
    float *up = Uq + ((size_t)(c >> 2) * Cout + co) * 4 + (c & 3);
    for (int rc = 0; rc < 2; ++rc)
        for (int cc = 0; cc < 2; ++cc) {
            const int na = 6 - rc, nb = 6 - cc, fb = w43_fbase(rc, cc);
            float t[6][3];                               // t = G_na g   (na x 3)
            for (int j = 0; j < 3; ++j) {
                float o[6];
                w43_g(na, g[j], g[3 + j], g[6 + j], o);
                for (int a = 0; a < 6; ++a) t[a][j] = o[a];
            }
            for (int a = 0; a < na; ++a) {
                float o[6];
                w43_g(nb, t[a][0], t[a][1], t[a][2], o);
                for (int b = 0; b < nb; ++b) up[(size_t)(fb + a * nb + b) * plane] = o[b];
            }
        }
}

// ---- input transform: thread = (class, channel quad, tile, lane of the quad), scalar floats like wino4_input_q4_kernel ----
template <int RC, int CC>
__device__ __forceinline__ void w43_input_tile(const W43Args &p, const __amdgpu_buffer_rsrc_t xrsrc, unsigned cq, unsigned t, unsigned e) {
    constexpr int NA = 6 - RC, NB = 6 - CC;
    unsigned n, tl, tyc, txc;
    p.divTC.divmod(t, n, tl);
    p.divAc.divmod(tl, tyc, txc);
    const int h0 = (RC ? 4 * p.ar + 3 * (int)tyc : 4 * (int)tyc) - 1, w0 = (CC ? 4 * p.ac + 3 * (int)txc : 4 * (int)txc) - 1;
    const int xbase = (int)(((n * (unsigned)p.Cq + cq) * (unsigned)(p.H * p.W)) * 4 + e);
    float dd[6][6];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int wi = w0 + b;
        const bool wok = (unsigned)wi < (unsigned)p.W;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int hi = h0 + a;
            const bool ok = wok && (unsigned)hi < (unsigned)p.H;
            dd[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     xrsrc, ok ? (xbase + (hi * p.W + wi) * 4) << 2 : (int)0x80000000, 0, 0));
        }
    }
    float m[6][6];
#pragma unroll
    for (int b = 0; b < NB; ++b) {                        // columns first: m[.][b] = B^T d[.][b]
        float d[6], o[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) d[a] = a < NA ? dd[a][b] : 0.f;
        w43_bt<NA>(d, o);
#pragma unroll
        for (int a = 0; a < NA; ++a) m[a][b] = o[a];
    }
    const size_t plane = (size_t)p.Cq * p.T * 4;
    float *vp = p.V + ((size_t)cq * p.T + t) * 4 + e;
    constexpr int FB = RC ? (CC ? 96 : 66) : (CC ? 36 : 0);
#pragma unroll
    for (int a = 0; a < NA; ++a) {                        // then rows
        float d[6], o[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) d[b] = b < NB ? m[a][b] : 0.f;
        w43_bt<NB>(d, o);
#pragma unroll
        for (int b = 0; b < NB; ++b) vp[(size_t)(FB + a * NB + b) * plane] = o[b];
    }
}
__global__ void __launch_bounds__(256) wino43_input_q4_kernel(const W43Args p, unsigned total) {
    const unsigned stride = gridDim.x * 256;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, p.x_bytes, 0x00020000);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const unsigned e = i & 3, it = i >> 2;
        unsigned cls, r, cq, t;
        p.divCT.divmod(it, cls, r);
        p.divT.divmod(r, cq, t);
        switch (cls) {
        case 0: w43_input_tile<0, 0>(p, xrsrc, cq, t, e); break;
        case 1: w43_input_tile<0, 1>(p, xrsrc, cq, t, e); break;
        case 2: w43_input_tile<1, 0>(p, xrsrc, cq, t, e); break;
        default: w43_input_tile<1, 1>(p, xrsrc, cq, t, e); break;
        }
    }
}

// ---- output transform + fused tail: thread = (class, channel quad, tile) on float4s like wino4_output_q4_kernel ----
template <int RC, int CC>
__device__ __forceinline__ void w43_output_tile(const W43Args &p, const float4 *M4, unsigned coq, unsigned t,
                                                const __amdgpu_buffer_rsrc_t yrsrc, const __amdgpu_buffer_rsrc_t rrsrc, float4 *plane, int pitch) {
    constexpr int NA = 6 - RC, NB = 6 - CC, MA = 4 - RC, MB = 4 - CC;
    constexpr int FB = RC ? (CC ? 96 : 66) : (CC ? 36 : 0);
    unsigned n, tl, tyc, txc;
    p.divTC.divmod(t, n, tl);
    p.divAc.divmod(tl, tyc, txc);
    const int h0 = RC ? 4 * p.ar + 3 * (int)tyc : 4 * (int)tyc, w0 = CC ? 4 * p.ac + 3 * (int)txc : 4 * (int)txc;
    const size_t mplane = (size_t)p.Cq * p.T;
    const float4 *mp = M4 + (size_t)coq * p.T + t;
    w43_f4 s[4][6];
#pragma unroll
    for (int b = 0; b < NB; ++b) {                        // columns: s[.][b] = A^T m[.][b]
        w43_f4 m[6], o[4];
#pragma unroll
        for (int a = 0; a < 6; ++a)
            m[a] = a < NA ? __builtin_bit_cast(w43_f4, mp[(size_t)(FB + a * NB + b) * mplane]) : (w43_f4){0.f, 0.f, 0.f, 0.f};
        w43_at<NA>(m, o);
#pragma unroll
        for (int a = 0; a < MA; ++a) s[a][b] = o[a];
    }
    float bs[4], sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) load_chan_params(p.ep, (int)coq * 4 + e, bs[e], sc[e], sh[e]);
    const float4 bias = make_float4(bs[0], bs[1], bs[2], bs[3]), scale = make_float4(sc[0], sc[1], sc[2], sc[3]);
    const float4 shift = make_float4(sh[0], sh[1], sh[2], sh[3]);
    int off[MA][MB];
    float4 rs[MA][MB];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b) {                    // (exact tiling: every output pixel of a tile is inside the map)
            off[a][b] = (int)((((n * (unsigned)p.Cq + coq) * (unsigned)p.H + (unsigned)(h0 + a)) * (unsigned)p.W + (unsigned)(w0 + b)) << 4);
            rs[a][b] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[a][b], 0, 0));
        }
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        w43_f4 d[6], o[4];
#pragma unroll
        for (int b = 0; b < 6; ++b) d[b] = b < NB ? s[a][b] : (w43_f4){0.f, 0.f, 0.f, 0.f};
        w43_at<NB>(d, o);
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const float4 v = apply_epilogue4(p.ep, bias, scale, shift, rs[a][b], 4, __builtin_bit_cast(float4, o[b]));
            if (plane) plane[(h0 + a + 1) * pitch + w0 + b + 1] = v;
            if (p.y)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                                       yrsrc, off[a][b], 0, 0);
        }
    }
}
__global__ void __launch_bounds__(256) wino43_output_q4_kernel(const W43Args p, unsigned total) {
    const unsigned stride = gridDim.x * 256;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ep.res), 0, p.ep.res ? p.x_bytes : 0u, 0x00020000);
    const float4 *M4 = reinterpret_cast<const float4 *>(p.M);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        unsigned cls, r, coq, t;
        p.divCT.divmod(i, cls, r);
        p.divT.divmod(r, coq, t);
        switch (cls) {
        case 0: w43_output_tile<0, 0>(p, M4, coq, t, yrsrc, rrsrc, nullptr, 0); break;
        case 1: w43_output_tile<0, 1>(p, M4, coq, t, yrsrc, rrsrc, nullptr, 0); break;
        case 2: w43_output_tile<1, 0>(p, M4, coq, t, yrsrc, rrsrc, nullptr, 0); break;
        default: w43_output_tile<1, 1>(p, M4, coq, t, yrsrc, rrsrc, nullptr, 0); break;
        }
    }
}

// ---- the LDS kernel: one workgroup = one image x G channel quads, whole planes (zero border included) in LDS.
//   P0   M (the products of conv k) -> an LDS slab [121][G * TC] by LDS-DMA, lane-linear pieces; the plane's interior <- the
//        residual of conv k (or x for a lone input transform, or zeros)
//   P1a  item = (quad, tile, frequency column b): s[.][b] = A^T m[.][b], in place in the slab          (one 1-D transform per thread)
//   P1b  item = (quad, tile, output row a'):      y[a'][.] = A^T s[a'][.], fused tail (the residual is what the plane cell holds),
//        into the plane
//   P2a  item = (quad, tile, patch column j):     m[.][j] = B^T d[.][j] out of the plane -> the slab region
//   P2b  item = (quad, tile, frequency row a):    V[a][.] = B^T m[a][.] -> V of conv k + 1 (16-byte stores)
//   P3   the plane's interior -> y, pixel order, when something other than conv k + 1 reads it
// Items are ordered tile class first, so a wave runs one variant of its 1-D transform (a pass depends on the class of ONE
// dimension only).  Lone input transform: P0 (x), P2.  Lone output transform: P0, P1, P3.  Every pass applies the same w43_at /
// w43_bt to the same values as the whole-tile kernels above: chained and unchained plans agree bit for bit.
struct W43LdsArgs {
    W43Args a;
    int gt;                // G * TC: tiles of one class in a workgroup
    int pcells;            // cells per LDS plane, (H + 2) * (W + 2) rounded so that consecutive quads start 4 banks apart
    int from_m;            // the slab phase runs (M is the source); otherwise x is
    FastDiv divGt, div2Gt, divPcells, divPitch, divHW, divW;
};

// geometry of tile tl of class (rc, cc): first output pixel
__device__ __forceinline__ void w43_tile_origin(const W43Args &p, int rc, int cc, unsigned tl, int &h0, int &w0) {
    unsigned tyc, txc;
    p.divAc.divmod(tl, tyc, txc);
    h0 = rc ? 4 * p.ar + 3 * (int)tyc : 4 * (int)tyc;
    w0 = cc ? 4 * p.ac + 3 * (int)txc : 4 * (int)txc;
}

__global__ void __launch_bounds__(512) wino43_lds_kernel(const W43LdsArgs q) {
    extern __shared__ float4 w43_lds[];
    const W43Args &p = q.a;
    const unsigned tid = threadIdx.x, bd = blockDim.x, lane = tid & 63u;
    const unsigned n = blockIdx.y, cq0 = blockIdx.x * (unsigned)p.G, gt = (unsigned)q.gt;
    const int pitch = p.W + 2;
    const unsigned HW = (unsigned)(p.H * p.W);
    float4 *plane = w43_lds;                                      // [G][pcells]
    // [121][gt]: the products M; P1a transforms the columns of every tile IN PLACE (a thread owns one column of one tile), P1b
    // reads rows; P2a leaves the half-transformed patches here in the same (frequency, tile) addressing, P2b reads rows.  Lanes
    // are consecutive tiles: every access of a wave is to consecutive 16-byte cells.
    float4 *slab = w43_lds + (size_t)p.G * q.pcells;
    float4 *prm = slab + (size_t)W43_GROUPS * gt;                 // [3][G]
    // ---- P0 ----
    if (q.from_m) {
        const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.M), 0, (unsigned)((size_t)W43_GROUPS * p.Cq * p.T * 16), 0x00020000);
        const unsigned cells = (unsigned)W43_GROUPS * gt;
        for (unsigned i0 = tid - lane; i0 < cells; i0 += bd) {
            const unsigned i = i0 + lane;
            if (i < cells) {
                unsigned f, r, cql, tl;
                q.divGt.divmod(i, f, r);
                p.divTC.divmod(r, cql, tl);
                const unsigned src = ((f * (unsigned)p.Cq + cq0 + cql) * (unsigned)p.T + n * (unsigned)p.TC + tl) << 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(mrsrc, (__attribute__((address_space(3))) float *)(slab + i0), 16, (int)src, 0, 0, 0);
            }
        }
        if (tid < 3u * (unsigned)p.G) {
            const unsigned which = tid / (unsigned)p.G, cql = tid - which * (unsigned)p.G;
            const float *src = which == 0 ? p.ep.bias : which == 1 ? p.ep.scale : p.ep.shift;
            const float fill = which == 1 ? 1.f : 0.f;
            prm[tid] = src ? *reinterpret_cast<const float4 *>(src + (size_t)(cq0 + cql) * 4) : make_float4(fill, fill, fill, fill);
        }
    }
    {
        // the planes: zero border; interior = the residual (from_m), x (lone input transform) or zeros.  Every request of a
        // thread goes out before its first LDS write (one round trip, not one per cell)
        const float4 *src4 = reinterpret_cast<const float4 *>(q.from_m ? p.ep.res : p.x);
        const unsigned cells = (unsigned)p.G * (unsigned)q.pcells;
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(src4), 0, src4 ? p.x_bytes : 0u, 0x00020000);
        for (unsigned c0 = tid; c0 < cells; c0 += 4 * bd) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned c = c0 + (unsigned)u * bd;
                unsigned cql, rem, r_, x_;
                q.divPcells.divmod(c, cql, rem);
                q.divPitch.divmod(rem, r_, x_);
                const int h = (int)r_ - 1, w = (int)x_ - 1;
                const bool in = c < cells && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                const unsigned off = (((n * (unsigned)p.Cq + cq0 + cql) * HW) + (unsigned)h * (unsigned)p.W + (unsigned)w) << 4;
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srsrc, in ? (int)off : (int)0x80000000, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned c = c0 + (unsigned)u * bd;
                if (c < cells) plane[c] = v[u];
            }
        }
    }
    __syncthreads();
    // Item numbering of a pass whose 1-D transform depends on the class of dimension D: lines of class 0 first (6 or 4 per tile),
    // then lines of class 1 (5 or 3 per tile); inside a class: line index slowest, then the OTHER dimension's class, then
    // (quad, tile) -- consecutive lanes are consecutive tiles.
    // line counts: P1a columns b < nb (class cc), P1b rows a' < ma (class rc), P2a columns j < nb (cc), P2b rows a < na (rc)
    auto decode = [&](unsigned j, unsigned n0, unsigned n1, unsigned &cd, unsigned &line, unsigned &co, unsigned &r) {
        // n0 / n1 lines per tile for class 0 / 1 of the pass's dimension; 2 gt tiles per class of that dimension
        const unsigned per0 = n0 * 2u * gt;
        cd = j >= per0;
        const unsigned jj = cd ? j - per0 : j;
        unsigned rem;
        q.div2Gt.divmod(jj, line, rem);
        co = rem >= gt;
        r = co ? rem - gt : rem;
        (void)n1;
    };
    if (q.from_m) {
        // ---- P1a: columns of the frequency tile ----
        for (unsigned j = tid; j < 22u * gt; j += bd) {
            unsigned cc, b, rc, r;
            decode(j, 6, 5, cc, b, rc, r);
            const int na = 6 - (int)rc, nb = 6 - (int)cc, fb = w43_fbase((int)rc, (int)cc);
            w43_f4 m[6], o[4];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                m[k] = k < na ? __builtin_bit_cast(w43_f4, slab[(unsigned)(fb + k * nb + (int)b) * gt + r]) : (w43_f4){0.f, 0.f, 0.f, 0.f};
            if (rc) w43_at<5>(m, o);
            else w43_at<6>(m, o);
            const int ma = 4 - (int)rc;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < ma) slab[(unsigned)(fb + a * nb + (int)b) * gt + r] = __builtin_bit_cast(float4, o[a]);
        }
        __syncthreads();
        // ---- P1b: rows of the output tile, fused tail, into the plane ----
        for (unsigned j = tid; j < 14u * gt; j += bd) {
            unsigned rc, a, cc, r, cql, tl;
            decode(j, 4, 3, rc, a, cc, r);
            p.divTC.divmod(r, cql, tl);
            int h0, w0;
            w43_tile_origin(p, (int)rc, (int)cc, tl, h0, w0);
            const int nb = 6 - (int)cc, mb = 4 - (int)cc, fb = w43_fbase((int)rc, (int)cc);
            w43_f4 d[6], o[4];
#pragma unroll
            for (int b = 0; b < 6; ++b)
                d[b] = b < nb ? __builtin_bit_cast(w43_f4, slab[(unsigned)(fb + (int)a * nb + b) * gt + r]) : (w43_f4){0.f, 0.f, 0.f, 0.f};
            if (cc) w43_at<5>(d, o);
            else w43_at<6>(d, o);
            const float4 bias = prm[cql], scale = prm[p.G + cql], shift = prm[2 * p.G + cql];
            float4 *row = plane + (size_t)cql * q.pcells + (h0 + (int)a + 1) * pitch + w0 + 1;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < mb) row[b] = apply_epilogue4(p.ep, bias, scale, shift, row[b], 4, __builtin_bit_cast(float4, o[b]));
        }
        __syncthreads();
    }
    if (p.V) {
        // ---- P2a: columns of the input patch ----
        for (unsigned j = tid; j < 22u * gt; j += bd) {
            unsigned cc, col, rc, r, cql, tl;
            decode(j, 6, 5, cc, col, rc, r);
            p.divTC.divmod(r, cql, tl);
            int h0, w0;
            w43_tile_origin(p, (int)rc, (int)cc, tl, h0, w0);
            const int na = 6 - (int)rc, nb = 6 - (int)cc, fb = w43_fbase((int)rc, (int)cc);
            const float4 *pl = plane + (size_t)cql * q.pcells + h0 * pitch + w0 + (int)col;      // plane coordinates include the border
            w43_f4 d[6], o[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = k < na ? __builtin_bit_cast(w43_f4, pl[k * pitch]) : (w43_f4){0.f, 0.f, 0.f, 0.f};
            if (rc) w43_bt<5>(d, o);
            else w43_bt<6>(d, o);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < na) slab[(unsigned)(fb + k * nb + (int)col) * gt + r] = __builtin_bit_cast(float4, o[k]);
        }
        __syncthreads();
        // ---- P2b: rows of the frequency tile -> V ----
        const size_t vplane = (size_t)p.Cq * p.T;
        for (unsigned j = tid; j < 22u * gt; j += bd) {
            unsigned rc, a, cc, r, cql, tl;
            decode(j, 6, 5, rc, a, cc, r);
            p.divTC.divmod(r, cql, tl);
            const int nb = 6 - (int)cc, fb = w43_fbase((int)rc, (int)cc);
            w43_f4 d[6], o[6];
#pragma unroll
            for (int b = 0; b < 6; ++b)
                d[b] = b < nb ? __builtin_bit_cast(w43_f4, slab[(unsigned)(fb + (int)a * nb + b) * gt + r]) : (w43_f4){0.f, 0.f, 0.f, 0.f};
            if (cc) w43_bt<5>(d, o);
            else w43_bt<6>(d, o);
            float4 *vp = reinterpret_cast<float4 *>(p.V) + (size_t)(cq0 + cql) * p.T + (size_t)n * p.TC + tl + (size_t)(fb + (int)a * nb) * vplane;
#pragma unroll
            for (int b = 0; b < 6; ++b)
                if (b < nb) vp[(size_t)b * vplane] = __builtin_bit_cast(float4, o[b]);
        }
    }
    // ---- P3: y to memory, pixel order (the G planes of this workgroup are one contiguous run) ----
    if (q.from_m && p.y) {
        float4 *yp = reinterpret_cast<float4 *>(p.y) + ((size_t)n * p.Cq + cq0) * HW;
        for (unsigned i = tid; i < (unsigned)p.G * HW; i += bd) {
            unsigned cql, px, h, w;
            q.divHW.divmod(i, cql, px);
            q.divW.divmod(px, h, w);
            yp[i] = plane[(size_t)cql * q.pcells + (h + 1) * (unsigned)pitch + w + 1];
        }
    }
}
