// What caps the fp32 MFMA pipe?  Same MFMA stream, ingredients added one by one.
//  V0: MFMAs on register operands          V1: + ds_read_b128 fragments per chunk
//  V2: + ds_write_b128 + s_barrier         V3: + 5 buffer loads per chunk (L2-resident)
//  V4: V3 with the real kernel's LDS stores (4 x b128 per thread per chunk, data from the loads)
//  V5: V4 with the real kernel's global loads (2 x b128 filter rows + 8 gathered dwords with a
//      per-chunk tap/bounds computation, OOB-select, scalar plane offsets), 2 chunks ahead
//  V6: V5 + the epilogue (64 dword stores per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int TM, int TN>
__global__ void __launch_bounds__(256) mix(const float *g, float *out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) smem[i] = (float)(i % 97) * 0.01f;
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *fa = V >= 13 ? smem + (lane >> 5) * 512 + (lane & 31) * 4 : smem + (lane & 31) * 20 + (lane >> 5) * 4;
    const float *fb = V >= 13 ? smem + 4096 + (lane >> 5) * 512 + (lane & 31) * 4 : smem + 4096 + (lane & 31) * 20 + (lane >> 5) * 4;
    float4 st = make_float4(1.f, 2.f, 3.f, 4.f);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, V == 14 ? (1 << 28) + (1 << 24) : 1 << 26, 0x00020000);
    float4 ld0 = st, ld1 = st;
    constexpr bool RS = V == 5 || V == 6 || V == 8;   // real LDS tile layout
    float4 lb0[2] = {st, st}, lb1[2] = {st, st}, la0[2] = {st, st}, la1[2] = {st, st};
    for (int c = 0; c < chunks; ++c) {
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[u][a] = V >= 1 ? *reinterpret_cast<const float4 *>(fa + (V >= 13 ? a * 128 + u * 1024 : a * 640 + u * 8) + (c & 1) * (RS ? 2560 : 2048))
                                  : make_float4(c + a, u, lane, 1.f);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[u][b] = V >= 1 ? *reinterpret_cast<const float4 *>(fb + (RS ? 1024 : 0) + (V >= 13 ? b * 128 + u * 1024 : b * 640 + u * 8) + (c & 1) * (RS ? 2560 : 2048))
                                  : make_float4(c - b, u, lane, 2.f);
        }
        if (V == 13 || V == 14 || V == 16) {
            // x: [32][32 quads][32][32][4]; w: [K/4][128][4]; one b128 per (pixel, channel quad)
            const int tap = (c >> 3) % 9, cq0 = (c & 7) * 4;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int pix = blockIdx.x * 128 + (tid & 127);
            const int n = pix >> 10, h = (pix >> 5) & 31, w = pix & 31;
            const bool ok = (unsigned)(h + dy) < 32u && (unsigned)(w + dx) < 32u;
            const int voff = ok ? (((V == 14 ? n : n & 31) * 32 * 1024 + (h + dy) * 32 + (w + dx)) << 4) : (int)0x80000000;
            lb1[0] = lb0[0]; lb1[1] = lb0[1]; la1[0] = la0[0]; la1[1] = la0[1];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int soff = ((cq0 + (tid >> 7) + ps * 2) * 1024) << 4;
                // V16: 16-byte loads at 4-byte alignment, 24 bytes between lanes (row-packed 3-channel stem)
                const int vo = V == 16 ? (ok ? (((n & 31) * 32 * 1024 + (h + dy) * 32) << 4) + (w + dx) * 24 + 4 : (int)0x80000000) : voff;
                lb0[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int aoff = (V == 14 ? (1 << 28) : (1 << 25)) + ((((tid >> 7) + i * 2) * 128 + (tid & 127)) << 4);
                la0[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, aoff, (c % 72) * 4 * 128 * 16, 0));
            }
        } else if (V == 5 || V == 6 || V == 7 || V >= 9) {
            // 3x3 taps (V9: gathers as 2 x b128; V10: no tap math; V11: B loads only; V12: A loads only) over a [32][128][32][32] tensor, 8 chunks of 16 channels per tap
            const int tap = (c >> 3) % 9, cin0 = (c & 7) * 16;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int pix = blockIdx.x * 128 + (tid & 127);
            const int n = pix >> 10, h = (pix >> 5) & 31, w = pix & 31;
            const bool ok = (unsigned)(h + dy) < 32u && (unsigned)(w + dx) < 32u;
            const int voff = V == 10 ? ((n & 31) * 128 * 1024 + h * 32 + w) << 2
                           : ok ? (((n & 31) * 128 * 1024 + (h + dy) * 32 + (w + dx)) << 2) : (int)0x80000000;
            lb1[0] = lb0[0]; lb1[1] = lb0[1]; la1[0] = la0[0]; la1[1] = la0[1];
#pragma unroll
            for (int ps = 0; ps < 2 && V != 12; ++ps) {
                if (V == 9) {
                    const int soff = ((cin0 + ps * 8) * 1024) << 2;
                    lb0[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (voff & ~15) + (tid >> 7) * 4096, soff, 0));
                    continue;
                }
                float tv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int soff = ((cin0 + ((tid >> 7) + ps * 2) * 4 + e) * 1024) << 2;
                    tv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
                }
                lb0[ps] = make_float4(tv[0], tv[1], tv[2], tv[3]);
            }
#pragma unroll
            for (int i = 0; i < 2 && V != 11; ++i) {
                const int v = tid + i * 256;
                const int aoff = (1 << 25) + (((v >> 2) * 1152 + (v & 3) * 4) << 2);
                la0[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, aoff, (c % 72) * 64, 0));
            }
        } else if (V >= 3) {
            int off = ((blockIdx.x * 256 + tid) * 16 + (c & 63) * 65536) & ((1 << 24) - 1);
            ld1 = ld0;
            ld0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4096 * (e + 1), 0, 0));
                ld0.x += t * 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float av = s4 == 0 ? af[u][a].x : s4 == 1 ? af[u][a].y : s4 == 2 ? af[u][a].z : af[u][a].w;
                        const float bv = s4 == 0 ? bf[u][b].x : s4 == 1 ? bf[u][b].y : s4 == 2 ? bf[u][b].z : bf[u][b].w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                    }
        if (V == 13 || V == 14) {
            float *Ab = smem + ((c + 1) & 1) * 2048, *Bb = smem + 4096 + ((c + 1) & 1) * 2048;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                *reinterpret_cast<float4 *>(Bb + ((tid >> 7) + ps * 2) * 512 + (tid & 127) * 4) = lb1[ps];
                *reinterpret_cast<float4 *>(Ab + ((tid >> 7) + ps * 2) * 512 + (tid & 127) * 4) = la1[ps];
            }
            __syncthreads();
        } else if (V == 5 || V == 6 || V == 8) {
            if (V == 8) { lb1[0] = lb1[1] = la1[0] = la1[1] = ld1; }
            // the real [row][k] tiles: row stride 20 floats, 128 rows x 16 k per operand
            float *Ab = smem + ((c + 1) & 1) * 2560, *Bb = smem + 5120 + ((c + 1) & 1) * 2560;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
                *reinterpret_cast<float4 *>(Bb + (tid & 127) * 20 + ((tid >> 7) + ps * 2) * 4) = lb1[ps];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int v = tid + i * 256;
                *reinterpret_cast<float4 *>(Ab + (v >> 2) * 20 + (v & 3) * 4) = la1[i];
            }
            __syncthreads();
        } else if (V == 7 || V >= 9) {
            *reinterpret_cast<float4 *>(smem + ((c + 1) & 1) * 2048 + tid * 4) = lb1[0];
            *reinterpret_cast<float4 *>(smem + ((c + 1) & 1) * 2048 + 1024 + tid * 4) = lb1[1];
            *reinterpret_cast<float4 *>(smem + 4096 + ((c + 1) & 1) * 2048 + tid * 4) = la1[0];
            *reinterpret_cast<float4 *>(smem + 4096 + ((c + 1) & 1) * 2048 + 1024 + tid * 4) = la1[1];
            __syncthreads();
        } else if (V >= 4) {
            *reinterpret_cast<float4 *>(smem + ((c + 1) & 1) * 2048 + tid * 4) = ld1;
            *reinterpret_cast<float4 *>(smem + ((c + 1) & 1) * 2048 + 1024 + tid * 4) = ld1;
            *reinterpret_cast<float4 *>(smem + 4096 + ((c + 1) & 1) * 2048 + tid * 4) = ld1;
            *reinterpret_cast<float4 *>(smem + 4096 + ((c + 1) & 1) * 2048 + 1024 + tid * 4) = ld1;
            __syncthreads();
        } else if (V >= 2) {
            *reinterpret_cast<float4 *>(smem + ((c + 1) & 1) * 2048 + tid * 4) = V >= 3 ? ld1 : st;
            *reinterpret_cast<float4 *>(smem + 4096 + ((c + 1) & 1) * 2048 + tid * 4) = st;
            __syncthreads();
        }
    }
    if (V == 6) {
        const int wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
        float *o = out + 4096 + (size_t)(blockIdx.x & 2047) * 128;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + a * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                    o[(size_t)row * 262144 + wn * 64 + b * 32 + (lane & 31)] = acc[a][b][r];
                }
        return;
    }
    float s = 0;
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 1234.5f) out[tid] = s;
}

// V15: the Q4 pattern (V13) with direct global->LDS loads (buffer_load_dwordx4 ... lds, gfx950): no
// VGPR staging, no ds_write; three LDS buffers so a chunk's loads are in flight for two chunk times.
__global__ void __launch_bounds__(256) mix_dma(const float *g, float *out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [3][A 2048 | B 2048] floats
    typedef __attribute__((address_space(3))) void lds_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, (1 << 28) + (1 << 24), 0x00020000);
    const int pix = blockIdx.x * 128 + (tid & 127);
    const int n = pix >> 10, h = (pix >> 5) & 31, w = pix & 31;
    auto issue = [&](int c) {
        float *buf = smem + (c % 3) * 4096;
        const int tap = (c >> 3) % 9, cq0 = (c & 7) * 4;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const bool ok = (unsigned)(h + dy) < 32u && (unsigned)(w + dx) < 32u;
        const int voff = ok ? ((n * 32 * 1024 + (h + dy) * 32 + (w + dx)) << 4) : (int)0x80000000;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int kq = (wave >> 1) + ps * 2;                                   // wave-uniform k-quad
            const int soff = ((cq0 + kq) * 1024) << 4;
            // B: wave writes 64 consecutive columns of plane kq;  A: 64 consecutive rows of plane kq
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t *)(buf + 2048 + (kq * 128 + (wave & 1) * 64) * 4), 16, voff, soff, 0, 0);
            const int aoff = (1 << 28) + ((kq * 128 + (tid & 127)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t *)(buf + (kq * 128 + (wave & 1) * 64) * 4), 16, aoff, (c % 72) * 4 * 128 * 16, 0, 0);
        }
    };
    // (hipcc waits vmcnt(0) before any LDS read that follows a DMA load, so the next chunk's loads
    //  are issued AFTER this chunk's fragment reads: one chunk time of latency cover, two buffers)
    issue(0);
    __syncthreads();
    const float *fa = smem + (lane >> 5) * 512 + (lane & 31) * 4;
    for (int c = 0; c < chunks; ++c) {
        const float *A = fa + (c % 3) * 4096, *B = fa + (c % 3) * 4096 + 2048;
        float4 af[2][2], bf[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[u][t] = *reinterpret_cast<const float4 *>(A + ((wave >> 1) * 64 + t * 32) * 4 + u * 1024);
                bf[u][t] = *reinterpret_cast<const float4 *>(B + ((wave & 1) * 64 + t * 32) * 4 + u * 1024);
            }
        __builtin_amdgcn_sched_barrier(0);
        issue(c + 1);
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
        __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 1234.5f) out[tid] = s;
}

void run_dma(const float *g, float *out, int per_cu, int chunks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)mix_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    mix_dma<<<256 * per_cu, 256, 49152>>>(g, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) mix_dma<<<256 * per_cu, 256, 49152>>>(g, out, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = 256.0 * per_cu * 4 * chunks * 8 * 4 * 4096.0;
    printf("V15 tile 128x128  %d/CU: %.3f ms  %.1f TFLOP/s   (LDS-DMA)\n", per_cu, ms, flops / ms / 1e9);
}

template <int V, int TM, int TN>
void run(const float *g, float *out, int per_cu, int chunks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mix<V, TM, TN><<<256 * per_cu, 256, 40960>>>(g, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) mix<V, TM, TN><<<256 * per_cu, 256, 40960>>>(g, out, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = 256.0 * per_cu * 4 * chunks * 8 * TM * TN * 4096.0;
    printf("V%-2d tile %dx%d  %d/CU: %.3f ms  %.1f TFLOP/s\n", V, TM * 32 * 2, TN * 32 * 2, per_cu, ms, flops / ms / 1e9);
}

int main() {
    float *g, *out; (void)hipMalloc(&g, (1 << 28) + (1 << 24)); (void)hipMalloc(&out, (size_t)4096 * 4 + (size_t)128 * 262144 * 4); const size_t gbytes = ((size_t)1 << 28) + (1 << 24);
    (void)hipMemset(g, 0, gbytes);
    if (getenv("RANDOM_DATA")) {      // MFMA power (and with it the sustained clock) depends on the operand bits
        std::vector<float> h(gbytes / 4);
        unsigned s = 12345u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        (void)hipMemcpy(g, h.data(), gbytes, hipMemcpyHostToDevice);
        printf("operands: random in [-1, 1)\n");
    } else printf("operands: zeros\n");
    // short K loops (72 / 36 / 16 chunks per workgroup, many workgroups): what block turnover costs
    for (int per : {1, 2, 3, 8}) {
        printf("-- 128x128 tiles, 72 chunks, %d workgroups per CU\n", per);
        run<2, 2, 2>(g, out, per, 72); run<3, 2, 2>(g, out, per, 72); run<4, 2, 2>(g, out, per, 72);
        run<5, 2, 2>(g, out, per, 72); run<6, 2, 2>(g, out, per, 72);
        run<7, 2, 2>(g, out, per, 72); run<8, 2, 2>(g, out, per, 72);
        run<9, 2, 2>(g, out, per, 72); run<10, 2, 2>(g, out, per, 72);
        run<11, 2, 2>(g, out, per, 72); run<12, 2, 2>(g, out, per, 72); run<13, 2, 2>(g, out, per, 72); run<14, 2, 2>(g, out, per, 72); run<16, 2, 2>(g, out, per, 72); run_dma(g, out, per, 72);
    }
    return 0;
    for (int ch : {72, 36, 16}) {
        printf("-- %d chunks per workgroup, 8 workgroups per CU\n", ch);
        run<2, 1, 1>(g, out, 8, ch); run<3, 1, 1>(g, out, 8, ch);
        run<2, 2, 2>(g, out, 8, ch); run<3, 2, 2>(g, out, 8, ch);
    }
    for (int per : {1, 2, 4}) {
        run<0, 1, 1>(g, out, per, 2000 / per); run<1, 1, 1>(g, out, per, 2000 / per);
        run<2, 1, 1>(g, out, per, 2000 / per); run<3, 1, 1>(g, out, per, 2000 / per);
        run<0, 2, 2>(g, out, per, 600 / per); run<1, 2, 2>(g, out, per, 600 / per);
        run<2, 2, 2>(g, out, per, 600 / per); run<3, 2, 2>(g, out, per, 600 / per);
    }
    return 0;
}
