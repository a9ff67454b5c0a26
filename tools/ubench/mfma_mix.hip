// What caps the fp32 MFMA pipe?  Same MFMA stream, ingredients added one by one.
//  V0: MFMAs on register operands          V1: + ds_read_b128 fragments per chunk
//  V2: + ds_write_b128 + s_barrier         V3: + 5 buffer loads per chunk (L2-resident)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int TM, int TN>
__global__ void __launch_bounds__(256) mix(const float *g, float *out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) smem[i] = (float)(i % 97) * 0.01f;
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *fa = smem + (lane & 31) * 20 + (lane >> 5) * 4;
    const float *fb = smem + 4096 + (lane & 31) * 20 + (lane >> 5) * 4;
    float4 st = make_float4(1.f, 2.f, 3.f, 4.f);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, 1 << 26, 0x00020000);
    float4 ld0 = st, ld1 = st;
    for (int c = 0; c < chunks; ++c) {
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[u][a] = V >= 1 ? *reinterpret_cast<const float4 *>(fa + a * 640 + u * 8 + (c & 1) * 2048)
                                  : make_float4(c + a, u, lane, 1.f);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[u][b] = V >= 1 ? *reinterpret_cast<const float4 *>(fb + b * 640 + u * 8 + (c & 1) * 2048)
                                  : make_float4(c - b, u, lane, 2.f);
        }
        if (V >= 3) {
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
        if (V >= 2) {
            *reinterpret_cast<float4 *>(smem + ((c + 1) & 1) * 2048 + tid * 4) = V >= 3 ? ld1 : st;
            *reinterpret_cast<float4 *>(smem + 4096 + ((c + 1) & 1) * 2048 + tid * 4) = st;
            __syncthreads();
        }
    }
    float s = 0;
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 1234.5f) out[tid] = s;
}

template <int V, int TM, int TN>
void run(const float *g, float *out, int per_cu, int chunks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mix<V, TM, TN><<<256 * per_cu, 256, 32768>>>(g, out, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) mix<V, TM, TN><<<256 * per_cu, 256, 32768>>>(g, out, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = 256.0 * per_cu * 4 * chunks * 8 * TM * TN * 4096.0;
    printf("V%d tile %dx%d  %d/CU: %.3f ms  %.1f TFLOP/s\n", V, TM * 32 * 2, TN * 32 * 2, per_cu, ms, flops / ms / 1e9);
}

int main() {
    float *g, *out; (void)hipMalloc(&g, 1 << 26); (void)hipMalloc(&out, 4096); (void)hipMemset(g, 0, 1 << 26);
    for (int per : {1, 2, 4}) {
        run<0, 1, 1>(g, out, per, 2000 / per); run<1, 1, 1>(g, out, per, 2000 / per);
        run<2, 1, 1>(g, out, per, 2000 / per); run<3, 1, 1>(g, out, per, 2000 / per);
        run<0, 2, 2>(g, out, per, 600 / per); run<1, 2, 2>(g, out, per, 600 / per);
        run<2, 2, 2>(g, out, per, 600 / per); run<3, 2, 2>(g, out, per, 600 / per);
    }
    return 0;
}
