// How do vector-ALU / LDS instructions share a SIMD with a stream of fp32 MFMAs (v_mfma_f32_16x16x4_f32, 8 passes)?
//  mode 0: every wave issues MFMAs back to back, K independent v_fma_f32 (or ds_read_b32 with LDS=1) after each MFMA
//  mode 1: wave 0 of each SIMD runs ONLY the vector instructions (the same count), the others only MFMAs
// Reports cycles per MFMA per SIMD (32 = the pipe's own rate) for WPS waves per SIMD and K = 0..8.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_valu tools/ubench/mfma_valu.hip && tools/ubench/bin/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int LDS, int MODE, int BIG>
__global__ void __launch_bounds__(1024) probe(float *out, int iters, long long *cyc) {
    __shared__ float sm[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) sm[i] = i * 0.001f;
    __syncthreads();
    // BIG: v_mfma_f32_32x32x2_f32 (16 passes, 4 accumulators of 16 registers) instead of 16x16x4 (8 passes, 8 of 4)
    f32x4 acc[8];
    f32x16 accb[4];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane * 0.01f + i;
    const float a = lane * 0.5f, b = 1.f + lane;
    const bool valu_only = MODE == 1 && wave < 4, mfma_only = MODE == 1 && wave >= 4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (!valu_only) {
                if (BIG) accb[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accb[m & 3], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            }
            if (!mfma_only) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (LDS) v[k] += sm[(lane + k * 64 + m * 512 + it) & 4095];
                    else v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + v[i + 8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += accb[i][r];
    if (s == 1234.5f) out[tid] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int K, int LDS, int MODE, int BIG = 0>
void run(int wps, float *out, long long *cyc) {
    const int iters = 2000, waves = wps * 4;
    probe<K, LDS, MODE, BIG><<<256, waves * 64>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<K, LDS, MODE, BIG><<<256, waves * 64>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    // MFMAs per SIMD: mode 0 every wave issues 8 per iteration, mode 1 the wps - 1 MFMA waves
    const int mw = MODE == 1 ? wps - 1 : wps;
    const double per = mw ? (double)h[MODE == 1 ? 4 : 0] / ((double)iters * 8 * mw) : 0;
    printf("%s mode %d %s K=%d waves/SIMD=%d: %.1f counter ticks per MFMA per SIMD (wave 0 ran %lld ticks, wave %d %lld), %.1f us\n", BIG ? "32x32x2" : "16x16x4", MODE,
           LDS ? "ds_read" : "v_fma  ", K, wps, per, h[0], waves - 1, h[waves - 1], ms * 1e3);
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 256 * 16 * 8);
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 0, 0>(wps, out, cyc); run<1, 0, 0>(wps, out, cyc); run<2, 0, 0>(wps, out, cyc); run<4, 0, 0>(wps, out, cyc);
        run<7, 0, 0>(wps, out, cyc); run<8, 0, 0>(wps, out, cyc);
        run<1, 1, 0>(wps, out, cyc); run<2, 1, 0>(wps, out, cyc); run<4, 1, 0>(wps, out, cyc);
    }
    for (int wps = 2; wps <= 3; ++wps) {
        run<1, 0, 1>(wps, out, cyc); run<2, 0, 1>(wps, out, cyc); run<4, 0, 1>(wps, out, cyc); run<8, 0, 1>(wps, out, cyc);
        run<2, 1, 1>(wps, out, cyc); run<4, 1, 1>(wps, out, cyc);
    }
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 0, 0, 1>(wps, out, cyc); run<1, 0, 0, 1>(wps, out, cyc); run<2, 0, 0, 1>(wps, out, cyc); run<4, 0, 0, 1>(wps, out, cyc);
        run<8, 0, 0, 1>(wps, out, cyc); run<14, 0, 0, 1>(wps, out, cyc);
    }
    for (int wps = 2; wps <= 3; ++wps) {
        run<1, 0, 1, 1>(wps, out, cyc); run<2, 0, 1, 1>(wps, out, cyc); run<4, 0, 1, 1>(wps, out, cyc); run<8, 0, 1, 1>(wps, out, cyc);
        run<16, 0, 1, 1>(wps, out, cyc);
    }
    return 0;
}
