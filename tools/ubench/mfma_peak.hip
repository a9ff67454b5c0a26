// fp32 MFMA peak probe: waves/SIMD x accumulators x instruction shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ void __launch_bounds__(256) probe(float *out, int iters, long long *cyc) {
    float a = threadIdx.x * 0.001f, b = 1.0f - threadIdx.x * 0.002f;
    long long t0 = clock64();
    if (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 123.456f) out[threadIdx.x] = s;
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        if (s == 123.456f) out[threadIdx.x] = s;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int SHAPE>
void run(int blocks_per_cu, int iters) {
    float *out; long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    probe<NACC, SHAPE><<<grid, 256>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC, SHAPE><<<grid, 256>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double flop_per = SHAPE == 32 ? 4096.0 : 2048.0;
    double flops = (double)grid * 4 * iters * 8 * NACC * flop_per;
    printf("shape %2d  acc %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s   clock64 ticks %lld (%.0f MHz equiv)\n", SHAPE, NACC, blocks_per_cu, ms,
           flops / ms / 1e9, c, c / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}

int main() {
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clockRate attr %d kHz\n", clk);
    for (int w : {1, 2, 4, 8}) {
        run<1, 32>(w, 4000); run<2, 32>(w, 2000); run<4, 32>(w, 1000);
        run<1, 16>(w, 8000); run<2, 16>(w, 4000); run<4, 16>(w, 2000);
    }
    // long run to see sustained clocks
    run<4, 32>(2, 20000);
    run<4, 32>(2, 20000);
    return 0;
}
