// fp32 vector-ALU issue rates on gfx950: v_fma_f32 against v_pk_fma_f32 (plain and with op_sel operand splats), v_pk_mul / v_pk_add,
// 16 independent accumulators per lane, 1 / 2 / 4 / 8 waves per SIMD.  Prints cycles per wave-instruction per SIMD and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o bin/valu_rate valu_rate.hip && bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) rate(float *out, int iters, float a, float b) {
    float acc[16];
    v2 pacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = (float)(threadIdx.x + i); pacc[i] = (v2){(float)i, (float)threadIdx.x}; }
    float x = a + threadIdx.x, w = b;
    v2 px = (v2){x, x + 1.f}, pw = (v2){w, w + 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(w));
                if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(px), "v"(pw));
                if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(pacc[i]) : "v"(px), "v"(pw));
                if constexpr (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(pacc[i]) : "v"(px));
                if constexpr (KIND == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(pacc[i]) : "v"(px));
                if constexpr (KIND == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
                if constexpr (KIND == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "s"(b));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + pacc[i].x + pacc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel splat", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32", "v_fma_f32 (sgpr operand)"};
    const double flops_per_lane[] = {2, 4, 4, 2, 2, 1, 2};
    void (*kern[])(float *, int, float, float) = {rate<0>, rate<1>, rate<2>, rate<3>, rate<4>, rate<5>, rate<6>};
    const int iters = 4000;
    for (int k = 0; k < 7; ++k)
        for (int wps : {1, 2, 4, 8}) {                       // 256-thread blocks: one wave per SIMD each
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kern[k], dim3(cus * wps), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double instr_per_simd = (double)iters * 64 * wps;
            const double tf = flops_per_lane[k] * 64 * instr_per_simd * cus * 4 / (best * 1e-3) / 1e12;
            printf("%-28s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n", names[k], wps,
                   best, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4, tf);
        }
    return 0;
}
