// What does a partner wave's work cost an MFMA-only wave on the same SIMD?
// 512-thread workgroups, one per CU: waves 0-3 issue 24 v_mfma_f32_32x32x2_f32 per step (6 independent
// accumulators) and meet waves 4-7 at one s_barrier per step; waves 4-7 do, per step, K instructions of
// one kind.  Output: ns per step for each (kind, K); the MFMA-bound step is 24*64 cycles.
//   hipcc --offload-arch=gfx950 -O3 -o bin/pc_interference pc_interference.hip && bin/pc_interference
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void keep(float4 (&v)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
}

template <int KIND, int K>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
kern(float *out, const float *in, int steps, int in_bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(1);
        f32x16 acc[6];
        for (int f = 0; f < 6; ++f)
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
        float a = in[tid], b = in[tid + 512];
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 6; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
            asm volatile("s_barrier" ::: "memory");
        }
        float t = 0.f;
        for (int f = 0; f < 6; ++f)
            for (int r = 0; r < 16; ++r) t += acc[f][r];
        out[blockIdx.x * 512 + tid] = t;
        return;
    }
    // partner waves
    float4 v[6];
    for (int i = 0; i < 6; ++i) v[i] = make_float4(in[tid + i], in[tid + 7 + i], in[tid + 13 + i], in[tid + 17 + i]);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, in_bytes, 0x00020000);
    float4 *lds4 = reinterpret_cast<float4 *>(smem) + (wave - 4) * 64 * 8 + lane;
    float4 pend[K > 0 ? K : 1];
    for (int i = 0; i < (K > 0 ? K : 1); ++i) pend[i] = v[i % 6];
    for (int s = 0; s < steps; ++s) {
        if constexpr (KIND == 8) {                 // K buffer loads, consumed ONE STEP LATER
#pragma unroll
            for (int i = 0; i < K; ++i) v[i % 6].x += pend[i].x;
#pragma unroll
            for (int i = 0; i < K; ++i)
                pend[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid * 16 + i * 8192 + (s & 7) * 65536) & (in_bytes - 1), 0, 0));
        } else if constexpr (KIND == 9) {          // K LDS-DMA pieces, waited for one step later
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < K; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) float *)(smem + ((wave - 4) * 8 + (i % 8)) * 256),
                                                         16, (tid * 16 + i * 8192 + (s & 7) * 65536) & (in_bytes - 1), 0, 0, 0);
        } else if constexpr (KIND == 10) {         // K OUT-OF-RANGE buffer loads (no memory traffic), consumed next step
#pragma unroll
            for (int i = 0; i < K; ++i) v[i % 6].x += pend[i].x;
#pragma unroll
            for (int i = 0; i < K; ++i)
                pend[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)0x80000000, 0, 0));
        } else if constexpr (KIND == 1) {                 // K dependent-free v_fma_f32 (4 chains x float4 lanes)
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float *p = reinterpret_cast<float *>(&v[i % 6]);
                p[i % 4] = __builtin_fmaf(p[i % 4], 1.0001f, 0.5f);
            }
            keep(v);
        } else if constexpr (KIND == 2) {          // K ds_write_b128
#pragma unroll
            for (int i = 0; i < K; ++i) lds4[(i % 8) * 64] = v[i % 6];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 3) {          // K buffer_load_dwordx4 into registers (L2 hits), consumed next step
            float4 t[K > 0 ? K : 1];
#pragma unroll
            for (int i = 0; i < K; ++i)
                t[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid * 16 + i * 8192 + (s & 7) * 65536) & (in_bytes - 1), 0, 0));
#pragma unroll
            for (int i = 0; i < K; ++i) v[i % 6].x += t[i].x;
        } else if constexpr (KIND == 4) {          // K LDS-DMA pieces (buffer_load_dwordx4 ... lds)
#pragma unroll
            for (int i = 0; i < K; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) float *)(smem + ((wave - 4) * 8 + (i % 8)) * 256),
                                                         16, (tid * 16 + i * 8192 + (s & 7) * 65536) & (in_bytes - 1), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 5) {          // K ds_read_b128
            float4 t[K > 0 ? K : 1];
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = lds4[(i % 8) * 64];
#pragma unroll
            for (int i = 0; i < K; ++i) v[i % 6].x += t[i].x;
        } else if constexpr (KIND == 6) {          // K v_pk_fma_f32-able float2 ops
#pragma unroll
            for (int i = 0; i < K; ++i) {
                v[i % 6].x = __builtin_fmaf(v[i % 6].x, 1.0001f, 0.5f);
                v[i % 6].y = __builtin_fmaf(v[i % 6].y, 1.0001f, 0.5f);
            }
            keep(v);
        } else if constexpr (KIND == 7) {          // K integer VALU (address arithmetic)
            int *q = reinterpret_cast<int *>(&v[0]);
#pragma unroll
            for (int i = 0; i < K; ++i) q[i % 24] = q[i % 24] * 3 + i;
            keep(v);
        }
        asm volatile("s_barrier" ::: "memory");
    }
    float t = 0.f;
    for (int i = 0; i < 6; ++i) t += v[i].x + v[i].y + v[i].z + v[i].w;
    out[blockIdx.x * 512 + tid] = t;
}

template <int KIND, int K>
void run(const char *name, float *out, float *in, int in_bytes, int blocks) {
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)kern<KIND, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    kern<KIND, K><<<blocks, 512, 65536>>>(out, in, steps, in_bytes);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        kern<KIND, K><<<blocks, 512, 65536>>>(out, in, steps, in_bytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-28s K=%3d  %7.1f ns/step  (%.0f TFLOP/s MFMA at %d blocks)\n", name, K, best * 1e6 / steps,
           blocks * 4.0 * 24 * 4096 / (best * 1e-3 / steps) / 1e12, blocks);
}

int main() {
    const int in_bytes = 1 << 22;
    float *in, *out;
    hipMalloc(&in, in_bytes); hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(in_bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 9) & 1023) / 1024.f - 0.5f;
    hipMemcpy(in, h.data(), in_bytes, hipMemcpyHostToDevice);
    for (int blocks : {256, 64}) {
        run<0, 0>("mfma only", out, in, in_bytes, blocks);
        run<0, 0>("mfma only", out, in, in_bytes, blocks);
        run<8, 6>("buffer_load, used next step", out, in, in_bytes, blocks);
        run<8, 12>("buffer_load, used next step", out, in, in_bytes, blocks);
        run<10, 6>("OOB buffer_load, next step", out, in, in_bytes, blocks);
        run<10, 12>("OOB buffer_load, next step", out, in, in_bytes, blocks);
        run<9, 6>("lds-dma, waited next step", out, in, in_bytes, blocks);
        run<9, 12>("lds-dma, waited next step", out, in, in_bytes, blocks);
        run<0, 0>("mfma only", out, in, in_bytes, blocks);
        run<1, 24>("v_fma_f32", out, in, in_bytes, blocks);
        run<1, 96>("v_fma_f32", out, in, in_bytes, blocks);
        run<1, 192>("v_fma_f32", out, in, in_bytes, blocks);
        run<6, 96>("2x v_fma (pk-able)", out, in, in_bytes, blocks);
        run<7, 96>("int mul-add", out, in, in_bytes, blocks);
        run<2, 6>("ds_write_b128", out, in, in_bytes, blocks);
        run<2, 12>("ds_write_b128", out, in, in_bytes, blocks);
        run<2, 24>("ds_write_b128", out, in, in_bytes, blocks);
        run<5, 12>("ds_read_b128", out, in, in_bytes, blocks);
        run<5, 24>("ds_read_b128", out, in, in_bytes, blocks);
        run<3, 6>("buffer_load_dwordx4 -> vgpr", out, in, in_bytes, blocks);
        run<3, 12>("buffer_load_dwordx4 -> vgpr", out, in, in_bytes, blocks);
        run<4, 3>("buffer_load_dwordx4 lds", out, in, in_bytes, blocks);
        run<4, 6>("buffer_load_dwordx4 lds", out, in, in_bytes, blocks);
        run<4, 12>("buffer_load_dwordx4 lds", out, in, in_bytes, blocks);
    }
    return 0;
}
