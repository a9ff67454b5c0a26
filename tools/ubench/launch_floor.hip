// What does one more dependent kernel cost?  Chains of N small kernels, in a stream and as a captured hipGraph:
//   empty (1 WG), touch (W WGs each read-modify-write 1 KB -- a real dependence on the previous kernel),
//   and touch + agent-scope release/acquire fence with a per-"tile" arrival counter (the split-K "last arriver
//   reduces" pattern) to price the fence against a second kernel launch.
//   hipcc --offload-arch=gfx950 -O3 -o bin/launch_floor launch_floor.hip && bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float *p) { if (p == nullptr) p[0] = 1.f; }
__global__ void __launch_bounds__(256) k_touch(float *p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    p[i] = p[i] * 1.0001f + 1.f;
}
// S splits per tile: every WG writes a 16 KB slab, the last arriver of the tile sums the S slabs
__global__ void __launch_bounds__(256) k_arrive(float *slab, float *out, int *cnt, int S) {
    const int tile = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
    float4 *mine = reinterpret_cast<float4 *>(slab) + ((size_t)z * gridDim.x + tile) * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[r * 256 + tid] = make_float4(z, tile, r, tid);
    __threadfence();
    __syncthreads();
    __shared__ int last;
    if (tid == 0) last = atomicAdd(&cnt[tile], 1) == S - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    float4 acc[4] = {};
    for (int s = 0; s < S; ++s) {
        const float4 *sp = reinterpret_cast<const float4 *>(slab) + ((size_t)s * gridDim.x + tile) * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 v = sp[r * 256 + tid];
            acc[r].x += v.x; acc[r].y += v.y; acc[r].z += v.z; acc[r].w += v.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) reinterpret_cast<float4 *>(out)[(size_t)tile * 1024 + r * 256 + tid] = acc[r];
    if (tid == 0) cnt[tile] = 0;
}
// the same with agent-scope (sc1) write-through stores / loads instead of L2 write-back + invalidate fences
__global__ void __launch_bounds__(256) k_arrive_sc1(float *slab, float *out, int *cnt, int S, int bytes) {
    const int tile = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, bytes, 0x00020000);
    typedef unsigned u4 __attribute__((__vector_size__(16)));
    const int mine = (int)((((size_t)z * gridDim.x + tile) * 1024 + tid) * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, make_float4(z, tile, r, tid)), rs, mine + r * 4096, 0, 16);
    __builtin_amdgcn_s_waitcnt(0);                      // every write-through store acknowledged
    __syncthreads();
    __shared__ int last;
    if (tid == 0) last = __hip_atomic_fetch_add(&cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
    __syncthreads();
    if (!last) return;
    float4 acc[4] = {};
    for (int s = 0; s < S; ++s) {
        const int off = (int)((((size_t)s * gridDim.x + tile) * 1024 + tid) * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + r * 4096, 0, 16));
            acc[r].x += v.x; acc[r].y += v.y; acc[r].z += v.z; acc[r].w += v.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) reinterpret_cast<float4 *>(out)[(size_t)tile * 1024 + r * 256 + tid] = acc[r];
    if (tid == 0) __hip_atomic_store(&cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(256) k_slab_only(float *slab) {
    const int tile = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
    float4 *mine = reinterpret_cast<float4 *>(slab) + ((size_t)z * gridDim.x + tile) * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[r * 256 + tid] = make_float4(z, tile, r, tid);
}
__global__ void __launch_bounds__(256) k_reduce(const float *slab, float *out, int S) {
    const int tile = blockIdx.x, tid = threadIdx.x;
    float4 acc[4] = {};
    for (int s = 0; s < S; ++s) {
        const float4 *sp = reinterpret_cast<const float4 *>(slab) + ((size_t)s * gridDim.x + tile) * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 v = sp[r * 256 + tid];
            acc[r].x += v.x; acc[r].y += v.y; acc[r].z += v.z; acc[r].w += v.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) reinterpret_cast<float4 *>(out)[(size_t)tile * 1024 + r * 256 + tid] = acc[r];
}

template <class F> double chain(hipStream_t st, int n, bool graph, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    if (graph) {
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch();
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    } else { for (int i = 0; i < n; ++i) launch(); hipStreamSynchronize(st); }
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        if (graph) hipGraphLaunch(ge, st); else for (int i = 0; i < n; ++i) launch();
        hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    return best / n;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *p, *slab, *out; int *cnt;
    CK(hipMalloc(&p, 64 << 20)); CK(hipMalloc(&slab, 256 << 20)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&cnt, 1 << 20));
    CK(hipMemset(cnt, 0, 1 << 20)); CK(hipMemset(p, 0, 64 << 20));
    const int n = 400;
    for (int graph = 0; graph < 2; ++graph) {
        printf("%s: empty %.2f us/kernel", graph ? "graph " : "stream", chain(st, n, graph, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, p); }));
        for (int w : {1, 64, 256, 1024, 4096})
            printf("  touch[%d WG] %.2f", w, chain(st, n, graph, [&] { hipLaunchKernelGGL(k_touch, dim3(w), dim3(256), 0, st, p); }));
        printf("\n");
    }
    for (int tiles : {24, 48, 96, 192})
        for (int S : {2, 4, 8, 16}) {
            double fused = chain(st, 200, true, [&] { hipLaunchKernelGGL(k_arrive, dim3(tiles, S), dim3(256), 0, st, slab, out, cnt, S); });
            double two = chain(st, 200, true, [&] {
                hipLaunchKernelGGL(k_slab_only, dim3(tiles, S), dim3(256), 0, st, slab);
                hipLaunchKernelGGL(k_reduce, dim3(tiles), dim3(256), 0, st, slab, out, S);
            });
            double sc1 = chain(st, 200, true, [&] { hipLaunchKernelGGL(k_arrive_sc1, dim3(tiles, S), dim3(256), 0, st, slab, out, cnt, S, 256 << 20); });
            printf("tiles %3d splits %2d: slab+fence+last-arriver %.2f us   sc1 last-arriver %.2f us   slab kernel + reduce kernel %.2f us\n", tiles, S, fused, sc1, two);
        }
    return 0;
}
