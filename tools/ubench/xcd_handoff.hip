// Does a kernel-to-kernel hand-off get faster when the consumer workgroup runs on the XCD whose L2 the producer workgroup
// just filled?  (the staged F(4x4,3x3) family hands M and V from kernel to kernel.)
//
// Layout = the Winograd-domain tensors': [F = 36 frequency planes][slabs][piece], a slab = what one consumer workgroup
// reads = 36 pieces of `piece` float4, one per plane (layer3 of ResNet-18 at batch 32: 512 chain workgroups x 36 x 64 float4
// = 18.4 MB; layer2: 1024 x 36 x 49..; layer4: 9.2 MB).  Producer block b writes slab (b + shift) % blocks; consumer block b
// reads slab b with every 16-byte load of a thread in flight at once.  Block b runs on XCD b % 8, so shift 0 and 8 are
// "same XCD", shift 1..7 "another XCD".  Timed: loops of {producer, consumer} and of {producer} alone, the difference is
// the consumer.  `--dirty MB`: a third kernel streams that many MB between the two (what the real GEMM's own operand
// traffic does to the L2).
//   hipcc --offload-arch=gfx950 -O3 -o bin/xcd_handoff xcd_handoff.hip && bin/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int F = 36;

// plane = float4 per frequency plane (blocks * piece + padding); contig: the slab's 36 pieces back to back instead
__device__ __forceinline__ size_t addr(int f, int slab, int e, int piece, size_t plane, int contig) {
    return contig ? ((size_t)slab * F + f) * piece + e : (size_t)f * plane + (size_t)slab * piece + e;
}
template <int TAG>       // TAG = 10 * case + shift index: separates the rows of a rocprofv3 --kernel-trace --stats summary
__global__ void __launch_bounds__(256) producer(float4 *M, int blocks, int piece, int shift, float seed, size_t plane, int contig) {
    const int slab = (int)((blockIdx.x + (unsigned)shift) % (unsigned)blocks);
    for (int i = threadIdx.x; i < F * piece; i += 256) {
        const int f = i / piece, e = i - f * piece;
        M[addr(f, slab, e, piece, plane, contig)] = make_float4(seed, (float)f, (float)slab, (float)e);
    }
}

template <int PER, int TAG>      // 16-byte loads per thread, all in flight
__global__ void __launch_bounds__(256) consumer(const float4 *M, float4 *out, int blocks, int piece, size_t plane, int contig) {
    const int slab = blockIdx.x;
    float4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * 256;
        const int f = i / piece, e = i - f * piece;
        v[k] = (i < F * piece) ? M[addr(f, slab, e, piece, plane, contig)] : make_float4(0, 0, 0, 0);
    }
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < PER; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    out[(size_t)slab * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) streamer(const float4 *src, float4 *dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = src[i];
        v.x += 1.f;
        dst[i] = v;
    }
}

int main(int argc, char **argv) {
    double dirty_mb = 0;
    int pad_bytes = 0, contig = 0, both_sides = 0, iters = 200;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--dirty") && i + 1 < argc) dirty_mb = atof(argv[++i]);
        if (!strcmp(argv[i], "--pad") && i + 1 < argc) pad_bytes = atoi(argv[++i]);       // bytes added to every frequency plane
        if (!strcmp(argv[i], "--contig")) contig = 1;                                      // slab-contiguous layout
        if (!strcmp(argv[i], "--both")) both_sides = 1;                                    // dirty traffic after the consumer too
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    }
    printf("plane padding %d bytes, %s layout\n", pad_bytes, contig ? "slab-contiguous" : "[frequency][slab][piece]");
    struct Case { const char *name; int blocks, piece; };
    // piece = G quads x tiles float4 per plane and workgroup; blocks = N x Cq / G
    const Case cases[] = {{"layer2 (G=1 x 49 tiles, 1024 WGs, 28.9 MB)", 1024, 49},
                          {"layer3 (G=4 x 16 tiles, 512 WGs, 18.9 MB)", 512, 64},
                          {"layer4 (G=8 x 4 tiles, 512 WGs, 9.4 MB)", 512, 32}};
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float4 *scratch_a = nullptr, *scratch_b = nullptr;
    const size_t dirty4 = (size_t)(dirty_mb * 1e6 / 16);
    if (dirty4) {
        CK(hipMalloc(&scratch_a, dirty4 * 16));
        CK(hipMalloc(&scratch_b, dirty4 * 16));
        CK(hipMemset(scratch_a, 0, dirty4 * 16));
    }
    printf("dirty traffic between producer and consumer: %.1f MB read + %.1f MB written\n", dirty_mb, dirty_mb);
    for (const Case &c : cases) {
        const size_t n4 = (size_t)F * c.blocks * c.piece;
        const size_t plane = (size_t)c.blocks * c.piece + pad_bytes / 16;
        float4 *M, *out;
        CK(hipMalloc(&M, (size_t)F * plane * 16));
        CK(hipMalloc(&out, (size_t)c.blocks * 256 * 16));
        const int per = (F * c.piece + 255) / 256;
        const int ci = (int)(&c - cases);
        auto launch_pair = [&](auto tag, int shift, float seed, bool with_consumer) {
            constexpr int T = decltype(tag)::value;
            producer<T><<<c.blocks, 256, 0, st>>>(M, c.blocks, c.piece, shift, seed, plane, contig);
            if (dirty4) streamer<<<1024, 256, 0, st>>>(scratch_a, scratch_b, dirty4);
            if (with_consumer) {
                if (per <= 5) consumer<5, T><<<c.blocks, 256, 0, st>>>(M, out, c.blocks, c.piece, plane, contig);
                else if (per <= 7) consumer<7, T><<<c.blocks, 256, 0, st>>>(M, out, c.blocks, c.piece, plane, contig);
                else consumer<9, T><<<c.blocks, 256, 0, st>>>(M, out, c.blocks, c.piece, plane, contig);
                // between the consumer and the NEXT producer too: otherwise that producer rewrites lines the consumer's
                // XCD still caches, and the price of taking them back lands in the loop's difference
                if (dirty4 && both_sides) streamer<<<1024, 256, 0, st>>>(scratch_a, scratch_b, dirty4);
            }
        };
        auto loop = [&](int shift, bool with_consumer, int iters) {
            using std::integral_constant;
            for (int it = 0; it < iters; ++it) {
                const int key = ci * 10 + (shift == 0 ? 0 : shift == 1 ? 1 : 2);
                switch (key) {
                case 0: launch_pair(integral_constant<int, 0>{}, shift, (float)it, with_consumer); break;
                case 1: launch_pair(integral_constant<int, 1>{}, shift, (float)it, with_consumer); break;
                case 2: launch_pair(integral_constant<int, 2>{}, shift, (float)it, with_consumer); break;
                case 10: launch_pair(integral_constant<int, 10>{}, shift, (float)it, with_consumer); break;
                case 11: launch_pair(integral_constant<int, 11>{}, shift, (float)it, with_consumer); break;
                case 12: launch_pair(integral_constant<int, 12>{}, shift, (float)it, with_consumer); break;
                case 20: launch_pair(integral_constant<int, 20>{}, shift, (float)it, with_consumer); break;
                case 21: launch_pair(integral_constant<int, 21>{}, shift, (float)it, with_consumer); break;
                default: launch_pair(integral_constant<int, 22>{}, shift, (float)it, with_consumer); break;
                }
            }
        };
        auto timed = [&](int shift, bool with_consumer) {
            loop(shift, with_consumer, 20);
            CK(hipStreamSynchronize(st));
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, st));
                loop(shift, with_consumer, iters);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            return best / iters * 1e3f;   // us per iteration
        };
        printf("%s: %.1f MB per hand-off\n", c.name, n4 * 16 / 1e6);
        const int shifts[] = {0, 1, 3};
        for (int s : shifts) {
            const float both = timed(s, true), prod = timed(s, false);
            printf("  shift %d (%s XCD): producer%s %.2f us, + consumer %.2f us -> consumer %.2f us = %.2f TB/s\n", s,
                   s % 8 == 0 ? "same" : "other", dirty4 ? " + stream" : "", prod, both, both - prod,
                   n4 * 16 / ((both - prod) * 1e-6) / 1e12);
        }
        CK(hipFree(M));
        CK(hipFree(out));
    }
    return 0;
}
