// Is an NCHW conv's output stream slow because of HOW it is laid over HBM?  102.76 MB (8 x 64 planes of 224 x 224
// floats) written by 1568 workgroups of 256 threads, 64 KB each, no compute:
//   pattern "planes": a workgroup writes RUN contiguous bytes in each of 64 KB / RUN channel planes (planes are
//                     200,704 bytes apart) -- what a (64 channels x 256 pixels) conv tile does with RUN = 1 KB;
//   pattern "linear": the same 64 KB contiguous.
// Stores are 16 bytes per lane (b128) or 4 bytes per lane (b32, the MFMA C-layout's natural width).
//   hipcc --offload-arch=gfx950 -O3 -o bin/write_pattern write_pattern.hip && bin/write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr size_t PLANE = 224 * 224 * 4, IMG = 64 * PLANE, TOTAL = 8 * IMG;

template <int LANE_BYTES>
__global__ void __launch_bounds__(256) write_planes(char *y, int run_bytes, int tiles_per_img, float v) {
    // tile t of image n covers bytes [t*run_px.., ) of `planes_per_wg` planes ... generalised: the workgroup's 64 KB are
    // (64 KB / run) runs; run r lives in plane (r % 64) at offset ((blockIdx % tiles) * (64 KB / 64) ...) -- keep it simple:
    const int n = blockIdx.x / tiles_per_img, t = blockIdx.x % tiles_per_img;
    const int runs = 65536 / run_bytes;                    // runs per workgroup
    // the workgroup owns, in `planes` = min(64, runs) planes, a contiguous stretch of 64 KB / planes bytes
    const int planes = runs < 64 ? runs : 64;
    const int per_plane = 65536 / planes;                  // bytes per plane (= run_bytes when runs <= 64)
    const int groups = 64 / planes;                        // workgroups that share one 64-plane image slab
    const int pg = t % groups, seg = t / groups;
    for (int off = threadIdx.x * LANE_BYTES; off < 65536; off += 256 * LANE_BYTES) {
        const int pl = off / per_plane, in = off % per_plane;
        char *dst = y + (size_t)n * IMG + (size_t)(pg * planes + pl) * PLANE + (size_t)seg * per_plane + in;
        if (LANE_BYTES == 16) *reinterpret_cast<float4 *>(dst) = make_float4(v, v, v, v);
        else *reinterpret_cast<float *>(dst) = v;
    }
}

int main() {
    char *y;
    CK(hipMalloc(&y, TOTAL + (1 << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = (int)(TOTAL / 65536), per_img = wgs / 8;
    for (int lane_bytes : {16, 4})
        for (int run : {256, 1024, 4096, 16384, 65536}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 10; ++i) {
                    if (lane_bytes == 16) hipLaunchKernelGGL(write_planes<16>, dim3(wgs), dim3(256), 0, 0, y, run, per_img, 1.f + i);
                    else hipLaunchKernelGGL(write_planes<4>, dim3(wgs), dim3(256), 0, 0, y, run, per_img, 1.f + i);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms / 10 < best) best = ms / 10;
            }
            printf("%2d B/lane, runs of %5d B in %2d planes per workgroup: %6.1f us  %5.0f GB/s\n", lane_bytes, run,
                   65536 / run < 64 ? 65536 / run : 64, best * 1e3, TOTAL / best / 1e6);
        }
    CK(hipMemsetAsync(y, 0, TOTAL, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(y, i, TOTAL, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    printf("hipMemsetAsync of the same bytes: %6.1f us  %5.0f GB/s\n", best * 1e3, TOTAL / best / 1e6);
    return 0;
}
