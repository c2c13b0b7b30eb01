// micro-benchmark: how fast can a kernel pull one camera frame (361 KB) out of pinned host memory (zero-copy) compared with a
// hipMemcpyAsync of the same buffer?  Decides whether the tracker's H2D copy could be folded into k_clahe_lut.
//   hipcc --offload-arch=gfx950 -O3 tools/zc_read.hip -o tools/bin/zc_read && tools/bin/zc_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
int main()
{
    const int W = 752, H = 480, bytes = W * H, n16 = bytes / 16;
    uint8_t *h; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    for (int i = 0; i < bytes; i++) h[i] = (uint8_t)(i * 7);
    uint4 *hd; CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    uint4 *d; CK(hipMalloc(&d, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {16, 64, 150, 360, 1024};
    for (int g : grids) for (int bs : {64, 256}) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; rep++) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_pull, dim3(g), dim3(bs), 0, s, hd, d, n16);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("zero-copy kernel pull  grid %4d x %3d : %7.1f us  (%5.1f GB/s)\n", g, bs, best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
    float best = 1e9f;
    for (int rep = 0; rep < 20; rep++) {
        CK(hipEventRecord(e0, s));
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("hipMemcpyAsync H2D (pinned)            : %7.1f us  (%5.1f GB/s)\n", best * 1e3, bytes / (best * 1e-3) / 1e9);
    return 0;
}
