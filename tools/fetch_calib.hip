// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE for 16-byte-per-lane GATHERS on gfx950 (tools, not product).
//
// MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reports exactly 1/2 of the bytes of a wide coalesced
// streaming read; other access patterns are uncalibrated.  k_fb_klt3 fetches 16-byte row segments that never coalesce
// with their neighbours (one 128-byte line look-up per request).  This program issues, over a buffer far larger than
// L2 + Infinity Cache (4 GiB), patterns whose set of touched lines is known:
//   dense    : lane i reads bytes [16 i, 16 i + 16)                  -> every byte once          (reference pattern)
//   g128     : one 16-byte read per 128-byte line (stride 128 B)     -> every line once, 12.5 % of it used
//   g64      : one 16-byte read per 64-byte half line (stride 64 B)
//   g256     : one 16-byte read per 256 B (every other line)
//   rows16   : LK-like: 16 consecutive "rows" of pitch 832 B, 16 bytes each, per 3-lane group, random row origin
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (and a second pass with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum):
// counted bytes per touched line tell whether a 16-byte gather is charged 32, 64 (= a full 128-byte fill, like dense) or
// something else; the kernel times give the achieved line rate.  Prints one JSON line with times and line counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int STRIDE>
__global__ __launch_bounds__(256) void k_gather(const uint8_t *__restrict__ buf, size_t n_req, uint32_t *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i < n_req; i += step) {
        const u32x4 v = *(const u32x4 *)(buf + i * STRIDE);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// LK-like: group g (3 lanes) fetches rows r = sub + 3 k of a 16-row block at a pseudo-random origin inside "its" image
__global__ __launch_bounds__(64) void k_rows16(const uint8_t *__restrict__ buf, size_t n_img, size_t img_bytes, int pitch, int rows,
                                               int blocks_per_img, uint32_t *__restrict__ sink)
{
    const int lane = threadIdx.x, l16 = lane & 15, g = l16 / 3, sub = l16 - 3 * g;
    if (l16 >= 15) return;
    const size_t img = blockIdx.x / blocks_per_img;
    const uint32_t kp = (blockIdx.x % blocks_per_img) * 20 + (lane >> 4) * 5 + g;
    uint32_t h = (uint32_t)img * 2654435761u + kp * 40503u + 12345u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const int y0 = (int)(h % (uint32_t)(rows - 16)), x0 = (int)((h >> 12) % (uint32_t)(pitch - 32)) & ~3;
    const uint8_t *p = buf + img * img_bytes + (size_t)(y0 + sub) * pitch + x0;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < 5 || sub == 0) {
            const u32x4 v = *(const u32x4 *)(p + (size_t)3 * k * pitch);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)4 << 30;
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&sink, 256));
    CK(hipMemset(buf, 1, bytes)); CK(hipMemset(sink, 0, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[5] = {0, 0, 0, 0, 0};
    const int grid = 256 * 32;
    auto timeit = [&](int idx, auto launch) -> int {
        launch();                                        // warm-up (also the dispatch rocprofv3 sees first)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[idx], e0, e1));
        return 0;
    };
    if (timeit(0, [&] { hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, sink); })) return 1;
    if (timeit(1, [&] { hipLaunchKernelGGL(k_gather<128>, dim3(grid), dim3(256), 0, 0, buf, bytes / 128, sink); })) return 1;
    if (timeit(2, [&] { hipLaunchKernelGGL(k_gather<64>, dim3(grid), dim3(256), 0, 0, buf, bytes / 64, sink); })) return 1;
    if (timeit(3, [&] { hipLaunchKernelGGL(k_gather<256>, dim3(grid), dim3(256), 0, 0, buf, bytes / 256, sink); })) return 1;
    // 4096 "images" of 1 MiB (pitch 832 x 1260 rows), 16 blocks of 20 keypoints each: 4096 * 320 keypoints * 16 rows
    const size_t n_img = 4096, img_bytes = (size_t)1 << 20; const int pitch = 832, rows = 1260, bpi = 16;
    if (timeit(4, [&] { hipLaunchKernelGGL(k_rows16, dim3((unsigned)(n_img * bpi)), dim3(64), 0, 0, buf, n_img, img_bytes, pitch, rows, bpi, sink); })) return 1;
    printf("{\"buffer_bytes\": %zu, \"dense_ms\": %.4f, \"g128_ms\": %.4f, \"g64_ms\": %.4f, \"g256_ms\": %.4f, \"rows16_ms\": %.4f, "
           "\"dense_requests\": %zu, \"g128_lines\": %zu, \"g64_requests\": %zu, \"g256_lines\": %zu, \"rows16_requests\": %zu}\n",
           bytes, ms[0], ms[1], ms[2], ms[3], ms[4], bytes / 16, bytes / 128, bytes / 64, bytes / 256, n_img * bpi * 20 * 16);
    return 0;
}
