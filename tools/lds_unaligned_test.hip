// Does gfx950 serve ds_read_b96 at BYTE granularity, and at what cost?  (lk3.hip reads the twelve bytes of a staged row from the
// window's first column.)  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_test.hip -o tools/bin/lds_unaligned_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_check(uint32_t *out, const uint8_t *in)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 32 + 64];
    for (int i = threadIdx.x; i < 64 * 32 + 64; i += 64) lds[i] = in[i];
    __syncthreads();
    for (int off = 0; off < 16; off++) {
        const u32x3_a1 v = *(const u32x3_a1 *)(lds + threadIdx.x * 32 + off);
        out[(off * 64 + threadIdx.x) * 3 + 0] = v.x; out[(off * 64 + threadIdx.x) * 3 + 1] = v.y; out[(off * 64 + threadIdx.x) * 3 + 2] = v.z;
    }
}
template <int MODE>
__global__ void k_time(uint32_t *out, int reps, int offmask)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 29 + 16];
    for (int i = threadIdx.x; i < 64 * 29 + 16; i += 64) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    const int off = (threadIdx.x * 7) & offmask;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint32_t *row = lds + threadIdx.x * 29 + 4 * m + (r & 3);
            if (MODE == 0) {
                const uint32_t a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
                acc += __builtin_amdgcn_alignbyte(a1, a0, off) ^ __builtin_amdgcn_alignbyte(a2, a1, off) ^ __builtin_amdgcn_alignbyte(a3, a2, off);
            } else {
                const u32x3_a1 v = *(const u32x3_a1 *)((const uint8_t *)row + off);
                acc += v.x ^ v.y ^ v.z;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main()
{
    std::vector<uint8_t> h(64 * 32 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 37 + (i >> 8));
    uint8_t *din; uint32_t *dout;
    hipMalloc(&din, h.size()); hipMalloc(&dout, 1 << 24);
    hipMemcpy(din, h.data(), h.size(), hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(dout, din);
    std::vector<uint32_t> o(16 * 64 * 3);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int off = 0; off < 16; off++) for (int t = 0; t < 64; t++) for (int k = 0; k < 3; k++) {
        uint32_t e = 0;
        for (int b = 0; b < 4; b++) e |= (uint32_t)h[t * 32 + off + 4 * k + b] << (8 * b);
        if (e != o[(off * 64 + t) * 3 + k]) bad++;
    }
    printf("unaligned ds_read_b96: %s (%d wrong dwords)\n", bad ? "WRONG" : "correct at every byte offset 0..15", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) for (int offmask = 0; offmask <= 3; offmask += 3) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k_time<0><<<256 * 48, 64>>>(dout, 2000, offmask); else k_time<1><<<256 * 48, 64>>>(dout, 2000, offmask);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s, offsets %s: %.3f ms  (%.2f ns per wave-row)\n", mode ? "ds_read_b96 at byte address" : "4 x ds_read_b32 + 3 x v_alignbyte", offmask ? "0..3 mixed" : "all 0",
               best, best * 1e6 / (256.0 * 48 * 2000 * 4) * 1024 /*per SIMD*/);
    }
    return bad ? 1 : 0;
}
