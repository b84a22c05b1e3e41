// global_load_lds_dwordx4 through inline asm: where does the data land with 16 active lanes and 4-byte-aligned sources?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string.h>
__device__ __forceinline__ void dma16(const void *g, uint32_t lds)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off\n\ts_nop 1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__global__ void k(const uint8_t *src, uint32_t *out, int shift)
{
    __shared__ uint32_t pad[100];
    __shared__ uint32_t ring[4][64 * 4];
    const uint32_t lane = threadIdx.x;
    pad[lane] = lane;
    for (int i = lane; i < 4 * 256; i += 64) (&ring[0][0])[i] = 0xDEAD0000u + i;
    __syncthreads();
    if (lane < 16) {
        const uint8_t *p = src + lane * 1000 + shift;
        dma16(p, (uint32_t)(uintptr_t)&ring[1][0]);
        dma16(p + 16, (uint32_t)(uintptr_t)&ring[2][0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = lane; i < 4 * 256; i += 64) out[i] = (&ring[0][0])[i];
    out[1024 + lane] = pad[lane];
}
int main()
{
    std::vector<uint8_t> h(64 * 1000 + 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 7 + (i >> 8));
    uint8_t *d; uint32_t *o; hipMalloc(&d, h.size()); hipMalloc(&o, 8192); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    for (int shift : {0, 4, 8, 12, 2}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        std::vector<uint32_t> r(2048); hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
        int ok = 0, untouched = 0;
        for (int s = 1; s <= 2; s++)
            for (int lane = 0; lane < 64; lane++)
                for (int w = 0; w < 4; w++) {
                    const uint32_t got = r[s * 256 + lane * 4 + w];
                    uint32_t want; memcpy(&want, &h[lane * 1000 + shift + (s - 1) * 16 + 4 * w], 4);
                    if (lane < 16) ok += got == want; else untouched += got == 0xDEAD0000u + s * 256 + lane * 4 + w;
                }
        int other = 0; for (int i = 0; i < 256; i++) other += r[i] == 0xDEAD0000u + i; for (int i = 768; i < 1024; i++) other += r[i] == 0xDEAD0000u + i;
        printf("shift %2d: %d / 128 words right, %d / 384 inactive lanes untouched, %d / 512 other slots untouched\n", shift, ok, untouched, other);
    }
    return 0;
}
