#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>
__global__ __launch_bounds__(256) void k_chain(uint32_t *out, int iters, int active, uint64_t seed)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (lane >= active) return;
    uint64_t x = seed + tid; uint32_t a = tid + (uint32_t)seed;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { a += 5; a ^= 0x55; a += 7; a ^= 0x33; a += 11; a ^= 0x77; a += 13; a ^= 0x11; }          // 8 x 32-bit
        if (KIND == 1) { x <<= (a & 3); x >>= 1; x <<= (a & 1); x >>= 1; x <<= 2; x >>= 1; x <<= 1; x >>= 2; x |= 0x8000000000ull; }   // 8 x 64-bit shifts
        if (KIND == 2) { a = (a >> 3) & 0xff; a += 0x1234567; a = (a >> 2) & 0xfff; a += 0x7654321; a = (a >> 1) & 0xffff; a += 0x1111111; a = (a >> 4) & 0xff; a += 0x2222222; }   // bfe + add
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (tid == 0) { out[0] = (uint32_t)((t1 - t0) * 100 / iters); out[1] = a + (uint32_t)x; }
}
int main()
{
    uint32_t *d; hipMalloc(&d, 64); uint32_t h[2];
    for (int active : {1, 16, 32, 33, 48, 63, 64}) {
        hipLaunchKernelGGL(k_chain<0>, dim3(78), dim3(256), 0, 0, d, 20000, active, 12345ull); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        float a0 = h[0] / 100.0f;
        hipLaunchKernelGGL(k_chain<1>, dim3(78), dim3(256), 0, 0, d, 20000, active, 12345ull); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        float a1 = h[0] / 100.0f;
        hipLaunchKernelGGL(k_chain<2>, dim3(78), dim3(256), 0, 0, d, 20000, active, 12345ull); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("active=%2d  cycles per 8 ops: add/xor32 %.1f   shift64 %.1f   bfe+add %.1f\n", active, a0, a1, h[0] / 100.0f);
    }
    // one wave per block instead of four
    for (int active : {1, 64}) {
        hipLaunchKernelGGL(k_chain<0>, dim3(78), dim3(64), 0, 0, d, 20000, active, 12345ull); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("1 wave/WG active=%2d add/xor32 %.1f\n", active, h[0] / 100.0f);
    }
    return 0;
}
