// scratch micro-benchmark: dependent LDS reads / ALU chains at the FSE kernel's launch shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void k_lds_chase(uint32_t *out, int iters, int active, int ldsbytes_dummy)
{
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = (i * 7 + 13) & 8191;
    __syncthreads();
    if (lane >= active) return;
    uint32_t idx = tid;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) idx = lds[idx];
    uint64_t t1 = __builtin_readcyclecounter();
    if (tid == 0) { out[0] = (uint32_t)((t1 - t0) / iters); out[1] = idx; }
}
__global__ __launch_bounds__(256) void k_alu_chain(uint32_t *out, int iters, int active, uint64_t seed)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (lane >= active) return;
    uint64_t x = seed + tid; uint32_t a = tid;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        // 8 dependent ops: 64-bit shift, bfe, add, 32-bit shift, and, sub, 64-bit shift, add
        x = x << (a & 7); a = (uint32_t)(x >> 32) >> 3; a += 5; a = a >> 1; a &= 0xffff; a = 77 - a; x = x >> (a & 3); a += (uint32_t)x;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (tid == 0) { out[0] = (uint32_t)((t1 - t0) * 100 / iters); out[1] = a; }
}
__global__ __launch_bounds__(256) void k_walk(uint32_t *out, int iters, int active)
{
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = (i * 2654435761u) | 0x00011000;
    __syncthreads();
    if (lane >= active) return;
    uint32_t s0 = tid & 511, s1 = (tid * 3) & 511, s2 = (tid * 5) & 255; uint64_t W = 0x123456789abcdefull * (tid + 1);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        uint32_t c0 = lds[s0], c1 = lds[512 + s1], c2 = lds[1024 + s2];
        uint32_t n0 = (c0 >> 8) & 15, n1 = (c1 >> 8) & 15, n2 = (c2 >> 8) & 15;
        uint64_t S = W << ((c0 >> 12) & 31);
        uint32_t h0 = (uint32_t)(S >> 32); S <<= n0; uint32_t h1 = (uint32_t)(S >> 32); S <<= n1; uint32_t h2 = (uint32_t)(S >> 32);
        s0 = ((c0 >> 17) + (n0 ? h0 >> (32 - n0) : 0)) & 511; s1 = ((c1 >> 17) + (n1 ? h1 >> (32 - n1) : 0)) & 511; s2 = ((c2 >> 17) + (n2 ? h2 >> (32 - n2) : 0)) & 255;
        W = W * 6364136223846793005ull + 1442695040888963407ull;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (tid == 0) { out[0] = (uint32_t)((t1 - t0) / iters); out[1] = s0 + s1 + s2; }
}
int main()
{
    uint32_t *d; hipMalloc(&d, 64); uint32_t h[2];
    for (int lds : {32768, 150000}) for (int active : {1, 7, 64}) {
        hipFuncSetAttribute((const void *)k_lds_chase, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
        hipFuncSetAttribute((const void *)k_walk, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
        hipLaunchKernelGGL(k_lds_chase, dim3(78), dim3(256), lds, 0, d, 20000, active, 0); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("lds=%d active=%d  lds_chase cycles/step %u\n", lds, active, h[0]);
        hipLaunchKernelGGL(k_alu_chain, dim3(78), dim3(256), 0, 0, d, 20000, active, 12345ull); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("lds=%d active=%d  alu_chain cycles per 8-op group %.2f\n", lds, active, h[0] / 100.0);
        hipLaunchKernelGGL(k_walk, dim3(78), dim3(256), lds, 0, d, 20000, active); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("lds=%d active=%d  walk cycles/iter %u\n", lds, active, h[0]);
    }
    return 0;
}
