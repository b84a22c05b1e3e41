#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void k_chain(uint32_t *out, int iters, int n, int stride, uint64_t seed)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if ((lane % stride) != 0 || lane / stride >= n) return;
    uint32_t a = tid + (uint32_t)seed;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { a = (a >> 3) & 0xff; a += 0x1234567; a = (a >> 2) & 0xfff; a += 0x7654321; a = (a >> 1) & 0xffff; a += 0x1111111; a = (a >> 4) & 0xff; a += 0x2222222; }
    uint64_t t1 = __builtin_readcyclecounter();
    if (tid == 0) { out[0] = (uint32_t)((t1 - t0) * 100 / iters); out[1] = a; }
}
int main()
{
    uint32_t *d; hipMalloc(&d, 64); uint32_t h[2];
    for (int stride : {1, 2, 4, 8, 9, 16, 32})
        for (int n : {1, 2, 4, 7, 8}) {
            if (n * stride > 64 + stride - 1) continue;
            hipLaunchKernelGGL(k_chain, dim3(78), dim3(256), 0, 0, d, 20000, n, stride, 12345ull); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
            printf("stride=%2d n=%d  cycles/iter %.1f\n", stride, n, h[0] / 100.0f);
        }
    return 0;
}
