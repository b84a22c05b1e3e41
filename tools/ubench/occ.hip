// How many workgroups of T threads with L bytes of LDS stay on a CU at once?  3 x 256 workgroups that each spin for a fixed time:
// the launch lasts one spin when three fit per CU, two when they do not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
extern __shared__ uint32_t dyn[];
template <int V> __global__ void k_spin(uint32_t *out, long long spin)
{
    uint32_t keep[V];
    for (int i = 0; i < V; i++) keep[i] = threadIdx.x * (i + 1);
    dyn[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { for (int i = 0; i < V; i++) keep[i] = keep[i] * 1664525u + dyn[(keep[(i + 1) % V] >> 7) % blockDim.x]; }
    uint32_t a = 0; for (int i = 0; i < V; i++) a ^= keep[i];
    if (a == 0x12345) out[0] = a;
}
template <int V> float run(int threads, int lds, int wgs)
{
    uint32_t *d; (void)hipMalloc(&d, 64);
    (void)hipFuncSetAttribute((const void *)k_spin<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_spin<V>, dim3(wgs), dim3(threads), lds, 0, d, 100000LL);       // 100 MHz clock: 1 ms
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_spin<V>, dim3(wgs), dim3(threads), lds, 0, d, 100000LL);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); (void)hipFree(d);
    return ms;
}
int main()
{
    const int cases[][2] = {{256, 49120}, {256, 53744}, {320, 49120}, {320, 53744}, {320, 54256}, {384, 53744}, {512, 53744}};
    for (auto &c : cases) {
        printf("threads %3d lds %5d: few registers %.2f ms, ~100 registers %.2f ms  (768 workgroups spinning 1 ms each: 1 ms = three per CU)\n",
               c[0], c[1], run<4>(c[0], c[1], 768), run<88>(c[0], c[1], 768));
    }
    return 0;
}
