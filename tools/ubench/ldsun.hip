// unaligned 8-byte LDS stores / 16-byte loads: what the compiler emits and whether the hardware does them
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint8_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t st[64][48];
    const uint32_t lane = threadIdx.x;
    for (int i = 0; i < 48; i++) st[lane][i] = 0xEE;
    __syncthreads();
    const uint64_t v = 0x0807060504030201ull + lane * 0x1010101010101010ull;
    __attribute__((address_space(3))) uint8_t *p = (__attribute__((address_space(3))) uint8_t *)&st[lane][lane % 24];
    *(__attribute__((address_space(3), aligned(1))) uint64_t *)p = v;       // hmm: aligned(1) on the pointee
    __syncthreads();
    for (int i = 0; i < 48; i++) out[lane * 48 + i] = st[lane][i];
}
int main()
{
    uint8_t *o; hipMalloc(&o, 64 * 48); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    std::vector<uint8_t> r(64 * 48); hipMemcpy(r.data(), o, r.size(), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int i = 0; i < 48; i++) {
            const int off = lane % 24; uint8_t want = 0xEE;
            if (i >= off && i < off + 8) want = (uint8_t)((i - off + 1) + lane * 0x10);
            bad += r[lane * 48 + i] != want;
        }
    printf("unaligned ds_write_b64: %d wrong bytes\n", bad);
    return 0;
}
