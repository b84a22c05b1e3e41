// unaligned LDS READS (ds_read_b32 / b64 / b128 / u16 at byte addresses that are not multiples of their size): does gfx950 do them?
// (ldsun.hip found that an unaligned ds_write_b64 does not write what it should.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
__global__ void k(uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t st[4096];
    const uint32_t lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) st[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)st + 32 * lane + (lane % 16);
    uint32_t r32, r16; uint64_t r64; uint32_t q0, q1, q2, q3;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r32) : "v"(a) : "memory");
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r16) : "v"(a) : "memory");
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r64) : "v"(a) : "memory");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 r128;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r128) : "v"(a) : "memory");
    q0 = r128.x; q1 = r128.y; q2 = r128.z; q3 = r128.w;
    uint32_t *o = out + lane * 8;
    o[0] = r32; o[1] = r16; o[2] = (uint32_t)r64; o[3] = (uint32_t)(r64 >> 32); o[4] = q0; o[5] = q1; o[6] = q2; o[7] = q3;
}
int main()
{
    uint32_t *o; hipMalloc(&o, 64 * 32); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    std::vector<uint32_t> r(64 * 8); hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    uint8_t st[4096 + 32]; for (int i = 0; i < 4096 + 32; i++) st[i] = (uint8_t)(i * 7 + (i >> 8));
    int bad32 = 0, bad16 = 0, bad64 = 0, bad128 = 0;
    for (int lane = 0; lane < 64; lane++) {
        const int a = 32 * lane + lane % 16; uint32_t w[4]; memcpy(w, st + a, 16);
        bad32 += r[lane * 8] != w[0]; bad16 += r[lane * 8 + 1] != (w[0] & 0xFFFF);
        bad64 += r[lane * 8 + 2] != w[0] || r[lane * 8 + 3] != w[1];
        bad128 += r[lane * 8 + 4] != w[0] || r[lane * 8 + 5] != w[1] || r[lane * 8 + 6] != w[2] || r[lane * 8 + 7] != w[3];
    }
    printf("unaligned LDS reads, lanes with wrong data of 64 (16 of them aligned): b32 %d  u16 %d  b64 %d  b128 %d\n", bad32, bad16, bad64, bad128);
    return 0;
}
