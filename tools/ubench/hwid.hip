// Where do the waves of a workgroup land?  4 waves per workgroup, 42.7 KiB of LDS (three workgroups per CU), every workgroup stays
// for a while; per workgroup: XCC, CU, the SIMD of each wave, the LDS base / size of the allocation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k_where(uint32_t *out, int spin)
{
    __shared__ uint32_t pad[10680];
    const int tid = threadIdx.x, wave = tid >> 6;
    uint32_t hw, lds, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[tid * 41 % 10680] = hw;
    uint64_t t0 = __builtin_readcyclecounter();
    uint32_t a = tid;
    while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) a = a * 1664525u + pad[a % 10680];
    if ((tid & 63) == 0) { uint32_t *o = out + (blockIdx.x * 4 + wave) * 4; o[0] = hw; o[1] = lds; o[2] = xcc; o[3] = a; }
}
int main()
{
    const int nwg = 2048;
    uint32_t *d; hipMalloc(&d, nwg * 64); std::vector<uint32_t> h(nwg * 16);
    hipLaunchKernelGGL(k_where, dim3(nwg), dim3(256), 0, 0, d, 2000000);
    hipMemcpy(h.data(), d, nwg * 64, hipMemcpyDeviceToHost);
    int distinct = 0, by_wave = 0;
    std::map<uint32_t, std::vector<int>> cu;
    for (int b = 0; b < nwg; b++) {
        uint32_t simd[4], mask = 0;
        for (int w = 0; w < 4; w++) { simd[w] = (h[(b * 4 + w) * 4] >> 4) & 3; mask |= 1u << simd[w]; }
        distinct += mask == 15; by_wave += simd[0] == 0 && simd[1] == 1 && simd[2] == 2 && simd[3] == 3;
        const uint32_t hw = h[b * 16], xcc = h[b * 16 + 2] & 0xF;
        cu[(xcc << 16) | ((hw >> 8) & 0xF) | (((hw >> 13) & 7) << 4)].push_back(b);      // CU_ID 11:8, SE_ID 15:13
        if (b < 12 || (b >= 768 && b < 776)) {
            const uint32_t lds = h[b * 16 + 1];
            printf("wg %4d xcc %u se %u cu %2u  simd %u %u %u %u  wave slot %2u %2u %2u %2u  lds base %3u size %3u (raw %08x)\n", b, xcc, (hw >> 13) & 7, (hw >> 8) & 0xF,
                   simd[0], simd[1], simd[2], simd[3], h[b * 16] & 0xF, h[b * 16 + 4] & 0xF, h[b * 16 + 8] & 0xF, h[b * 16 + 12] & 0xF, lds & 0xFF, (lds >> 12) & 0x1FF, lds);
        }
    }
    printf("%d workgroups: %d with four distinct SIMDs, %d with wave w on SIMD w; %zu distinct (xcc, se, cu)\n", nwg, distinct, by_wave, cu.size());
    int shown = 0;
    for (auto &kv : cu) { if (shown++ >= 4) break; printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d(seq wave simd %u, lds %u)", b, (h[(b * 4 + 1) * 4] >> 4) & 3, h[b * 16 + 1] & 0xFF); printf("\n"); }
    return 0;
}
