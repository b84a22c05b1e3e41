// Where do the workgroups of two kernels on two queues land?  (HW_REG_XCC_ID per workgroup)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return x; }
__global__ void big(unsigned *o, int spin) { __shared__ unsigned pad[7000]; if (threadIdx.x == 0) o[blockIdx.x] = xcc(); for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100); if (spin < 0) o[0] = pad[threadIdx.x]; }
__global__ void small(unsigned *o, int spin) { if (threadIdx.x == 0) o[blockIdx.x] = xcc(); for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100); }
static void show(const char *name, const std::vector<unsigned> &v)
{
    int match = 0; std::vector<int> per(16, 0);
    for (size_t i = 0; i < v.size(); i++) { match += v[i] == i % 8; per[v[i] & 15]++; }
    printf("%s: %zu workgroups, %d on XCD b %% 8; first 32:", name, v.size(), match);
    for (size_t i = 0; i < 32 && i < v.size(); i++) printf(" %u", v[i]);
    printf("; per XCD:"); for (int i = 0; i < 8; i++) printf(" %d", per[i]); printf("\n");
}
int main()
{
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned *da, *db; hipMalloc(&da, 4096 * 4); hipMalloc(&db, 4096 * 4);
    for (int round = 0; round < 3; round++) {
        hipMemsetAsync(da, 0xff, 4096 * 4, a); hipMemsetAsync(db, 0xff, 4096 * 4, b); hipDeviceSynchronize();
        if (round == 2) hipLaunchKernelGGL(small, dim3(13), dim3(64), 0, a, db + 2048, 0);      // (something odd in front: does the rotation carry over?)
        if (round == 1) hipLaunchKernelGGL(small, dim3(128), dim3(64), 0, b, db, 300);          // the small kernel first
        hipLaunchKernelGGL(big, dim3(2048), dim3(256), 0, a, da, 300);
        if (round != 1) hipLaunchKernelGGL(small, dim3(128), dim3(64), 0, b, db, 300);
        hipDeviceSynchronize();
        std::vector<unsigned> va(2048), vb(128);
        hipMemcpy(va.data(), da, 2048 * 4, hipMemcpyDeviceToHost); hipMemcpy(vb.data(), db, 128 * 4, hipMemcpyDeviceToHost);
        printf("round %d\n", round); show("  big  ", va); show("  small", vb);
    }
    return 0;
}
