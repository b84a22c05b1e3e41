// Is pinning caller memory on the fly (hipHostRegister) cheaper than staging through pinned rings?  And how fast do N
// threads copy out of pinned memory?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = 1ull << 30;
    uint8_t *p = (uint8_t *)aligned_alloc(4096, N); memset(p, 3, N);
    uint8_t *d; CK(hipMalloc((void **)&d, N));
    for (size_t sz : {(size_t)32 << 20, (size_t)256 << 20, N}) {
        double t = now(); CK(hipHostRegister(p, sz, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpy(p, d, sz, hipMemcpyDeviceToHost)); double t2 = now();
        CK(hipHostUnregister(p)); double t3 = now();
        printf("register %4zu MiB: %.2f ms (%.1f GB/s), D2H into it %.1f GB/s, unregister %.2f ms\n", sz >> 20, (t1 - t) * 1e3, sz / (t1 - t) / 1e9, sz / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
    }
    uint8_t *h; CK(hipHostMalloc((void **)&h, N, hipHostMallocDefault)); memset(h, 1, N);
    uint8_t *h2; CK(hipHostMalloc((void **)&h2, N, hipHostMallocNonCoherent)); memset(h2, 1, N);
    for (int nt : {1, 4, 8, 16, 32}) {
        for (int which = 0; which < 2; which++) {
            const uint8_t *src = which ? h2 : h;
            std::vector<std::thread> th;
            double t = now();
            for (int k = 0; k < nt; k++) th.emplace_back([=] { memcpy(p + (N / nt) * k, src + (N / nt) * k, N / nt); });
            for (auto &x : th) x.join();
            printf("%2d threads pinned(%s)->pageable: %.1f GB/s\n", nt, which ? "noncoherent" : "default", N / (now() - t) / 1e9);
        }
    }
    return 0;
}
