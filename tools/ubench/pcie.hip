// PCIe legs of the host pipeline: pinned H2D / D2H bandwidth alone and both at once, per copy size.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = 1ull << 30;
    uint8_t *h1, *h2, *d1, *d2;
    CK(hipHostMalloc((void **)&h1, N, hipHostMallocDefault)); CK(hipHostMalloc((void **)&h2, N, hipHostMallocDefault));
    CK(hipMalloc((void **)&d1, N)); CK(hipMalloc((void **)&d2, N));
    memset(h1, 1, N); memset(h2, 2, N);
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (size_t sz : {(size_t)64 << 10, (size_t)1 << 20, (size_t)32 << 20, (size_t)256 << 20, N}) {
        const int reps = (int)(N / sz > 64 ? 64 : N / sz < 2 ? 2 : N / sz);
        CK(hipMemcpyAsync(d1, h1, sz, hipMemcpyHostToDevice, a)); CK(hipStreamSynchronize(a));
        double t = now();
        for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(d1 + (size_t)i * sz % N, h1 + (size_t)i * sz % N, sz, hipMemcpyHostToDevice, a));
        CK(hipStreamSynchronize(a));
        const double h2d = reps * (double)sz / (now() - t) / 1e9;
        t = now();
        for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(h2 + (size_t)i * sz % N, d2 + (size_t)i * sz % N, sz, hipMemcpyDeviceToHost, b));
        CK(hipStreamSynchronize(b));
        const double d2h = reps * (double)sz / (now() - t) / 1e9;
        t = now();
        for (int i = 0; i < reps; i++) {
            CK(hipMemcpyAsync(d1 + (size_t)i * sz % N, h1 + (size_t)i * sz % N, sz, hipMemcpyHostToDevice, a));
            CK(hipMemcpyAsync(h2 + (size_t)i * sz % N, d2 + (size_t)i * sz % N, sz, hipMemcpyDeviceToHost, b));
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        const double both = 2.0 * reps * (double)sz / (now() - t) / 1e9;
        printf("size %8zu KiB: H2D %6.1f GB/s  D2H %6.1f GB/s  both %6.1f GB/s (sum)\n", sz >> 10, h2d, d2h, both);
    }
    // latency of a tiny copy + sync, and of an empty kernel + sync
    for (int k = 0; k < 2; k++) {
        double t = now();
        for (int i = 0; i < 200; i++) { CK(hipMemcpyAsync(k ? (void *)h2 : (void *)d1, k ? (void *)d2 : (void *)h1, 4096, k ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, a)); CK(hipStreamSynchronize(a)); }
        printf("%s 4 KiB + sync: %.1f us\n", k ? "D2H" : "H2D", (now() - t) / 200 * 1e6);
    }
    // pageable memcpy speed, one thread
    uint8_t *p = (uint8_t *)malloc(N); memset(p, 3, N);
    double t = now(); memcpy(p, h2, N); printf("memcpy pinned->pageable 1 thread: %.1f GB/s\n", N / (now() - t) / 1e9);
    t = now(); memcpy(h1, p, N); printf("memcpy pageable->pinned 1 thread: %.1f GB/s\n", N / (now() - t) / 1e9);
    return 0;
}
