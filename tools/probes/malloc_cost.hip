// probe: what hipMalloc / hipFree / hipHostMalloc cost by size on this box (ms per call)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    for (size_t mb : {1, 16, 64, 256, 1024, 4096, 16384}) {
        void* p[4]; double t0 = now();
        for (int i = 0; i < 4; ++i) hipMalloc(&p[i], mb << 20);
        double t1 = now();
        for (int i = 0; i < 4; ++i) hipFree(p[i]);
        double t2 = now();
        std::printf("hipMalloc %6zu MB: %.3f ms per call, hipFree %.3f ms\n", mb, (t1 - t0) / 4, (t2 - t1) / 4);
    }
    for (size_t mb : {1, 32, 256}) {
        void* p; double t0 = now(); hipHostMalloc(&p, mb << 20, hipHostMallocDefault); double t1 = now(); hipHostFree(p); double t2 = now();
        std::printf("hipHostMalloc %4zu MB: %.3f ms, hipHostFree %.3f ms\n", mb, t1 - t0, t2 - t1);
    }
    return 0;
}
