// probe: pinned host memory by hipHostMalloc against malloc'd memory on transparent huge pages + hipHostRegister (ms), and the H2D/D2H rate from both
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    const size_t n = (size_t)256 << 20;
    void* d; hipMalloc(&d, n);
    for (int huge = 0; huge < 2; ++huge) {
        double t0 = now();
        void* p = mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char* q = (char*)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
        if (huge) madvise(q, n, MADV_HUGEPAGE);
        memset(q, 1, n);
        double t1 = now();
        hipError_t e = hipHostRegister(q, n, hipHostRegisterDefault);
        double t2 = now();
        hipMemcpy(d, q, n, hipMemcpyHostToDevice);
        double t3 = now();
        hipMemcpy(d, q, n, hipMemcpyHostToDevice);
        double t4 = now();
        hipMemcpy(q, d, n, hipMemcpyDeviceToHost);
        double t5 = now();
        hipHostUnregister(q);
        double t6 = now();
        munmap(p, n + (2 << 20));
        double t7 = now();
        std::printf("%s pages: map+touch %.1f ms, hipHostRegister %.1f ms (%s), H2D %.1f / %.1f ms (%.1f GB/s), D2H %.1f ms, unregister %.1f ms, munmap %.1f ms\n", huge ? "huge" : "4-KB", t1 - t0, t2 - t1,
                    hipGetErrorString(e), t3 - t2, t4 - t3, n / (t4 - t3) / 1e6, t5 - t4, t6 - t5, t7 - t6);
    }
    double t0 = now(); void* h; hipHostMalloc(&h, n, hipHostMallocDefault); double t1 = now(); memset(h, 1, n); double t2 = now();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t3 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t4 = now(); hipHostFree(h); double t5 = now();
    std::printf("hipHostMalloc 256 MB %.1f ms, touch %.1f ms, H2D %.1f / %.1f ms (%.1f GB/s), hipHostFree %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, n / (t4 - t3) / 1e6, t5 - t4);
    return 0;
}
