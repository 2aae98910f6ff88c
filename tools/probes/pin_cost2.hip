// probe: mmap + MADV_HUGEPAGE + hipHostRegister WITHOUT touching first; a kernel writes into the registered block (same address on the device?), the host reads it after the sync
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(unsigned* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)(i * 2654435761u); }
int main() {
    hipFree(nullptr);
    for (unsigned flags : {hipHostRegisterDefault, hipHostRegisterMapped | hipHostRegisterPortable}) {
        const size_t n = (size_t)256 << 20;
        double t0 = now();
        void* p = mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char* q = (char*)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
        madvise(q, n, MADV_HUGEPAGE);
        double t1 = now();
        hipError_t e = hipHostRegister(q, n, flags);
        double t2 = now();
        void* dp = nullptr; hipError_t e2 = hipHostGetDevicePointer(&dp, q, 0);
        hipLaunchKernelGGL(fill, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (unsigned*)q, n / 4);
        hipError_t e3 = hipDeviceSynchronize();
        double t3 = now();
        hipLaunchKernelGGL(fill, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (unsigned*)q, n / 4);
        hipDeviceSynchronize();
        double t4 = now();
        size_t bad = 0; const unsigned* u = (const unsigned*)q;
        for (size_t i = 0; i < n / 4; i += 4099) if (u[i] != (unsigned)(i * 2654435761u)) bad++;
        std::printf("flags %u: mmap %.2f ms, register (untouched) %.1f ms (%s), device pointer %s same=%d, kernel fill %.1f ms then %.1f ms (%.1f GB/s) (%s), wrong words %zu\n", flags, t1 - t0, t2 - t1,
                    hipGetErrorString(e), hipGetErrorString(e2), dp == (void*)q, t3 - t2, t4 - t3, n / (t4 - t3) / 1e6, hipGetErrorString(e3), bad);
        hipHostUnregister(q); munmap(p, n + (2 << 20));
    }
    {
        const size_t n = (size_t)256 << 20; void* h; hipHostMalloc(&h, n, hipHostMallocDefault);
        hipLaunchKernelGGL(fill, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (unsigned*)h, n / 4); hipDeviceSynchronize();
        double t3 = now(); hipLaunchKernelGGL(fill, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (unsigned*)h, n / 4); hipDeviceSynchronize(); double t4 = now();
        std::printf("hipHostMalloc: kernel fill %.1f ms (%.1f GB/s)\n", t4 - t3, n / (t4 - t3) / 1e6);
    }
    return 0;
}
