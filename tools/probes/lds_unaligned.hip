// probe: are byte-unaligned 32-bit / 16-bit LDS accesses exact on gfx950 (unaligned access mode)? Each lane writes a dword at byte
// offset 5 * lane + 1 of a zeroed LDS array, then every byte is read back one by one and compared with what a byte-wise writer would
// have left. Prints "ok" or the first difference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t __attribute__((aligned(1))) u16u;
__global__ void k(uint8_t* out, uint32_t* rd) {
    __shared__ __attribute__((aligned(16))) uint8_t s[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) s[i] = 0;
    __syncthreads();
    *reinterpret_cast<u32u*>(s + 7 * l + 1) = 0x01020304u * (uint32_t)(l + 1);
    *reinterpret_cast<u16u*>(s + 7 * l + 5) = (uint16_t)(0x1111u * (uint32_t)(l % 15 + 1));
    __syncthreads();
    for (int i = l; i < 1024; i += 64) out[i] = s[i];
    rd[l] = *reinterpret_cast<u32u*>(s + 3 * l + 2);      // unaligned dword READ
}
int main() {
    uint8_t* d; uint32_t* r;
    hipMalloc(&d, 1024); hipMalloc(&r, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r);
    std::vector<uint8_t> h(1024), e(1024, 0); std::vector<uint32_t> hr(64);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), r, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { uint32_t v = 0x01020304u * (uint32_t)(l + 1); std::memcpy(&e[7 * l + 1], &v, 4); uint16_t w = (uint16_t)(0x1111u * (uint32_t)(l % 15 + 1)); std::memcpy(&e[7 * l + 5], &w, 2); }
    for (int i = 0; i < 1024; ++i) if (h[i] != e[i]) { std::printf("write differs at byte %d: %u vs %u\n", i, h[i], e[i]); return 1; }
    for (int l = 0; l < 64; ++l) { uint32_t v; std::memcpy(&v, &e[3 * l + 2], 4); if (v != hr[l]) { std::printf("read differs at lane %d\n", l); return 1; } }
    std::printf("ok: unaligned LDS dword / short writes and dword reads are exact\n");
    return 0;
}
