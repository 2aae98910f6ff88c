// probe: what does a wave64 integer VALU instruction cost a gfx950 SIMD when W wavefronts share it? Every wave runs N rounds of 8
// independent chains x 8 instructions (v_add / v_xor / v_and_or / v_perm / v_alignbyte / v_lshl_add / v_sad_u8 / v_bfe); grid = CUs x
// 4 SIMDs x W waves. Prints SIMD cycles per wave-instruction (from s_memtime around the loop, max over waves) for W = 1, 2, 4, 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(uint32_t* out, unsigned long long* cyc, int n) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + blockIdx.x;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < n; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t x = a[i];
            x = x + 0x01010101u;
            x = x ^ (uint32_t)r;
            x = (x & 0x03030303u) | 0x20202020u;
            x = __builtin_amdgcn_perm(x, a[(i + 1) & 7], 0x02010003u);
            x = __builtin_amdgcn_alignbyte(x, a[(i + 2) & 7], 3);
            x = (x << 2) + a[(i + 3) & 7];
            x = __builtin_amdgcn_sad_u8(x, 0u, a[(i + 4) & 7]);
            x = (x >> 3) & 0xffffu;
            a[i] = x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    std::printf("CUs %d, clock %d kHz\n", cus, clk);
    const int n = 2000;
    for (int W : {1, 2, 4, 8}) {
        const int threads = 256, blocks = cus * W;      // 4 waves per block: one per SIMD (the hardware spreads a block's waves over the SIMDs)
        uint32_t* out; unsigned long long* cyc;
        hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double instr = (double)n * 64;      // VALU instructions per wave (8 chains x 8)
        std::printf("W = %d waves per SIMD: kernel %.3f ms; timer ticks per wave-instruction, median wave %.2f -> per SIMD %.2f; by the wall clock at %.2f GHz: %.2f cycles per wave-instruction per SIMD\n",
                    W, ms, (double)h[h.size() / 2] / instr, (double)h[h.size() / 2] / instr / W, clk / 1e6, ms * 1e-3 * clk * 1e3 / (instr * W));
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
