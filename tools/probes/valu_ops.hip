// probe: SIMD cycles per wave64 instruction, by instruction, at 8 waves per SIMD (wall clock at the reported clock rate). 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(uint32_t* out, int n) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + blockIdx.x;
    const uint32_t c = out[0];
    for (int r = 0; r < n; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t x = a[i];
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 1) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 2) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(c), "v"(a[(i + 1) & 7]));
            if (OP == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(c), "v"(a[(i + 1) & 7]));
            if (OP == 4) asm volatile("v_alignbyte_b32 %0, %1, %2, 3" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 5) asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 6) asm volatile("v_sad_u8 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(c), "v"(a[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_bfe_u32 %0, %1, 3, 16" : "=v"(x) : "v"(x));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 10) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(x) : "v"(x));
            if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(x));
            if (OP == 12) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(c), "v"(a[(i + 1) & 7]));
            if (OP == 13) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 14) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 15) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(x), "v"(c) : "vcc");
            if (OP == 17) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(x) : "v"(x), "v"(c) : "s20", "s21");
            if (OP == 18) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(x), "v"(c) : "vcc");
            if (OP == 19) asm volatile("v_cmp_lt_u32 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(x) : "v"(x), "v"(c) : "s20", "s21");
            if (OP == 20) asm volatile("v_min_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 21) asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(c), "v"(a[(i + 1) & 7]));
            if (OP == 22) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 23) asm volatile("v_or_b32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 24) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c));
            if (OP == 25) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(x) : "v"(c), "v"(x));
            if (OP == 26) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c));
            if (OP == 27) asm volatile("v_readlane_b32 s20, %0, 5" :: "v"(x) : "s20");
            if (OP == 28) asm volatile("v_add_u32 %0, 0x01010101, %1" : "=v"(x) : "v"(x));
            if (OP == 29) asm volatile("v_add_u32 %0, s20, %1" : "=v"(x) : "v"(x) : "s20");
            if (OP == 16) asm volatile("s_add_u32 s20, s20, 1\n\ts_and_b32 s21, s21, s20" ::: "s20", "s21", "scc");
            a[i] = x;
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int cus, int clk, uint32_t* out) {
    const int n = 400, W = 8, blocks = cus * W;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 4); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)n * 64 * ((OP == 16 || OP == 18 || OP == 19) ? 2 : 1);
    std::printf("%-18s %.2f cycles per wave-instruction per SIMD\n", name, ms * 1e-3 * clk * 1e3 / (instr * W));
}
int main() {
    int cus = 0, clk = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0); hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    uint32_t* out; hipMalloc(&out, ((size_t)cus * 8 * 256 + 1) * 4); hipMemset(out, 0, 4);
    run<0>("v_add_u32", cus, clk, out); run<1>("v_xor_b32", cus, clk, out); run<2>("v_and_or_b32", cus, clk, out); run<3>("v_perm_b32", cus, clk, out);
    run<4>("v_alignbyte_b32", cus, clk, out); run<5>("v_lshl_add_u32", cus, clk, out); run<6>("v_sad_u8", cus, clk, out); run<7>("v_bfe_u32", cus, clk, out);
    run<8>("v_mul_lo_u32", cus, clk, out); run<9>("v_cndmask_b32", cus, clk, out); run<10>("v_lshlrev_b32", cus, clk, out); run<11>("v_mov_b32 dpp", cus, clk, out);
    run<12>("v_add3_u32", cus, clk, out); run<13>("v_pk_add_u16", cus, clk, out); run<14>("v_bcnt_u32_b32", cus, clk, out); run<15>("v_cmp_lt_u32", cus, clk, out);
    run<16>("s_add + s_and", cus, clk, out);
    run<17>("v_cndmask e64 sgpr", cus, clk, out); run<18>("v_cmp vcc + cndmask", cus, clk, out); run<19>("v_cmp sgpr + cndmask", cus, clk, out);
    run<20>("v_min_u32", cus, clk, out); run<21>("v_med3_i32", cus, clk, out); run<22>("v_and_b32", cus, clk, out); run<23>("v_or_b32", cus, clk, out);
    run<24>("v_sub_u32", cus, clk, out); run<25>("v_lshrrev_b32 (vgpr shift)", cus, clk, out); run<26>("v_mov_b32", cus, clk, out); run<27>("v_readlane_b32", cus, clk, out);
    run<28>("v_add_u32 literal", cus, clk, out); run<29>("v_add_u32 sgpr", cus, clk, out);
    return 0;
}
