// probe: what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for a KNOWN number of bytes, by access pattern -- the calibration
// MI355X_MICROARCH.md (HBM) asks for before an absolute is trusted: "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read (16 B/lane) ... other access widths are uncalibrated". Every kernel below moves N bytes of a buffer far
// larger than the Infinity Cache exactly once (reads) or writes N bytes once; tools/fetch_calib.sh runs the binary under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE and divides. The patterns are those of the path's kernels:
//   r1      one byte per lane, consecutive lanes consecutive bytes                      (the gathers of single codes)
//   r4      one aligned dword per lane                                                   (K2: a tile row of a record, 256 bytes per wavefront)
//   r4u     one UNALIGNED dword per lane (base + 1)                                      (K2's rows start anywhere)
//   r16     one aligned 16-byte load per lane                                            (the guide's case)
//   r16u    one UNALIGNED 16-byte load per lane (base + 5), lanes 16 bytes apart         (k_pileup_runs: 16 read bases per piece)
//   r16k1   k_pileup_runs' own mix: per lane an unaligned 16-byte load + the dword before it, pieces of one task 16-272 bytes apart in
//           runs of ~11 lanes (an op's pieces are consecutive, the next op starts a few bytes on)
//   w16 / w16u / w1   stores: aligned 16 bytes, unaligned 16 bytes, single bytes
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/fetch_calib tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); std::exit(1); } } while (0)

__global__ void k_r1(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    uint32_t acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_r4(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    uint32_t acc = 0;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i + 4 <= n; i += (long long)gridDim.x * blockDim.x * 4) acc += *reinterpret_cast<const uint32_t*>(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_r4u(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    uint32_t acc = 0;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4 + 1; i + 4 <= n; i += (long long)gridDim.x * blockDim.x * 4) acc += *reinterpret_cast<const u32u*>(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_r16(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    uint32_t acc = 0;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16; i + 16 <= n; i += (long long)gridDim.x * blockDim.x * 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_r16u(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    uint32_t acc = 0;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16 + 5; i + 16 <= n; i += (long long)gridDim.x * blockDim.x * 16) {
        const u32x4u v = *reinterpret_cast<const u32x4u*>(p + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// per wavefront a "task" of 64 pieces: runs of 11 consecutive 16-byte pieces, every run starting 3 bytes behind the end of the one before (an
// insertion), the dword before every piece loaded as well; the tasks of a launch tile the buffer
__global__ void k_r16k1(const uint8_t* __restrict__ p, long long n, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int run = lane / 11, in_run = lane % 11;
    const long long task_bytes = 6 * (11 * 16 + 3);      // what a task's 64 pieces span (the last run is short)
    uint32_t acc = 0;
    for (long long t = wave; (t + 1) * task_bytes + 32 <= n; t += n_waves) {
        const long long a = 8 + t * task_bytes + (long long)run * (11 * 16 + 3) + in_run * 16;
        const u32x4u v = *reinterpret_cast<const u32x4u*>(p + a);
        const uint32_t b = *reinterpret_cast<const u32u*>(p + a - 4);
        acc += v.x ^ v.y ^ v.z ^ v.w ^ b;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_w16(uint8_t* __restrict__ p, long long n) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16; i + 16 <= n; i += (long long)gridDim.x * blockDim.x * 16)
        *reinterpret_cast<uint4*>(p + i) = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void k_w16u(uint8_t* __restrict__ p, long long n) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16 + 5; i + 16 <= n; i += (long long)gridDim.x * blockDim.x * 16) {
        u32x4u v; v.x = (uint32_t)i; v.y = 1u; v.z = 2u; v.w = 3u;
        *reinterpret_cast<u32x4u*>(p + i) = v;
    }
}
__global__ void k_w1(uint8_t* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (uint8_t)i;
}

int main(int argc, char** argv) {
    const long long n = (argc > 1 ? std::atoll(argv[1]) : 2048ll) << 20;      // bytes per kernel (default 2 GiB: eight times the Infinity Cache)
    uint8_t* p = nullptr; uint32_t* out = nullptr;
    CHECK(hipMalloc(&p, (size_t)n + 4096));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(p, 1, (size_t)n + 4096));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(4096), block(256);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto timed = [&](const char* name, auto launch) {
        launch();      // (untimed: the counters count both launches; the script divides by two)
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        std::printf("%s bytes %lld ms %.3f GB/s %.0f\n", name, n, ms, (double)n / ms * 1e-6);
    };
    timed("k_r1", [&] { hipLaunchKernelGGL(k_r1, grid, block, 0, 0, p, n, out); });
    timed("k_r4", [&] { hipLaunchKernelGGL(k_r4, grid, block, 0, 0, p, n, out); });
    timed("k_r4u", [&] { hipLaunchKernelGGL(k_r4u, grid, block, 0, 0, p, n, out); });
    timed("k_r16", [&] { hipLaunchKernelGGL(k_r16, grid, block, 0, 0, p, n, out); });
    timed("k_r16u", [&] { hipLaunchKernelGGL(k_r16u, grid, block, 0, 0, p, n, out); });
    timed("k_r16k1", [&] { hipLaunchKernelGGL(k_r16k1, grid, block, 0, 0, p, n, out); });
    timed("k_w16", [&] { hipLaunchKernelGGL(k_w16, grid, block, 0, 0, p, n); });
    timed("k_w16u", [&] { hipLaunchKernelGGL(k_w16u, grid, block, 0, 0, p, n); });
    timed("k_w1", [&] { hipLaunchKernelGGL(k_w1, grid, block, 0, 0, p, n); });
    CHECK(hipDeviceSynchronize());
    return 0;
}
