// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/wait_probe tools/probes/wait_probe.hip ; run: tools/probes/wait_probe <0 stream sync / 1 blocking event / 2 device blocking schedule>
// How much CPU does a host thread burn while it waits for the device? (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <chrono>
__global__ void spin(long long cycles, int* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
static double cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    if (mode == 2) hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
    int* d; hipMalloc(&d, 4);
    for (int rep = 0; rep < 6; ++rep) {
        const long long cyc = 100000LL * 5 * (rep < 3 ? 1 : 4);   // wall_clock64 ticks at 100 MHz: 5 ms / 20 ms
        const double c0 = cpu_ms(), t0 = now_ms();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, d);
        if (mode == 0) hipStreamSynchronize(s);
        else { hipEventRecord(ev, s); hipEventSynchronize(ev); }
        std::printf("mode %d rep %d: wall %.2f ms, thread cpu %.2f ms\n", mode, rep, now_ms() - t0, cpu_ms() - c0);
    }
    return 0;
}
