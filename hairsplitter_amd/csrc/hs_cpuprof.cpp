// hs_cpuprof.cpp -- opt-in sampling profiler of the host side (HS_CPU_PROFILE=<file>): SIGPROF on process CPU time, the
// interrupted program counter of whichever thread was running is recorded; at exit the samples are written as
// "<module> <offset> <count>" lines (tools/cpuprof_report.py resolves them with addr2line). Diagnostic only.
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <sys/time.h>
#include <ucontext.h>

namespace {
constexpr size_t kCap = 1 << 20;
void* g_pc[kCap];
std::atomic<size_t> g_n{0};
std::string g_out;

void on_prof(int, siginfo_t*, void* uc) {
    const size_t i = g_n.fetch_add(1, std::memory_order_relaxed);
    if (i < kCap) g_pc[i] = (void*)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
}

void dump() {
    struct itimerval off;
    std::memset(&off, 0, sizeof off);
    setitimer(ITIMER_PROF, &off, nullptr);
    const size_t n = std::min(g_n.load(), kCap);
    std::map<std::pair<std::string, size_t>, size_t> hist;
    for (size_t i = 0; i < n; ++i) {
        Dl_info di;
        if (dladdr(g_pc[i], &di) && di.dli_fname) hist[{di.dli_fname, (size_t)((char*)g_pc[i] - (char*)di.dli_fbase)}]++;
        else hist[{"?", (size_t)g_pc[i]}]++;
    }
    if (FILE* f = std::fopen(g_out.c_str(), "w")) {
        std::fprintf(f, "# samples %zu (1 ms of process CPU each)\n", n);
        for (auto& kv : hist) std::fprintf(f, "%s %zx %zu\n", kv.first.first.c_str(), kv.first.second, kv.second);
        std::fclose(f);
    }
}

}  // namespace

// diagnostic entry points (not part of include/hairsplitter_hip.h): bracket the region of interest
extern "C" void hs_cpuprof_start(const char* out_file) {
    g_out = out_file ? out_file : "cpu_prof.txt";
    g_n.store(0);
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    struct itimerval it;
    it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 2000;
    it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, nullptr);
}
extern "C" void hs_cpuprof_stop(void) { dump(); }
