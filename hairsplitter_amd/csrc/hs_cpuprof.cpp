// hs_cpuprof.cpp -- opt-in sampling profiler of the host side (HS_CPU_PROFILE=<file>): SIGPROF on process CPU time, the
// interrupted program counter of whichever thread was running is recorded; at exit the samples are written as
// "<module> <offset> <via> <count>" lines (tools/cpuprof_report.py resolves them with addr2line); <via> = the innermost frame
// of the interrupted stack that lies in this library, i.e. the function of ours that called into libc / the HIP runtime.
// Diagnostic only.
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <link.h>
#include <map>
#include <string>
#include <sys/time.h>
#include <ucontext.h>

namespace {
constexpr size_t kCap = 1 << 20;
void* g_pc[kCap];
void* g_via[kCap];
unsigned long g_rsi[kCap];       // second syscall argument at the sample: the request of an ioctl
uintptr_t g_lo = 0, g_hi = 0, g_base = 0;     // executable segment of this library
int (*g_unw_init)(void*, void*) = nullptr;    // libunwind.so.8 of the image, loaded by name (no headers here)
int (*g_unw_step)(void*) = nullptr;
int (*g_unw_reg)(void*, int, unsigned long*) = nullptr;
std::atomic<size_t> g_n{0};
std::string g_out;

void on_prof(int, siginfo_t*, void* uc) {
    const size_t i = g_n.fetch_add(1, std::memory_order_relaxed);
    if (i >= kCap) return;
    const uintptr_t pc = (uintptr_t)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
    g_pc[i] = (void*)pc;
    g_rsi[i] = (unsigned long)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RSI];
    uintptr_t via = 0;
    if (pc >= g_lo && pc < g_hi) via = pc;
    else if (g_unw_init) {      // unwind from the interrupted context (libunwind, local) to the first frame of this library
        alignas(16) unsigned long cursor[160];                    // unw_cursor_t is 127 words on x86-64
        if (g_unw_init(cursor, uc) == 0) {
            for (int k = 0; k < 48 && g_unw_step(cursor) > 0; ++k) {
                unsigned long ip = 0;
                if (g_unw_reg(cursor, 16 /* UNW_X86_64_RIP */, &ip) != 0) break;
                if (ip >= g_lo && ip < g_hi) { via = ip; break; }
            }
        }
    }
    g_via[i] = (void*)via;
}

int find_self(struct dl_phdr_info* info, size_t, void* self) {
    for (int k = 0; k < info->dlpi_phnum; ++k) {
        const ElfW(Phdr)& ph = info->dlpi_phdr[k];
        if (ph.p_type != PT_LOAD || !(ph.p_flags & PF_X)) continue;
        const uintptr_t lo = info->dlpi_addr + ph.p_vaddr, hi = lo + ph.p_memsz;
        if ((uintptr_t)self >= lo && (uintptr_t)self < hi) { g_lo = lo; g_hi = hi; g_base = info->dlpi_addr; return 1; }
    }
    return 0;
}

void dump() {
    struct itimerval off;
    std::memset(&off, 0, sizeof off);
    setitimer(ITIMER_PROF, &off, nullptr);
    const size_t n = std::min(g_n.load(), kCap);
    std::map<std::pair<std::pair<std::string, size_t>, size_t>, size_t> hist;
    for (size_t i = 0; i < n; ++i) {
        Dl_info di;
        const size_t via = g_via[i] ? (size_t)((uintptr_t)g_via[i] - g_base) : 0;
        if (dladdr(g_pc[i], &di) && di.dli_fname) hist[{{di.dli_fname, (size_t)((char*)g_pc[i] - (char*)di.dli_fbase)}, via}]++;
        else hist[{{"?", (size_t)g_pc[i]}, via}]++;
    }
    if (FILE* f = std::fopen(g_out.c_str(), "w")) {
        std::fprintf(f, "# samples %zu (1 ms of process CPU each)\n", n);
        Dl_info self;
        std::fprintf(f, "# self %s\n", dladdr((void*)&on_prof, &self) && self.dli_fname ? self.dli_fname : "?");
        {   // samples inside libc's ioctl(): by request code
            std::map<unsigned long, size_t> io;
            for (size_t i = 0; i < n; ++i) {
                Dl_info di;
                if (dladdr(g_pc[i], &di) && di.dli_sname && std::strcmp(di.dli_sname, "ioctl") == 0) io[g_rsi[i]]++;
            }
            for (auto& kv : io) std::fprintf(f, "# ioctl request 0x%lx: %zu samples\n", kv.first, kv.second);
        }
        for (auto& kv : hist) std::fprintf(f, "%s %zx %zx %zu\n", kv.first.first.first.c_str(), kv.first.first.second, kv.first.second, kv.second);
        std::fclose(f);
    }
}

}  // namespace

// diagnostic entry points (not part of include/hairsplitter_hip.h): bracket the region of interest
extern "C" void hs_cpuprof_start(const char* out_file) {
    g_out = out_file ? out_file : "cpu_prof.txt";
    g_n.store(0);
    dl_iterate_phdr(find_self, (void*)&on_prof);
    if (void* lu = dlopen("libunwind.so.8", RTLD_NOW | RTLD_GLOBAL)) {
        g_unw_init = (int (*)(void*, void*))dlsym(lu, "_ULx86_64_init_local");
        g_unw_step = (int (*)(void*))dlsym(lu, "_ULx86_64_step");
        g_unw_reg = (int (*)(void*, int, unsigned long*))dlsym(lu, "_ULx86_64_get_reg");
        if (!g_unw_step || !g_unw_reg) g_unw_init = nullptr;
    }
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
