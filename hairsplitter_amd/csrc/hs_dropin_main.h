/* hs_dropin_main.h -- how the two drop-in executables start and end (shared by HS_call_variants.c and HS_separate_reads.c).
 *
 * Destroying the parsed inputs and results, the HIP runtime, and -- on the kernel side, after _exit -- a 4 GB address space with
 * pinned staging buffers and the device state takes 0.4-0.6 s on the 500-contig job: a quarter of the stage. None of it is of
 * interest to the caller, who waits for the exit status (hairsplitter.py:670-679). So: (1) nothing is destroyed in user space
 * (hs_main_process_exits), (2) the work runs in a child forked BEFORE anything touches the GPU; when its output files are
 * complete it reports its status through a pipe and the parent exits with it at once, while the child's teardown finishes in
 * the background. A child that dies without reporting is waited for and its status passed on.
 *   - Until it has reported, the child dies with its parent (PR_SET_PDEATHSIG) and the parent passes SIGTERM / SIGINT / SIGHUP /
 *     SIGQUIT on to it: a caller that kills the process it started (a timeout) leaves nothing behind that still holds the GPU
 *     or writes the output files. After the report only the teardown is left and the tie is cut.
 *   - HS_NO_DETACH=1 runs everything in the process that was started. The same happens by itself when the process holds a descriptor
 *     of the GPU when main() begins (/dev/kfd or a render node among /proc/self/fd: somebody -- rocprofv3 and friends, under whatever
 *     name -- has initialised the GPU before main(), and neither a fork nor an exec after that is safe), and, as a second line, when a
 *     known tool is named in the environment (ROCP_TOOL_LIBRARIES, HSA_TOOLS_LIB, an LD_PRELOAD with rocprof / roctracer / the HSA or
 *     HIP runtime): profile the drop-ins as they are, no switch needed. */
#ifndef HS_DROPIN_MAIN_H
#define HS_DROPIN_MAIN_H
#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/prctl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include "../../include/hairsplitter_hip.h"
void hs_teardown_probe(void);      /* (diagnostics, not part of the C ABI header) */
void hs_dropin_finish(void);
void hs_call_variants_epilogue(void);
void hs_cpuprof_start(const char* out_file);
void hs_cpuprof_stop(void);

static double hs_dropin_now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void hs_dropin_stamp(const char* what) {      /* HS_TIMING: wall-clock stamps of the process (epoch ms), to be set against the caller's own */
    if (!getenv("HS_TIMING")) return;
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    fprintf(stderr, "[hs timing] stamp %s pid %d at %.1f ms\n", what, (int)getpid(), ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6);
}
static int (*hs_dropin_stage)(int, char**);
static void (*hs_dropin_epilogue)(void);      /* work the stage may still do once its outputs are complete (HS_call_variants: the precomputed .gro); may be NULL */
static int hs_dropin_run(int argc, char** argv) {
    const double t0 = hs_dropin_now_ms();
    hs_main_process_exits(1);
    if (getenv("HS_CPU_PROFILE")) hs_cpuprof_start(getenv("HS_CPU_PROFILE"));      /* sampling profile of the host side (tools/cpuprof_report.py) */
    const int rc = hs_dropin_stage(argc, argv);
    if (getenv("HS_CPU_PROFILE")) hs_cpuprof_stop();
    fflush(NULL);
    if (getenv("HS_EXIT_PROBE")) { const double tp = hs_dropin_now_ms(); hs_teardown_probe(); fprintf(stderr, "[hs exit probe] explicit teardown %.1f ms\n", hs_dropin_now_ms() - tp); }
    if (getenv("HS_TIMING")) fprintf(stderr, "[hs timing] main: entry to exit %.1f ms\n", hs_dropin_now_ms() - t0);
    return rc;
}
static volatile pid_t hs_dropin_child = 0;
static void hs_dropin_forward(int sig) { if (hs_dropin_child > 0) kill(hs_dropin_child, sig); }
static int hs_dropin_tool_preloaded(void) {      // a profiler / tracer of the ROCm stack in the process (they initialise the GPU before main())
    const char* names[] = {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCTRACER_DOMAIN"};
    for (unsigned i = 0; i < sizeof names / sizeof names[0]; ++i) { const char* v = getenv(names[i]); if (v && v[0]) return 1; }
    const char* pre = getenv("LD_PRELOAD");      // (other preloads -- sanitizers, exec guards -- do not touch the GPU)
    if (pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "rocprofiler") || strstr(pre, "libhsa") || strstr(pre, "amdhip"))) return 1;
    return 0;
}
/* The positive test: does this process hold a descriptor of the GPU (the compute node /dev/kfd or a render node /dev/dri/renderD*)? Whoever
 * initialised the GPU before main() -- under whatever name -- has opened them. 1: yes, or it cannot be told (no /proc/self/fd): no fork and
 * no exec then; 0: none. */
static int hs_dropin_gpu_is_open(void) {
    DIR* d = opendir("/proc/self/fd");
    if (!d) return 1;
    int found = 0;
    struct dirent* e;
    while (!found && (e = readdir(d)) != NULL) {
        if (e->d_name[0] == '.') continue;
        char path[64], target[256];
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (strncmp(target, "/dev/kfd", 8) == 0 || strncmp(target, "/dev/dri/", 9) == 0) found = 1;
    }
    closedir(d);
    return found;
}
static int hs_dropin_run_all(int argc, char** argv) {      /* one process: the stage, then its epilogue, then the exit */
    const int rc = hs_dropin_run(argc, argv);
    if (rc == 0 && hs_dropin_epilogue) { hs_dropin_epilogue(); fflush(NULL); }
    hs_dropin_finish();
    hs_dropin_stamp("leaving");
    return rc;
}
/* The stage parses gigabytes into freshly mapped memory and leaves them to the process end: with 4-KB pages that is a million page faults on the way in and
 * a million pages to give back on the way out (0.15 s of the 500-contig job's 1.35 s). glibc (>= 2.35) asks for transparent huge pages for what malloc maps
 * when it is STARTED with GLIBC_TUNABLES=glibc.malloc.hugetlb=1 -- so the executable starts itself again with that setting, once, before anything of it
 * has touched the GPU (checked, not assumed: hs_dropin_gpu_is_open; not when the caller set GLIBC_TUNABLES itself). HS_NO_REEXEC=1: never. This call stays above every hs_* call. */
static void hs_dropin_with_huge_pages(char** argv, int wanted) {
    if (!wanted || getenv("HS_NO_REEXEC") || getenv("GLIBC_TUNABLES") || hs_dropin_tool_preloaded() || hs_dropin_gpu_is_open()) return;
    setenv("HS_NO_REEXEC", "1", 1);
    setenv("GLIBC_TUNABLES", "glibc.malloc.hugetlb=1", 1);
    execv("/proc/self/exe", argv);
    unsetenv("GLIBC_TUNABLES");      /* (no /proc, or the exec was refused: on with the process as it is) */
}
static int hs_dropin_main2(int (*stage)(int, char**), void (*epilogue)(void), int argc, char** argv) {
    int pfd[2];
    hs_dropin_stage = stage;
    hs_dropin_epilogue = epilogue;
    hs_dropin_with_huge_pages(argv, epilogue != NULL);      /* (HS_call_variants: the stage that parses the job's text; HS_separate_reads maps the arrays it left: the restart would only cost it 30 ms) */
    hs_dropin_stamp("main entered");
    if (getenv("HS_NO_DETACH") || hs_dropin_tool_preloaded() || hs_dropin_gpu_is_open() || pipe(pfd) != 0) _exit(hs_dropin_run_all(argc, argv));
    const pid_t parent = getpid();
    const pid_t pid = fork();
    if (pid < 0) _exit(hs_dropin_run_all(argc, argv));
    if (pid == 0) {
        close(pfd[0]);
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        if (getppid() != parent) _exit(1);      /* the parent went away between fork and prctl */
        hs_dropin_stamp("worker started");
        const int rc = hs_dropin_run(argc, argv);
        hs_dropin_stamp("worker done");
        prctl(PR_SET_PDEATHSIG, 0);             /* the outputs are complete: the parent is about to leave, the teardown goes on */
        if (write(pfd[1], &rc, sizeof rc) != (ssize_t)sizeof rc) _exit(rc ? rc : 1);
        close(pfd[1]);
        {   /* whoever reads this program's output sees its end now; the descriptors stay taken (by /dev/null), so that nothing the epilogue
             * opens -- the companion file, a mapping, a device node -- becomes "standard output" */
            const int nul = open("/dev/null", O_RDWR);
            if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); if (nul > 2) close(nul); } else { close(0); close(1); }
        }
        if (rc == 0 && hs_dropin_epilogue) hs_dropin_epilogue();      /* (the caller has its exit status; this runs beside whatever it starts next) */
        hs_dropin_finish();
        { const int nul = open("/dev/null", O_RDWR); if (nul >= 0) { dup2(nul, 2); if (nul > 2) close(nul); } else close(2); }
        _exit(rc);
    }
    close(pfd[1]);
    hs_dropin_child = pid;
    {
        struct sigaction sa;
        sa.sa_handler = hs_dropin_forward; sigemptyset(&sa.sa_mask); sa.sa_flags = SA_RESTART;
        sigaction(SIGTERM, &sa, NULL); sigaction(SIGINT, &sa, NULL); sigaction(SIGHUP, &sa, NULL); sigaction(SIGQUIT, &sa, NULL);
    }
    int rc = 1;
    if (read(pfd[0], &rc, sizeof rc) == (ssize_t)sizeof rc) { hs_dropin_stamp("status received"); _exit(rc); }
    int st = 0;
    if (waitpid(pid, &st, 0) == pid) {
        if (WIFEXITED(st)) _exit(WEXITSTATUS(st));
        if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); kill(getpid(), WTERMSIG(st)); }      /* end the way the child ended */
    }
    _exit(1);
}
static int hs_dropin_main(int (*stage)(int, char**), int argc, char** argv) { return hs_dropin_main2(stage, NULL, argc, argv); }
#endif
