/* Drop-in executable: same argv, exit codes and output files as the reference's HS_call_variants; a thin host over the C ABI.
 *
 * How the process ends. Destroying the parsed inputs and results, the HIP runtime, and -- on the kernel side, after _exit --
 * a 4 GB address space with pinned staging buffers and the device state takes 0.4-0.6 s on the 500-contig job: a quarter of the
 * stage. None of it is of interest to the caller, who waits for the exit status (hairsplitter.py:670-679). So: (1) nothing is
 * destroyed in user space (hs_main_process_exits), (2) the work runs in a child forked BEFORE anything touches the GPU; when
 * its output files are complete it reports its status through a pipe and the parent exits with it at once, while the child's
 * teardown finishes in the background. A child that dies without reporting is waited for and its status passed on.
 * HS_NO_DETACH=1 runs everything in one process. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/wait.h>
#include "../../include/hairsplitter_hip.h"
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static int run(int argc, char** argv) {
    const double t0 = now_ms();
    hs_main_process_exits(1);
    const int rc = hs_call_variants_main(argc, argv);
    fflush(NULL);
    if (getenv("HS_TIMING")) fprintf(stderr, "[hs timing] main: entry to exit %.1f ms\n", now_ms() - t0);
    return rc;
}
int main(int argc, char** argv) {
    int pfd[2];
    if (getenv("HS_NO_DETACH") || pipe(pfd) != 0) _exit(run(argc, argv));
    const pid_t pid = fork();
    if (pid < 0) _exit(run(argc, argv));
    if (pid == 0) {
        close(pfd[0]);
        const int rc = run(argc, argv);
        if (write(pfd[1], &rc, sizeof rc) != (ssize_t)sizeof rc) _exit(rc ? rc : 1);
        close(pfd[1]);
        close(0); close(1); close(2);      /* whoever reads this program's output sees its end now */
        _exit(rc);
    }
    close(pfd[1]);
    int rc = 1;
    if (read(pfd[0], &rc, sizeof rc) == (ssize_t)sizeof rc) _exit(rc);
    int st = 0;
    if (waitpid(pid, &st, 0) == pid && WIFEXITED(st)) _exit(WEXITSTATUS(st));
    _exit(1);
}
