/* LD_PRELOAD sampling profiler for images without perf/gdb: SIGALRM every 200 us of wall time (ITIMER_REAL is a high-resolution timer; ITIMER_PROF ticks with the
 * scheduler, 100-250 Hz), the interrupted PC goes into
 * a buffer; at exit the PCs that fall into the library named by PCSAMPLE_LIB are written as offsets into it (one per line)
 * to PCSAMPLE_OUT, ready for `llvm-addr2line -f -e <lib>` (tools/pcsample_report.py).
 *   gcc -O2 -shared -fPIC -o tools/pcsample.so tools/pcsample.c -ldl
 *   PCSAMPLE_LIB=liblurkhip.so PCSAMPLE_OUT=/tmp/pcs.txt LD_PRELOAD=tools/pcsample.so python ... */
#define _GNU_SOURCE
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>

#define MAX_SAMPLES (1u << 22)
static uintptr_t* g_pc;
static volatile uint32_t g_n;

static void on_prof(int sig, siginfo_t* si, void* ctx) {
    (void)sig, (void)si;
    const ucontext_t* uc = (const ucontext_t*)ctx;
    uint32_t i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
    if (i < MAX_SAMPLES) g_pc[i] = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP];
}

__attribute__((constructor)) static void start(void) {
    if (!getenv("PCSAMPLE_OUT")) return;
    g_pc = (uintptr_t*)calloc(MAX_SAMPLES, sizeof(uintptr_t));
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGALRM, &sa, NULL);
    struct itimerval it = {{0, 200}, {0, 200}};
    setitimer(ITIMER_REAL, &it, NULL);
}

__attribute__((destructor)) static void stop(void) {
    const char* out = getenv("PCSAMPLE_OUT");
    if (!out || !g_pc) return;
    struct itimerval it = {{0, 0}, {0, 0}};
    setitimer(ITIMER_REAL, &it, NULL);
    const char* want = getenv("PCSAMPLE_LIB");
    uintptr_t lo = 0, hi = 0, base = 0;
    FILE* m = fopen("/proc/self/maps", "r");
    char line[1024];
    while (m && fgets(line, sizeof line, m)) {
        if (!want || !strstr(line, want)) continue;
        unsigned long a, b, off;
        if (sscanf(line, "%lx-%lx %*s %lx", &a, &b, &off) != 3) continue;
        if (!lo || a < lo) lo = a;
        if (b > hi) hi = b;
        if (off == 0) base = a;
    }
    if (m) fclose(m);
    FILE* f = fopen(out, "w");
    if (!f) return;
    uint32_t n = g_n < MAX_SAMPLES ? g_n : MAX_SAMPLES, inside = 0;
    for (uint32_t i = 0; i < n; i++)
        if (g_pc[i] >= lo && g_pc[i] < hi) fprintf(f, "0x%lx\n", (unsigned long)(g_pc[i] - base)), inside++;
    fprintf(stderr, "pcsample: %u samples, %u in %s\n", n, inside, want ? want : "(all)");
    fclose(f);
}
