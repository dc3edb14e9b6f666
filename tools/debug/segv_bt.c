/* LD_PRELOAD shim: native backtrace on SIGSEGV (the image has no debugger).  Later attempts to replace the SIGSEGV
   handler are swallowed.   gcc -shared -fPIC -o segv_bt.so segv_bt.c -ldl
   LD_PRELOAD=./segv_bt.so python -m pytest -p no:faulthandler ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

static int installed = 0;
typedef int (*sigaction_fn)(int, const struct sigaction*, struct sigaction*);

static void handler(int sig, siginfo_t* si, void* ctx) {
    void* frames[64];
    int n = backtrace(frames, 64);
    int fd = open("gpurun_out/segv_native_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);  /* pytest captures fd 2 */
    if (fd < 0) fd = 2;
    dprintf(fd, "\n==== SIGSEGV at address %p, native backtrace (%d frames) ====\n", si->si_addr, n);
    backtrace_symbols_fd(frames, n, fd);
    _exit(139);
}

int sigaction(int signum, const struct sigaction* act, struct sigaction* old) {
    sigaction_fn real = (sigaction_fn)dlsym(RTLD_NEXT, "sigaction");
    if (signum == SIGSEGV && installed && act) return real(signum, NULL, old);  /* keep ours */
    return real(signum, act, old);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    sa.sa_sigaction = handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO;
    sigaction_fn real = (sigaction_fn)dlsym(RTLD_NEXT, "sigaction");
    real(SIGSEGV, &sa, NULL);
    installed = 1;
}
