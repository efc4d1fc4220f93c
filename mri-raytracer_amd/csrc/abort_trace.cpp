// mrirt_install_abort_trace: make a process abort self-describing (host code only; opt-in, see include/mrirt.h).
//
// Why it exists: a GPU memory fault, a hardware exception or a queue error is reported by the ROCm runtime as a line on
// stderr followed by abort() — from the runtime's own event thread, while the application's thread sits in some
// synchronising call.  A test runner that redirects file descriptor 2 into a temporary file per test (pytest's default
// capture) loses that line when the process dies, and what is left is a bare "Aborted" at an innocent synchronisation
// point (round 3: one such abort in ten full runs of the GPU suite).  The handler installed here writes,
// to a descriptor the caller duplicated BEFORE any redirection:
//   - the native backtrace of the aborting thread (which library called abort: the HSA event thread, glibc's heap
//     checks, an uncaught C++ exception, ...), and
//   - the tail of whatever file descriptor 2 currently points at when that is a regular file (the runner's capture
//     file: the runtime's own message),
// then hands over to the handler that was installed before it (Python's faulthandler, or the default action).
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/mrirt.h"

namespace {

int g_fd = -1;
struct sigaction g_prev;

void put(const char* s) { (void)!write(g_fd, s, strlen(s)); }

void on_abort(int sig, siginfo_t* info, void* ctx) {
    if (g_fd >= 0) {
        put("\n==== mrirt abort trace: native stack of the aborting thread ====\n");
        void* frames[64];
        const int n = backtrace(frames, 64);
        backtrace_symbols_fd(frames, n, g_fd);
        struct stat st;
        if (fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            put("==== tail of the redirected stderr (fd 2 is a regular file) ====\n");
            static char buf[65536];
            const off_t want = st.st_size < (off_t)sizeof buf ? st.st_size : (off_t)sizeof buf;
            const ssize_t got = pread(2, buf, (size_t)want, st.st_size - want);
            if (got > 0) (void)!write(g_fd, buf, (size_t)got);
            put("\n");
        }
        put("==== end of mrirt abort trace ====\n");
    }
    // hand over: the previous handler (faulthandler dumps the Python stacks and re-raises), or the default action
    if ((g_prev.sa_flags & SA_SIGINFO) && g_prev.sa_sigaction != nullptr) {
        g_prev.sa_sigaction(sig, info, ctx);
    } else if (!(g_prev.sa_flags & SA_SIGINFO) && g_prev.sa_handler != SIG_DFL && g_prev.sa_handler != SIG_IGN && g_prev.sa_handler != nullptr) {
        g_prev.sa_handler(sig);
    }
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

}  // namespace

extern "C" int mrirt_install_abort_trace(int fd) {
    if (fd < 0) return MRIRT_ERR_ARG;
    void* warm[4];
    (void)backtrace(warm, 4);            // the first call loads libgcc's unwinder: not something to do inside a signal handler
    g_fd = fd;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    struct sigaction prev;
    if (sigaction(SIGABRT, &sa, &prev) != 0) return MRIRT_ERR_ARG;
    if (prev.sa_sigaction != on_abort || !(prev.sa_flags & SA_SIGINFO)) g_prev = prev;      // (installing twice keeps the first chain)
    return MRIRT_OK;
}
