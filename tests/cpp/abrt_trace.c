/* Diagnostic for the test sessions ("Robustness record"): when the process receives SIGABRT, write
 * the NATIVE call stack of the thread that raised it to stderr -- python's faulthandler shows the python frames of the
 * threads, which says where the main thread was, not who called abort() -- then hand over to the handler that was installed
 * before (faulthandler's) or to the default action.  Loaded by tests/conftest.py in GPU sessions.
 * gcc -O1 -g -shared -fPIC -o libabrt_trace.so abrt_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static struct sigaction g_prev;
static void (*g_dump)(int fd);      /* libcariboulite_hip.so's clhip_debug_ops_dump (write(2) only): the library's last 256 registrations,
                                     * releases and copies of host memory it does not own -- what a fault address on the host heap is set against */
static int g_fd = -1;       /* a file of our own: under pytest fd 2 is a capture file that dies with the process */

static void put(const char *s) { if (write(2, s, strlen(s)) < 0) { } if (g_fd >= 0 && write(g_fd, s, strlen(s)) < 0) { } }

static void on_abrt(int sig, siginfo_t *info, void *ctx)
{
    void *frames[96];
    char line[96];
    long tid = (long)syscall(SYS_gettid), pid = (long)getpid();
    int n = 0, i = 0;
    /* (no printf in a signal handler) */
    put("\n==== SIGABRT: native stack of the raising thread (tid ");
    { long v = tid; char tmp[24]; int k = 0; if (!v) tmp[k++] = '0'; while (v) { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (k) line[i++] = tmp[--k]; line[i] = 0; put(line); }
    put(tid == pid ? ", the main thread) ====\n" : ", NOT the main thread) ====\n");
    n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    if (g_fd >= 0) { backtrace_symbols_fd(frames, n, g_fd); fsync(g_fd); }
    put("==== end of native stack ====\n");
    if (g_dump) { g_dump(2); if (g_fd >= 0) { g_dump(g_fd); fsync(g_fd); } }
    if (g_prev.sa_flags & SA_SIGINFO) { if (g_prev.sa_sigaction) { g_prev.sa_sigaction(sig, info, ctx); return; } }
    else if (g_prev.sa_handler != SIG_DFL && g_prev.sa_handler != SIG_IGN) { g_prev.sa_handler(sig); return; }
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

void abrt_trace_set_dump(void (*fn)(int)) { g_dump = fn; }

int abrt_trace_install(const char *path)
{
    if (path && *path) g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abrt;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &g_prev);
}
