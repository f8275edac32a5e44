/* LD_PRELOAD shim for crash hunts: prints the C backtrace of the thread that raised SIGABRT/SIGSEGV
 * (the HIP runtime's aborts carry no Python frames) and exits without writing a core file.
 * build: gcc -shared -fPIC -O1 -o gpurun_out/abort_trace.so tools/probe/abort_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static int g_err = 2;   /* the stderr the process started with (pytest redirects fd 2 while a test runs) */

static void handler(int sig) {
  void* frames[64];
  char msg[128];
  int n = snprintf(msg, sizeof msg, "\n=== abort_trace: signal %d in tid %ld ===\n", sig, (long)syscall(SYS_gettid));
  (void)!write(g_err, msg, n);
  n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, g_err);
  (void)!write(g_err, "=== end ===\n", 12);
  _exit(128 + sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  int fd = dup(2);
  if (fd >= 0) g_err = fd;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sigaction(SIGABRT, &sa, NULL);
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
}
