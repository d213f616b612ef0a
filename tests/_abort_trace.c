/* Test infrastructure (loaded by tests/conftest.py, never by the product): on SIGABRT / SIGSEGV / SIGBUS write the native
 * backtrace of the thread that received the signal to a file -- pytest captures stderr per test and Python's faulthandler shows
 * Python frames only, so an abort() inside a runtime library otherwise leaves no trace of who called it. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static int trace_fd = -1;
static void on_signal(int sig) {
  void *frames[96];
  const int n = backtrace(frames, 96);
  char head[64];
  const int len = snprintf(head, sizeof head, "\n==== signal %d, native backtrace ====\n", sig);
  const int fds[2] = {trace_fd, 2};
  for (int i = 0; i < 2; ++i) {
    if (fds[i] < 0) continue;
    if (write(fds[i], head, (size_t)len) < 0) continue;
    backtrace_symbols_fd(frames, n, fds[i]);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
int abort_trace_install(const char *path) {
  (void)backtrace((void *[1]){0}, 1); /* loads libgcc now, not inside the handler */
  trace_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_signal;
  sigaction(SIGABRT, &sa, NULL);
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
  return trace_fd >= 0 ? 0 : -1;
}
