// Feasibility probe: two PROCESSES on one GPU share an uncached device buffer through hipIpcMemHandle and hand a flag
// back and forth from kernels (the mechanism of a low-latency scalar all-reduce over xGMI peers).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
#include <unistd.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("[%d] HIP error %s at line %d\n", getpid(), hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

// each side: for it in 1..n: write (it) into the peer's word, wait until my own word == it
__global__ void pingpong(volatile unsigned long long *mine, unsigned long long *peer, int n, unsigned long long *fail) {
  for (int it = 1; it <= n; ++it) {
    __hip_atomic_store(peer, (unsigned long long)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    long spin = 0;
    while (__hip_atomic_load((unsigned long long *)mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)it) {
      if (++spin > 2000000000L) { *fail = it; return; }
    }
  }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  pid_t pid = fork();
  const bool child = pid == 0;
  int rd = child ? p2c[0] : c2p[0], wr = child ? c2p[1] : p2c[1];
  CK(hipSetDevice(0));
  unsigned long long *buf = nullptr, *fail = nullptr;
  CK(hipExtMallocWithFlags((void **)&buf, 4096, hipDeviceMallocUncached));
  CK(hipMemset(buf, 0, 4096));
  CK(hipMalloc((void **)&fail, 8));
  CK(hipMemset(fail, 0, 8));
  hipIpcMemHandle_t mine, theirs;
  CK(hipIpcGetMemHandle(&mine, buf));
  if (write(wr, &mine, sizeof(mine)) != (ssize_t)sizeof(mine)) return 1;
  if (read(rd, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) return 1;
  unsigned long long *peer = nullptr;
  CK(hipIpcOpenMemHandle((void **)&peer, theirs, hipIpcMemLazyEnablePeerAccess));
  char c = 1;  // both sides have opened
  if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 1;
  const int n = 20000;
  double t0 = now();
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, buf, peer, n, fail);
  CK(hipDeviceSynchronize());
  double t1 = now();
  unsigned long long f = 0;
  CK(hipMemcpy(&f, fail, 8, hipMemcpyDeviceToHost));
  printf("[%s] %d handshakes in %.3f ms -> %.2f us each, fail=%llu\n", child ? "child" : "parent", n, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, f);
  if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 1;  // nobody unmaps while the other still runs
  CK(hipIpcCloseMemHandle(peer));
  if (!child) { int st; waitpid(pid, &st, 0); }
  return f ? 3 : 0;
}
