// Micro-benchmark: cost of back-to-back dependent launches on one stream (host enqueue rate vs device dispatch gap).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel(double *p) { if (p && threadIdx.x == 1024) p[0] = 1.0; }
__global__ void small_kernel(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  double *d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1, ev[64]; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto &e : ev) hipEventCreate(&e);
  const int N = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    // (a) empty kernels
    hipStreamSynchronize(st);
    double t0 = now(); hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, d);
    double t1 = now(); hipEventRecord(e1, st); hipStreamSynchronize(st); double t2 = now();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty 1x64      : host enqueue %.2f us/launch, device %.2f us/launch, wall %.2f us\n", (t1 - t0) / N * 1e6, ms * 1e3 / N, (t2 - t0) / N * 1e6);
    // (b) 2048 x 256 small kernels (like the persistent SpMV grid)
    t0 = now(); hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(small_kernel, dim3(2048), dim3(256), 0, st, d, 2048 * 256);
    t1 = now(); hipEventRecord(e1, st); hipStreamSynchronize(st); t2 = now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("small 2048x256  : host enqueue %.2f us/launch, device %.2f us/launch, wall %.2f us\n", (t1 - t0) / N * 1e6, ms * 1e3 / N, (t2 - t0) / N * 1e6);
    // (c) with an event record after every launch
    t0 = now(); hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(small_kernel, dim3(2048), dim3(256), 0, st, d, 2048 * 256); hipEventRecord(ev[i & 63], st); }
    t1 = now(); hipEventRecord(e1, st); hipStreamSynchronize(st); t2 = now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("small + event   : host enqueue %.2f us/launch, device %.2f us/launch, wall %.2f us\n", (t1 - t0) / N * 1e6, ms * 1e3 / N, (t2 - t0) / N * 1e6);
    // (d) 20000 x 64 (like the ILU apply grid)
    t0 = now(); hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(small_kernel, dim3(20000), dim3(64), 0, st, d, 20000 * 64);
    t1 = now(); hipEventRecord(e1, st); hipStreamSynchronize(st); t2 = now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("small 20000x64  : host enqueue %.2f us/launch, device %.2f us/launch, wall %.2f us\n", (t1 - t0) / N * 1e6, ms * 1e3 / N, (t2 - t0) / N * 1e6);
    // (e) graph of 9 kernels replayed
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 9; ++i) hipLaunchKernelGGL(small_kernel, dim3(2048), dim3(256), 0, st, d, 2048 * 256);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    t0 = now(); hipEventRecord(e0, st);
    for (int i = 0; i < N / 9; ++i) hipGraphLaunch(ge, st);
    t1 = now(); hipEventRecord(e1, st); hipStreamSynchronize(st); t2 = now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 9      : host enqueue %.2f us/kernel, device %.2f us/kernel, wall %.2f us\n", (t1 - t0) / (N / 9 * 9) * 1e6, ms * 1e3 / (N / 9 * 9), (t2 - t0) / (N / 9 * 9) * 1e6);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
