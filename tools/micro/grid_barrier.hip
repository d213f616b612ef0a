// What would a persistent per-iteration BiCGStab kernel pay per phase boundary?  ("measure the barrier
// alone first".)  K phases of a streaming pass y <- a*y + x over a vector sized like the 8-GPU share of the bench grid
// (1.25M rows), run (A) as K dependent launches and (B) as ONE persistent launch whose phases are separated by an
// XCD-hierarchical grid barrier: per-XCC arrival counter -> the XCD's last arriver does ONE agent-scope release and arrives at the
// top counter -> the chip's last arriver bumps the generation word; everybody polls the generation with relaxed sc1 loads and
// does one agent-scope acquire (MI355X_MICROARCH.md: barrier-xcd; the vector written in phase p is read by OTHER workgroups in
// phase p+1 -- shifted index -- so the release / acquire pair is not optional).  Prints us per phase for both forms and the
// difference = what a barrier costs more (or less) than a kernel boundary.
// Build: hipcc --offload-arch=gfx950 -O2 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier [rows] [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
struct Bar { unsigned xcc[8 * 16]; unsigned top; unsigned pad0[15]; unsigned gen; unsigned pad1[15]; unsigned flat; unsigned pad2[15]; unsigned cen[8]; unsigned failed; };  // one counter per 64-byte line
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }
// every spin is bounded: a barrier that cannot complete sets B->failed and lets the kernel run out instead of hanging the GPU
#define SPIN_LIMIT 20000000u
// nper[x]: workgroups of THIS launch on XCD x (in-kernel census); every workgroup calls this with all threads
__device__ void grid_barrier(Bar *B, const unsigned *nper, unsigned nxcd_used, unsigned &gen_seen) {
  if (nper[0] == 0xffffffffu) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    const unsigned target = gen_seen + 1;
    if (__hip_atomic_fetch_add(&B->xcc[x * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nper[x] - 1u) {
      __hip_atomic_store(&B->xcc[x * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this XCD's dirty lines
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (__hip_atomic_fetch_add(&B->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd_used - 1u) {
        __hip_atomic_store(&B->top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&B->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned spins = 0;
    while (__hip_atomic_load(&B->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT || __hip_atomic_load(&B->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(&B->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    gen_seen = target;
  }
  __syncthreads();
}
__device__ __forceinline__ void phase(double *y, const double *x, long n, int p) {
  // reads what ANOTHER workgroup wrote in the previous phase (index shifted by a quarter of the vector)
  const long shift = (n / 4) * (p & 3);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long j = i + shift; if (j >= n) j -= n;
    y[(p & 1) * n + i] = 0.5 * y[((p + 1) & 1) * n + j] + x[i];
  }
}
__global__ __launch_bounds__(256) void phase_kernel(double *y, const double *x, long n, int p) { phase(y, x, n, p); }
__global__ __launch_bounds__(256) void persistent_kernel(double *y, const double *x, long n, int K, Bar *B, const unsigned *nper_unused, unsigned nx_unused, int with_work) {
  const unsigned *nper; unsigned nx;
  // census of THIS launch (placement is not a contract and may differ from launch to launch), closed by a flat counter barrier
  __shared__ unsigned cen[8];
  __shared__ unsigned nxs;
  unsigned gen = 0;
  if (threadIdx.x == 0) {
    gen = __hip_atomic_load(&B->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&B->cen[xcc_id()], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&B->flat, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&B->flat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
    if (spins >= SPIN_LIMIT) __hip_atomic_store(&B->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned nx0 = 0;
    for (int i = 0; i < 8; ++i) { cen[i] = __hip_atomic_load(&B->cen[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); nx0 += cen[i] != 0; }
    nxs = nx0;
  }
  __syncthreads();
  nper = cen; nx = nxs;
  for (int p = 0; p < K; ++p) {
    if (__hip_atomic_load(&B->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (with_work) phase(y, x, n, p);
    grid_barrier(B, nper, nx, gen);
  }
}
__global__ void census_kernel(unsigned *cnt) { if (threadIdx.x == 0) atomicAdd(&cnt[xcc_id()], 1u); }
int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1253160;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *y, *x; CK(hipMalloc(&y, 2 * n * 8)); CK(hipMalloc(&x, n * 8));
  CK(hipMemset(y, 0, 2 * n * 8)); CK(hipMemset(x, 0, n * 8));
  Bar *B; CK(hipMalloc(&B, sizeof(Bar))); CK(hipMemset(B, 0, sizeof(Bar)));
  unsigned *nper; CK(hipMalloc(&nper, 8 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 400;
  for (int wgs : {256, 512, 1024}) {
    if (argc > 2 && atoi(argv[2]) != wgs) continue;
    // census: where do the workgroups of a grid of this size run?  (placement is not a contract: counted, not assumed; a grid that
    // is not fully resident would deadlock the barrier -- 1024 x 256 threads = 4 workgroups per CU is resident)
    CK(hipMemset(nper, 0, 32));
    hipLaunchKernelGGL(census_kernel, dim3(wgs), dim3(256), 0, st, nper);
    unsigned h[8]; CK(hipMemcpyAsync(h, nper, 32, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    unsigned nx = 0; for (unsigned v : h) nx += v != 0;
    float msA = 0, msB = 0, msC = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int p = 0; p < K; ++p) hipLaunchKernelGGL(phase_kernel, dim3(wgs), dim3(256), 0, st, y, x, n, p);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&msA, e0, e1));
      CK(hipMemsetAsync(B, 0, sizeof(Bar), st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent_kernel, dim3(wgs), dim3(256), 0, st, y, x, n, K, B, nper, nx, 1);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&msB, e0, e1));
      CK(hipMemsetAsync(B, 0, sizeof(Bar), st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent_kernel, dim3(wgs), dim3(256), 0, st, y, x, n, K, B, nper, nx, 0);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&msC, e0, e1));
    }
    Bar hb; CK(hipMemcpy(&hb, B, sizeof(Bar), hipMemcpyDeviceToHost));
    if (hb.failed) printf("!! the barrier did not complete (spin limit): numbers below are meaningless\n");
    printf("rows %ld, %4d workgroups (XCDs used %u, per XCD %u..): %d phases as launches %.2f us/phase | persistent + grid barrier %.2f us/phase | barrier alone %.2f us | barrier - boundary = %+.2f us\n",
           n, wgs, nx, h[0], K, msA * 1e3 / K, msB * 1e3 / K, msC * 1e3 / K, (msB - msA) * 1e3 / K);
  }
  return 0;
}
