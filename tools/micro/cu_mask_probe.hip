// Does this box honour CU masks on HIP streams?  (the multi-rank Krylov loop lets every wavefront of a
// chip-filling kernel wait for a peer rank; on the one-GPU test box that is only testable if each rank's stream owns a disjoint
// set of CUs -- hipExtStreamCreateWithCUMask.)
//   1. census: a 4096-workgroup kernel on a stream masked to the lower / upper half of the CU bits records (XCC, SE, SH, CU) of
//      every workgroup -> distinct CUs per stream, CUs per XCC, overlap of the two sets;
//   2. starvation test: stream A is filled with wavefronts that spin (bounded) on a flag only a kernel on stream B raises.  With
//      disjoint masks B runs beside A and the flag arrives within microseconds; if the masks are ignored B waits for a slot
//      until A's spins run out.
// Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip ; run: ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }
__device__ __forceinline__ unsigned hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__global__ void census_kernel(unsigned *out) {
  if (threadIdx.x == 0) {
    const unsigned h = hw_id();  // gfx9: [11:8] CU_ID, [12] SH_ID, [15:13] SE_ID
    out[blockIdx.x] = (xcc_id() << 16) | (h & 0xff00u);
  }
  // stay a little so that the launch spreads over every CU it may use
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000ull) __builtin_amdgcn_s_sleep(8);  // 20 us
}
__global__ __launch_bounds__(64) void spin_kernel(const unsigned *flag, unsigned long long limit_ticks, unsigned long long *waited, unsigned *timed_out) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long dt = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    dt = wall_clock64() - t0;
    if (dt > limit_ticks) { if (threadIdx.x == 0) atomicAdd(timed_out, 1u); break; }
    __builtin_amdgcn_s_sleep(4);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *waited = wall_clock64() - t0;
}
__global__ void raise_kernel(unsigned *flag) { if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static int census(hipStream_t s, std::set<unsigned> &cus, int per_xcc[16]) {
  const int nb = 4096;
  unsigned *d;
  CK(hipMalloc(&d, nb * sizeof(unsigned)));
  hipLaunchKernelGGL(census_kernel, dim3(nb), dim3(64), 0, s, d);
  std::vector<unsigned> h(nb);
  CK(hipMemcpyAsync(h.data(), d, nb * sizeof(unsigned), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  for (unsigned v : h) cus.insert(v);
  for (int i = 0; i < 16; ++i) per_xcc[i] = 0;
  for (unsigned v : cus) per_xcc[(v >> 16) & 15]++;
  CK(hipFree(d));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.name, ncu);
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> lo(words, 0), hi(words, 0);
  for (int i = 0; i < ncu; ++i) (i < ncu / 2 ? lo : hi)[i / 32] |= 1u << (i % 32);
  hipStream_t sa, sb, sf;
  hipError_t e = hipExtStreamCreateWithCUMask(&sa, words, lo.data());
  if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\nCU_MASK_HONOURED 0\n", hipGetErrorString(e)); return 0; }
  CK(hipExtStreamCreateWithCUMask(&sb, words, hi.data()));
  CK(hipStreamCreateWithFlags(&sf, hipStreamNonBlocking));
  std::set<unsigned> ca, cb, cf;
  int pa[16], pb[16], pf[16];
  if (census(sf, cf, pf) || census(sa, ca, pa) || census(sb, cb, pb)) return 2;
  int overlap = 0;
  for (unsigned v : ca) overlap += (int)cb.count(v);
  printf("census: unmasked stream %zu CUs, lower-half mask %zu CUs, upper-half mask %zu CUs, overlap %d\n", cf.size(), ca.size(), cb.size(), overlap);
  printf("CUs per XCC  unmasked:"); for (int i = 0; i < 8; ++i) printf(" %d", pf[i]);
  printf("  lower:"); for (int i = 0; i < 8; ++i) printf(" %d", pa[i]);
  printf("  upper:"); for (int i = 0; i < 8; ++i) printf(" %d", pb[i]);
  printf("\n");
  // starvation test
  unsigned *flag, *tmo;
  unsigned long long *waited;
  CK(hipMalloc(&flag, 4)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&waited, 8));
  for (int masked = 1; masked >= 0; --masked) {
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(tmo, 0, 4)); CK(hipMemset(waited, 0, 8));
    CK(hipDeviceSynchronize());
    hipStream_t A = masked ? sa : sf, B = masked ? sb : sf;
    hipStream_t B2 = B;
    if (!masked) CK(hipStreamCreateWithFlags(&B2, hipStreamNonBlocking));
    // 65536 one-wavefront workgroups: far more than the chip holds; bounded spin of 0.5 s (100 MHz ticks)
    hipLaunchKernelGGL(spin_kernel, dim3(65536), dim3(64), 0, A, flag, 50000000ull, waited, tmo);
    hipLaunchKernelGGL(raise_kernel, dim3(1), dim3(64), 0, B2, flag);
    CK(hipDeviceSynchronize());
    unsigned h_tmo = 0; unsigned long long h_w = 0;
    CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_w, waited, 8, hipMemcpyDeviceToHost));
    printf("starvation test (%s): workgroup 0 waited %.1f us, %u workgroups ran out of time\n", masked ? "disjoint CU masks" : "no masks, two streams", h_w / 100.0, h_tmo);
    if (masked) printf("CU_MASK_HONOURED %d\n", (overlap == 0 && ca.size() < cf.size() && h_tmo == 0) ? 1 : 0);
  }
  return 0;
}
