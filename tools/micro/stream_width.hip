// Micro-benchmark: does the jagged SpMV's access mix stream faster with 16-byte value loads?
//   A: per row (lane) five diagonals, each an 8-byte value + a 2-byte column code (what spmv_jds16_kernel issues)
//   B: the same entries as two (double2, ushort2) pairs + one single (two rows' worth of padding never read)
// Both gather x through small offsets (served by the vector L1 like the near columns of the real matrix), accumulate left to
// right and store y as 8 bytes per lane.  10M rows, persistent grid of 2048 x 256 threads, slices of 64 rows per wavefront.
// Build / run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/sw tools/micro/stream_width.hip && /tmp/sw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void kernel_a(const double *__restrict__ val, const unsigned short *__restrict__ code,
                                                const double *__restrict__ x, double *__restrict__ y, int nslices, int nrows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int s = (blockIdx.x * 4 + w); s < nslices; s += gridDim.x * 4) {
    const size_t base = (size_t)s * 5 * 64;
    double v[5];
    int c[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      v[j] = __builtin_nontemporal_load(val + base + j * 64 + lane);
      c[j] = (int)__builtin_nontemporal_load(code + base + j * 64 + lane);
    }
    const int row = s * 64 + lane;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) acc += v[j] * x[min(row + c[j], nrows - 1)];
    y[row] = acc;
  }
}

__global__ __launch_bounds__(256) void kernel_b(const double2 *__restrict__ val2, const ushort2 *__restrict__ code2,
                                                const double *__restrict__ val1, const unsigned short *__restrict__ code1,
                                                const double *__restrict__ x, double *__restrict__ y, int nslices, int nrows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int s = (blockIdx.x * 4 + w); s < nslices; s += gridDim.x * 4) {
    const size_t base2 = (size_t)s * 2 * 64, base1 = (size_t)s * 64;
    double2 v2[2];
    ushort2 c2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(val2 + base2 + j * 64 + lane));
      v2[j].x = t.x;
      v2[j].y = t.y;
      const unsigned cc = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(code2 + base2 + j * 64 + lane));
      c2[j].x = (unsigned short)(cc & 0xffffu);
      c2[j].y = (unsigned short)(cc >> 16);
    }
    const double v1 = __builtin_nontemporal_load(val1 + base1 + lane);
    const int c1 = (int)__builtin_nontemporal_load(code1 + base1 + lane);
    const int row = s * 64 + lane;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc += v2[j].x * x[min(row + (int)c2[j].x, nrows - 1)];
      acc += v2[j].y * x[min(row + (int)c2[j].y, nrows - 1)];
    }
    acc += v1 * x[min(row + c1, nrows - 1)];
    y[row] = acc;
  }
}

int main() {
  const int nrows = 10 * 1000 * 1000 / 64 * 64, nslices = nrows / 64;
  const size_t nnz = (size_t)nrows * 5;
  std::vector<double> hv(nnz, 1.0);
  std::vector<unsigned short> hc(nnz);
  for (size_t i = 0; i < nnz; ++i) hc[i] = (unsigned short)((i * 2654435761u) % 300);
  double *val, *x, *y;
  unsigned short *code;
  CK(hipMalloc(&val, nnz * 8)); CK(hipMalloc(&code, nnz * 2)); CK(hipMalloc(&x, (size_t)nrows * 8)); CK(hipMalloc(&y, (size_t)nrows * 8));
  CK(hipMemcpy(val, hv.data(), nnz * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(code, hc.data(), nnz * 2, hipMemcpyHostToDevice));
  CK(hipMemset(x, 0, (size_t)nrows * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = nnz * 10.0 + (double)nrows * 16.0;
  for (int wgs : {1024, 2048, 4096}) {
    for (int variant = 0; variant < 2; ++variant) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 10; ++it) {
          if (variant == 0)
            hipLaunchKernelGGL(kernel_a, dim3(wgs), dim3(256), 0, 0, val, code, x, y, nslices, nrows);
          else  // the same device buffers reinterpreted: pairs in the first 4/5, singles behind them
            hipLaunchKernelGGL(kernel_b, dim3(wgs), dim3(256), 0, 0, reinterpret_cast<const double2 *>(val),
                               reinterpret_cast<const ushort2 *>(code), val + (size_t)nrows * 4, code + (size_t)nrows * 4, x, y, nslices, nrows);
        }
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("%s  %4d workgroups: %.1f us per launch, %.2f TB/s of %.0f MB\n", variant ? "B (16-byte pairs)" : "A (8-byte values) ", wgs,
             best * 100.0, bytes / (best * 1e-4) / 1e12, bytes / 1e6);
    }
  }
  return 0;
}
