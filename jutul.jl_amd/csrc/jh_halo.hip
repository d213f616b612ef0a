// pack / unpack kernels of the ghost-cell halo exchange (consistent!, ext/JutulPartitionedArraysExt/linalg.jl:46)
#include "jh_internal.hpp"
namespace jh {
__global__ void halo_pack_kernel(double *buf, const double *v, const int32_t *idx, int64_t n, int bs) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * bs; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = i / bs;
    int e = (int)(i - k * bs);
    buf[i] = v[(size_t)idx[k] * bs + e];
  }
}
__global__ void halo_unpack_kernel(double *v, const double *buf, const int32_t *idx, int64_t n, int bs) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * bs; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = i / bs;
    int e = (int)(i - k * bs);
    v[(size_t)idx[k] * bs + e] = buf[i];
  }
}
void halo_pack_launch(hipStream_t s, double *buf, const double *v, const int32_t *idx, int64_t n, int bs) {
  int g = (int)std::max<int64_t>(1, std::min<int64_t>((n * bs + 255) / 256, 1024));
  hipLaunchKernelGGL(halo_pack_kernel, dim3(g), dim3(256), 0, s, buf, v, idx, n, bs);
}
void halo_unpack_launch(hipStream_t s, double *v, const double *buf, const int32_t *idx, int64_t n, int bs) {
  int g = (int)std::max<int64_t>(1, std::min<int64_t>((n * bs + 255) / 256, 1024));
  hipLaunchKernelGGL(halo_unpack_kernel, dim3(g), dim3(256), 0, s, v, buf, idx, n, bs);
}
}  // namespace jh
