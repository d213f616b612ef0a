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
// ---- mailbox all-reduce ------------------------------------------------------------------------------------------------
// The BiCGStab scalars (1-2 doubles, three times per iteration) are reduced without a collective library call: every rank
// owns a mailbox in uncached device memory that all ranks of the node have mapped (hipIpc); a one-wavefront kernel writes
// the rank's contribution into every mailbox (xGMI peer stores, system-scope release on the flag), waits until all ranks
// have written into its own, and sums in rank order -- so every rank obtains the same bits.  Two slot sets alternate with
// the parity of the epoch: a rank can be at most one all-reduce ahead of the slowest one, because it cannot finish epoch
// e+1 before everybody has contributed to it, i.e. has finished reading epoch e.
size_t mailbox_bytes() { return sizeof(Mailbox); }

__global__ __launch_bounds__(64) void mailbox_allreduce_kernel(MailArgs A, double *p, int n, int op) { mailbox_allreduce_body(A, p, n, op); }
void mailbox_allreduce_launch(hipStream_t s, const MailArgs &A, double *p, int n, int op) {
  hipLaunchKernelGGL(mailbox_allreduce_kernel, dim3(1), dim3(64), 0, s, A, p, n, op);
}

// self-test of the consumer-side all-reduce (xr_push / xr_sum): every wavefront of a many-workgroup launch collects the peers' sums
__global__ __launch_bounds__(64) void xr_selftest_kernel(MailArgs A, double v0, double v1, double *out) {
  XrRegs R;
  xr_load(A, R);
  double s0 = v0, s1 = v1;
  if (blockIdx.x == 0) xr_push(A, s0, s1);
  xr_sum(A, R, blockIdx.x == 0, s0, s1);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = s0; out[2 * blockIdx.x + 1] = s1; }
}
void xr_selftest_launch(hipStream_t s, const MailArgs &A, double v0, double v1, double *out, int nblocks) {
  hipLaunchKernelGGL(xr_selftest_kernel, dim3(nblocks), dim3(64), 0, s, A, v0, v1, out);
}

// ---- push halo ------------------------------------------------------------------------------------------------------------
// Inside the Krylov loop the ghost exchange needs no collective library call either: the producer of the vector (the fused
// ILU(0) apply, or this pack kernel) stores every boundary row straight into the landing buffers of the ranks that hold it
// as a ghost (peer-mapped uncached memory, xGMI stores).  The finish kernel, next on the stream (so those stores are
// released), raises this rank's flag in every neighbour's mailbox, waits for the neighbours' flags in its own, and copies
// the landed values into the ghost rows.  HALO_S landing buffers are used round robin by the exchange counter.
__global__ void halo_push_pack_kernel(double *const *dst, const double *v, const int32_t *idx, int64_t n, int bs) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    double *d = dst[k];
    for (int e = 0; e < bs; ++e) __hip_atomic_store(d + e, v[(size_t)idx[k] * bs + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void halo_push_finish_kernel(Mailbox *self, Mailbox *const *peers, const int32_t *nbr, int n_nbr,
                                                               int rank, unsigned long long epoch, const double *landing, double *v,
                                                               const int32_t *recv_idx, int64_t n_recv, int bs, MailErr *err,
                                                               unsigned long long timeout_ticks) {
  const int par = (int)(epoch % (unsigned long long)HALO_S);
  if (threadIdx.x < n_nbr) {
    const int q = nbr[threadIdx.x];
    if (blockIdx.x == 0)  // my rows for q are in place (previous kernel on this stream): tell q
      __hip_atomic_store(&peers[q]->hflag[par][rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!mailbox_wait(&self->hflag[par][q], epoch, timeout_ticks, self) && blockIdx.x == 0) mailbox_wait_failed(self, err, 2ull, q, epoch);
  }
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_recv * bs; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / bs;
    const int e = (int)(i - k * bs);
    v[(size_t)recv_idx[k] * bs + e] = __hip_atomic_load(landing + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void halo_push_pack_launch(hipStream_t s, double *const *dst, const double *v, const int32_t *idx, int64_t n, int bs) {
  int g = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(halo_push_pack_kernel, dim3(g), dim3(256), 0, s, dst, v, idx, n, bs);
}
void halo_push_finish_launch(hipStream_t s, Mailbox *self, Mailbox *const *peers, const int32_t *nbr, int n_nbr, int rank, uint64_t epoch,
                             const double *landing, double *v, const int32_t *recv_idx, int64_t n_recv, int bs, MailErr *err,
                             uint64_t timeout_ticks) {
  int g = (int)std::max<int64_t>(1, std::min<int64_t>((n_recv * bs + 2047) / 2048, 16));
  hipLaunchKernelGGL(halo_push_finish_kernel, dim3(g), dim3(256), 0, s, self, peers, nbr, n_nbr, rank, (unsigned long long)epoch,
                     landing, v, recv_idx, n_recv, bs, err, (unsigned long long)timeout_ticks);
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
