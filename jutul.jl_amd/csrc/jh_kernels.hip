// Vector kernels, deterministic reductions and the CSR SpMV (a-10) for gfx950.
//
// SpMV design (StaticCSR/mat.jl:24-68 semantics): rows of a TPFA Jacobian are short (~5 block entries on a tet
// mesh), so one wavefront per row would idle 90% of its lanes.  Instead a 256-thread workgroup owns a TILE of
// consecutive rows holding <= TILE_NNZ entries: every lane streams (val, col) pairs fully coalesced, gathers
// x[col], stages the products in LDS, and then one lane per row reduces its LDS row segment left->right (the
// reference's accumulation order).  Tiles are mapped to XCDs in contiguous chunks so that each XCD's private
// L2 only ever holds one window of x.  Bound: HBM; algorithmic bytes 12*nnz + 20*n.
#include <cstdlib>

#include "jh_internal.hpp"

namespace jh {
int comm_size(jh_context ctx);

__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
  // workgroup b runs on XCD b % 8 (observed placement, performance only): give each XCD a contiguous chunk
  int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
  return (b % NUM_XCD) * chunk + b / NUM_XCD;
}

// ---------------------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------------------
__global__ void fill_kernel(double *x, int64_t n, double v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void axpby_kernel(double *y, double a, const double *x, double b, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}
__global__ void permute_in_kernel(double *dst, const double *src, const int32_t *perm, int64_t n, int bs) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * bs; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / bs;
    int e = (int)(i - r * bs);
    dst[i] = src[(int64_t)perm[r] * bs + e];
  }
}
__global__ void permute_out_kernel(double *dst, const double *src, const int32_t *perm, int64_t n, int bs) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * bs; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / bs;
    int e = (int)(i - r * bs);
    dst[(int64_t)perm[r] * bs + e] = src[i];
  }
}
// dst[k] = src[slot[k]] (gather) or dst[slot[k]] = src[k] (scatter), blocks of bb doubles
__global__ void gather_blocks_kernel(double *dst, const double *src, const int32_t *slot, int64_t nblk, int bb, bool scatter) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nblk * bb; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = i / bb;
    int e = (int)(i - k * bb);
    int64_t j = (int64_t)slot[k] * bb + e;
    if (scatter) dst[j] = src[i]; else dst[i] = src[j];
  }
}

static inline int grid_for(int64_t n, int threads = 256) {
  int64_t g = (n + threads - 1) / threads;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;  // grid-stride beyond 8 blocks/CU
  return (int)g;
}

void k_fill(hipStream_t s, double *x, int64_t n, double v) {
  if (n) hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n, v);
}
void k_copy(hipStream_t s, double *dst, const double *src, int64_t n) {
  if (n) JH_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, s));
}
void k_axpby(hipStream_t s, double *y, double a, const double *x, double b, int64_t n) {
  if (n) hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, a, x, b, n);
}
void k_negate(hipStream_t s, double *dst, const double *src, int64_t n) { k_axpby(s, dst, -1.0, src, 0.0, n); }
void k_permute_in(hipStream_t s, double *dst, const double *src, const int32_t *perm, int64_t n, int bs) {
  if (n) hipLaunchKernelGGL(permute_in_kernel, dim3(grid_for(n * bs)), dim3(256), 0, s, dst, src, perm, n, bs);
}
void k_permute_out(hipStream_t s, double *dst, const double *src, const int32_t *perm, int64_t n, int bs) {
  if (n) hipLaunchKernelGGL(permute_out_kernel, dim3(grid_for(n * bs)), dim3(256), 0, s, dst, src, perm, n, bs);
}
// component e of a [bs, n] device-ordered vector -> contiguous host-ordered array of n doubles (perm == nullptr: same order)
__global__ void component_out_kernel(double *dst, const double *src, const int32_t *perm, int64_t n, int bs, int e) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
    dst[perm ? (int64_t)perm[r] : r] = src[r * bs + e];
}
void k_component_out(hipStream_t s, double *dst, const double *src, const int32_t *perm, int64_t n, int bs, int e) {
  if (n) hipLaunchKernelGGL(component_out_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, src, perm, n, bs, e);
}
void k_gather_blocks(hipStream_t s, double *dst, const double *src, const int32_t *slot, int64_t nblk, int bb, bool scatter) {
  if (nblk) hipLaunchKernelGGL(gather_blocks_kernel, dim3(grid_for(nblk * bb)), dim3(256), 0, s, dst, src, slot, nblk, bb, scatter);
}

// ---------------------------------------------------------------------------------------------------------
// reductions: per-block partial (wave shuffle + LDS), then a single-block ordered final sum => deterministic
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// maximum that keeps a NaN (Julia's max / maximum propagate NaN: increment_norm, variable_change_report, models.jl:955-1038;
// fmax drops it)
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = nanmax(v, __shfl_down(v, off, 64));
  return v;
}
template <bool MAX>
__device__ __forceinline__ double block_reduce(double v, double *sm) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = MAX ? wave_max(v) : wave_sum(v);
  if (lane == 0) sm[w] = v;
  __syncthreads();
  int nw = blockDim.x >> 6;
  double r = 0.0;
  if (threadIdx.x == 0) {
    r = sm[0];
    for (int i = 1; i < nw; ++i) r = MAX ? nanmax(r, sm[i]) : r + sm[i];
  }
  __syncthreads();
  return r;  // valid on thread 0
}

constexpr int RED_BLOCKS = 1024;

__global__ __launch_bounds__(256) void dot2_partial_kernel(const double *a, const double *b, const double *c, const double *d,
                                                           int64_t n, double *part, size_t stride) {
  __shared__ double sm[8];
  double s0 = 0, s1 = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    s0 += a[i] * b[i];
    if (c) s1 += c[i] * d[i];
  }
  double r0 = block_reduce<false>(s0, sm);
  double r1 = c ? block_reduce<false>(s1, sm) : 0.0;
  if (threadIdx.x == 0) {
    part[blockIdx.x] = r0;
    if (c) part[stride + blockIdx.x] = r1;
  }
}
__global__ __launch_bounds__(256) void absmax_partial_kernel(const double *r, int64_t ncell, int bs, int e, double *part) {
  __shared__ double sm[8];
  double m = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * blockDim.x) {
    double v = fabs(r[i * bs + e]);
    m = (v > m || v != v) ? v : m;  // NaN propagates like Julia's maximum(abs, ...)
  }
  double r0 = block_reduce<true>(m, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r0;
}
// final stage: see final_reduce_body (jh_internal.hpp)
template <bool MAX>
__global__ __launch_bounds__(FIN_THREADS) void final_reduce_kernel(const double *part, size_t stride, int nparts, int count, double *out,
                                                                   const double *done, MailArgs mail) {
  if (done && *done != 0.0) return;
  final_reduce_body<MAX>(part, stride, nparts, count, out);
  if (mail.self) mailbox_allreduce_body(mail, out, count, MAX ? 1 : 0);  // over the ranks, in the same launch
}

void ensure_partials(jh_context ctx, size_t min_stride) {
  if (ctx->partials.n == 0 || ctx->partial_stride < min_stride) {
    JH_HIP(hipStreamSynchronize(ctx->stream));
    ctx->partial_stride = std::max<size_t>(65536, min_stride);  // also used by tile-level partials
    // (Cached memory.  Uncached partials -- tried in round 3 to make the cross-XCD hand-over of the opt-in in-kernel second stage
    // coherent by construction -- made processes with several rank threads abort now and then inside the runtime; the opt-in
    // path relies on write-through stores by the producers and an agent-scope acquire by the one workgroup that reads them.)
    ctx->partials.alloc(ctx->partial_stride * 4);
  }
}
void k_final_reduce(jh_context ctx, int nparts, int count, int slot, bool is_max, const double *done, const MailArgs *mail) {
  const MailArgs ma = mail ? *mail : MailArgs();
  if (is_max)
    hipLaunchKernelGGL(final_reduce_kernel<true>, dim3(1), dim3(FIN_THREADS), 0, ctx->stream, ctx->partials.p, ctx->partial_stride, nparts, count, ctx->scalars.p + slot, done, ma);
  else
    hipLaunchKernelGGL(final_reduce_kernel<false>, dim3(1), dim3(FIN_THREADS), 0, ctx->stream, ctx->partials.p, ctx->partial_stride, nparts, count, ctx->scalars.p + slot, done, ma);
}

// mail: the sums are all-reduced over the ranks in the second stage's launch (mailboxes)
void k_dot2_to(jh_context ctx, const double *a, const double *b, const double *c, const double *d, int64_t n, double *out, const MailArgs *mail) {
  ensure_partials(ctx, 0);
  int g = grid_for(n);
  if (g > RED_BLOCKS) g = RED_BLOCKS;
  hipLaunchKernelGGL(dot2_partial_kernel, dim3(g), dim3(256), 0, ctx->stream, a, b, c, d, n, ctx->partials.p, ctx->partial_stride);
  hipLaunchKernelGGL(final_reduce_kernel<false>, dim3(1), dim3(FIN_THREADS), 0, ctx->stream, ctx->partials.p, ctx->partial_stride, g,
                     c ? 2 : 1, out, (const double *)nullptr, mail ? *mail : MailArgs());
}
void k_dot2(jh_context ctx, const double *a, const double *b, const double *c, const double *d, int64_t n, int slot, const MailArgs *mail) {
  k_dot2_to(ctx, a, b, c, d, n, ctx->scalars.p + slot, mail);
}
void k_dot(jh_context ctx, const double *a, const double *b, int64_t n, int slot) { k_dot2(ctx, a, b, nullptr, nullptr, n, slot); }

void k_absmax_strided(jh_context ctx, const double *r, int64_t ncell, int bs, int slot) {
  ensure_partials(ctx, 0);
  int g = grid_for(ncell);
  if (g > RED_BLOCKS) g = RED_BLOCKS;
  for (int e = 0; e < bs; ++e) {
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(g), dim3(256), 0, ctx->stream, r, ncell, bs, e, ctx->partials.p);
    hipLaunchKernelGGL(final_reduce_kernel<true>, dim3(1), dim3(FIN_THREADS), 0, ctx->stream, ctx->partials.p, ctx->partial_stride, g, 1,
                       ctx->scalars.p + slot + e, (const double *)nullptr, MailArgs());
  }
}

// Per variable e: sum|a - b|, max|a - b|, sum|a|, max|a| over the first ncell cells (b == nullptr: b = 0, so the two pairs
// coincide) -- increment_norm (models.jl:955-965) and variable_change_report (models.jl:1023-1038) on the device.
// Sums are NaN/Inf propagating, so a non-finite increment shows up in them (check_increment).
__global__ __launch_bounds__(256) void absstats_partial_kernel(const double *a, const double *b, int64_t ncell, int bs, int e, double *part,
                                                               size_t stride) {
  __shared__ double sm[8];
  double sd = 0, md = 0, sa = 0, ma = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = a[i * bs + e];
    const double d = fabs(b ? x - b[i * bs + e] : x), v = fabs(x);
    sd += d;
    md = (d > md || d != d) ? d : md;
    sa += v;
    ma = (v > ma || v != v) ? v : ma;
  }
  const double r0 = block_reduce<false>(sd, sm), r1 = block_reduce<true>(md, sm), r2 = block_reduce<false>(sa, sm), r3 = block_reduce<true>(ma, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = r0;
    part[stride + blockIdx.x] = r1;
    part[2 * stride + blockIdx.x] = r2;
    part[3 * stride + blockIdx.x] = r3;
  }
}
__global__ __launch_bounds__(FIN_THREADS) void absstats_final_kernel(const double *part, size_t stride, int nparts, double *out) {
  final_reduce_body<false>(part, stride, nparts, 1, out);
  final_reduce_body<true>(part + stride, stride, nparts, 1, out + 1);
  final_reduce_body<false>(part + 2 * stride, stride, nparts, 1, out + 2);
  final_reduce_body<true>(part + 3 * stride, stride, nparts, 1, out + 3);
}
void k_absstats(jh_context ctx, const double *a, const double *b, int64_t ncell, int bs, int slot) {
  ensure_partials(ctx, 0);
  int g = grid_for(ncell);
  if (g > RED_BLOCKS) g = RED_BLOCKS;
  for (int e = 0; e < bs; ++e) {
    hipLaunchKernelGGL(absstats_partial_kernel, dim3(g), dim3(256), 0, ctx->stream, a, b, ncell, bs, e, ctx->partials.p, ctx->partial_stride);
    hipLaunchKernelGGL(absstats_final_kernel, dim3(1), dim3(FIN_THREADS), 0, ctx->stream, ctx->partials.p, ctx->partial_stride, g,
                       ctx->scalars.p + slot + 4 * e);
  }
}

// Device scalars -> host.  A D2H copy + hipStreamSynchronize costs a DMA packet and an interrupt-driven wake-up per read;
// the seam path reads scalars three times per Newton iteration (convergence, increment norms, change report).  Instead a
// one-wavefront kernel stores the values into pinned, host-coherent memory followed by a sequence number, and the host spins
// on that (the protocol of the Krylov loop's iteration records).  Option read_sync = 1 restores the copy + synchronise.
__global__ void publish_scalars_kernel(const double *sc, int count, double *dst, double seq) {
  if ((int)threadIdx.x < count) __hip_atomic_store(dst + threadIdx.x, sc[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(dst + (JH_NSCALARS - 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Two halves, so that a caller can enqueue work that does not depend on the scalars between the publishing kernel and the host's
// wait for it (the device goes on while the host reads).  begin returns the record's sequence number (0: the synchronous path).
double read_scalars_begin(jh_context ctx, int slot, int count) {
  const bool use_sync = ctx->opt.read_sync != 0;
  if (use_sync || !ctx->h_rd || count >= JH_NSCALARS - 1) return 0.0;
  const double seq = (double)(++ctx->rd_seq);
  hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->scalars.p + slot, count, ctx->h_rd, seq);
  return seq;
}
void read_scalars_end(jh_context ctx, double seq, int slot, int count, double *out) {
  if (seq == 0.0) {
    JH_HIP(hipMemcpyAsync(ctx->h_scalars + slot, ctx->scalars.p + slot, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    JH_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < count; ++i) out[i] = ctx->h_scalars[slot + i];
    return;
  }
  volatile double *r = ctx->h_rd;
  SpinPacer pace;
  for (;;) {
    if (r[JH_NSCALARS - 1] == seq) break;
    if (pace.due()) {  // the stream must still be busy, otherwise the kernel was lost (fault)
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipSuccess) {
        if (r[JH_NSCALARS - 1] == seq) break;
        JH_THROW("device scalars were not published");
      } else if (q != hipErrorNotReady) {
        JH_HIP(q);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int i = 0; i < count; ++i) out[i] = r[i];
}
void read_scalars(jh_context ctx, int slot, int count, double *out) {
  read_scalars_end(ctx, read_scalars_begin(ctx, slot, count), slot, count, out);
}
double read_scalar(jh_context ctx, int slot) {
  double v;
  read_scalars(ctx, slot, 1, &v);
  return v;
}

// ---------------------------------------------------------------------------------------------------------
// SpMV: y = alpha*A*x (+ beta*y)
// ---------------------------------------------------------------------------------------------------------
constexpr int SPMV_WIN = 128;  // x entries staged in LDS on either side of a tile's own rows
template <int BS, int DOT, bool XW>
__global__ __launch_bounds__(TILE_THREADS) void spmv_tile_kernel(const int32_t *__restrict__ tile_row, int ntiles, int ncols,
                                                                 const int32_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ col,
                                                                 const double *__restrict__ val,
                                                                 const double *__restrict__ x, double *__restrict__ y,
                                                                 double alpha, double beta, const double *__restrict__ dw,
                                                                 int dot_rows, double *__restrict__ part, size_t pstride,
                                                                 const double *done, int t_begin) {
  if (done && *done != 0.0) return;
  __shared__ double prod[TILE_NNZ * BS];
  __shared__ int32_t rp[TILE_ROWS + 1];
  __shared__ double red[8];
  __shared__ double xw[XW ? TILE_ROWS + 2 * SPMV_WIN : 1];
  const int tid = threadIdx.x;
  double d0 = 0.0, d1 = 0.0;  // fused dot partials of this lane
  // Persistent over tiles: workgroup b belongs to XCD b % 8 and walks that XCD's contiguous chunk of tiles with stride
  // gridDim/8, so a fused dot product needs ONE block reduction and ONE partial per workgroup (<= 2048 partials).
  // (ntiles tiles starting at tile t_begin: the split SpMV of a rank-local subdomain launches interior and boundary tiles
  // separately)
  const int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
  const int xcd = blockIdx.x % NUM_XCD, wg = blockIdx.x / NUM_XCD, wgs = gridDim.x / NUM_XCD;
  for (int tl = wg; tl < chunk; tl += wgs) {
  if (xcd * chunk + tl >= ntiles) break;
  const int t = t_begin + xcd * chunk + tl;
  if (tl != wg) __syncthreads();  // the previous tile's LDS contents are dead only once every lane is done with them
  const int4 td = reinterpret_cast<const int4 *>(tile_row)[t];  // tile descriptor (Pattern::tile_desc)
  const int r0 = td.x, nrows = td.y, base = td.z, cnt = td.w;
  if (nrows == 1 && cnt > TILE_NNZ) {
    // long row: the whole workgroup strides over one row, then block-reduces
    double acc[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) acc[e] = 0.0;
    for (int k = tid; k < cnt; k += TILE_THREADS) {
      const int c = col[base + k];
      if (BS == 1) {
        acc[0] += val[base + k] * x[c];
      } else {
        const double *A = val + (size_t)(base + k) * BS * BS;
#pragma unroll
        for (int d = 0; d < BS; ++d) {
          double xd = x[(size_t)c * BS + d];
#pragma unroll
          for (int e = 0; e < BS; ++e) acc[e] += A[d * BS + e] * xd;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      double s = block_reduce<false>(acc[e], red);
      if (tid == 0) {
        size_t o = (size_t)r0 * BS + e;
        const double yv = (beta == 0.0) ? alpha * s : alpha * s + beta * y[o];
        y[o] = yv;
        if (DOT && r0 < dot_rows) { d0 += yv * dw[o]; if (DOT == 2) d1 += yv * yv; }
      }
    }
    continue;
  }
  for (int i = tid; i <= nrows; i += TILE_THREADS) rp[i] = rowptr[r0 + i] - base;
  if (BS == 1) {
    // all (col, val) loads of the lane's entries are issued before the dependent x gathers: 4 + 4 + 4 loads in flight
    constexpr int KPT = TILE_NNZ / TILE_THREADS;
    int cidx[KPT];
    double vv[KPT], xg[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = tid + j * TILE_THREADS;
      cidx[j] = (k < cnt) ? col[base + k] : 0;
      vv[j] = (k < cnt) ? val[base + k] : 0.0;
    }
    if (XW) {
      // In the device ordering (compact graph blocks) >80% of a tile's columns lie within SPMV_WIN rows of the tile: that
      // slice of x is fetched with two coalesced loads per lane, staged in LDS and gathered from there; only the remaining
      // columns are gathered from global memory (6x fewer scattered cache-line requests).
      const int w0 = max(0, r0 - SPMV_WIN);
      const int wn = min(ncols, r0 + TILE_ROWS + SPMV_WIN) - w0;  // <= TILE_ROWS + 2*SPMV_WIN = 2*TILE_THREADS
      const double xa = (tid < wn) ? x[w0 + tid] : 0.0;
      const double xb = (tid + TILE_THREADS < wn) ? x[w0 + tid + TILE_THREADS] : 0.0;
      xw[tid] = xa;
      xw[tid + TILE_THREADS] = xb;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const unsigned loc = (unsigned)(cidx[j] - w0);
        xg[j] = lds_or_global(xw, loc, loc < (unsigned)wn, x, (size_t)cidx[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < KPT; ++j) xg[j] = x[cidx[j]];
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = tid + j * TILE_THREADS;
      if (k < cnt) prod[k] = vv[j] * xg[j];
    }
  }
  for (int k = tid; k < cnt && BS > 1; k += TILE_THREADS) {
    const int c = col[base + k];
    if (BS == 1) {
      prod[k] = val[base + k] * x[c];
    } else {
      const double *A = val + (size_t)(base + k) * BS * BS;
      double xv[BS], p[BS];
#pragma unroll
      for (int d = 0; d < BS; ++d) xv[d] = x[(size_t)c * BS + d];
#pragma unroll
      for (int e = 0; e < BS; ++e) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < BS; ++d) s += A[d * BS + e] * xv[d];
        p[e] = s;
      }
#pragma unroll
      for (int e = 0; e < BS; ++e) prod[k * BS + e] = p[e];
    }
  }
  __syncthreads();
  if (tid < nrows) {
    double s[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) s[e] = 0.0;
    for (int j = rp[tid]; j < rp[tid + 1]; ++j) {
#pragma unroll
      for (int e = 0; e < BS; ++e) s[e] = s[e] + prod[j * BS + e];
    }
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      size_t o = (size_t)(r0 + tid) * BS + e;
      const double yv = (beta == 0.0) ? alpha * s[e] : alpha * s[e] + beta * y[o];
      y[o] = yv;
      if (DOT && r0 + tid < dot_rows) { d0 += yv * dw[o]; if (DOT == 2) d1 += yv * yv; }
    }
  }
  }  // tile loop
  if (DOT) {  // one partial per workgroup
    __syncthreads();
    const double p0 = block_reduce<false>(d0, red);
    if (tid == 0) part[blockIdx.x] = p0;
    if (DOT == 2) {
      const double p1 = block_reduce<false>(d1, red);
      if (tid == 0) part[pstride + blockIdx.x] = p1;
    }
  }
}


// Software-pipelined variant of the scalar SpMV with a fused dot (the BiCGStab workhorse): a tile costs a chain of dependent
// memory latencies (descriptor -> entries -> far x), and with 8 resident workgroups per CU one tile in flight per workgroup
// leaves the memory system under-subscribed.  Here the first-level loads of tile i+1 (column ids, values, x window, row
// offsets) are issued before tile i is processed out of registers and LDS.
struct SpmvTilePre {
  int r0, nrows, base, cnt;
  int cidx[TILE_NNZ / TILE_THREADS];
  double vv[TILE_NNZ / TILE_THREADS];
  double xa, xb;
  int rpv;
};
template <int DOT>
__global__ __launch_bounds__(TILE_THREADS) void spmv_pipe_kernel(const int32_t *__restrict__ tile_desc, int ntiles, int ncols,
                                                                 const int32_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ col,
                                                                 const double *__restrict__ val,
                                                                 const double *__restrict__ x, double *__restrict__ y,
                                                                 double alpha, double beta, const double *__restrict__ dw,
                                                                 int dot_rows, double *__restrict__ part, size_t pstride,
                                                                 const double *done, int t_begin) {
  if (done && *done != 0.0) return;
  constexpr int KPT = TILE_NNZ / TILE_THREADS;
  __shared__ double prod[TILE_NNZ];
  __shared__ int32_t rp[TILE_ROWS + 1];
  __shared__ double red[8];
  __shared__ double xw[TILE_ROWS + 2 * SPMV_WIN];
  const int tid = threadIdx.x;
  double d0 = 0.0, d1 = 0.0;
  const int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
  const int xcd = blockIdx.x % NUM_XCD, wg = blockIdx.x / NUM_XCD, wgs = gridDim.x / NUM_XCD;
  auto in_range = [&](int tl) { return tl < chunk && xcd * chunk + tl < ntiles; };
  auto issue = [&](int tl, SpmvTilePre &T) {  // descriptor + first-level loads of tile tl (entries beyond the tile: zeros)
    const int4 td = reinterpret_cast<const int4 *>(tile_desc)[t_begin + xcd * chunk + tl];
    T.r0 = td.x; T.nrows = td.y; T.base = td.z; T.cnt = td.w;
    if (T.nrows == 1 && T.cnt > TILE_NNZ) return;  // long row: handled synchronously
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = tid + j * TILE_THREADS;
      T.cidx[j] = (k < T.cnt) ? col[T.base + k] : 0;
      T.vv[j] = (k < T.cnt) ? val[T.base + k] : 0.0;
    }
    const int w0 = max(0, T.r0 - SPMV_WIN);
    const int wn = min(ncols, T.r0 + TILE_ROWS + SPMV_WIN) - w0;
    T.xa = (tid < wn) ? x[w0 + tid] : 0.0;
    T.xb = (tid + TILE_THREADS < wn) ? x[w0 + tid + TILE_THREADS] : 0.0;
    T.rpv = (tid < T.nrows) ? rowptr[T.r0 + tid] - T.base : T.cnt;
  };
  SpmvTilePre cur, nxt;
  int tl = wg;
  bool have = in_range(tl);
  if (have) issue(tl, cur);
  while (have) {
    const int tln = tl + wgs;
    const bool have_next = in_range(tln);
    if (have_next) issue(tln, nxt);
    const int r0 = cur.r0, nrows = cur.nrows, base = cur.base, cnt = cur.cnt;
    if (nrows == 1 && cnt > TILE_NNZ) {  // long row: the whole workgroup strides over one row, then block-reduces
      double acc = 0.0;
      for (int k = tid; k < cnt; k += TILE_THREADS) acc += val[base + k] * x[col[base + k]];
      const double s = block_reduce<false>(acc, red);
      if (tid == 0) {
        const double yv = (beta == 0.0) ? alpha * s : alpha * s + beta * y[r0];
        y[r0] = yv;
        if (r0 < dot_rows) { d0 += yv * dw[r0]; if (DOT == 2) d1 += yv * yv; }
      }
    } else {
      const int w0 = max(0, r0 - SPMV_WIN);
      const int wn = min(ncols, r0 + TILE_ROWS + SPMV_WIN) - w0;
      rp[tid] = cur.rpv;
      if (tid == 0) rp[nrows] = cnt;
      xw[tid] = cur.xa;
      xw[tid + TILE_THREADS] = cur.xb;
      __syncthreads();
      double xg[KPT];
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const unsigned loc = (unsigned)(cur.cidx[j] - w0);
        xg[j] = lds_or_global(xw, loc, loc < (unsigned)wn, x, (size_t)cur.cidx[j]);
      }
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * TILE_THREADS;
        if (k < cnt) prod[k] = cur.vv[j] * xg[j];
      }
      __syncthreads();
      if (tid < nrows) {
        double s = 0.0;
        for (int j = rp[tid]; j < rp[tid + 1]; ++j) s = s + prod[j];
        const int o = r0 + tid;
        const double yv = (beta == 0.0) ? alpha * s : alpha * s + beta * y[o];
        y[o] = yv;
        if (o < dot_rows) { d0 += yv * dw[o]; if (DOT == 2) d1 += yv * yv; }
      }
    }
    __syncthreads();  // LDS is reused by the next tile
    cur = nxt;
    tl = tln;
    have = have_next;
  }
  __syncthreads();
  const double p0 = block_reduce<false>(d0, red);
  if (tid == 0) part[blockIdx.x] = p0;
  if (DOT == 2) {
    const double p1 = block_reduce<false>(d1, red);
    if (tid == 0) part[pstride + blockIdx.x] = p1;
  }
}

// second stage of a fused SpMV dot; with dot->allreduce also over the ranks (same launch if the mailboxes are on)
void spmv_dot_reduce(jh_context ctx, const SpmvDot *dot, int nparts, const double *done) {
  const int cnt = dot->mode == 2 ? 2 : 1;
  MailArgs ma;
  const bool fused = dot->allreduce && comm_mail_args(ctx, cnt, &ma);
  k_final_reduce(ctx, nparts, cnt, dot->slot, false, done, fused ? &ma : nullptr);
  if (dot->allreduce && !fused) comm_allreduce_dev(ctx, ctx->scalars.p + dot->slot, cnt, 0);
}

int k_spmv(jh_context ctx, const Pattern &P, const double *val, const double *x, double *y, double alpha, double beta,
           const SpmvDot *dot, const double *done, const SpmvRange *rng) {
  if (P.n == 0) return 0;
  const int t_begin = rng ? rng->t0 : 0;
  const int ntl = rng ? rng->t1 - rng->t0 : P.ntiles;
  const int part_off = rng ? rng->part_off : 0;
  auto reduce = [&](int nparts) { spmv_dot_reduce(ctx, dot, nparts, done); };
  if (ntl <= 0) {
    if (dot && rng && rng->reduce && part_off > 0) reduce(part_off);
    return 0;
  }
  int chunk = (ntl + NUM_XCD - 1) / NUM_XCD;
  const int wg_per_xcd = ctx->opt.spmv_waves_per_xcd > 0 ? std::max<int>(1, (int)ctx->opt.spmv_waves_per_xcd / 4) : 256;  // default: 32 CUs x 8 workgroups
  const int mode = dot ? dot->mode : 0;
  const int per_xcd = mode ? std::min(chunk, wg_per_xcd) : chunk;  // plain SpMV: one tile per workgroup
  dim3 grid(per_xcd * NUM_XCD), block(TILE_THREADS);
  if (mode) ensure_partials(ctx, (size_t)grid.x + part_off);
  const double *dw = dot ? dot->w : nullptr;
  const int drows = dot ? (int)dot->n_rows : 0;
  double *part = ctx->partials.p + part_off;
  const size_t ps = ctx->partial_stride;
  const bool xwin = ctx->opt.spmv_window != 0;
#define JH_SPMV(BSV, DV)                                                                                                      \
  if (BSV == 1 && xwin) JH_SPMV_X(BSV, DV, true); else JH_SPMV_X(BSV, DV, false)
#define JH_SPMV_X(BSV, DV, XWV) hipLaunchKernelGGL((spmv_tile_kernel<BSV, DV, (BSV == 1) && XWV>), grid, block, 0, ctx->stream, P.d_tile_desc.p, ntl, (int)P.n, P.d_rowptr.p, P.d_col.p, val, x, y, alpha, beta, dw, drows, part, ps, done, t_begin)
  const bool pipe = ctx->opt.spmv_pipe != 0;
#define JH_SPMV_PIPE(DV) hipLaunchKernelGGL((spmv_pipe_kernel<DV>), grid, block, 0, ctx->stream, P.d_tile_desc.p, ntl, (int)P.n, P.d_rowptr.p, P.d_col.p, val, x, y, alpha, beta, dw, drows, part, ps, done, t_begin)
  switch (P.bs * 10 + mode) {
    case 10: JH_SPMV(1, 0); break;
    case 11: if (xwin && pipe) JH_SPMV_PIPE(1); else JH_SPMV(1, 1); break;
    case 12: if (xwin && pipe) JH_SPMV_PIPE(2); else JH_SPMV(1, 2); break;
    case 20: JH_SPMV(2, 0); break;
    case 21: JH_SPMV(2, 1); break;
    case 22: JH_SPMV(2, 2); break;
    case 30: JH_SPMV(3, 0); break;
    case 31: JH_SPMV(3, 1); break;
    case 32: JH_SPMV(3, 2); break;
    default: JH_THROW("unsupported block size");
  }
#undef JH_SPMV
#undef JH_SPMV_X
#undef JH_SPMV_PIPE
  // second stage of the fused dot: over the partials of all launches that make up this product
  if (mode && (!rng || rng->reduce)) reduce(part_off + (int)grid.x);
  return (int)grid.x;
}

// unit_diagonalize!: ghost rows -> -I, r_ghost -> 0 (ext/JutulPartitionedArraysExt/linalg.jl:18-35)
__global__ void unit_diag_kernel(const int32_t *rowptr, const int32_t *col, double *val, double *r, int64_t n_owned, int64_t n, int bs) {
  int64_t row = n_owned + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (row >= n) return;
  for (int e = 0; e < bs; ++e) r[row * bs + e] = 0.0;
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
    double *A = val + (size_t)k * bs * bs;
    for (int i = 0; i < bs * bs; ++i) A[i] = 0.0;
    if (col[k] == row)
      for (int e = 0; e < bs; ++e) A[e * bs + e] = -1.0;
  }
}
// apply_scaling_to_linearized_system! (linsolve/default.jl:325-352): kind 1 = :diagonal, F = 1/|A_ii[j,j]| (scalar
// entries ~ 0 -> 1, default.jl:354-385), J <- diag(F) J, r <- F .* r ; kind 2 = :dt, J <- dt J, r <- dt r.
__global__ void scale_system_kernel(const int32_t *rowptr, const int32_t *diag, double *val, double *r, int64_t n, int bs, int kind, double dt) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double F[3];
  for (int e = 0; e < bs; ++e) {
    if (kind == 2) { F[e] = dt; continue; }
    const double a = val[(size_t)diag[i] * bs * bs + e * bs + e];
    // scalar variant guards `A_ii ≈ 0` (default.jl:374-379); isapprox against 0 with the default rtol holds only for 0
    F[e] = (bs == 1 && a == 0.0) ? 1.0 : 1.0 / fabs(a);
  }
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
    for (int d = 0; d < bs; ++d)
      for (int e = 0; e < bs; ++e) val[(size_t)k * bs * bs + d * bs + e] *= F[e];  // D_mat * block: row e scaled by F[e]
  for (int e = 0; e < bs; ++e) r[i * bs + e] *= F[e];
}
void k_scale_system(hipStream_t s, const Pattern &P, double *val, double *r, int kind, double dt) {
  if (P.n) hipLaunchKernelGGL(scale_system_kernel, dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, s, P.d_rowptr.p, P.d_diag.p, val, r, P.n, P.bs, kind, dt);
}

__global__ void zero_slots_kernel(double *val, const int32_t *slots, int64_t n, int bb) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * bb; i += (int64_t)gridDim.x * blockDim.x)
    val[(size_t)slots[i / bb] * bb + (i % bb)] = 0.0;
}
// shadow slots of a multigraph pattern: the off-diagonal of all but the last parallel face is not stored (Pattern::shadow_slots)
void k_zero_slots(hipStream_t s, double *val, const int32_t *slots, int64_t n, int bb) {
  if (n) hipLaunchKernelGGL(zero_slots_kernel, dim3(grid_for(n * bb)), dim3(256), 0, s, val, slots, n, bb);
}

void k_unit_diag(hipStream_t s, const Pattern &P, double *val, double *r, int64_t n_owned) {
  int64_t ng = P.n - n_owned;
  if (ng <= 0) return;
  hipLaunchKernelGGL(unit_diag_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, P.d_rowptr.p, P.d_col.p, val, r, n_owned, P.n, P.bs);
}

}  // namespace jh
