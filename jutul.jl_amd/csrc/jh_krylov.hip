// BiCGStab (a-14) fully resident on the device + one Newton iteration (perform_step!, simulator.jl:392-455).
//
// Reference: linear_solve!(sys, GenericKrylov(:bicgstab)) (linsolve/krylov.jl:71-182) drives Krylov.jl's
// bicgstab! with N = prec (right, single process) or M = prec (left, distributed path); per iteration
// 2 SpMV + 2 preconditioner applies + 4 dots + 1 norm + ~6 unfused BLAS-1 passes.  Here the scalars
// (rho, alpha, omega, beta) never leave HBM: update kernels recompute them from the reduced dot products,
// BLAS-1 work is fused into three kernels, and only ||r|| is read back once per iteration for the stopping
// test ||r|| <= atol + rtol*||r0||.
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "jh_internal.hpp"

namespace jh {
void ilu_apply(jh_ilu M, const double *b, double *x);
bool ilu_can_fuse_gather(jh_ilu M);
void ilu_apply_fused(jh_ilu M, const IluGather &G, double *x, bool pack = false);
bool ilu_can_fuse_product(jh_ilu M);
void ilu_apply_mul(jh_ilu M, const IluGather &G, const double *b, double *x, double *q, const SpmvDot *dot, const double *done, bool pack);
int ilu_eprod(jh_ilu M, const double *x, double *q, const SpmvDot *dot, const double *done);
void ilu_eprod_refresh(jh_ilu M);
bool ilu_can_pack_halo(jh_ilu M);
bool ilu_can_apply_pack(jh_ilu M);
void ilu_apply_pack(jh_ilu M, const double *b, double *x, int n_owned_rows);
void halo_exchange_begin(jh_tpfa d, double *v, int bs, bool packed = false);
void halo_exchange_end(jh_tpfa d);
void ilu_factor(jh_ilu M);
void halo_exchange(jh_tpfa d, double *v, int bs, bool packed = false, bool push = false);
void comm_check_errors(jh_context ctx);
int comm_size(jh_context ctx);
bool comm_xrank_consumer(jh_context ctx);
}  // namespace jh
using namespace jh;

struct jh_krylov_s {
  jh_context ctx = nullptr;
  jh_csr A = nullptr;
  int64_t len = 0;      // doubles per vector
  int64_t len_dot = 0;  // owned part (dots / norms)
  DevBuf<double> r, p, c, s, q, y, z, d, v, t, xalt;
  int64_t min_its = 1;  // IterativeSolverConfig.min_iterations (krylov.jl:120-131)
  int64_t last_path[6] = {0, 0, 0, 0, 0, 0};  // jh_krylov_last_path
  int64_t launches = 0;  // kernels enqueued by the iteration being enqueued (counted at the launch sites of the loop)
  int64_t cur_it = 0;  // Krylov iteration the launches being enqueued belong to (profiling marks of speculative ones are dropped)
  // GMRES workspace: Krylov basis (grows on demand, Krylov.jl `restart = false`), device/pinned Hessenberg column
  std::vector<std::unique_ptr<DevBuf<double>>> gV;
  DevBuf<double> gh;
  double *gh_host = nullptr;
  size_t gh_cap = 0;
  // optional in-solve profiling with HIP events on the context stream (PrecondWrapper-style time/count,
  // linsolve/krylov.jl:5-25): [0] spmv, [1] preconditioner apply
  bool profiling = false;
  int prof_stride = 1;  // time the launches of every prof_stride-th Krylov iteration only (an event pair costs ~4.5 us of stream time)
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_kind;
  std::vector<int64_t> ev_it;
  size_t ev_used = 0;
  double prof_ms[2] = {0, 0};
  int64_t prof_cnt[2] = {0, 0};
  ~jh_krylov_s() {
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    if (gh_host) (void)hipHostFree(gh_host);
  }
  void mark(int kind, hipStream_t st) {  // call before and after the profiled launch (kind 2: continuation of a kind-0 launch)
    if (!profiling) return;
    if (prof_stride > 1 && cur_it % prof_stride != 1) return;
    if (ev_used == ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      ev_pool.push_back(e);
      ev_kind.push_back(kind);
      ev_it.push_back(0);
    }
    ev_kind[ev_used] = kind;
    ev_it[ev_used] = cur_it;
    (void)hipEventRecord(ev_pool[ev_used++], st);
  }
  void collect(int64_t last_it = INT64_MAX) {  // marks of iterations > last_it were speculative: not counted
    if (!profiling || ev_used < 2) { ev_used = 0; return; }
    (void)hipEventSynchronize(ev_pool[ev_used - 1]);
    for (size_t i = 0; i + 1 < ev_used; i += 2) {
      float ms = 0;
      if (ev_it[i] > last_it) continue;
      if (hipEventElapsedTime(&ms, ev_pool[i], ev_pool[i + 1]) == hipSuccess) {
        if (ev_kind[i] == 2) { prof_ms[0] += ms; continue; }  // second launch of a split SpMV: time only
        prof_ms[ev_kind[i]] += ms;
        prof_cnt[ev_kind[i]]++;
      }
    }
    ev_used = 0;
  }
};

namespace {
// omega = <t,s>/<t,t>.  When the first half-step is already exact (s == 0, e.g. ILU(0) == LU on a chain or a 1-cell
// problem) Krylov.jl's formula is 0/0; the guard makes that case converge (x = x + alpha*y, r = 0) instead of NaN.
__device__ __forceinline__ double bicg_omega(double ts, double tt) { return tt == 0.0 ? 0.0 : ts / tt; }
// scalar slots in ctx->scalars
// (rho, ||r||^2) live in two ping-pong pairs so that one fused reduction can write (rho_next, ||r_next||^2) contiguously
enum { S_PAIR0 = 0 /* rho, rr */, S_CV = 3, S_TS = 4, S_TT = 5, S_PAIR1 = 6 /* rho, rr */, S_ERR = 8 /* ..10 */ };

// s = r - alpha*v, alpha = rho/cv
__global__ void bicg_s_kernel(double *s, const double *r, const double *v, const double *sc, int rho_slot, int64_t n) {
  const double alpha = sc[rho_slot] / sc[S_CV];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s[i] = r[i] - alpha * v[i];
}
__global__ void bicg_publish_kernel(double *sc, int pair_slot, double eps, double *rec, double seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) publish_record(sc, pair_slot, eps, rec, seq);
}
// fused: x_out = x_in + alpha*y + omega*z ; r = s - omega*t ; partial sums of <c,r> and <r,r> over the first nd entries.
// x ping-pongs between two buffers so that the iterate of iteration k survives the speculatively enqueued iteration k+1.
// pend.part != nullptr: (<t,s>, <t,t>) still lie in the partials of the second product -- every wavefront sums them itself
// (PendSum), workgroup 0 stores them to sc[S_TS], sc[S_TT].  NT threads; one partial per workgroup and sum.
template <int NT>
__global__ __launch_bounds__(NT) void bicg_xr_dots_kernel(const double *x_in, double *x_out, double *r, const double *y, const double *z,
                                                          const double *s, const double *t, const double *c, double *sc,
                                                          int rho_slot, int64_t n, int64_t nd, double *part, size_t stride,
                                                          const double *done, PendSum pend) {
  if (done && *done != 0.0) return;
  __shared__ double sm[2 * (NT / 64)];
  double ts = 0.0, tt = 0.0;
  if (pend.part) {
    if (pend.mail.self) {  // several ranks: wavefront 0 of every workgroup collects the peers' sums, the others take them from LDS
      __shared__ double xs2[2];
      if (threadIdx.x < 64) {
        XrRegs xr;
        xr_load(pend.mail, xr);
        pend_sum_wave(pend, ts, tt);
        if (blockIdx.x == 0) xr_push(pend.mail, ts, tt);
        xr_sum(pend.mail, xr, blockIdx.x == 0, ts, tt);
        if (threadIdx.x == 0) { xs2[0] = ts; xs2[1] = tt; }
      }
      __syncthreads();
      ts = xs2[0]; tt = xs2[1];
    } else {
      pend_sum_wave(pend, ts, tt);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[S_TS] = ts; sc[S_TT] = tt; }
  } else {  // (workgroup 0 may be storing the pair when it is pending: read only when it is not)
    ts = sc[S_TS]; tt = sc[S_TT];
  }
  const double alpha = sc[rho_slot] / sc[S_CV];
  const double omega = bicg_omega(ts, tt);
  double d0 = 0.0, d1 = 0.0;
  // two entries per lane and access (16-byte loads and stores; the vectors are 256-byte aligned device allocations)
  const int64_t n2 = n >> 1;
  for (int64_t i2 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i2 < n2; i2 += (int64_t)gridDim.x * blockDim.x) {
    const double2 xv = reinterpret_cast<const double2 *>(x_in)[i2], yv = reinterpret_cast<const double2 *>(y)[i2];
    const double2 zv = reinterpret_cast<const double2 *>(z)[i2], sv = reinterpret_cast<const double2 *>(s)[i2];
    const double2 tv = reinterpret_cast<const double2 *>(t)[i2];
    double2 xo, ro;
    xo.x = (xv.x + alpha * yv.x) + omega * zv.x;  // x_aux = x + alpha*y;  x = x_aux + omega*z
    xo.y = (xv.y + alpha * yv.y) + omega * zv.y;
    ro.x = sv.x - omega * tv.x;
    ro.y = sv.y - omega * tv.y;
    reinterpret_cast<double2 *>(x_out)[i2] = xo;
    reinterpret_cast<double2 *>(r)[i2] = ro;
    const int64_t i = 2 * i2;
    if (i + 1 < nd) {
      const double2 cv = reinterpret_cast<const double2 *>(c)[i2];
      d0 += cv.x * ro.x; d1 += ro.x * ro.x;
      d0 += cv.y * ro.y; d1 += ro.y * ro.y;
    } else if (i < nd) {
      d0 += c[i] * ro.x; d1 += ro.x * ro.x;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd length: the last entry
    const int64_t i = n - 1;
    double xi = x_in[i];
    xi += alpha * y[i];
    xi += omega * z[i];
    x_out[i] = xi;
    const double ri = s[i] - omega * t[i];
    r[i] = ri;
    if (i < nd) { d0 += c[i] * ri; d1 += ri * ri; }
  }
  // block reduction in a fixed order
  constexpr int NWV = NT / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); }
  if (lane == 0) { sm[w] = d0; sm[NWV + w] = d1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a[NWV], b[NWV];
#pragma unroll
    for (int i = 0; i < NWV; ++i) { a[i] = sm[i]; b[i] = sm[NWV + i]; }
#pragma unroll
    for (int m = NWV; m > 1; m >>= 1) {
#pragma unroll
      for (int i = 0; i < m / 2; ++i) { a[i] = a[2 * i] + a[2 * i + 1]; b[i] = b[2 * i] + b[2 * i + 1]; }
    }
    part[blockIdx.x] = a[0];
    part[stride + blockIdx.x] = b[0];
  }
}
// (rho_next, ||r||^2) of the LAST enqueued iteration, whose partials no later kernel consumes: the same sums in the same order as
// a consuming kernel would form them (pend_sum_wave), then the iteration's record.  One wavefront.
__global__ __launch_bounds__(64) void bicg_pend_publish_kernel(PendSum pend, double *sc, const double *done, double eps, double *rec, double seq) {
  if (done && *done != 0.0) return;
  double p0, p1;
  XrRegs xr;
  if (pend.mail.self) xr_load(pend.mail, xr);
  pend_sum_wave(pend, p0, p1);
  if (pend.mail.self) {
    xr_push(pend.mail, p0, p1);
    xr_sum(pend.mail, xr, true, p0, p1);
  }
  if (threadIdx.x == 0) {
    sc[pend.out_slot] = p0;
    sc[pend.out_slot + 1] = p1;
    publish_record(sc, pend.out_slot, eps, rec, seq);
  }
}
// second stage of the reduction above: (rho_next, ||r||^2) -> sc[out_slot..+1]; without a communicator the same launch
// publishes the iteration's record (otherwise bicg_publish_kernel does, after the all-reduce)
__global__ __launch_bounds__(FIN_THREADS) void bicg_reduce_publish_kernel(const double *part, size_t stride, int nparts, double *sc,
                                                                          int out_slot, const double *done, double eps, double *rec,
                                                                          double seq, MailArgs mail) {
  if (done && *done != 0.0) return;
  final_reduce_body<false>(part, stride, nparts, 2, sc + out_slot);
  if (mail.self) mailbox_allreduce_body(mail, sc + out_slot, 2, 0);  // over the ranks, in the same launch
  if (rec && threadIdx.x == 0) publish_record(sc, out_slot, eps, rec, seq);
}
// p = r + beta*(p - omega*v), beta = (rho'/rho)*(alpha/omega)
__global__ void bicg_p_kernel(double *p, const double *r, const double *v, const double *sc, int rho_slot, int rho_next_slot, int64_t n) {
  const double rho = sc[rho_slot], rho_next = sc[rho_next_slot];
  const double alpha = rho / sc[S_CV];
  const double omega = bicg_omega(sc[S_TS], sc[S_TT]);
  const double beta = (rho_next / rho) * (alpha / omega);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double pa = p[i] - omega * v[i];
    p[i] = r[i] + beta * pa;
  }
}
// start of a solve on one rank without left preconditioning: x = 0, r = p = c = b, partial sums of rho = <c,r> = ||r||^2 = <b,b>
__global__ __launch_bounds__(256) void bicg_init_kernel(double *x, double *r, double *p, double *c, const double *b, int64_t n, double *part,
                                                        size_t stride) {
  __shared__ double sm[4];
  double d = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = b[i];
    x[i] = 0.0;
    r[i] = v;
    p[i] = v;
    c[i] = v;
    d += v * v;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
  if (lane == 0) sm[w] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s4 = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    part[blockIdx.x] = s4;           // <c, r>
    part[stride + blockIdx.x] = s4;  // <r, r>
  }
}
inline dim3 vgrid(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048))); }
}  // namespace

extern "C" int32_t jh_krylov_create(jh_csr A, jh_krylov *out) {
  return guard([&] {
    if (!A || !out) JH_THROW("null argument");
    jh::select_device(A->ctx);
    auto K = std::make_unique<jh_krylov_s>();
    K->ctx = A->ctx;
    K->A = A;
    K->len = A->pat->n * A->pat->bs;
    K->len_dot = K->len;
    if (A->disc && A->disc->halo.active) K->len_dot = A->disc->halo.n_owned * A->pat->bs;
    for (DevBuf<double> *b : {&K->r, &K->p, &K->c, &K->s, &K->q, &K->y, &K->z, &K->d}) b->alloc(K->len);
    *out = K.release();
  });
}
extern "C" int32_t jh_krylov_destroy(jh_krylov K) {
  return guard([&] { delete K; });
}

namespace jh {

// Wait (spinning on pinned memory, no stream synchronisation) until the record with sequence number seq has been published.
static void wait_published(jh_context ctx, int rec, double seq, double *out) {
  volatile double *r = ctx->h_pub + rec * JH_PUB_LEN;
  SpinPacer pace;
  for (;;) {
    if (r[15] == seq) break;
    if (pace.due()) {  // the stream must still be busy, otherwise the record was lost (kernel fault)
      comm_check_errors(ctx);    // a peer that never arrived in a mailbox all-reduce / push halo (time-limited waits)
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipSuccess) {
        if (r[15] == seq) break;
        JH_THROW("Krylov iteration record was not published");
      } else if (q != hipErrorNotReady) {
        JH_HIP(q);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int i = 0; i < 9; ++i) out[i] = r[i];
}

// returns status; x must hold len doubles
//
// Host/device protocol: iteration k+1 is enqueued BEFORE the host looks at the outcome of iteration k, so the stream never
// drains between iterations and host-side launch cost (RCCL calls on the distributed path) is hidden.  The iterate
// ping-pongs between x and K->xalt, which keeps x_k intact while the speculative iteration k+1 runs; on convergence the
// device raises sc[S_DONE] and the kernels of the speculative iteration return immediately.  The host follows the solve
// through records in pinned memory (wait_published): no stream synchronisation or D2H copy per iteration.
int bicgstab(jh_krylov K, jh_ilu M, int side, const double *b_in, double *x, double rtol, double atol, int64_t itmax,
             int64_t *iters_out, double *hist, int64_t hist_cap) {
  jh_context ctx = K->ctx;
  hipStream_t st = ctx->stream;
  const Pattern &P = *K->A->pat;
  jh_tpfa disc = K->A->disc;
  const bool dist = disc && disc->halo.active;
  if (dist) K->len_dot = disc->halo.n_owned * P.bs;
  const int64_t n = K->len, nd = K->len_dot;
  const bool left = (side == JH_SIDE_LEFT) && M, right = (side == JH_SIDE_RIGHT) && M;
  if (left && K->v.n == 0) { K->v.alloc(n); K->t.alloc(n); }
  if (K->xalt.n == 0) K->xalt.alloc(n);
  double *X[2] = {x, K->xalt.p};
  double *sc = ctx->scalars.p;
  const double *done = sc + S_DONE;
  JH_HIP(hipMemsetAsync(sc + S_DONE, 0, sizeof(double), st));
  auto prec = [&](double *in, double *out) {
    // parray_preconditioner_apply! (ext/.../linalg.jl:78-88): ghost part of the input zeroed, local apply
    if (dist && n > nd) k_fill(st, in + nd, n - nd, 0.0);
    K->mark(1, st);
    ilu_apply(M, in, out);
    K->mark(1, st);
  };
  auto dot2 = [&](const double *a, const double *bb, const double *c2, const double *d2, int slot) {
    MailArgs ma;
    if (comm_mail_args(ctx, c2 ? 2 : 1, &ma)) { k_dot2(ctx, a, bb, c2, d2, nd, slot, &ma); return; }  // all-reduce in the reduction launch
    k_dot2(ctx, a, bb, c2, d2, nd, slot);
    comm_allreduce_dev(ctx, sc + slot, c2 ? 2 : 1, 0);
  };
  const int64_t rows_dot = nd / P.bs;
  // Right preconditioning with column-scaled pivot-only factors: every product of the loop is A * (M^-1 v), and its in-block part
  // is formed inside the apply (ilu_apply_mul); the out-of-block entries -- ghost columns among them, hence after the ghost
  // exchange -- follow in ilu_eprod.  No SpMV launch, no jagged copy of the matrix.
  const bool want_overlap = ctx->opt.halo_overlap != 0;
  const bool fmul = right && ilu_can_fuse_gather(M) && ilu_can_fuse_product(M) && !(dist && want_overlap);
  const bool fmul_pack = fmul && dist && ilu_can_pack_halo(M);
  // the matrix does not change during the solve: multiply out of its jagged-slice copy when it has one (jh_sell.hip)
  // (the copy itself is enqueued behind the start-of-solve reduction: it runs while the host waits for ||r0||)
  const bool jagged = !fmul && sell_usable(K->A);
  if (fmul) ilu_eprod_refresh(M);
  // x = M^-1 (input), out = A x with the fused dot; event pairs: [1] the apply launch, [0] the out-of-block launch
  auto apply_mul = [&](const IluGather &G, double *bin, double *xv, double *out, const SpmvDot &dot) {
    if (bin && dist && n > nd) k_fill(st, bin + nd, n - nd, 0.0);  // ghost part of the preconditioner's input (linalg.jl:78-88)
    K->mark(1, st);
    ilu_apply_mul(M, G, bin, xv, out, &dot, done, fmul_pack);
    K->mark(1, st);
    if (dist) halo_exchange(disc, xv, P.bs, fmul_pack, true);  // consistent!(x) before the ghost columns are multiplied
    K->mark(0, st);
    const int nparts = ilu_eprod(M, xv, out, &dot, done);
    K->mark(0, st);
    spmv_dot_reduce(ctx, &dot, nparts, done);
  };
  // One rank, fused right preconditioning, jagged product: nobody has to exchange the dots, so the kernel that needs a dot next
  // sums its partials itself (PendSum) -- <c, A y> in the s-update apply, (<t,s>, <t,t>) in the x,r kernel, (rho', ||r||^2) in the
  // next iteration's p-update apply, whose workgroup 0 also publishes the record.  Four one-workgroup launches per iteration gone;
  // the producers run with larger workgroups so that a full-chip launch leaves at most PEND_MAX partials.
  // Several ranks: the same, with the sums over the ranks finished by the consumers too (PendSum::mail) -- where every rank has
  // compute units of its own (comm_xrank_consumer); otherwise the dots go through the reduction launch that carries the all-reduce.
  const bool xrank = comm_xrank_consumer(ctx) && !(dist && want_overlap);
  // (the CSR tile product -- block matrices, rows of more than 8 entries -- feeds the consumers as well: up to PEND_LIMIT partials)
  const bool pend_ok = ctx->opt.consumer_reduce && !fmul && right && ilu_can_fuse_gather(M) && (comm_size(ctx) == 1 || xrank) &&
                       !(dist && want_overlap);
  auto pend_mail = [&](PendSum &ps) { if (xrank && !comm_mail_args(ctx, 2, &ps.mail)) JH_THROW("consumer-side all-reduce without mailboxes"); };
  // ... and the push halo's signal / wait / copy runs inside the product kernel that reads the vector (HaloFold) instead of a
  // finish launch: five launches per iteration as on one rank
  const bool halo_fold = pend_ok && jagged && xrank && dist && disc->halo.push_enabled && ilu_can_pack_halo(M) && P.bs == 1;
  PendSum pend_spmv;  // partials of the last product with a fused dot
  // 16 wavefronts per workgroup: <= 512 partials for the consumers above, and the faster product at every size measured (10M rows
  // 0.160 vs 0.174-0.182 ms with 4, one workgroup per CU; 1.25M rows 26.1 vs 27.4 us) -- also where the dots still go through the
  // reduction launch (several ranks)
  const int spmv_waves = ctx->opt.spmv_waves ? (int)ctx->opt.spmv_waves : 16;
  auto spmv = [&](double *in, double *out, const SpmvDot *dot, bool packed = false) {
    HaloFold hf;
    const bool fold = halo_fold && packed && spmv_waves == 16 && halo_fold_args(disc, in, &hf);
    if (dist && !fold) halo_exchange(disc, in, P.bs, packed, true);  // consistent!(X) before every mul! (ext/.../linalg.jl:46)
    K->mark(0, st);
    if (jagged) {
      // the event pair brackets the product kernel alone (what rocprofv3 reports for it); the second stage of its fused dot,
      // with the all-reduce over the ranks, follows -- or is left to the consuming kernel
      const int nparts = k_spmv_sell(K->A, in, out, 1.0, 0.0, dot, done, false, spmv_waves, fold ? &hf : nullptr);
      K->mark(0, st);
      if (dot && dot->mode) {
        if (pend_ok && nparts <= PEND_LIMIT) {
          pend_spmv.part = ctx->partials.p; pend_spmv.stride = (unsigned)ctx->partial_stride; pend_spmv.nparts = nparts;
          pend_spmv.count = dot->mode == 2 ? 2 : 1; pend_spmv.out_slot = dot->slot;
          pend_mail(pend_spmv);
        } else {
          pend_spmv = PendSum();
          spmv_dot_reduce(ctx, dot, nparts, done);
        }
      }
      return;
    }
    if (pend_ok && dot && dot->mode) {  // the second stage of the fused dot is left to the consuming kernel here, too
      SpmvRange all{0, P.ntiles, 0, false};
      const int nparts = k_spmv(ctx, P, K->A->val.p, in, out, 1.0, 0.0, dot, done, &all);
      K->mark(0, st);
      if (nparts <= PEND_LIMIT) {
        pend_spmv.part = ctx->partials.p; pend_spmv.stride = (unsigned)ctx->partial_stride; pend_spmv.nparts = nparts;
        pend_spmv.count = dot->mode == 2 ? 2 : 1; pend_spmv.out_slot = dot->slot;
        pend_mail(pend_spmv);
      } else {
        pend_spmv = PendSum();
        spmv_dot_reduce(ctx, dot, nparts, done);
      }
      return;
    }
    k_spmv(ctx, P, K->A->val.p, in, out, 1.0, 0.0, dot, done);  // dot->allreduce: summed over the ranks in there
    K->mark(0, st);
  };
  K->cur_it = 0;
  ensure_partials(ctx, 4096);
  if (!dist && !left && comm_size(ctx) == 1) {  // (several ranks without a halo plan still all-reduce every dot)
    // x = 0, r = p = c = b and the partial sums of <b,b> in one pass over b (instead of a fill, four copies and a dot)
    dim3 g = vgrid(n);
    hipLaunchKernelGGL(bicg_init_kernel, g, dim3(256), 0, st, x, K->r.p, K->p.p, K->c.p, b_in, n, ctx->partials.p, ctx->partial_stride);
    k_final_reduce(ctx, (int)g.x, 2, S_PAIR0, false);
  } else {
    k_fill(st, x, n, 0.0);
    k_copy(st, K->c.p, b_in, n);  // scratch copy of b (the ghost zeroing of the preconditioner must not touch b)
    if (dist) halo_exchange(disc, K->c.p, P.bs, false, true);  // consistent!(b) (ext/.../krylov.jl:54); a push exchange where enabled
    if (left) prec(K->c.p, K->r.p); else k_copy(st, K->r.p, K->c.p, n);  // r0 = M^-1 b
    k_copy(st, K->p.p, K->r.p, n);
    if (left) k_copy(st, K->c.p, K->r.p, n);  // c = r0 (without left preconditioning c holds it already)
    // rho = <c,r>, ||r||^2 -> pair 0
    dot2(K->c.p, K->r.p, K->r.p, K->r.p, S_PAIR0);
  }
  double h2[2];
  const double rd_seq = read_scalars_begin(ctx, S_PAIR0, 2);
  if (jagged && !sell_refresh(K->A)) JH_THROW("jagged copy of the matrix failed");
  read_scalars_end(ctx, rd_seq, S_PAIR0, 2, h2);
  double rho = h2[0];
  double rnorm = std::sqrt(h2[1]);
  // min_iterations > 1 (krylov.jl:120-131, 199-206): the reference hands Krylov.jl atol = rtol = 1e-20 and lets a callback
  // stop the solve once ||r_k|| <= atol + rtol*||r_0|| AND k >= min_iterations ("user-requested exit": solved stays false)
  const bool manual = K->min_its > 1;
  const double eps_user = atol + rtol * rnorm;
  const double eps_auto = manual ? 1e-20 + 1e-20 * rnorm : eps_user;
  auto eps_at = [&](int64_t k) { return (manual && k < K->min_its) ? eps_auto : eps_user; };
  if (hist && hist_cap > 0) hist[0] = rnorm;
  int status = 0;
  bool solved = rnorm <= eps_auto;
  if (rho == 0.0 && !solved) status = 2;  // "Breakdown b'c = 0"
  // right preconditioning: the s- and p-updates are fused into the gather phase of the ILU(0) apply
  const bool fuse = right && ilu_can_fuse_gather(M);
  const int ghost_from = dist ? (int)(nd / P.bs) : 0x7fffffff;
  const int lag = ctx->opt.sync_loop ? 0 : 1;
  // Rank-local subdomain: the device order is [interior blocks | boundary blocks | ghost blocks].  Opt-in
  // (option halo_overlap = 1): a fused half-iteration "v = A * N^-1(update)" runs as
  //   compute stream: ILU(0) apply | SpMV(interior tiles) | wait | SpMV(boundary tiles)
  //   comm stream   :              | pack, send/recv, unpack |
  // hiding the ghost exchange of the preconditioned vector (consistent!, linalg.jl:46) behind the interior SpMV.
  // Off by default: on one GPU with a real RCCL self send/recv (tools/overlap_probe.py, the 1.25M-cell share of the 8-GPU
  // run) the two cross-stream event hops (~8 us each) and the second SpMV launch cost more than the 22 us exchange they
  // hide (247 vs 214 us per iteration); also splitting the ILU apply doubled its latency-bound time (261 us).
  const bool overlap = dist && fuse && want_overlap && P.interior_tiles >= 0 && !disc->halo.push_enabled;
  // the fused ILU(0) apply also fills the halo send buffer with the rows neighbouring ranks hold as ghosts
  const bool pack = dist && fuse && ilu_can_pack_halo(M);
  auto fused_half = [&](IluGather &G, double *pv, double *out, const SpmvDot &dot) {
    K->mark(1, st);
    ilu_apply_fused(M, G, pv, pack);
    K->mark(1, st);
    halo_exchange_begin(disc, pv, P.bs, pack);
    SpmvRange r1{0, P.interior_tiles, 0, false};
    K->mark(0, st);
    const int g1 = k_spmv(ctx, P, K->A->val.p, pv, out, 1.0, 0.0, &dot, done, &r1);
    K->mark(0, st);
    halo_exchange_end(disc);
    SpmvRange r2{P.interior_tiles, P.ntiles, g1, true};
    K->mark(2, st);
    k_spmv(ctx, P, K->A->val.p, pv, out, 1.0, 0.0, &dot, done, &r2);
    K->mark(2, st);
  };
  double seq_of[2] = {0, 0};
  double *pend_rec = nullptr, pend_seq = 0, pend_eps = 0;  // record of the previous iteration still to be published (by the next apply)
  int pend_pair = 0;
  PendSum pend_xr;  // partials of the last x,r kernel still to be summed (by the next apply)
  // pair holding (rho, rr) at the start of iteration k; the next one goes to the other pair
  auto pair_of = [](int64_t k) { return (k & 1) ? (int)S_PAIR0 : (int)S_PAIR1; };
  auto enqueue = [&](int64_t k) {
    K->cur_it = k;
    const int rs = pair_of(k), rn = pair_of(k + 1);
    const double *xin = X[(k - 1) & 1];
    double *xout = X[k & 1];
    double *yy = K->p.p;
    bool v_done = false, y_packed = false, z_packed = false;
    if (fuse && k > 1) {  // p = r + beta*(p - omega*q) of the previous iteration, then y = N^-1 p
      IluGather G;
      G.mode = 2; G.r = K->r.p; G.q = K->q.p; G.out = K->p.p; G.sc = sc; G.done = done;
      G.rho_slot = pair_of(k - 1); G.rho_next_slot = rs; G.cv_slot = S_CV; G.ts_slot = S_TS; G.n_owned_rows = ghost_from;
      if (pend_rec) { G.pub_rec = pend_rec; G.pub_seq = pend_seq; G.pub_pair = pend_pair; G.pub_eps = pend_eps; G.sc_rw = sc; pend_rec = nullptr; }
      if (pend_xr.part) { G.pend = pend_xr; G.sc_rw = sc; pend_xr = PendSum(); }
      yy = K->y.p;
      if (fmul) {
        SpmvDot d1{1, K->c.p, S_CV, rows_dot, true};
        apply_mul(G, nullptr, yy, K->q.p, d1);
        v_done = true;
      } else if (overlap) {
        // NB: q is an input of this gather (p-update) and the output of the SpMV, which follows it on the compute stream
        SpmvDot d1{1, K->c.p, S_CV, rows_dot, true};
        fused_half(G, yy, K->q.p, d1);
        v_done = true;
      } else {
        K->mark(1, st);
        ilu_apply_fused(M, G, K->y.p, pack);
        K->mark(1, st);
        y_packed = pack;
      }
    } else if (fmul) {  // first iteration: y = N^-1 p, q = A y
      SpmvDot d1{1, K->c.p, S_CV, rows_dot, true};
      apply_mul(IluGather(), K->p.p, K->y.p, K->q.p, d1);
      yy = K->y.p;
      v_done = true;
    } else if (right && pack && ilu_can_apply_pack(M)) {  // first iteration: y = N^-1 p, ghost input zeroed, send rows leave with it
      K->mark(1, st);
      ilu_apply_pack(M, K->p.p, K->y.p, ghost_from);
      K->mark(1, st);
      yy = K->y.p;
      y_packed = true;
    } else if (right) { prec(K->p.p, K->y.p); yy = K->y.p; }
    double *vv = K->q.p;
    if (v_done) {
    } else if (left) {
      spmv(yy, K->q.p, nullptr);
      prec(K->q.p, K->v.p);
      vv = K->v.p;
      dot2(K->c.p, vv, nullptr, nullptr, S_CV);
    } else {
      SpmvDot d1{1, K->c.p, S_CV, rows_dot, true};  // <c, A y> fused into the SpMV epilogue
      spmv(yy, K->q.p, &d1, y_packed);
    }
    double *zz = K->s.p;
    bool t_done = false;
    if (fuse) {  // s = r - alpha*q fused into z = N^-1 s
      IluGather G;
      G.mode = 1; G.r = K->r.p; G.q = vv; G.out = K->s.p; G.sc = sc; G.done = done;
      G.rho_slot = rs; G.cv_slot = S_CV; G.n_owned_rows = ghost_from;
      if (pend_spmv.part) { G.pend = pend_spmv; G.sc_rw = sc; pend_spmv = PendSum(); }  // <c, A y> of the product just enqueued
      zz = K->z.p;
      if (fmul) {
        SpmvDot d2{2, K->s.p, S_TS, rows_dot, true};
        apply_mul(G, nullptr, zz, K->d.p, d2);
        t_done = true;
      } else if (overlap) {
        SpmvDot d2{2, K->s.p, S_TS, rows_dot, true};
        fused_half(G, zz, K->d.p, d2);
        t_done = true;
      } else {
        K->mark(1, st);
        ilu_apply_fused(M, G, K->z.p, pack);
        K->mark(1, st);
        z_packed = pack;
      }
    } else {
      hipLaunchKernelGGL(bicg_s_kernel, vgrid(n), dim3(256), 0, st, K->s.p, K->r.p, vv, sc, rs, n);
      if (right) { prec(K->s.p, K->z.p); zz = K->z.p; }
    }
    double *tt = K->d.p;
    if (t_done) {
    } else if (left) {
      spmv(zz, K->d.p, nullptr);
      prec(K->d.p, K->t.p);
      tt = K->t.p;
      dot2(tt, K->s.p, tt, tt, S_TS);
    } else {
      SpmvDot d2{2, K->s.p, S_TS, rows_dot, true};  // <t,s>, <t,t> fused
      spmv(zz, K->d.p, &d2, z_packed);
    }
    const double seq = (double)(++ctx->pub_seq);
    seq_of[k & 1] = seq;
    double *rec = ctx->h_pub + (k & 1) * JH_PUB_LEN;
    // (rho_next, ||r||^2) -> the other pair
    if (pend_ok) {
      // 512-thread workgroups, at most 512 of them: the partials are summed by the kernel that needs rho_next -- the next
      // iteration's first apply, which also publishes this iteration's record -- or, behind the last enqueued iteration, by a
      // one-wavefront kernel that forms the same sums in the same order
      const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>(((n >> 1) + 511) / 512, 512)));
      double *xpart = ctx->partials.p + 2 * ctx->partial_stride;  // (the product's partials are still being read by this launch)
      hipLaunchKernelGGL(bicg_xr_dots_kernel<512>, g, dim3(512), 0, st, xin, xout, K->r.p, yy, zz, K->s.p, tt, K->c.p, sc, rs, n, nd,
                         xpart, ctx->partial_stride, done, pend_spmv);
      pend_spmv = PendSum();
      PendSum px;
      px.part = xpart; px.stride = (unsigned)ctx->partial_stride; px.nparts = (int)g.x; px.count = 2; px.out_slot = rn;
      pend_mail(px);
      if (lag == 1 && k < itmax) { pend_xr = px; pend_rec = rec; pend_seq = seq; pend_pair = rn; pend_eps = eps_at(k); }
      else hipLaunchKernelGGL(bicg_pend_publish_kernel, dim3(1), dim3(64), 0, st, px, sc, done, eps_at(k), rec, seq);
      return;
    }
    dim3 g = vgrid(n);
    hipLaunchKernelGGL(bicg_xr_dots_kernel<256>, g, dim3(256), 0, st, xin, xout, K->r.p, yy, zz, K->s.p, tt, K->c.p, sc, rs, n, nd,
                       ctx->partials.p, ctx->partial_stride, done, PendSum());
    MailArgs ma;
    const bool fused_ar = comm_mail_args(ctx, 2, &ma);  // mailboxes on: all-reduce + publish in the reduction launch itself
    // without a communicator the reduction launch also publishes the record
    hipLaunchKernelGGL(bicg_reduce_publish_kernel, dim3(1), dim3(FIN_THREADS), 0, st, ctx->partials.p, ctx->partial_stride, (int)g.x,
                       sc, rn, done, eps_at(k), (ctx->comm && !fused_ar) ? nullptr : rec, seq, fused_ar ? ma : MailArgs());
    if (ctx->comm && !fused_ar) {
      comm_allreduce_dev(ctx, sc + rn, 2, 0);
      // the record needs the all-reduced pair: it is published by the first kernel of the next iteration (fused ILU gather)
      // when there is one, otherwise by a one-thread kernel
      if (fuse && lag == 1 && k < itmax) { pend_rec = rec; pend_seq = seq; pend_pair = rn; pend_eps = eps_at(k); }
      else hipLaunchKernelGGL(bicg_publish_kernel, dim3(1), dim3(64), 0, st, sc, rn, eps_at(k), rec, seq);
    }
    // p-update: deferred into the next iteration's first ILU apply when fused
    if (!fuse) hipLaunchKernelGGL(bicg_p_kernel, vgrid(n), dim3(256), 0, st, K->p.p, K->r.p, vv, sc, rs, rn, n);
  };
  K->last_path[0] = pend_ok; K->last_path[1] = pend_ok && xrank; K->last_path[2] = jagged; K->last_path[3] = fmul;
  K->last_path[4] = dist && disc->halo.push_enabled; K->last_path[5] = halo_fold;
  int64_t it = 0, enq = 0;
  // Slot sets of the mailboxes / landing buffers are chosen by HOST-counted epochs, and the launches of a speculative iteration past
  // convergence consume theirs without executing: two consecutive EXECUTED epochs are then (epochs per iteration x lag + 1) apart
  // and must not meet in one set (MAIL_S / HALO_S, jh_internal.hpp).  Counted on the first iteration, on every rank alike; a loop
  // that gains a fused dot or an exchange fails here instead of corrupting a neighbour's reads silently.
  const uint64_t ar0 = comm_mail_epoch(ctx), hx0 = dist ? disc->halo.push_epoch : 0;
  while (!solved && it < itmax && status == 0) {
    while (enq < std::min<int64_t>(itmax, it + 1 + lag)) {
      enqueue(++enq);
      if (enq == 1 && lag > 0 && comm_size(ctx) > 1) {
        const uint64_t ar = comm_mail_epoch(ctx) - ar0, hx = dist ? disc->halo.push_epoch - hx0 : 0;
        if (ar * (uint64_t)lag + 1 >= (uint64_t)MAIL_S || hx * (uint64_t)lag + 1 >= (uint64_t)HALO_S)
          JH_THROW("BiCGStab: " + std::to_string(ar) + " all-reduce and " + std::to_string(hx) + " halo epochs per iteration with " +
                   std::to_string(lag) + " speculative iteration(s) do not fit the " + std::to_string(MAIL_S) + " / " +
                   std::to_string(HALO_S) + " slot sets (MAIL_S / HALO_S)");
      }
    }
    ++it;
    double h[9];
    wait_published(ctx, (int)(it & 1), seq_of[it & 1], h);
    const double rho_cur = h[pair_of(it)], cv = h[S_CV];
    const double alpha = rho_cur / cv;
    rnorm = std::sqrt(h[pair_of(it + 1) + 1]);
    if (hist && it < hist_cap) hist[it] = rnorm;
    solved = h[8] != 0.0;
    if (alpha == 0.0 || alpha != alpha) status = 2;
  }
  if (solved) status = (manual && rnorm > eps_auto) ? 3 : 0;
  else if (status == 0 && it >= itmax) status = 1;
  if (it & 1) k_copy(st, x, X[1], n);  // iterate of the last accepted iteration
  if (dist) halo_exchange(disc, x, P.bs, false, true);  // consistent!(x) (ext/.../krylov.jl:75)
  K->A->jval_fresh = false;  // the values may change before the next solve
  *iters_out = it;
  K->collect(it);
  JH_HIP(hipGetLastError());
  comm_check_errors(ctx);
  return status;
}

// w -= h*v with h read from device memory (modified Gram-Schmidt step; no host round trip between the dot and the update)
__global__ void gmres_axpy_dev_kernel(double *w, const double *v, const double *h, int64_t n) {
  const double a = *h;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) w[i] -= a * v[i];
}

// GMRES (Krylov.jl 0.9 `gmres!`, third-party, restated from its published algorithm: MGS Arnoldi + Givens rotations,
// restart = false, x0 = 0, stop on |zeta_{k+1}| <= atol + rtol*||r0||; call sites linsolve/krylov.jl:214-218 and
// ext/JutulPartitionedArraysExt/krylov.jl:126-148).  side as in bicgstab.  Returns 0 solved / 1 itmax / 2 breakdown.
int gmres(jh_krylov K, jh_ilu M, int side, const double *b_in, double *x, double rtol, double atol, int64_t itmax,
          int64_t *iters_out, double *hist, int64_t hist_cap) {
  jh_context ctx = K->ctx;
  hipStream_t st = ctx->stream;
  const Pattern &P = *K->A->pat;
  jh_tpfa disc = K->A->disc;
  const bool dist = disc && disc->halo.active;
  if (dist) K->len_dot = disc->halo.n_owned * P.bs;
  const int64_t n = K->len, nd = K->len_dot;
  const bool left = (side == JH_SIDE_LEFT) && M, right = (side == JH_SIDE_RIGHT) && M;
  if (K->v.n == 0) { K->v.alloc(n); K->t.alloc(n); }
  if (K->gh_cap < (size_t)itmax + 4) {
    if (K->gh_host) (void)hipHostFree(K->gh_host);
    K->gh_cap = (size_t)itmax + 4;
    K->gh.alloc(K->gh_cap);
    JH_HIP(hipHostMalloc((void **)&K->gh_host, K->gh_cap * sizeof(double), hipHostMallocDefault));
  }
  auto basis = [&](size_t j) -> double * {
    while (K->gV.size() <= j) { K->gV.emplace_back(new DevBuf<double>()); K->gV.back()->alloc(n); }
    return K->gV[j]->p;
  };
  auto prec = [&](double *in, double *out) {
    if (dist && n > nd) k_fill(st, in + nd, n - nd, 0.0);
    K->mark(1, st);
    ilu_apply(M, in, out);
    K->mark(1, st);
  };
  auto dot_to = [&](const double *a, const double *bb, double *out) {
    k_dot2_to(ctx, a, bb, nullptr, nullptr, nd, out);
    comm_allreduce_dev(ctx, out, 1, 0);
  };
  auto read = [&](int count) {
    JH_HIP(hipMemcpyAsync(K->gh_host, K->gh.p, sizeof(double) * count, hipMemcpyDeviceToHost, st));
    JH_HIP(hipStreamSynchronize(st));
  };
  double *w = K->q.p, *tmp = K->v.p, *pp = K->t.p;
  k_fill(st, x, n, 0.0);
  k_copy(st, tmp, b_in, n);
  if (dist) halo_exchange(disc, tmp, P.bs);
  if (left) prec(tmp, w); else k_copy(st, w, tmp, n);  // r0 = M^-1 b
  dot_to(w, w, K->gh.p);
  read(1);
  const double beta = std::sqrt(K->gh_host[0]);
  const bool manual = K->min_its > 1;  // see bicgstab
  const double eps_user = atol + rtol * beta;
  const double eps_auto = manual ? 1e-20 + 1e-20 * beta : eps_user;
  if (hist && hist_cap > 0) hist[0] = beta;
  int64_t it = 0;
  int status = 0;
  bool solved = beta <= eps_auto;
  double rnorm_last = beta;
  std::vector<double> R, cs, sn, z;  // R packed by columns: column k holds k+1 entries
  std::vector<size_t> colptr(1, 0);
  z.push_back(beta);
  if (!solved) k_axpby(st, basis(0), 1.0 / beta, w, 0.0, n);
  while (!solved && it < itmax && status == 0) {
    const size_t k = (size_t)it;
    ++it;
    double *vk = basis(k);
    double *in = vk;
    if (right) { k_copy(st, tmp, vk, n); prec(tmp, pp); in = pp; }
    else if (dist) { k_copy(st, pp, vk, n); in = pp; }  // the halo exchange writes ghost entries: keep the basis vector intact
    if (dist) halo_exchange(disc, in, P.bs);
    K->mark(0, st);
    k_spmv(ctx, P, K->A->val.p, in, w, 1.0, 0.0);
    K->mark(0, st);
    if (left) { prec(w, tmp); k_copy(st, w, tmp, n); }
    for (size_t i = 0; i <= k; ++i) {  // modified Gram-Schmidt
      dot_to(basis(i), w, K->gh.p + i);
      hipLaunchKernelGGL(gmres_axpy_dev_kernel, vgrid(n), dim3(256), 0, st, w, basis(i), K->gh.p + i, n);
    }
    dot_to(w, w, K->gh.p + k + 1);
    read((int)k + 2);
    std::vector<double> h(K->gh_host, K->gh_host + k + 1);
    const double hbis = std::sqrt(K->gh_host[k + 1]);
    for (size_t i = 0; i < k; ++i) {  // previous rotations
      const double a = cs[i] * h[i] + sn[i] * h[i + 1];
      h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
      h[i] = a;
    }
    const double rr = std::hypot(h[k], hbis);  // sym_givens
    const double c = rr == 0.0 ? 1.0 : h[k] / rr, s2 = rr == 0.0 ? 0.0 : hbis / rr;
    cs.push_back(c);
    sn.push_back(s2);
    h[k] = rr;
    R.insert(R.end(), h.begin(), h.end());
    colptr.push_back(R.size());
    z.push_back(-s2 * z[k]);
    z[k] = c * z[k];
    const double rnorm = std::fabs(z[k + 1]);
    if (hist && it < hist_cap) hist[it] = rnorm;
    rnorm_last = rnorm;
    solved = rnorm <= ((manual && it < K->min_its) ? eps_auto : eps_user);
    if (!solved) {
      if (hbis == 0.0) { status = 2; break; }
      k_axpby(st, basis(k + 1), 1.0 / hbis, w, 0.0, n);
    }
  }
  // y = R \ z, x = N^-1 (V y)
  const size_t m = cs.size();
  std::vector<double> y(m);
  for (size_t jj = m; jj-- > 0;) {
    double acc = z[jj];
    for (size_t l = jj + 1; l < m; ++l) acc -= R[colptr[l] + jj] * y[l];
    y[jj] = acc / R[colptr[jj] + jj];
  }
  k_fill(st, tmp, n, 0.0);
  for (size_t jj = 0; jj < m; ++jj) k_axpby(st, tmp, y[jj], basis(jj), 1.0, n);
  if (right) prec(tmp, x); else k_copy(st, x, tmp, n);
  if (dist) halo_exchange(disc, x, P.bs);
  if (solved) status = (manual && rnorm_last > eps_auto) ? 3 : 0;
  else if (status == 0 && it >= itmax) status = 1;
  *iters_out = it;
  K->collect();
  JH_HIP(hipGetLastError());
  return status;
}

}  // namespace jh

extern "C" int32_t jh_gmres(jh_krylov K, jh_ilu M, int32_t side, jh_vec b, jh_vec x, double rtol, double atol, int64_t itmax,
                            int64_t *iters, int32_t *status, double *hist, int64_t hist_cap) {
  return guard([&] {
    if (!K || !b || !x || !iters || !status) JH_THROW("null argument");
    if (b->len != K->len || x->len != K->len) JH_THROW("dimension mismatch in gmres");
    jh::select_device(K->ctx);
    *status = jh::gmres(K, M, side, b->d.p, x->d.p, rtol, atol, itmax, iters, hist, hist_cap);
  });
}

extern "C" int32_t jh_krylov_set_min_iterations(jh_krylov K, int64_t min_iterations) {
  return guard([&] {
    if (!K) JH_THROW("null handle");
    K->min_its = min_iterations < 1 ? 1 : min_iterations;
  });
}

// Which path the last BiCGStab solve of this workspace took: out6 = [0] 1 = dot products finished by the consuming kernels (no
// reduction launches), [1] 1 = over several ranks as well (mailbox granules), [2] 1 = products out of the jagged-slice copy,
// [3] 1 = products fused into the preconditioner apply, [4] 1 = ghost exchanges of the loop are push halos, [5] 1 = their
// hand-shake runs inside the product kernel (no finish launch).
extern "C" int32_t jh_krylov_last_path(jh_krylov K, int64_t *out6) {
  return guard([&] {
    if (!K || !out6) JH_THROW("null argument");
    for (int i = 0; i < 6; ++i) out6[i] = K->last_path[i];
  });
}

extern "C" int32_t jh_krylov_profile(jh_krylov K, int32_t enable, int32_t reset, double *ms2, int64_t *count2) {
  return guard([&] {
    if (!K) JH_THROW("null handle");
    if (ms2) { ms2[0] = K->prof_ms[0]; ms2[1] = K->prof_ms[1]; }
    if (count2) { count2[0] = K->prof_cnt[0]; count2[1] = K->prof_cnt[1]; }
    if (reset) { K->prof_ms[0] = K->prof_ms[1] = 0; K->prof_cnt[0] = K->prof_cnt[1] = 0; }
    K->profiling = enable != 0;
    K->prof_stride = enable > 1 ? enable : 1;
  });
}

extern "C" int32_t jh_bicgstab(jh_krylov K, jh_ilu M, int32_t side, jh_vec b, jh_vec x, double rtol, double atol,
                               int64_t itmax, int64_t *iters, int32_t *status, double *hist, int64_t hist_cap) {
  return guard([&] {
    if (!K || !b || !x || !iters || !status) JH_THROW("null argument");
    if (b->len != K->len || x->len != K->len) JH_THROW("dimension mismatch in bicgstab");
    jh::select_device(K->ctx);
    *status = jh::bicgstab(K, M, side, b->d.p, x->d.p, rtol, atol, itmax, iters, hist, hist_cap);
  });
}

extern "C" int32_t jh_newton_step(jh_law L, jh_csr A, jh_ilu M, jh_krylov K, jh_vec r, jh_vec dx, double dt, double tol,
                                  int32_t force_solve, double rtol, double atol, int64_t itmax, int32_t side,
                                  jh_newton_report *rep) {
  return guard([&] {
    if (!L || !A || !K || !r || !dx || !rep) JH_THROW("null argument");
    jh_context ctx = L->ctx;
    jh::select_device(ctx);
    hipStream_t st = ctx->stream;
    std::memset(rep, 0, sizeof(*rep));
    hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
    float ms = 0;
    jh_tpfa disc = L->disc;
    const bool dist = disc->halo.active;
    const int64_t n_owned = dist ? disc->halo.n_owned : disc->nc;
    // -- assembly (update_state_dependents! + update_linearized_system!, simulator.jl:404-418)
    JH_HIP(hipEventRecord(e0, st));
    if (dist) halo_exchange(disc, L->X.p, L->N);  // parray_synchronize_primary_variables (interface.jl:189-220)
    k_assemble(L, dt, A, r);
    if (dist) k_unit_diag(st, *A->pat, A->val.p, r->d.p, n_owned);  // post_update_linearized_system! (overloads.jl:270-275)
    JH_HIP(hipEventRecord(e1, st));
    // -- convergence (check_convergence, models.jl:818-883)
    k_absmax_strided(ctx, r->d.p, n_owned, L->N, S_ERR);
    comm_allreduce_dev(ctx, ctx->scalars.p + S_ERR, L->N, 1);
    // force_solve: 0 = solve unless converged (check_before_solve, simulator.jl:435-441); 1 = always solve
    // (iteration <= min_nonlinear_iterations, :484); -1 = assemble + check only (solve = false past max_iter, :566-570).
    // With force_solve = 1 the outcome of the check decides nothing here: it is read at the end of the step, together with the
    // event pairs, instead of idling the device for a host round trip in front of the factorisation (the S_ERR slots are not
    // touched by the solve).
    const bool defer_check = force_solve > 0;
    auto read_check = [&] {
      double err[3] = {0, 0, 0};
      read_scalars(ctx, S_ERR, L->N, err);
      float ms_a = 0;
      JH_HIP(hipEventElapsedTime(&ms_a, e0, e1));
      rep->assembly_ms = ms_a;
      bool conv = true;
      for (int e = 0; e < L->N; ++e) {
        if (e < 2) rep->error[e] = err[e];
        if (!(err[e] < tol)) conv = false;
      }
      rep->converged = conv ? 1 : 0;
      return conv;
    };
    if (!defer_check) {
      const bool conv = read_check();
      if (force_solve < 0 || (conv && force_solve == 0)) return;
    }
    // -- linear solve.  The host does not wait between the phases (every wait is an idle gap on the device): the event pairs are
    // read once at the end of the step
    hipEvent_t *ev = ctx->ev_step;
    // (a failing factorisation / solve -- a communication time-out, say -- must not lose the deferred check: the report carries
    // error / converged / assembly_ms as it would have with the check in front)
    struct DeferredCheck {
      bool armed;
      decltype(read_check) &fn;
      ~DeferredCheck() { if (armed) { try { (void)fn(); } catch (...) {} } }
    } deferred{defer_check, read_check};
    JH_HIP(hipEventRecord(ev[0], st));
    if (M) ilu_factor(M);
    JH_HIP(hipEventRecord(ev[1], st));
    JH_HIP(hipEventRecord(ev[2], st));
    std::vector<double> h((size_t)itmax + 2, 0.0);
    int64_t iters = 0;
    // dx receives the solution x, then is negated in place
    int status = jh::bicgstab(K, M, M ? side : JH_SIDE_NONE, r->d.p, dx->d.p, rtol, atol, itmax, &iters, h.data(), (int64_t)h.size());
    k_negate(st, dx->d.p, dx->d.p, dx->len);  // update_dx_from_vector!: dx = -x (default.jl:444-446)
    JH_HIP(hipEventRecord(ev[3], st));
    rep->linear_iterations = iters;
    rep->linear_status = status;
    rep->lin_res0 = h[0];
    rep->lin_res = h[(size_t)std::min<int64_t>(iters, (int64_t)h.size() - 1)];
    const bool bad = status != 0 && rep->lin_res / rep->lin_res0 > 1.0;
    // -- update (update_primary_variables!, models.jl:928-953)
    if (!bad) {
      JH_HIP(hipEventRecord(ev[4], st));
      k_update_primary(L, dx->d.p, 1.0, L->limits.n ? L->limits.p : nullptr);
      JH_HIP(hipEventRecord(ev[5], st));
    }
    JH_HIP(hipEventSynchronize(bad ? ev[3] : ev[5]));
    if (defer_check) { deferred.armed = false; (void)read_check(); }
    JH_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    rep->precond_ms = ms;
    JH_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
    rep->linear_solve_ms = ms;
    if (bad) JH_THROW("Bad linear solve: residual increased (linsolve/krylov.jl:161-166)");
    JH_HIP(hipEventElapsedTime(&ms, ev[4], ev[5]));
    rep->update_ms = ms;
  });
}
