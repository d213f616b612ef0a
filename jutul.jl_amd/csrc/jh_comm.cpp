// RCCL-over-xGMI communication for the domain-decomposed path (a-16/a-17).
//
// Reference semantics (ext/JutulPartitionedArraysExt): `consistent!(v)` copies owner values to the ghost copies
// on neighbouring ranks (linalg.jl:46, krylov.jl:54,75, interface.jl:200); dots/norms are local sums over OWNED
// entries + a scalar all-reduce.  One process per GPU; the library talks RCCL directly on the context's HIP
// stream so the Krylov loop never returns to the host language between collectives.  librccl is dlopen'ed
// lazily: single-GPU use has no RCCL dependency, and inside a PyTorch process the already-loaded librccl.so.1
// is reused.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "jh_internal.hpp"

namespace jh {

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl &rccl() {
  static Rccl R;
  if (R.h) return R;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names) {
    R.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (R.h) break;
  }
  if (!R.h) JH_THROW(std::string("cannot dlopen librccl: ") + dlerror());
#define JH_SYM(field, name)                                     \
  *(void **)(&R.field) = dlsym(R.h, name);                      \
  if (!R.field) JH_THROW(std::string("librccl lacks ") + name);
  JH_SYM(GetUniqueId, "ncclGetUniqueId");
  JH_SYM(CommInitRank, "ncclCommInitRank");
  JH_SYM(CommDestroy, "ncclCommDestroy");
  JH_SYM(CommCount, "ncclCommCount");
  JH_SYM(AllReduce, "ncclAllReduce");
  JH_SYM(Send, "ncclSend");
  JH_SYM(Recv, "ncclRecv");
  JH_SYM(GroupStart, "ncclGroupStart");
  JH_SYM(GroupEnd, "ncclGroupEnd");
  JH_SYM(GetErrorString, "ncclGetErrorString");
#undef JH_SYM
  return R;
}

#define JH_NCCL(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t _r = (expr);                                                                      \
    if (_r != ncclSuccess) JH_THROW(std::string("RCCL error: ") + rccl().GetErrorString(_r) + " in " #expr); \
  } while (0)

// In-process rendezvous group: the analogue of the reference's DebugPArrayBackend / JuliaPArrayBackend
// (src/ext/partitionedarrays_ext.jl:37-39) -- several ranks live in ONE process (one host thread each, any
// device) and exchange through host memory.  Exercises the same pack/unpack kernels, halo plans, ghost handling
// and reduction placement as the RCCL path; used by tests on a single GPU.
struct LocalGroup {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  std::vector<int> dev;                             // device of every rank's context (-1: planning context), jh_comm_init_local
  std::vector<std::vector<double>> slot;            // per rank contribution (allreduce)
  std::vector<std::vector<std::vector<double>>> box;  // box[src][dst] halo payload
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    uint64_t g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

struct Comm {
  ncclComm_t comm = nullptr;
  LocalGroup *local = nullptr;
  int nranks = 1, rank = 0;
  // mailbox all-reduce (jh_comm_ipc_*): every rank's mailbox mapped into this process
  Mailbox *mail_self = nullptr;
  std::vector<Mailbox *> mail_peer;  // [nranks], mail_peer[rank] == mail_self
  DevBuf<Mailbox *> d_mail_peer;
  uint64_t mail_epoch = 0;  // scalar all-reduces enqueued so far (every rank enqueues the same sequence; see Mailbox)
  bool exclusive_cus = false;  // every rank has compute units of its own (jh_comm_set_exclusive): kernels may wait for peers in all wavefronts
  // what the library has observed about the ranks' devices: -1 nothing yet, 0 all ranks on devices of their own, 1 at least two
  // ranks on one device (PCI bus ids exchanged through the mailboxes by jh_comm_ipc_attach; in-process ranks: their device numbers)
  int shared_device = -1, shared_a = -1, shared_b = -1;
  // time limit of the in-solve waits in 100 MHz ticks (JH_COMM_TIMEOUT_S; default 600 s: ranks may legitimately enter a reduction
  // far apart -- lazy first-solve set-up, JIT compilation on one rank, several ranks time-sharing a GPU -- and a timeout is sticky)
  uint64_t wait_ticks = 60000000000ull;
  bool mail_attached = false, mail_enabled = false;
  // host-language halo backend (jh_comm_set_halo_callback): the packed send buffer is staged to the host and exchanged there
  jh_halo_callback halo_cb = nullptr;
  void *halo_cb_user = nullptr;
  std::vector<double> cb_send, cb_recv;
  volatile MailErr *mail_err = nullptr;  // pinned + mapped: code != 0 after a timed-out wait
  int rccl_ranks = 0;                    // ncclCommCount of the RCCL communicator (0: none)
};

int comm_size(jh_context ctx) { return ctx->comm ? ctx->comm->nranks : 1; }
int comm_rank(jh_context ctx) { return ctx->comm ? ctx->comm->rank : 0; }

void mailbox_allreduce_launch(hipStream_t s, const MailArgs &A, double *p, int n, int op);
size_t mailbox_bytes();
constexpr int MAIL_MAX_RANKS = MAIL_R, MAIL_MAX_VALUES = MAIL_V;

static MailArgs next_mail_args(jh_context ctx, uint64_t timeout_ticks) {
  Comm &c = *ctx->comm;
  MailArgs A;
  A.self = c.mail_self; A.peers = c.d_mail_peer.p; A.rank = c.rank; A.nranks = c.nranks;
  A.epoch = ++c.mail_epoch; A.timeout_ticks = timeout_ticks; A.err = const_cast<MailErr *>(c.mail_err);
  return A;
}
uint64_t comm_mail_epoch(jh_context ctx) { return ctx->comm ? ctx->comm->mail_epoch : 0; }
bool comm_mail_args(jh_context ctx, int n, MailArgs *out) {
  if (!ctx->comm || ctx->comm->nranks == 1 || !ctx->comm->mail_enabled || n > MAIL_MAX_VALUES) return false;
  *out = next_mail_args(ctx, ctx->comm->wait_ticks);
  return true;
}
// timeout_ticks: 100 MHz ticks, 0 = wait like a collective
static void mailbox_allreduce(jh_context ctx, double *p, int n, int op, uint64_t timeout_ticks) {
  mailbox_allreduce_launch(ctx->stream, next_mail_args(ctx, timeout_ticks), p, n, op);
}
// A wait inside a mailbox all-reduce / push halo ran out of time on this rank: fail the running call with the culprit's
// name instead of returning numbers computed from a missing contribution.  Sticky: the communicator is unusable afterwards.
void comm_check_errors(jh_context ctx) {
  if (!ctx->comm || !ctx->comm->mail_err) return;
  const volatile MailErr *e = ctx->comm->mail_err;
  if (e->code == 0) return;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  JH_THROW("rank " + std::to_string(ctx->comm->rank) + " of " + std::to_string(ctx->comm->nranks) + ": " +
           (e->code == 1 ? "mailbox all-reduce" : "push halo exchange") + " timed out after " +
           std::to_string(ctx->comm->wait_ticks / 100000000ull) + " s waiting for rank " + std::to_string(e->peer) + " (epoch " +
           std::to_string(e->epoch) + ", " + std::to_string(e->count) + " timed-out waits so far); the peer process died, hung or "
           "runs a different sequence of collectives.  Set JH_BENCH_NO_MAILBOX=1 / JH_BENCH_NO_PUSH=1 to fall back to RCCL.");
}
// consumer-side all-reduces (PendSum::mail): mailboxes on and every rank owns its compute units; option xrank_consumer 0 / 1
// overrides (-1: as the host declared through jh_comm_set_exclusive)
bool comm_xrank_consumer(jh_context ctx) {
  if (!ctx->comm || ctx->comm->nranks == 1 || !ctx->comm->mail_enabled) return false;
  const int64_t o = ctx->opt.xrank_consumer;
  return o < 0 ? ctx->comm->exclusive_cus : o != 0;
}
int comm_timeouts(jh_context ctx) { return (ctx->comm && ctx->comm->mail_err) ? (int)ctx->comm->mail_err->count : 0; }

// in-stream all-reduce of n doubles living in device memory; no-op without a communicator
void comm_allreduce_dev(jh_context ctx, double *p, int n, int op) {
  if (!ctx->comm || ctx->comm->nranks == 1) return;
  if (ctx->comm->mail_enabled && n <= MAIL_MAX_VALUES) { mailbox_allreduce(ctx, p, n, op, ctx->comm->wait_ticks); return; }
  if (ctx->comm->local) {
    LocalGroup &G = *ctx->comm->local;
    const int r = ctx->comm->rank;
    std::vector<double> mine(n);
    jh::copy_d2h(mine.data(), p, sizeof(double) * n, ctx->stream);
    JH_HIP(hipStreamSynchronize(ctx->stream));
    G.slot[r] = mine;
    G.barrier();
    std::vector<double> out(n, 0.0);
    for (int i = 0; i < n; ++i) {
      double acc = G.slot[0][i];
      for (int q = 1; q < G.n; ++q) acc = (op == 1) ? std::max(acc, G.slot[q][i]) : acc + G.slot[q][i];  // fixed order
      out[i] = acc;
    }
    G.barrier();  // everyone has read the slots
    jh::copy_h2d(p, out.data(), sizeof(double) * n, ctx->stream);
    JH_HIP(hipStreamSynchronize(ctx->stream));
    return;
  }
  if (!ctx->comm->comm) JH_THROW("all-reduce without an RCCL communicator or enabled mailboxes");
  JH_NCCL(rccl().AllReduce(p, p, (size_t)n, ncclFloat64, op == 1 ? ncclMax : ncclSum, ctx->comm->comm, ctx->stream));
}

void halo_push_pack_launch(hipStream_t s, double *const *dst, const double *v, const int32_t *idx, int64_t n, int bs);
void halo_push_finish_launch(hipStream_t s, Mailbox *self, Mailbox *const *peers, const int32_t *nbr, int n_nbr, int rank, uint64_t epoch,
                             const double *landing, double *v, const int32_t *recv_idx, int64_t n_recv, int bs, MailErr *err,
                             uint64_t timeout_ticks);
void halo_pack_launch(hipStream_t s, double *buf, const double *v, const int32_t *idx, int64_t n, int bs);
void halo_unpack_launch(hipStream_t s, double *v, const double *buf, const int32_t *idx, int64_t n, int bs);

// consistent!(v): pack owned values -> grouped ncclSend/ncclRecv per neighbour -> unpack into ghost rows, all on stream s
// Push exchange (see jh_halo.hip).  packed: the producer of v has already stored the boundary rows into the neighbours' landing
// buffers of parity (push_epoch + 1) & 1 (halo_push_targets).
static void halo_push(jh_tpfa d, double *v, int bs, hipStream_t s, bool packed, uint64_t timeout_ticks = ~0ull) {
  auto &H = d->halo;
  Comm &c = *d->ctx->comm;
  if (timeout_ticks == ~0ull) timeout_ticks = c.wait_ticks;
  const uint64_t e = ++H.push_epoch;
  const int par = (int)(e % HALO_S);
  if (H.n_send && !packed) halo_push_pack_launch(s, H.d_push_dst[par].p, v, H.d_send_idx.p, H.n_send, bs);
  halo_push_finish_launch(s, c.mail_self, c.d_mail_peer.p, H.d_nbr.p, (int)H.nbr.size(), c.rank, e, H.landing + par * H.landing_stride, v,
                          H.d_recv_idx.p, H.n_recv, bs, const_cast<MailErr *>(c.mail_err), timeout_ticks);
}
bool halo_fold_args(jh_tpfa d, double *v, HaloFold *out) {
  auto &H = d->halo;
  if (!H.active || !H.push_enabled || !d->ctx->comm) return false;
  Comm &c = *d->ctx->comm;
  if (H.d_ready.n == 0) {
    H.d_ready.alloc(1);
    JH_HIP(hipMemsetAsync(H.d_ready.p, 0, sizeof(unsigned long long), d->ctx->stream));
  }
  const uint64_t e = ++H.push_epoch;
  HaloFold F;
  F.self = c.mail_self; F.peers = c.d_mail_peer.p; F.nbr = H.d_nbr.p; F.n_nbr = (int)H.nbr.size(); F.rank = c.rank;
  F.epoch = e; F.landing = H.landing + (e % HALO_S) * H.landing_stride; F.xg = v; F.recv_idx = H.d_recv_idx.p; F.n_recv = (int)H.n_recv;
  F.interior_rows = (int)std::max<int64_t>(0, d->pat->interior_rows);
  F.ready = H.d_ready.p; F.err = const_cast<MailErr *>(c.mail_err); F.timeout_ticks = c.wait_ticks;
  *out = F;
  return true;
}
// where the producer of the NEXT pushed vector must store send slot k (N doubles each); nullptr when pushing is off
double *const *halo_push_targets(jh_tpfa d) {
  auto &H = d->halo;
  return H.push_enabled ? H.d_push_dst[(H.push_epoch + 1) % HALO_S].p : nullptr;
}

// packed: the producer of v has already written the send buffer (fused ILU(0) apply); push: inside the Krylov loop, use the
// push exchange when it is enabled
static void halo_exchange_on(jh_tpfa d, double *v, int bs, hipStream_t s, bool packed = false, bool push = false) {
  auto &H = d->halo;
  jh_context ctx = d->ctx;
  if (!ctx->comm) JH_THROW("halo exchange without a communicator (jh_comm_init)");
  if (push && H.push_enabled) { halo_push(d, v, bs, s, packed); return; }
  if (H.n_send && !packed) halo_pack_launch(s, H.d_send_buf.p, v, H.d_send_idx.p, H.n_send, bs);
  if (ctx->comm->local) {
    LocalGroup &G = *ctx->comm->local;
    const int me = ctx->comm->rank;
    std::vector<double> hs((size_t)H.n_send * bs), hr((size_t)H.n_recv * bs);
    if (H.n_send) jh::copy_d2h(hs.data(), H.d_send_buf.p, hs.size() * sizeof(double), s);
    JH_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < H.nbr.size(); ++i)
      G.box[me][H.nbr[i]].assign(hs.begin() + H.send_ptr[i] * bs, hs.begin() + H.send_ptr[i + 1] * bs);
    G.barrier();
    for (size_t i = 0; i < H.nbr.size(); ++i) {
      const std::vector<double> &in = G.box[H.nbr[i]][me];
      if ((int64_t)in.size() != (H.recv_ptr[i + 1] - H.recv_ptr[i]) * bs) JH_THROW("halo plan mismatch between ranks");
      std::copy(in.begin(), in.end(), hr.begin() + H.recv_ptr[i] * bs);
    }
    G.barrier();
    if (H.n_recv) {
      jh::copy_h2d(H.d_recv_buf.p, hr.data(), hr.size() * sizeof(double), s);
      halo_unpack_launch(s, v, H.d_recv_buf.p, H.d_recv_idx.p, H.n_recv, bs);
      JH_HIP(hipStreamSynchronize(s));
    }
    return;
  }
  if (ctx->comm->halo_cb) {
    Comm &c = *ctx->comm;
    c.cb_send.resize((size_t)H.n_send * bs);
    c.cb_recv.resize((size_t)H.n_recv * bs);
    if (H.n_send) jh::copy_d2h(c.cb_send.data(), H.d_send_buf.p, c.cb_send.size() * sizeof(double), s);
    JH_HIP(hipStreamSynchronize(s));
    if (c.halo_cb(c.halo_cb_user, c.cb_send.data(), (int64_t)c.cb_send.size(), c.cb_recv.data(), (int64_t)c.cb_recv.size(), bs) != 0)
      JH_THROW("halo callback reported an error");
    if (H.n_recv) {
      jh::copy_h2d(H.d_recv_buf.p, c.cb_recv.data(), c.cb_recv.size() * sizeof(double), s);
      halo_unpack_launch(s, v, H.d_recv_buf.p, H.d_recv_idx.p, H.n_recv, bs);
      JH_HIP(hipStreamSynchronize(s));  // cb_recv is reused by the next exchange
    }
    return;
  }
  if (!ctx->comm->comm) JH_THROW("halo exchange needs an RCCL communicator, the in-process backend or a halo callback");
  Rccl &R = rccl();
  JH_NCCL(R.GroupStart());
  for (size_t i = 0; i < H.nbr.size(); ++i) {
    int64_t ns = H.send_ptr[i + 1] - H.send_ptr[i], nr = H.recv_ptr[i + 1] - H.recv_ptr[i];
    if (ns) JH_NCCL(R.Send(H.d_send_buf.p + H.send_ptr[i] * bs, (size_t)(ns * bs), ncclFloat64, H.nbr[i], ctx->comm->comm, s));
    // direct: the ghosts owned by this neighbour are consecutive device rows -> receive straight into the vector
    double *dst = H.direct_recv ? v + (size_t)H.recv_row0[i] * bs : H.d_recv_buf.p + H.recv_ptr[i] * bs;
    if (nr) JH_NCCL(R.Recv(dst, (size_t)(nr * bs), ncclFloat64, H.nbr[i], ctx->comm->comm, s));
  }
  JH_NCCL(R.GroupEnd());
  if (H.n_recv && !H.direct_recv) halo_unpack_launch(s, v, H.d_recv_buf.p, H.d_recv_idx.p, H.n_recv, bs);
}

void halo_exchange(jh_tpfa d, double *v, int bs, bool packed, bool push) {
  if (!d->halo.active) return;
  halo_exchange_on(d, v, bs, d->ctx->stream, packed, push);
}

// Overlapped form.  begin: everything enqueued on the compute stream so far (in particular the rows that are sent) is
// what the exchange sees; it runs on the context's communication stream while the compute stream goes on with work
// that neither writes v's owned boundary rows nor touches its ghost rows.  end: the compute stream waits for the ghosts.
void halo_exchange_begin(jh_tpfa d, double *v, int bs, bool packed) {
  if (!d->halo.active) return;
  jh_context ctx = d->ctx;
  if (!ctx->comm_stream) {
    ctx->comm_stream = ctx->new_stream();  // (under the context's CU mask, if it has one)
    JH_HIP(hipEventCreateWithFlags(&ctx->ev_halo_ready, hipEventDisableTiming));
    JH_HIP(hipEventCreateWithFlags(&ctx->ev_halo_done, hipEventDisableTiming));
  }
  JH_HIP(hipEventRecord(ctx->ev_halo_ready, ctx->stream));
  JH_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_halo_ready, 0));
  halo_exchange_on(d, v, bs, ctx->comm_stream, packed);
  JH_HIP(hipEventRecord(ctx->ev_halo_done, ctx->comm_stream));
}
void halo_exchange_end(jh_tpfa d) {
  if (!d->halo.active) return;
  jh_context ctx = d->ctx;
  JH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_halo_done, 0));
}

}  // namespace jh

using namespace jh;

extern "C" int32_t jh_comm_unique_id(char *id128) {
  return guard([&] {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    JH_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, 128);
  });
}

extern "C" int32_t jh_comm_init(jh_context ctx, int32_t nranks, int32_t rank, const char *id128) {
  return guard([&] {
    if (!ctx) JH_THROW("null context");
    if (ctx->comm) JH_THROW("communicator already initialised");
    jh::select_device(ctx);
    auto c = std::make_unique<Comm>();
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    JH_NCCL(rccl().CommInitRank(&c->comm, nranks, id, rank));
    int cnt = 0;
    JH_NCCL(rccl().CommCount(c->comm, &cnt));
    if (cnt != nranks) JH_THROW("RCCL communicator has " + std::to_string(cnt) + " ranks, " + std::to_string(nranks) + " were requested");
    c->rccl_ranks = cnt;
    ctx->comm = c.release();
  });
}

extern "C" int32_t jh_comm_local_group_create(int32_t nranks, void **group) {
  return guard([&] {
    if (nranks < 1 || !group) JH_THROW("bad arguments");
    auto *G = new LocalGroup();
    G->n = nranks;
    G->slot.resize(nranks);
    G->dev.assign((size_t)nranks, -2);  // -2: rank not initialised yet
    G->box.assign(nranks, std::vector<std::vector<double>>(nranks));
    *group = G;
  });
}
extern "C" int32_t jh_comm_local_group_destroy(void *group) {
  return guard([&] { delete (LocalGroup *)group; });
}
extern "C" int32_t jh_comm_init_local(jh_context ctx, void *group, int32_t rank) {
  return guard([&] {
    if (!ctx || !group) JH_THROW("null argument");
    // (planning contexts may join: the communicator's bookkeeping is host state; everything that computes refuses them)
    if (ctx->comm) JH_THROW("communicator already initialised");
    auto *G = (LocalGroup *)group;
    if (rank < 0 || rank >= G->n) JH_THROW("rank out of range");
    auto c = std::make_unique<Comm>();
    c->local = G;
    c->nranks = G->n;
    c->rank = rank;
    {
      std::lock_guard<std::mutex> lk(G->m);
      G->dev[(size_t)rank] = ctx->plan_only ? -1 : ctx->device;
    }
    ctx->comm = c.release();
  });
}

// a failed set-up self-test must leave the fallback (RCCL) path usable: forget the recorded time-out
static void clear_mail_error(Comm &c) {
  std::memset((void *)c.mail_err, 0, sizeof(MailErr));
  const unsigned long long zero = 0;
  (void)hipMemcpy(&c.mail_self->abort, &zero, sizeof(zero), hipMemcpyHostToDevice);
}

// What actually carries the data of this rank -- for the host to report and to check (bench.py fails loudly when the ranks it
// asked for are not the ranks that run).  out8: [0] ranks of the communicator, [1] this rank, [2] ranks RCCL counts in its
// communicator (ncclCommCount; 0 = no RCCL communicator), [3] 1 = scalar all-reduces through the mailboxes, [4] 1 = ghost
// exchanges through the host callback, [5] 1 = in-process backend, [6] timed-out waits so far, [7] wait limit in seconds.
extern "C" int32_t jh_comm_info(jh_context ctx, int64_t *out8) {
  return guard([&] {
    if (!ctx || !out8) JH_THROW("null argument");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    out8[0] = 1;
    if (!ctx->comm) return;
    Comm &c = *ctx->comm;
    out8[0] = c.nranks; out8[1] = c.rank; out8[2] = c.rccl_ranks; out8[3] = c.mail_enabled ? (comm_xrank_consumer(ctx) ? 3 : 1) : 0;
    out8[4] = c.halo_cb ? 1 : 0; out8[5] = c.local ? 1 : 0; out8[6] = comm_timeouts(ctx); out8[7] = (int64_t)(c.wait_ticks / 100000000ull);
  });
}
// [0] 1 = the Krylov-loop ghost exchange is the push halo, [1] 1 = receives land directly in the ghost rows, [2] cells sent,
// [3] cells received, [4] neighbour ranks, [5] owned cells
extern "C" int32_t jh_halo_info(jh_tpfa d, int64_t *out6) {
  return guard([&] {
    if (!d || !out6) JH_THROW("null argument");
    const auto &H = d->halo;
    out6[0] = H.push_enabled ? 1 : 0; out6[1] = H.direct_recv ? 1 : 0; out6[2] = H.n_send; out6[3] = H.n_recv;
    out6[4] = (int64_t)H.nbr.size(); out6[5] = H.active ? H.n_owned : d->nc;
  });
}

// ---- mailbox all-reduce: the solver's 1-2 scalar reductions over peer-mapped device memory ------------------------------
extern "C" int32_t jh_comm_init_ipc_only(jh_context ctx, int32_t nranks, int32_t rank) {
  return guard([&] {
    jh::require_device(ctx);
    if (ctx->comm) JH_THROW("communicator already initialised");
    if (nranks < 1 || rank < 0 || rank >= nranks) JH_THROW("bad rank / size");
    auto c = std::make_unique<Comm>();
    c->nranks = nranks;
    c->rank = rank;
    ctx->comm = c.release();
  });
}

extern "C" int32_t jh_comm_set_halo_callback(jh_context ctx, jh_halo_callback fn, void *user) {
  return guard([&] {
    if (!ctx || !ctx->comm) JH_THROW("no communicator");
    if (ctx->comm->local) JH_THROW("the in-process backend exchanges by itself");
    ctx->comm->halo_cb = fn;
    ctx->comm->halo_cb_user = user;
  });
}

extern "C" int32_t jh_comm_ipc_export(jh_context ctx, char *handle64) {
  return guard([&] {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!ctx || !ctx->comm || !handle64) JH_THROW("jh_comm_ipc_export needs an initialised communicator");
    Comm &c = *ctx->comm;
    if (c.local) JH_THROW("the in-process backend has no mailboxes");
    if (c.nranks > MAIL_MAX_RANKS) JH_THROW("mailbox all-reduce supports at most 16 ranks");
    jh::select_device(ctx);
    if (!c.mail_self) {
      // uncached (fine-grained) device memory: peer stores over xGMI become visible without cache maintenance
      JH_HIP(hipExtMallocWithFlags((void **)&c.mail_self, mailbox_bytes(), hipDeviceMallocUncached));
      JH_HIP(hipMemset(c.mail_self, 0, mailbox_bytes()));
      JH_HIP(hipHostMalloc((void **)&c.mail_err, sizeof(MailErr), hipHostMallocMapped | hipHostMallocCoherent));
      std::memset((void *)c.mail_err, 0, sizeof(MailErr));
      c.wait_ticks = (uint64_t)std::max<int64_t>(0, ctx->opt.comm_timeout_ms) * 100000ull;  // 100 MHz ticks; 0 = unbounded
    }
    hipIpcMemHandle_t h;
    JH_HIP(hipIpcGetMemHandle(&h, c.mail_self));
    std::memcpy(handle64, &h, 64);
  });
}

// handles: nranks x 64 bytes in rank order (every rank's jh_comm_ipc_export, gathered by the host).  Maps the peers and runs a
// self-test: 64 all-reduces (sum and max) whose answers every rank can compute.  *ok = 1 if all of them were right.  The
// mailboxes are only used after jh_comm_ipc_enable(ctx, 1) -- call it with the AND of every rank's *ok.
extern "C" int32_t jh_comm_ipc_attach(jh_context ctx, const char *handles, int32_t *ok) {
  return guard([&] {
    if (!ctx || !ctx->comm || !handles || !ok) JH_THROW("null argument");
    Comm &c = *ctx->comm;
    if (!c.mail_self) JH_THROW("jh_comm_ipc_export first");
    jh::select_device(ctx);
    *ok = 0;
    c.mail_peer.assign(c.nranks, nullptr);
    for (int r = 0; r < c.nranks; ++r) {
      if (r == c.rank) { c.mail_peer[r] = c.mail_self; continue; }
      hipIpcMemHandle_t h;
      std::memcpy(&h, handles + (size_t)r * 64, 64);
      void *ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return;  // *ok stays 0: the caller falls back to RCCL
      }
      c.mail_peer[r] = (Mailbox *)ptr;
    }
    c.d_mail_peer.upload(c.mail_peer, ctx->stream);
    c.mail_attached = true;
    // self-test on the solver's scalar area (slots 24..31 are free)
    double *dev = ctx->scalars.p + 24;
    bool good = true;
    for (int t = 0; t < 64 && good; ++t) {
      const int n = 1 + t % MAIL_MAX_VALUES, op = (t / 8) % 2;
      double mine[MAIL_MAX_VALUES], want[MAIL_MAX_VALUES], got[MAIL_MAX_VALUES];
      for (int i = 0; i < n; ++i) {
        auto f = [&](int r) { return (double)((r + 1) * (i + 3) + t) * (((r + t + i) & 1) ? 0.5 : -0.25); };
        mine[i] = f(c.rank);
        double acc = f(0);
        for (int r = 1; r < c.nranks; ++r) acc = op ? std::max(acc, f(r)) : acc + f(r);
        want[i] = acc;
      }
      jh::copy_h2d(dev, mine, sizeof(double) * n, ctx->stream);
      mailbox_allreduce(ctx, dev, n, op, 500000000ull);  // 5 s
      jh::copy_d2h(got, dev, sizeof(double) * n, ctx->stream);
      JH_HIP(hipStreamSynchronize(ctx->stream));
      if (c.mail_err->code) good = false;
      for (int i = 0; i < n; ++i) good = good && got[i] == want[i];
    }
    if (good) {
      // Which ranks sit on which device: every rank contributes a 48-bit digest of its device's PCI bus id (exact in a double) at
      // its own position of a zero vector, sums over the ranks, MAIL_MAX_VALUES ranks per reduction.  Equal digests = one device:
      // jh_comm_set_exclusive(ctx, 1) is then refused unless the context is CU-masked.
      char bus[64] = {0};
      if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), ctx->device) != hipSuccess) {
        // (no bus id: fall back to the device ordinal -- ranks that each see "device 0" then count as sharing one, the safe answer)
        (void)hipGetLastError();
        snprintf(bus, sizeof(bus), "ordinal-%d", ctx->device);
      }
      uint64_t h = 0xcbf29ce484222325ull;
      for (const char *q = bus; *q; ++q) h = (h ^ (uint64_t)(unsigned char)*q) * 0x100000001b3ull;
      const double mine = (double)(((h ^ (h >> 48)) & 0xffffffffffffull) | 1ull);
      std::vector<double> ids((size_t)c.nranks, 0.0);
      for (int r0 = 0; r0 < c.nranks && good; r0 += MAIL_MAX_VALUES) {
        const int n = std::min(MAIL_MAX_VALUES, c.nranks - r0);
        double v[MAIL_MAX_VALUES] = {0};
        if (c.rank >= r0 && c.rank < r0 + n) v[c.rank - r0] = mine;
        jh::copy_h2d(dev, v, sizeof(double) * n, ctx->stream);
        mailbox_allreduce(ctx, dev, n, 0, 500000000ull);
        jh::copy_d2h(v, dev, sizeof(double) * n, ctx->stream);
        JH_HIP(hipStreamSynchronize(ctx->stream));
        if (c.mail_err->code) good = false;
        for (int i = 0; i < n; ++i) ids[(size_t)(r0 + i)] = v[i];
      }
      if (good) {
        c.shared_device = 0;
        for (int a = 0; a < c.nranks && !c.shared_device; ++a)
          for (int b = a + 1; b < c.nranks; ++b)
            if (ids[(size_t)a] == ids[(size_t)b]) { c.shared_device = 1; c.shared_a = a; c.shared_b = b; break; }
      }
    }
    if (!good) clear_mail_error(c);  // the fallback path must stay usable
    *ok = good ? 1 : 0;
  });
}

// 1: every rank of the communicator runs on a device of its own; 0: at least two ranks share one; -1: not observed yet (the
// mailboxes were never attached).  What jh_comm_set_exclusive checks; for hosts that want to decide themselves.
static int devices_distinct(Comm &c) {
  if (c.local) {  // in-process ranks: compare the device numbers the ranks registered
    LocalGroup &G = *c.local;
    std::lock_guard<std::mutex> lk(G.m);
    c.shared_device = -1;
    bool all = true;
    for (int d : G.dev) all = all && d != -2;
    if (all) {
      c.shared_device = 0;
      for (int a = 0; a < G.n && !c.shared_device; ++a)
        for (int b = a + 1; b < G.n; ++b)
          if (G.dev[(size_t)a] == G.dev[(size_t)b]) { c.shared_device = 1; c.shared_a = a; c.shared_b = b; break; }
    }
  }
  return c.shared_device < 0 ? -1 : (c.shared_device ? 0 : 1);
}
extern "C" int32_t jh_comm_devices_distinct(jh_context ctx, int32_t *distinct) {
  return guard([&] {
    if (!ctx || !ctx->comm || !distinct) JH_THROW("jh_comm_devices_distinct needs a communicator");
    *distinct = devices_distinct(*ctx->comm);
  });
}

// exclusive = 1: no two ranks of the communicator share compute units (one process per GPU, or CU-masked streams on one GPU,
// jh_context_set_cu_mask) -- the host knows, the library cannot.  Lets the Krylov loop finish its dot products over the ranks
// inside the consuming kernels and fold the push-halo hand-shake into the product kernel (no reduction / finish launches).
extern "C" int32_t jh_comm_set_exclusive(jh_context ctx, int32_t exclusive) {
  return guard([&] {
    if (!ctx || !ctx->comm) JH_THROW("no communicator");
    if (exclusive != 0) {
      // the liveness assumption, checked against what the library has seen: every wavefront of the chip-filling kernels spins until
      // all peers have pushed, so two ranks on one device without CU masks of their own can keep each other off the chip
      if (devices_distinct(*ctx->comm) == 0 && ctx->ncu >= ctx->ncu_total)
        JH_THROW("jh_comm_set_exclusive(1) refused: ranks " + std::to_string(ctx->comm->shared_a) + " and " + std::to_string(ctx->comm->shared_b) +
                 " of the communicator run on the same device and this context has no CU mask (jh_context_set_cu_mask): kernels that wait "
                 "for a peer in every wavefront would keep that peer off the chip.  Leave it 0 (reduction launches) or give every rank "
                 "compute units of its own.");
    }
    ctx->comm->exclusive_cus = exclusive != 0;
  });
}

namespace jh {
void xr_selftest_launch(hipStream_t s, const MailArgs &A, double v0, double v1, double *out, int nblocks);
}  // namespace jh
// Self-test of the consumer-side all-reduce before jh_comm_set_exclusive(ctx, 1): 16 launches of 256 one-wavefront workgroups in
// which EVERY wavefront collects the peers' sums (time limit 5 s per wait); *ok = 1 if all of them hold the sums in rank order.
// Collective: every rank calls it; enable the consumer-side path only if all ranks report 1 (like jh_comm_ipc_attach).
extern "C" int32_t jh_comm_xrank_selftest(jh_context ctx, int32_t *ok) {
  return guard([&] {
    if (!ctx || !ctx->comm || !ok) JH_THROW("jh_comm_xrank_selftest needs a communicator");
    Comm &c = *ctx->comm;
    *ok = 0;
    if (!c.mail_attached || !c.mail_enabled || c.nranks < 2) return;
    jh::select_device(ctx);
    const int nb = 256;
    DevBuf<double> out;
    out.alloc(2 * nb);
    std::vector<double> got(2 * nb);
    bool good = true;
    for (int t = 0; t < 16 && good; ++t) {
      auto f0 = [&](int r) { return (double)((r + 1) * (t + 3)) * (((r + t) & 1) ? 0.5 : -0.25) + 1e-3 * r; };
      auto f1 = [&](int r) { return 1.0 / (double)(r + t + 1); };
      double w0 = f0(0), w1 = f1(0);
      for (int r = 1; r < c.nranks; ++r) { w0 += f0(r); w1 += f1(r); }  // rank order, like the kernel
      xr_selftest_launch(ctx->stream, next_mail_args(ctx, 500000000ull), f0(c.rank), f1(c.rank), out.p, nb);
      jh::copy_d2h(got.data(), out.p, got.size() * sizeof(double), ctx->stream);
      JH_HIP(hipStreamSynchronize(ctx->stream));
      if (c.mail_err->code) good = false;
      for (int b = 0; b < nb; ++b) good = good && got[2 * b] == w0 && got[2 * b + 1] == w1;
    }
    if (!good) clear_mail_error(c);  // the reduction-launch path must stay usable
    *ok = good ? 1 : 0;
  });
}

extern "C" int32_t jh_comm_ipc_enable(jh_context ctx, int32_t enable) {
  return guard([&] {
    if (!ctx || !ctx->comm) JH_THROW("no communicator");
    if (enable && !ctx->comm->mail_attached) JH_THROW("jh_comm_ipc_attach first");
    ctx->comm->mail_enabled = enable != 0;
  });
}

// ---- push halo set-up ---------------------------------------------------------------------------------------------------
extern "C" int32_t jh_halo_ipc_export(jh_tpfa d, char *handle64) {
  return guard([&] {
    if (!d || !handle64) JH_THROW("null argument");
    auto &H = d->halo;
    if (!H.active) JH_THROW("jh_halo_create first");
    if (!d->ctx->comm || !d->ctx->comm->mail_attached) JH_THROW("the push halo needs attached mailboxes (jh_comm_ipc_attach)");
    jh::select_device(d->ctx);
    if (!H.landing) {
      H.landing_stride = std::max<int64_t>(1, H.n_recv * d->N);
      JH_HIP(hipExtMallocWithFlags((void **)&H.landing, sizeof(double) * HALO_S * H.landing_stride, hipDeviceMallocUncached));
      JH_HIP(hipMemset(H.landing, 0, sizeof(double) * HALO_S * H.landing_stride));
    }
    hipIpcMemHandle_t h;
    JH_HIP(hipIpcGetMemHandle(&h, H.landing));
    std::memcpy(handle64, &h, 64);
  });
}

// nbr_handles: n_nbr x 64 bytes, the landing-buffer handles of the neighbours in halo-plan order.  nbr_offset[i]: first cell
// of MY segment in neighbour i's receive order; nbr_stride[i]: neighbour i's total number of received cells (both gathered
// by the host from the neighbours' halo plans).
extern "C" int32_t jh_halo_ipc_attach(jh_tpfa d, const char *nbr_handles, const int64_t *nbr_offset, const int64_t *nbr_stride, int32_t *ok) {
  return guard([&] {
    if (!d || !ok) JH_THROW("null argument");
    auto &H = d->halo;
    if (!H.landing) JH_THROW("jh_halo_ipc_export first");
    Comm &c = *d->ctx->comm;
    jh::select_device(d->ctx);
    *ok = 0;
    const int nn = (int)H.nbr.size();
    H.peer_landing.assign(nn, nullptr);
    for (int i = 0; i < nn; ++i) {
      if (H.nbr[i] == c.rank) { H.peer_landing[i] = H.landing; continue; }  // self exchange (one-rank proxies)
      hipIpcMemHandle_t h;
      std::memcpy(&h, nbr_handles + (size_t)i * 64, 64);
      void *ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return; }
      H.peer_landing[i] = (double *)ptr;
    }
    for (int par = 0; par < HALO_S; ++par) {
      std::vector<double *> dst((size_t)H.n_send);
      for (int i = 0; i < nn; ++i) {
        const int64_t stride = std::max<int64_t>(1, nbr_stride[i] * d->N);
        for (int64_t k = H.send_ptr[i]; k < H.send_ptr[i + 1]; ++k)
          dst[k] = H.peer_landing[i] + par * stride + (nbr_offset[i] + (k - H.send_ptr[i])) * d->N;
      }
      H.d_push_dst[par].upload(dst, d->ctx->stream);
    }
    H.d_nbr.upload(H.nbr, d->ctx->stream);
    JH_HIP(hipStreamSynchronize(d->ctx->stream));
    H.push_attached = true;
    *ok = 1;
  });
}

// One time-limited push exchange of v; *ok = 1 if every ghost row of v then equals expected_ghosts (n_recv x N doubles in
// receive-list order).  Call on all ranks; enable the push halo only if all of them report 1.
extern "C" int32_t jh_halo_ipc_selftest(jh_tpfa d, jh_vec v, const double *expected_ghosts, int32_t *ok) {
  return guard([&] {
    if (!d || !v || !ok) JH_THROW("null argument");
    auto &H = d->halo;
    *ok = 0;
    if (!H.push_attached) return;
    Comm &c = *d->ctx->comm;
    hipStream_t s = d->ctx->stream;
    halo_push(d, v->d.p, v->bs, s, false, 500000000ull);
    std::vector<double> got((size_t)H.n_recv * v->bs);
    if (H.n_recv) {
      halo_pack_launch(s, H.d_recv_buf.p, v->d.p, H.d_recv_idx.p, H.n_recv, v->bs);  // gather the ghost rows
      jh::copy_d2h(got.data(), H.d_recv_buf.p, got.size() * sizeof(double), s);
    }
    JH_HIP(hipStreamSynchronize(s));
    bool good = c.mail_err->code == 0;
    for (size_t i = 0; i < got.size(); ++i) good = good && got[i] == expected_ghosts[i];
    if (!good) clear_mail_error(c);
    *ok = good ? 1 : 0;
  });
}

extern "C" int32_t jh_halo_ipc_enable(jh_tpfa d, int32_t enable) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    if (enable && !d->halo.push_attached) JH_THROW("jh_halo_ipc_attach first");
    d->halo.push_enabled = enable != 0;
  });
}

extern "C" int32_t jh_comm_finalize(jh_context ctx) {
  return guard([&] {
    if (!ctx || !ctx->comm) return;
    JH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) JH_HIP(hipStreamSynchronize(ctx->comm_stream));
    if (ctx->comm->comm) rccl().CommDestroy(ctx->comm->comm);
    Comm &c = *ctx->comm;
    for (int r = 0; r < (int)c.mail_peer.size(); ++r)
      if (r != c.rank && c.mail_peer[r]) (void)hipIpcCloseMemHandle(c.mail_peer[r]);
    if (c.mail_self) (void)hipFree(c.mail_self);
    if (c.mail_err) (void)hipHostFree((void *)c.mail_err);
    delete ctx->comm;
    ctx->comm = nullptr;
  });
}

extern "C" int32_t jh_allreduce(jh_context ctx, double *values, int32_t n, int32_t op) {
  return guard([&] {
    if (!ctx) JH_THROW("null context");
    if (!ctx->comm || ctx->comm->nranks == 1) return;
    if (n > 16) JH_THROW("jh_allreduce handles at most 16 scalars");
    jh::require_device(ctx);
    jh::select_device(ctx);
    double *dev = ctx->scalars.p + 16;
    jh::copy_h2d(dev, values, sizeof(double) * n, ctx->stream);
    comm_allreduce_dev(ctx, dev, n, op);
    jh::copy_d2h(values, dev, sizeof(double) * n, ctx->stream);
    JH_HIP(hipStreamSynchronize(ctx->stream));
    comm_check_errors(ctx);
  });
}

// Halo plan: host gives, per neighbour rank, the local owned cells to send and the local ghost cells to receive
// (1-based HOST numbering of this rank's discretisation).
extern "C" int32_t jh_halo_create(jh_tpfa d, int64_t n_owned, int32_t n_nbr, const int32_t *nbr_rank, const int64_t *send_ptr,
                                  const int64_t *send_cells, const int64_t *recv_ptr, const int64_t *recv_cells) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    jh::DeviceScope dev(d->ctx);  // (planning contexts: the plan's index tables are built and checked, nothing is allocated)
    const Pattern &P = *d->pat;
    auto &H = d->halo;
    H.n_owned = n_owned;
    H.nbr.assign(nbr_rank, nbr_rank + n_nbr);
    H.send_ptr.assign(send_ptr, send_ptr + n_nbr + 1);
    H.recv_ptr.assign(recv_ptr, recv_ptr + n_nbr + 1);
    H.n_send = H.send_ptr[n_nbr] - H.send_ptr[0];
    H.n_recv = H.recv_ptr[n_nbr] - H.recv_ptr[0];
    if (H.send_ptr[0] != 0 || H.recv_ptr[0] != 0) JH_THROW("send_ptr/recv_ptr must start at 0");
    std::vector<int32_t> si(H.n_send), ri(H.n_recv);
    for (int64_t i = 0; i < H.n_send; ++i) {
      int64_t c = send_cells[i] - 1;
      if (c < 0 || c >= n_owned) JH_THROW("send cell is not an owned cell");
      si[i] = P.iperm.empty() ? (int32_t)c : P.iperm[c];
      // dots, norms and unit_diagonalize! take the owned cells to be the device rows [0, n_owned): true only if the
      // discretisation was created with the same n_owned (jh_tpfa_create keeps the ghosts as the last device rows)
      if (si[i] >= n_owned) JH_THROW("halo plan and discretisation disagree on n_owned: an owned cell is stored behind the owned device rows (pass the same n_owned to jh_tpfa_create)");
    }
    for (int64_t i = 0; i < H.n_recv; ++i) {
      int64_t c = recv_cells[i] - 1;
      if (c < n_owned || c >= d->nc) JH_THROW("recv cell is not a ghost cell");
      ri[i] = P.iperm.empty() ? (int32_t)c : P.iperm[c];
      if (ri[i] < n_owned) JH_THROW("halo plan and discretisation disagree on n_owned: a ghost cell is stored among the owned device rows (pass the same n_owned to jh_tpfa_create)");
    }
    // ghosts grouped by owner (dd.local_subdomain(ghost_order="owner")): every neighbour's receive list is a run of
    // consecutive device rows and the unpack kernel is not needed
    H.recv_row0.assign(n_nbr, 0);
    H.direct_recv = true;
    for (int32_t i = 0; i < n_nbr && H.direct_recv; ++i) {
      if (H.recv_ptr[i + 1] > H.recv_ptr[i]) H.recv_row0[i] = ri[H.recv_ptr[i]];
      for (int64_t k = H.recv_ptr[i] + 1; k < H.recv_ptr[i + 1]; ++k)
        if (ri[k] != ri[k - 1] + 1) { H.direct_recv = false; break; }
    }
    H.send_idx_host = si;
    ++H.epoch;
    hipStream_t s = d->ctx->stream;
    H.d_send_idx.upload(si, s);
    H.d_recv_idx.upload(ri, s);
    H.d_send_buf.alloc(std::max<size_t>(1, (size_t)H.n_send * d->N));
    H.d_recv_buf.alloc(std::max<size_t>(1, (size_t)H.n_recv * d->N));
    jh::stream_sync(s);
    H.active = true;
  });
}
