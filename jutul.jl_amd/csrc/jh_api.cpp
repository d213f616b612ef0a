// C-ABI glue: contexts, vectors, matrices, laws.  See include/jutul_hip.h for the reference seams.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <unordered_map>

#include <map>
#include <mutex>

#include <malloc.h>

#include "jh_internal.hpp"

namespace jh {
thread_local bool tl_plan_only = false;  // jh_internal.hpp, "PLANNING CONTEXTS"
thread_local uint64_t *tl_plan_crc = nullptr;

void halo_exchange(jh_tpfa d, double *v, int bs, bool packed = false, bool push = false);
}
using namespace jh;

// ---- copies of pageable host memory through a page-locked bounce buffer (see jh_internal.hpp) -----------------------------------
namespace jh {
namespace {
// One per host thread at a time: a copy waits for its stream while it owns the buffer, and the stream of one rank may hold a
// kernel that waits for another rank (push halo, mailbox all-reduce) -- ranks that are threads of one process (the in-process test
// backend) must not queue behind each other here.  A thread leases a buffer from a pool on its first copy and hands it back when it
// exits; the buffers themselves live as long as the process (no runtime calls from thread-exit or process-exit destructors).
struct Bounce {
  void *buf = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};  // "the DMA out of / into half i has finished"
  static constexpr size_t CAP = (size_t)16 << 20, HALF = CAP / 2;
  int ev_dev = -1;
  void ensure() {
    if (!buf) JH_HIP(hipHostMalloc(&buf, CAP, hipHostMallocPortable));  // usable with the streams of any device
    int dev = 0;
    JH_HIP(hipGetDevice(&dev));
    if (dev != ev_dev) {  // events belong to a device: a thread that moves to another one gets new ones
      for (auto &e : ev) { if (e) (void)hipEventDestroy(e); JH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
      ev_dev = dev;
    }
  }
};
struct BouncePool {
  std::mutex m;
  std::vector<Bounce *> idle;
};
BouncePool &bounce_pool() {
  static BouncePool *p = new BouncePool();  // never destroyed
  return *p;
}
struct BounceLease {
  Bounce *b = nullptr;
  ~BounceLease() {
    if (!b) return;
    BouncePool &P = bounce_pool();
    std::lock_guard<std::mutex> lk(P.m);
    P.idle.push_back(b);
  }
};
Bounce &bounce() {
  static thread_local BounceLease lease;
  if (!lease.b) {
    BouncePool &P = bounce_pool();
    std::lock_guard<std::mutex> lk(P.m);
    if (!P.idle.empty()) { lease.b = P.idle.back(); P.idle.pop_back(); }
    else lease.b = new Bounce();
  }
  return *lease.b;
}
// Option upload_bounce (process-wide: the copy helpers have no context): 0 hands the caller's pointer to hipMemcpyAsync +
// hipStreamSynchronize directly -- the path on which round 3 saw "Memory access fault by GPU" once in ~10 runs of the GPU suite and
// which no isolated reproducer (tools/micro/upload_lifetime.hip) could break since; kept switchable so that the second copy can
// be measured off and the fault hunted with the real workload (tools/upload_soak.sh).
// (process-wide; JH_UPLOAD_BOUNCE=0 in the environment of the PROCESS sets the initial state: tools/upload_soak.sh)
std::atomic<int> g_upload_bounce{[] { const char *e = getenv("JH_UPLOAD_BOUNCE"); return (e && e[0] == '0') ? 0 : 1; }()};
bool page_locked(const void *host) {
  if (!g_upload_bounce.load(std::memory_order_relaxed)) return true;  // (treated like page-locked memory: copied directly)
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, host) != hipSuccess) {
    (void)hipGetLastError();  // "invalid value" for ordinary host memory: not an error of ours
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
}  // namespace
void copy_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s) {
  if (!bytes) return;
  if (page_locked(src_host)) {
    JH_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    JH_HIP(hipStreamSynchronize(s));
    return;
  }
  Bounce &B = bounce();
  B.ensure();
  // two halves in alternation: the host fills one while the DMA engine drains the other (the host copy dominates -- 8 MB take
  // ~1 ms to copy and 0.15 ms to transfer -- so this hides the transfers, not the copies)
  int k = 0;
  for (size_t off = 0; off < bytes; off += Bounce::HALF, k ^= 1) {
    const size_t len = std::min(Bounce::HALF, bytes - off);
    char *half = (char *)B.buf + (size_t)k * Bounce::HALF;
    if (off >= 2 * Bounce::HALF) JH_HIP(hipEventSynchronize(B.ev[k]));  // the transfer that last read this half
    std::memcpy(half, (const char *)src_host + off, len);
    JH_HIP(hipMemcpyAsync((char *)dst_dev + off, half, len, hipMemcpyHostToDevice, s));
    JH_HIP(hipEventRecord(B.ev[k], s));
  }
  JH_HIP(hipStreamSynchronize(s));  // the caller may free or reuse its array when this returns
}
void copy_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s) {
  if (!bytes) return;
  if (page_locked(dst_host)) {
    JH_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
    JH_HIP(hipStreamSynchronize(s));
    return;
  }
  Bounce &B = bounce();
  B.ensure();
  // the transfer of chunk i+1 runs while the host copies chunk i out of the other half
  const size_t nchunks = (bytes + Bounce::HALF - 1) / Bounce::HALF;
  auto issue = [&](size_t i) {
    const size_t off = i * Bounce::HALF, len = std::min(Bounce::HALF, bytes - off);
    JH_HIP(hipMemcpyAsync((char *)B.buf + (i & 1) * Bounce::HALF, (const char *)src_dev + off, len, hipMemcpyDeviceToHost, s));
    JH_HIP(hipEventRecord(B.ev[i & 1], s));
  };
  issue(0);
  for (size_t i = 0; i < nchunks; ++i) {
    JH_HIP(hipEventSynchronize(B.ev[i & 1]));
    if (i + 1 < nchunks) issue(i + 1);  // (the other half: its previous contents were copied out in the last round)
    const size_t off = i * Bounce::HALF, len = std::min(Bounce::HALF, bytes - off);
    std::memcpy((char *)dst_host + off, (const char *)B.buf + (i & 1) * Bounce::HALF, len);
  }
}
}  // namespace jh

namespace {
inline void check(int32_t rc) {  // nested C-ABI call failed: the message is already in the thread-local slot
  if (rc != 0) {
    char buf[2048];
    jh_last_error(buf, sizeof(buf));
    JH_THROW(std::string(buf));
  }
}
}  // namespace

// ---- options ----------------------------------------------------------------------------------------------------
namespace jh {
// Option setup_heap (PROCESS-WIDE, glibc): the set-up allocates and frees ~4 GB of tables and temporaries per 10M cells, and every
// vector of more than the allocator's mmap threshold is a fresh mapping handed back on free.  1: no block is mmap'ed and the trim
// threshold is raised, so that freed heap memory is kept and later tables reuse pages that are already mapped; 0: mmap'ed blocks
// are allowed again and the free heap memory is returned to the system (malloc_trim).  What it changes for the rest of the process, permanently: glibc stops adapting its mmap / trim thresholds
// dynamically once either has been set by hand (mallopt), and switching the option off does not bring that back -- it sets the
// values the dynamic scheme converges to in a program that has freed large blocks (32 MB / 64 MB).  A HOST decision, hence an
// option, off unless asked for, and never read from JH_OPTIONS (a per-context seed must not flip process-wide state).  Since
// round 6 the large tables are mapped up front (prefault_pages), which removes most of what this option was for.
static void apply_setup_heap(bool on) {
#if defined(__GLIBC__)
  if (on) {
    // (M_MMAP_THRESHOLD cannot do this: glibc rejects values above 32 MB, and the set-up's vectors are 40-320 MB at 10M cells --
    // round 5's mallopt(M_MMAP_THRESHOLD, 1 << 30) was a silent no-op.  With no mmap'ed blocks at all the heap grows to the PEAK of
    // the live tables, 2.0 GB at 10M cells, instead of 4.2 GB of mappings that come and go.)
    mallopt(M_MMAP_MAX, 0);
    mallopt(M_TRIM_THRESHOLD, INT32_MAX);
  } else {
    mallopt(M_MMAP_MAX, 65536);  // glibc's default
    mallopt(M_TRIM_THRESHOLD, 64 << 20);
    malloc_trim(0);
  }
#else
  (void)on;
#endif
}
static std::atomic<int> g_setup_heap{0};
bool Options::set(const char *key, int64_t v) {
  if (std::strcmp(key, "upload_bounce") == 0) g_upload_bounce.store(v != 0, std::memory_order_relaxed);
  if (std::strcmp(key, "setup_heap") == 0 && g_setup_heap.exchange(v != 0) != (v != 0)) apply_setup_heap(v != 0);
  // launch geometries that go into kernels unchanged: only the shapes the kernels are written for (a factor workgroup of fewer
  // than 64 threads has no wavefront to walk its rows; a partial wavefront would share rows with another one)
  if (std::strcmp(key, "ilu_factor_threads") == 0 && v != 0 && v != 64 && v != 128 && v != 256 && v != 512)
    JH_THROW("option ilu_factor_threads: 0 (default), 64, 128, 256 or 512");
#define JH_OPT_SET(name, def) if (std::strcmp(key, #name) == 0) { name = v; return true; }
  JH_OPTION_LIST(JH_OPT_SET)
#undef JH_OPT_SET
  return false;
}
bool Options::get(const char *key, int64_t *v) const {
#define JH_OPT_GET(name, def) if (std::strcmp(key, #name) == 0) { *v = name; return true; }
  JH_OPTION_LIST(JH_OPT_GET)
#undef JH_OPT_GET
  return false;
}
void Options::seed_from_env() {
  const char *e = getenv("JH_OPTIONS");
  if (!e) return;
  std::string str(e);
  size_t pos = 0;
  while (pos < str.size()) {
    size_t end = str.find(',', pos);
    if (end == std::string::npos) end = str.size();
    const std::string item = str.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    const std::string key = item.substr(0, eq);
    int64_t val = 1;
    if (eq != std::string::npos) {  // integers only: "sync_loop=on" must not silently become 0
      const char *b = item.c_str() + eq + 1;
      char *endp = nullptr;
      val = std::strtoll(b, &endp, 10);
      if (endp == b || *endp != '\0') JH_THROW("JH_OPTIONS: value of '" + key + "' is not an integer: '" + std::string(b) + "'");
    }
    // process-wide switches are the host program's to flip, by an explicit jh_context_set_option -- not a side effect of creating
    // a context under some environment
    if (key == "setup_heap" || key == "upload_bounce")
      JH_THROW("JH_OPTIONS: '" + key + "' changes process-wide state and can only be set through jh_context_set_option");
    if (!set(key.c_str(), val)) JH_THROW("JH_OPTIONS: unknown option '" + key + "'");
  }
}
}  // namespace jh
extern "C" int32_t jh_context_set_option(jh_context ctx, const char *key, int64_t value) {
  return guard([&] {
    if (!ctx || !key) JH_THROW("null argument");
    if (!ctx->opt.set(key, value)) JH_THROW(std::string("unknown option '") + key + "'");
  });
}
extern "C" int32_t jh_context_get_option(jh_context ctx, const char *key, int64_t *value) {
  return guard([&] {
    if (!ctx || !key || !value) JH_THROW("null argument");
    if (!ctx->opt.get(key, value)) JH_THROW(std::string("unknown option '") + key + "'");
  });
}

// ---- context --------------------------------------------------------------------------------------------------
extern "C" int32_t jh_context_create(int32_t device_id, jh_context *out) {
  return guard([&] {
    if (!out) JH_THROW("null argument");
    int ndev = 0;
    JH_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) JH_THROW("no such HIP device: " + std::to_string(device_id));
    JH_HIP(hipSetDevice(device_id));
    auto c = std::make_unique<jh_context_s>();
    c->device = device_id;
    c->opt.seed_from_env();
    JH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    JH_HIP(hipDeviceGetAttribute(&c->ncu_total, hipDeviceAttributeMultiprocessorCount, device_id));
    c->ncu = c->ncu_total;
    JH_HIP(hipEventCreate(&c->ev0));
    JH_HIP(hipEventCreate(&c->ev1));
    for (auto &e : c->ev_step) JH_HIP(hipEventCreate(&e));
    c->scalars.alloc(jh::JH_NSCALARS);
    JH_HIP(hipMemsetAsync(c->scalars.p, 0, jh::JH_NSCALARS * sizeof(double), c->stream));
    JH_HIP(hipHostMalloc((void **)&c->h_scalars, jh::JH_NSCALARS * sizeof(double), hipHostMallocDefault));
    JH_HIP(hipHostMalloc((void **)&c->h_pub, 2 * jh::JH_PUB_LEN * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_pub, 0, 2 * jh::JH_PUB_LEN * sizeof(double));
    JH_HIP(hipHostMalloc((void **)&c->h_rd, jh::JH_NSCALARS * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_rd, 0, jh::JH_NSCALARS * sizeof(double));
    JH_HIP(hipStreamSynchronize(c->stream));
    *out = c.release();
  });
}
// A context without a device: plans the set-up tables (jh_internal.hpp, "PLANNING CONTEXTS"); every compute entry point refuses it.
extern "C" int32_t jh_context_create_host(jh_context *out) {
  return guard([&] {
    if (!out) JH_THROW("null argument");
    auto c = std::make_unique<jh_context_s>();
    c->device = -1;
    c->plan_only = true;
    c->opt.seed_from_env();
    *out = c.release();
  });
}
extern "C" int32_t jh_context_plan_checksum(jh_context ctx, int64_t *out) {
  return guard([&] {
    if (!ctx || !out) JH_THROW("null argument");
    if (!ctx->plan_only) JH_THROW("jh_context_plan_checksum is for planning contexts (jh_context_create_host)");
    *out = (int64_t)ctx->plan_crc;
  });
}
extern "C" int32_t jh_context_destroy(jh_context ctx) {
  return guard([&] {
    if (!ctx) return;
    if (ctx->plan_only) { delete ctx; return; }
    (void)hipSetDevice(ctx->device);
    (void)jh_comm_finalize(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto e : ctx->ev_step) if (e) (void)hipEventDestroy(e);
    if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
    if (ctx->h_pub) (void)hipHostFree(ctx->h_pub);
    if (ctx->h_rd) (void)hipHostFree(ctx->h_rd);
    if (ctx->comm_stream) {
      (void)hipStreamSynchronize(ctx->comm_stream);
      (void)hipEventDestroy(ctx->ev_halo_ready);
      (void)hipEventDestroy(ctx->ev_halo_done);
      (void)hipStreamDestroy(ctx->comm_stream);
    }
    ctx->partials.release();
    ctx->scalars.release();
    ctx->stage.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
  });
}
// Restricts the context's stream -- every kernel of the context -- to the compute units [first_cu, first_cu + n_cus) of the
// device's CU-mask bit order (hipExtStreamCreateWithCUMask; on gfx942 / gfx950 consecutive bits alternate over the XCDs, so a
// contiguous range takes the same share of every XCD).  n_cus <= 0 removes the mask.  For several ranks on ONE device (tests,
// proxies of a multi-GPU run): with disjoint ranges no rank's spinning kernel can keep another rank's kernel off the chip.  Call
// it before any other object of the context is created; persistent grids are sized by the CUs the mask leaves.
hipStream_t jh_context_s::new_stream() const {
  hipStream_t ns = nullptr;
  if (ncu >= ncu_total) {
    JH_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
  } else {
    std::vector<uint32_t> mask((size_t)(ncu_total + 31) / 32, 0u);
    for (int i = cu_first; i < cu_first + ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    JH_HIP(hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
  }
  return ns;
}
extern "C" int32_t jh_context_set_cu_mask(jh_context ctx, int32_t first_cu, int32_t n_cus) {
  return guard([&] {
    if (!ctx) JH_THROW("null context");
    jh::select_device(ctx);
    JH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) JH_THROW("jh_context_set_cu_mask after the context's communication stream was created: set the mask first");
    if (n_cus > 0 && (first_cu < 0 || first_cu + n_cus > ctx->ncu_total))
      JH_THROW("CU range outside the device's " + std::to_string(ctx->ncu_total) + " compute units");
    const int old_first = ctx->cu_first, old_n = ctx->ncu;
    ctx->cu_first = n_cus > 0 ? first_cu : 0;
    ctx->ncu = n_cus > 0 ? n_cus : ctx->ncu_total;
    hipStream_t ns = nullptr;
    try { ns = ctx->new_stream(); } catch (...) { ctx->cu_first = old_first; ctx->ncu = old_n; throw; }
    (void)hipStreamDestroy(ctx->stream);
    ctx->stream = ns;
  });
}
extern "C" int32_t jh_synchronize(jh_context ctx) {
  return guard([&] {
    if (!ctx) JH_THROW("null context");
    jh::select_device(ctx);
    JH_HIP(hipStreamSynchronize(ctx->stream));
  });
}
extern "C" int32_t jh_timer_start(jh_context ctx) {
  return guard([&] { jh::require_device(ctx); JH_HIP(hipEventRecord(ctx->ev0, ctx->stream)); });
}
extern "C" int32_t jh_timer_stop_ms(jh_context ctx, double *ms) {
  return guard([&] {
    jh::require_device(ctx);
    JH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    JH_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0;
    JH_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f;
  });
}

// ---- vectors ----------------------------------------------------------------------------------------------------
static jh_vec make_vec(jh_context ctx, std::shared_ptr<Pattern> pat) {
  auto v = std::make_unique<jh_vec_s>();
  v->ctx = ctx;
  v->pat = pat;
  v->bs = pat->bs;
  v->len = pat->n * pat->bs;
  jh::select_device(ctx);
  v->d.alloc(v->len);
  JH_HIP(hipMemsetAsync(v->d.p, 0, v->len * sizeof(double), ctx->stream));
  return v.release();
}
extern "C" int32_t jh_vec_create(jh_tpfa d, jh_vec *out) {
  return guard([&] {
    if (!d || !out) JH_THROW("null argument");
    *out = make_vec(d->ctx, d->pat);
  });
}
extern "C" int32_t jh_vec_create_for(jh_csr A, jh_vec *out) {
  return guard([&] {
    if (!A || !out) JH_THROW("null argument");
    *out = make_vec(A->ctx, A->pat);
  });
}
extern "C" int32_t jh_vec_destroy(jh_vec v) {
  return guard([&] { delete v; });
}
static void upload_cells(jh_context ctx, const Pattern &P, double *dst, const double *host, int64_t n, int bs) {
  jh::select_device(ctx);
  if (P.perm.empty()) {
    jh::copy_h2d(dst, host, n * bs * sizeof(double), ctx->stream);
  } else {
    ctx->ensure_stage(n * bs);
    jh::copy_h2d(ctx->stage.p, host, n * bs * sizeof(double), ctx->stream);
    k_permute_in(ctx->stream, dst, ctx->stage.p, P.d_perm.p, n, bs);
  }
  JH_HIP(hipStreamSynchronize(ctx->stream));
}
static void download_cells(jh_context ctx, const Pattern &P, double *host, const double *src, int64_t n, int bs) {
  jh::select_device(ctx);
  if (P.perm.empty()) {
    jh::copy_d2h(host, src, n * bs * sizeof(double), ctx->stream);
  } else {
    ctx->ensure_stage(n * bs);
    k_permute_out(ctx->stream, ctx->stage.p, src, P.d_perm.p, n, bs);
    jh::copy_d2h(host, ctx->stage.p, n * bs * sizeof(double), ctx->stream);
  }
  JH_HIP(hipStreamSynchronize(ctx->stream));
}
extern "C" int32_t jh_vec_upload(jh_vec v, const double *host) {
  return guard([&] {
    if (!v || !host) JH_THROW("null argument");
    upload_cells(v->ctx, *v->pat, v->d.p, host, v->pat->n, v->bs);
  });
}
extern "C" int32_t jh_vec_download(jh_vec v, double *host) {
  return guard([&] {
    if (!v || !host) JH_THROW("null argument");
    download_cells(v->ctx, *v->pat, host, v->d.p, v->pat->n, v->bs);
  });
}
extern "C" int32_t jh_vec_fill(jh_vec v, double value) {
  return guard([&] { k_fill(v->ctx->stream, v->d.p, v->len, value); });
}
extern "C" int32_t jh_vec_copy(jh_vec dst, jh_vec src) {
  return guard([&] {
    if (dst->len != src->len) JH_THROW("dimension mismatch");
    k_copy(dst->ctx->stream, dst->d.p, src->d.p, dst->len);
  });
}
extern "C" int32_t jh_vec_axpby(jh_vec y, double a, jh_vec x, double b) {
  return guard([&] {
    if (y->len != x->len) JH_THROW("dimension mismatch");
    k_axpby(y->ctx->stream, y->d.p, a, x->d.p, b, y->len);
  });
}
extern "C" int32_t jh_vec_negate_into(jh_vec dx, jh_vec x) {
  return guard([&] {
    if (dx->len != x->len) JH_THROW("dimension mismatch");
    k_negate(dx->ctx->stream, dx->d.p, x->d.p, dx->len);
  });
}
extern "C" int32_t jh_vec_dot(jh_vec a, jh_vec b, double *out) {
  return guard([&] {
    if (a->len != b->len) JH_THROW("dimension mismatch");
    jh::select_device(a->ctx);
    k_dot(a->ctx, a->d.p, b->d.p, a->len, 10);
    *out = read_scalar(a->ctx, 10);
  });
}
extern "C" int32_t jh_vec_length(jh_vec v, int64_t *n) {
  return guard([&] { *n = v->len; });
}

// ---- matrices ---------------------------------------------------------------------------------------------------
extern "C" int32_t jh_csr_create(jh_tpfa d, jh_csr *out) {
  return guard([&] {
    if (!d || !out) JH_THROW("null argument");
    jh::DeviceScope dev(d->ctx);  // (planning contexts: a matrix without values, for jh_ilu0_create's symbolic phase)
    auto A = std::make_unique<jh_csr_s>();
    A->ctx = d->ctx;
    A->pat = d->pat;
    A->disc = d;
    size_t cnt = (size_t)d->nnzb * d->N * d->N;
    A->val.alloc(cnt);
    A->val.zero(d->ctx->stream);
    *out = A.release();
  });
}
extern "C" int32_t jh_csr_create_from_pattern(jh_context ctx, int64_t n, int32_t bs, const int64_t *rowptr,
                                              const int64_t *colidx, const double *nz, jh_csr *out) {
  return guard([&] {
    if (!ctx || !rowptr || !colidx || !out) JH_THROW("null argument");
    if (bs < 1 || bs > 3) JH_THROW("block size must be 1..3");
    if (rowptr[0] != 1) JH_THROW("rowptr must be 1-based");
    jh::DeviceScope dev(ctx);  // (planning contexts: the pattern only)
    auto pat = std::make_shared<Pattern>();
    pat->ctx = ctx;
    pat->n = n;
    pat->bs = bs;
    pat->nnzb = rowptr[n] - 1;
    pat->nnzb_host = pat->nnzb;
    if (pat->nnzb > 2000000000LL) JH_THROW("too many non-zeros for 32-bit device indices");
    pat->rowptr.resize(n + 1);
    pat->col.resize(pat->nnzb);
    pat->diag.assign(n, -1);
    for (int64_t i = 0; i <= n; ++i) pat->rowptr[i] = (int32_t)(rowptr[i] - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int64_t k = rowptr[i] - 1; k < rowptr[i + 1] - 1; ++k) {
        int64_t c = colidx[k] - 1;
        if (c < 0 || c >= n) JH_THROW("column index out of range");
        if (k > rowptr[i] - 1 && colidx[k] <= colidx[k - 1]) JH_THROW("columns must be ascending within a row (mat.jl:73-76)");
        pat->col[k] = (int32_t)c;
        if (c == i) pat->diag[i] = (int32_t)k;
      }
    pat->build_tiles();
    pat->upload();
    auto A = std::make_unique<jh_csr_s>();
    A->ctx = ctx;
    A->pat = pat;
    size_t cnt = (size_t)pat->nnzb * bs * bs;
    A->val.alloc(cnt);
    if (nz && !jh::tl_plan_only) jh::copy_h2d(A->val.p, nz, cnt * sizeof(double), ctx->stream);
    else A->val.zero(ctx->stream);
    jh::stream_sync(ctx->stream);
    *out = A.release();
  });
}
extern "C" int32_t jh_csr_destroy(jh_csr A) {
  return guard([&] { delete A; });
}
extern "C" int32_t jh_csr_sizes(jh_csr A, int64_t *n, int64_t *nnzb, int32_t *bs) {
  return guard([&] {
    if (!A) JH_THROW("null handle");
    if (n) *n = A->pat->n;
    if (nnzb) *nnzb = A->pat->nnzb_host;
    if (bs) *bs = A->pat->bs;
  });
}
extern "C" int32_t jh_csr_set_values(jh_csr A, const double *nz) {
  return guard([&] {
    if (!A || !nz) JH_THROW("null argument");
    jh_context ctx = A->ctx;
    jh::select_device(ctx);
    const Pattern &P = *A->pat;
    int bb = P.bs * P.bs;
    size_t cnt = (size_t)P.nnzb_host * bb;  // the host pattern; shadow slots of a multigraph read the zeroed spare slot behind it
    if (P.nz_hslot.empty()) {
      jh::copy_h2d(A->val.p, nz, cnt * sizeof(double), ctx->stream);
    } else {
      ctx->ensure_stage(cnt + bb);
      jh::copy_h2d(ctx->stage.p, nz, cnt * sizeof(double), ctx->stream);
      JH_HIP(hipMemsetAsync(ctx->stage.p + cnt, 0, bb * sizeof(double), ctx->stream));
      k_gather_blocks(ctx->stream, A->val.p, ctx->stage.p, P.d_nz_hslot.p, P.nnzb, bb, false);
    }
    JH_HIP(hipStreamSynchronize(ctx->stream));
  });
}
extern "C" int32_t jh_csr_get_values(jh_csr A, double *nz) {
  return guard([&] {
    if (!A || !nz) JH_THROW("null argument");
    jh_context ctx = A->ctx;
    jh::select_device(ctx);
    const Pattern &P = *A->pat;
    int bb = P.bs * P.bs;
    size_t cnt = (size_t)P.nnzb_host * bb;
    if (P.nz_hslot.empty()) {
      jh::copy_d2h(nz, A->val.p, cnt * sizeof(double), ctx->stream);
    } else {
      ctx->ensure_stage(cnt + bb);  // + the spare slot the shadow slots of a multigraph scatter into
      k_gather_blocks(ctx->stream, ctx->stage.p, A->val.p, P.d_nz_hslot.p, P.nnzb, bb, true);
      jh::copy_d2h(nz, ctx->stage.p, cnt * sizeof(double), ctx->stream);
    }
    JH_HIP(hipStreamSynchronize(ctx->stream));
  });
}
extern "C" int32_t jh_spmv(jh_csr A, jh_vec x, jh_vec y, double alpha, double beta) {
  return guard([&] {
    if (!A || !x || !y) JH_THROW("null argument");
    int64_t len = A->pat->n * A->pat->bs;
    if (x->len != len || y->len != len) JH_THROW("DimensionMismatch (mat.jl:27-28)");
    if (x == y) JH_THROW("x and y must not alias");
    jh::select_device(A->ctx);
    k_spmv(A->ctx, *A->pat, A->val.p, x->d.p, y->d.p, alpha, beta);
    JH_HIP(hipGetLastError());
  });
}
extern "C" int32_t jh_spmv_jagged(jh_csr A, jh_vec x, jh_vec y, double alpha, double beta) {
  return guard([&] {
    if (!A || !x || !y) JH_THROW("null argument");
    int64_t len = A->pat->n * A->pat->bs;
    if (x->len != len || y->len != len) JH_THROW("DimensionMismatch (mat.jl:27-28)");
    if (x == y) JH_THROW("x and y must not alias");
    jh::select_device(A->ctx);
    const bool keep = A->ctx->opt.jds_keep != 0;  // tools/spmv_probe.py: time the product without the refresh
    if (!(keep && A->jval_fresh) && !sell_refresh(A)) JH_THROW("matrix has no jagged-slice form (block size > 1 or more than 8 entries in a row)");
    k_spmv_sell(A, x->d.p, y->d.p, alpha, beta, nullptr, nullptr);
    if (!keep) A->jval_fresh = false;
    JH_HIP(hipGetLastError());
  });
}
extern "C" int32_t jh_spmv_info(jh_csr A, int64_t *out6) {
  return guard([&] {
    if (!A || !out6) JH_THROW("null argument");
    Pattern &P = *A->pat;
    jh::DeviceScope dev(A->ctx);  // (planning contexts: the jagged layout is built, not uploaded)
    if (!P.jag.built) P.build_jagged();
    int32_t kmax = 0;
    for (int64_t i = 0; i < P.n; ++i) kmax = std::max(kmax, P.rowptr[i + 1] - P.rowptr[i]);
    out6[0] = (P.jag.usable && A->ctx->opt.spmv_jagged) ? 1 : 0;
    out6[1] = (P.jag.usable && P.jag.d_col.n == 0) ? 1 : 0;
    out6[2] = kmax;
    out6[3] = P.jag.usable ? P.jag.nslices : 0;
    out6[4] = P.ntiles;
    out6[5] = P.jag.usable ? (int64_t)P.jag.d_far.n : 0;
  });
}
extern "C" int32_t jh_scale_system(jh_csr A, jh_vec r, int32_t kind, double dt) {
  return guard([&] {
    if (!A || !r) JH_THROW("null argument");
    if (kind == 0) return;
    if (kind != 1 && kind != 2) JH_THROW("scaling must be 0 (:none), 1 (:diagonal) or 2 (:dt)");
    if (r->len != A->pat->n * A->pat->bs) JH_THROW("dimension mismatch");
    jh::select_device(A->ctx);
    k_scale_system(A->ctx->stream, *A->pat, A->val.p, r->d.p, kind, dt);
    JH_HIP(hipGetLastError());
  });
}
extern "C" int32_t jh_unit_diagonalize(jh_csr A, jh_vec r, int64_t n_owned) {
  return guard([&] {
    if (!A || !r) JH_THROW("null argument");
    jh::require_device(A->ctx);
    k_unit_diag(A->ctx->stream, *A->pat, A->val.p, r->d.p, n_owned);
  });
}

// ---- conservation law -----------------------------------------------------------------------------------------------
extern "C" int32_t jh_law_create(jh_tpfa d, int32_t kind, const double *params, jh_law *out) {
  return guard([&] {
    if (!d || !out) JH_THROW("null argument");
    int N = (kind == JH_LAW_TWOPHASE) ? 2 : 1;
    if (kind < 0 || kind > 2) JH_THROW("unknown law kind");
    if (N != d->N) JH_THROW("law needs block_n = " + std::to_string(N) + " but the discretisation has " + std::to_string(d->N));
    jh::select_device(d->ctx);
    auto L = std::make_unique<jh_law_s>();
    L->ctx = d->ctx;
    L->disc = d;
    L->kind = kind;
    L->N = N;
    if (params) std::memcpy(L->par, params, 7 * sizeof(double));
    L->X.alloc((size_t)d->nc * N);
    L->X0.alloc((size_t)d->nc * N);
    L->Tnz.alloc(d->nnzb);
    hipStream_t s = d->ctx->stream;
    JH_HIP(hipMemsetAsync(L->X.p, 0, L->X.n * sizeof(double), s));
    JH_HIP(hipMemsetAsync(L->X0.p, 0, L->X0.n * sizeof(double), s));
    JH_HIP(hipMemsetAsync(L->Tnz.p, 0, L->Tnz.n * sizeof(double), s));
    k_set_diag_data(s, L->Tnz.p, d->pat->d_diag.p, nullptr, nullptr, d->nc, true, 1.0);  // vol = 1 (Poisson default)
    JH_HIP(hipStreamSynchronize(s));
    *out = L.release();
  });
}
extern "C" int32_t jh_law_destroy(jh_law L) {
  return guard([&] { delete L; });
}
extern "C" int32_t jh_law_set_data(jh_law L, int32_t which, const double *host) {
  return guard([&] {
    if (!L || !host) JH_THROW("null argument");
    jh_context ctx = L->ctx;
    jh_tpfa d = L->disc;
    jh::select_device(ctx);
    hipStream_t s = ctx->stream;
    if (which == JH_FACE_TRANS || which == JH_FACE_GDZ) {
      ctx->ensure_stage(std::max<size_t>(d->nf, 1));
      jh::copy_h2d(ctx->stage.p, host, d->nf * sizeof(double), s);
      if (which == JH_FACE_TRANS) {
        k_gather_face_data(s, L->Tnz.p, d->d_nz_face.p, ctx->stage.p, d->nnzb, false);
      } else {
        if (L->gnz.n == 0) { L->gnz.alloc(d->nnzb); JH_HIP(hipMemsetAsync(L->gnz.p, 0, d->nnzb * sizeof(double), s)); }
        k_gather_face_data(s, L->gnz.p, d->d_nz_face.p, ctx->stage.p, d->nnzb, true);
        L->has_gdz = true;
      }
    } else if (which == JH_CELL_VOLUME) {
      ctx->ensure_stage(d->nc);
      jh::copy_h2d(ctx->stage.p, host, d->nc * sizeof(double), s);
      k_set_diag_data(s, L->Tnz.p, d->pat->d_diag.p, d->pat->perm.empty() ? nullptr : d->pat->d_perm.p, ctx->stage.p, d->nc, false, 0.0);
    } else {
      JH_THROW("unknown data id");
    }
    JH_HIP(hipStreamSynchronize(s));
  });
}
extern "C" int32_t jh_law_set_state(jh_law L, const double *X) {
  return guard([&] {
    if (!L || !X) JH_THROW("null argument");
    upload_cells(L->ctx, *L->disc->pat, L->X.p, X, L->disc->nc, L->N);
  });
}
extern "C" int32_t jh_law_set_state0(jh_law L, const double *X0) {
  return guard([&] {
    if (!L || !X0) JH_THROW("null argument");
    upload_cells(L->ctx, *L->disc->pat, L->X0.p, X0, L->disc->nc, L->N);
  });
}
extern "C" int32_t jh_law_get_state(jh_law L, double *X) {
  return guard([&] {
    if (!L || !X) JH_THROW("null argument");
    download_cells(L->ctx, *L->disc->pat, X, L->X.p, L->disc->nc, L->N);
  });
}
extern "C" int32_t jh_law_get_variable(jh_law L, int32_t which, int32_t e, double *out) {
  return guard([&] {
    if (!L || !out) JH_THROW("null argument");
    if (which < 0 || which > 1) JH_THROW("which must be 0 (state) or 1 (state0)");
    if (e < 0 || e >= L->N) JH_THROW("no such primary variable");
    jh_context ctx = L->ctx;
    const Pattern &P = *L->disc->pat;
    const int64_t nc = L->disc->nc;
    jh::select_device(ctx);
    ctx->ensure_stage(nc);
    k_component_out(ctx->stream, ctx->stage.p, which ? L->X0.p : L->X.p, P.perm.empty() ? nullptr : P.d_perm.p, nc, L->N, e);
    jh::copy_d2h(out, ctx->stage.p, nc * sizeof(double), ctx->stream);
    JH_HIP(hipStreamSynchronize(ctx->stream));
  });
}
// The ranges THIS library page-locked.  A range somebody else registered (hipErrorHostMemoryAlreadyRegistered: fully or partly
// page-locked already, which is what the caller asked for) is not ours to release: jh_host_unregister leaves it alone.
namespace {
std::mutex g_reg_mutex;
std::map<void *, size_t> g_registered;
}  // namespace
extern "C" int32_t jh_host_register(void *ptr, int64_t bytes) {
  return guard([&] {
    if (!ptr || bytes <= 0) JH_THROW("bad host range");
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return; }
    JH_HIP(e);
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_registered[ptr] = (size_t)bytes;
  });
}
extern "C" int32_t jh_host_unregister(void *ptr) {
  return guard([&] {
    if (!ptr) JH_THROW("null argument");
    {
      std::lock_guard<std::mutex> lk(g_reg_mutex);
      auto it = g_registered.find(ptr);
      if (it == g_registered.end()) return;  // not pinned by jh_host_register: another component's registration stays
      g_registered.erase(it);
    }
    JH_HIP(hipHostUnregister(ptr));
  });
}
extern "C" int32_t jh_law_update_state0(jh_law L) {
  return guard([&] { k_copy(L->ctx->stream, L->X0.p, L->X.p, (int64_t)L->X.n); });
}
extern "C" int32_t jh_law_reset_state(jh_law L) {
  return guard([&] { k_copy(L->ctx->stream, L->X.p, L->X0.p, (int64_t)L->X.n); });
}
extern "C" int32_t jh_law_set_sources(jh_law L, int64_t n, const int64_t *cells, const double *values) {
  return guard([&] {
    if (!L) JH_THROW("null argument");
    if (n < 0 || (n > 0 && (!cells || !values))) JH_THROW("bad source list");
    jh::select_device(L->ctx);
    // forces are re-applied in every Newton iteration (update_equations_and_apply_forces!): an unchanged list costs nothing
    if (L->src_set && (int64_t)L->h_src_cells.size() == n && std::equal(cells, cells + n, L->h_src_cells.begin()) &&
        std::equal(values, values + n * L->N, L->h_src_vals.begin(), [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0; }))
      return;
    const Pattern &P = *L->disc->pat;
    // pre-sum duplicates: the reference adds every source in sequence (variable_poisson.jl:78-84)
    std::vector<int32_t> cs;
    std::vector<double> vs;
    std::unordered_map<int64_t, int64_t> where;  // host cell -> position in cs (work proportional to the list, not to nc)
    where.reserve((size_t)n * 2);
    for (int64_t i = 0; i < n; ++i) {
      int64_t c = cells[i] - 1;
      if (c < 0 || c >= L->disc->nc) JH_THROW("source cell out of range");
      auto it = where.find(c);
      if (it == where.end()) {
        where.emplace(c, (int64_t)cs.size());
        cs.push_back(P.iperm.empty() ? (int32_t)c : P.iperm[c]);
        for (int e = 0; e < L->N; ++e) vs.push_back(values[i * L->N + e]);
      } else {
        for (int e = 0; e < L->N; ++e) vs[it->second * L->N + e] += values[i * L->N + e];
      }
    }
    JH_HIP(hipStreamSynchronize(L->ctx->stream));  // an assembly in flight may still read the old list
    L->nsrc = (int64_t)cs.size();
    if (L->nsrc) {
      L->src_cell.upload(cs, L->ctx->stream);
      L->src_val.upload(vs, L->ctx->stream);
      JH_HIP(hipStreamSynchronize(L->ctx->stream));
    }
    L->h_src_cells.assign(cells, cells + n);
    L->h_src_vals.assign(values, values + n * L->N);
    L->src_set = true;
  });
}
extern "C" int32_t jh_assemble(jh_law L, double dt, jh_csr A, jh_vec r) {
  return guard([&] {
    if (!L || !A || !r) JH_THROW("null argument");
    if (A->pat != L->disc->pat) JH_THROW("matrix does not belong to the law's discretisation");
    if (r->len != L->disc->nc * L->N) JH_THROW("residual has wrong length");
    if (L->kind != JH_LAW_POISSON && !(dt > 0)) JH_THROW("dt must be positive");
    jh::select_device(L->ctx);
    k_assemble(L, dt, A, r);
    JH_HIP(hipGetLastError());
  });
}
extern "C" int32_t jh_convergence(jh_law L, jh_vec r, int64_t n_owned, double *err) {
  return guard([&] {
    if (!L || !r || !err) JH_THROW("null argument");
    jh::select_device(L->ctx);
    if (n_owned <= 0) n_owned = L->disc->nc;
    k_absmax_strided(L->ctx, r->d.p, n_owned, L->N, 12);
    read_scalars(L->ctx, 12, L->N, err);
  });
}
extern "C" int32_t jh_update_primary(jh_law L, jh_vec dx, double w, const double *limits) {
  return guard([&] {
    if (!L || !dx) JH_THROW("null argument");
    jh::select_device(L->ctx);
    const double *lim_dev = nullptr;
    if (limits) {
      jh::copy_h2d(L->ctx->scalars.p + 16, limits, 5 * L->N * sizeof(double), L->ctx->stream);
      lim_dev = L->ctx->scalars.p + 16;
      JH_HIP(hipStreamSynchronize(L->ctx->stream));
    }
    k_update_primary(L, dx->d.p, w, lim_dev);
    JH_HIP(hipGetLastError());
  });
}
extern "C" int32_t jh_law_set_update_limits(jh_law L, const double *limits) {
  return guard([&] {
    if (!L) JH_THROW("null argument");
    jh::select_device(L->ctx);
    JH_HIP(hipStreamSynchronize(L->ctx->stream));
    if (!limits) { L->limits.release(); return; }
    std::vector<double> h(limits, limits + 5 * L->N);
    L->limits.upload(h, L->ctx->stream);
    JH_HIP(hipStreamSynchronize(L->ctx->stream));
  });
}
extern "C" int32_t jh_increment_norm(jh_law L, jh_vec dx, int64_t n_owned, double *out) {
  return guard([&] {
    if (!L || !dx || !out) JH_THROW("null argument");
    if (dx->len != L->disc->nc * L->N) JH_THROW("increment has wrong length");
    jh::select_device(L->ctx);
    if (n_owned <= 0 || n_owned > L->disc->nc) n_owned = L->disc->nc;
    k_absstats(L->ctx, dx->d.p, nullptr, n_owned, L->N, S_STATS);
    double h[12];
    read_scalars(L->ctx, S_STATS, 4 * L->N, h);
    for (int e = 0; e < L->N; ++e) { out[2 * e] = h[4 * e]; out[2 * e + 1] = h[4 * e + 1]; }
  });
}
extern "C" int32_t jh_law_change_report(jh_law L, int64_t n_owned, double *out) {
  return guard([&] {
    if (!L || !out) JH_THROW("null argument");
    jh::select_device(L->ctx);
    if (n_owned <= 0 || n_owned > L->disc->nc) n_owned = L->disc->nc;
    k_absstats(L->ctx, L->X.p, L->X0.p, n_owned, L->N, S_STATS);
    read_scalars(L->ctx, S_STATS, 4 * L->N, out);
  });
}
extern "C" int32_t jh_halo_exchange(jh_tpfa d, jh_vec v) {
  return guard([&] {
    if (!d || !v) JH_THROW("null argument");
    jh::select_device(d->ctx);
    halo_exchange(d, v->d.p, v->bs);
  });
}
extern "C" int32_t jh_halo_exchange_state(jh_law L) {
  return guard([&] {
    if (!L) JH_THROW("null argument");
    jh::select_device(L->ctx);
    halo_exchange(L->disc, L->X.p, L->N);
  });
}

// ---- scalar matrix layouts at the boundary (a-3/a-4) -------------------------------------------------------------------
namespace {
// 0-based scalar slot of entry (e, d) of the k-th block (of `len`) in block row i; rp = 0-based block rowptr of row i
inline int64_t scalar_slot(int layout, int64_t nnzb, int N, int64_t rp, int64_t len, int64_t k, int e, int d) {
  if (layout == JH_LAYOUT_EQUATION_MAJOR) return (int64_t)e * (nnzb * N) + (int64_t)N * rp + (int64_t)d * len + k;
  return (int64_t)N * N * rp + (int64_t)e * (len * N) + k * N + d;  // entity major
}
void check_layout(int layout) {
  if (layout < 0 || layout > 2) JH_THROW("unknown matrix layout");
}
// host rowptr (0-based) of a matrix: from the discretisation's reference tables or, for raw patterns, the pattern itself
std::vector<int64_t> host_block_rowptr(jh_csr A) {
  const Pattern &P = *A->pat;
  std::vector<int64_t> rp(P.n + 1);
  if (A->disc) {
    A->disc->build_tables();
    for (int64_t i = 0; i <= P.n; ++i) rp[i] = A->disc->h_rowptr[i] - 1;
  } else {
    for (int64_t i = 0; i <= P.n; ++i) rp[i] = P.rowptr[i];
  }
  return rp;
}
}  // namespace

extern "C" int32_t jh_tpfa_get_pattern_layout(jh_tpfa d, int32_t layout, int64_t *rowptr, int64_t *colidx) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    check_layout(layout);
    if (layout == JH_LAYOUT_BLOCK_MAJOR || d->N == 1) { check(jh_tpfa_get_pattern(d, rowptr, colidx)); return; }
    d->build_tables();
    const int N = d->N;
    const int64_t nc = d->nc, nnzb = d->pat->nnzb_host;
    for (int64_t i = 0; i < nc; ++i) {
      const int64_t rp = d->h_rowptr[i] - 1, len = d->h_rowptr[i + 1] - d->h_rowptr[i];
      for (int e = 0; e < N; ++e) {
        const int64_t row = (layout == JH_LAYOUT_EQUATION_MAJOR) ? (int64_t)e * nc + i : (int64_t)N * i + e;
        const int64_t start = scalar_slot(layout, nnzb, N, rp, len, 0, e, 0);
        if (rowptr) { rowptr[row] = start + 1; rowptr[row + 1] = start + len * N + 1; }
        if (colidx)
          for (int64_t k = 0; k < len; ++k)
            for (int dd = 0; dd < N; ++dd) {
              const int64_t j = d->h_colidx[rp + k] - 1;
              const int64_t col = (layout == JH_LAYOUT_EQUATION_MAJOR) ? (int64_t)dd * nc + j : (int64_t)N * j + dd;
              colidx[scalar_slot(layout, nnzb, N, rp, len, k, e, dd)] = col + 1;
            }
      }
    }
  });
}

extern "C" int32_t jh_tpfa_get_positions_layout(jh_tpfa d, int32_t layout, int64_t *pos_acc, int64_t *pos_flux) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    check_layout(layout);
    if (layout == JH_LAYOUT_BLOCK_MAJOR || d->N == 1) { check(jh_tpfa_get_positions(d, pos_acc, pos_flux)); return; }
    d->build_tables();
    const int N = d->N;
    const int64_t NN = (int64_t)N * N, nnzb = d->pat->nnzb_host;
    auto conv = [&](const std::vector<int64_t> &blk, int64_t count, int64_t *out, auto row_of) {
      if (!out) return;
      for (int64_t idx = 0; idx < count; ++idx) {
        const int64_t i = row_of(idx);  // 0-based block row holding the entry
        const int64_t rp = d->h_rowptr[i] - 1, len = d->h_rowptr[i + 1] - d->h_rowptr[i];
        const int64_t p = (blk[NN * idx] - 1) / NN;  // host block slot (block-major position of (1,1))
        for (int e = 0; e < N; ++e)
          for (int dd = 0; dd < N; ++dd) out[NN * idx + (int64_t)e * N + dd] = scalar_slot(layout, nnzb, N, rp, len, p - rp, e, dd) + 1;
      }
    };
    conv(d->pos_acc, d->nc, pos_acc, [&](int64_t c) { return c; });
    conv(d->pos_flux, d->nhf, pos_flux, [&](int64_t k) { return d->hf_other[k] - 1; });  // entry (row = other, col = self)
  });
}

static void values_layout(jh_csr A, int layout, double *nz, bool to_host) {
  const Pattern &P = *A->pat;
  const int N = P.bs;
  const int64_t NN = (int64_t)N * N, nnzb = P.nnzb_host;
  std::vector<double> blk((size_t)nnzb * NN);
  std::vector<int64_t> rp = host_block_rowptr(A);
  if (to_host) check(jh_csr_get_values(A, blk.data()));
  for (int64_t i = 0; i < P.n; ++i) {
    const int64_t len = rp[i + 1] - rp[i];
    for (int64_t k = 0; k < len; ++k)
      for (int e = 0; e < N; ++e)
        for (int dd = 0; dd < N; ++dd) {
          const int64_t s = scalar_slot(layout, nnzb, N, rp[i], len, k, e, dd);
          const int64_t b = (rp[i] + k) * NN + (int64_t)dd * N + e;
          if (to_host) nz[s] = blk[b]; else blk[b] = nz[s];
        }
  }
  if (!to_host) check(jh_csr_set_values(A, blk.data()));
}
extern "C" int32_t jh_csr_get_values_layout(jh_csr A, int32_t layout, double *nz) {
  return guard([&] {
    if (!A || !nz) JH_THROW("null argument");
    check_layout(layout);
    if (layout == JH_LAYOUT_BLOCK_MAJOR || A->pat->bs == 1) { check(jh_csr_get_values(A, nz)); return; }
    values_layout(A, layout, nz, true);
  });
}
extern "C" int32_t jh_csr_set_values_layout(jh_csr A, int32_t layout, const double *nz) {
  return guard([&] {
    if (!A || !nz) JH_THROW("null argument");
    check_layout(layout);
    if (layout == JH_LAYOUT_BLOCK_MAJOR || A->pat->bs == 1) { check(jh_csr_set_values(A, nz)); return; }
    values_layout(A, layout, const_cast<double *>(nz), false);
  });
}
extern "C" int32_t jh_vec_upload_layout(jh_vec v, int32_t layout, const double *host) {
  return guard([&] {
    if (!v || !host) JH_THROW("null argument");
    check_layout(layout);
    if (layout != JH_LAYOUT_EQUATION_MAJOR || v->bs == 1) { check(jh_vec_upload(v, host)); return; }
    const int64_t n = v->pat->n;
    std::vector<double> tmp((size_t)v->len);
    for (int64_t c = 0; c < n; ++c)
      for (int e = 0; e < v->bs; ++e) tmp[c * v->bs + e] = host[(int64_t)e * n + c];
    check(jh_vec_upload(v, tmp.data()));
  });
}
extern "C" int32_t jh_vec_download_layout(jh_vec v, int32_t layout, double *host) {
  return guard([&] {
    if (!v || !host) JH_THROW("null argument");
    check_layout(layout);
    if (layout != JH_LAYOUT_EQUATION_MAJOR || v->bs == 1) { check(jh_vec_download(v, host)); return; }
    const int64_t n = v->pat->n;
    std::vector<double> tmp((size_t)v->len);
    check(jh_vec_download(v, tmp.data()));
    for (int64_t c = 0; c < n; ++c)
      for (int e = 0; e < v->bs; ++e) host[(int64_t)e * n + c] = tmp[c * v->bs + e];
  });
}
