// Internal declarations of libjutul_hip.so (gfx950 only).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <atomic>
#include <thread>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/jutul_hip.h"

namespace jh {

// ---- error handling -------------------------------------------------------------------------------------
void set_error(const std::string &msg);
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
#define JH_THROW(msg) throw ::jh::Error(std::string(msg))
#define JH_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      throw ::jh::Error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                        std::to_string(__LINE__) + " (" #expr ")");                                     \
  } while (0)

template <class F>
static inline int32_t guard(F &&f) noexcept {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    set_error(e.what());
    return -1;
  } catch (...) {
    set_error("unknown error");
    return -2;
  }
}

// ---- host-side parallel loop for the set-up phases (integer work over independent rows / blocks) ------------------------------
// fn(begin, end) on contiguous index ranges, one per hardware thread (at most 64); exceptions are rethrown on the caller.
// Cores the set-up may use: hardware threads, capped by the cgroup CPU quota (a container that shows 256 logical CPUs under a
// 16-CPU quota throttles 64 busy threads instead of running them).
static inline int64_t setup_cores() {
  static const int64_t cached = [] {
    int64_t nt = (int64_t)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      long long period = 0;
      if (fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && q[0] >= '0' && q[0] <= '9') {
        const long long quota = atoll(q);
        if (quota > 0) nt = std::min<int64_t>(nt, std::max<int64_t>(1, (quota + period - 1) / period));
      }
      fclose(f);
    }
    return std::max<int64_t>(1, nt);
  }();
  return cached;
}
// host threads of the set-up: setup_cores(), or JH_SETUP_THREADS; at most 64
static inline int setup_threads() {
  int64_t nt = setup_cores();
  if (const char *e = getenv("JH_SETUP_THREADS")) nt = atoi(e);
  return (int)std::max<int64_t>(1, std::min<int64_t>(nt, 64));
}
template <class F>
static inline void parallel_ranges(int64_t n, int64_t min_per_thread, F &&fn) {
  int64_t nt = setup_threads();
  nt = std::max<int64_t>(1, std::min<int64_t>(nt, n / std::max<int64_t>(1, min_per_thread)));
  if (nt <= 1) { fn((int64_t)0, n); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  for (int64_t t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      try { fn(n * t / nt, n * (t + 1) / nt); } catch (...) { err[(size_t)t] = std::current_exception(); }
    });
  for (auto &x : th) x.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);
}

// fn(t, nt) on exactly nt threads (the caller is thread 0); exceptions are rethrown on the caller
template <class F>
static inline void parallel_team(int nt, F &&fn) {
  if (nt <= 1) { fn(0, 1); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  for (int t = 1; t < nt; ++t)
    th.emplace_back([&, t] {
      try { fn(t, nt); } catch (...) { err[(size_t)t] = std::current_exception(); }
    });
  try { fn(0, nt); } catch (...) { err[0] = std::current_exception(); }
  for (auto &x : th) x.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);
}
// Stable counting sort on the set-up's threads: out[start[key[i]] ...] = i for i = 0 .. n-1 in ascending i per key; start has nkeys + 1
// entries (filled here).  The same result as the serial two-pass loop, for any thread count.
static inline void counting_sort_indices(const int32_t *key, int64_t n, int64_t nkeys, std::vector<int32_t> &start, int32_t *out) {
  int nt = setup_threads();
  if (n < (1 << 18) || (int64_t)nt * nkeys > n) nt = 1;
  start.assign((size_t)nkeys + 1, 0);
  if (nt == 1) {
    for (int64_t i = 0; i < n; ++i) start[(size_t)key[i] + 1]++;
    for (int64_t k = 0; k < nkeys; ++k) start[(size_t)k + 1] += start[(size_t)k];
    std::vector<int32_t> cur(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < n; ++i) out[cur[(size_t)key[i]]++] = (int32_t)i;
    return;
  }
  std::vector<std::vector<int32_t>> hist((size_t)nt);
  parallel_team(nt, [&](int t, int) {
    std::vector<int32_t> &h = hist[(size_t)t];
    h.assign((size_t)nkeys, 0);
    for (int64_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) h[(size_t)key[i]]++;
  });
  for (int64_t k = 0; k < nkeys; ++k) {  // per key: the threads' shares in thread order
    int32_t run = start[(size_t)k];
    for (int t = 0; t < nt; ++t) { const int32_t c = hist[(size_t)t][(size_t)k]; hist[(size_t)t][(size_t)k] = run; run += c; }
    start[(size_t)k + 1] = run;
  }
  parallel_team(nt, [&](int t, int) {
    std::vector<int32_t> &cur = hist[(size_t)t];
    for (int64_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) out[cur[(size_t)key[i]]++] = (int32_t)i;
  });
}

// Barrier of a parallel_team: spins briefly, then yields (the team may share cores with other set-up threads)
struct TeamBarrier {
  const int nt;
  std::atomic<int> count{0};
  std::atomic<int> phase{0};
  explicit TeamBarrier(int n) : nt(n) {}
  void wait() {
    const int ph = phase.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == nt - 1) {
      count.store(0, std::memory_order_relaxed);
      phase.store(ph + 1, std::memory_order_release);
      return;
    }
    for (int spins = 0; phase.load(std::memory_order_acquire) == ph; ++spins) {
      if (spins < 2000) __builtin_ia32_pause();
      else std::this_thread::yield();
    }
  }
};

// ---- level-synchronous breadth-first search on a team of host threads, in the SERIAL queue order -------------------------------
// Continues the search whose queue is o[head, tail) (one complete level) until it dies out and returns the new tail.  The order
// written to o[] is exactly what the serial loop
//     for (; head < tail; ++head) for (k over the neighbours w of o[head]) if (accept(w) && mark[w] != st) { mark[w] = st; o[tail++] = w; }
// produces, for any thread count: while a level is expanded, mark[w] = -(i + 1) is the claim of frontier position i on the
// unvisited w and the smallest position wins -- the vertex the serial search discovers w from -- then every thread emits the
// claims it placed that still hold, in position and neighbour order, and the threads' lists are concatenated in thread order.
// mark[w] == st means visited; stamps are positive, claims negative, any other value counts as unvisited.  levels (optional):
// the queue position where every level starts is appended.  Nothing else may write mark[] of accepted vertices meanwhile.
template <class Accept>
static inline int64_t bfs_parallel(const int64_t *ptr, const int32_t *nbr, int32_t *mark, int32_t st, int32_t *o, int64_t head, int64_t tail,
                                   int nt, Accept accept, std::vector<int64_t> *levels) {
  nt = std::max(1, std::min(nt, 64));
  TeamBarrier bar(nt);
  std::vector<int64_t> cnt((size_t)nt * 8, 0);  // (one cache line per thread)
  int64_t head_out = head, tail_out = tail;
  // Levels of fewer than `small` vertices (the first steps of any search; every level of a long thin graph -- a 1-D column has as
  // many levels as cells) are not worth three barriers: thread 0 runs them alone, as the plain serial loop, while the team waits once.
  constexpr int64_t small = 512;
  int64_t sh_h = head, sh_tl = tail;  // (written by thread 0 in front of a barrier, read by the team behind it)
  parallel_team(nt, [&](int t, int) {
    std::vector<int32_t> buf;
    std::vector<std::pair<int32_t, int32_t>> claimed;  // (claim, vertex) of every claim this thread placed, in serial order
    int64_t h = head, tl = tail;
    while (h < tl) {
      if (tl - h < small) {
        if (t == 0) {
          while (h < tl && tl - h < small) {
            if (levels) levels->push_back(h);
            const int64_t level_end = tl;
            for (; h < level_end; ++h) {
              const int32_t v = o[h];
              for (int64_t k = ptr[v]; k < ptr[v + 1]; ++k) {
                const int32_t w = nbr[k];
                if (accept(w) && mark[w] != st) { mark[w] = st; o[tl++] = w; }
              }
            }
          }
          sh_h = h;
          sh_tl = tl;
        }
        bar.wait();
        h = sh_h;
        tl = sh_tl;
        bar.wait();  // (thread 0 may write sh_* again only after everybody has read them)
        continue;
      }
      if (t == 0 && levels) levels->push_back(h);
      const int64_t F = tl - h, i0 = h + F * t / nt, i1 = h + F * (t + 1) / nt;
      claimed.clear();
      for (int64_t i = i0; i < i1; ++i) {  // claims
        // the queue runs ahead of the random accesses (input numberings may be arbitrary: every access a cache miss)
        if (i + 8 < i1) __builtin_prefetch(&ptr[o[i + 8]]);
        if (i + 4 < i1) __builtin_prefetch(&nbr[ptr[o[i + 4]]]);
        if (i + 2 < i1) { const int32_t v2 = o[i + 2]; for (int64_t k = ptr[v2]; k < ptr[v2 + 1]; ++k) __builtin_prefetch(&mark[nbr[k]]); }
        const int32_t v = o[i], mine = -(int32_t)(i - h) - 1;
        for (int64_t k = ptr[v]; k < ptr[v + 1]; ++k) {
          const int32_t w = nbr[k];
          if (!accept(w)) continue;
          int32_t m = __atomic_load_n(&mark[w], __ATOMIC_RELAXED);
          while (m != st && !(m < 0 && m >= mine)) {
            if (__atomic_compare_exchange_n(&mark[w], &m, mine, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
              claimed.emplace_back(mine, w);
              break;
            }
          }
        }
      }
      bar.wait();
      buf.clear();
      for (const auto &cw : claimed)  // the claims that held: discovered from this position, in position and neighbour order
        if (__atomic_load_n(&mark[cw.second], __ATOMIC_RELAXED) == cw.first) {
          __atomic_store_n(&mark[cw.second], st, __ATOMIC_RELAXED);
          buf.push_back(cw.second);
        }
      cnt[(size_t)t * 8] = (int64_t)buf.size();
      bar.wait();
      int64_t off = 0, total = 0;
      for (int u = 0; u < nt; ++u) { if (u == t) off = total; total += cnt[(size_t)u * 8]; }
      std::copy(buf.begin(), buf.end(), o + tl + off);
      bar.wait();  // (cnt[] is written again only behind the next level's first barrier)
      h = tl;
      tl += total;
    }
    if (t == 0) { head_out = h; tail_out = tl; }
  });
  (void)head_out;
  return tail_out;
}

// Pacing of the health checks inside a host spin on pinned memory: due() is true once every `period_ms` of waiting.  The check
// itself is a hipStreamQuery, which is NOT free for the device: it puts a marker packet behind the last enqueued kernel, and the
// kernel enqueued next waits for it (measured round 4: 6.3 instead of 1.4 us in front of the first kernel of every BiCGStab
// iteration while the host polled the stream every 16 384 spins).
struct SpinPacer {
  uint64_t spins = 0;
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  bool due(int period_ms = 50) {
    if ((++spins & 0xfff) != 0) return false;
    const auto now = std::chrono::steady_clock::now();
    if (now - last < std::chrono::milliseconds(period_ms)) return false;
    last = now;
    return true;
  }
};

// MB of memory this process has touched for the first time so far (minor page faults x page size): printed per set-up phase next
// to the seconds (setup_timing = 1).  On virtualised hosts a first touch costs 1.5-3.5 s per GB on one thread (measured on the
// build box, round 6), which makes fresh allocations the largest single term of the set-up there.
double setup_faulted_mb();

// Fresh memory of the large set-up arrays.  A vector of tens of MB is a new mapping whose pages are faulted in one by one by the
// thread that fills it; on a virtualised host every such fault leaves the guest (build box, round 6: 1.6-6 s per GB on one thread,
// 12 of the 20 s of a 10M-cell set-up).  prefault_pages asks the kernel to map the whole range in ONE call
// (madvise MADV_POPULATE_WRITE, Linux >= 5.14: 0.3-0.4 s per GB on the same box, no worse than first touch on a healthy host);
// where the call is not available nothing happens and the pages are faulted on first touch as before.  Ranges below 4 MB are left
// alone.  (Touching the pages from all host cores instead was measured twice, GPU box round 4 and build box round 5: no gain.)
void prefault_pages(void *p, size_t bytes);
// v.resize(n) / v.assign(n, value) with the new storage mapped up front
template <class T>
static inline void resize_parallel(std::vector<T> &v, size_t n) {
  if (n > v.capacity()) {
    v.reserve(n);
    prefault_pages((void *)(v.data() + v.size()), (n - v.size()) * sizeof(T));
  }
  v.resize(n);
}
template <class T>
static inline void assign_prefaulted(std::vector<T> &v, size_t n, const T &value) {
  if (n > v.capacity()) {
    v.clear();
    v.reserve(n);
    prefault_pages((void *)v.data(), n * sizeof(T));
  }
  v.assign(n, value);
}
template <class T>
static inline void reserve_prefaulted(std::vector<T> &v, size_t n) {
  if (n > v.capacity()) {
    v.reserve(n);
    prefault_pages((void *)(v.data() + v.size()), (n - v.size()) * sizeof(T));
  }
}

// adjacency of the cells from the neighbourship (jh_setup.cpp): bucketed counting sort, rows in ascending face order
void build_adjacency_buckets(int64_t nc, int64_t nf, const int64_t *N, int32_t *Nhost, bool self_loops_ok, bool signed_faces,
                             std::vector<int64_t> &ptr, std::vector<int32_t> &nbr, std::vector<int32_t> &sface);

// ---- graph partitioner (jh_partition.cpp): recursive bisection + Fiduccia-Mattheyses refinement --------------------------------
struct PGraph {
  int64_t n;           // cells 0 .. n-1 take part; neighbour ids >= n are ignored (ghost cells of a rank-local subdomain)
  const int64_t *ptr;  // adjacency, without self loops
  const int32_t *nbr;
  const double *w;     // edge weights or nullptr (all 1)
};
void partition_bisect(const PGraph &G, std::vector<int32_t> &&cells, int64_t nparts, double imbalance, int64_t max_part,
                      std::vector<int32_t> &label, int32_t rim_cell);
void partition_on_bfs_numbering(int64_t nc, const int64_t *ptr, const int32_t *nbr, const double *wk, const int32_t *sface, const double *fw,
                                double fw_scale, double w_floor, int64_t nparts, double imbalance, int64_t max_part, int32_t *label_out,
                                const std::function<void(const char *)> &lap);

// ---- tiling constants for the CSR row-segment kernels ------------------------------------------------
// Value of a vector entry that lives either in an LDS window or in global memory.  Two loads -- an LDS read every lane issues and
// a global load under the exec mask of the lanes outside the window -- because `in ? lds[i] : g[j]` compiles to ONE flat_load
// through a selected generic pointer: flat loads count on vmcnt AND lgkmcnt and return out of order with respect to both, so
// every wait behind them is a wait for everything.
#ifdef __HIPCC__
__device__ __forceinline__ double lds_or_global(const double *lds, unsigned i, bool in, const double *g, size_t j) {
  // explicit address spaces: loads through generic pointers are merged again (SimplifyCFG) before the address spaces are inferred
  double v = *(const __attribute__((address_space(3))) double *)(lds + (in ? i : 0u));
  if (!in) v = *(const __attribute__((address_space(1))) double *)(g + j);
  return v;
}
#endif
constexpr int TILE_THREADS = 256;  // 4 wavefronts
constexpr int TILE_NNZ = 1024;     // block-nnz staged in LDS per workgroup
constexpr int TILE_ROWS = 256;     // rows per tile (one row-reduce thread each; row id fits uint8)
constexpr int NUM_XCD = 8;
constexpr int JH_NSCALARS = 64;   // device scalars of a context
constexpr int S_STATS = 32;       // [32, 44): k_absstats results (4 per variable, N <= 3)

// ---- device buffer helper ----------------------------------------------------------------------------
// Host <-> device copies of caller-owned (pageable) memory.  They go through a page-locked bounce buffer of the library instead of
// handing the caller's pointer to hipMemcpyAsync: for pageable memory the runtime pins the range on the fly and caches the pinning,
// and a heap that shrinks and grows again under it (large numpy / std::vector buffers come from the brk heap once glibc has raised
// its mmap threshold) ends in "Memory access fault by GPU ... on address <host heap page>" -- seen once in ~10 runs of the GPU
// test suite (round 3), always inside an upload of a freshly computed array.  Memory that IS page-locked (jh_host_register,
// hipHostMalloc) is copied directly.  Both calls return when the copy is complete; `s` orders it after earlier work of the stream.
void copy_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t s);
void copy_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t s);

// PLANNING CONTEXTS (jh_context_create_host): a context without a device.  The set-up entry points -- jh_tpfa_create*, the table
// getters, jh_csr_create, jh_ilu0_create and its info getters -- run their host phases (connectivity, device ordering, pattern,
// tiles, ILU(0) symbolic phase and layouts) and skip every allocation, upload and memset; everything that computes on the device
// refuses such a context (require_device).  tl_plan_only is raised by DeviceScope for the duration of one entry point on the
// calling thread: all device work of the set-up is issued from that thread (the worker threads of parallel_ranges only fill host
// arrays).  There is no CPU compute path behind this: it exists so that the set-up tables can be checked and timed on a box
// without a GPU (tests -m "not gpu", tools/setup_probe.py).
extern thread_local bool tl_plan_only;
// Option plan_checksum = 1: every table a real context would upload is folded into a running 64-bit checksum of the planning
// context instead (jh_context_plan_checksum) -- in upload order, lengths included -- so that ALL device tables of a set-up
// (pattern, tiles, jagged layouts, ILU(0) levels / maps / factorisation programs) can be compared between two builds, thread
// counts or machines without a GPU.
extern thread_local uint64_t *tl_plan_crc;
static inline void plan_crc_update(const void *data, size_t bytes) {
  uint64_t h = *tl_plan_crc ^ (0x9e3779b97f4a7c15ull * (uint64_t)(bytes + 1));
  const unsigned char *b = (const unsigned char *)data;
  size_t i = 0;
  for (; i + 8 <= bytes; i += 8) {
    uint64_t w;
    memcpy(&w, b + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  for (; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  *tl_plan_crc = h;
}

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (count && !tl_plan_only) JH_HIP(hipMalloc((void **)&p, count * sizeof(T)));
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    if (count > n) alloc(count);
    if (tl_plan_only) { if (tl_plan_crc) plan_crc_update(h, count * sizeof(T)); return; }
    if (count) jh::copy_h2d(p, h, count * sizeof(T), s);
  }
  void upload(const std::vector<T> &h, hipStream_t s) {
    alloc(h.size());
    if (tl_plan_only) { if (tl_plan_crc) plan_crc_update(h.data(), h.size() * sizeof(T)); return; }
    if (!h.empty()) jh::copy_h2d(p, h.data(), h.size() * sizeof(T), s);
  }
  // zero-fill on the stream (nothing to do for a planning context)
  void zero(hipStream_t s) {
    if (p && n && !tl_plan_only) JH_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

struct Comm;  // RCCL state (jh_comm.cpp)

// ---- per-context options (jh_context_set_option / jh_context_get_option; include/jutul_hip.h lists them) ------------------------
// Every choice a caller can legitimately make about kernels and paths is a named integer option of the CONTEXT, read where the
// object that depends on it is built (layouts: jh_tpfa_create / jh_ilu0_create / first solve) or per call (launch geometry, paths).
// JH_OPTIONS="key=value,key=value" in the environment seeds the options of every new context (one getenv: A/B runs of bench.py).
#define JH_OPTION_LIST(X)                                                                                                      \
  X(consumer_reduce, 1)       /* BiCGStab: second stage of the fused dots inside the consuming kernel (PendSum) */            \
  X(spmv_jagged, 1)           /* Krylov loop multiplies out of the jagged-slice copy when the matrix has one */               \
  X(spmv_col_bits, 0)         /* jagged column ids: 16 / 32, 0 = by size (16 from 3M rows) */                                  \
  X(spmv_waves_per_xcd, 0)    /* persistent wavefronts per XCD of the SpMV kernels, 0 = what is resident at once */            \
  X(spmv_nontemporal, -1)     /* jagged SpMV, 32-bit columns: matrix stream non-temporal; -1 = by working set (read when the layout is built) */ \
  X(spmv_waves, 0)            /* wavefronts per workgroup of the jagged SpMV in the Krylov loop: 4 / 8 / 16, 0 = 16 */           \
  X(spmv_window, 1)           /* CSR tile SpMV: LDS window of x */                                                             \
  X(spmv_pipe, 1)             /* CSR tile SpMV with a fused dot: software-pipelined variant */                                 \
  X(sync_loop, 0)             /* 1: the host waits for iteration k before enqueueing k+1 (no speculative iteration) */         \
  X(halo_overlap, 0)          /* ghost exchange on a second stream behind the interior SpMV tiles */                           \
  X(fused_product, 0)         /* D-ILU storage + product inside the preconditioner apply (set before jh_ilu0_create) */        \
  X(fuse_gather, 1)           /* BiCGStab vector updates inside the gather phase of the ILU(0) apply */                        \
  X(fused_pack, 1)            /* the ILU(0) apply fills the halo send buffer */                                                \
  X(ilu_jagged, 1)            /* chunk-jagged factor layout (set before jh_ilu0_create) */                                     \
  X(ilu_threads, 0)           /* threads per block of the row-major sweeps: 64 / 128 / 256 / 512, 0 = default */               \
  X(ilu_factor_kernel, -1)    /* pivot-only refactorisation: 0 workgroup per block, 1 wavefront per block, -1 = by pattern */  \
  X(ilu_factor_threads, 0)  /* program-driven refactorisation: threads per block (0: 512, rows form 256) */                      \
  X(ilu_factor_wave_per_row, 1)                                                                                                \
  X(ilu_diag_factor, 1)       /* pivot-only kernels for triangle-free block patterns (set before jh_ilu0_create) */            \
  X(ilu_prog, 1)              /* factorisation programs (set before jh_ilu0_create) */                                         \
  X(ilu_lean_upload, 0)       /* jagged layout + direct factor kernels: do not upload the row-major structure arrays (set before jh_ilu0_create; NOT yet run on a GPU) */                                         \
  X(ilu_factor_global, 0)     /* force the per-level refactorisation kernels (set before jh_ilu0_create) */                    \
  X(asm_pipe, 1)              /* persistent pipelined assembly kernel (scalar laws) */                                         \
  X(asm_pipe2_wgs, 0)         /* 2x2-block laws: workgroups per XCD of the pipelined kernel, 0 = tile kernel */                \
  X(block_order, 0)           /* device blocks: 0 recursive bisection, 1 breadth-first "onion" (set before jh_tpfa_create) */  \
  X(block_weights, 1)         /* device blocks: cut weak couplings first when jh_tpfa_create_weighted is given face weights (0: ignore them) */ \
  X(read_sync, 0)             /* device scalars through copy + stream synchronise instead of the pinned record */              \
  X(comm_timeout_ms, 600000)  /* limit of the mailbox / push-halo waits inside kernels, 0 = wait like a collective */          \
  X(upload_bounce, 1)         /* PROCESS-WIDE: caller arrays cross PCIe through the library's page-locked bounce buffer; 0 = straight from the caller's (pageable) memory */ \
  X(setup_heap, 0)            /* PROCESS-WIDE (glibc): large blocks from the heap, freed heap memory kept -> later set-up tables reuse mapped pages; 0 restores and trims */ \
  X(setup_timing, 0)          /* print the set-up phases */                                                                    \
  X(plan_checksum, 0)         /* planning contexts: checksum every table that would be uploaded (jh_context_plan_checksum) */  \
  X(xrank_consumer, -1)       /* several ranks: dots all-reduced inside the consuming kernels + push-halo hand-shake inside the product; -1 = when the host declared exclusive compute units (jh_comm_set_exclusive) */ \
  X(jds_keep, 0)              /* jh_spmv_jagged: do not refresh the jagged copy (timing probe) */
struct Options {
#define JH_OPT_FIELD(name, def) int64_t name = def;
  JH_OPTION_LIST(JH_OPT_FIELD)
#undef JH_OPT_FIELD
  bool set(const char *key, int64_t v);         // false: unknown key
  bool get(const char *key, int64_t *v) const;
  void seed_from_env();                         // JH_OPTIONS
};

// Mailbox of the node-local scalar all-reduce / push halo (jh_comm_ipc_*, jh_halo_ipc_*; protocol in jh_halo.hip)
constexpr int MAIL_R = 16, MAIL_V = 8;
// Slot sets of the scalar all-reduces, used round robin by epoch.  A rank is at most one executed all-reduce ahead of the slowest
// one (it cannot finish epoch e' before everybody has contributed to e', i.e. has finished reading the epoch before it), so two
// live sets exist at any time; the epoch is counted by the HOST per enqueued all-reduce and the launches of a speculative Krylov
// iteration past convergence consume their epochs without executing (on every rank alike: the done flag derives from all-reduced
// bits), which leaves gaps of a few epochs between consecutive executed ones -- eight sets keep any two live epochs apart.
constexpr int MAIL_S = 8;
// Landing buffers / flags of the push halo, round robin by exchange epoch: consecutive EXECUTED exchanges must not share a set (a
// neighbour may still be copying out of the previous one), and the launches of a speculative Krylov iteration skip their epochs.
constexpr int HALO_S = 4;
struct Mailbox {
  double val[MAIL_S][MAIL_R][MAIL_V];
  unsigned long long flag[MAIL_S][MAIL_R];
  // consumer-side all-reduce of one or two sums (xr_* below): per source rank four self-validating 8-byte granules
  // {epoch tag : 32 | half of a double : 32} -- no flag, no second round trip: a reader that sees four matching tags has the data
  unsigned long long gran[MAIL_S][MAIL_R][4];
  unsigned long long hflag[HALO_S][MAIL_R];  // push halo: epoch of the last exchange whose data rank r has delivered here
  unsigned long long abort;             // != 0 once a wait of THIS rank has timed out: its later waits give up at once
};
// What a timed-out wait leaves behind for the host (pinned, host-coherent memory): code 1 = scalar all-reduce, 2 = push halo;
// peer = the rank whose flag never arrived; epoch = the exchange that was waited for.  code is written last.
struct MailErr {
  unsigned long long code, peer, epoch, count;
};
// everything a kernel needs to all-reduce a few scalars through the mailboxes; self == nullptr: no all-reduce requested
struct MailArgs {
  Mailbox *self = nullptr;
  Mailbox *const *peers = nullptr;
  int rank = 0, nranks = 1;
  unsigned long long epoch = 0;          // 1 + the number of all-reduces enqueued before this one (host counter, equal on all ranks)
  unsigned long long timeout_ticks = 0;  // 100 MHz ticks, 0 = wait like a collective
  MailErr *err = nullptr;
};

// Push-halo hand-shake inside the product kernel (consistent! before mul!, ext/JutulPartitionedArraysExt/linalg.jl:37-55, without
// a launch of its own): workgroup 0 of the product raises this rank's flag in the neighbours' mailboxes (the rows were pushed by
// the preceding kernel), waits for theirs, copies the landed values into the ghost rows of the vector with write-through stores
// and publishes `ready`; the other workgroups multiply the interior slices first and look at `ready` (one relaxed poll, one
// agent-scope acquire per workgroup) before their first slice that can touch a ghost column.  self == nullptr: no fold.
struct HaloFold {
  Mailbox *self = nullptr;
  Mailbox *const *peers = nullptr;
  const int32_t *nbr = nullptr;
  int n_nbr = 0, rank = 0;
  unsigned long long epoch = 0;        // push epoch of this exchange (its parity selects the landing buffer)
  const double *landing = nullptr;     // this rank's landing buffer of that parity, receive-list order
  double *xg = nullptr;                // the vector whose ghost rows are filled
  const int32_t *recv_idx = nullptr;   // device row of every received cell
  int n_recv = 0;
  int interior_rows = 0;               // rows [0, interior_rows) have no ghost column
  unsigned long long *ready = nullptr; // device word: epoch of the last exchange whose ghosts are in place
  MailErr *err = nullptr;
  unsigned long long timeout_ticks = 0;
};

}  // namespace jh

// ---- handle structs ------------------------------------------------------------------------------------
constexpr int jh_num_xcd = 8;
struct jh_context_s {
  int device = 0;
  jh::Options opt;
  hipStream_t stream = nullptr;
  int ncu_total = 256, ncu = 256;  // compute units of the device / of the stream (jh_context_set_cu_mask)
  int cu_first = 0;                // first bit of the CU mask (ncu < ncu_total: every stream of the context carries it)
  hipStream_t new_stream() const;  // a non-blocking stream under the context's CU mask (jh_api.cpp)
  int cus_per_xcd() const { return std::max(1, ncu / jh_num_xcd); }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_step[6] = {};  // jh_newton_step: factor / solve / update brackets, read once at the end of the step
  // reduction scratch: partial sums [NSLOT][max_blocks], device scalars, pinned host mirror
  jh::DevBuf<double> partials;
  size_t partial_stride = 0;
  jh::DevBuf<double> scalars;  // JH_NSCALARS doubles: [0, 11) Krylov, [12, 15) convergence, [16, 32) limits / all-reduce staging, [20] S_DONE, [32, 48) absstats
  double *h_scalars = nullptr; // pinned, JH_NSCALARS doubles
  jh::DevBuf<double> stage;    // staging for permuted uploads/downloads
  double *h_pub = nullptr;     // pinned + coherent: 2 records of JH_PUB_LEN doubles the solver loop publishes to (see jh_krylov.hip)
  uint64_t pub_seq = 0;        // sequence number of the last published record
  double *h_rd = nullptr;      // pinned + coherent: JH_NSCALARS doubles read_scalars publishes to, [JH_NSCALARS - 1] = sequence number
  uint64_t rd_seq = 0;
  jh::Comm *comm = nullptr;
  // second stream + events of the overlapped halo exchange (created on first use, jh_comm.cpp)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_halo_ready = nullptr, ev_halo_done = nullptr;
  void ensure_stage(size_t n) {
    if (stage.n < n) stage.alloc(n);
  }
  bool plan_only = false;  // jh_context_create_host: no device behind this context (device == -1, no stream)
  uint64_t plan_crc = 0;   // option plan_checksum: running checksum of everything a real context would have uploaded
};

namespace jh {
// First statement of a set-up entry point that also serves planning contexts: selects the context's device, or raises
// tl_plan_only until the entry point returns.
struct DeviceScope {
  bool prev;
  uint64_t *prev_crc;
  explicit DeviceScope(jh_context c) : prev(tl_plan_only), prev_crc(tl_plan_crc) {
    if (c->plan_only) { tl_plan_only = true; tl_plan_crc = c->opt.plan_checksum ? &c->plan_crc : nullptr; }
    else JH_HIP(hipSetDevice(c->device));
  }
  ~DeviceScope() { tl_plan_only = prev; tl_plan_crc = prev_crc; }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};
// every entry point that computes on the device
static inline void require_device(jh_context c) {
  if (!c) JH_THROW("null context");
  if (c->plan_only) JH_THROW("this context has no device (jh_context_create_host): it only plans the set-up tables");
}
// first statement of every entry point that works on the device
static inline void select_device(jh_context c) {
  require_device(c);
  JH_HIP(hipSetDevice(c->device));
}
static inline void stream_sync(hipStream_t s) {
  if (!tl_plan_only) JH_HIP(hipStreamSynchronize(s));
}
}  // namespace jh

namespace jh {

// Sparsity pattern in DEVICE numbering shared by matrices / vectors / preconditioners.
struct Pattern {
  jh_context ctx = nullptr;
  int64_t n = 0;     // block rows
  int64_t nnzb = 0;  // block non-zeros on the device
  // Multigraph neighbourships (several faces between one cell pair): the reference's pattern holds ONE entry per pair
  // (sparse() merges duplicates, conservation.jl:486-505; nnzb_host counts those), the device keeps one slot per face so that
  // every half-face has its own transmissibility.  Per pair the slot of the highest face id is the primary one; the others
  // ("shadow" slots) map to no host slot and their off-diagonal value is stored as zero -- the reference assigns, not adds,
  // the off-diagonal per half-face (ad.jl:74-76), so the last parallel face wins there while residual and diagonal sum all.
  int64_t nnzb_host = 0;
  std::vector<int32_t> shadow_slots;  // device slots of the shadow entries (usually empty)
  std::vector<char> shadow_flag;      // per device slot, only allocated when shadow_slots is not empty
  DevBuf<int32_t> d_shadow_slots;
  bool is_shadow(int64_t k) const { return !shadow_flag.empty() && shadow_flag[(size_t)k]; }
  int bs = 1;
  std::vector<int32_t> rowptr, col, diag;  // host copies, 0-based, device numbering
  // device order <-> host order (empty => identity)
  std::vector<int32_t> perm;      // perm[i] = host row at device position i
  std::vector<int32_t> iperm;     // iperm[host] = device position
  std::vector<int32_t> nz_hslot;  // device block slot -> host block slot (empty => identity)
  std::vector<int32_t> block_ptr; // contiguous device row blocks (block-Jacobi partition), may be empty
  // rank-local subdomains: blocks [0, interior_blocks) = rows [0, interior_rows) have no ghost neighbour (their SpMV rows /
  // ILU blocks neither feed nor need the halo exchange); then the boundary blocks, then the ghost blocks.  -1: not split.
  int64_t interior_rows = -1;
  int32_t interior_blocks = -1;
  int32_t interior_tiles = -1;    // tiles [0, interior_tiles) cover exactly the interior rows
  // tiles
  std::vector<int32_t> tile_row;
  int32_t ntiles = 0;
  std::vector<int32_t> tile_desc;  // per tile (first row, rows, first entry, entries): one 16-byte load instead of the chain
                                   // tile_row -> rowptr -> entries at the start of every tile
  DevBuf<int32_t> d_tile_desc;
  // device copies
  DevBuf<int32_t> d_rowptr, d_col, d_diag, d_perm, d_nz_hslot, d_tile_row;
  // Jagged-slice form of the pattern for the SpMV of the Krylov loop (jh_sell.hip): rows in slices of 64 (one wavefront), inside
  // a slice sorted by length, entries stored diagonal by diagonal -> every lane streams its row with fully coalesced loads,
  // accumulates left->right in registers (the reference's order, mat.jl:41-61) and needs neither row pointers, LDS nor barriers.
  struct Jagged {
    bool built = false, usable = false;  // usable: bs == 1 and at most JDS_KMAX entries per row
    int32_t nslices = 0, kmax = 0;
    bool nontemporal = true;  // value / column stream with non-temporal loads (working set beyond the Infinity Cache)
    int64_t nent = 0;
    DevBuf<int32_t> d_base;  // [nslices + 1] first entry of every slice
    DevBuf<uint8_t> d_cnt;   // [nslices * 16] byte j: rows of the slice with more than j entries
    DevBuf<uint8_t> d_perm;  // [nslices * 64] lane -> row offset inside the slice
    DevBuf<int32_t> d_col;   // [nent] column ids, jagged order (matrices below 3M rows)
    // 16-bit column codes (larger matrices; exactly one of d_col / d_col16 is filled): code < JDS_FAR: column = window origin of the slice + code; else entry code - JDS_FAR of
    // the slice's list of far columns
    DevBuf<uint16_t> d_col16;  // [nent + 64]
    DevBuf<int32_t> d_win;     // [2 * nslices] window origin (may be negative), first far entry
    DevBuf<int32_t> d_far;     // far columns of all slices (+ padding: lanes past a diagonal's count decode, too)
    DevBuf<uint16_t> d_src;  // [nent] CSR slot each jagged entry is copied from, relative to the first slot of its slice
  } jag;
  void build_tiles();
  void upload();
  void build_jagged();  // lazily, on the first Krylov solve
};

}  // namespace jh

struct jh_tpfa_s {
  jh_context ctx = nullptr;
  int64_t nc = 0, nf = 0, nhf = 0, nnzb = 0;  // nnzb: device slots (== pat->nnzb); the host pattern has pat->nnzb_host
  int N = 1;
  std::vector<int32_t> Nhost;  // 2*nf, 1-based, as given (cells are < 2^31: jh_tpfa_create checks)
  std::shared_ptr<jh::Pattern> pat;
  // per device nnz: signed face id (+(f+1) if row cell == N[1,f], -(f+1) if == N[2,f], 0 on the diagonal)
  std::vector<int32_t> nz_face;
  jh::DevBuf<int32_t> d_nz_face;
  // lazily built reference-exact host tables (1-based)
  bool tables_built = false;
  std::vector<int64_t> face_pos, hf_self, hf_other, hf_face, hf_sign, h_rowptr, h_colidx, pos_acc, pos_flux;
  void build_tables();
  // halo plan (device indices), see jh_comm.cpp
  struct Halo {
    int64_t n_owned = 0;
    std::vector<int32_t> nbr;
    std::vector<int64_t> send_ptr, recv_ptr;     // per neighbour offsets into send/recv index lists
    jh::DevBuf<int32_t> d_send_idx, d_recv_idx;  // device row indices
    jh::DevBuf<double> d_send_buf, d_recv_buf;
    int64_t n_send = 0, n_recv = 0;
    bool active = false;
    std::vector<int32_t> send_idx_host; // host copy of d_send_idx (the ILU(0) apply can fill the send buffer itself)
    uint64_t epoch = 0;                // bumped by every jh_halo_create
    bool direct_recv = false;          // each neighbour's ghosts are consecutive device rows: no unpack
    std::vector<int32_t> recv_row0;    // first device row of every neighbour's ghosts (direct_recv)
    // "push" exchange over peer-mapped memory (jh_halo_ipc_*): senders store their boundary rows straight into the
    // receiver's landing buffer (uncached, HALO_S sets x n_recv x N doubles, receive-list order)
    double *landing = nullptr;                 // this rank's landing buffer
    int64_t landing_stride = 0;                // doubles per set
    std::vector<double *> peer_landing;        // per neighbour: that rank's landing buffer, mapped here
    jh::DevBuf<double *> d_push_dst[jh::HALO_S]; // per send slot and landing set: where the cell's N doubles go
    jh::DevBuf<int32_t> d_nbr;                 // neighbour ranks on the device
    uint64_t push_epoch = 0;
    bool push_attached = false, push_enabled = false;
    jh::DevBuf<unsigned long long> d_ready;    // HaloFold::ready
  } halo;
};

struct jh_vec_s {
  jh_context ctx = nullptr;
  std::shared_ptr<jh::Pattern> pat;  // for the permutation (may be null => identity, plain vector)
  int64_t len = 0;                   // doubles
  int bs = 1;
  jh::DevBuf<double> d;
};

struct jh_csr_s {
  jh_context ctx = nullptr;
  std::shared_ptr<jh::Pattern> pat;
  jh_tpfa disc = nullptr;  // non-owning, may be null
  jh::DevBuf<double> val;  // nnzb * bs * bs, device slot order, each block column-major
  jh::DevBuf<double> jval; // the same values in jagged-slice order: refreshed at the start of every Krylov solve (jh_sell.hip)
  bool jval_fresh = false; // only between sell_refresh() and the end of that solve
};

struct jh_law_s {
  jh_context ctx = nullptr;
  jh_tpfa disc = nullptr;
  int kind = 0, N = 1;
  double par[7] = {1, 1, 0, 0, 1, 1, 0};
  jh::DevBuf<double> X, X0;  // [N, nc] device order
  jh::DevBuf<double> limits; // 5*N update limits for jh_newton_step (jh_law_set_update_limits); empty: none
  jh::DevBuf<double> Tnz;    // per device nnz: T_f off-diagonal, accumulation coefficient on the diagonal
  jh::DevBuf<double> gnz;    // per device nnz: signed gdz (self -> other); empty when no gravity
  bool has_gdz = false;
  int64_t nsrc = 0;
  jh::DevBuf<int32_t> src_cell;
  jh::DevBuf<double> src_val;
  std::vector<int64_t> h_src_cells;  // the list as last handed over (jh_law_set_sources skips an unchanged one)
  std::vector<double> h_src_vals;
  bool src_set = false;
  // JH_LAW_CUSTOM: the assembly kernel compiled at run time from the user's source (jh_custom.cpp)
  hipModule_t custom_module = nullptr;
  hipFunction_t custom_kernel = nullptr;
  jh::DevBuf<double> custom_par;
  ~jh_law_s() {
    if (custom_module) (void)hipModuleUnload(custom_module);
  }
};

namespace jh {

// ---- kernel launch wrappers (jh_kernels.hip) ----------------------------------------------------------------
void k_fill(hipStream_t s, double *x, int64_t n, double v);
void k_copy(hipStream_t s, double *dst, const double *src, int64_t n);
void k_axpby(hipStream_t s, double *y, double a, const double *x, double b, int64_t n);
void k_negate(hipStream_t s, double *dst, const double *src, int64_t n);
void k_permute_in(hipStream_t s, double *dst, const double *src_hostorder, const int32_t *perm, int64_t n, int bs);
void k_permute_out(hipStream_t s, double *dst_hostorder, const double *src, const int32_t *perm, int64_t n, int bs);
void k_component_out(hipStream_t s, double *dst_hostorder, const double *src, const int32_t *perm, int64_t n, int bs, int e);
void k_gather_blocks(hipStream_t s, double *dst, const double *src, const int32_t *slot, int64_t nblk, int bb,
                     bool scatter);
// dot products: result(s) land in ctx->scalars[slot..]; deterministic two-stage reduction
void k_dot(jh_context ctx, const double *a, const double *b, int64_t n, int slot);
void k_dot2(jh_context ctx, const double *a, const double *b, const double *c, const double *d, int64_t n, int slot, const MailArgs *mail = nullptr);
void k_dot2_to(jh_context ctx, const double *a, const double *b, const double *c, const double *d, int64_t n, double *out, const MailArgs *mail = nullptr);
void k_absmax_strided(jh_context ctx, const double *r, int64_t ncell, int bs, int slot);
void k_absstats(jh_context ctx, const double *a, const double *b, int64_t ncell, int bs, int slot);  // 4 scalars per variable from `slot`
double read_scalar(jh_context ctx, int slot);                  // sync + D2H
void read_scalars(jh_context ctx, int slot, int count, double *out);
double read_scalars_begin(jh_context ctx, int slot, int count);  // publishing kernel enqueued; work enqueued next overlaps the host's wait
void read_scalars_end(jh_context ctx, double seq, int slot, int count, double *out);
// Optional dot-product epilogue of the SpMV: mode 1 -> sum(w .* y) ; mode 2 -> sum(y .* w), sum(y .* y); rows >= n_rows
// (ghost rows) do not contribute.  Results land in ctx->scalars[slot (, slot+1)] after an ordered final reduction.
struct SpmvDot {
  int mode = 0;
  const double *w = nullptr;
  int slot = 0;
  int64_t n_rows = 0;
  bool allreduce = false;  // the result is also summed over the ranks (in the reduction kernel itself when the mailboxes are on)
};
constexpr int JH_PUB_LEN = 16;  // doubles per published record: [0..8) scalars, [8] converged flag, [15] sequence number
constexpr int S_DONE = 20;      // ctx->scalars slot: != 0 once the running Krylov solve has converged (speculative launches exit)
// done != nullptr: the launch is skipped on the device when *done != 0 (speculative Krylov iteration past convergence).
// rng: only tiles [t0, t1); the fused-dot partials go behind part_off earlier ones and the second reduction stage runs
// only if `reduce` (last launch of a split product).  Returns the number of partials this launch produced.
struct SpmvRange {
  int t0 = 0, t1 = 0, part_off = 0;
  bool reduce = true;
};
int k_spmv(jh_context ctx, const Pattern &P, const double *val, const double *x, double *y, double alpha, double beta,
           const SpmvDot *dot = nullptr, const double *done = nullptr, const SpmvRange *rng = nullptr);
// jh_sell.hip: jagged-slice SpMV with the same fused-dot contract; A->jval must be fresh (sell_refresh)
constexpr int JDS_KMAX = 8;
constexpr int JDS_FAR = 0xE000;   // first 16-bit column code that is an index into the slice's far list
constexpr int JDS_BACK = 0x7000;  // the slice's column window starts this many rows before its first row
bool sell_refresh(jh_csr A);  // false: the matrix has no jagged form (block size > 1 or long rows) -> CSR tile kernels
bool sell_usable(jh_csr A);   // the same answer without launching the copy (builds the jagged pattern on first use)
int k_spmv_sell(jh_csr A, const double *x, double *y, double alpha, double beta, const SpmvDot *dot, const double *done,
                bool reduce_now = true, int waves = 4, const HaloFold *fold = nullptr);
// jh_comm.cpp: the next push exchange of v as a HaloFold for the kernel that reads v (advances the push epoch); false: the push
// halo is off
bool halo_fold_args(jh_tpfa d, double *v, HaloFold *out);
void spmv_dot_reduce(jh_context ctx, const SpmvDot *dot, int nparts, const double *done);
void ensure_partials(jh_context ctx, size_t min_stride);
void k_final_reduce(jh_context ctx, int nparts, int count, int slot, bool is_max, const double *done = nullptr,
                    const MailArgs *mail = nullptr);
// jh_comm.cpp: fills *out (and advances the epoch) if n scalars can be all-reduced through the mailboxes right now
bool comm_mail_args(jh_context ctx, int n, MailArgs *out);
uint64_t comm_mail_epoch(jh_context ctx);  // scalar all-reduces enqueued so far (0 without a communicator)
void comm_allreduce_dev(jh_context ctx, double *p, int n, int op);
void k_unit_diag(hipStream_t s, const Pattern &P, double *val, double *r, int64_t n_owned);
void k_zero_slots(hipStream_t s, double *val, const int32_t *slots, int64_t n, int bb);
void k_scale_system(hipStream_t s, const Pattern &P, double *val, double *r, int kind, double dt);

// ---- assembly (jh_assembly.hip) -------------------------------------------------------------------------------
void k_gather_face_data(hipStream_t s, double *nzdata, const int32_t *nz_face, const double *face_data, int64_t nnzb,
                        bool signed_data);
void k_set_diag_data(hipStream_t s, double *nzdata, const int32_t *diag, const int32_t *perm, const double *cell_data,
                     int64_t n, bool use_const, double cval);
void k_assemble(jh_law L, double dt, jh_csr A, jh_vec r);
void k_update_primary(jh_law L, const double *dx, double w, const double *limits);

}  // namespace jh

namespace jh {
// Second stage of a fused dot product, run by its CONSUMER: the kernel that needs the scalar next sums the producer's partials
// itself -- every wavefront redundantly, in one fixed order (per-lane left to right, then an xor butterfly), so all of them hold
// the same bits -- instead of a one-workgroup kernel between producer and consumer (4.8 us + a kernel boundary each, four per
// BiCGStab iteration: 16 % of an iteration at 1.25M cells per GPU).  No atomics, no flags, no spinning: the kernel boundary that
// is there anyway orders partials and consumer.  Workgroup 0 of the consumer also stores the sums to sc[out_slot (, +1)] for the
// kernels behind it.  Producers that feed a PendSum run with at most PEND_MAX workgroups (8 loads per lane and sum).
constexpr int PEND_MAX = 512;     // partials a wavefront sums per batch (8 loads per lane and sum, all in flight together)
constexpr int PEND_LIMIT = 2048;  // most partials a consumer sums itself (the CSR tile product leaves up to 256 per XCD): batches in sequence
struct PendSum {
  const double *part = nullptr;  // nullptr: nothing pending, the scalars are in sc[]
  unsigned stride = 0;           // second sum at part + stride
  int nparts = 0, count = 0;     // count 1 or 2
  int out_slot = 0;
  // Several ranks (mail.self != nullptr): the local sums are all-reduced by the consumer as well (xr_push / xr_sum): wavefront 0 of
  // workgroup 0 stores them into every peer's mailbox, every consuming wavefront collects the peers' sums from its own mailbox and
  // adds them in rank order -- the same bits on every wavefront of every rank, still no launch between producer and consumer.
  // Every wavefront of a chip-filling kernel then waits for the PEERS' workgroup 0: only where each rank has compute units of
  // its own (one process per GPU, or CU-masked streams on a shared one), option xrank_consumer.
  MailArgs mail;
};
// BiCGStab vector update fused into the gather phase of the ILU(0) apply (see ilu_apply_chunked_kernel)
struct IluGather {
  int mode = 0;             // 1: s = r - alpha*q ; 2: p = r + beta*(p - omega*q)
  const double *r = nullptr, *q = nullptr;
  double *out = nullptr;    // s (mode 1) or p (mode 2, read-modify-write)
  const double *sc = nullptr;
  int rho_slot = 0, rho_next_slot = 0, cv_slot = 0, ts_slot = 0;
  int n_owned_rows = 0x7fffffff;  // rows >= n_owned_rows are ghosts: their preconditioner input is zero (linalg.jl:78-88)
  const double *done = nullptr;   // launch is a no-op when *done != 0
  // distributed runs: block 0 first publishes the previous iteration's record (publish_record), which needs the
  // all-reduced scalars and therefore cannot be written by the kernel that produced them
  double *pub_rec = nullptr, *sc_rw = nullptr;
  double pub_seq = 0.0, pub_eps = 0.0;
  int pub_pair = 0;
  // mode 1: <c, A y> (-> cv_slot) still lies in the product's partials; mode 2: (rho_next, ||r||^2) (-> rho_next_slot, +1) in the
  // partials of bicg_xr_dots_kernel -- summed by every wavefront of the apply, stored (and, mode 2, published) by workgroup 0
  PendSum pend;
};
}  // namespace jh

#if defined(__HIPCC__)
namespace jh {
// One record per Krylov iteration goes to pinned host memory so that the host can follow the solve without a stream
// synchronisation: [0..8) scalars, [8] converged flag, [15] sequence number (written last, after a system-scope fence).
__device__ __forceinline__ void publish_record(double *sc, int pair_slot, double eps, double *rec, double seq) {
  const double rr = sc[pair_slot + 1];
  const double conv = (sqrt(rr) <= eps) ? 1.0 : 0.0;
  if (conv != 0.0) sc[S_DONE] = 1.0;  // later launches of this solve (one speculative iteration) become no-ops
  for (int i = 0; i < 8; ++i) __hip_atomic_store(rec + i, sc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(rec + 8, conv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(rec + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// PendSum: called by all 64 lanes of a wavefront; every lane returns the same sums (s1 only when count == 2)
__device__ __forceinline__ void pend_sum_wave(const PendSum &P, double &s0, double &s1) {
  const int lane = threadIdx.x & 63;
  double a0 = 0.0, a1 = 0.0;
  for (int base = 0; base < P.nparts; base += PEND_MAX) {  // (one batch unless the producer was the CSR tile product of a large matrix)
    double v0[PEND_MAX / 64], v1[PEND_MAX / 64];
#pragma unroll
    for (int j = 0; j < PEND_MAX / 64; ++j) {  // all loads first
      const int i = base + lane + 64 * j;
      v0[j] = (i < P.nparts) ? P.part[i] : 0.0;
      v1[j] = (P.count == 2 && i < P.nparts) ? P.part[P.stride + i] : 0.0;
    }
    if (base == 0) { a0 = v0[0]; a1 = v1[0]; } else { a0 += v0[0]; a1 += v1[0]; }
#pragma unroll
    for (int j = 1; j < PEND_MAX / 64; ++j) { a0 += v0[j]; a1 += v1[j]; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {  // a + b == b + a: both partners of an exchange end with the same bits
    a0 += __shfl_xor(a0, off, 64);
    a1 += __shfl_xor(a1, off, 64);
  }
  s0 = a0;
  s1 = a1;
}
// All-reduce of p[0..n) (n <= MAIL_V, global or LDS memory, in place) over the ranks of the node.  Called by ALL threads of a
// workgroup of >= 64 threads (threads 0..63 work, everybody joins the barrier); op 0 sum, 1 max (NaN propagating).  Each rank
// stores its contribution into every rank's mailbox (xGMI peer stores, system-scope release on the epoch flag), waits
// until all ranks have written into its own, and sums in rank order -- identical bits everywhere.  MAIL_S slot sets are used
// round robin by epoch (see Mailbox).
// A wait that ran out of time: tell the host who was missing (it turns that into a jh_last_error failure, comm_check_errors)
// and make every later wait of this rank give up at once, so that the kernels already enqueued drain instead of each
// sitting out its own time limit.
__device__ __forceinline__ void mailbox_wait_failed(Mailbox *self, MailErr *err, unsigned long long code, int peer, unsigned long long epoch) {
  if (__hip_atomic_load(&self->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {  // keep the first culprit
    __hip_atomic_store(&err->peer, (unsigned long long)peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&err->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&err->code, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // (plain load + store: a read-modify-write on host-pinned memory needs PCIe atomics; concurrent time-outs may lose a count)
  __hip_atomic_store(&err->count, __hip_atomic_load(&err->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1ull, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&self->abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spin until *flag >= epoch; false when the time limit ran out (or an earlier wait of this rank already had)
__device__ __forceinline__ bool mailbox_wait(const unsigned long long *flag, unsigned long long epoch, unsigned long long timeout_ticks,
                                             Mailbox *self) {
  const unsigned long long t0 = wall_clock64();  // constant 100 MHz counter
  unsigned spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
    if (timeout_ticks) {
      if (wall_clock64() - t0 > timeout_ticks) return false;
      if ((++spins & 63u) == 0 && __hip_atomic_load(&self->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
__device__ __forceinline__ void mailbox_allreduce_body(const MailArgs &A, double *p, int n, int op) {
  const unsigned long long epoch = A.epoch;
  const int lane = threadIdx.x, par = (int)(epoch % (unsigned long long)MAIL_S);
  if (lane < A.nranks) {
    Mailbox *dst = A.peers[lane];
    for (int i = 0; i < n; ++i) __hip_atomic_store(&dst->val[par][A.rank][i], p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->flag[par][A.rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // Inside a solve the limit is generous (seconds: ranks may legitimately enter a reduction far apart); a rank that never
    // arrives is reported through A.err instead of hanging every rank of the node until the job is killed.
    if (!mailbox_wait(&A.self->flag[par][lane], epoch, A.timeout_ticks, A.self)) mailbox_wait_failed(A.self, A.err, 1ull, lane, epoch);
  }
  __syncthreads();
  if (lane < n) {
    double acc = __hip_atomic_load(&A.self->val[par][0][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int r = 1; r < A.nranks; ++r) {
      const double v = __hip_atomic_load(&A.self->val[par][r][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      acc = op == 1 ? ((v > acc || v != v) ? v : acc) : acc + v;
    }
    p[lane] = acc;
  }
  __syncthreads();
}

// ---- push-halo hand-shake inside a kernel (HaloFold) ---------------------------------------------------------------------------
// exchanger: all NT threads of ONE workgroup
template <int NT>
__device__ __forceinline__ void halo_fold_exchange(const HaloFold &H) {
  const int par = (int)(H.epoch % (unsigned long long)HALO_S);
  if ((int)threadIdx.x < H.n_nbr) {
    const int q = H.nbr[threadIdx.x];
    // my rows for q are in place (stored by the previous kernel on this stream): tell q, then wait for q's
    __hip_atomic_store(&H.peers[q]->hflag[par][H.rank], H.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!mailbox_wait(&H.self->hflag[par][q], H.epoch, H.timeout_ticks, H.self)) mailbox_wait_failed(H.self, H.err, 2ull, q, H.epoch);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H.n_recv; i += NT) {
    const double v = __hip_atomic_load(H.landing + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(H.xg + H.recv_idx[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: visible to every XCD
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every writing wavefront drains before the flag
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(H.ready, H.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer: all threads of a workgroup, once, before the first gather that can touch a ghost row
__device__ __forceinline__ void halo_fold_wait(const HaloFold &H) {
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      unsigned spins = 0;
      while (__hip_atomic_load(H.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < H.epoch) {
        // (the exchanger reports a peer that never arrives; its own time limit ends this wait a little later)
        if (H.timeout_ticks && (++spins & 15u) == 0 && wall_clock64() - t0 > H.timeout_ticks + 100000000ull) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 (one wavefront per workgroup)
  }
  __syncthreads();
}

// ---- consumer-side all-reduce of a PendSum over the ranks ------------------------------------------------------------------------
// ext/JutulPartitionedArraysExt/krylov.jl:51-105 sums every dot over the ranks (MPI all-reduce inside PartitionedArrays' dot).
// Granule i of a rank's contribution: (epoch tag << 32) | 32 bits of the sums (i = 0, 1: low / high half of s0; 2, 3: of s1).
struct XrRegs {
  unsigned long long g[4];  // lane r < nranks, r != rank: rank r's granules as last loaded
};
__device__ __forceinline__ void xr_load(const MailArgs &A, XrRegs &R) {
  const int lane = threadIdx.x & 63;
  if (lane < A.nranks && lane != A.rank) {
    const unsigned long long *g = A.self->gran[A.epoch % (unsigned long long)MAIL_S][lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) R.g[i] = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// ONE wavefront of the launch (wavefront 0 of workgroup 0), all 64 lanes: the local sums go to every peer
__device__ __forceinline__ void xr_push(const MailArgs &A, double s0, double s1) {
  const int lane = threadIdx.x & 63;
  if (lane < A.nranks && lane != A.rank) {
    unsigned long long *g = A.peers[lane]->gran[A.epoch % (unsigned long long)MAIL_S][A.rank];
    const unsigned long long tag = (A.epoch & 0xffffffffull) << 32;
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(s0), b1 = (unsigned long long)__double_as_longlong(s1);
    __hip_atomic_store(g + 0, tag | (b0 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(g + 1, tag | (b0 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(g + 2, tag | (b1 & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(g + 3, tag | (b1 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// All 64 lanes of a wavefront, R from an earlier xr_load: waits until every peer's granules carry this epoch's tag, then replaces
// (s0, s1) -- the local sums, the same bits in every lane -- by the sums over the ranks in rank order.  report: this wavefront
// tells the host about a wait that ran out of time (one reporter per launch is enough; everybody gives up).
__device__ __forceinline__ void xr_sum(const MailArgs &A, XrRegs &R, bool report, double &s0, double &s1) {
  const int lane = threadIdx.x & 63;
  const unsigned long long tag = A.epoch & 0xffffffffull;
  double v0 = 0.0, v1 = 0.0;
  if (lane < A.nranks && lane != A.rank) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0, backoff = 1;
    for (;;) {
      if ((R.g[0] >> 32) == tag && (R.g[1] >> 32) == tag && (R.g[2] >> 32) == tag && (R.g[3] >> 32) == tag) break;
      if (A.timeout_ticks) {
        bool giveup = wall_clock64() - t0 > A.timeout_ticks;
        if (!giveup && (++spins & 63u) == 0) giveup = __hip_atomic_load(&A.self->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        if (giveup) {
          if (report) mailbox_wait_failed(A.self, A.err, 1ull, lane, A.epoch);
          break;
        }
      }
      // back off: thousands of wavefronts polling uncached memory take bandwidth from whoever is still working (a rank that waits
      // long is waiting for a straggler anyway) -- 128 cycles first, doubling up to ~4k
      for (unsigned k = 0; k < backoff; ++k) __builtin_amdgcn_s_sleep(2);
      if (backoff < 32u) backoff <<= 1;
      xr_load(A, R);
    }
    v0 = __longlong_as_double((long long)((R.g[0] & 0xffffffffull) | (R.g[1] << 32)));
    v1 = __longlong_as_double((long long)((R.g[2] & 0xffffffffull) | (R.g[3] << 32)));
  }
  double a0 = 0.0, a1 = 0.0;
  for (int r = 0; r < A.nranks; ++r) {  // rank order: identical bits on every rank
    const double t0 = (r == A.rank) ? s0 : __shfl(v0, r, 64);
    const double t1 = (r == A.rank) ? s1 : __shfl(v1, r, 64);
    a0 = r == 0 ? t0 : a0 + t0;
    a1 = r == 0 ? t1 : a1 + t1;
  }
  s0 = a0;
  s1 = a1;
}

// Second stage of the deterministic two-stage reductions: ONE 1024-thread block sums (or maxes) nparts partials of
// `count` slots in a fixed order.  The loads of a thread are independent (8 accumulators), so even ~50k partials cost a few
// microseconds instead of a serial latency chain.  out[k] is written by thread 0.
// (Folding this stage into the producing kernel -- "the last workgroup to arrive reduces" -- was measured and rejected:
// 2048 same-address device-scope arrivals serialise, 10M-cell SpMV+dot 0.223 -> 0.279 ms with write-through partials and
// 0.357 ms with a release fence per workgroup.)
constexpr int FIN_THREADS = 1024;
template <bool MAX>
__device__ __forceinline__ void final_reduce_body(const double *part, size_t stride, int nparts, int count, double *out) {
  __shared__ double sm[FIN_THREADS / 64];
  for (int k = 0; k < count; ++k) {
    const double *p = part + k * stride;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.0;
    int i = threadIdx.x;
    for (; i + 7 * FIN_THREADS < nparts; i += 8 * FIN_THREADS) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[i + j * FIN_THREADS];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = MAX ? ((v[j] > a[j] || v[j] != v[j]) ? v[j] : a[j]) : a[j] + v[j];
    }
    for (int j = 0; i < nparts; i += FIN_THREADS, ++j) {
      const double v = p[i];
      a[j & 7] = MAX ? ((v > a[j & 7] || v != v) ? v : a[j & 7]) : a[j & 7] + v;
    }
    double s = a[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s = MAX ? ((a[j] > s || a[j] != a[j]) ? a[j] : s) : s + a[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double o = __shfl_down(s, off, 64);
      s = MAX ? ((o > s || o != o) ? o : s) : s + o;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = sm[0];
      for (int w = 1; w < FIN_THREADS / 64; ++w) r = MAX ? ((sm[w] > r || sm[w] != sm[w]) ? sm[w] : r) : r + sm[w];
      out[k] = r;
    }
    __syncthreads();
  }
}
}  // namespace jh
#endif

// ---- ILU (jh_ilu.hip) -------------------------------------------------------------------------------------------------
struct jh_ilu_s;
// ---- Krylov (jh_krylov.hip) ---------------------------------------------------------------------------------------------
struct jh_krylov_s;
