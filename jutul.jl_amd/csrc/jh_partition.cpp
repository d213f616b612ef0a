// Graph partitioner: recursive bisection, every cut grown breadth-first from a pseudo-peripheral cell and improved by
// Fiduccia-Mattheyses boundary refinement, independent sub-problems on all host cores.  Two users:
//  * jh_tpfa_create (JH_REORDER_BLOCKS): the device blocks = block-Jacobi ILU(0) partition.  The reference hands this job to
//    Metis (precond/ilu.jl:37-60 -> generate_lookup, partitioning.jl:20-51); compact blocks cut fewer couplings than blocks
//    grown breadth-first along the rim of the already assigned region (13% instead of 18% of the half-faces at 512 cells per
//    block on the tet lattice), which is worth 10% of the BiCGStab iterations;
//  * jh_partition_graph: the coordinate-free stand-in for the reference's MetisPartitioner for the domain decomposition
//    (partitioning.jl:29-51, generate_metis_graph :64-78; optional face weights like its |A|-weighted graph).
// A partition is an INPUT of the hot path: any valid one reproduces the reference's semantics.  Host integer work only.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <numeric>
#include <queue>
#include <thread>

#include "jh_internal.hpp"

namespace jh {
namespace {

struct Job {
  int32_t lab = 0;  // the cells carry this label; the k parts get lab .. lab + k - 1
  int64_t k = 1;
  int64_t off = 0, n = 0;  // the job's cells: cells[off .. off + n) of the shared list (children split their parent's segment)
  int32_t start = -1;      // a cell on the job's rim if one is known (an end of the parent's sweep), else -1
  int par = 1;             // host threads this job may use for its own sweeps (the top of the tree has fewer jobs than threads)
};

struct Bisector {
  const PGraph &G;
  std::vector<int32_t> &label;
  std::vector<int32_t> cells, order;  // the jobs' cell lists / breadth-first orders, one segment per job: no allocation per job
  std::vector<int32_t> mark, locked;  // shared stamp arrays: concurrent jobs own disjoint cells, stamps are globally fresh
  std::vector<int32_t> dkey;          // breadth-first distance from one end, then distance difference between the two ends
  std::atomic<int32_t> stamp{0};
  double imbalance;
  int64_t max_part;
  static constexpr int64_t hint_min = 65536;  // jobs above this size are shared between the threads, see partition_bisect
  static constexpr int64_t par_min = 1 << 17;  // sweeps over at least this many cells run on the job's own team of threads
  Bisector(const PGraph &g, std::vector<int32_t> &lab, std::vector<int32_t> &&c, double imb, int64_t mp)
      : G(g), label(lab), cells(std::move(c)), imbalance(imb), max_part(mp) {
    resize_parallel(order, cells.size());
    resize_parallel(mark, (size_t)g.n);
    resize_parallel(locked, (size_t)g.n);
    resize_parallel(dkey, (size_t)g.n);
  }

  inline double weight(int64_t k) const { return G.w ? G.w[k] : 1.0; }
  // Labels of cells that belong to another job may change under our eyes (that job runs on another thread); they can never equal
  // one of OUR labels (every job owns a private label range), so any value read is as good as any other -- relaxed atomics
  // make that a defined read.
  inline int32_t lab_of(int32_t c) const { return __atomic_load_n(&label[c], __ATOMIC_RELAXED); }
  inline void set_lab(int32_t c, int32_t l) { __atomic_store_n(&label[c], l, __ATOMIC_RELAXED); }

  // breadth-first order of the n cells c[] (label lab) from `start` into o[] (restarts cover disconnected pieces); returns the
  // last cell reached from the first start (a far end of that piece).  levels (optional): the positions in o[] where a
  // breadth-first level begins.
  // dmode 1: dkey[v] = level of v; 2: dkey[v] -= level of v (after a mode-1 sweep from the other end: distance difference).
  // par > 1 (large jobs, dmode 0): the first piece is swept by a team of threads -- same order (bfs_parallel).
  int32_t bfs_order(int32_t lab, const int32_t *c, int64_t n, int32_t start, int32_t *o, std::vector<int64_t> *levels = nullptr,
                    int dmode = 0, int par = 1) {
    const int32_t st = ++stamp;
    if (levels) levels->clear();
    int32_t far = start, s = start;
    int64_t next_restart = 0, tail = 0;
    int32_t level = -1;
    bool first = true;
    while (tail < n) {
      if (s < 0) {
        while (next_restart < n && mark[c[next_restart]] == st) ++next_restart;
        if (next_restart == n) break;
        s = c[next_restart];
      }
      int64_t head = tail, level_end = head;
      mark[s] = st;
      o[tail++] = s;
      if (first && par > 1 && dmode == 0 && n >= par_min) {
        tail = bfs_parallel(G.ptr, G.nbr, mark.data(), st, o, head, tail, par,
                            [&](int32_t w) { return w < G.n && lab_of(w) == lab; }, levels);
        head = tail;
      }
      while (head < tail) {
        if (head == level_end) {
          if (levels) levels->push_back(head);
          level_end = tail;
          ++level;
        }
        const int32_t v = o[head++];
        if (dmode == 1) dkey[v] = level;
        else if (dmode == 2) dkey[v] -= level;
        for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
          const int32_t w = G.nbr[k];
          if (w < G.n && lab_of(w) == lab && mark[w] != st) { mark[w] = st; o[tail++] = w; }
        }
      }
      if (first) { far = o[tail - 1]; first = false; }
      s = -1;
    }
    return far;
  }

  // Fiduccia-Mattheyses passes on the cut between labels a and b (only the job's cells carry them); near_cut: a superset of the
  // cells that have a neighbour on the other side
  void fm_refine(int32_t a, int32_t b, int64_t ncells, const int32_t *near_cut, int64_t n_near, int64_t &na, int64_t &nb,
                 int64_t max_a, int64_t max_b, int64_t min_a, int64_t min_b) {
    auto gain_of = [&](int32_t v) {  // reduction of the cut weight if v changes side
      const int32_t mine = lab_of(v), other = mine == a ? b : a;
      double g = 0.0;
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
        const int32_t o = G.nbr[k];
        if (o >= G.n) continue;
        const int32_t lo = lab_of(o);
        if (lo == other) g += weight(k);
        else if (lo == mine) g -= weight(k);
      }
      return g;
    };
    auto on_cut = [&](int32_t v) {
      const int32_t mine = lab_of(v);
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
        const int32_t o = G.nbr[k];
        if (o >= G.n) continue;
        const int32_t lo = lab_of(o);
        if ((lo == a || lo == b) && lo != mine) return true;
      }
      return false;
    };
    // the cells on the cut, afterwards kept up to date from the moves (a pass touches the cut's surroundings only)
    std::vector<int32_t> cand;
    for (int64_t i = 0; i < n_near; ++i) if (on_cut(near_cut[i])) cand.push_back(near_cut[i]);
    for (int pass = 0; pass < 4; ++pass) {
      const int32_t st = ++stamp;
      std::priority_queue<std::pair<double, int32_t>> pq;
      for (int32_t v : cand) if (on_cut(v)) pq.push({gain_of(v), v});
      std::vector<int32_t> moved;
      double total = 0.0, best = 0.0;
      size_t best_len = 0;
      const size_t max_moves = std::max<size_t>(64, (size_t)ncells / 8);
      while (!pq.empty() && moved.size() < max_moves) {
        auto [g, v] = pq.top();
        pq.pop();
        if (locked[v] == st) continue;
        const double gn = gain_of(v);
        if (gn != g) { pq.push({gn, v}); continue; }  // stale entry: re-queue with the current gain
        const bool from_a = lab_of(v) == a;
        if (from_a ? (nb + 1 > max_b || na - 1 < min_a) : (na + 1 > max_a || nb - 1 < min_b)) continue;  // would break the balance
        set_lab(v, from_a ? b : a);
        if (from_a) { --na; ++nb; } else { ++na; --nb; }
        locked[v] = st;
        moved.push_back(v);
        total += g;
        if (total > best + 1e-12) { best = total; best_len = moved.size(); }
        if (total < best - 50.0 * std::max(1.0, std::fabs(g))) break;  // far below the best prefix: give up this pass
        for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
          const int32_t o = G.nbr[k];
          if (o < G.n && (lab_of(o) == a || lab_of(o) == b) && locked[o] != st) pq.push({gain_of(o), o});
        }
      }
      for (size_t i = moved.size(); i > best_len; --i) {  // roll back to the best prefix
        const int32_t v = moved[i - 1];
        const bool in_a = lab_of(v) == a;
        set_lab(v, in_a ? b : a);
        if (in_a) { --na; ++nb; } else { ++na; --nb; }
      }
      if (best_len == 0) break;
      // next pass: the old candidates plus the surroundings of what moved
      const int32_t st2 = ++stamp;
      for (int32_t v : cand) mark[v] = st2;
      for (size_t i = 0; i < best_len; ++i) {
        const int32_t v = moved[i];
        if (mark[v] != st2) { mark[v] = st2; cand.push_back(v); }
        for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
          const int32_t o = G.nbr[k];
          if (o < G.n && (lab_of(o) == a || lab_of(o) == b) && mark[o] != st2) { mark[o] = st2; cand.push_back(o); }
        }
      }
    }
  }

  // one bisection: job -> (A, B)
  void split(const Job &job, Job &A, Job &B) {
    const int64_t k1 = job.k / 2, k2 = job.k - k1, n = job.n;
    const int32_t la = job.lab, lb = job.lab + (int32_t)k1;  // side B takes the labels lab + k1 ...
    int32_t *c = cells.data() + job.off, *o = order.data() + job.off;
    // side sizes: proportional to the part counts, each side within `imbalance` of its share, no side too small for its parts
    // and -- max_part > 0 -- none so large that a leaf would have to exceed max_part cells
    int64_t min_a = k1, min_b = k2, cap_a = n - k2, cap_b = n - k1;
    if (max_part > 0) {
      cap_a = std::min(cap_a, k1 * max_part);
      cap_b = std::min(cap_b, k2 * max_part);
      min_a = std::max(min_a, n - cap_b);
      min_b = std::max(min_b, n - cap_a);
      if (min_a > cap_a || min_b > cap_b) JH_THROW("partitioner: the part size cap is infeasible (internal error)");
    }
    int64_t target_a = (int64_t)std::llround((double)n * (double)k1 / (double)job.k);
    target_a = std::max(min_a, std::min(cap_a, target_a));
    // grow side A breadth-first from a pseudo-peripheral cell: the far end of a first sweep, or -- large jobs, whose sweeps are
    // the serial part of the run -- an end of the parent's sweep (for all jobs that would cost 4% more cut: the many small ones
    // take the first sweep)
    std::vector<int64_t> levels;
    int32_t far = job.start;
    const int par = n >= par_min ? job.par : 1;
    if (far < 0 || n <= hint_min || lab_of(far) != la) far = bfs_order(la, c, n, c[0], o, nullptr, 0, par);
    int64_t na, nb, lo, hi;
    // Small jobs only: they make the final shapes (the cut area doubles with every level) and cost nothing extra in wall
    // time, the large jobs at the top are the serial part of the run.  Cut on the 2M-cell lattice 13.6 % -> 13.1 %; BiCGStab
    // iterations per step at 2.5M / 5M / 10M / 20M cells: 23.7 / 24.25 / 24.03 / 24.1 without, 22.05 / 25.0 / 22.2 / 24.12 with
    // a limit of 40 000 cells, 24.0 / 24.0 / 23.5 / 23.25 with 300 000, 24.0 / 23.6 / 22.7 / 24.0 for all jobs (the count of a
    // single solve is an integer that moves by one for any change: averages 24.0, 23.3, 23.7, 23.6).
    constexpr int64_t two_ended_max = 40000;
    if (n <= two_ended_max) {
      // Order the cells by (distance from one end) - (distance from the other end) and cut at the share: the cut is the
      // bisector between the two ends -- flat -- instead of a sphere around one of them.
      const int32_t other = bfs_order(la, c, n, far, o, nullptr, 1);
      bfs_order(la, c, n, other, o, nullptr, 2);  // dkey = d(far) - d(other): small near `far`
      int32_t kmin = INT32_MAX, kmax = INT32_MIN;
      for (int64_t i = 0; i < n; ++i) { kmin = std::min(kmin, dkey[o[i]]); kmax = std::max(kmax, dkey[o[i]]); }
      std::vector<int64_t> pos((size_t)(kmax - kmin) + 2, 0);
      for (int64_t i = 0; i < n; ++i) pos[(size_t)(dkey[o[i]] - kmin) + 1]++;
      for (size_t k = 1; k < pos.size(); ++k) pos[k] += pos[k - 1];
      levels.assign(pos.begin(), pos.end() - 1);  // start of every key class in the sorted list
      // stable in reverse sweep order: inside a key class the cells nearest to `far` come first
      for (int64_t i = n - 1; i >= 0; --i) { const int32_t v = o[i]; c[pos[(size_t)(dkey[v] - kmin)]++] = v; }
      std::copy(c, c + n, o);
      // neighbours differ by at most 2 in the key: the cut lies within two key classes of the one holding position target_a
      const int64_t lv = (int64_t)(std::upper_bound(levels.begin(), levels.end(), target_a) - levels.begin()) - 1;
      lo = levels[std::max<int64_t>(0, lv - 2)];
      hi = lv + 3 < (int64_t)levels.size() ? levels[lv + 3] : n;
    } else {
      bfs_order(la, c, n, far, o, &levels, 0, par);
      // neighbours are at most one breadth-first level apart: the cut lies inside the level of position target_a and its two
      // neighbours
      const int64_t lv = (int64_t)(std::upper_bound(levels.begin(), levels.end(), target_a) - levels.begin()) - 1;  // level of o[target_a]
      lo = levels[std::max<int64_t>(0, lv - 1)];
      hi = lv + 2 < (int64_t)levels.size() ? levels[lv + 2] : n;
    }
    parallel_team(par, [&](int t, int nt) {
      for (int64_t i = target_a + (n - target_a) * t / nt, e = target_a + (n - target_a) * (t + 1) / nt; i < e; ++i) set_lab(o[i], lb);
    });
    na = target_a;
    nb = n - target_a;
    const int64_t max_a = std::min(cap_a, target_a + (int64_t)std::floor(imbalance * (double)target_a));
    const int64_t max_b = std::min(cap_b, (n - target_a) + (int64_t)std::floor(imbalance * (double)(n - target_a)));
    fm_refine(la, lb, n, o + lo, hi - lo, na, nb, max_a, max_b, min_a, min_b);
    if (na < k1 || nb < k2) JH_THROW("partitioner lost a part (internal error)");
    // the children split the segment and keep the breadth-first order
    int64_t wa = 0, wb = na;
    if (par > 1) {  // stable two-way split on the job's team: count per range, then fill
      std::vector<int64_t> cnt_a((size_t)par + 1, 0);
      TeamBarrier bar(par);
      parallel_team(par, [&](int t, int nt) {
        const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
        int64_t a = 0;
        for (int64_t i = i0; i < i1; ++i) a += lab_of(o[i]) == la;
        cnt_a[(size_t)t + 1] = a;
        bar.wait();
        int64_t pa = 0;
        for (int u = 0; u < t; ++u) pa += cnt_a[(size_t)u + 1];
        int64_t pb = na + (i0 - pa);
        for (int64_t i = i0; i < i1; ++i) {
          const int32_t v = o[i];
          if (lab_of(v) == la) c[pa++] = v; else c[pb++] = v;
        }
      });
      for (int u = 0; u < par; ++u) wa += cnt_a[(size_t)u + 1];
      wb = n - wa + na;
    } else {
      for (int64_t i = 0; i < n; ++i) {
        const int32_t v = o[i];
        if (lab_of(v) == la) c[wa++] = v; else c[wb++] = v;
      }
    }
    if (wa != na || wb != n) JH_THROW("partitioner: side counts are inconsistent (internal error)");
    A.lab = la; A.k = k1; A.off = job.off; A.n = na;
    B.lab = lb; B.k = k2; B.off = job.off + na; B.n = nb;
    A.par = B.par = std::max(1, job.par / 2);
    A.start = lab_of(o[0]) == la ? o[0] : -1;  // where this sweep began / ended, unless the refinement moved the cell across
    B.start = lab_of(o[n - 1]) == lb ? o[n - 1] : -1;
  }

  // A job of at most hint_min cells is finished on a private copy of its subgraph: local ids 0 .. n-1 in the order of the global
  // ids (every comparison between cells -- the priority queue's tie-break -- comes out the same), the cell list, the neighbour
  // order and the label values as they are, so the parts are those the shared arrays would give; but label / mark / locked / dkey
  // and the adjacency of 64k cells stay in the thread's cache instead of being scattered over the arrays of the whole graph
  // (10M cells, 19 582 parts, one thread of the build box: 14.1-15.2 -> 13.5 s; the parts are bit-identical).
  void finish_locally(const Job &job) {
    const int64_t n = job.n;
    const int32_t *c = cells.data() + job.off;
    std::vector<int32_t> ids(c, c + n);
    std::sort(ids.begin(), ids.end());
    for (int64_t i = 0; i < n; ++i) dkey[ids[i]] = (int32_t)i;  // (dkey of the job's own cells: nobody else reads or writes it)
    std::vector<int64_t> lptr((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t v = ids[i];
      int64_t d = 0;
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) { const int32_t w = G.nbr[k]; d += w < G.n && lab_of(w) == job.lab; }
      lptr[i + 1] = lptr[i] + d;
    }
    std::vector<int32_t> lnbr((size_t)lptr[n]);
    std::vector<double> lw(G.w ? (size_t)lptr[n] : 0);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t v = ids[i];
      int64_t p = lptr[i];
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
        const int32_t w = G.nbr[k];
        if (w < G.n && lab_of(w) == job.lab) { if (G.w) lw[p] = G.w[k]; lnbr[p++] = dkey[w]; }
      }
    }
    std::vector<int32_t> llab((size_t)n, job.lab), lcells((size_t)n);
    for (int64_t i = 0; i < n; ++i) lcells[i] = dkey[c[i]];
    const PGraph L{n, lptr.data(), lnbr.data(), G.w ? lw.data() : nullptr};
    Bisector sub(L, llab, std::move(lcells), imbalance, max_part);
    std::vector<Job> todo;
    Job root = job;
    root.off = 0;
    root.start = -1;  // (jobs of this size look for their far end themselves)
    root.par = 1;
    todo.push_back(root);
    while (!todo.empty()) {
      const Job cur = todo.back();
      todo.pop_back();
      if (cur.k <= 1) continue;
      Job A, B;
      sub.split(cur, A, B);
      todo.push_back(A);
      todo.push_back(B);
    }
    for (int64_t i = 0; i < n; ++i) set_lab(ids[i], llab[i]);
  }
};

}  // namespace

// label[c] in 0 .. nparts-1 for the cells listed (label must be 0 for them on entry and outside 0..nparts-1 elsewhere, e.g. -1);
// neighbours with index >= G.n are ignored.  max_part > 0 caps the size of every part.  rim_cell: a cell known to lie on the rim
// of the graph (e.g. the last one a breadth-first search reached) or -1.
void partition_bisect(const PGraph &G, std::vector<int32_t> &&cells, int64_t nparts, double imbalance, int64_t max_part,
                      std::vector<int32_t> &label, int32_t rim_cell) {
  if (nparts <= 1 || cells.empty()) return;
  const int64_t ncells = (int64_t)cells.size();
  Bisector bis(G, label, std::move(cells), imbalance, max_part);
  std::mutex m;
  std::condition_variable cv;
  std::vector<Job> queue;
  int64_t active = 0;
  std::exception_ptr err;
  {
    Job j;
    j.lab = 0;
    j.k = nparts;
    j.off = 0;
    j.n = ncells;
    j.start = rim_cell;
    queue.push_back(j);
  }
  int nt = (int)setup_cores();  // hardware threads, capped by the cgroup CPU quota
  if (const char *e = getenv("JH_SETUP_THREADS")) nt = atoi(e);
  nt = std::max(1, std::min(nt, 64));
  queue[0].par = nt;  // the root job sweeps on all threads, its children on half of them each, ...
  auto worker = [&] {
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !queue.empty() || active == 0 || err; });
        if (err || queue.empty()) return;  // queue empty here means active == 0: all done
        size_t best = 0;                   // largest job first: the big splits are the serial part
        for (size_t i = 1; i < queue.size(); ++i) if (queue[i].n > queue[best].n) best = i;
        job = queue[best];
        queue.erase(queue.begin() + (ptrdiff_t)best);
        ++active;
      }
      try {
        // small jobs are finished by the thread that holds them (no queue traffic below 64k cells)
        std::vector<Job> local;
        local.push_back(job);
        while (!local.empty()) {
          const Job cur = local.back();
          local.pop_back();
          if (cur.k <= 1) continue;
          if (cur.n <= Bisector::hint_min) { bis.finish_locally(cur); continue; }
          Job A, B;
          bis.split(cur, A, B);
          if (cur.n > Bisector::hint_min && nt > 1) {
            std::lock_guard<std::mutex> lk(m);
            queue.push_back(A);
            queue.push_back(B);
            cv.notify_all();
          } else {
            local.push_back(A);
            local.push_back(B);
          }
        }
      } catch (...) {
        std::lock_guard<std::mutex> lk(m);
        if (!err) err = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> lk(m);
        --active;
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(worker);
  worker();
  for (auto &x : th) x.join();
  if (err) std::rethrow_exception(err);
}


// Partition of the vertices 0 .. nc-1 of the graph (ptr, nbr; neighbours >= nc are ignored) into nparts, label_out[v] in 0 .. nparts-1.
// Weight of adjacency entry k: wk[k], or |fw[|sface[k]| - 1]| x fw_scale (a weight per face, sface = signed 1-based face id of the
// entry; zero and non-finite weights count as w_floor), or 1.
// lap(name): called after each phase (set-up timing).
void partition_on_bfs_numbering(int64_t nc, const int64_t *ptr, const int32_t *nbr, const double *wk, const int32_t *sface, const double *fw,
                                double fw_scale, double w_floor, int64_t nparts, double imbalance, int64_t max_part, int32_t *label_out,
                                const std::function<void(const char *)> &lap) {
  const bool weighted = wk || (fw && sface);
  // The input numbering may be arbitrary (the bench grid's is scrambled): every sweep of the partitioner would miss the
  // cache on every cell.  One breadth-first renumbering first (neighbours end up close in memory), the bisections run on the
  // renumbered graph of the owned cells, the labels are mapped back.
  std::vector<int32_t> ord, newid, seen;
  resize_parallel(ord, (size_t)nc);
  resize_parallel(newid, (size_t)nc);
  resize_parallel(seen, (size_t)nc);  // 0: not reached yet, 1: reached (negative while a level is expanded: bfs_parallel)
  int nt_bfs = (int)setup_cores();
  if (const char *e = getenv("JH_SETUP_THREADS")) nt_bfs = atoi(e);
  if (nc < (1 << 17)) nt_bfs = 1;
  int64_t first_piece = 0;  // cells of the first connected piece: its last cell is a far end of the graph
  int64_t tail = 0;
  for (int64_t s0 = 0; s0 < nc && tail < nc; ++s0) {
    if (seen[s0] != 0) continue;
    seen[s0] = 1;
    ord[tail] = (int32_t)s0;
    // the serial queue order on all host cores (pieces past the first one are usually small: bfs_parallel costs them a team
    // start, so they take the plain loop)
    if (nt_bfs > 1 && tail == 0) {
      tail = bfs_parallel(ptr, nbr, seen.data(), 1, ord.data(), tail, tail + 1, nt_bfs,
                          [&](int32_t o) { return o < nc; }, nullptr);
    } else {
      int64_t h = tail++;
      for (; h < tail; ++h) {
        if (h + 8 < tail) __builtin_prefetch(&ptr[ord[h + 8]]);  // the queue runs ahead of the random accesses
        if (h + 4 < tail) __builtin_prefetch(&nbr[ptr[ord[h + 4]]]);
        const int32_t c = ord[h];
        for (int64_t k = ptr[c]; k < ptr[c + 1]; ++k) {
          const int32_t o = nbr[k];
          if (o < nc && seen[o] == 0) { seen[o] = 1; ord[tail++] = o; }
        }
      }
    }
    if (s0 == 0) first_piece = tail;
  }
  parallel_ranges(nc, 1 << 18, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) newid[ord[i]] = (int32_t)i; });
  { std::vector<int32_t>().swap(seen); }
  lap("  blocks: renumber");
  std::vector<int64_t> ptr2;
  resize_parallel(ptr2, (size_t)nc + 1);
  parallel_ranges(nc, 65536, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) {
      // (ord[] is an arbitrary place of the input arrays: the loop prefetches ahead of itself, here and in the fill below)
      if (i + 16 < e) __builtin_prefetch(&ptr[ord[i + 16]]);
      if (i + 8 < e) __builtin_prefetch(&nbr[ptr[ord[i + 8]]]);
      const int32_t c = ord[i];
      int64_t deg = 0;
      for (int64_t k = ptr[c]; k < ptr[c + 1]; ++k) deg += nbr[k] < nc;
      ptr2[i + 1] = deg;
    }
  });
  for (int64_t i = 0; i < nc; ++i) ptr2[i + 1] += ptr2[i];
  std::vector<int32_t> nbr2;
  resize_parallel(nbr2, (size_t)ptr2[nc]);
  std::vector<double> w2;
  if (weighted) resize_parallel(w2, (size_t)ptr2[nc]);
  parallel_ranges(nc, 65536, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) {
      if (i + 16 < e) __builtin_prefetch(&ptr[ord[i + 16]]);
      if (i + 12 < e) { const int64_t k12 = ptr[ord[i + 12]]; __builtin_prefetch(&nbr[k12]); if (fw && sface && !wk) __builtin_prefetch(&sface[k12]); }
      if (i + 6 < e) {
        const int32_t c6 = ord[i + 6];
        for (int64_t k = ptr[c6]; k < ptr[c6 + 1]; ++k) {
          if (nbr[k] < nc) __builtin_prefetch(&newid[nbr[k]]);
          if (fw && sface && !wk) __builtin_prefetch(&fw[std::abs(sface[k]) - 1]);
        }
      }
      const int32_t c = ord[i];
      int64_t w = ptr2[i];
      for (int64_t k = ptr[c]; k < ptr[c + 1]; ++k)
        if (nbr[k] < nc) {
          if (weighted) {
            if (wk) {
              w2[w] = wk[k];
            } else {
              const double v = std::fabs(fw[std::abs(sface[k]) - 1]);
              // a sealed (zero) or non-finite coupling still costs something to cut: w_floor = 1e-3 of the mean, so that the
              // refinement does not see free moves across it (the reference's integer weights have a floor of one unit = a tenth of the
              // mean, generate_metis_graph, partitioning.jl:64-78; positive finite weights are passed on as they are)
              const double wv = (std::isfinite(v) ? v : 0.0) * fw_scale;
              w2[w] = wv > 0.0 ? wv : w_floor;
            }
          }
          nbr2[w++] = newid[nbr[k]];
        }
    }
  });
  std::vector<int32_t> lab2, cells;
  resize_parallel(lab2, (size_t)nc);
  resize_parallel(cells, (size_t)nc);
  parallel_ranges(nc, 1 << 18, [&](int64_t b, int64_t e) { std::iota(cells.begin() + b, cells.begin() + e, (int32_t)b); });
  PGraph G{nc, ptr2.data(), nbr2.data(), weighted ? w2.data() : nullptr};
  lap("  blocks: graph");
  partition_bisect(G, std::move(cells), nparts, imbalance, max_part, lab2, (int32_t)(first_piece - 1));
  lap("  blocks: bisection");
  parallel_ranges(nc, 1 << 18, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) label_out[ord[i]] = lab2[i]; });
}

}  // namespace jh

using namespace jh;

// partition(N, nparts[, face weights]) in the role of the reference's MetisPartitioner (partitioning.jl:29-51): out[c] in
// 1..nparts, every part non-empty (nparts <= nc), sizes within `imbalance` (e.g. 0.03) per bisection.
extern "C" int32_t jh_partition_graph(int64_t nc, int64_t nf, const int64_t *N, const double *face_weights, int64_t nparts,
                                      double imbalance, int64_t *out) {
  return guard([&] {
    if (nc < 1 || nf < 0 || !out || (nf > 0 && !N)) JH_THROW("bad arguments");
    if (nparts < 1 || nparts > nc) JH_THROW("nparts must be in 1..nc");
    if (nc > 2000000000LL || nf > (int64_t)INT32_MAX - 1) JH_THROW("graph too large (cells and faces are 32-bit here)");
    if (!(imbalance >= 0.0)) imbalance = 0.03;
    // adjacency (bucketed counting sort, rows in ascending face order; faces connecting a cell to itself are skipped)
    std::vector<int64_t> ptr;
    std::vector<int32_t> nbr, face;
    build_adjacency_buckets(nc, nf, N, nullptr, true, false, ptr, nbr, face);
    // the bisections run on a breadth-first renumbering of the cells (an arbitrary input numbering would miss the
    // cache on every access of every sweep; build box, 3M scrambled cells, weighted: 8 parts 4.1 -> 2.4 s, 64 parts 4.7 -> 2.4 s, the cut
    // weight unchanged to four digits)
    std::vector<int32_t> label;
    resize_parallel(label, (size_t)nc);
    double w_floor = 0.0;  // 1e-3 of the mean |weight|
    if (face_weights && nf > 0) {
      double sum = 0.0;
      for (int64_t f = 0; f < nf; ++f) { const double v = std::fabs(face_weights[f]); sum += std::isfinite(v) ? v : 0.0; }
      w_floor = 1e-3 * sum / (double)nf;
    }
    partition_on_bfs_numbering(nc, ptr.data(), nbr.data(), nullptr, face_weights ? face.data() : nullptr, face_weights, 1.0, w_floor, nparts, imbalance, 0,
                               label.data(), [](const char *) {});
    for (int64_t c = 0; c < nc; ++c) out[c] = (int64_t)label[c] + 1;
  });
}
