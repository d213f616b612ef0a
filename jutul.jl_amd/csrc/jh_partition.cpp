// Graph partitioner for the domain decomposition (a-16): the reference hands the cell graph to Metis (k-way on the
// |A|-weighted graph, partitioning.jl:29-51,64-78).  Metis is third-party and absent; a partition vector is an INPUT of the
// path, so any valid partition reproduces the reference's semantics -- this is the in-house stand-in for grids without
// usable coordinates (dd.partition_rcb needs centroids): recursive bisection, each cut grown breadth-first from a
// pseudo-peripheral cell and improved by Fiduccia-Mattheyses boundary refinement on the (optionally face-weighted) graph.
// Host integer work only.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <queue>

#include "jh_internal.hpp"

namespace jh {
namespace {

struct Graph {
  int64_t n = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> nbr;
  std::vector<double> w;  // edge weights (1 when none were given)
};

Graph build_graph(int64_t nc, int64_t nf, const int64_t *N, const double *fw) {
  Graph G;
  G.n = nc;
  G.ptr.assign(nc + 1, 0);
  for (int64_t f = 0; f < nf; ++f) {
    const int64_t l = N[2 * f], r = N[2 * f + 1];
    if (l < 1 || l > nc || r < 1 || r > nc) JH_THROW("neighborship entry out of range (utils.jl:822: max(N) <= nc)");
    if (l == r) continue;
    G.ptr[l]++;
    G.ptr[r]++;
  }
  for (int64_t c = 0; c < nc; ++c) G.ptr[c + 1] += G.ptr[c];
  G.nbr.resize(G.ptr[nc]);
  G.w.resize(G.ptr[nc]);
  std::vector<int64_t> cur(G.ptr.begin(), G.ptr.end() - 1);
  for (int64_t f = 0; f < nf; ++f) {
    const int64_t l = N[2 * f] - 1, r = N[2 * f + 1] - 1;
    if (l == r) continue;
    const double wf = fw ? std::fabs(fw[f]) : 1.0;
    G.nbr[cur[l]] = (int32_t)r; G.w[cur[l]++] = wf;
    G.nbr[cur[r]] = (int32_t)l; G.w[cur[r]++] = wf;
  }
  return G;
}

// breadth-first order of the cells with label `lab` starting at `start` (restarts cover disconnected pieces); returns the last
// cell reached from the first start (a far end of the piece)
int32_t bfs_order(const Graph &G, const std::vector<int32_t> &label, int32_t lab, const std::vector<int32_t> &cells, int32_t start,
                  std::vector<int32_t> &order, std::vector<int32_t> &mark, int32_t stamp) {
  order.clear();
  order.reserve(cells.size());
  int32_t far = start;
  size_t next_restart = 0;
  int32_t s = start;
  bool first = true;
  while (order.size() < cells.size()) {
    if (s < 0) {
      while (next_restart < cells.size() && mark[cells[next_restart]] == stamp) ++next_restart;
      if (next_restart == cells.size()) break;
      s = cells[next_restart];
    }
    size_t head = order.size();
    mark[s] = stamp;
    order.push_back(s);
    while (head < order.size()) {
      const int32_t v = order[head++];
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
        const int32_t o = G.nbr[k];
        if (label[o] == lab && mark[o] != stamp) { mark[o] = stamp; order.push_back(o); }
      }
    }
    if (first) { far = order.back(); first = false; }
    s = -1;
  }
  return far;
}

// Fiduccia-Mattheyses passes on the cut between labels a and b (only cells of these two labels move).  max_a / max_b: the
// most cells either side may hold.
void fm_refine(const Graph &G, std::vector<int32_t> &label, int32_t a, int32_t b, const std::vector<int32_t> &cells, int64_t &na,
               int64_t &nb, int64_t max_a, int64_t max_b, int64_t min_a, int64_t min_b, std::vector<int32_t> &locked, int32_t &stamp) {
  auto gain_of = [&](int32_t v) {  // reduction of the cut weight if v changes side
    const int32_t mine = label[v], other = mine == a ? b : a;
    double g = 0.0;
    for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
      const int32_t lo = label[G.nbr[k]];
      if (lo == other) g += G.w[k];
      else if (lo == mine) g -= G.w[k];
    }
    return g;
  };
  for (int pass = 0; pass < 6; ++pass) {
    ++stamp;
    std::priority_queue<std::pair<double, int32_t>> pq;
    for (int32_t v : cells) {
      bool boundary = false;
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1] && !boundary; ++k) {
        const int32_t lo = label[G.nbr[k]];
        boundary = (lo == a || lo == b) && lo != label[v];
      }
      if (boundary) pq.push({gain_of(v), v});
    }
    std::vector<int32_t> moved;
    double total = 0.0, best = 0.0;
    size_t best_len = 0;
    const size_t max_moves = std::max<size_t>(64, cells.size() / 8);
    while (!pq.empty() && moved.size() < max_moves) {
      auto [g, v] = pq.top();
      pq.pop();
      if (locked[v] == stamp) continue;
      const double gn = gain_of(v);
      if (gn != g) { pq.push({gn, v}); continue; }  // stale entry: re-queue with the current gain
      const bool from_a = label[v] == a;
      if (from_a ? (nb + 1 > max_b || na - 1 < min_a) : (na + 1 > max_a || nb - 1 < min_b)) continue;  // would break the balance
      label[v] = from_a ? b : a;
      if (from_a) { --na; ++nb; } else { ++na; --nb; }
      locked[v] = stamp;
      moved.push_back(v);
      total += g;
      if (total > best + 1e-12) { best = total; best_len = moved.size(); }
      if (total < best - 50.0 * std::max(1.0, std::fabs(g))) break;  // far below the best prefix: give up this pass
      for (int64_t k = G.ptr[v]; k < G.ptr[v + 1]; ++k) {
        const int32_t o = G.nbr[k];
        if ((label[o] == a || label[o] == b) && locked[o] != stamp) pq.push({gain_of(o), o});
      }
    }
    for (size_t i = moved.size(); i > best_len; --i) {  // roll back to the best prefix
      const int32_t v = moved[i - 1];
      const bool in_a = label[v] == a;
      label[v] = in_a ? b : a;
      if (in_a) { --na; ++nb; } else { ++na; --nb; }
    }
    if (best_len == 0) break;
  }
}

}  // namespace
}  // namespace jh

using namespace jh;

// partition(N, nparts[, face weights]) in the role of the reference's MetisPartitioner (partitioning.jl:29-51): out[c] in
// 1..nparts, every part non-empty (nparts <= nc), sizes within `imbalance` (e.g. 0.03) of nc/nparts.
extern "C" int32_t jh_partition_graph(int64_t nc, int64_t nf, const int64_t *N, const double *face_weights, int64_t nparts,
                                      double imbalance, int64_t *out) {
  return guard([&] {
    if (nc < 1 || nf < 0 || !out || (nf > 0 && !N)) JH_THROW("bad arguments");
    if (nparts < 1 || nparts > nc) JH_THROW("nparts must be in 1..nc");
    if (nc > 2000000000LL) JH_THROW("graph too large");
    if (!(imbalance >= 0.0)) imbalance = 0.03;
    Graph G = build_graph(nc, nf, N, face_weights);
    std::vector<int32_t> label(nc, 0), mark(nc, 0), locked(nc, 0), order;
    int32_t stamp = 0;
    struct Job { int32_t lab; int64_t k; std::vector<int32_t> cells; };  // split the cells of label `lab` into k parts lab..lab+k-1
    std::vector<Job> stack;
    {
      Job j; j.lab = 0; j.k = nparts; j.cells.resize(nc);
      std::iota(j.cells.begin(), j.cells.end(), 0);
      stack.push_back(std::move(j));
    }
    while (!stack.empty()) {
      Job job = std::move(stack.back());
      stack.pop_back();
      if (job.k == 1) continue;
      const int64_t k1 = job.k / 2, k2 = job.k - k1, n = (int64_t)job.cells.size();
      const int32_t la = job.lab, lb = job.lab + (int32_t)k1;  // side B takes the labels lab + k1 ...
      int64_t target_a = (int64_t)std::llround((double)n * (double)k1 / (double)job.k);
      target_a = std::max<int64_t>(k1, std::min<int64_t>(n - k2, target_a));
      // grow side A breadth-first from a pseudo-peripheral cell (two sweeps find a far end of the piece)
      ++stamp;
      int32_t far = bfs_order(G, label, la, job.cells, job.cells[0], order, mark, stamp);
      ++stamp;
      far = bfs_order(G, label, la, job.cells, far, order, mark, stamp);
      ++stamp;
      bfs_order(G, label, la, job.cells, far, order, mark, stamp);
      for (int64_t i = target_a; i < n; ++i) label[order[i]] = lb;
      int64_t na = target_a, nb = n - target_a;
      const int64_t slack_a = (int64_t)std::floor(imbalance * (double)n * (double)k1 / (double)job.k);
      const int64_t slack_b = (int64_t)std::floor(imbalance * (double)n * (double)k2 / (double)job.k);
      fm_refine(G, label, la, lb, job.cells, na, nb, target_a + slack_a, (n - target_a) + slack_b, k1, k2, locked, stamp);
      // both sides must keep enough cells for their parts
      Job A, B;
      A.lab = la; A.k = k1; B.lab = lb; B.k = k2;
      for (int32_t v : job.cells) (label[v] == la ? A.cells : B.cells).push_back(v);
      if ((int64_t)A.cells.size() < k1 || (int64_t)B.cells.size() < k2) JH_THROW("partitioner lost a part (internal error)");
      stack.push_back(std::move(A));
      stack.push_back(std::move(B));
    }
    for (int64_t c = 0; c < nc; ++c) out[c] = (int64_t)label[c] + 1;
  });
}
