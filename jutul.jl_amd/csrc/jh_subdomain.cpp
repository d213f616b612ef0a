// Rank-local subdomains and coordinate partitions on the host, on all cores (no GPU work):
//  * jh_subdomain_*: what PArraySimulator builds per rank from a partition vector (ext/JutulPartitionedArraysExt/interface.jl:38-63
//    with submap_cells(buffer = 0), dd/subdomains.jl:77-182): local cells = [owned (ascending global id) ..., ghosts ...], the
//    faces with both cells local, the local neighbourship, and the halo plan (who sends which owned cells / receives which
//    ghosts).  ghost_order 0: ghosts by ascending global id (the reference's order); 1: by (owning rank, global id) -- every
//    neighbour's ghosts are then consecutive local cells and the device library receives straight into the vectors.
//  * jh_partition_rcb: recursive coordinate bisection, the build-side stand-in for the reference's MetisPartitioner when
//    centroids are at hand (partitioning.jl:29-51; Metis is third-party).  A partition vector is an INPUT of the path.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <numeric>

#include "jh_internal.hpp"

struct jh_subdomain_s {
  int64_t n_owned = 0, n_local = 0;
  std::vector<int64_t> cells, faces, N_local;  // 1-based global ids / local pairs [l0, r0, l1, r1, ...]
  std::vector<int32_t> neighbors;              // 0-based ranks, ascending
  std::vector<int64_t> send_ptr, send_idx, recv_ptr, recv_idx;  // 1-based local cells per neighbour
};

using namespace jh;

namespace {

// out = the indices i in [0, n) with pred(i), ascending -- count, prefix, fill on all cores
template <class Pred>
void select_ascending(int64_t n, Pred pred, std::vector<int64_t> &out) {
  const int64_t chunk = 1 << 16, nchunks = (n + chunk - 1) / chunk;
  std::vector<int64_t> cnt((size_t)nchunks + 1, 0);
  parallel_ranges(nchunks, 4, [&](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) {
      int64_t k = 0;
      for (int64_t i = c * chunk, e = std::min(n, (c + 1) * chunk); i < e; ++i) k += pred(i) ? 1 : 0;
      cnt[c + 1] = k;
    }
  });
  for (int64_t c = 0; c < nchunks; ++c) cnt[c + 1] += cnt[c];
  resize_parallel(out, (size_t)cnt[nchunks]);
  parallel_ranges(nchunks, 4, [&](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) {
      int64_t w = cnt[c];
      for (int64_t i = c * chunk, e = std::min(n, (c + 1) * chunk); i < e; ++i)
        if (pred(i)) out[w++] = i;
    }
  });
}

}  // namespace

extern "C" int32_t jh_subdomain_create(int64_t nc, int64_t nf, const int64_t *N, const int64_t *partition, int64_t rank,
                                       int32_t ghost_order, jh_subdomain *out) {
  return guard([&] {
    if (nc < 1 || nf < 0 || !partition || !out || (nf > 0 && !N)) JH_THROW("bad arguments");
    if (ghost_order != 0 && ghost_order != 1) JH_THROW("ghost_order must be 0 (global id) or 1 (owner, global id)");
    if (rank < 1) JH_THROW("rank is 1-based");
    auto S = std::make_unique<jh_subdomain_s>();
    // 0: foreign, 1: owned, 2: ghost (a foreign cell that shares a face with an owned one)
    std::vector<unsigned char> kind;
    resize_parallel(kind, (size_t)nc);
        parallel_ranges(nc, 1 << 18, [&](int64_t b, int64_t e) {
      for (int64_t c = b; c < e; ++c) {
        if (partition[c] < 1) JH_THROW("partition ids must be >= 1");
        kind[c] = partition[c] == rank ? 1 : 0;
      }
    });
    unsigned char *kd = kind.data();
    parallel_ranges(nf, 1 << 18, [&](int64_t b, int64_t e) {
      for (int64_t f = b; f < e; ++f) {
        const int64_t l = N[2 * f] - 1, r = N[2 * f + 1] - 1;
        if (l < 0 || l >= nc || r < 0 || r >= nc) JH_THROW("neighborship entry out of range (utils.jl:822: max(N) <= nc)");
        const bool ml = partition[l] == rank, mr = partition[r] == rank;
        if (ml && !mr) __atomic_store_n(&kd[r], (unsigned char)2, __ATOMIC_RELAXED);  // (several faces may mark the same cell)
        if (mr && !ml) __atomic_store_n(&kd[l], (unsigned char)2, __ATOMIC_RELAXED);
      }
    });
    std::vector<int64_t> owned, ghosts;
    select_ascending(nc, [&](int64_t c) { return kd[c] == 1; }, owned);
    if (owned.empty()) {  // an empty part is legal, a rank beyond the partition's parts is a caller error (0- vs 1-based ranks)
      int64_t pmax = 0;
      for (int64_t c = 0; c < nc; ++c) pmax = std::max(pmax, partition[c]);
      if (rank > pmax) JH_THROW("jh_subdomain_create: rank " + std::to_string(rank) + " but the partition vector has parts 1.." +
                                std::to_string(pmax) + " (rank is 1-based like the partition ids)");
    }
    select_ascending(nc, [&](int64_t c) { return kd[c] == 2; }, ghosts);
    if (ghost_order == 1)
      std::stable_sort(ghosts.begin(), ghosts.end(), [&](int64_t a, int64_t b) { return partition[a] < partition[b]; });
    const int64_t no = (int64_t)owned.size(), ng = (int64_t)ghosts.size();
    S->n_owned = no;
    S->n_local = no + ng;
    if (S->n_local > 2000000000LL) JH_THROW("subdomain too large");
    std::vector<int32_t> g2l;  // global cell -> local cell (0-based), -1: not local
    resize_parallel(g2l, (size_t)nc);
    parallel_ranges(nc, 1 << 18, [&](int64_t b, int64_t e) { std::fill(g2l.begin() + b, g2l.begin() + e, -1); });
    resize_parallel(S->cells, (size_t)(no + ng));
    parallel_ranges(no, 1 << 16, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) { g2l[owned[i]] = (int32_t)i; S->cells[i] = owned[i] + 1; }
    });
    for (int64_t i = 0; i < ng; ++i) { g2l[ghosts[i]] = (int32_t)(no + i); S->cells[no + i] = ghosts[i] + 1; }
    // faces with both cells local (ghost-ghost faces included, as in the reference's submap), local neighbourship
    select_ascending(nf, [&](int64_t f) { return g2l[N[2 * f] - 1] >= 0 && g2l[N[2 * f + 1] - 1] >= 0; }, S->faces);
    const int64_t nfl = (int64_t)S->faces.size();
    resize_parallel(S->N_local, (size_t)(2 * nfl));
    parallel_ranges(nfl, 1 << 16, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) {
        const int64_t f = S->faces[i];
        S->N_local[2 * i] = g2l[N[2 * f] - 1] + 1;
        S->N_local[2 * i + 1] = g2l[N[2 * f + 1] - 1] + 1;
        S->faces[i] = f + 1;
      }
    });
    // halo plan.  Receive: per owning rank the local ids of its ghosts, in ghost order.  Send: per neighbour the owned cells
    // that are ghosts over there = owned endpoints of the faces crossing to it, ascending global id, each once.
    std::vector<std::pair<int64_t, int64_t>> rg((size_t)ng);  // (owner, local id)
    for (int64_t i = 0; i < ng; ++i) rg[i] = {partition[ghosts[i]], no + i + 1};
    std::stable_sort(rg.begin(), rg.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<std::pair<int64_t, int64_t>> sc;  // (neighbour, owned global cell)
    {
      std::mutex m;
      parallel_ranges(nf, 1 << 18, [&](int64_t b, int64_t e) {
        std::vector<std::pair<int64_t, int64_t>> loc;
        for (int64_t f = b; f < e; ++f) {
          const int64_t l = N[2 * f] - 1, r = N[2 * f + 1] - 1;
          const bool ml = partition[l] == rank, mr = partition[r] == rank;
          if (ml && !mr) loc.push_back({partition[r], l});
          if (mr && !ml) loc.push_back({partition[l], r});
        }
        std::lock_guard<std::mutex> lk(m);
        sc.insert(sc.end(), loc.begin(), loc.end());
      });
    }
    std::sort(sc.begin(), sc.end());
    sc.erase(std::unique(sc.begin(), sc.end()), sc.end());
    S->recv_ptr.push_back(0);
    S->send_ptr.push_back(0);
    size_t si = 0;
    for (size_t i = 0; i < rg.size();) {
      const int64_t s = rg[i].first;
      S->neighbors.push_back((int32_t)(s - 1));
      for (; i < rg.size() && rg[i].first == s; ++i) S->recv_idx.push_back(rg[i].second);
      S->recv_ptr.push_back((int64_t)S->recv_idx.size());
      if (si < sc.size() && sc[si].first < s) JH_THROW("halo plan: a send list without a matching receive list (internal error)");
      for (; si < sc.size() && sc[si].first == s; ++si) S->send_idx.push_back(g2l[sc[si].second] + 1);
      S->send_ptr.push_back((int64_t)S->send_idx.size());
    }
    if (si != sc.size()) JH_THROW("halo plan: a send list without a matching receive list (internal error)");
    *out = S.release();
  });
}

extern "C" int32_t jh_subdomain_sizes(jh_subdomain s, int64_t *sizes) {
  return guard([&] {
    if (!s || !sizes) JH_THROW("null argument");
    sizes[0] = s->n_owned;
    sizes[1] = s->n_local;
    sizes[2] = (int64_t)s->faces.size();
    sizes[3] = (int64_t)s->neighbors.size();
    sizes[4] = (int64_t)s->send_idx.size();
  });
}

extern "C" int32_t jh_subdomain_get(jh_subdomain s, int64_t *cells, int64_t *faces, int64_t *N_local, int32_t *neighbors,
                                    int64_t *send_ptr, int64_t *send_idx, int64_t *recv_ptr, int64_t *recv_idx) {
  return guard([&] {
    if (!s) JH_THROW("null argument");
    auto put = [](auto *dst, const auto &v) {
      if (!dst || v.empty()) return;
      parallel_ranges((int64_t)v.size(), 1 << 18, [&](int64_t b, int64_t e) { std::copy(v.begin() + b, v.begin() + e, dst + b); });
    };
    put(cells, s->cells);
    put(faces, s->faces);
    put(N_local, s->N_local);
    put(neighbors, s->neighbors);
    put(send_ptr, s->send_ptr);
    put(send_idx, s->send_idx);
    put(recv_ptr, s->recv_ptr);
    put(recv_idx, s->recv_idx);
  });
}

extern "C" int32_t jh_subdomain_destroy(jh_subdomain s) {
  return guard([&] { delete s; });
}

// Recursive coordinate bisection: the cells of a part are split at the median of the axis with the largest extent, sizes
// proportional to the part counts; ties by cell id (deterministic).  X: nc points of `dim` coordinates each, point-major.
extern "C" int32_t jh_partition_rcb(int64_t nc, int32_t dim, const double *X, int64_t nparts, int64_t *out) {
  return guard([&] {
    if (nc < 1 || dim < 1 || dim > 3 || !X || !out) JH_THROW("bad arguments");
    if (nparts < 1 || nparts > nc) JH_THROW("nparts must be in 1..nc");
    if (nc > 2000000000LL) JH_THROW("too many cells");
    // (a NaN would break the strict weak ordering nth_element relies on)
    parallel_ranges(nc * dim, 1 << 18, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i)
        if (!std::isfinite(X[i])) JH_THROW("jh_partition_rcb: non-finite coordinate of cell " + std::to_string(i / dim + 1));
    });
    std::vector<int32_t> idx;
    resize_parallel(idx, (size_t)nc);
    parallel_ranges(nc, 1 << 18, [&](int64_t b, int64_t e) { std::iota(idx.begin() + b, idx.begin() + e, (int32_t)b); });
    struct Job { int64_t b, e, k, first; };
    std::vector<Job> level{{0, nc, nparts, 1}};
    while (!level.empty()) {
      std::vector<Job> next;
      std::mutex m;
      // the jobs of one level are independent; the few large ones at the top also use the cores inside (extent scan)
      parallel_ranges((int64_t)level.size(), 1, [&](int64_t j0, int64_t j1) {
        for (int64_t j = j0; j < j1; ++j) {
          const Job J = level[j];
          const int64_t n = J.e - J.b;
          if (J.k == 1) {
            for (int64_t i = J.b; i < J.e; ++i) out[idx[i]] = J.first;
            continue;
          }
          double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
          {
            std::mutex mm;
            parallel_ranges(n, 1 << 18, [&](int64_t b, int64_t e) {
              double l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
              for (int64_t i = J.b + b; i < J.b + e; ++i)
                for (int d = 0; d < dim; ++d) {
                  const double v = X[(int64_t)idx[i] * dim + d];
                  l[d] = std::min(l[d], v);
                  h[d] = std::max(h[d], v);
                }
              std::lock_guard<std::mutex> lk(mm);
              for (int d = 0; d < dim; ++d) { lo[d] = std::min(lo[d], l[d]); hi[d] = std::max(hi[d], h[d]); }
            });
          }
          int ax = 0;
          for (int d = 1; d < dim; ++d) if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
          const int64_t kl = J.k / 2;
          int64_t nl = (int64_t)std::llround((double)n * (double)kl / (double)J.k);
          nl = std::max<int64_t>(kl, std::min<int64_t>(n - (J.k - kl), nl));
          // select on contiguous (coordinate, cell) pairs: the comparator stays in cache
          struct Key {
            double x;
            int32_t c;
            bool operator<(const Key &o) const { return x != o.x ? x < o.x : c < o.c; }
          };
          std::vector<Key> key;
          resize_parallel(key, (size_t)n);
          parallel_ranges(n, 1 << 18, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) key[i] = {X[(int64_t)idx[J.b + i] * dim + ax], idx[J.b + i]};
          });
          std::nth_element(key.begin(), key.begin() + nl, key.end());
          parallel_ranges(n, 1 << 18, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) idx[J.b + i] = key[i].c;
          });
          std::lock_guard<std::mutex> lk(m);
          next.push_back({J.b, J.b + nl, kl, J.first});
          next.push_back({J.b + nl, J.e, J.k - kl, J.first + kl});
        }
      });
      level.swap(next);
    }
  });
}
