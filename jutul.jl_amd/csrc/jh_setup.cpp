// Host-side setup of the TPFA discretisation: connectivity (a-1), pattern (a-3), positions (a-4), device
// ordering, CSR tiles.  Runs once per Simulator setup; everything here is integer work.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <queue>
#include <string>
#include <cstdlib>
#include <sys/mman.h>
#include <sys/resource.h>
#include <unistd.h>

#include "jh_internal.hpp"

namespace jh {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const std::string &last_error() { return g_err; }
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
void prefault_pages(void *p, size_t bytes) {
  static const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  if (!p || bytes < ((size_t)4 << 20)) return;
  const uintptr_t a = ((uintptr_t)p + page - 1) & ~(uintptr_t)(page - 1), e = ((uintptr_t)p + bytes) & ~(uintptr_t)(page - 1);
  if (e > a) (void)madvise((void *)a, (size_t)(e - a), MADV_POPULATE_WRITE);  // EINVAL on older kernels: first touch as before
}
double setup_faulted_mb() {
  struct rusage ru;
  if (getrusage(RUSAGE_SELF, &ru) != 0) return 0.0;
  return (double)ru.ru_minflt * (double)sysconf(_SC_PAGESIZE) / 1048576.0;
}

// --------------------------------------------------------------------------------------------------------------
// Pattern: tiles + upload
// --------------------------------------------------------------------------------------------------------------
void Pattern::build_tiles() {
  tile_row.clear();
  tile_row.push_back(0);
  // block entries carry bs*bs values: halve the entry budget of a tile for bs > 1 (keeps the assembly kernel's LDS
  // footprint near 30 KB so that several workgroups stay resident per CU)
  const int64_t max_nnz = bs == 1 ? TILE_NNZ : TILE_NNZ / 2;
  int64_t r = 0;
  while (r < n) {
    int64_t r1 = r;
    int64_t base = rowptr[r];
    // always take at least one row (a row longer than TILE_NNZ forms its own tile: "long row" path)
    ++r1;
    // a tile never straddles the interior / boundary limit of a rank-local subdomain (split SpMV, see jh_krylov.hip)
    const int64_t lim = (interior_rows > r) ? interior_rows : n;
    while (r1 < lim && (r1 - r) < TILE_ROWS && (rowptr[r1 + 1] - base) <= max_nnz) ++r1;
    tile_row.push_back((int32_t)r1);
    if (r1 == interior_rows) interior_tiles = (int32_t)tile_row.size() - 1;
    r = r1;
  }
  ntiles = (int32_t)tile_row.size() - 1;
  if (interior_rows == 0) interior_tiles = 0;
  tile_desc.resize((size_t)ntiles * 4);
  for (int32_t t = 0; t < ntiles; ++t) {
    const int32_t a = tile_row[t], b = tile_row[t + 1];
    tile_desc[4 * t + 0] = a;
    tile_desc[4 * t + 1] = b - a;
    tile_desc[4 * t + 2] = rowptr[a];
    tile_desc[4 * t + 3] = rowptr[b] - rowptr[a];
  }
}

void Pattern::upload() {
  hipStream_t s = ctx->stream;
  d_rowptr.upload(rowptr, s);
  d_col.upload(col, s);
  d_diag.upload(diag, s);
  d_tile_row.upload(tile_row, s);
  d_tile_desc.upload(tile_desc, s);
  if (!perm.empty()) d_perm.upload(perm, s);
  if (!nz_hslot.empty()) d_nz_hslot.upload(nz_hslot, s);
  stream_sync(s);
}

// --------------------------------------------------------------------------------------------------------------
// adjacency in host numbering (0-based), without the diagonal
// --------------------------------------------------------------------------------------------------------------
struct Adj {
  std::vector<int64_t> ptr;
  std::vector<int32_t> nbr;
  std::vector<int32_t> sface;  // signed face id: +(f+1) if this cell is N[1,f], -(f+1) otherwise
};

// Adjacency of the cells from the neighbourship N (2 x nf, 1-based), without the diagonal: ptr[nc + 1], and per entry the
// neighbour (0-based) and the face id -- signed (+(f+1) if the cell is N[1,f], -(f+1) otherwise) or plain f+1.  Rows in ascending
// face order.  self_loops_ok: faces with N[1,f] == N[2,f] are skipped instead of refused.  Nhost (optional): the 32-bit copy of N.
// The input numbering may be arbitrary (the bench grid's is scrambled): counting and filling cell by cell would be one cache miss
// per half-face, twice.  Instead the half-faces are first distributed over BUCKETS of consecutive cells (a counting sort on the
// high bits of the cell id: sequential reads, one write stream per bucket), then every bucket -- 16 384 cells, its slice of every
// array fits the L2 of a core -- is counted and filled on its own.  Threads take face ranges in ascending order and write a
// bucket's entries in thread order, so a cell's entries arrive in ascending face order: the rows are what a serial fill in face
// order produces, for any thread count, without a sort.
void build_adjacency_buckets(int64_t nc, int64_t nf, const int64_t *N, int32_t *Nhost, bool self_loops_ok, bool signed_faces,
                             std::vector<int64_t> &ptr, std::vector<int32_t> &nbr, std::vector<int32_t> &sface) {
  constexpr int64_t BW = 16384;  // cells per bucket (cell id within a bucket: 14 bits, kept in a 16-bit side array)
  const int64_t nbk = (nc + BW - 1) / BW;
  int nt = setup_threads();
  if (nf < (1 << 16)) nt = 1;
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, nf / 4096 + 1));
  std::vector<std::vector<int64_t>> hist((size_t)nt);
  parallel_team(nt, [&](int t, int) {
    std::vector<int64_t> &h = hist[(size_t)t];
    h.assign((size_t)nbk, 0);
    for (int64_t f = nf * t / nt, f1 = nf * (t + 1) / nt; f < f1; ++f) {
      const int64_t l = N[2 * f], r = N[2 * f + 1];
      if (l < 1 || l > nc || r < 1 || r > nc)
        JH_THROW("neighborship entry out of range (utils.jl:822: max(N) <= nc)");
      if (Nhost) { Nhost[2 * f] = (int32_t)l; Nhost[2 * f + 1] = (int32_t)r; }
      if (l == r) {
        if (self_loops_ok) continue;
        JH_THROW("face connecting a cell to itself is not supported");
      }
      h[(size_t)((l - 1) / BW)]++;
      h[(size_t)((r - 1) / BW)]++;
    }
  });
  std::vector<int64_t> bstart((size_t)nbk + 1, 0);
  for (int64_t b = 0; b < nbk; ++b) {  // per bucket: the threads' shares in thread order
    int64_t run = bstart[(size_t)b];
    for (int t = 0; t < nt; ++t) { const int64_t c = hist[(size_t)t][(size_t)b]; hist[(size_t)t][(size_t)b] = run; run += c; }
    bstart[(size_t)b + 1] = run;
  }
  const int64_t nhf = bstart[(size_t)nbk];
  resize_parallel(ptr, (size_t)nc + 1);
  resize_parallel(nbr, (size_t)nhf);
  resize_parallel(sface, (size_t)nhf);
  std::vector<uint16_t> lcell;  // cell of an entry, relative to its bucket
  resize_parallel(lcell, (size_t)nhf);
  parallel_team(nt, [&](int t, int) {  // entries in bucket order (inside a bucket: face order), written into the final arrays
    int64_t *cur = hist[(size_t)t].data();
    for (int64_t f = nf * t / nt, f1 = nf * (t + 1) / nt; f < f1; ++f) {
      const int64_t l = N[2 * f] - 1, r = N[2 * f + 1] - 1;
      if (l == r) continue;
      const int64_t pl = cur[l / BW]++;
      nbr[pl] = (int32_t)r; sface[pl] = (int32_t)(f + 1); lcell[pl] = (uint16_t)(l % BW);
      const int64_t pr = cur[r / BW]++;
      nbr[pr] = (int32_t)l; sface[pr] = signed_faces ? -(int32_t)(f + 1) : (int32_t)(f + 1); lcell[pr] = (uint16_t)(r % BW);
    }
  });
  ptr[0] = 0;
  parallel_ranges(nbk, 4, [&](int64_t b0, int64_t b1) {  // every bucket: count per cell, row pointers, stable scatter by cell
    std::vector<int32_t> cnt((size_t)BW + 1), tn, tf;
    for (int64_t b = b0; b < b1; ++b) {
      const int64_t e0 = bstart[(size_t)b], e1 = bstart[(size_t)b + 1], c0 = b * BW, ncb = std::min(BW, nc - c0);
      std::fill(cnt.begin(), cnt.begin() + ncb + 1, 0);
      for (int64_t e = e0; e < e1; ++e) cnt[(size_t)lcell[e] + 1]++;
      for (int64_t c = 0; c < ncb; ++c) { cnt[(size_t)c + 1] += cnt[(size_t)c]; ptr[c0 + c + 1] = e0 + cnt[(size_t)c + 1]; }
      tn.assign(nbr.begin() + e0, nbr.begin() + e1);
      tf.assign(sface.begin() + e0, sface.begin() + e1);
      for (int64_t e = e0; e < e1; ++e) {
        const int64_t w = e0 + cnt[(size_t)lcell[e]]++;
        nbr[w] = tn[(size_t)(e - e0)];
        sface[w] = tf[(size_t)(e - e0)];
      }
    }
  });
}
static Adj build_adjacency(int64_t nc, int64_t nf, const int64_t *N, int32_t *Nhost) {
  Adj A;
  build_adjacency_buckets(nc, nf, N, Nhost, false, true, A.ptr, A.nbr, A.sface);
  return A;
}

// --------------------------------------------------------------------------------------------------------------
// device ordering: compact blocks grown by BFS; blocks in creation order, cells in growth order
// --------------------------------------------------------------------------------------------------------------
// breadth-first order of one block's cells from a centre of the block (middle of a longest shortest path, two sweeps): the
// dependency depth of the block's triangular solves becomes its radius.  cells[0 .. n): the block's cells in ascending order.
// The three sweeps run on a private copy of the block's subgraph (local ids = positions in cells[], neighbour order as in A): the
// cells of a block are scattered over the host numbering, so every sweep over A itself would miss the cache on every cell; the
// copy costs one such pass.  slot[] (one entry per cell of the grid, any content on entry, shared by all threads: the blocks are
// disjoint) maps a cell to its position; an entry is trusted only if cells[] confirms it, so stale values of other blocks are harmless.
struct CentreScratch { std::vector<int32_t> lptr, lnbr, dist, bq, out; };
static void centre_bfs_order(const Adj &A, int32_t *cells, int64_t n, int32_t *slot, CentreScratch &S) {
  if (n < 3) return;
  for (int64_t j = 0; j < n; ++j) {
    if (j + 8 < n) __builtin_prefetch(&slot[cells[j + 8]], 1);
    __atomic_store_n(&slot[cells[j]], (int32_t)j, __ATOMIC_RELAXED);
  }
  S.lptr.resize((size_t)n + 1);
  S.lnbr.clear();
  S.lptr[0] = 0;
  for (int64_t j = 0; j < n; ++j) {
    if (j + 12 < n) __builtin_prefetch(&A.ptr[cells[j + 12]]);
    if (j + 8 < n) __builtin_prefetch(&A.nbr[A.ptr[cells[j + 8]]]);
    if (j + 4 < n) { const int32_t c4 = cells[j + 4]; for (int64_t k = A.ptr[c4]; k < A.ptr[c4 + 1]; ++k) __builtin_prefetch(&slot[A.nbr[k]]); }
    const int32_t c = cells[j];
    for (int64_t k = A.ptr[c]; k < A.ptr[c + 1]; ++k) {
      const int32_t o = A.nbr[k];
      const uint32_t q = (uint32_t)__atomic_load_n(&slot[o], __ATOMIC_RELAXED);
      if (q < (uint32_t)n && cells[q] == o) S.lnbr.push_back((int32_t)q);
    }
    S.lptr[(size_t)j + 1] = (int32_t)S.lnbr.size();
  }
  const int32_t *lp = S.lptr.data(), *ln = S.lnbr.data();
  S.dist.resize((size_t)n);
  int32_t *dist = S.dist.data();
  std::vector<int32_t> &bq = S.bq;
  auto sweep = [&](int32_t s) {
    bq.clear();
    bq.push_back(s);
    std::fill(dist, dist + n, -1);
    dist[s] = 0;
    for (size_t h = 0; h < bq.size(); ++h) {
      const int32_t c = bq[h];
      for (int32_t k = lp[c]; k < lp[c + 1]; ++k) {
        const int32_t o = ln[k];
        if (dist[o] < 0) { dist[o] = dist[c] + 1; bq.push_back(o); }
      }
    }
    return bq.back();
  };
  const int32_t u = sweep(0);
  const int32_t w = sweep(u);
  int32_t m = w;
  for (int32_t step = dist[w] / 2; step > 0; --step)
    for (int32_t k = lp[m]; k < lp[m + 1]; ++k) {
      const int32_t o = ln[k];
      if (dist[o] == dist[m] - 1) { m = o; break; }
    }
  sweep(m);
  // the reached piece first, the rest (not connected to it) in the given order
  S.out.clear();
  for (int32_t q : bq) S.out.push_back(cells[q]);
  if ((int64_t)bq.size() < n)
    for (int64_t j = 0; j < n; ++j) if (dist[j] < 0) S.out.push_back(cells[j]);
  std::copy(S.out.begin(), S.out.end(), cells);
}

namespace {
// JH_SETUP_TIMING=1: seconds per set-up phase on stderr
struct PhaseTimer {
  bool on = false;
  explicit PhaseTimer(bool enabled) : on(enabled) {}
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  double mb = setup_faulted_mb();
  void lap(const char *what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    const double m = setup_faulted_mb();
    fprintf(stderr, "[jutul_hip setup] %-28s %.3f s  %6.0f MB first-touched\n", what, std::chrono::duration<double>(n - t).count(), m - mb);
    t = n;
    mb = m;
  }
};
}  // namespace

// Blocks by recursive graph bisection + Fiduccia-Mattheyses refinement (jh_partition.cpp) -- the default.  Compact blocks cut
// fewer couplings than blocks grown along the rim of the assigned region ("onion", below): 13% instead of 18% of the
// half-faces at 512 cells per block on the tet lattice = 24.0 instead of 26.6 BiCGStab iterations at 2M cells, and a
// dependency depth of the radius, not the diameter.  Block ids follow the bisection tree (neighbouring blocks are close in
// memory); inside a block the cells are ordered breadth-first from a centre.
// fw, fw_scale: coupling per face (nf doubles; |.| x fw_scale has mean 1, non-finite values count as 0) or nullptr -- the cuts then prefer weak couplings (the
// reference partitions the |A|-weighted graph of the matrix for its block-Jacobi ILU(0), precond/ilu.jl:37-60 ->
// generate_metis_graph, partitioning.jl:64-78)
static void blocks_by_bisection(const Adj &A, int64_t nc, int64_t block_rows, std::vector<int32_t> &perm, std::vector<int32_t> &block_ptr, bool timing,
                                const double *fw, double fw_scale) {
  const int64_t nparts = std::max<int64_t>(1, (nc + block_rows / 2) / block_rows);
  const int64_t max_part = std::max<int64_t>(block_rows + block_rows / 8, (nc + nparts - 1) / nparts);
  std::vector<int32_t> label;
  resize_parallel(label, A.ptr.size() - 1);
  parallel_ranges((int64_t)label.size(), 1 << 18, [&](int64_t b, int64_t e) { std::fill(label.begin() + b, label.begin() + e, -1); });
  PhaseTimer pt(timing);
  // (on a breadth-first renumbering of the owned cells: jh_partition.cpp)
  partition_on_bfs_numbering(nc, A.ptr.data(), A.nbr.data(), nullptr, fw ? A.sface.data() : nullptr, fw, fw_scale, 1e-3, nparts, 0.04, max_part, label.data(),
                             [&](const char *what) { pt.lap(what); });
  resize_parallel(perm, (size_t)nc);
  counting_sort_indices(label.data(), nc, nparts, block_ptr, perm.data());  // every block's cells in ascending order
  // (the blocks are disjoint: one cell -> position array serves all threads, see centre_bfs_order)
  std::vector<int32_t> dist;
  resize_parallel(dist, A.ptr.size() - 1);
  parallel_ranges(nparts, 8, [&](int64_t b0, int64_t b1) {
    CentreScratch S;
    for (int64_t b = b0; b < b1; ++b) centre_bfs_order(A, perm.data() + block_ptr[b], block_ptr[b + 1] - block_ptr[b], dist.data(), S);
  });
  pt.lap("  blocks: order inside");
}

static void order_blocks(const Adj &A, int64_t nc_all, int64_t nc, int64_t block_rows, std::vector<int32_t> &perm,
                         std::vector<int32_t> &block_ptr, int64_t &interior_rows, int32_t &interior_blocks, bool onion, bool timing,
                         const double *fw, double fw_scale) {
  // cells >= nc (ghosts of a rank-local subdomain) are never absorbed; they form the last block
  perm.clear();
  perm.reserve(nc_all);
  block_ptr.assign(1, 0);
  // onion (option block_order = 1): round 1's blocks grown along the rim of the assigned region; default: graph bisection (above)
  if (!onion) blocks_by_bisection(A, nc, block_rows, perm, block_ptr, timing, fw, fw_scale);
  std::vector<int32_t> blk(nc_all, -1);
  for (int64_t c = nc; c < nc_all; ++c) blk[c] = INT32_MAX;
  std::vector<int32_t> cand;  // frontier candidates for the next seed (FIFO)
  size_t cand_head = 0;
  std::vector<int32_t> q;
  q.reserve(block_rows);
  // Order inside an onion block: by BFS layer from the seed and, inside a layer, by a greedy colouring of the layer's own
  // adjacencies.  ILU(0) depends on the ordering only through the orientation of the edges (which endpoint is eliminated first);
  // growth order would chain the cells of a layer one after the other (a 512-cell block has ~65 dependency levels), colouring the
  // layer keeps every cross-layer orientation and leaves (layers x colours) ~ 25 levels for the level-scheduled triangular solves.
  std::vector<int32_t> depth(onion ? nc_all : 0, 0), colour(onion ? nc_all : 0, -1), qpos(onion ? nc_all : 0, 0);
  std::vector<int32_t> qsorted;
  int64_t next_unassigned = 0;
  int32_t b = 0;
  while (onion && (int64_t)perm.size() < nc) {
    int32_t seed = -1;
    while (cand_head < cand.size()) {
      int32_t c = cand[cand_head++];
      if (blk[c] < 0) { seed = c; break; }
    }
    if (seed < 0) {
      while (next_unassigned < nc && blk[next_unassigned] >= 0) ++next_unassigned;
      seed = (int32_t)next_unassigned;
    }
    // grow
    q.clear();
    q.push_back(seed);
    blk[seed] = b;
    depth[seed] = 0;
    size_t head = 0;
    while (head < q.size() && (int64_t)q.size() < block_rows) {
      int32_t c = q[head++];
      for (int64_t k = A.ptr[c]; k < A.ptr[c + 1] && (int64_t)q.size() < block_rows; ++k) {
        int32_t o = A.nbr[k];
        if (blk[o] < 0) { blk[o] = b; q.push_back(o); depth[o] = depth[c] + 1; }
      }
    }
    {
      for (size_t i = 0; i < q.size(); ++i) { qpos[q[i]] = (int32_t)i; colour[q[i]] = -1; }
      for (int32_t c : q) {  // greedy colouring inside the BFS layer, in growth order
        unsigned used = 0;
        for (int64_t k = A.ptr[c]; k < A.ptr[c + 1]; ++k) {
          const int32_t o = A.nbr[k];
          if (blk[o] == b && depth[o] == depth[c] && colour[o] >= 0 && colour[o] < 32) used |= 1u << colour[o];
        }
        int col = 0;
        while (col < 31 && (used >> col) & 1u) ++col;
        colour[c] = col;
      }
      qsorted = q;
      std::sort(qsorted.begin(), qsorted.end(), [&](int32_t a, int32_t c) {
        if (depth[a] != depth[c]) return depth[a] < depth[c];
        if (colour[a] != colour[c]) return colour[a] < colour[c];
        return qpos[a] < qpos[c];
      });
    }
    // unassigned neighbours of the block become seed candidates
    for (int32_t c : q)
      for (int64_t k = A.ptr[c]; k < A.ptr[c + 1]; ++k)
        if (blk[A.nbr[k]] < 0) cand.push_back(A.nbr[k]);
    if (cand_head > (1u << 20) && cand_head * 2 > cand.size()) {  // compact the FIFO
      cand.erase(cand.begin(), cand.begin() + cand_head);
      cand_head = 0;
    }
    perm.insert(perm.end(), qsorted.begin(), qsorted.end());
    block_ptr.push_back((int32_t)perm.size());
    ++b;
  }
  // merge small fragments into the preceding block when the result stays within 5/4 of the target
  std::vector<int32_t> merged(1, 0);
  int64_t cap = block_rows + block_rows / 4;
  for (size_t i = 1; onion && i < block_ptr.size(); ++i) {
    int64_t sz = block_ptr[i] - block_ptr[i - 1];
    int64_t prev = merged.size() > 1 ? merged.back() - merged[merged.size() - 2] : 0;
    if (merged.size() > 1 && sz < block_rows / 4 && prev + sz <= cap)
      merged.back() = block_ptr[i];
    else
      merged.push_back(block_ptr[i]);
  }
  if (onion) block_ptr.swap(merged);
  // rank-local subdomain: blocks touching a ghost cell ("boundary" blocks) go behind the interior ones, so that the halo
  // exchange can run while the interior part of an ILU(0) apply / SpMV is computed.  Block-Jacobi ILU(0) does not depend
  // on the order of the blocks, and the order inside a block is kept.
  interior_rows = -1;
  interior_blocks = -1;
  if (nc < nc_all) {
    const size_t nb = block_ptr.size() - 1;
    std::vector<char> bnd(nb, 0);
    for (size_t b = 0; b < nb; ++b)
      for (int32_t i = block_ptr[b]; i < block_ptr[b + 1] && !bnd[b]; ++i) {
        const int32_t c = perm[i];
        for (int64_t k = A.ptr[c]; k < A.ptr[c + 1]; ++k)
          if (A.nbr[k] >= nc) { bnd[b] = 1; break; }
      }
    std::vector<int32_t> perm2, bp2(1, 0);
    perm2.reserve(nc_all);
    for (int pass = 0; pass < 2; ++pass) {
      for (size_t b = 0; b < nb; ++b)
        if (bnd[b] == pass) {
          perm2.insert(perm2.end(), perm.begin() + block_ptr[b], perm.begin() + block_ptr[b + 1]);
          bp2.push_back((int32_t)perm2.size());
        }
      if (pass == 0) { interior_rows = (int64_t)perm2.size(); interior_blocks = (int32_t)bp2.size() - 1; }
    }
    perm.swap(perm2);
    block_ptr.swap(bp2);
  }
  // ghost cells: their rows become -I (unit_diagonalize!), so any grouping works; keep the blocks LDS-sized
  for (int64_t c = nc; c < nc_all; ++c) {
    perm.push_back((int32_t)c);
    if ((c - nc + 1) % block_rows == 0 || c + 1 == nc_all) block_ptr.push_back((int32_t)(c + 1));
  }
}

static void order_by_partition(const Adj &A, int64_t nc, const int64_t *partition, std::vector<int32_t> &perm,
                               std::vector<int32_t> &block_ptr, bool bfs_inside = false) {
  int64_t np = 0;
  for (int64_t c = 0; c < nc; ++c) {
    if (partition[c] < 1) JH_THROW("partition ids must be >= 1 (par_ilu0.jl:49)");
    np = std::max(np, partition[c]);
  }
  std::vector<int64_t> cnt(np + 1, 0);
  for (int64_t c = 0; c < nc; ++c) cnt[partition[c]]++;
  block_ptr.assign(np + 1, 0);
  for (int64_t b = 0; b < np; ++b) block_ptr[b + 1] = block_ptr[b] + (int32_t)cnt[b + 1];
  perm.resize(nc);
  std::vector<int32_t> cur(block_ptr.begin(), block_ptr.end() - 1);
  for (int64_t c = 0; c < nc; ++c) perm[cur[partition[c] - 1]++] = (int32_t)c;  // findall order inside a part
  if (bfs_inside) {  // JH_REORDER_BLOCKS with a partition: the caller's blocks, ordered inside for short dependency chains
    // (the parts are disjoint: one cell -> position array serves all threads, as in blocks_by_bisection)
    std::vector<int32_t> dist;
    resize_parallel(dist, (size_t)nc);
    parallel_ranges(np, 8, [&](int64_t b0, int64_t b1) {
      CentreScratch S;
      for (int64_t b = b0; b < b1; ++b) centre_bfs_order(A, perm.data() + block_ptr[b], block_ptr[b + 1] - block_ptr[b], dist.data(), S);
    });
  }
}

}  // namespace jh

using namespace jh;


// --------------------------------------------------------------------------------------------------------------
// jh_tpfa_create
// --------------------------------------------------------------------------------------------------------------
extern "C" int32_t jh_tpfa_create(jh_context ctx, int64_t nc, int64_t nf, const int64_t *N, int32_t block_n,
                                  int32_t reorder, const int64_t *partition, int64_t block_rows, int64_t n_owned,
                                  jh_tpfa *out) {
  return jh_tpfa_create_weighted(ctx, nc, nf, N, nullptr, block_n, reorder, partition, block_rows, n_owned, out);
}
extern "C" int32_t jh_tpfa_create_weighted(jh_context ctx, int64_t nc, int64_t nf, const int64_t *N, const double *face_weights,
                                           int32_t block_n, int32_t reorder, const int64_t *partition, int64_t block_rows,
                                           int64_t n_owned, jh_tpfa *out) {
  return guard([&] {
    if (!ctx || !out) JH_THROW("null argument");
    if (nc < 1 || nf < 0) JH_THROW("bad sizes");
    // device indices are 32-bit (4-byte column ids are a third of the SpMV's index+value stream): one GPU holds the
    // pattern of up to 2^31 - 1 scalar non-zeros (~400M cells on a tet grid), beyond that the grid must be partitioned
    if (block_n < 1 || block_n > 3) JH_THROW("block_n must be 1..3");
    if (nc + 2 * nf > (int64_t)INT32_MAX / ((int64_t)block_n * block_n))
      JH_THROW("discretisation too large for 32-bit device indices: partition it across more ranks");
    if (nc > 2000000000LL || 2 * nf + nc > 2000000000LL) JH_THROW("problem too large for 32-bit device indices");
    jh::DeviceScope dev(ctx);  // (planning contexts: host phases only)
    auto d = std::make_unique<jh_tpfa_s>();
    d->ctx = ctx;
    d->nc = nc;
    d->nf = nf;
    d->nhf = 2 * nf;
    d->N = block_n;
    PhaseTimer pt(ctx->opt.setup_timing != 0);
    resize_parallel(d->Nhost, (size_t)(2 * nf));
    Adj A = build_adjacency(nc, nf, N, d->Nhost.data());  // (validates N and fills its 32-bit copy)
    pt.lap("adjacency");
    if (n_owned <= 0 || n_owned > nc) n_owned = nc;

    auto pat = std::make_shared<Pattern>();
    pat->ctx = ctx;
    pat->n = nc;
    pat->bs = block_n;
    if (partition) {
      order_by_partition(A, nc, partition, pat->perm, pat->block_ptr, reorder == JH_REORDER_BLOCKS);
    } else if (reorder == JH_REORDER_BLOCKS) {
      // default: 512-row blocks; below ~2M rows the ILU(0) apply is bound by per-block latency, not bandwidth, and twice
      // as many half-size blocks fill the chip better (1.25M cells: apply 48 -> 40 us at equal iteration counts)
      if (block_rows <= 0) {
        block_rows = nc < 2000000 ? 256 : 512;
        // The defaults are sized for tet / hex grids (<= 6 faces per cell: ~2500 entries per block).  Cells with many faces
        // (polyhedral / PEBI grids) get proportionally fewer rows per block, so that a block's factors and dependency levels
        // stay within what one workgroup holds in LDS.
        const double deg = nc > 0 ? (double)A.ptr[nc] / (double)nc : 0.0;
        // (2M-cell polyhedral grid, 15.5 faces per cell: 96 / 128 / 160 / 192 / 256 rows -> 90 / 86 / 92 / 77 / 85 Newton it/s; from
        // 192 rows the factorisation program of a block exceeds 64 KB of LDS: 3.1 -> 4.5 ms)
        if (deg > 7.0) block_rows = std::max<int64_t>(64, (int64_t)(block_rows * 5.0 / deg) / 32 * 32);
      }
      // |coupling| per face for the bisection, mean 1 (the refinement's give-up thresholds are absolute); non-finite -> 0
      // (no scaled copy of the weights: the factor goes with the pointer and is applied where the partitioner's graph is built)
      const double *fw = nullptr;
      double fw_scale = 1.0;
      if (face_weights && nf > 0 && ctx->opt.block_weights != 0) {
        double sum = 0.0;
        for (int64_t f = 0; f < nf; ++f) { const double v = std::fabs(face_weights[f]); sum += std::isfinite(v) ? v : 0.0; }
        if (sum > 0.0) { fw = face_weights; fw_scale = (double)nf / sum; }
      }
      order_blocks(A, nc, n_owned, block_rows, pat->perm, pat->block_ptr, pat->interior_rows, pat->interior_blocks,
                   ctx->opt.block_order == 1, ctx->opt.setup_timing != 0, fw, fw_scale);
    } else if (reorder != JH_REORDER_NONE) {
      JH_THROW("unknown reorder mode");
    }
    pt.lap("device ordering (blocks)");
    bool ident = pat->perm.empty();
    if (!ident)
      for (int64_t i = n_owned; i < nc; ++i)
        if (pat->perm[i] < n_owned) JH_THROW("partition must keep the ghost cells (>= n_owned) as the last device rows");
    if (!ident) {
      resize_parallel(pat->iperm, (size_t)nc);
      parallel_ranges(nc, 1 << 16, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) pat->iperm[pat->perm[i]] = (int32_t)i; });
    }
    // device CSR: row i = host cell perm[i]; columns mapped to device numbering, ascending; diagonal present
    resize_parallel(pat->rowptr, (size_t)nc + 1);
    pat->rowptr[0] = 0;
    parallel_ranges(nc, 1 << 16, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) {
        const int64_t h = ident ? i : pat->perm[i];
        pat->rowptr[i + 1] = (int32_t)(A.ptr[h + 1] - A.ptr[h]) + 1;
      }
    });
    for (int64_t i = 0; i < nc; ++i) pat->rowptr[i + 1] += pat->rowptr[i];
    pat->nnzb = pat->rowptr[nc];
    d->nnzb = pat->nnzb;
    resize_parallel(pat->col, (size_t)pat->nnzb);
    resize_parallel(pat->diag, (size_t)nc);
    resize_parallel(d->nz_face, (size_t)pat->nnzb);
    // host CSR (host numbering) rowptr to compute host slots: row h has (distinct neighbours of h) + 1 entries, ascending host cols
    std::vector<int64_t> hrp;
    resize_parallel(hrp, (size_t)nc + 1);
    bool multigraph = false;
    {
      std::vector<int32_t> ucount;
      resize_parallel(ucount, (size_t)nc);
      std::vector<char> dup_in_range;
      parallel_ranges(nc, 65536, [&](int64_t b, int64_t e) {
        // distinct neighbours of every host cell (rows are short: sorted by insertion on the stack; long rows take std::sort)
        int32_t small[64];
        std::vector<int32_t> big;
        for (int64_t h = b; h < e; ++h) {
          const int64_t k0 = A.ptr[h], deg = A.ptr[h + 1] - k0;
          int32_t *nb = small;
          if (deg > 64) { big.assign(A.nbr.begin() + k0, A.nbr.begin() + k0 + deg); std::sort(big.begin(), big.end()); nb = big.data(); }
          else
            for (int64_t j = 0; j < deg; ++j) {
              const int32_t v = A.nbr[k0 + j];
              int64_t q = j;
              for (; q > 0 && small[q - 1] > v; --q) small[q] = small[q - 1];
              small[q] = v;
            }
          int32_t u = deg > 0;
          for (int64_t j = 1; j < deg; ++j) u += nb[j] != nb[j - 1];
          ucount[h] = u;
        }
      });
      for (int64_t h = 0; h < nc; ++h) {
        hrp[h + 1] = hrp[h] + ucount[h] + 1;
        if (ucount[h] != (int32_t)(A.ptr[h + 1] - A.ptr[h])) multigraph = true;
      }
    }
    pat->nnzb_host = hrp[nc];
    if (!ident || multigraph) resize_parallel(pat->nz_hslot, (size_t)pat->nnzb);
    if (multigraph) pat->shadow_flag.assign(pat->nnzb, 0);
    // rows are independent: sorted device columns, signed face ids, diagonal slot, host slots -- on all host cores.  Device row i
    // is host cell perm[i], an arbitrary place of the host arrays: the loop runs ahead of itself with prefetches (row pointer of
    // row i + 12, neighbour list of row i + 8, the neighbours' device numbers of row i + 4).
    struct Ent { int32_t col, sf, hcol; };
    parallel_ranges(nc, 4096, [&](int64_t r_begin, int64_t r_end) {
      Ent small[64];
      std::vector<Ent> big;
      int32_t rk[64];
      std::vector<int32_t> rkbig;
      const int32_t *perm = ident ? nullptr : pat->perm.data(), *iperm = ident ? nullptr : pat->iperm.data();
      for (int64_t i = r_begin; i < r_end; ++i) {
        if (!ident) {
          if (i + 12 < r_end) __builtin_prefetch(&A.ptr[perm[i + 12]]);
          if (i + 8 < r_end) { const int64_t k8 = A.ptr[perm[i + 8]]; __builtin_prefetch(&A.nbr[k8]); __builtin_prefetch(&A.sface[k8]); }
          if (i + 4 < r_end) { const int32_t h4 = perm[i + 4]; for (int64_t k = A.ptr[h4]; k < A.ptr[h4 + 1]; ++k) __builtin_prefetch(&iperm[A.nbr[k]]); }
        }
        const int64_t h = ident ? i : perm[i];
        const int64_t k0 = A.ptr[h], deg = A.ptr[h + 1] - k0, len = deg + 1;
        Ent *t = small;
        if (len > 64) { big.resize((size_t)len); t = big.data(); }
        // ascending device columns; parallel faces (multigraph) by ascending face id: the last one is the pair's primary slot
        auto before = [](const Ent &x, const Ent &y) { return x.col != y.col ? x.col < y.col : std::abs(x.sf) < std::abs(y.sf); };
        t[0] = Ent{(int32_t)i, 0, (int32_t)h};
        for (int64_t j = 0; j < deg; ++j) {
          const int32_t o = A.nbr[k0 + j];
          const Ent v{ident ? o : iperm[o], A.sface[k0 + j], o};
          if (len > 64) { t[j + 1] = v; continue; }
          int64_t q = j + 1;
          for (; q > 0 && before(v, t[q - 1]); --q) t[q] = t[q - 1];
          t[q] = v;
        }
        if (len > 64) std::sort(t, t + len, before);
        const int32_t base = pat->rowptr[i];
        for (int64_t j = 0; j < len; ++j) {
          pat->col[base + j] = t[j].col;
          d->nz_face[base + j] = t[j].sf;
          if (t[j].col == (int32_t)i) pat->diag[i] = base + (int32_t)j;
          if (multigraph && j + 1 < len && t[j + 1].col == t[j].col) pat->shadow_flag[base + j] = 1;
        }
        if (!ident || multigraph) {
          // host slot = position of host column in host row h (ascending distinct host columns incl. diagonal); shadow slots
          // map to the spare slot behind the host pattern (jh_csr_get/set_values)
          int32_t *r = rk;
          if (len > 64) { rkbig.resize((size_t)len); r = rkbig.data(); }
          for (int64_t j = 0; j < len; ++j) r[j] = (int32_t)j;
          // entries by (host column, position): equal host columns (parallel faces) share a rank
          auto hless = [&](int32_t x, int32_t y) { return t[x].hcol != t[y].hcol ? t[x].hcol < t[y].hcol : x < y; };
          if (len > 64) std::sort(r, r + len, hless);
          else
            for (int64_t j = 1; j < len; ++j) {
              const int32_t v = r[j];
              int64_t q = j;
              for (; q > 0 && hless(v, r[q - 1]); --q) r[q] = r[q - 1];
              r[q] = v;
            }
          int64_t rank = -1;
          for (int64_t j = 0; j < len; ++j) {
            if (j == 0 || t[r[j]].hcol != t[r[j - 1]].hcol) ++rank;
            const int32_t slot = base + r[j];
            pat->nz_hslot[slot] = (multigraph && pat->shadow_flag[slot]) ? (int32_t)pat->nnzb_host : (int32_t)(hrp[h] + rank);
          }
        }
      }
    });
    for (int64_t i = 0; i < nc; ++i)
      if (pat->rowptr[i + 1] - pat->rowptr[i] > TILE_NNZ / 2) JH_THROW("cell with more than 511 faces is not supported");
    if (multigraph) {
      for (int64_t k = 0; k < pat->nnzb; ++k)
        if (pat->shadow_flag[k]) pat->shadow_slots.push_back((int32_t)k);
      pat->d_shadow_slots.upload(pat->shadow_slots, ctx->stream);
    }
    pt.lap("device CSR pattern");
    pat->build_tiles();
    pat->upload();
    pt.lap("tiles + upload");
    d->d_nz_face.upload(d->nz_face, ctx->stream);
    jh::stream_sync(ctx->stream);
    d->pat = pat;
    *out = d.release();
  });
}

extern "C" int32_t jh_tpfa_destroy(jh_tpfa d) {
  return guard([&] {
    if (!d) return;
    auto &H = d->halo;  // push-halo mappings
    for (double *p : H.peer_landing)
      if (p && p != H.landing) (void)hipIpcCloseMemHandle(p);
    if (H.landing) (void)hipFree(H.landing);
    delete d;
  });
}

extern "C" int32_t jh_tpfa_sizes(jh_tpfa d, int64_t *nc, int64_t *nf, int64_t *nhf, int64_t *nnzb, int32_t *block_n) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    if (nc) *nc = d->nc;
    if (nf) *nf = d->nf;
    if (nhf) *nhf = d->nhf;
    if (nnzb) *nnzb = d->pat->nnzb_host;  // the reference's pattern: one entry per cell pair
    if (block_n) *block_n = d->N;
  });
}

// --------------------------------------------------------------------------------------------------------------
// Reference-exact host tables (1-based), built on demand for the getters / the Julia glue.
//   get_facepos (utils.jl:813-874), get_connection (flux.jl:128-142), pattern (conservation.jl:486-505 +
//   StaticCSR/mat.jl:73-76), positions (conservation.jl:143-216, equations.jl:95-113,163-174).
// --------------------------------------------------------------------------------------------------------------
void jh_tpfa_s::build_tables() {
  if (tables_built) return;
  const int32_t *Nn = Nhost.data();
  face_pos.assign(nc + 1, 0);
  for (int i = 0; i < 2; ++i)
    for (int64_t j = 0; j < nf; ++j) face_pos[Nn[2 * j + i]]++;
  face_pos[0] = 1;
  for (int64_t c = 0; c < nc; ++c) face_pos[c + 1] += face_pos[c];
  hf_face.resize(nhf);
  {
    std::vector<int64_t> cur(nc);
    for (int64_t c = 0; c < nc; ++c) cur[c] = face_pos[c] - 1;
    for (int i = 0; i < 2; ++i)
      for (int64_t j = 0; j < nf; ++j) hf_face[cur[Nn[2 * j + i] - 1]++] = j + 1;
    for (int64_t c = 0; c < nc; ++c) std::sort(hf_face.begin() + (face_pos[c] - 1), hf_face.begin() + (face_pos[c + 1] - 1));
  }
  hf_self.resize(nhf);
  hf_other.resize(nhf);
  hf_sign.resize(nhf);
  for (int64_t c = 1; c <= nc; ++c)
    for (int64_t k = face_pos[c - 1]; k <= face_pos[c] - 1; ++k) {
      int64_t f = hf_face[k - 1];
      int64_t l = Nn[2 * (f - 1)], r = Nn[2 * (f - 1) + 1];
      hf_self[k - 1] = c;
      if (l == c) { hf_sign[k - 1] = 1; hf_other[k - 1] = r; }
      else { hf_sign[k - 1] = -1; hf_other[k - 1] = l; }
    }
  // pattern: per row sorted unique {others} U {row}
  h_rowptr.assign(nc + 1, 1);
  h_colidx.clear();
  h_colidx.reserve(nnzb);
  std::vector<int64_t> tmp;
  for (int64_t c = 1; c <= nc; ++c) {
    tmp.clear();
    for (int64_t k = face_pos[c - 1]; k <= face_pos[c] - 1; ++k) tmp.push_back(hf_other[k - 1]);
    tmp.push_back(c);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    h_colidx.insert(h_colidx.end(), tmp.begin(), tmp.end());
    h_rowptr[c] = h_rowptr[c - 1] + (int64_t)tmp.size();
  }
  // positions, BlockMajorLayout flat index: (pos-1)*N^2 + N*(d-1) + e
  auto find_pos = [&](int64_t row, int64_t colv) -> int64_t {
    for (int64_t p = h_rowptr[row - 1]; p <= h_rowptr[row] - 1; ++p)
      if (h_colidx[p - 1] == colv) return p;
    return 0;
  };
  int64_t NN = (int64_t)N * N;
  pos_acc.resize(NN * nc);
  pos_flux.resize(NN * nhf);
  for (int64_t c = 1; c <= nc; ++c) {
    int64_t pos = find_pos(c, c);
    for (int e = 1; e <= N; ++e)
      for (int dd = 1; dd <= N; ++dd) pos_acc[NN * (c - 1) + (e - 1) * N + (dd - 1)] = (pos - 1) * NN + N * (dd - 1) + e;
  }
  for (int64_t k = 1; k <= nhf; ++k) {
    int64_t pos = find_pos(hf_other[k - 1], hf_self[k - 1]);
    for (int e = 1; e <= N; ++e)
      for (int dd = 1; dd <= N; ++dd) pos_flux[NN * (k - 1) + (e - 1) * N + (dd - 1)] = (pos - 1) * NN + N * (dd - 1) + e;
  }
  tables_built = true;
}

extern "C" int32_t jh_tpfa_get_conn(jh_tpfa d, int64_t *face_pos, int64_t *self, int64_t *other, int64_t *face,
                                    int64_t *face_sign) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    d->build_tables();
    if (face_pos) std::memcpy(face_pos, d->face_pos.data(), sizeof(int64_t) * (d->nc + 1));
    if (self) std::memcpy(self, d->hf_self.data(), sizeof(int64_t) * d->nhf);
    if (other) std::memcpy(other, d->hf_other.data(), sizeof(int64_t) * d->nhf);
    if (face) std::memcpy(face, d->hf_face.data(), sizeof(int64_t) * d->nhf);
    if (face_sign) std::memcpy(face_sign, d->hf_sign.data(), sizeof(int64_t) * d->nhf);
  });
}

extern "C" int32_t jh_tpfa_get_pattern(jh_tpfa d, int64_t *rowptr, int64_t *colidx) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    d->build_tables();
    if (rowptr) std::memcpy(rowptr, d->h_rowptr.data(), sizeof(int64_t) * (d->nc + 1));
    if (colidx) std::memcpy(colidx, d->h_colidx.data(), sizeof(int64_t) * d->h_colidx.size());
  });
}

extern "C" int32_t jh_tpfa_get_positions(jh_tpfa d, int64_t *pos_acc, int64_t *pos_flux) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    d->build_tables();
    if (pos_acc) std::memcpy(pos_acc, d->pos_acc.data(), sizeof(int64_t) * d->pos_acc.size());
    if (pos_flux) std::memcpy(pos_flux, d->pos_flux.data(), sizeof(int64_t) * d->pos_flux.size());
  });
}

extern "C" int32_t jh_tpfa_get_ordering(jh_tpfa d, int64_t *perm, int64_t *nblocks, int64_t *block_ptr,
                                        int64_t block_ptr_cap) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    const Pattern &P = *d->pat;
    if (perm)
      for (int64_t i = 0; i < d->nc; ++i) perm[i] = (P.perm.empty() ? i : P.perm[i]) + 1;
    int64_t nb = P.block_ptr.empty() ? 0 : (int64_t)P.block_ptr.size() - 1;
    if (nblocks) *nblocks = nb;
    if (block_ptr) {
      if (block_ptr_cap < nb + 1) JH_THROW("block_ptr buffer too small");
      for (int64_t i = 0; i <= nb; ++i) block_ptr[i] = P.block_ptr[i];
    }
  });
}

extern "C" int32_t jh_tpfa_get_split(jh_tpfa d, int64_t *interior_rows, int64_t *interior_blocks, int64_t *interior_tiles) {
  return guard([&] {
    if (!d) JH_THROW("null handle");
    const Pattern &P = *d->pat;
    if (interior_rows) *interior_rows = P.interior_rows;
    if (interior_blocks) *interior_blocks = P.interior_blocks;
    if (interior_tiles) *interior_tiles = P.interior_tiles;
  });
}

extern "C" int32_t jh_last_error(char *buf, int64_t cap) {
  const std::string &e = jh::last_error();
  if (buf && cap > 0) {
    int64_t n = std::min<int64_t>(cap - 1, (int64_t)e.size());
    std::memcpy(buf, e.data(), n);
    buf[n] = 0;
  }
  return (int32_t)e.size();
}

extern "C" int32_t jh_version(void) { return 100; }
