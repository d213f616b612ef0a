// Jagged-slice SpMV for the Krylov loop (a-10; mul!, StaticCSR/mat.jl:24-61).
//
// The CSR tile kernel (jh_kernels.hip) pays for the generality of CSR on every product: row pointers, products staged in LDS,
// three workgroup barriers per tile.  A TPFA Jacobian has short rows of nearly equal length (a cell and its faces), and inside
// the Krylov loop its values do not change.  So every solve starts by copying the values into a jagged-slice layout (one
// 18-byte-per-entry pass, ~1.2 SpMVs) and then multiplies out of that:
//   * rows in slices of 64 = one wavefront, one lane per row;
//   * inside a slice the rows are sorted by length (the lane -> row map is one byte per row) and the entries are stored
//     "diagonal by diagonal": the j-th entries of all rows that have one are consecutive.  Lane l reads entry j at
//     base_j + l: every load of the value / column stream is one fully coalesced 512 / 256-byte wavefront access, and there is
//     no padding whatever the length distribution is;
//   * the lane accumulates its row left -> right in registers -- the reference's order (ascending columns, mat.jl:41-61), so
//     the result has the same bits as the CSR kernel's -- and stores y directly.  No row pointers, no LDS, no barrier.
// Waves are persistent over an XCD-contiguous range of slices and software-pipelined: the descriptor of slice i+2 and the
// entries of slice i+1 are in flight while slice i gathers x and accumulates.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <unordered_map>

#include "jh_internal.hpp"

namespace jh {

void Pattern::build_jagged() {
  if (jag.built) return;
  jag.built = true;
  jag.usable = false;
  if (bs != 1 || n == 0) return;
  int32_t kmax = 0;
  for (int64_t i = 0; i < n; ++i) kmax = std::max(kmax, rowptr[i + 1] - rowptr[i]);
  if (kmax > JDS_KMAX) return;
  const int32_t ns = (int32_t)((n + 63) / 64);
  // 16-bit codes pay where the product streams from HBM for long (10M rows: 0.177 instead of 0.181-0.188 ms); on a small matrix
  // the extra pipeline stage costs more than the bytes save (1.25M rows: 28.5 instead of 27.1 us)
  const int64_t force = ctx->opt.spmv_col_bits;  // 16 / 32: test switch (read when the layout is built)
  const bool col16 = force ? force == 16 : n >= 3000000;
  // (only the column array of the chosen width is built: the other one is 100 / 200 MB at 10M rows)
  std::vector<int32_t> base(ns + 1, 0), jc(col16 ? 0 : nnzb + 64, 0);  // + 64: lanes past a diagonal's count load too
  std::vector<uint16_t> src;
  assign_prefaulted(src, (size_t)nnzb + 64, (uint16_t)0);                // CSR slot of a jagged entry, relative to the first slot of its slice (< 512)
  std::vector<uint8_t> cnt((size_t)ns * 16, 0), perm((size_t)ns * 64, 0);
  // 16-bit column codes: a slice's columns lie almost all in a window around its rows (the device order keeps neighbours close);
  // the few outside go to a per-slice list
  std::vector<uint16_t> jc16;
  assign_prefaulted(jc16, (size_t)(col16 ? nnzb + 64 : 0), (uint16_t)0);
  std::vector<int32_t> win((size_t)ns * 2, 0), far;
  // Slices are independent (no padding: slice s starts at entry rowptr[64 s]); only the lists of far columns are appended in slice
  // order -- every range of slices collects its own and the ranges are concatenated in order afterwards.  All host cores.
  struct FarPart { int64_t s0, s1; std::vector<int32_t> far; };
  std::vector<FarPart> parts;
  std::mutex parts_m;
  parallel_ranges(ns, 2048, [&](int64_t s_begin, int64_t s_end) {
    FarPart P{s_begin, s_end, {}};
    int lanes[64];
    for (int64_t s = s_begin; s < s_end; ++s) {
      const int64_t r0 = s * 64;
      const int64_t cb = r0 - JDS_BACK;
      const size_t far0 = P.far.size();
      win[(size_t)s * 2] = (int32_t)cb;
      win[(size_t)s * 2 + 1] = (int32_t)far0;  // (relative to the range's list until the lists are joined)
      const int nr = (int)std::min<int64_t>(64, n - r0);
      std::iota(lanes, lanes + nr, 0);
      std::stable_sort(lanes, lanes + nr, [&](int a, int b) {
        return rowptr[r0 + a + 1] - rowptr[r0 + a] > rowptr[r0 + b + 1] - rowptr[r0 + b];
      });
      int64_t pos = rowptr[r0];
      base[s] = (int32_t)pos;
      for (int l = 0; l < nr; ++l) perm[(size_t)s * 64 + l] = (uint8_t)lanes[l];
      for (int j = 0; j < kmax; ++j) {
        int c = 0;
        for (int l = 0; l < nr; ++l) {
          const int64_t row = r0 + lanes[l];
          if (rowptr[row + 1] - rowptr[row] > j) {
            const int32_t cj = col[rowptr[row] + j];
            if (!col16) {
              jc[pos] = cj;
            } else if (cj >= cb && cj - cb < JDS_FAR) {
              jc16[pos] = (uint16_t)(cj - cb);
            } else {
              jc16[pos] = (uint16_t)(JDS_FAR + (P.far.size() - far0));  // (a slice has at most 512 entries)
              P.far.push_back(cj);
            }
            src[pos] = (uint16_t)(rowptr[row] + j - rowptr[r0]);
            ++pos;
            ++c;
          } else {
            break;  // sorted by length: no later lane has a j-th entry either
          }
        }
        cnt[(size_t)s * 16 + j] = (uint8_t)c;
      }
    }
    std::lock_guard<std::mutex> lk(parts_m);
    parts.push_back(std::move(P));
  });
  std::sort(parts.begin(), parts.end(), [](const FarPart &a, const FarPart &b) { return a.s0 < b.s0; });
  for (const FarPart &P : parts) {
    const int32_t off = (int32_t)far.size();
    if (off) for (int64_t s = P.s0; s < P.s1; ++s) win[(size_t)s * 2 + 1] += off;
    far.insert(far.end(), P.far.begin(), P.far.end());
  }
  const int64_t pos = rowptr[n];
  base[ns] = (int32_t)pos;
  jag.nslices = ns;
  jag.kmax = kmax;
  jag.nent = pos;
  hipStream_t st = ctx->stream;
  jag.d_base.upload(base, st);
  jag.d_cnt.upload(cnt, st);
  jag.d_perm.upload(perm, st);
  if (col16) {
    far.resize(far.size() + (0x10000 - JDS_FAR), 0);  // any code of any slice decodes to a valid position
    jag.d_col16.upload(jc16, st);
    jag.d_win.upload(win, st);
    jag.d_far.upload(far, st);
  } else {
    jag.d_col.upload(jc, st);
  }
  jag.d_src.upload(src, st);
  // stream policy of the 32-bit kernel (see stream_load): the Krylov loop's working set -- jagged values and columns, the ILU(0)
  // factors (about as many), ~10 vectors -- against the Infinity Cache
  // (measured on the bench lattice: plain loads win up to 1.25M rows, the two tie at 1.6M = 256 MB by this count, non-temporal wins
  // from 2M; the rank-local matrix of an 8-rank 10M-cell run, ghost rows included, counts 211 MB)
  const int64_t nt = ctx->opt.spmv_nontemporal;
  jag.nontemporal = nt >= 0 ? nt != 0 : 12.0 * (double)nnzb + 100.0 * (double)n > 235e6;
  stream_sync(st);
  jag.usable = true;
}

namespace {

// Value / column stream of the product: read once per launch.  NT: non-temporal loads, so that the vectors keep the L2 / Infinity
// Cache -- right when matrix, factors and vectors together pass the 256 MB Infinity Cache (10M rows: 0.160 vs 0.165 ms, 2M rows:
// 38.4 vs 41.0 us and the ILU apply next to it 42.6 vs 45.7 us); when they fit (the 1.25M-row share of an 8-GPU run) plain loads
// let the matrix stay on the die between the two products of an iteration: 23.7 vs 26.0 us, 119 vs 124 us per BiCGStab iteration.
template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// A slice's entries are one contiguous range of the CSR array (64 rows of at most 8 entries): a wavefront stages that range in LDS
// with coalesced loads and writes the slice's jagged range from there -- value array read once and written once, plus two
// bytes of map per entry (the gather val[src[i]] straight from global memory moved 1.41 GB for 1.0 GB at 10M rows).
__global__ __launch_bounds__(256) void jagged_copy_kernel(double *__restrict__ jval, const double *__restrict__ val,
                                                          const uint16_t *__restrict__ src, const int32_t *__restrict__ base,
                                                          const int32_t *__restrict__ rowptr, int nslices, int nrows) {
  __shared__ double buf[4][64 * JDS_KMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int s = blockIdx.x * 4 + w; s < nslices; s += gridDim.x * 4) {  // wave-uniform
    const int c0 = rowptr[s * 64], cnt = rowptr[min(s * 64 + 64, nrows)] - c0, b0 = base[s];
    for (int i = lane; i < cnt; i += 64) buf[w][i] = val[c0 + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < cnt; i += 64) jval[b0 + i] = buf[w][src[b0 + i]];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

struct JDesc {
  int base;
  uint4 c;  // 16 counts, one byte each
};
template <int KU>
struct JEnt {
  int col[KU];
  double val[KU];
  int prow;
};
template <int J>
__device__ __forceinline__ int jcount(const uint4 &c) {
  const unsigned w = J < 4 ? c.x : (J < 8 ? c.y : (J < 12 ? c.z : c.w));
  return (int)((w >> (8 * (J & 3))) & 0xffu);
}

// Measured on the 10M-cell bench matrix (tools/spmv_probe.py; CSR tile kernel 189 us): lane = row with plain loads and a
// store in sorted-by-length lane order 173 us; value / column stream with non-temporal loads (read once: keeps the vectors in
// L2 / Infinity Cache) -8 us; results transposed through LDS so that y leaves in row order -10 us: 154 us.  An LDS window of x
// for the near columns (what the CSR tile kernel does) made it SLOWER here (160 us: the workgroup barrier and the LDS round
// trip cost more than the vector L1 serving the gathers); with every column served from LDS it would be 145 us, i.e. the
// 18% far columns cost ~15 us.  The four wavefronts of a workgroup take four consecutive slices and meet at one barrier per
// step: kept in step they share the vector-L1 lines of their neighbouring rows' x entries (free-running wavefronts: 165 us).
// NW: wavefronts per workgroup = slices per step.  One dot partial per workgroup: 16 wavefronts keep the partials of a full-chip
// launch within PEND_MAX, so that the kernel that consumes the dot can sum them itself (PendSum) instead of a reduction launch.
template <int NW>
__device__ __forceinline__ void jds_dot_partial(double d0, double d1, int dotv, double *red, double *part, size_t pstride) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); if (dotv == 2) d1 += __shfl_down(d1, off, 64); }
  if (lane == 0) { red[w] = d0; red[NW + w] = d1; }
  __syncthreads();
  if (tid == 0) {  // fixed order: pairs of neighbours, then pairs of pairs, ...
    double a[NW], b[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) { a[i] = red[i]; b[i] = red[NW + i]; }
#pragma unroll
    for (int m = NW; m > 1; m >>= 1) {
#pragma unroll
      for (int i = 0; i < m / 2; ++i) { a[i] = a[2 * i] + a[2 * i + 1]; b[i] = b[2 * i] + b[2 * i + 1]; }
    }
    part[blockIdx.x] = a[0];
    if (dotv == 2) part[pstride + blockIdx.x] = b[0];
  }
}
// Steps of a persistent product workgroup: workgroup b runs on XCD b % 8 (observed placement, performance only) and every XCD
// sweeps one contiguous eighth of the super-slices (NW slices = 64 NW rows, one per wavefront of the workgroup).  FOLD (push-halo
// hand-shake inside the kernel, HaloFold): workgroup 0 is the exchanger and takes no slices; the others sweep an eighth of the
// INTERIOR super-slices first and an eighth of the rest -- boundary and ghost rows, the last device rows -- afterwards, behind
// halo_fold_wait.
template <int NW, bool FOLD>
struct JSteps {
  int ilo, ni, blo, nb;  // this XCD's interior / other super-slices
  int t0, dt;            // first step and step stride of this workgroup
  int w, nslices;
  __device__ __forceinline__ JSteps(int nslices_, int interior_rows, int w_) : w(w_), nslices(nslices_) {
    const int nsup = (nslices + NW - 1) / NW;
    const int xcd = blockIdx.x % NUM_XCD, wg = blockIdx.x / NUM_XCD, wgs = gridDim.x / NUM_XCD;
    const int sbs = FOLD ? min(nsup, max(0, interior_rows) / (64 * NW)) : nsup;
    const int ci = (sbs + NUM_XCD - 1) / NUM_XCD, cb = (nsup - sbs + NUM_XCD - 1) / NUM_XCD;
    ilo = xcd * ci;
    ni = max(0, min(sbs, ilo + ci) - ilo);
    blo = sbs + xcd * cb;
    nb = max(0, min(nsup, blo + cb) - blo);
    const int ex = (FOLD && xcd == 0) ? 1 : 0;  // the exchanger is workgroup 0 of XCD 0
    t0 = wg - ex;
    dt = wgs - ex;
  }
  __device__ __forceinline__ int steps() const { return ni + nb; }
  // slice of this wavefront at step t; >= nslices: none
  __device__ __forceinline__ int slice(int t) const {
    if (t >= ni + nb) return 0x3fffffff;
    return NW * (t < ni ? ilo + t : blo + (t - ni)) + w;
  }
};
template <int KU, int DOT, int NW, bool NT, bool FOLD>
__global__ __launch_bounds__(64 * NW) void spmv_jds_kernel(const int32_t *__restrict__ base, const uint4 *__restrict__ cnt16,
                                                       const uint8_t *__restrict__ perm, const int32_t *__restrict__ jcol,
                                                       const double *__restrict__ jval, int nslices, int nrows,
                                                       const double *__restrict__ x, double *__restrict__ y, double alpha, double beta,
                                                       const double *__restrict__ dw, int dot_rows, double *__restrict__ part,
                                                       size_t pstride, const double *done, HaloFold H) {
  if (done && *done != 0.0) return;
  __shared__ double tr[NW][64];
  __shared__ double red[2 * NW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (FOLD && blockIdx.x == 0) {  // the exchanger: ghost rows of x, no slices, an empty dot partial
    halo_fold_exchange<64 * NW>(H);
    if (DOT && tid == 0) { part[0] = 0.0; if (DOT == 2) part[pstride] = 0.0; }
    return;
  }
  const double *xx = FOLD ? H.xg : x;  // (FOLD: the ghost rows change during the launch -- not through the restrict pointer)
  const JSteps<NW, FOLD> S(nslices, H.interior_rows, w);
  const int T = S.steps(), dt = S.dt;  // uniform over the workgroup (the barrier below)
  bool crossed = false;
  double d0 = 0.0, d1 = 0.0;
  auto ldesc = [&](int s, JDesc &D) {
    if (s < nslices) { D.base = base[s]; D.c = cnt16[s]; }
    else { D.base = 0; D.c = make_uint4(0, 0, 0, 0); }
  };
  auto lent = [&](int s, const JDesc &D, JEnt<KU> &E) {
    int off = D.base;
#define JH_LENT(J)                                                                                                       \
    if (J < KU) { /* unconditional: lanes past the count read the next diagonal's entries (arrays padded by 64), unused */ \
      E.col[J < KU ? J : 0] = stream_load<NT>(jcol + off + lane);                                                        \
      E.val[J < KU ? J : 0] = stream_load<NT>(jval + off + lane);                                                        \
      off += jcount<J>(D.c);                                                                                             \
    }
    JH_LENT(0) JH_LENT(1) JH_LENT(2) JH_LENT(3) JH_LENT(4) JH_LENT(5) JH_LENT(6) JH_LENT(7)
#undef JH_LENT
    E.prow = (int)perm[(size_t)min(s, nslices - 1) * 64 + lane];
  };
  int t = S.t0;
  JDesc dc, dn, dnn;
  JEnt<KU> ec, en;
  ldesc(S.slice(t), dc);
  ldesc(S.slice(t + dt), dn);
  lent(S.slice(t), dc, ec);
  for (; t < T; t += dt) {
    const int s = S.slice(t);
    ldesc(S.slice(t + 2 * dt), dnn);
    lent(S.slice(t + dt), dn, en);
    __syncthreads();  // keeps the wavefronts of the workgroup on neighbouring slices
    if (FOLD && !crossed && t >= S.ni) { halo_fold_wait(H); crossed = true; }  // the ghost rows of x are in place from here on
    double xg[KU];
    // (unconditional as well: an unused lane holds some other entry's column id, a valid index)
#define JH_GATH(J) if (J < KU) xg[J < KU ? J : 0] = xx[ec.col[J < KU ? J : 0]];
    JH_GATH(0) JH_GATH(1) JH_GATH(2) JH_GATH(3) JH_GATH(4) JH_GATH(5) JH_GATH(6) JH_GATH(7)
#undef JH_GATH
    double acc = 0.0;
#define JH_ACC(J) if (J < KU) { const double t = acc + ec.val[J < KU ? J : 0] * xg[J < KU ? J : 0]; acc = (lane < jcount<J>(dc.c)) ? t : acc; }
    JH_ACC(0) JH_ACC(1) JH_ACC(2) JH_ACC(3) JH_ACC(4) JH_ACC(5) JH_ACC(6) JH_ACC(7)
#undef JH_ACC
    // lane -> row: through the wavefront's LDS strip (the LDS operations of one wavefront complete in order), so that y is
    // stored -- and the dot weights are read -- as 512 contiguous bytes
    const int nr = (s < nslices) ? min(64, nrows - s * 64) : 0;
    if (lane < nr) tr[w][ec.prow] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < nr) {
      const int row = s * 64 + lane;
      const double a = tr[w][lane];
      const double yv = (beta == 0.0) ? alpha * a : alpha * a + beta * y[row];
      y[row] = yv;
      if (DOT && row < dot_rows) { d0 += yv * dw[row]; if (DOT == 2) d1 += yv * yv; }
    }
    __builtin_amdgcn_wave_barrier();
    dc = dn; dn = dnn;
    ec = en;
  }
  if (DOT) jds_dot_partial<NW>(d0, d1, DOT, red, part, pstride);  // one partial per workgroup, reduced in a fixed order
}

// The same product with 16-bit column codes (Pattern::Jagged::d_col16): 10 instead of 12 bytes per entry.  One more pipeline
// stage: the codes of slice i+2 are in flight, those of slice i+1 are decoded (window origin + code, or -- a few percent of
// the entries -- a load from the slice's far list) while slice i gathers x and accumulates.  Lanes past a diagonal's count hold
// another entry's code: clamped into [0, nrows) / padded far list, so every gather address is valid.
template <int KU>
struct JRaw {
  unsigned short c16[KU];
  double val[KU];
  int prow;
};
struct JDesc16 {
  int base, cb, fo;
  uint4 c;
};
template <int KU, int DOT, int NW, bool FOLD>
__global__ __launch_bounds__(64 * NW) void spmv_jds16_kernel(const int32_t *__restrict__ base, const uint4 *__restrict__ cnt16,
                                                         const uint8_t *__restrict__ perm, const uint16_t *__restrict__ jcol,
                                                         const int2 *__restrict__ win, const int32_t *__restrict__ far,
                                                         const double *__restrict__ jval, int nslices, int nrows,
                                                         const double *__restrict__ x, double *__restrict__ y, double alpha, double beta,
                                                         const double *__restrict__ dw, int dot_rows, double *__restrict__ part,
                                                         size_t pstride, const double *done, HaloFold H) {
  if (done && *done != 0.0) return;
  __shared__ double tr[NW][64];
  __shared__ double red[2 * NW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (FOLD && blockIdx.x == 0) {  // the exchanger (see spmv_jds_kernel)
    halo_fold_exchange<64 * NW>(H);
    if (DOT && tid == 0) { part[0] = 0.0; if (DOT == 2) part[pstride] = 0.0; }
    return;
  }
  const double *xx = FOLD ? H.xg : x;
  const JSteps<NW, FOLD> S(nslices, H.interior_rows, w);
  const int T = S.steps(), dt = S.dt;  // uniform over the workgroup (the barrier below)
  bool crossed = false;
  double d0 = 0.0, d1 = 0.0;
  auto ldesc = [&](int s, JDesc16 &D) {
    if (s < nslices) { D.base = base[s]; D.c = cnt16[s]; const int2 wf = win[s]; D.cb = wf.x; D.fo = wf.y; }
    else { D.base = 0; D.c = make_uint4(0, 0, 0, 0); D.cb = 0; D.fo = 0; }
  };
  auto lraw = [&](int s, const JDesc16 &D, JRaw<KU> &E) {
    int off = D.base;
#define JH_LRAW(J)                                                                                                       \
    if (J < KU) { /* unconditional: lanes past the count read the next diagonal's entries (arrays padded by 64), unused */ \
      E.c16[J < KU ? J : 0] = stream_load<true>(jcol + off + lane);  /* (16-bit codes: matrices of 3M rows and more) */  \
      E.val[J < KU ? J : 0] = stream_load<true>(jval + off + lane);                                                      \
      off += jcount<J>(D.c);                                                                                             \
    }
    JH_LRAW(0) JH_LRAW(1) JH_LRAW(2) JH_LRAW(3) JH_LRAW(4) JH_LRAW(5) JH_LRAW(6) JH_LRAW(7)
#undef JH_LRAW
    E.prow = (int)perm[(size_t)min(s, nslices - 1) * 64 + lane];
  };
  auto decode = [&](const JDesc16 &D, const JRaw<KU> &R, JEnt<KU> &E) {
#pragma unroll
    for (int j = 0; j < KU; ++j) {
      const int code = (int)R.c16[j];
      int c = min(max(D.cb + code, 0), nrows - 1);
      if (code >= JDS_FAR) c = far[D.fo + code - JDS_FAR];
      E.col[j] = c;
      E.val[j] = R.val[j];
    }
    E.prow = R.prow;
  };
  int t = S.t0;
  JDesc16 dc, dn, dnn, dnnn;
  JRaw<KU> rn, rnn;
  JEnt<KU> ec, en;
  ldesc(S.slice(t), dc);
  ldesc(S.slice(t + dt), dn);
  ldesc(S.slice(t + 2 * dt), dnn);
  lraw(S.slice(t), dc, rn);
  decode(dc, rn, ec);
  lraw(S.slice(t + dt), dn, rn);
  for (; t < T; t += dt) {
    const int s = S.slice(t);
    ldesc(S.slice(t + 3 * dt), dnnn);
    lraw(S.slice(t + 2 * dt), dnn, rnn);
    decode(dn, rn, en);  // (its far-list loads complete during this slice's gathers)
    __syncthreads();  // keeps the wavefronts of the workgroup on neighbouring slices
    if (FOLD && !crossed && t >= S.ni) { halo_fold_wait(H); crossed = true; }
    double xg[KU];
#define JH_GATH(J) if (J < KU) xg[J < KU ? J : 0] = xx[ec.col[J < KU ? J : 0]];
    JH_GATH(0) JH_GATH(1) JH_GATH(2) JH_GATH(3) JH_GATH(4) JH_GATH(5) JH_GATH(6) JH_GATH(7)
#undef JH_GATH
    double acc = 0.0;
#define JH_ACC(J) if (J < KU) { const double t = acc + ec.val[J < KU ? J : 0] * xg[J < KU ? J : 0]; acc = (lane < jcount<J>(dc.c)) ? t : acc; }
    JH_ACC(0) JH_ACC(1) JH_ACC(2) JH_ACC(3) JH_ACC(4) JH_ACC(5) JH_ACC(6) JH_ACC(7)
#undef JH_ACC
    const int nr = (s < nslices) ? min(64, nrows - s * 64) : 0;
    if (lane < nr) tr[w][ec.prow] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < nr) {
      const int row = s * 64 + lane;
      const double a = tr[w][lane];
      const double yv = (beta == 0.0) ? alpha * a : alpha * a + beta * y[row];
      y[row] = yv;
      if (DOT && row < dot_rows) { d0 += yv * dw[row]; if (DOT == 2) d1 += yv * yv; }
    }
    __builtin_amdgcn_wave_barrier();
    dc = dn; dn = dnn; dnn = dnnn;
    ec = en;
    rn = rnn;
  }
  if (DOT) jds_dot_partial<NW>(d0, d1, DOT, red, part, pstride);  // one partial per workgroup, reduced in a fixed order
}

}  // namespace

// Copies the current values of A into the jagged-slice order; true if the Krylov loop can multiply with k_spmv_sell.
bool sell_usable(jh_csr A) {
  if (!A->ctx->opt.spmv_jagged) return false;
  Pattern &P = *A->pat;
  if (!P.jag.built) P.build_jagged();
  return P.jag.usable;
}
bool sell_refresh(jh_csr A) {
  A->jval_fresh = false;
  if (!sell_usable(A)) return false;
  Pattern &P = *A->pat;
  if (A->jval.n < (size_t)P.jag.nent + 64) {
    A->jval.alloc((size_t)P.jag.nent + 64);
    JH_HIP(hipMemsetAsync(A->jval.p + P.jag.nent, 0, 64 * sizeof(double), A->ctx->stream));  // (the padding the last diagonal's lanes load)
  }
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>((P.jag.nslices + 3) / 4, 8192));
  hipLaunchKernelGGL(jagged_copy_kernel, dim3(g), dim3(256), 0, A->ctx->stream, A->jval.p, A->val.p, P.jag.d_src.p, P.jag.d_base.p,
                     P.d_rowptr.p, P.jag.nslices, (int)P.n);
  A->jval_fresh = true;
  return true;
}

// Resident workgroups per CU of a kernel (occupancy query, cached per kernel): the persistent grids are sized by it, so that no
// workgroup of a launch has to wait for another one to finish (with 68-70 VGPRs the 16-bit kernels hold 7, not 8, wavefronts per SIMD).
template <class KernelT>
static int resident_per_cu(KernelT kernel, int threads) {
  static std::mutex m;
  static std::unordered_map<const void *, int> cache;
  std::lock_guard<std::mutex> lk(m);
  auto it = cache.find((const void *)kernel);
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
  cache[(const void *)kernel] = nb;
  return nb;
}

// reduce_now = false: the caller launches the second stage of the fused dot itself (spmv_dot_reduce with the returned count) or
// hands the partials to the consuming kernel (PendSum).  waves: wavefronts per workgroup, 4 / 8 / 16 (see spmv_jds_kernel).
// fold: push-halo hand-shake inside the launch (HaloFold); the caller checked sell_can_fold.
int k_spmv_sell(jh_csr A, const double *x, double *y, double alpha, double beta, const SpmvDot *dot, const double *done, bool reduce_now,
                int waves, const HaloFold *fold) {
  jh_context ctx = A->ctx;
  const Pattern &P = *A->pat;
  const auto &J = P.jag;
  if (!A->jval_fresh) JH_THROW("jagged-slice SpMV without fresh values (sell_refresh)");
  if (waves != 4 && waves != 8 && waves != 16) JH_THROW("jagged-slice SpMV: 4, 8 or 16 wavefronts per workgroup");
  const int mode = dot ? dot->mode : 0;
  const int nsup = (J.nslices + waves - 1) / waves;
  const bool folded = fold && fold->self;
  // steps of the busiest XCD (JSteps)
  int chunk = (nsup + NUM_XCD - 1) / NUM_XCD;
  if (folded) {
    const int sbs = std::min<int>(nsup, std::max(0, fold->interior_rows) / (64 * waves));
    chunk = (sbs + NUM_XCD - 1) / NUM_XCD + (nsup - sbs + NUM_XCD - 1) / NUM_XCD;
  }
  const double *dw = dot ? dot->w : nullptr;
  const int drows = dot ? (int)dot->n_rows : 0;
  const HaloFold H = folded ? *fold : HaloFold();
  dim3 block(64 * waves);
  int nparts = 0;
  auto launch = [&](auto kernel, auto... args) {
    // persistent grid: what is resident at once (32 CUs per XCD, fewer under a CU mask), or option spmv_waves_per_xcd
    const int cap = ctx->opt.spmv_waves_per_xcd > 0 ? std::max<int>(1, (int)ctx->opt.spmv_waves_per_xcd / waves)
                                                    : ctx->cus_per_xcd() * resident_per_cu(kernel, 64 * waves);
    // (folded: one more workgroup per XCD -- XCD 0's first one is the exchanger -- and never fewer than two)
    dim3 grid((unsigned)(std::max(folded ? 2 : 1, std::min(chunk + (folded ? 1 : 0), cap)) * NUM_XCD));
    if (mode) ensure_partials(ctx, (size_t)grid.x);
    hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, args..., A->jval.p, J.nslices, (int)P.n, x, y, alpha, beta, dw, drows,
                       ctx->partials.p, ctx->partial_stride, done, H);
    nparts = (int)grid.x;
  };
#define JH_JDS(KU, DV, NWV, FO)                                                                                                             \
  do {                                                                                                                                 \
    if (J.nontemporal) launch(spmv_jds_kernel<KU, DV, NWV, true, FO>, J.d_base.p, reinterpret_cast<const uint4 *>(J.d_cnt.p), J.d_perm.p, J.d_col.p); \
    else launch(spmv_jds_kernel<KU, DV, NWV, false, FO>, J.d_base.p, reinterpret_cast<const uint4 *>(J.d_cnt.p), J.d_perm.p, J.d_col.p);              \
  } while (0)
#define JH_JDS16(KU, DV, NWV, FO)                                                                                               \
  launch(spmv_jds16_kernel<KU, DV, NWV, FO>, J.d_base.p, reinterpret_cast<const uint4 *>(J.d_cnt.p), J.d_perm.p, J.d_col16.p, \
         reinterpret_cast<const int2 *>(J.d_win.p), J.d_far.p)
#define JH_JDS_M(K, KU, NWV, FO) do { if (mode == 0) K(KU, 0, NWV, FO); else if (mode == 1) K(KU, 1, NWV, FO); else K(KU, 2, NWV, FO); } while (0)
#define JH_JDS_W(K, KU) do { if (folded) JH_JDS_M(K, KU, 16, true); else if (waves == 4) JH_JDS_M(K, KU, 4, false); else if (waves == 8) JH_JDS_M(K, KU, 8, false); else JH_JDS_M(K, KU, 16, false); } while (0)
  if (folded && waves != 16) JH_THROW("folded push halo: 16 wavefronts per workgroup");
  if (J.d_col.n == 0) {  // 16-bit column codes (default)
    if (J.kmax <= 5) JH_JDS_W(JH_JDS16, 5); else JH_JDS_W(JH_JDS16, 8);
  } else {
    if (J.kmax <= 5) JH_JDS_W(JH_JDS, 5); else JH_JDS_W(JH_JDS, 8);
  }
#undef JH_JDS_W
#undef JH_JDS_M
#undef JH_JDS
#undef JH_JDS16
  if (mode && reduce_now) spmv_dot_reduce(ctx, dot, nparts, done);
  return nparts;
}

}  // namespace jh
