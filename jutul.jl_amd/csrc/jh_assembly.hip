// Fused TPFA flux + residual/Jacobian assembly (a-5 + a-6) for gfx950.
//
// Reference: update_half_face_flux_tpfa! materialises Dual arrays acc[ne,nc], hf[ne,nhf] (conservation.jl:
// 558-626) and fill_conservation_eq! re-reads them and scatters through 8-byte position tables
// (conservation.jl:373-430).  Here the half-face data is permuted ONCE at setup into the Jacobian's CSR
// order ("sorted half-face permutation": per device nnz a transmissibility and a signed face id), so that
// the flux of cell `row` towards `col[k]` and ALL its derivatives are produced while streaming row `row`:
//     r[row]            = acc + sum_k q(row -> col_k)
//     J[row,row]        = d acc/d x_row + sum_k d q_k / d x_row
//     J[row,col_k]      = d q_k / d x_col_k   ( == -d q(col_k -> row)/d x_col_k, what the reference stores
//                                               from the neighbour's half-face; two-point fluxes are
//                                               antisymmetric, conservation.jl:397-415)
// Every nzval slot is written exactly once, coalesced, with no position table and no intermediate Dual
// arrays.  A 256-thread workgroup owns a tile of consecutive rows (<= 1024 entries): lanes stream
// (col, T) pairs, gather the neighbour's primary variables, stage flux values/derivatives in LDS and one
// lane per row reduces its row segment.  Bound: HBM; algorithmic bytes 44*nc + 20*nhf for N = 1.
#include "jh_internal.hpp"

namespace jh {

__device__ __forceinline__ int xcd_tile_a(int b, int ntiles) {
  int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
  return (b % NUM_XCD) * chunk + b / NUM_XCD;
}

// ---- forward-mode duals over the 2N primary variables of (self, other) ------------------------------------------
// Sparse: M is the compile-time set of partials that can be non-zero; the others are never computed.  A dense Dual<4> product
// costs 1 + 8 multiplies and 4 adds whatever its operands depend on (0 * x cannot be folded under IEEE rules); in the
// two-phase flux a density depends on one of the four variables, a mobility on two, and 45 % of the kernel's VALU instructions
// were products with structural zeros.  For finite values the results are bit-identical to the dense duals (x + 0*y == x).
template <unsigned M>
struct Dual {
  double v;
  double d[4];
};
#define JH_HAS(M, i) ((((M) >> (i)) & 1u) != 0u)
__device__ __forceinline__ Dual<0u> dconst(double v) {
  Dual<0u> r; r.v = v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = 0.0;
  return r;
}
template <int IDX> __device__ __forceinline__ Dual<(1u << IDX)> dvar(double v) {
  Dual<(1u << IDX)> r; r.v = v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = (i == IDX) ? 1.0 : 0.0;
  return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ Dual<(A | B)> operator+(Dual<A> a, Dual<B> b) {
  Dual<(A | B)> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r.d[i] = (JH_HAS(A, i) && JH_HAS(B, i)) ? a.d[i] + b.d[i] : JH_HAS(A, i) ? a.d[i] : JH_HAS(B, i) ? b.d[i] : 0.0;
  return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ Dual<(A | B)> operator-(Dual<A> a, Dual<B> b) {
  Dual<(A | B)> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r.d[i] = (JH_HAS(A, i) && JH_HAS(B, i)) ? a.d[i] - b.d[i] : JH_HAS(A, i) ? a.d[i] : JH_HAS(B, i) ? -b.d[i] : 0.0;
  return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ Dual<(A | B)> operator*(Dual<A> a, Dual<B> b) {
  Dual<(A | B)> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r.d[i] = (JH_HAS(A, i) && JH_HAS(B, i)) ? a.d[i] * b.v + a.v * b.d[i]
             : JH_HAS(A, i)                 ? a.d[i] * b.v
             : JH_HAS(B, i)                 ? a.v * b.d[i]
                                            : 0.0;
  return r;
}
template <unsigned A> __device__ __forceinline__ Dual<A> operator*(double s, Dual<A> a) {
  Dual<A> r; r.v = s * a.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = JH_HAS(A, i) ? s * a.d[i] : 0.0;
  return r;
}
template <unsigned A> __device__ __forceinline__ Dual<A> ddiv(Dual<A> a, double s) {
  Dual<A> r; r.v = a.v / s;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = JH_HAS(A, i) ? a.d[i] / s : 0.0;
  return r;
}
template <unsigned A> __device__ __forceinline__ Dual<A> dexp(Dual<A> a) {
  double e = exp(a.v);
  Dual<A> r; r.v = e;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = JH_HAS(A, i) ? e * a.d[i] : 0.0;
  return r;
}
// c ? a : b (the upwind choice between two mobilities that depend on different cells)
template <unsigned A, unsigned B> __device__ __forceinline__ Dual<(A | B)> dselect(bool c, Dual<A> a, Dual<B> b) {
  Dual<(A | B)> r; r.v = c ? a.v : b.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = c ? (JH_HAS(A, i) ? a.d[i] : 0.0) : (JH_HAS(B, i) ? b.d[i] : 0.0);
  return r;
}

struct LawPar {
  double rho0[2], comp[2], mu[2], p_ref;
  double inv_mu[2];  // 1.0 / mu[ph], formed once on the host (the same IEEE quotient the per-lane division produced)
};
// argument of the density's exponential; ONE expression for every place that evaluates it, so that an exponential computed
// once and handed over through LDS has the bits of one computed in place
__device__ __forceinline__ double density_arg(const LawPar &P, int ph, double p) { return P.comp[ph] * (p - P.p_ref); }
template <unsigned A> __device__ __forceinline__ Dual<A> density(const LawPar &P, int ph, Dual<A> p) {
  return P.rho0[ph] * dexp(P.comp[ph] * (p - dconst(P.p_ref)));  // rho0*exp(c*(p-p0))
}
// the same density from E = exp(density_arg(p)) evaluated elsewhere (once per cell of the tile instead of once per entry):
// identical operations on identical inputs, hence the same bits as density()
template <int VAR> __device__ __forceinline__ Dual<(1u << VAR)> density_from_exp(const LawPar &P, int ph, double E) {
  Dual<(1u << VAR)> r = dvar<VAR>(E);
  r.d[VAR] = E * (P.comp[ph] * 1.0);
  return P.rho0[ph] * r;
}

// flux of the scalar laws across one half-face with its derivatives w.r.t. the self / other primary variable (shared by the
// tile kernel and its pipelined variant)
template <int KIND>
__device__ __forceinline__ void flux_scalar(double Us, double Uo, double T, double gz, const LawPar &par, double &q, double &dself,
                                            double &dother) {
  if (KIND == JH_LAW_POISSON) {  // q = -K[face]*(U_other - U_self)  (variable_poisson.jl:99,124)
    q = -(T * (Uo - Us));
    dself = T;
    dother = -T;
  } else {
    auto ps = dvar<0>(Us);
    auto po = dvar<1>(Uo);
    auto rs = density(par, 0, ps);
    auto ro = density(par, 0, po);
    auto ravg = 0.5 * (rs + ro);             // face_average (flux.jl:372-375)
    auto dphi = (ps - po) + gz * ravg;       // two_point_potential_drop (flux.jl:335-338)
    auto f = (T / par.mu[0]) * (ravg * dphi);
    q = f.v;
    dself = f.d[0];
    dother = f.d[1];
  }
}
// flux of the two-phase law across one half-face: values of both equations, derivatives w.r.t. the self cell's (p, S) in ds
// (column-major (e, d)) and w.r.t. the other cell's in dother (shared by the tile kernel and the pipelined kernel)
__device__ __forceinline__ void flux_twophase(double ps_, double ss_, double po_, double so_, const double Eself[2],
                                              const double Eoth[2], double T, double gz, const LawPar &par, double q2[2],
                                              double ds[4], double dother[4]) {
  auto ps = dvar<0>(ps_);
  auto ss_w = dvar<1>(ss_);
  auto po = dvar<2>(po_);
  auto so_w = dvar<3>(so_);
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    auto rs = density_from_exp<0>(par, ph, Eself[ph]);
    auto ro = density_from_exp<2>(par, ph, Eoth[ph]);
    auto ravg = 0.5 * (rs + ro);
    auto dphi = (ps - po) + gz * ravg;
    auto ss = ph == 0 ? ss_w : dconst(1.0) - ss_w;
    auto so = ph == 0 ? so_w : dconst(1.0) - so_w;
    auto ms = par.inv_mu[ph] * ((ss * ss) * rs);
    auto mo = par.inv_mu[ph] * ((so * so) * ro);
    auto up = dselect(dphi.v < 0.0, mo, ms);  // SPU upwind (flux.jl:382-405)
    auto q = T * (up * dphi);
    q2[ph] = q.v;
    ds[0 * 2 + ph] = q.d[0];  // (e = ph, d = 0), column-major
    ds[1 * 2 + ph] = q.d[1];
    dother[0 * 2 + ph] = q.d[2];
    dother[1 * 2 + ph] = q.d[3];
  }
}
// accumulation term of the two-phase law for one row: (M - M0)/dt with its derivatives w.r.t. the row's (p, S)
__device__ __forceinline__ void accum_twophase(double p, double sw_, double p0, double sw0, const double E[2], double vol,
                                               double dt, const LawPar &par, double ar[2], double ap[4]) {
  auto sw = dvar<1>(sw_);
  auto so = dconst(1.0) - sw;
  auto Mw = vol * (density_from_exp<0>(par, 0, E[0]) * sw);
  auto Mo = vol * (density_from_exp<0>(par, 1, E[1]) * so);
  double Mw0 = vol * (density(par, 0, dconst(p0)).v * sw0);
  double Mo0 = vol * (density(par, 1, dconst(p0)).v * (1.0 - sw0));
  auto aw = ddiv(Mw - dconst(Mw0), dt);
  auto ao = ddiv(Mo - dconst(Mo0), dt);
  ar[0] = aw.v; ar[1] = ao.v;
  ap[0] = aw.d[0]; ap[1] = ao.d[0]; ap[2] = aw.d[1]; ap[3] = ao.d[1];  // column-major (e, d)
}

// Store of a tile's 2x2 blocks.  A lane that stores its own block issues four 8-byte stores 32 bytes apart from its neighbours':
// every store instruction touches a quarter of each line.  The off-diagonal blocks go to their (free by now) dsv slots next to
// the parked diagonal blocks instead, and the tile's nzval range leaves the CU as one linear copy, 16 bytes per lane.
template <int KPT>
__device__ __forceinline__ void store_blocks2(double *__restrict__ nz, double *dsv, const double (*off)[4], const bool *isdiag,
                                              int base, int cnt, int tid) {
#pragma unroll
  for (int kk = 0; kk < KPT; ++kk) {
    const int k = tid + kk * TILE_THREADS;
    if (k < cnt && !isdiag[kk]) {
      reinterpret_cast<double2 *>(dsv)[k * 2] = make_double2(off[kk][0], off[kk][1]);
      reinterpret_cast<double2 *>(dsv)[k * 2 + 1] = make_double2(off[kk][2], off[kk][3]);
    }
  }
  __syncthreads();
  double2 *dst = reinterpret_cast<double2 *>(nz + (size_t)base * 4);
  for (int i = tid; i < cnt * 2; i += TILE_THREADS) dst[i] = reinterpret_cast<const double2 *>(dsv)[i];
}
// ---- the kernel -------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(TILE_THREADS) void assemble_tile_kernel(
    const int32_t *__restrict__ tile_row, int ntiles, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ diag, const double *__restrict__ Tnz, const double *__restrict__ gnz, const double *__restrict__ X,
    const double *__restrict__ X0, double *__restrict__ nz, double *__restrict__ r, double dt, LawPar par, int reg_row,
    int nrows_total) {
  constexpr int N = (KIND == JH_LAW_TWOPHASE) ? 2 : 1;
  constexpr int NN = N * N;
  constexpr int TNNZ = (N == 1) ? TILE_NNZ : TILE_NNZ / 2;  // tile entry budget (Pattern::build_tiles)
  __shared__ double qv[TNNZ * N];    // flux values per entry
  __shared__ __align__(16) double dsv[TNNZ * NN];  // d q / d x_self per entry (column-major N x N)
  // Rows staged on either side of the tile's own rows.  0: a +-128-row window (82% instead of 66% of the neighbours in
  // LDS) did not shorten the launch (0.331 vs 0.335 ms) and raised the measured HBM traffic from 1.48 to 1.65 GB.
  constexpr int WIN = 0;
  __shared__ double xs[(TILE_ROWS + 2 * WIN) * N];  // primary variables of rows [w0, w0 + wn)
  __shared__ int32_t rp[TILE_ROWS + 1];
  __shared__ uint8_t rowof[TNNZ];
  const int t = xcd_tile_a(blockIdx.x, ntiles);
  if (t >= ntiles) return;
  const int4 td = reinterpret_cast<const int4 *>(tile_row)[t];  // tile descriptor (Pattern::tile_desc): one load, not a chain
  const int r0 = td.x, nrows = td.y, base = td.z;
  const int cnt = td.w;  // TPFA rows are short: cnt <= TILE_NNZ is guaranteed by the host (checked)
  const int tid = threadIdx.x;
  constexpr int KPT = TNNZ / TILE_THREADS;  // entries per lane
  // the (col, T) stream of the lane's entries is requested first: it is in flight while the row table is built
  int cidx[KPT];
  double Tk[KPT];
#pragma unroll
  for (int kk = 0; kk < KPT; ++kk) {
    const int k = tid + kk * TILE_THREADS;
    cidx[kk] = (k < cnt) ? col[base + k] : 0;
    Tk[kk] = (k < cnt) ? Tnz[base + k] : 0.0;
  }
  const int w0 = max(0, r0 - WIN);
  const int wn = min(nrows_total, r0 + TILE_ROWS + WIN) - w0;
  const int own = r0 - w0;  // offset of the tile's first row inside the window
  for (int i = tid; i <= nrows; i += TILE_THREADS) rp[i] = rowptr[r0 + i] - base;
  for (int i = tid; i < wn * N; i += TILE_THREADS) xs[i] = X[(size_t)w0 * N + i];
  __syncthreads();
  if (tid < nrows)
    for (int j = rp[tid]; j < rp[tid + 1]; ++j) rowof[j] = (uint8_t)tid;
  __syncthreads();
  // ---- phase 1: one lane per CSR entry -------------------------------------------------------------------
  // The off-diagonal blocks stay in registers until phase 3 so that every nzval line of the tile is written once,
  // full and coalesced (8-byte diagonal stores from the row lanes cost a 32-byte HBM write each: +0.32 GB at 10M
  // cells, measured with WRITE_SIZE).
  double off[KPT][NN];
  bool isdiag[KPT];
#pragma unroll
  for (int kk = 0; kk < KPT; ++kk) {
    const int k = tid + kk * TILE_THREADS;
    isdiag[kk] = false;
    if (k >= cnt) continue;
    const int lr = rowof[k];
    const int c = cidx[kk];
    const double T = Tk[kk];
    const unsigned cw = (unsigned)(c - w0);  // position of the neighbour inside the LDS window (if < wn)
    const bool inl = cw < (unsigned)wn;
    if (c == r0 + lr) {  // diagonal slot: no flux; its block is produced by the row lane in phase 2
#pragma unroll
      for (int e = 0; e < N; ++e) qv[k * N + e] = 0.0;
#pragma unroll
      for (int i = 0; i < NN; ++i) dsv[k * NN + i] = 0.0;
      isdiag[kk] = true;
      continue;
    }
    if constexpr (N == 1) {
      const double gz = gnz ? gnz[base + k] : 0.0;
      double q, ds, dother_;
      flux_scalar<KIND>(xs[own + lr], lds_or_global(xs, cw, inl, X, (size_t)c), T, gz, par, q, ds, dother_);
      qv[k] = q;
      dsv[k] = ds;
      off[kk][0] = dother_;
    } else {
      const double gz = gnz ? gnz[base + k] : 0.0;
      const double ps = xs[(own + lr) * 2], ss = xs[(own + lr) * 2 + 1];
      const double po = lds_or_global(xs, cw * 2, inl, X, (size_t)c * 2);
      const double so = lds_or_global(xs, cw * 2 + 1, inl, X, (size_t)c * 2 + 1);
      double Eself[2], Eoth[2], q2[2], ds[4];
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        Eself[ph] = exp(density_arg(par, ph, ps));
        Eoth[ph] = exp(density_arg(par, ph, po));
      }
      flux_twophase(ps, ss, po, so, Eself, Eoth, T, gz, par, q2, ds, off[kk]);
#pragma unroll
      for (int e = 0; e < 2; ++e) qv[k * 2 + e] = q2[e];
#pragma unroll
      for (int i = 0; i < 4; ++i) dsv[k * 4 + i] = ds[i];
    }
  }
  __syncthreads();
  // ---- phase 2: one lane per row ----------------------------------------------------------------------------------
  if (tid < nrows) {
    const int row = r0 + tid;
    double ar[N], ap[NN];
    int dk = -1;
    // accumulation term (update_accumulation!, conservation.jl:558-568): (M - M0)/dt as a Dual wrt own cell
    {
      // diagonal slot of Tnz carries the accumulation coefficient vol*phi of this row
      dk = diag[row] - base;
      const double vol = Tnz[base + dk];
      if (KIND == JH_LAW_POISSON) {
        const double U = xs[own + tid];
        if (dt > 0.0) {
          ar[0] = (vol * U - vol * X0[row]) / dt;
          ap[0] = vol / dt;
        } else {  // stationary variant with the 1e-10*U regulariser on host cell 1 (variable_poisson.jl:101-104)
          ar[0] = (row == reg_row) ? 1e-10 * U : 0.0;
          ap[0] = (row == reg_row) ? 1e-10 : 0.0;
        }
      } else if (KIND == JH_LAW_COMPRESSIBLE) {
        auto p = dvar<0>(xs[own + tid]);
        auto M = vol * density(par, 0, p);
        double M0 = vol * density(par, 0, dconst(X0[row])).v;
        auto a = ddiv(M - dconst(M0), dt);
        ar[0] = a.v;
        ap[0] = a.d[0];
      } else if constexpr (N == 2) {
        const double p = xs[(own + tid) * 2], E[2] = {exp(density_arg(par, 0, p)), exp(density_arg(par, 1, p))};
        accum_twophase(p, xs[(own + tid) * 2 + 1], X0[(size_t)row * 2], X0[(size_t)row * 2 + 1], E, vol, dt, par, ar, ap);
      }
    }
    for (int j = rp[tid]; j < rp[tid + 1]; ++j) {
#pragma unroll
      for (int e = 0; e < N; ++e) ar[e] = ar[e] + qv[j * N + e];
#pragma unroll
      for (int i = 0; i < NN; ++i) ap[i] = ap[i] + dsv[j * NN + i];
    }
#pragma unroll
    for (int e = 0; e < N; ++e) r[(size_t)row * N + e] = ar[e];
    if (dk >= 0) {  // park the diagonal block in its (otherwise unused) LDS slot for the coalesced store of phase 3
#pragma unroll
      for (int i = 0; i < NN; ++i) dsv[dk * NN + i] = ap[i];
    }
  }
  __syncthreads();
  // ---- phase 3: every lane stores its entries; nzval lines leave the CU complete ---------------------------------------
  if constexpr (N == 2) {
    store_blocks2<KPT>(nz, dsv, off, isdiag, base, cnt, tid);
  } else {
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
      const int k = tid + kk * TILE_THREADS;
      if (k >= cnt) continue;
      double *blk = nz + (size_t)(base + k) * NN;
#pragma unroll
      for (int i = 0; i < NN; ++i) blk[i] = isdiag[kk] ? dsv[k * NN + i] : off[kk][i];
    }
  }
}


// ---- software-pipelined, persistent variant for the scalar laws -------------------------------------------------------------
// Same arithmetic and tile structure as assemble_tile_kernel; a tile is a chain of dependent latencies (descriptor ->
// (col, T) / row data -> neighbour gather), so a workgroup walks its tiles (XCD-contiguous chunks like the SpMV) and issues
// the first-level loads of its next tile before it computes the current one.  The diagonal slot's Tnz value IS the row's
// accumulation coefficient and its position IS diag[row]: the entry lane that owns the slot hands both to the row lane
// through LDS, so the row phase needs no dependent global load (and the diag array is not read at all).
template <int N, int KPT>
struct AsmPre {
  int r0, nrows, base, cnt;
  int cidx[KPT];
  double Tk[KPT], gz[KPT];
  double xrow[N], x0[N];
  int rpv;
};
template <int KIND>
__global__ __launch_bounds__(TILE_THREADS) void assemble_pipe_kernel(
    const int32_t *__restrict__ tile_desc, int ntiles, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const double *__restrict__ Tnz, const double *__restrict__ gnz, const double *__restrict__ X, const double *__restrict__ X0,
    double *__restrict__ nz, double *__restrict__ r, double dt, LawPar par, int reg_row, int nrows_total) {
  constexpr int N = (KIND == JH_LAW_TWOPHASE) ? 2 : 1;
  constexpr int NN = N * N;
  constexpr int TNNZ = (N == 1) ? TILE_NNZ : TILE_NNZ / 2;  // tile entry budget (Pattern::build_tiles)
  constexpr int KPT = TNNZ / TILE_THREADS;
  __shared__ double qv[TNNZ * N];
  __shared__ __align__(16) double dsv[TNNZ * NN];
  // primary variables of the TILE_ROWS rows from the tile's first row on: the tile's own rows (<= TILE_ROWS; ~100 for 2x2
  // blocks) and the rows after them, which hold most of the remaining neighbours (the device blocks are contiguous)
  __shared__ double xs[TILE_ROWS * N];
  __shared__ double volrow[TILE_ROWS];
  __shared__ uint16_t rp[TILE_ROWS + 2];  // entry offsets inside the tile (<= TILE_NNZ)
  __shared__ uint16_t diagk[TILE_ROWS];
  __shared__ uint8_t rowof[TNNZ];
  const int tid = threadIdx.x;
  const int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
  const int xcd = blockIdx.x % NUM_XCD, wg = blockIdx.x / NUM_XCD, wgs = gridDim.x / NUM_XCD;
  auto in_range = [&](int tl) { return tl < chunk && xcd * chunk + tl < ntiles; };
  auto issue = [&](int tl, AsmPre<N, KPT> &P) {
    const int4 td = reinterpret_cast<const int4 *>(tile_desc)[xcd * chunk + tl];
    P.r0 = td.x; P.nrows = td.y; P.base = td.z; P.cnt = td.w;
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
      const int k = tid + kk * TILE_THREADS;
      P.cidx[kk] = (k < P.cnt) ? col[P.base + k] : 0;
      P.Tk[kk] = (k < P.cnt) ? Tnz[P.base + k] : 0.0;
      P.gz[kk] = (gnz && k < P.cnt) ? gnz[P.base + k] : 0.0;
    }
    const bool rowlane = tid < P.nrows;
    const bool winlane = P.r0 + tid < nrows_total;  // TILE_THREADS == TILE_ROWS: one window row per lane
#pragma unroll
    for (int e = 0; e < N; ++e) {
      P.xrow[e] = winlane ? X[(size_t)(P.r0 + tid) * N + e] : 0.0;
      P.x0[e] = rowlane ? X0[(size_t)(P.r0 + tid) * N + e] : 0.0;
    }
    P.rpv = rowlane ? rowptr[P.r0 + tid] - P.base : P.cnt;
  };
  AsmPre<N, KPT> cur, nxt;
  int tl = wg;
  bool have = in_range(tl);
  if (have) issue(tl, cur);
  while (have) {
    const int tln = tl + wgs;
    const bool have_next = in_range(tln);
    if (have_next) issue(tln, nxt);
    const int r0 = cur.r0, nrows = cur.nrows, base = cur.base, cnt = cur.cnt;
    const int wn = min(TILE_ROWS, nrows_total - r0);  // rows in the LDS window
    rp[tid] = (uint16_t)cur.rpv;
    if (tid == 0) rp[nrows] = (uint16_t)cnt;
#pragma unroll
    for (int e = 0; e < N; ++e) xs[tid * N + e] = cur.xrow[e];
    __syncthreads();
    if (tid < nrows)
      for (int j = rp[tid]; j < rp[tid + 1]; ++j) rowof[j] = (uint8_t)tid;
    __syncthreads();
    // phase 1: one lane per CSR entry; off-diagonals stay in registers until the coalesced store of phase 3
    double off[KPT][NN];
    bool isdiag[KPT];
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
      const int k = tid + kk * TILE_THREADS;
      isdiag[kk] = false;
#pragma unroll
      for (int i = 0; i < NN; ++i) off[kk][i] = 0.0;
      if (k >= cnt) continue;
      const int lr = rowof[k];
      const int c = cur.cidx[kk];
      if (c == r0 + lr) {  // diagonal slot: no flux; hand the accumulation coefficient and the slot to the row lane
#pragma unroll
        for (int e = 0; e < N; ++e) qv[k * N + e] = 0.0;
#pragma unroll
        for (int i = 0; i < NN; ++i) dsv[k * NN + i] = 0.0;
        volrow[lr] = cur.Tk[kk];
        diagk[lr] = (uint16_t)k;
        isdiag[kk] = true;
        continue;
      }
      const unsigned cw = (unsigned)(c - r0);
      const bool inl = cw < (unsigned)wn;
      if constexpr (N == 1) {
        const double Uo = lds_or_global(xs, cw, inl, X, (size_t)c);
        double q, ds, dother_;
        flux_scalar<KIND>(xs[lr], Uo, cur.Tk[kk], cur.gz[kk], par, q, ds, dother_);
        qv[k] = q;
        dsv[k] = ds;
        off[kk][0] = dother_;
      } else {
        const double ps = xs[lr * 2], ss = xs[lr * 2 + 1];
        const double po = lds_or_global(xs, cw * 2, inl, X, (size_t)c * 2);
        const double so = lds_or_global(xs, cw * 2 + 1, inl, X, (size_t)c * 2 + 1);
        double Eself[2], Eoth[2], q2[2], ds[4];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          Eself[ph] = exp(density_arg(par, ph, ps));
          Eoth[ph] = exp(density_arg(par, ph, po));
        }
        flux_twophase(ps, ss, po, so, Eself, Eoth, cur.Tk[kk], cur.gz[kk], par, q2, ds, off[kk]);
#pragma unroll
        for (int e = 0; e < N; ++e) qv[k * N + e] = q2[e];
#pragma unroll
        for (int i = 0; i < NN; ++i) dsv[k * NN + i] = ds[i];
      }
    }
    __syncthreads();
    // phase 2: one lane per row
    if (tid < nrows) {
      const int row = r0 + tid;
      const double vol = volrow[tid];
      double ar[N], ap[NN];
      if (KIND == JH_LAW_POISSON) {
        const double U = xs[tid];
        if (dt > 0.0) {
          ar[0] = (vol * U - vol * cur.x0[0]) / dt;
          ap[0] = vol / dt;
        } else {  // stationary variant with the 1e-10*U regulariser on host cell 1 (variable_poisson.jl:101-104)
          ar[0] = (row == reg_row) ? 1e-10 * U : 0.0;
          ap[0] = (row == reg_row) ? 1e-10 : 0.0;
        }
      } else if (KIND == JH_LAW_COMPRESSIBLE) {
        auto p = dvar<0>(xs[tid]);
        auto M = vol * density(par, 0, p);
        const double M0 = vol * density(par, 0, dconst(cur.x0[0])).v;
        auto a = ddiv(M - dconst(M0), dt);
        ar[0] = a.v;
        ap[0] = a.d[0];
      } else if constexpr (N == 2) {
        const double p = xs[tid * 2], E[2] = {exp(density_arg(par, 0, p)), exp(density_arg(par, 1, p))};
        accum_twophase(p, xs[tid * 2 + 1], cur.x0[0], cur.x0[1], E, vol, dt, par, ar, ap);
      }
      for (int j = rp[tid]; j < rp[tid + 1]; ++j) {
#pragma unroll
        for (int e = 0; e < N; ++e) ar[e] = ar[e] + qv[j * N + e];
#pragma unroll
        for (int i = 0; i < NN; ++i) ap[i] = ap[i] + dsv[j * NN + i];
      }
#pragma unroll
      for (int e = 0; e < N; ++e) r[(size_t)row * N + e] = ar[e];
      const int dk = diagk[tid];  // park the diagonal block in its LDS slot for the coalesced store
#pragma unroll
      for (int i = 0; i < NN; ++i) dsv[dk * NN + i] = ap[i];
    }
    __syncthreads();
    // phase 3: every lane stores its entries; nzval lines leave the CU complete
    if constexpr (N == 2) {
      store_blocks2<KPT>(nz, dsv, off, isdiag, base, cnt, tid);
    } else {
#pragma unroll
      for (int kk = 0; kk < KPT; ++kk) {
        const int k = tid + kk * TILE_THREADS;
        if (k < cnt) nz[base + k] = isdiag[kk] ? dsv[k] : off[kk][0];
      }
    }
    __syncthreads();  // LDS is reused by the next tile
    cur = nxt;
    tl = tln;
    have = have_next;
  }
}

__global__ void sources_kernel(double *r, const int32_t *cell, const double *val, int64_t n, int N) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n * N) return;
  int64_t s = i / N;
  int e = (int)(i - s * N);
  // one source entry per (cell, e); duplicates of the same cell must be pre-summed by the host wrapper
  r[(size_t)cell[s] * N + e] += val[i];
}

__global__ void gather_face_kernel(double *nzdata, const int32_t *nz_face, const double *face_data, int64_t nnzb, bool sgn) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < nnzb; k += (int64_t)gridDim.x * blockDim.x) {
    int f = nz_face[k];
    if (f == 0) continue;  // diagonal slot holds cell data
    double v = face_data[(f > 0 ? f : -f) - 1];
    nzdata[k] = (sgn && f < 0) ? -v : v;
  }
}
__global__ void set_diag_kernel(double *nzdata, const int32_t *diag, const int32_t *perm, const double *cell_data, int64_t n,
                                bool use_const, double cval) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    nzdata[diag[i]] = use_const ? cval : cell_data[perm ? perm[i] : i];
}

void k_gather_face_data(hipStream_t s, double *nzdata, const int32_t *nz_face, const double *face_data, int64_t nnzb, bool sgn) {
  int g = (int)std::min<int64_t>((nnzb + 255) / 256, 4096);
  if (nnzb) hipLaunchKernelGGL(gather_face_kernel, dim3(g), dim3(256), 0, s, nzdata, nz_face, face_data, nnzb, sgn);
}
void k_set_diag_data(hipStream_t s, double *nzdata, const int32_t *diag, const int32_t *perm, const double *cell_data, int64_t n,
                     bool use_const, double cval) {
  int g = (int)std::min<int64_t>((n + 255) / 256, 4096);
  if (n) hipLaunchKernelGGL(set_diag_kernel, dim3(g), dim3(256), 0, s, nzdata, diag, perm, cell_data, n, use_const, cval);
}

void custom_launch(jh_law L, double dt, jh_csr A, jh_vec r);  // jh_custom.cpp

void k_assemble(jh_law L, double dt, jh_csr A, jh_vec r) {
  const Pattern &P = *A->pat;
  jh_context ctx = L->ctx;
  if (L->kind == JH_LAW_CUSTOM) {
    custom_launch(L, dt, A, r);
    if (L->nsrc)
      hipLaunchKernelGGL(sources_kernel, dim3((unsigned)((L->nsrc * L->N + 255) / 256)), dim3(256), 0, ctx->stream, r->d.p,
                         L->src_cell.p, L->src_val.p, L->nsrc, L->N);
    k_zero_slots(ctx->stream, A->val.p, P.d_shadow_slots.p, (int64_t)P.shadow_slots.size(), P.bs * P.bs);
    return;
  }
  LawPar par;
  par.rho0[0] = L->par[0]; par.rho0[1] = L->par[1];
  par.comp[0] = L->par[2]; par.comp[1] = L->par[3];
  par.mu[0] = L->par[4]; par.mu[1] = L->par[5];
  par.p_ref = L->par[6];
  par.inv_mu[0] = 1.0 / par.mu[0]; par.inv_mu[1] = 1.0 / par.mu[1];
  int reg_row = P.iperm.empty() ? 0 : P.iperm[0];
  int chunk = (P.ntiles + NUM_XCD - 1) / NUM_XCD;
  dim3 grid(chunk * NUM_XCD), block(TILE_THREADS);
  const double *g = L->has_gdz ? L->gnz.p : nullptr;
#define JH_ASM_ARGS P.d_tile_desc.p, P.ntiles, P.d_rowptr.p, P.d_col.p, P.d_diag.p, L->Tnz.p, g, L->X.p, L->X0.p, A->val.p, r->d.p, dt, par, reg_row, (int)P.n
  const bool pipe = ctx->opt.asm_pipe != 0;
  // persistent grid of the pipelined scalar kernels: 8 workgroups per CU like the SpMV
  dim3 pgrid((unsigned)(std::min(chunk, 256) * NUM_XCD));
  const int wg2 = (int)ctx->opt.asm_pipe2_wgs;  // per XCD; 0 = tile kernel (the default: the pipelined kernel is not faster on 2x2 blocks)
  const bool pipe2 = pipe && wg2 > 0;
  dim3 pgrid2((unsigned)(std::min(chunk, std::max(wg2, 1)) * NUM_XCD));
#define JH_ASM_PIPE_ARGS P.d_tile_desc.p, P.ntiles, P.d_rowptr.p, P.d_col.p, L->Tnz.p, g, L->X.p, L->X0.p, A->val.p, r->d.p, dt, par, reg_row, (int)P.n
  switch (L->kind) {
    case JH_LAW_POISSON:
      if (pipe) hipLaunchKernelGGL(assemble_pipe_kernel<JH_LAW_POISSON>, pgrid, block, 0, ctx->stream, JH_ASM_PIPE_ARGS);
      else hipLaunchKernelGGL(assemble_tile_kernel<JH_LAW_POISSON>, grid, block, 0, ctx->stream, JH_ASM_ARGS);
      break;
    case JH_LAW_COMPRESSIBLE:
      if (pipe) hipLaunchKernelGGL(assemble_pipe_kernel<JH_LAW_COMPRESSIBLE>, pgrid, block, 0, ctx->stream, JH_ASM_PIPE_ARGS);
      else hipLaunchKernelGGL(assemble_tile_kernel<JH_LAW_COMPRESSIBLE>, grid, block, 0, ctx->stream, JH_ASM_ARGS);
      break;
    case JH_LAW_TWOPHASE:
      // 2x2 blocks: 31.5 KB of LDS per workgroup, five of them per CU
      if (pipe2) hipLaunchKernelGGL(assemble_pipe_kernel<JH_LAW_TWOPHASE>, pgrid2, block, 0, ctx->stream, JH_ASM_PIPE_ARGS);
      else hipLaunchKernelGGL(assemble_tile_kernel<JH_LAW_TWOPHASE>, grid, block, 0, ctx->stream, JH_ASM_ARGS);
      break;
    default: JH_THROW("unknown law kind");
  }
#undef JH_ASM_ARGS
#undef JH_ASM_PIPE_ARGS
  if (L->nsrc)
    hipLaunchKernelGGL(sources_kernel, dim3((unsigned)((L->nsrc * L->N + 255) / 256)), dim3(256), 0, ctx->stream, r->d.p,
                       L->src_cell.p, L->src_val.p, L->nsrc, L->N);
  // multigraph neighbourships: only the last parallel face's off-diagonal is stored (Pattern::shadow_slots)
  k_zero_slots(ctx->stream, A->val.p, P.d_shadow_slots.p, (int64_t)P.shadow_slots.size(), P.bs * P.bs);
}

// update_primary_variables! / choose_increment (variables/utils.jl:146-174): scale -> abs -> rel -> lower -> upper
__global__ void update_primary_kernel(double *X, const double *dx, double w, int64_t n, int N, const double *lim) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * N; i += (int64_t)gridDim.x * blockDim.x) {
    int e = (int)(i % N);
    double v = X[i];
    double dv = w * dx[i];
    if (lim) {
      const double scale = lim[5 * e + 0], amax = lim[5 * e + 1], rmax = lim[5 * e + 2], lo = lim[5 * e + 3], hi = lim[5 * e + 4];
      if (scale == scale) dv = dv * scale;
      if (amax == amax) dv = copysign(1.0, dv) * fmin(fabs(dv), amax) * (dv == 0.0 ? 0.0 : 1.0);
      if (rmax == rmax) { double a = rmax * fabs(v); dv = copysign(1.0, dv) * fmin(fabs(dv), a) * (dv == 0.0 ? 0.0 : 1.0); }
      if (lo == lo) dv = fmax(dv, lo - v);
      if (hi == hi) dv = fmin(dv, hi - v);
    }
    X[i] = v + dv;
  }
}
void k_update_primary(jh_law L, const double *dx, double w, const double *limits_dev) {
  int64_t n = L->disc->nc;
  int g = (int)std::min<int64_t>((n * L->N + 255) / 256, 2048);
  hipLaunchKernelGGL(update_primary_kernel, dim3(g), dim3(256), 0, L->ctx->stream, L->X.p, dx, w, n, L->N, limits_dev);
}

}  // namespace jh
