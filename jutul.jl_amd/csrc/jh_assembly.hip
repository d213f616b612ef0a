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
template <int NP>
struct Dual {
  double v;
  double d[NP];
};
template <int NP> __device__ __forceinline__ Dual<NP> dconst(double v) {
  Dual<NP> r; r.v = v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = 0.0;
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> dvar(double v, int idx) {
  Dual<NP> r = dconst<NP>(v);
  r.d[idx] = 1.0;
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator+(Dual<NP> a, Dual<NP> b) {
  Dual<NP> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator-(Dual<NP> a, Dual<NP> b) {
  Dual<NP> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(Dual<NP> a, Dual<NP> b) {
  Dual<NP> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> operator*(double s, Dual<NP> a) {
  Dual<NP> r; r.v = s * a.v;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = s * a.d[i];
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> ddiv(Dual<NP> a, double s) {
  Dual<NP> r; r.v = a.v / s;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = a.d[i] / s;
  return r;
}
template <int NP> __device__ __forceinline__ Dual<NP> dexp(Dual<NP> a) {
  double e = exp(a.v);
  Dual<NP> r; r.v = e;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = e * a.d[i];
  return r;
}

struct LawPar {
  double rho0[2], comp[2], mu[2], p_ref;
};
template <int NP> __device__ __forceinline__ Dual<NP> density(const LawPar &P, int ph, Dual<NP> p) {
  return P.rho0[ph] * dexp(P.comp[ph] * (p - dconst<NP>(P.p_ref)));  // rho0*exp(c*(p-p0))
}

// the same density from E = exp(c*(p - p0)) evaluated once per cell (twophase_exp_kernel): identical operations on identical
// inputs, hence the same bits as density()
template <int NP> __device__ __forceinline__ Dual<NP> density_from_exp(const LawPar &P, int ph, double E, int var) {
  Dual<NP> a = dconst<NP>(0.0);  // c * (p - p0): only its partials are needed
  a.d[var] = P.comp[ph] * 1.0;
  Dual<NP> r; r.v = E;
#pragma unroll
  for (int i = 0; i < NP; ++i) r.d[i] = E * a.d[i];
  return P.rho0[ph] * r;
}
// E[2c + ph] = exp(comp[ph] * (p_c - p_ref)): the four software fp64 exponentials per entry of the two-phase flux become
// four loads (per-cell property pre-pass; 32 B per cell of extra traffic against ~16 exp evaluations per cell)
__global__ void twophase_exp_kernel(const double *__restrict__ X, double *__restrict__ E, int64_t nc, LawPar par) {
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nc; c += (int64_t)gridDim.x * blockDim.x) {
    const double p = X[2 * c];
    E[2 * c] = exp(par.comp[0] * (p - par.p_ref));
    E[2 * c + 1] = exp(par.comp[1] * (p - par.p_ref));
  }
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
    Dual<2> ps = dvar<2>(Us, 0), po = dvar<2>(Uo, 1);
    Dual<2> rs = density(par, 0, ps), ro = density(par, 0, po);
    Dual<2> ravg = 0.5 * (rs + ro);             // face_average (flux.jl:372-375)
    Dual<2> dphi = (ps - po) + gz * ravg;       // two_point_potential_drop (flux.jl:335-338)
    Dual<2> f = (T / par.mu[0]) * (ravg * dphi);
    q = f.v;
    dself = f.d[0];
    dother = f.d[1];
  }
}
// ---- the kernel -------------------------------------------------------------------------------------------------
template <int KIND, bool PRE = false>
__global__ __launch_bounds__(TILE_THREADS) void assemble_tile_kernel(
    const int32_t *__restrict__ tile_row, int ntiles, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const int32_t *__restrict__ diag, const double *__restrict__ Tnz, const double *__restrict__ gnz, const double *__restrict__ X,
    const double *__restrict__ X0, double *__restrict__ nz, double *__restrict__ r, double dt, LawPar par, int reg_row,
    int nrows_total, const double *__restrict__ Eexp) {
  constexpr int N = (KIND == JH_LAW_TWOPHASE) ? 2 : 1;
  constexpr int NN = N * N;
  constexpr int TNNZ = (N == 1) ? TILE_NNZ : TILE_NNZ / 2;  // tile entry budget (Pattern::build_tiles)
  __shared__ double qv[TNNZ * N];    // flux values per entry
  __shared__ double dsv[TNNZ * NN];  // d q / d x_self per entry (column-major N x N)
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
    if (KIND != JH_LAW_TWOPHASE) {
      const double gz = gnz ? gnz[base + k] : 0.0;
      double q, ds, dother_;
      flux_scalar<KIND>(xs[own + lr], inl ? xs[cw] : X[c], T, gz, par, q, ds, dother_);
      qv[k] = q;
      dsv[k] = ds;
      off[kk][0] = dother_;
    } else {
      const double gz = gnz ? gnz[base + k] : 0.0;
      Dual<4> ps = dvar<4>(xs[(own + lr) * 2], 0), ss_w = dvar<4>(xs[(own + lr) * 2 + 1], 1);
      Dual<4> po = dvar<4>(inl ? xs[cw * 2] : X[(size_t)c * 2], 2);
      Dual<4> so_w = dvar<4>(inl ? xs[cw * 2 + 1] : X[(size_t)c * 2 + 1], 3);
      // exp(c (p - p0)) of both cells and both phases: from the per-cell pre-pass (own rows: consecutive, served by the vector
      // L1; neighbours: one 16-byte gather next to the 16-byte gather of their primary variables)
      double Es[2], Eo[2];
      if (PRE) {
        const double2 es = reinterpret_cast<const double2 *>(Eexp)[r0 + lr], eo = reinterpret_cast<const double2 *>(Eexp)[c];
        Es[0] = es.x; Es[1] = es.y; Eo[0] = eo.x; Eo[1] = eo.y;
      }
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        Dual<4> rs = PRE ? density_from_exp<4>(par, ph, Es[ph], 0) : density(par, ph, ps);
        Dual<4> ro = PRE ? density_from_exp<4>(par, ph, Eo[ph], 2) : density(par, ph, po);
        Dual<4> ravg = 0.5 * (rs + ro);
        Dual<4> dphi = (ps - po) + gz * ravg;
        Dual<4> ss = ph == 0 ? ss_w : dconst<4>(1.0) - ss_w;
        Dual<4> so = ph == 0 ? so_w : dconst<4>(1.0) - so_w;
        Dual<4> ms = (1.0 / par.mu[ph]) * ((ss * ss) * rs);
        Dual<4> mo = (1.0 / par.mu[ph]) * ((so * so) * ro);
        Dual<4> up = (dphi.v < 0.0) ? mo : ms;  // SPU upwind (flux.jl:382-405)
        Dual<4> q = T * (up * dphi);
        qv[k * 2 + ph] = q.v;
        dsv[k * 4 + 0 * 2 + ph] = q.d[0];  // (e=ph, d=0) column-major
        dsv[k * 4 + 1 * 2 + ph] = q.d[1];
        off[kk][(0 * 2 + ph) % NN] = q.d[2];
        off[kk][(1 * 2 + ph) % NN] = q.d[3];
      }
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
        Dual<2> p = dvar<2>(xs[own + tid], 0);
        Dual<2> M = vol * density(par, 0, p);
        double M0 = vol * density(par, 0, dconst<2>(X0[row])).v;
        Dual<2> a = ddiv(M - dconst<2>(M0), dt);
        ar[0] = a.v;
        ap[0] = a.d[0];
      } else {
        Dual<4> p = dvar<4>(xs[(own + tid) * 2], 0), sw = dvar<4>(xs[(own + tid) * 2 + 1], 1);
        Dual<4> so = dconst<4>(1.0) - sw;
        const double p0 = X0[(size_t)row * 2], sw0 = X0[(size_t)row * 2 + 1];
        Dual<4> Mw = vol * ((PRE ? density_from_exp<4>(par, 0, Eexp[(size_t)row * 2], 0) : density(par, 0, p)) * sw);
        Dual<4> Mo = vol * ((PRE ? density_from_exp<4>(par, 1, Eexp[(size_t)row * 2 + 1], 0) : density(par, 1, p)) * so);
        double Mw0 = vol * (density(par, 0, dconst<4>(p0)).v * sw0);
        double Mo0 = vol * (density(par, 1, dconst<4>(p0)).v * (1.0 - sw0));
        Dual<4> aw = ddiv(Mw - dconst<4>(Mw0), dt), ao = ddiv(Mo - dconst<4>(Mo0), dt);
        ar[0] = aw.v; ar[1] = ao.v;
        ap[0] = aw.d[0]; ap[1] = ao.d[0]; ap[2] = aw.d[1]; ap[3] = ao.d[1];  // column-major (e, d)
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
#pragma unroll
  for (int kk = 0; kk < KPT; ++kk) {
    const int k = tid + kk * TILE_THREADS;
    if (k >= cnt) continue;
    double *blk = nz + (size_t)(base + k) * NN;
#pragma unroll
    for (int i = 0; i < NN; ++i) blk[i] = isdiag[kk] ? dsv[k * NN + i] : off[kk][i];
  }
}


// ---- software-pipelined, persistent variant for the scalar laws -------------------------------------------------------------
// Same arithmetic and tile structure as assemble_tile_kernel; a tile is a chain of dependent latencies (descriptor ->
// (col, T) / row data -> neighbour gather), so a workgroup walks its tiles (XCD-contiguous chunks like the SpMV) and issues
// the first-level loads of its next tile before it computes the current one.  The diagonal slot's Tnz value IS the row's
// accumulation coefficient and its position IS diag[row]: the entry lane that owns the slot hands both to the row lane
// through LDS, so the row phase needs no dependent global load (and the diag array is not read at all).
struct AsmPre {
  int r0, nrows, base, cnt;
  int cidx[TILE_NNZ / TILE_THREADS];
  double Tk[TILE_NNZ / TILE_THREADS], gz[TILE_NNZ / TILE_THREADS];
  double xrow, x0;
  int rpv;
};
template <int KIND>
__global__ __launch_bounds__(TILE_THREADS) void assemble_pipe_kernel(
    const int32_t *__restrict__ tile_desc, int ntiles, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const double *__restrict__ Tnz, const double *__restrict__ gnz, const double *__restrict__ X, const double *__restrict__ X0,
    double *__restrict__ nz, double *__restrict__ r, double dt, LawPar par, int reg_row) {
  constexpr int KPT = TILE_NNZ / TILE_THREADS;
  __shared__ double qv[TILE_NNZ];
  __shared__ double dsv[TILE_NNZ];
  __shared__ double xs[TILE_ROWS];
  __shared__ double volrow[TILE_ROWS];
  __shared__ int32_t rp[TILE_ROWS + 1];
  __shared__ uint16_t diagk[TILE_ROWS];
  __shared__ uint8_t rowof[TILE_NNZ];
  const int tid = threadIdx.x;
  const int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
  const int xcd = blockIdx.x % NUM_XCD, wg = blockIdx.x / NUM_XCD, wgs = gridDim.x / NUM_XCD;
  auto in_range = [&](int tl) { return tl < chunk && xcd * chunk + tl < ntiles; };
  auto issue = [&](int tl, AsmPre &P) {
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
    P.xrow = rowlane ? X[P.r0 + tid] : 0.0;
    P.x0 = rowlane ? X0[P.r0 + tid] : 0.0;
    P.rpv = rowlane ? rowptr[P.r0 + tid] - P.base : P.cnt;
  };
  AsmPre cur, nxt;
  int tl = wg;
  bool have = in_range(tl);
  if (have) issue(tl, cur);
  while (have) {
    const int tln = tl + wgs;
    const bool have_next = in_range(tln);
    if (have_next) issue(tln, nxt);
    const int r0 = cur.r0, nrows = cur.nrows, r1 = r0 + nrows, base = cur.base, cnt = cur.cnt;
    rp[tid] = cur.rpv;
    if (tid == 0) rp[nrows] = cnt;
    xs[tid] = cur.xrow;
    __syncthreads();
    if (tid < nrows)
      for (int j = rp[tid]; j < rp[tid + 1]; ++j) rowof[j] = (uint8_t)tid;
    __syncthreads();
    // phase 1: one lane per CSR entry; off-diagonals stay in registers until the coalesced store of phase 3
    double off[KPT];
    bool isdiag[KPT];
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
      const int k = tid + kk * TILE_THREADS;
      isdiag[kk] = false;
      off[kk] = 0.0;
      if (k >= cnt) continue;
      const int lr = rowof[k];
      const int c = cur.cidx[kk];
      if (c == r0 + lr) {  // diagonal slot: no flux; hand the accumulation coefficient and the slot to the row lane
        qv[k] = 0.0;
        dsv[k] = 0.0;
        volrow[lr] = cur.Tk[kk];
        diagk[lr] = (uint16_t)k;
        isdiag[kk] = true;
        continue;
      }
      const double Uo = (c >= r0 && c < r1) ? xs[c - r0] : X[c];
      double q, ds, dother_;
      flux_scalar<KIND>(xs[lr], Uo, cur.Tk[kk], cur.gz[kk], par, q, ds, dother_);
      qv[k] = q;
      dsv[k] = ds;
      off[kk] = dother_;
    }
    __syncthreads();
    // phase 2: one lane per row
    if (tid < nrows) {
      const int row = r0 + tid;
      const double vol = volrow[tid], U = xs[tid];
      double ar, ap;
      if (KIND == JH_LAW_POISSON) {
        if (dt > 0.0) {
          ar = (vol * U - vol * cur.x0) / dt;
          ap = vol / dt;
        } else {  // stationary variant with the 1e-10*U regulariser on host cell 1 (variable_poisson.jl:101-104)
          ar = (row == reg_row) ? 1e-10 * U : 0.0;
          ap = (row == reg_row) ? 1e-10 : 0.0;
        }
      } else {
        Dual<2> p = dvar<2>(U, 0);
        Dual<2> M = vol * density(par, 0, p);
        const double M0 = vol * density(par, 0, dconst<2>(cur.x0)).v;
        Dual<2> a = ddiv(M - dconst<2>(M0), dt);
        ar = a.v;
        ap = a.d[0];
      }
      for (int j = rp[tid]; j < rp[tid + 1]; ++j) {
        ar = ar + qv[j];
        ap = ap + dsv[j];
      }
      r[row] = ar;
      dsv[diagk[tid]] = ap;  // park the diagonal in its LDS slot for the coalesced store
    }
    __syncthreads();
    // phase 3: every lane stores its entries; nzval lines leave the CU complete
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
      const int k = tid + kk * TILE_THREADS;
      if (k < cnt) nz[base + k] = isdiag[kk] ? dsv[k] : off[kk];
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
  int reg_row = P.iperm.empty() ? 0 : P.iperm[0];
  int chunk = (P.ntiles + NUM_XCD - 1) / NUM_XCD;
  dim3 grid(chunk * NUM_XCD), block(TILE_THREADS);
  const double *g = L->has_gdz ? L->gnz.p : nullptr;
  const double *Eexp = nullptr;
  // Per-cell exp pre-pass for the two-phase law: measured on MI355X, 5M cells (profiles/r03_twophase_prepass_*): assembly 0.515 ->
  // 0.579 ms (pre-pass launch included) -- the 16-byte gathers that replace the exponentials cost more than the exponentials.
  // Opt-in (JH_ASM_PREPASS=1).
  static const bool prepass = getenv("JH_ASM_PREPASS") != nullptr;
  if (L->kind == JH_LAW_TWOPHASE && prepass) {
    if (L->Eexp.n < (size_t)P.n * 2) L->Eexp.alloc((size_t)P.n * 2);
    hipLaunchKernelGGL(twophase_exp_kernel, dim3((unsigned)std::min<int64_t>((P.n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, L->X.p,
                       L->Eexp.p, P.n, par);
    Eexp = L->Eexp.p;
  }
#define JH_ASM_ARGS P.d_tile_desc.p, P.ntiles, P.d_rowptr.p, P.d_col.p, P.d_diag.p, L->Tnz.p, g, L->X.p, L->X0.p, A->val.p, r->d.p, dt, par, reg_row, (int)P.n, Eexp
  static const bool pipe = getenv("JH_ASM_NO_PIPE") == nullptr;
  // persistent grid of the pipelined scalar kernels: 8 workgroups per CU like the SpMV
  dim3 pgrid((unsigned)(std::min(chunk, 256) * NUM_XCD));
#define JH_ASM_PIPE_ARGS P.d_tile_desc.p, P.ntiles, P.d_rowptr.p, P.d_col.p, L->Tnz.p, g, L->X.p, L->X0.p, A->val.p, r->d.p, dt, par, reg_row
  switch (L->kind) {
    case JH_LAW_POISSON:
      if (pipe) hipLaunchKernelGGL(assemble_pipe_kernel<JH_LAW_POISSON>, pgrid, block, 0, ctx->stream, JH_ASM_PIPE_ARGS);
      else hipLaunchKernelGGL((assemble_tile_kernel<JH_LAW_POISSON, false>), grid, block, 0, ctx->stream, JH_ASM_ARGS);
      break;
    case JH_LAW_COMPRESSIBLE:
      if (pipe) hipLaunchKernelGGL(assemble_pipe_kernel<JH_LAW_COMPRESSIBLE>, pgrid, block, 0, ctx->stream, JH_ASM_PIPE_ARGS);
      else hipLaunchKernelGGL((assemble_tile_kernel<JH_LAW_COMPRESSIBLE, false>), grid, block, 0, ctx->stream, JH_ASM_ARGS);
      break;
    case JH_LAW_TWOPHASE:
      if (Eexp) hipLaunchKernelGGL((assemble_tile_kernel<JH_LAW_TWOPHASE, true>), grid, block, 0, ctx->stream, JH_ASM_ARGS);
      else hipLaunchKernelGGL((assemble_tile_kernel<JH_LAW_TWOPHASE, false>), grid, block, 0, ctx->stream, JH_ASM_ARGS);
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
