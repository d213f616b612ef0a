/*
 * jutul_oracle.c -- CPU restatement of the Jutul.jl hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path (jutul.jl_amd/, libjutul_hip.so) never does.
 *
 * Parity status: the reference is pure Julia and cannot be executed in this environment (no julia
 * binary, no network), so this restatement is pinned by the reference's OWN known-answer tests and
 * fixtures (tests/test_oracle_kat.py: Poisson 3x1 -> [0,1/3,2/3], pico.mat neighborship, partition KATs,
 * unit-cube transmissibilities, heat identity, layout vectors) and by mathematical identities where the
 * reference has no test (L*U == A on the pattern, Jacobian vs finite differences).  The BiCGStab
 * restatement follows the published algorithm of Krylov.jl 0.9 (third-party, un-vendored): "parity
 * unpinned" for per-iterate bits, pinned on converged solution + iteration counts.
 *
 * Conventions: every index array crossing this API holds the reference's 1-based Int64 values
 * (context.jl:76-78: Float64 / Int64).  Each function cites the reference file:line it follows
 * (paths relative to the Jutul.jl source tree).
 *
 * Threading mirrors the reference's Polyester `@batch` loops with `#pragma omp parallel for
 * schedule(static)` (owner-computes over cells/rows, no atomics).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t I64;

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */
static int cmp_i64(const void *a, const void *b) {
  I64 x = *(const I64 *)a, y = *(const I64 *)b;
  return (x > y) - (x < y);
}
static void sort_i64(I64 *v, I64 n) {
  if (n < 24) { /* insertion sort: cell_faces lists are tiny (utils.jl:836-838 `sort!`) */
    for (I64 i = 1; i < n; ++i) {
      I64 x = v[i], j = i - 1;
      while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; }
      v[j + 1] = x;
    }
  } else {
    qsort(v, (size_t)n, sizeof(I64), cmp_i64);
  }
}

int jo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void jo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* A.1 connectivity                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* get_cell_faces + get_facepos, src/utils.jl:813-874.
 * N is the 2 x nf neighborship, column-major (N[2*j+i] == N[i+1, j+1]).
 * faces: length 2*nf, facepos: length nc+1 (1-based cumulative, facepos[0] == 1).
 * Scan order i=1:2 outer, j=1:nf inner (utils.jl:830-834), then each list is sort!ed (836-838). */
int jo_get_facepos(const I64 *N, I64 nf, I64 nc, I64 *faces, I64 *facepos) {
  I64 *cnt = (I64 *)calloc((size_t)nc + 1, sizeof(I64));
  if (!cnt) return -1;
  for (I64 i = 0; i < 2; ++i)
    for (I64 j = 0; j < nf; ++j) {
      I64 c = N[2 * j + i];
      if (c < 1 || c > nc) { free(cnt); return -2; } /* utils.jl:822 assert max_n <= nc */
      cnt[c - 1]++;
    }
  facepos[0] = 1; /* cumsum([1; counts]) utils.jl:869 */
  for (I64 c = 0; c < nc; ++c) facepos[c + 1] = facepos[c] + cnt[c];
  for (I64 c = 0; c < nc; ++c) cnt[c] = facepos[c] - 1; /* cursor, 0-based */
  for (I64 i = 0; i < 2; ++i)
    for (I64 j = 0; j < nf; ++j) {
      I64 c = N[2 * j + i];
      faces[cnt[c - 1]++] = j + 1;
    }
  for (I64 c = 0; c < nc; ++c) sort_i64(faces + (facepos[c] - 1), facepos[c + 1] - facepos[c]);
  free(cnt);
  return 0;
}

/* half_face_map (src/domains.jl:101-122) and get_connection / TwoPointPotentialFlowHardCoded
 * (src/conservation/flux.jl:128-142,172-190); get_facesigns (src/utils.jl:876-890) gives the same signs.
 * Outputs per half-face k: self, other, face, sign (+1 iff N[1,face]==self). */
int jo_half_face_map(const I64 *N, I64 nf, I64 nc, const I64 *faces, const I64 *facepos, I64 *self,
                     I64 *other, I64 *sign) {
  (void)nf;
  for (I64 c = 1; c <= nc; ++c)
    for (I64 k = facepos[c - 1]; k <= facepos[c] - 1; ++k) {
      I64 f = faces[k - 1];
      I64 l = N[2 * (f - 1)], r = N[2 * (f - 1) + 1];
      self[k - 1] = c;
      if (l == c) { /* flux.jl:129-131 / domains.jl:110-112 */
        sign[k - 1] = 1;
        other[k - 1] = r;
      } else {
        if (r != c) return -3; /* domains.jl:114 @assert r == i */
        sign[k - 1] = -1;
        other[k - 1] = l;
      }
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.2 transmissibilities                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* expand_perm (src/discretization/finite-volume.jl:156-218): K as symmetric dim x dim matrix.
 * np = number of permeability entries per cell. */
static int expand_perm(const double *K, int np, int dim, double Km[9]) {
  memset(Km, 0, 9 * sizeof(double));
  if (dim == 1) { Km[0] = K[0]; return np == 1 ? 0 : -1; }
  if (dim == 2) {
    double xx, xy = 0, yy;
    if (np == 1) { xx = yy = K[0]; }
    else if (np == 2) { xx = K[0]; yy = K[1]; }
    else if (np == 3) { xx = K[0]; xy = K[1]; yy = K[2]; }
    else return -1;
    Km[0] = xx; Km[1] = xy; Km[2] = xy; Km[3] = yy; /* row-major 2x2 */
    return 0;
  }
  double xx, yy, zz, xy = 0, xz = 0, yz = 0;
  if (np == 1) { xx = yy = zz = K[0]; }
  else if (np == 3) { xx = K[0]; yy = K[1]; zz = K[2]; }
  else if (np == 6) { xx = K[0]; xy = K[1]; xz = K[2]; yy = K[3]; yz = K[4]; zz = K[5]; }
  else return -1;
  Km[0] = xx; Km[1] = xy; Km[2] = xz;
  Km[3] = xy; Km[4] = yy; Km[5] = yz;
  Km[6] = xz; Km[7] = yz; Km[8] = zz;
  return 0;
}

/* compute_half_face_trans! + half_face_trans (finite-volume.jl:130-154,220-222):
 * T_hf = A * dot(K*C, sgn*n) / dot(C, C), C = x_face - x_cell.
 * cell_centroids [dim, nc], face_centroids [dim, nf], normals [dim, nf] (column-major), perm [np, nc]. */
int jo_half_face_trans(I64 nc, int dim, const double *cc, const double *fc, const double *fn,
                       const double *areas, const double *perm, int np, const I64 *faces,
                       const I64 *facepos, const I64 *signs, double *T_hf) {
  for (I64 c = 1; c <= nc; ++c)
    for (I64 k = facepos[c - 1]; k <= facepos[c] - 1; ++k) {
      I64 f = faces[k - 1];
      double sgn = (double)signs[k - 1];
      double Km[9];
      if (expand_perm(perm + (size_t)np * (c - 1), np, dim, Km)) return -1;
      double C[3] = {0, 0, 0}, Nn[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) {
        C[d] = fc[(size_t)dim * (f - 1) + d] - cc[(size_t)dim * (c - 1) + d];
        Nn[d] = sgn * fn[(size_t)dim * (f - 1) + d];
      }
      double num = 0, den = 0;
      for (int a = 0; a < dim; ++a) {
        double kc = 0;
        for (int b = 0; b < dim; ++b) kc += Km[a * dim + b] * C[b];
        num += kc * Nn[a];
        den += C[a] * C[a];
      }
      T_hf[k - 1] = areas[f - 1] * num / den;
    }
  return 0;
}

/* compute_face_trans (finite-volume.jl:224-233): harmonic sum in half-face storage order. */
int jo_face_trans(I64 nf, I64 nhf, const double *T_hf, const I64 *faces, double *T) {
  for (I64 f = 0; f < nf; ++f) T[f] = 0.0;
  for (I64 i = 0; i < nhf; ++i) T[faces[i] - 1] += 1.0 / T_hf[i];
  for (I64 f = 0; f < nf; ++f) T[f] = 1.0 / T[f];
  return 0;
}

/* compute_face_gdz (finite-volume.jl:304-313) */
int jo_face_gdz(const I64 *N, I64 nf, const double *z, double g, double *gdz) {
  for (I64 i = 0; i < nf; ++i) {
    I64 l = N[2 * i], r = N[2 * i + 1];
    gdz[i] = -g * (z[r - 1] - z[l - 1]);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.3 Jacobian pattern and positions                                                         */
/* ------------------------------------------------------------------------------------------ */

/* declare_pattern (conservation.jl:486-505) -> (I,J) = (self,other) U (i,i);
 * static_sparsity_sparse (StaticCSR/mat.jl:73-76) = CSR(sparse(J,I,V,m,n)): per row ascending unique
 * columns.  Cell-level pattern (BlockMajorLayout, or any layout with N == 1).
 * Call with colidx == NULL to get nnz via rowptr[nc]-1, then again with storage. */
int jo_csr_pattern(I64 nc, const I64 *facepos, const I64 *other, I64 *rowptr, I64 *colidx) {
  rowptr[0] = 1;
  I64 cap = 64;
  I64 *tmp = (I64 *)malloc((size_t)cap * sizeof(I64));
  for (I64 c = 1; c <= nc; ++c) {
    I64 n = facepos[c] - facepos[c - 1];
    if (n + 1 > cap) { cap = 2 * (n + 1); tmp = (I64 *)realloc(tmp, (size_t)cap * sizeof(I64)); }
    for (I64 k = 0; k < n; ++k) tmp[k] = other[facepos[c - 1] - 1 + k];
    tmp[n] = c;
    sort_i64(tmp, n + 1);
    I64 m = 0;
    for (I64 k = 0; k <= n; ++k)
      if (k == 0 || tmp[k] != tmp[k - 1]) { /* duplicates summed by sparse() -> one entry */
        if (colidx) colidx[rowptr[c - 1] - 1 + m] = tmp[k];
        ++m;
      }
    rowptr[c] = rowptr[c - 1] + m;
  }
  free(tmp);
  return 0;
}

/* alignment_linear_index (equations.jl:132-138).  layout: 0 EquationMajor, 1 EntityMajor, 2 BlockMajor. */
I64 jo_alignment_linear_index(I64 index_outer, I64 index_inner, I64 n_outer, I64 n_inner, int layout) {
  if (layout == 0) return n_outer * (index_inner - 1) + index_outer;
  return n_inner * (index_outer - 1) + index_inner;
}

/* find_sparse_position for StaticSparsityMatrixCSR (equations.jl:163-174): linear scan, 0 if absent */
static I64 find_sparse_position_csr(const I64 *rowptr, const I64 *colidx, I64 row, I64 col) {
  for (I64 p = rowptr[row - 1]; p <= rowptr[row] - 1; ++p)
    if (colidx[p - 1] == col) return p;
  return 0;
}
I64 jo_find_sparse_position_csr(const I64 *rowptr, const I64 *colidx, I64 row, I64 col) {
  return find_sparse_position_csr(rowptr, colidx, row, col);
}

/* Scalar-layout expansion of the cell-level pattern (models.jl:585-611 with equations.jl:349-392):
 * rows row(i,e), cols col(j,d) for all e,d in 1..N of every block (i,j); result CSR (ascending cols).
 * layout 0 EquationMajor: (e-1)*nc+i ; layout 1 EntityMajor: N*(i-1)+e. */
int jo_csr_pattern_scalar(I64 nc, int N, int layout, const I64 *b_rowptr, const I64 *b_colidx,
                          I64 *rowptr, I64 *colidx) {
  I64 n = nc * N;
  rowptr[0] = 1;
  for (I64 row = 1; row <= n; ++row) {
    I64 i, e;
    if (layout == 0) { e = (row - 1) / nc + 1; i = (row - 1) % nc + 1; }
    else { i = (row - 1) / N + 1; e = (row - 1) % N + 1; }
    (void)e;
    I64 len = (b_rowptr[i] - b_rowptr[i - 1]) * N;
    if (colidx) {
      I64 *dst = colidx + rowptr[row - 1] - 1;
      I64 m = 0;
      if (layout == 0) {
        for (int d = 1; d <= N; ++d)
          for (I64 p = b_rowptr[i - 1]; p <= b_rowptr[i] - 1; ++p)
            dst[m++] = (I64)(d - 1) * nc + b_colidx[p - 1];
      } else {
        for (I64 p = b_rowptr[i - 1]; p <= b_rowptr[i] - 1; ++p)
          for (int d = 1; d <= N; ++d) dst[m++] = (I64)N * (b_colidx[p - 1] - 1) + d;
      }
    }
    rowptr[row] = rowptr[row - 1] + len;
  }
  return 0;
}

/* Position tables (conservation.jl:143-216 with ad.jl:46-61,103-169; core_types.jl:774-783):
 *   pos_acc [(e-1)*np+d, cell]  -> slot of (row(cell,e), col(cell,d))              (diagonal_alignment!)
 *   pos_flux[(e-1)*np+d, k]     -> slot of (row(other_k,e), col(self_k,d))         (align_half_face_cells)
 * Block layout (equations.jl:95-113): ix = (pos-1)*N^2 + N*(d-1) + e with pos the block slot.
 * Scalar layouts: slot in the expanded CSR (s_rowptr/s_colidx from jo_csr_pattern_scalar; for N==1 the
 * cell-level pattern itself).  `active` (length nc, may be NULL): 0 marks a cell without equations
 * (ghost/boundary cell of a buffered subdomain) -> position 0 (conservation.jl:192-198). */
int jo_align(I64 nc, I64 nhf, int N, int layout, const I64 *rowptr, const I64 *colidx,
             const I64 *s_rowptr, const I64 *s_colidx, const I64 *self, const I64 *other,
             const I64 *active, I64 *pos_acc, I64 *pos_flux) {
  I64 NN = (I64)N * N;
  for (I64 c = 1; c <= nc; ++c)
    for (int e = 1; e <= N; ++e)
      for (int d = 1; d <= N; ++d) {
        I64 out;
        if (active && !active[c - 1]) out = 0;
        else if (layout == 2 || N == 1) {
          I64 pos = find_sparse_position_csr(rowptr, colidx, c, c);
          if (!pos) return -4; /* equations.jl:146-151 "Jacobian alignment failed" */
          out = (pos - 1) * NN + (I64)N * (d - 1) + e;
        } else {
          I64 row = jo_alignment_linear_index(c, e, nc, N, layout);
          I64 col = jo_alignment_linear_index(c, d, nc, N, layout);
          out = find_sparse_position_csr(s_rowptr, s_colidx, row, col);
          if (!out) return -4;
        }
        pos_acc[NN * (c - 1) + (I64)(e - 1) * N + (d - 1)] = out;
      }
  for (I64 k = 1; k <= nhf; ++k) {
    I64 s = self[k - 1], o = other[k - 1];
    for (int e = 1; e <= N; ++e)
      for (int d = 1; d <= N; ++d) {
        I64 out;
        if (active && (!active[s - 1] || !active[o - 1])) out = 0;
        else if (layout == 2 || N == 1) {
          I64 pos = find_sparse_position_csr(rowptr, colidx, o, s); /* row=other, col=self */
          if (!pos) return -4;
          out = (pos - 1) * NN + (I64)N * (d - 1) + e;
        } else {
          I64 row = jo_alignment_linear_index(o, e, nc, N, layout);
          I64 col = jo_alignment_linear_index(s, d, nc, N, layout);
          out = find_sparse_position_csr(s_rowptr, s_colidx, row, col);
          if (!out) return -4;
        }
        pos_flux[NN * (k - 1) + (I64)(e - 1) * N + (d - 1)] = out;
      }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.4 flux, accumulation (a-5) and fill (a-6)                                                */
/* ------------------------------------------------------------------------------------------ */

/* Law kinds.  Physics is expressed through the reference operators:
 *  0 POISSON      N=1  q = -K_f (U_o - U_s); acc = vol*(U-U0)/dt
 *                      (applications/test_systems/variable_poisson/variable_poisson.jl:108-133 with vol == 1;
 *                       optional regulariser +1e-10*U on cell 1 when dt <= 0 -> stationary variant :90-106)
 *  1 COMPRESSIBLE N=1  M = vol*phi*rho(p), rho = rho0*exp(c*(p-p0)); q = T_f * face_average(rho) / mu *
 *                      two_point_potential_drop(p_s,p_o,gdz_signed,rho_s,rho_o)
 *                      (flux.jl:335-338 two_point_potential_drop, :372-375 face_average)
 *  2 TWOPHASE     N=2  (p, S_w) immiscible: M_a = vol*phi*rho_a(p)*S_a, q_a = T_f*upwind(lambda_a*rho_a)*dPhi_a,
 *                      dPhi_a = two_point_potential_drop(...), SPU upwind flag q<0 ? other : self (flux.jl:382-405)
 *                      kr_a = S_a^2, S_o = 1-S_w.
 * The reference ships no concrete face_flux! for ConservationLaw (conservation.jl:642-653 are error stubs),
 * so kinds 1 and 2 are build-defined; the hot-path structure (a-5/a-6) is the reference's. */
typedef struct {
  int kind;
  int N;
  double dt;
  /* parameters: rho0[2], c[2], mu[2], p0 */
  double rho0[2], comp[2], mu[2], p_ref;
} jo_law;

/* "Dual" with partials wrt the N primary variables of `self` only (LocalStateAD, local_ad.jl:44-79) */
typedef struct { double v; double d[2]; } dual;

static inline dual d_const(double v) { dual r = {v, {0, 0}}; return r; }
static inline dual d_add(dual a, dual b) { dual r = {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1]}}; return r; }
static inline dual d_sub(dual a, dual b) { dual r = {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1]}}; return r; }
static inline dual d_mul(dual a, dual b) {
  dual r = {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1]}};
  return r;
}
static inline dual d_scale(double s, dual a) { dual r = {s * a.v, {s * a.d[0], s * a.d[1]}}; return r; }
static inline dual d_divs(dual a, double s) { dual r = {a.v / s, {a.d[0] / s, a.d[1] / s}}; return r; }
static inline dual d_exp(dual a) { double e = exp(a.v); dual r = {e, {e * a.d[0], e * a.d[1]}}; return r; }

static inline dual law_density(const jo_law *L, int ph, dual p) {
  /* rho = rho0*exp(c*(p-p0)) */
  dual x = d_scale(L->comp[ph], d_sub(p, d_const(L->p_ref)));
  return d_scale(L->rho0[ph], d_exp(x));
}

/* Per-cell state views: X [N, nc] column-major (primary vars), X0 previous step, vol [nc] accumulation
 * coefficient (vol*phi), Tf [nf], gdz [nf] (may be NULL -> 0). */

/* update_accumulation! (conservation.jl:558-568): acc[e,c] = (M[e,c]-M0[e,c])/dt as Dual. */
static inline void law_accumulation(const jo_law *L, const double *X, const double *X0, const double *vol,
                                    I64 c, dual acc[2]) {
  int N = L->N;
  if (L->kind == 0) {
    dual U = {X[c - 1], {1, 0}};
    if (L->dt > 0) {
      /* accumulation_term (conservation.jl:97-99): (M - M0)/dt with M = vol*U; vol == 1 reproduces
       * (U_self - U0)/dt of variable_poisson.jl:131 bit for bit */
      dual M = d_scale(vol[c - 1], U);
      acc[0] = d_divs(d_sub(M, d_const(vol[c - 1] * X0[c - 1])), L->dt);
    } else {
      acc[0] = d_const(0.0);
      if (c == 1) acc[0] = d_scale(1e-10, U); /* variable_poisson.jl:101-104 regulariser */
    }
  } else if (L->kind == 1) {
    dual p = {X[c - 1], {1, 0}};
    dual M = d_scale(vol[c - 1], law_density(L, 0, p));
    dual M0 = d_const(vol[c - 1] * law_density(L, 0, d_const(X0[c - 1])).v);
    acc[0] = d_divs(d_sub(M, M0), L->dt);
  } else {
    dual p = {X[N * (c - 1)], {1, 0}};
    dual sw = {X[N * (c - 1) + 1], {0, 1}};
    dual so = d_sub(d_const(1.0), sw);
    double p0 = X0[N * (c - 1)], sw0 = X0[N * (c - 1) + 1];
    dual Mw = d_scale(vol[c - 1], d_mul(law_density(L, 0, p), sw));
    dual Mo = d_scale(vol[c - 1], d_mul(law_density(L, 1, p), so));
    double Mw0 = vol[c - 1] * (law_density(L, 0, d_const(p0)).v * sw0);
    double Mo0 = vol[c - 1] * (law_density(L, 1, d_const(p0)).v * (1.0 - sw0));
    acc[0] = d_divs(d_sub(Mw, d_const(Mw0)), L->dt);
    acc[1] = d_divs(d_sub(Mo, d_const(Mo0)), L->dt);
  }
}

/* face_flux for one half-face (conservation.jl:586-626 calling contract :642-649): flux of eq e OUT of
 * `self` across `face`; partials wrt self only (neighbour frozen as value). */
static inline void law_half_face_flux(const jo_law *L, const double *X, const double *Tf, const double *gdz,
                                      I64 s, I64 o, I64 f, I64 sgn, dual q[2]) {
  int N = L->N;
  double T = Tf[f - 1];
  if (L->kind == 0) {
    dual Us = {X[s - 1], {1, 0}};
    dual Uo = d_const(X[o - 1]);
    /* return -K[face]*(U_other - U_self)   (variable_poisson.jl:99,124) */
    dual g = d_sub(Uo, Us);
    q[0] = d_scale(-1.0, d_scale(T, g));
    /* -(K*g): value -(K*(Uo-Us)), partial -(K*(-1)) = K */
  } else if (L->kind == 1) {
    double gz = gdz ? (double)sgn * gdz[f - 1] : 0.0; /* gdz stored left->right; signed to self->other */
    dual ps = {X[s - 1], {1, 0}};
    dual po = d_const(X[o - 1]);
    dual rs = law_density(L, 0, ps), ro = law_density(L, 0, po);
    dual ravg = d_scale(0.5, d_add(rs, ro));           /* face_average flux.jl:372-375 */
    dual dphi = d_add(d_sub(ps, po), d_scale(gz, ravg)); /* two_point_potential_drop flux.jl:335-338 */
    q[0] = d_scale(T / L->mu[0], d_mul(ravg, dphi));
  } else {
    double gz = gdz ? (double)sgn * gdz[f - 1] : 0.0;
    dual ps = {X[N * (s - 1)], {1, 0}};
    dual sws = {X[N * (s - 1) + 1], {0, 1}};
    dual po = d_const(X[N * (o - 1)]);
    dual swo = d_const(X[N * (o - 1) + 1]);
    for (int ph = 0; ph < 2; ++ph) {
      dual rs = law_density(L, ph, ps), ro = law_density(L, ph, po);
      dual ravg = d_scale(0.5, d_add(rs, ro));
      dual dphi = d_add(d_sub(ps, po), d_scale(gz, ravg));
      /* mobility*density upwinded (SPU): flow out of self when dphi >= 0 -> self is upstream.
       * upwind(): flag = q < 0 ? right : left with q oriented self->other (flux.jl:382-405). */
      dual ss = ph == 0 ? sws : d_sub(d_const(1.0), sws);
      dual so = ph == 0 ? swo : d_sub(d_const(1.0), swo);
      dual ms = d_scale(1.0 / L->mu[ph], d_mul(d_mul(ss, ss), rs));
      dual mo = d_scale(1.0 / L->mu[ph], d_mul(d_mul(so, so), ro));
      dual up = (dphi.v < 0.0) ? mo : ms;
      q[ph] = d_scale(T, d_mul(up, dphi));
    }
  }
}

/* a-5: update_accumulation! + update_half_face_flux_tpfa! : materialise the Dual arrays exactly like the
 * reference (acc[ne, nc], hf[ne, nhf]; value + np partials each). Arrays laid out [1+np, ne, n]. */
int jo_update_equation(const jo_law *L, I64 nc, I64 nhf, const I64 *facepos, const I64 *self,
                       const I64 *other, const I64 *face, const I64 *sign, const double *X, const double *X0,
                       const double *vol, const double *Tf, const double *gdz, double *acc, double *hf) {
  (void)nhf;
  int N = L->N;
  int w = 1 + N;
#pragma omp parallel for schedule(static)
  for (I64 c = 1; c <= nc; ++c) {
    dual a[2];
    law_accumulation(L, X, X0, vol, c, a);
    for (int e = 0; e < N; ++e) {
      double *dst = acc + ((size_t)(c - 1) * N + e) * w;
      dst[0] = a[e].v;
      for (int d = 0; d < N; ++d) dst[1 + d] = a[e].d[d];
    }
    for (I64 k = facepos[c - 1]; k <= facepos[c] - 1; ++k) {
      dual q[2];
      law_half_face_flux(L, X, Tf, gdz, self[k - 1], other[k - 1], face[k - 1], sign[k - 1], q);
      for (int e = 0; e < N; ++e) {
        double *dst = hf + ((size_t)(k - 1) * N + e) * w;
        dst[0] = q[e].v;
        for (int d = 0; d < N; ++d) dst[1 + d] = q[e].d[d];
      }
    }
  }
  return 0;
}

/* apply_forces! -> PoissonSource-style sources add `value` to the diagonal AD entry's value
 * (models.jl:889-901, variable_poisson.jl:78-84) BEFORE the fill (models.jl:729-733 then :762). */
int jo_apply_sources(int N, I64 nsrc, const I64 *cells, const double *values, double *acc) {
  int w = 1 + N;
  for (I64 i = 0; i < nsrc; ++i)
    for (int e = 0; e < N; ++e) acc[((size_t)(cells[i] - 1) * N + e) * w] += values[(size_t)i * N + e];
  return 0;
}

/* a-6: fill_conservation_eq! (conservation.jl:373-430) via threaded_fill_conservation_eq! (:366-371).
 * r is [N, nc] (entity/block-major view) -- for EquationMajorLayout pass r_stride_e = nc, r_stride_c = 1
 * (transposed view utils.jl:65-85), otherwise r_stride_e = 1, r_stride_c = N. */
int jo_fill_conservation_eq(int N, I64 nc, const I64 *facepos, const double *acc, const double *hf,
                            const I64 *pos_acc, const I64 *pos_flux, double *nz, double *r, I64 r_stride_e,
                            I64 r_stride_c) {
  int w = 1 + N;
  I64 NN = (I64)N * N;
#pragma omp parallel for schedule(static)
  for (I64 c = 1; c <= nc; ++c) {
    double acc_r[2] = {0, 0}, acc_p[2][2] = {{0, 0}, {0, 0}};
    for (int e = 0; e < N; ++e) {
      const double *a = acc + ((size_t)(c - 1) * N + e) * w;
      acc_r[e] = a[0];
      for (int p = 0; p < N; ++p) acc_p[e][p] = a[1 + p];
    }
    for (I64 i = facepos[c - 1]; i <= facepos[c] - 1; ++i) {
      for (int e = 0; e < N; ++e) acc_r[e] = acc_r[e] + hf[((size_t)(i - 1) * N + e) * w];
      I64 fpos_outer = pos_flux[NN * (i - 1)]; /* get_jacobian_pos(cell_flux, i, 1, 1) :397 */
      int is_inner = fpos_outer > 0;
      for (int p = 0; p < N; ++p)
        for (int e = 0; e < N; ++e) {
          double dq = hf[((size_t)(i - 1) * N + e) * w + 1 + p];
          acc_p[e][p] = acc_p[e][p] + dq;
          if (is_inner) {
            I64 fpos = pos_flux[NN * (i - 1) + (I64)e * N + p];
            nz[fpos - 1] = -dq; /* update_jacobian_inner! ad.jl:74-76: plain store */
          }
        }
    }
    for (int e = 0; e < N; ++e) r[(I64)e * r_stride_e + (c - 1) * r_stride_c] = acc_r[e];
    for (int p = 0; p < N; ++p)
      for (int e = 0; e < N; ++e) {
        I64 apos = pos_acc[NN * (c - 1) + (I64)e * N + p];
        if (apos > 0) nz[apos - 1] = acc_p[e][p];
      }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.4b generic-AD alternatives (SURVEY a-7): (i) cell-based GenericAutoDiffCache, (ii) face-based */
/*      PotentialFlow{:fvm}.  Independent formulations of the same residual / Jacobian, used as  */
/*      the checker of the device path for runtime-defined laws (jh_law_create_custom).          */
/* ------------------------------------------------------------------------------------------ */

/* Dual whose partials are taken w.r.t. the N <= 3 primary variables of ONE entity (local_ad(state, var),
 * equations.jl:562-566, new_entity_index :589-590): a state value of cell c is seeded iff c == var. */
typedef struct { double v; double d[3]; } gdual;
static inline gdual g_const(double v) { gdual r = {v, {0, 0, 0}}; return r; }
static inline gdual g_add(gdual a, gdual b) { gdual r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static inline gdual g_sub(gdual a, gdual b) { gdual r; r.v = a.v - b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static inline gdual g_mul(gdual a, gdual b) { gdual r; r.v = a.v * b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static inline gdual g_scale(double s, gdual a) { gdual r; r.v = s * a.v; for (int i = 0; i < 3; ++i) r.d[i] = s * a.d[i]; return r; }
static inline gdual g_adds(gdual a, double s) { a.v += s; return a; }
static inline gdual g_exp(gdual a) { double e = exp(a.v); gdual r; r.v = e; for (int i = 0; i < 3; ++i) r.d[i] = e * a.d[i]; return r; }
static inline gdual g_sqrt(gdual a) { double q = sqrt(a.v); gdual r; r.v = q; for (int i = 0; i < 3; ++i) r.d[i] = 0.5 * a.d[i] / q; return r; }
/* flux.jl:335-405 */
static inline gdual g_face_average(gdual a, gdual b) { return g_scale(0.5, g_add(a, b)); }
static inline gdual g_potential_drop(gdual ps, gdual po, double gdz, gdual rs, gdual ro) { return g_add(g_sub(ps, po), g_scale(gdz, g_face_average(rs, ro))); }
static inline gdual g_upwind(gdual drop, gdual vs, gdual vo) { return drop.v < 0.0 ? vo : vs; }

/* The laws the device tests define as user source (tests/test_gpu_parity.py), restated on gdual:
 *  law 1 "compressible" (N = 1), par = rho0, c, mu, p_ref:
 *      rho(p) = rho0 exp(c (p - p_ref)); q = (T/mu) face_average(rho_s, rho_o) two_point_potential_drop(...); M = rho(p)
 *  law 2 "reaction" (N = 2), par = a, b, k:
 *      mob = a + face_average(v_s, v_o)^2; q_u = T (mob (u_s - u_o)); q_v = (b T)(v_s - v_o) + gdz upwind(v_s - v_o, v_s, v_o)
 *      M_u = u + k u^3; M_v = sqrt(v) u */
static void gen_flux(int law, const gdual *s, const gdual *o, double T, double gdz, const double *par, gdual *q) {
  if (law == 1) {
    gdual rs = g_scale(par[0], g_exp(g_scale(par[1], g_adds(s[0], -par[3]))));
    gdual ro = g_scale(par[0], g_exp(g_scale(par[1], g_adds(o[0], -par[3]))));
    gdual dphi = g_potential_drop(s[0], o[0], gdz, rs, ro);
    q[0] = g_scale(T / par[2], g_mul(g_face_average(rs, ro), dphi));
  } else {
    gdual fa = g_face_average(s[1], o[1]);
    gdual mob = g_adds(g_mul(fa, fa), par[0]);
    q[0] = g_scale(T, g_mul(mob, g_sub(s[0], o[0])));
    gdual dv = g_sub(s[1], o[1]);
    q[1] = g_add(g_scale(par[1] * T, dv), g_scale(gdz, g_upwind(dv, s[1], o[1])));
  }
}
static void gen_mass(int law, const gdual *x, const double *par, gdual *M) {
  if (law == 1) {
    M[0] = g_scale(par[0], g_exp(g_scale(par[1], g_adds(x[0], -par[3]))));
  } else {
    M[0] = g_add(x[0], g_scale(par[2], g_mul(x[0], g_mul(x[0], x[0]))));
    M[1] = g_mul(g_sqrt(x[1]), x[0]);
  }
}
/* state of cell c seen from the entity `var` the partials are taken for */
static inline void gen_seed(int N, const double *X, I64 c, I64 var, gdual *x) {
  for (int e = 0; e < N; ++e) { x[e] = g_const(X[(size_t)N * (c - 1) + e]); if (c == var) x[e].d[e] = 1.0; }
}

/* (i) inner_update_equation_for_entity (equations.jl:578-594) + fill_equation_entries_impl! (ad/generic.jl:53-96):
 * for every cell i and every entry j of its stencil (the CSR row: var = colidx[j]) the WHOLE cell equation
 *     R_i = vol_i (M(x_i) - M(x0_i)) / dt + sum over the half-faces of i of q(x_i, x_other)  (+ source value)
 * is evaluated with duals seeded on `var`; the value of the diagonal entry is the residual, the partials of entry j are the
 * Jacobian block (row i, column var).  No flux antisymmetry is used anywhere. */
int jo_generic_cell_assemble(int law, int N, I64 nc, const I64 *facepos, const I64 *other, const I64 *face, const I64 *sign,
                             const I64 *rowptr, const I64 *colidx, const double *X, const double *X0, const double *vol,
                             const double *Tf, const double *gdz, double dt, const double *par, I64 nsrc, const I64 *src_cells,
                             const double *src_vals, double *nz, double *r) {
  if (N < 1 || N > 3 || (law != 1 && law != 2)) return -1;
  const I64 NN = (I64)N * N;
  double *srcv = (double *)calloc((size_t)nc * N, sizeof(double));
  for (I64 k = 0; k < nsrc; ++k)
    for (int e = 0; e < N; ++e) srcv[(size_t)(src_cells[k] - 1) * N + e] += src_vals[k * N + e];
#pragma omp parallel for schedule(static)
  for (I64 i = 1; i <= nc; ++i) {
    for (I64 j = rowptr[i - 1]; j <= rowptr[i] - 1; ++j) {
      const I64 var = colidx[j - 1];
      gdual xi[3], x0i[3], M[3], M0[3], eq[3];
      gen_seed(N, X, i, var, xi);
      for (int e = 0; e < N; ++e) x0i[e] = g_const(X0[(size_t)N * (i - 1) + e]);
      gen_mass(law, xi, par, M);
      gen_mass(law, x0i, par, M0);
      for (int e = 0; e < N; ++e) eq[e] = g_scale(vol[i - 1] / dt, g_sub(M[e], g_const(M0[e].v)));
      for (I64 k = facepos[i - 1]; k <= facepos[i] - 1; ++k) {
        gdual xo[3], q[3];
        gen_seed(N, X, other[k - 1], var, xo);
        const double gz = gdz ? (double)sign[k - 1] * gdz[face[k - 1] - 1] : 0.0;
        gen_flux(law, xi, xo, Tf[face[k - 1] - 1], gz, par, q);
        for (int e = 0; e < N; ++e) eq[e] = g_add(eq[e], q[e]);
      }
      for (int e = 0; e < N; ++e) {
        if (var == i) r[(size_t)N * (i - 1) + e] = eq[e].v + srcv[(size_t)(i - 1) * N + e];  /* fill_residual on the diagonal entry */
        for (int d = 0; d < N; ++d) nz[(size_t)(j - 1) * NN + (size_t)N * d + e] = eq[e].d[d];  /* update_jacobian_entry! */
      }
    }
  }
  free(srcv);
  return 0;
}

/* (ii) face-based PotentialFlow{:fvm} assembly (conservation/fvm_assembly.jl:175-283): ONE flux per face (left -> right),
 * differentiated w.r.t. each of its two stencil cells in turn; r[l] += q, r[r] -= q and the four Jacobian blocks
 * (l,l) += dq/dl, (r,l) -= dq/dl, (l,r) += dq/dr, (r,r) -= dq/dr; accumulation + sources on the diagonal.  Serial scatter-add
 * like the reference's loop.  N_lr: neighbourship [2, nf]. */
int jo_fvm_face_assemble(int law, int N, I64 nc, I64 nf, const I64 *N_lr, const I64 *rowptr, const I64 *colidx, const double *X,
                         const double *X0, const double *vol, const double *Tf, const double *gdz, double dt, const double *par,
                         I64 nsrc, const I64 *src_cells, const double *src_vals, double *nz, double *r) {
  if (N < 1 || N > 3 || (law != 1 && law != 2)) return -1;
  const I64 NN = (I64)N * N;
  memset(nz, 0, sizeof(double) * (size_t)(rowptr[nc] - 1) * NN);
  memset(r, 0, sizeof(double) * (size_t)nc * N);
  for (I64 c = 1; c <= nc; ++c) {  /* accumulation on the diagonal block */
    gdual x[3], x0[3], M[3], M0[3];
    gen_seed(N, X, c, c, x);
    for (int e = 0; e < N; ++e) x0[e] = g_const(X0[(size_t)N * (c - 1) + e]);
    gen_mass(law, x, par, M);
    gen_mass(law, x0, par, M0);
    I64 pos = find_sparse_position_csr(rowptr, colidx, c, c);
    if (!pos) return -4;
    for (int e = 0; e < N; ++e) {
      gdual a = g_scale(vol[c - 1] / dt, g_sub(M[e], g_const(M0[e].v)));
      r[(size_t)N * (c - 1) + e] += a.v;
      for (int d = 0; d < N; ++d) nz[(size_t)(pos - 1) * NN + (size_t)N * d + e] += a.d[d];
    }
  }
  for (I64 k = 0; k < nsrc; ++k)
    for (int e = 0; e < N; ++e) r[(size_t)N * (src_cells[k] - 1) + e] += src_vals[k * N + e];
  for (I64 f = 1; f <= nf; ++f) {
    const I64 l = N_lr[2 * (f - 1)], rc = N_lr[2 * (f - 1) + 1];
    const double gz = gdz ? gdz[f - 1] : 0.0;  /* stored left -> right */
    for (int wrt = 0; wrt < 2; ++wrt) {
      const I64 var = wrt == 0 ? l : rc;
      gdual xl[3], xr[3], q[3];
      gen_seed(N, X, l, var, xl);
      gen_seed(N, X, rc, var, xr);
      gen_flux(law, xl, xr, Tf[f - 1], gz, par, q);
      I64 pl = find_sparse_position_csr(rowptr, colidx, l, var), pr = find_sparse_position_csr(rowptr, colidx, rc, var);
      if (!pl || !pr) return -4;
      for (int e = 0; e < N; ++e) {
        if (wrt == 0) { r[(size_t)N * (l - 1) + e] += q[e].v; r[(size_t)N * (rc - 1) + e] -= q[e].v; }
        for (int d = 0; d < N; ++d) {
          nz[(size_t)(pl - 1) * NN + (size_t)N * d + e] += q[e].d[d];
          nz[(size_t)(pr - 1) * NN + (size_t)N * d + e] -= q[e].d[d];
        }
      }
    }
  }
  return 0;
}

/* convergence_criterion (equations.jl:619-629): per equation max_cells |r[e, :]| */
int jo_convergence(int N, I64 nc, const double *r, I64 r_stride_e, I64 r_stride_c, double *e_out) {
  for (int e = 0; e < N; ++e) {
    double m = 0;
    for (I64 c = 0; c < nc; ++c) {
      double v = fabs(r[(I64)e * r_stride_e + c * r_stride_c]);
      if (v > m || v != v) m = v;
    }
    e_out[e] = m;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.5 SpMV                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* mul!(y, A, x, alpha, beta) / csr_mul_add! / internal_muladd (StaticCSR/mat.jl:24-68).
 * Scalar (bs==1): v = v + A_ij*x_j left->right.  Block (bs==N): v = muladd(A_ij, x_j, v) with A_ij an
 * N x N column-major SMatrix (block_cpu.jl:1-17). */
int jo_spmv(I64 n, int bs, const I64 *rowptr, const I64 *colidx, const double *nz, const double *x,
            double *y, double alpha, double beta) {
  int do_inc = 1;
  if (beta == 0.0) do_inc = 0;
  else if (beta != 1.0) {
#pragma omp parallel for schedule(static)
    for (I64 i = 0; i < n * bs; ++i) y[i] *= beta; /* rmul!(y, beta) mat.jl:35 */
  }
  if (bs == 1) {
#pragma omp parallel for schedule(static)
    for (I64 row = 1; row <= n; ++row) {
      double v = 0.0;
      for (I64 p = rowptr[row - 1]; p <= rowptr[row] - 1; ++p) v = v + nz[p - 1] * x[colidx[p - 1] - 1];
      if (do_inc) y[row - 1] += alpha * v; else y[row - 1] = alpha * v;
    }
  } else {
    int N = bs, NN = bs * bs;
#pragma omp parallel for schedule(static)
    for (I64 row = 1; row <= n; ++row) {
      double v[4] = {0, 0, 0, 0};
      for (I64 p = rowptr[row - 1]; p <= rowptr[row] - 1; ++p) {
        const double *A = nz + (size_t)(p - 1) * NN;
        const double *xj = x + (size_t)(colidx[p - 1] - 1) * N;
        for (int e = 0; e < N; ++e) {
          double s = v[e];
          for (int d = 0; d < N; ++d) s += A[d * N + e] * xj[d];
          v[e] = s;
        }
      }
      for (int e = 0; e < N; ++e) {
        if (do_inc) y[(row - 1) * N + e] += alpha * v[e]; else y[(row - 1) * N + e] = alpha * v[e];
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.6 ILU(0)                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* small dense helpers for SMatrix{N,N} blocks (column-major), N <= 2 here but written for N <= 3 */
static void blk_inv(int N, const double *A, double *R) {
  if (N == 1) { R[0] = 1.0 / A[0]; return; }
  if (N == 2) {
    double a = A[0], c = A[1], b = A[2], d = A[3]; /* [a b; c d] col-major */
    double idet = 1.0 / (a * d - b * c);
    R[0] = d * idet; R[1] = -c * idet; R[2] = -b * idet; R[3] = a * idet;
    return;
  }
  /* N == 3: adjugate */
  double m[3][3];
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) m[i][j] = A[j * 3 + i];
  double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  double idet = 1.0 / (m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02);
  double inv[3][3];
  inv[0][0] = c00 * idet; inv[1][0] = c01 * idet; inv[2][0] = c02 * idet;
  inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * idet;
  inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * idet;
  inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * idet;
  inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * idet;
  inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * idet;
  inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * idet;
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[j * 3 + i] = inv[i][j];
}
/* C = A*B (col-major N x N) */
static void blk_mul(int N, const double *A, const double *B, double *C) {
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += A[k * N + i] * B[j * N + k];
      C[j * N + i] = s;
    }
}
static int blk_nonzero(int N, const double *A) {
  for (int i = 0; i < N * N; ++i) if (A[i] != 0.0) return 1;
  return 0;
}

/* The ILU factor object mirrors ILUFactorCSR (StaticCSR/ilu0.jl:190-203): strict-L CSR, strict-U CSR,
 * D (inverted at the end), and the value maps L_map/U_map/D_map into A's nzval.  For block-Jacobi
 * (par_ilu0.jl) one such object per block with a sorted `active` row set. */
typedef struct {
  I64 n; int bs;
  /* strict-L / strict-U CSR over the ACTIVE rows only: row pointers are indexed by the local index a (active[a] = global
   * row), 1-based entry positions; columns stay global.  (The reference stores n + 1 row pointers per block and scans all
   * n rows with `insorted`, ilu0.jl:13-54; same entries, but a set-up of O(rows of the block) instead of O(n) per block.) */
  I64 *l_rowptr, *l_col, *l_map; double *l_val;
  I64 *u_rowptr, *u_col, *u_map; double *u_val;
  I64 nact; I64 *active; /* sorted global rows; all rows when nact == n */
  I64 *d_map; double *d_val; /* per active row */
  const I64 *loc;            /* shared by all blocks: global row -> local index + 1 inside its own block */
  const I64 *owner;          /* shared: global row -> block id + 1 */
  I64 id;                    /* this block's id + 1 */
} jo_ilu;
#define JO_ACTIVE(F, row) ((F)->owner[(row) - 1] == (F)->id)

/* fixed_block (ilu0.jl:13-54): keep entries with keep(col,row,lower) && col in active, rows in active */
static void fixed_block(const jo_ilu *F, const I64 *rowptr, const I64 *colidx, int lower, I64 **o_rowptr,
                        I64 **o_col, I64 **o_map) {
  I64 na = F->nact;
  I64 *rp = (I64 *)malloc((size_t)(na + 1) * sizeof(I64));
  rp[0] = 1;
  for (I64 a = 0; a < na; ++a) {
    const I64 row = F->active[a];
    I64 ctr = 0;
    for (I64 i = rowptr[row - 1]; i <= rowptr[row] - 1; ++i) {
      I64 col = colidx[i - 1];
      int keep = lower ? (col < row) : (col > row);
      if (keep && JO_ACTIVE(F, col)) ++ctr;
    }
    rp[a + 1] = rp[a] + ctr;
  }
  I64 m = rp[na] - 1;
  I64 *cols = (I64 *)malloc((size_t)(m > 0 ? m : 1) * sizeof(I64));
  I64 *map = (I64 *)malloc((size_t)(m > 0 ? m : 1) * sizeof(I64));
  I64 idx = 0;
  for (I64 a = 0; a < na; ++a) {
    const I64 row = F->active[a];
    for (I64 i = rowptr[row - 1]; i <= rowptr[row] - 1; ++i) {
      I64 col = colidx[i - 1];
      int keep = lower ? (col < row) : (col > row);
      if (keep && JO_ACTIVE(F, col)) { map[idx] = i; cols[idx] = col; ++idx; }
    }
  }
  *o_rowptr = rp; *o_col = cols; *o_map = map;
}

/* U[k, j] lookup: getindex (mat.jl:11) on the U CSR, 0 when not stored; returns pointer or NULL */
static const double *u_lookup(const jo_ilu *F, I64 k, I64 j) {
  const I64 ka = F->loc[k - 1] - 1; /* k is an active row of this block (it is the column of a kept L entry) */
  I64 lo = F->u_rowptr[ka], hi = F->u_rowptr[ka + 1] - 1;
  while (lo <= hi) { /* SparseMatrixCSC getindex = binary search over the sorted column */
    I64 mid = (lo + hi) / 2;
    I64 c = F->u_col[mid - 1];
    if (c == j) return F->u_val + (size_t)(mid - 1) * F->bs * F->bs;
    if (c < j) lo = mid + 1; else hi = mid - 1;
  }
  return NULL;
}

/* ilu0_factor! (ilu0.jl:108-144) incl. update_values! through the maps (ilu0.jl:83-98,223-231) */
static void ilu_refactor(jo_ilu *F, const double *nz) {
  int N = F->bs, NN = N * N;
  I64 nl = F->l_rowptr[F->nact] - 1, nu = F->u_rowptr[F->nact] - 1;
  for (I64 i = 0; i < nl; ++i) memcpy(F->l_val + i * NN, nz + (F->l_map[i] - 1) * NN, sizeof(double) * NN);
  for (I64 i = 0; i < nu; ++i) memcpy(F->u_val + i * NN, nz + (F->u_map[i] - 1) * NN, sizeof(double) * NN);
  for (I64 i = 0; i < F->nact; ++i) memcpy(F->d_val + i * NN, nz + (F->d_map[i] - 1) * NN, sizeof(double) * NN);
  double tmp[9], Aik[9], inv[9];
  for (I64 a = 0; a < F->nact; ++a) {
    I64 i = F->active[a];
    I64 ls = F->l_rowptr[a], le = F->l_rowptr[a + 1] - 1;
    I64 us = F->u_rowptr[a], ue = F->u_rowptr[a + 1] - 1;
    for (I64 l_i = ls; l_i <= le; ++l_i) {
      I64 k = F->l_col[l_i - 1];
      const double *Dk = F->d_val + (size_t)(F->loc[k - 1] - 1) * NN;
      if (N == 1) {
        double A_ik = F->l_val[l_i - 1] * (1.0 / Dk[0]); /* nz_l[l_i]*inv(A_kk) :121 */
        F->l_val[l_i - 1] = A_ik;
        if (A_ik != 0.0) {
          for (I64 l_j = l_i + 1; l_j <= le; ++l_j) { /* process_partial_row! on rem_l_pos */
            const double *Akj = u_lookup(F, k, F->l_col[l_j - 1]);
            F->l_val[l_j - 1] -= A_ik * (Akj ? *Akj : 0.0);
          }
          const double *Aki = u_lookup(F, k, i);
          F->d_val[a] -= A_ik * (Aki ? *Aki : 0.0);
          for (I64 u_j = us; u_j <= ue; ++u_j) {
            const double *Akj = u_lookup(F, k, F->u_col[u_j - 1]);
            F->u_val[u_j - 1] -= A_ik * (Akj ? *Akj : 0.0);
          }
        }
      } else {
        blk_inv(N, Dk, inv);
        blk_mul(N, F->l_val + (size_t)(l_i - 1) * NN, inv, Aik); /* A_ik = nz_l * inv(A_kk) */
        memcpy(F->l_val + (size_t)(l_i - 1) * NN, Aik, sizeof(double) * NN);
        if (blk_nonzero(N, Aik)) {
          for (I64 l_j = l_i + 1; l_j <= le; ++l_j) {
            const double *Akj = u_lookup(F, k, F->l_col[l_j - 1]);
            if (Akj) { blk_mul(N, Aik, Akj, tmp); for (int t = 0; t < NN; ++t) F->l_val[(l_j - 1) * NN + t] -= tmp[t]; }
          }
          const double *Aki = u_lookup(F, k, i);
          if (Aki) { blk_mul(N, Aik, Aki, tmp); for (int t = 0; t < NN; ++t) F->d_val[a * NN + t] -= tmp[t]; }
          for (I64 u_j = us; u_j <= ue; ++u_j) {
            const double *Akj = u_lookup(F, k, F->u_col[u_j - 1]);
            if (Akj) { blk_mul(N, Aik, Akj, tmp); for (int t = 0; t < NN; ++t) F->u_val[(u_j - 1) * NN + t] -= tmp[t]; }
          }
        }
      }
    }
  }
  for (I64 a = 0; a < F->nact; ++a) { /* D[i] = inv(D[i]) :141-143 */
    if (N == 1) F->d_val[a] = 1.0 / F->d_val[a];
    else { blk_inv(N, F->d_val + a * NN, inv); memcpy(F->d_val + a * NN, inv, sizeof(double) * NN); }
  }
}

static jo_ilu *ilu_setup(I64 n, int bs, const I64 *rowptr, const I64 *colidx, const I64 *active, I64 nact, const I64 *loc,
                         const I64 *owner, I64 id) {
  jo_ilu *F = (jo_ilu *)calloc(1, sizeof(jo_ilu));
  F->n = n; F->bs = bs; F->nact = nact; F->loc = loc; F->owner = owner; F->id = id;
  F->active = (I64 *)malloc((size_t)(nact > 0 ? nact : 1) * sizeof(I64));
  for (I64 a = 0; a < nact; ++a) F->active[a] = active ? active[a] : a + 1;
  fixed_block(F, rowptr, colidx, 1, &F->l_rowptr, &F->l_col, &F->l_map);
  fixed_block(F, rowptr, colidx, 0, &F->u_rowptr, &F->u_col, &F->u_map);
  int NN = bs * bs;
  I64 nl = F->l_rowptr[nact] - 1, nu = F->u_rowptr[nact] - 1;
  F->l_val = (double *)calloc((size_t)(nl > 0 ? nl : 1) * NN, sizeof(double));
  F->u_val = (double *)calloc((size_t)(nu > 0 ? nu : 1) * NN, sizeof(double));
  F->d_val = (double *)calloc((size_t)(nact > 0 ? nact : 1) * NN, sizeof(double));
  F->d_map = (I64 *)calloc((size_t)(nact > 0 ? nact : 1), sizeof(I64));
  /* diagonal_block (ilu0.jl:56-81): diagonal must be stored and precede all upper entries */
  for (I64 a = 0; a < nact; ++a) {
    I64 row = F->active[a];
    for (I64 k = rowptr[row - 1]; k <= rowptr[row] - 1; ++k)
      if (colidx[k - 1] == row) { F->d_map[a] = k; break; }
  }
  return F;
}
static void ilu_free_one(jo_ilu *F) {
  if (!F) return;
  free(F->l_rowptr); free(F->l_col); free(F->l_map); free(F->l_val);
  free(F->u_rowptr); free(F->u_col); free(F->u_map); free(F->u_val);
  free(F->active); free(F->d_map); free(F->d_val); free(F);
}

/* ParallelILUFactorCSR (par_ilu0.jl:2-90): nblocks == 1 with partition == NULL is the serial ilu0_csr(A). */
typedef struct { I64 nblocks; jo_ilu **f; I64 n; int bs; I64 *loc, *owner; } jo_ilu_par;

/* ilu0_csr(A) (ilu0.jl:213-221) / ilu0_csr(A, partition) (par_ilu0.jl:47-55); partition values 1..nparts */
jo_ilu_par *jo_ilu0_csr(I64 n, int bs, const I64 *rowptr, const I64 *colidx, const double *nz,
                        const I64 *partition) {
  jo_ilu_par *P = (jo_ilu_par *)calloc(1, sizeof(jo_ilu_par));
  P->n = n; P->bs = bs;
  P->loc = (I64 *)calloc((size_t)(n > 0 ? n : 1), sizeof(I64));
  P->owner = (I64 *)calloc((size_t)(n > 0 ? n : 1), sizeof(I64));
  if (!partition) {
    P->nblocks = 1;
    P->f = (jo_ilu **)calloc(1, sizeof(jo_ilu *));
    for (I64 i = 0; i < n; ++i) { P->loc[i] = i + 1; P->owner[i] = 1; }
    P->f[0] = ilu_setup(n, bs, rowptr, colidx, NULL, n, P->loc, P->owner, 1);
    for (I64 a = 0; a < n; ++a) if (!P->f[0]->d_map[a]) { return NULL; }
    ilu_refactor(P->f[0], nz);
    return P;
  }
  I64 np = 0;
  for (I64 i = 0; i < n; ++i) { if (partition[i] < 1) return NULL; if (partition[i] > np) np = partition[i]; }
  P->nblocks = np;
  P->f = (jo_ilu **)calloc((size_t)np, sizeof(jo_ilu *));
  I64 *cnt = (I64 *)calloc((size_t)np + 1, sizeof(I64));
  for (I64 i = 0; i < n; ++i) cnt[partition[i]]++;
  I64 **lists = (I64 **)calloc((size_t)np, sizeof(I64 *));
  for (I64 b = 0; b < np; ++b) lists[b] = (I64 *)malloc((size_t)(cnt[b + 1] > 0 ? cnt[b + 1] : 1) * sizeof(I64));
  I64 *cur = (I64 *)calloc((size_t)np, sizeof(I64));
  for (I64 i = 0; i < n; ++i) { I64 b = partition[i] - 1; lists[b][cur[b]++] = i + 1; P->loc[i] = cur[b]; P->owner[i] = b + 1; } /* findall(isequal(b), p) */
#pragma omp parallel for schedule(dynamic, 1)
  for (I64 b = 0; b < np; ++b) P->f[b] = ilu_setup(n, bs, rowptr, colidx, lists[b], cnt[b + 1], P->loc, P->owner, b + 1);
#pragma omp parallel for schedule(dynamic, 1)
  for (I64 b = 0; b < np; ++b) ilu_refactor(P->f[b], nz);
  for (I64 b = 0; b < np; ++b) free(lists[b]);
  free(lists); free(cur); free(cnt);
  return P;
}

/* ilu0_csr!(LU, A) (ilu0.jl:223-231 / par_ilu0.jl:75-80) */
int jo_ilu0_refactor(jo_ilu_par *P, const double *nz) {
#pragma omp parallel for schedule(dynamic, 1)
  for (I64 b = 0; b < P->nblocks; ++b) ilu_refactor(P->f[b], nz);
  return 0;
}

/* invert_row! (ilu0.jl:156-166), forward/backward_substitute! (:168-181), ilu_solve! (:184-187) */
static void ilu_solve_one(const jo_ilu *F, double *x, const double *b) {
  int N = F->bs, NN = N * N;
  for (I64 a = 0; a < F->nact; ++a) { /* forward: unit diagonal, D = nothing */
    I64 row = F->active[a];
    if (N == 1) {
      double v = b[row - 1];
      for (I64 j = F->l_rowptr[a]; j <= F->l_rowptr[a + 1] - 1; ++j) v -= F->l_val[j - 1] * x[F->l_col[j - 1] - 1];
      x[row - 1] = v;
    } else {
      double v[3];
      for (int e = 0; e < N; ++e) v[e] = b[(row - 1) * N + e];
      for (I64 j = F->l_rowptr[a]; j <= F->l_rowptr[a + 1] - 1; ++j) {
        const double *A = F->l_val + (size_t)(j - 1) * NN;
        const double *xk = x + (size_t)(F->l_col[j - 1] - 1) * N;
        for (int e = 0; e < N; ++e) { double s = 0; for (int d = 0; d < N; ++d) s += A[d * N + e] * xk[d]; v[e] -= s; }
      }
      for (int e = 0; e < N; ++e) x[(row - 1) * N + e] = v[e];
    }
  }
  for (I64 a = F->nact - 1; a >= 0; --a) { /* backward: x = D*(x - U x) */
    I64 row = F->active[a];
    if (N == 1) {
      double v = x[row - 1];
      for (I64 j = F->u_rowptr[a]; j <= F->u_rowptr[a + 1] - 1; ++j) v -= F->u_val[j - 1] * x[F->u_col[j - 1] - 1];
      x[row - 1] = F->d_val[a] * v;
    } else {
      double v[3], o[3];
      for (int e = 0; e < N; ++e) v[e] = x[(row - 1) * N + e];
      for (I64 j = F->u_rowptr[a]; j <= F->u_rowptr[a + 1] - 1; ++j) {
        const double *A = F->u_val + (size_t)(j - 1) * NN;
        const double *xk = x + (size_t)(F->u_col[j - 1] - 1) * N;
        for (int e = 0; e < N; ++e) { double s = 0; for (int d = 0; d < N; ++d) s += A[d * N + e] * xk[d]; v[e] -= s; }
      }
      const double *D = F->d_val + (size_t)a * NN;
      for (int e = 0; e < N; ++e) { double s = 0; for (int d = 0; d < N; ++d) s += D[d * N + e] * v[d]; o[e] = s; }
      for (int e = 0; e < N; ++e) x[(row - 1) * N + e] = o[e];
    }
  }
}

/* ldiv!(x, LU, b) (ilu0.jl:233-236; par_ilu0.jl:82-88: blocks write disjoint rows of x) */
int jo_ilu0_apply(const jo_ilu_par *P, double *x, const double *b) {
#pragma omp parallel for schedule(dynamic, 1)
  for (I64 blk = 0; blk < P->nblocks; ++blk) ilu_solve_one(P->f[blk], x, b);
  return 0;
}

void jo_ilu0_free(jo_ilu_par *P) {
  if (!P) return;
  for (I64 b = 0; b < P->nblocks; ++b) ilu_free_one(P->f[b]);
  free(P->f); free(P->loc); free(P->owner); free(P);
}

/* Export factor values scattered back to A's pattern (for parity tests): lu[k] holds L (multipliers)
 * below the diagonal, inv(U_ii) on the diagonal, U above; entries outside every block are left as is. */
int jo_ilu0_export(const jo_ilu_par *P, double *lu) {
  int NN = P->bs * P->bs;
  for (I64 b = 0; b < P->nblocks; ++b) {
    const jo_ilu *F = P->f[b];
    I64 nl = F->l_rowptr[F->nact] - 1, nu = F->u_rowptr[F->nact] - 1;
    for (I64 i = 0; i < nl; ++i) memcpy(lu + (F->l_map[i] - 1) * NN, F->l_val + i * NN, sizeof(double) * NN);
    for (I64 i = 0; i < nu; ++i) memcpy(lu + (F->u_map[i] - 1) * NN, F->u_val + i * NN, sizeof(double) * NN);
    for (I64 i = 0; i < F->nact; ++i) memcpy(lu + (F->d_map[i] - 1) * NN, F->d_val + i * NN, sizeof(double) * NN);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.7 BiCGStab (Krylov.jl 0.9 `bicgstab!`, third-party, restated from its published algorithm; */
/*     call site src/linsolve/krylov.jl:71-182, workspace ext/.../krylov.jl:107-124)          */
/* ------------------------------------------------------------------------------------------ */

static double dotp(I64 n, const double *a, const double *b) {
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (I64 i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* side: 0 none, 1 left (M = prec), 2 right (N = prec; Jutul single-process default, linsolve/utils.jl:25).
 * History: ||r_k|| for k = 0..iters (length iters+1).  Returns: 0 solved, 1 itmax, 2 breakdown, -1 workspace allocation failed.
 * x0 = 0, c = r0 (shadow), stop on ||r|| <= atol + rtol*||r0||. */
/* Workspace of the Krylov solvers: ten vectors kept between solves (Krylov.jl's BicgstabSolver workspace is allocated once,
 * linsolve/krylov.jl:27-58), pages first touched by the threads that work on them. */
/* (per calling thread: ctypes releases the GIL during a call, two Python threads may solve at the same time) */
static __thread double *ws_buf = NULL;
static __thread I64 ws_len = 0;
static double *ws_get(I64 m) {
  if (ws_len != m) {
    free(ws_buf);
    ws_buf = (double *)malloc((size_t)(10 * m) * sizeof(double));
    ws_len = ws_buf ? m : 0;
    if (ws_buf)
      for (int k = 0; k < 10; ++k) {
        double *w = ws_buf + (size_t)k * m;
#pragma omp parallel for schedule(static)
        for (I64 i = 0; i < m; ++i) w[i] = 0.0;
      }
  }
  return ws_buf;
}
static void vcopy(I64 m, double *dst, const double *src) {
#pragma omp parallel for schedule(static)
  for (I64 i = 0; i < m; ++i) dst[i] = src[i];
}
/* y = a*x + b*y (the solver's kaxpy! / kaxpby! calls: BLAS level 1 on all cores, as OpenBLAS runs them for Jutul) */
static void vaxpby(I64 m, double a, const double *x, double b, double *y) {
#pragma omp parallel for schedule(static)
  for (I64 i = 0; i < m; ++i) y[i] = a * x[i] + b * y[i];
}

int jo_bicgstab(I64 n, int bs, const I64 *rowptr, const I64 *colidx, const double *nz, const jo_ilu_par *P,
                int side, const double *b, double *x, double rtol, double atol, I64 itmax, I64 *iters_out,
                double *hist, I64 hist_cap) {
  I64 m = n * bs;
  double *ws = ws_get(m);
  if (!ws) return -1;
  double *r = ws, *p = ws + m, *v = ws + 2 * m, *s = ws + 3 * m, *q = ws + 4 * m, *y = ws + 5 * m, *z = ws + 6 * m, *t = ws + 7 * m,
         *d = ws + 8 * m, *c = ws + 9 * m;
  int left = (side == 1 && P), right = (side == 2 && P);
#pragma omp parallel for schedule(static)
  for (I64 i = 0; i < m; ++i) x[i] = 0.0;
  if (left) jo_ilu0_apply(P, r, b); else vcopy(m, r, b); /* r0 = M^-1 b */
  vcopy(m, p, r);
  vcopy(m, c, r); /* c = r0 */
  double rho = dotp(m, c, r);
  double rnorm = sqrt(dotp(m, r, r));
  double eps = atol + rtol * rnorm;
  I64 it = 0;
  if (hist && hist_cap > 0) hist[0] = rnorm;
  int status = 0;
  if (rho == 0.0 && rnorm > 0) status = 2;
  int solved = rnorm <= eps;
  while (!solved && it < itmax && status == 0) {
    ++it;
    const double *yy = p;
    if (right) { jo_ilu0_apply(P, y, p); yy = y; }
    jo_spmv(n, bs, rowptr, colidx, nz, yy, q, 1.0, 0.0);
    const double *vv = q;
    if (left) { jo_ilu0_apply(P, v, q); vv = v; } else { vcopy(m, v, q); vv = v; }
    double alpha = rho / dotp(m, c, vv);
    vcopy(m, s, r);
    vaxpby(m, -alpha, vv, 1.0, s);  /* s = r - alpha*v */
    vaxpby(m, alpha, yy, 1.0, x);   /* x = x + alpha*y */
    const double *zz = s;
    if (right) { jo_ilu0_apply(P, z, s); zz = z; }
    jo_spmv(n, bs, rowptr, colidx, nz, zz, d, 1.0, 0.0);
    const double *tt = d;
    if (left) { jo_ilu0_apply(P, t, d); tt = t; }
    double ttn = dotp(m, tt, tt);
    /* 0/0 guard for an exact first half-step (s == 0): Krylov.jl's formula would give NaN */
    double omega = (ttn == 0.0) ? 0.0 : dotp(m, tt, s) / ttn;
    vaxpby(m, omega, zz, 1.0, x);   /* x = x + omega*z */
    vcopy(m, r, s);
    vaxpby(m, -omega, tt, 1.0, r);  /* r = s - omega*t */
    double rho_next = dotp(m, c, r);
    double beta = (rho_next / rho) * (alpha / omega);
    vaxpby(m, -omega, vv, 1.0, p);  /* p = p - omega*v */
    vaxpby(m, 1.0, r, beta, p);     /* p = r + beta*p */
    rho = rho_next;
    rnorm = sqrt(dotp(m, r, r));
    if (hist && it < hist_cap) hist[it] = rnorm;
    solved = rnorm <= eps;
    if (alpha == 0.0 || alpha != alpha) status = 2;
  }
  if (!solved && status == 0 && it >= itmax) status = 1;
  if (solved) status = 0;
  *iters_out = it;
  return status;
}

/* dst = src on all cores: used to hand the solver arrays whose pages are spread over the NUMA nodes like the loops that read
 * them (first touch), the way Julia's threaded initialisation would */
int jo_zero(I64 n8, double *dst) {
#pragma omp parallel for schedule(static)
  for (I64 i = 0; i < n8; ++i) dst[i] = 0.0;
  return 0;
}
int jo_touch_copy(I64 nbytes8, double *dst, const double *src) {
#pragma omp parallel for schedule(static)
  for (I64 i = 0; i < nbytes8; ++i) dst[i] = src[i];
  return 0;
}

/* GMRES (Krylov.jl 0.9 `gmres!`, third-party, restated from its published algorithm: MGS Arnoldi + Givens rotations,
 * restart = false so the basis grows with the iterations, x0 = 0; call site src/linsolve/krylov.jl:214-218).
 * Same conventions as jo_bicgstab.  "Parity unpinned" per iterate. */
int jo_gmres(I64 n, int bs, const I64 *rowptr, const I64 *colidx, const double *nz, const jo_ilu_par *P, int side,
             const double *b, double *x, double rtol, double atol, I64 itmax, I64 *iters_out, double *hist, I64 hist_cap) {
  I64 m = n * bs;
  int left = (side == 1 && P), right = (side == 2 && P);
  double *V = (double *)calloc((size_t)m * (size_t)(itmax + 1), sizeof(double));
  double *w = (double *)calloc((size_t)m, sizeof(double)), *tmp = (double *)calloc((size_t)m, sizeof(double));
  double *R = (double *)calloc((size_t)(itmax + 1) * (size_t)(itmax + 1), sizeof(double));
  double *cs = (double *)calloc((size_t)itmax + 1, sizeof(double)), *sn = (double *)calloc((size_t)itmax + 1, sizeof(double));
  double *z = (double *)calloc((size_t)itmax + 2, sizeof(double)), *h = (double *)calloc((size_t)itmax + 2, sizeof(double));
  for (I64 i = 0; i < m; ++i) x[i] = 0.0;
  if (left) jo_ilu0_apply(P, w, b); else memcpy(w, b, sizeof(double) * m);
  double beta = sqrt(dotp(m, w, w));
  double eps = atol + rtol * beta;
  if (hist && hist_cap > 0) hist[0] = beta;
  I64 it = 0, ld = itmax + 1;
  int status = 0, solved = beta <= eps;
  z[0] = beta;
  if (!solved) for (I64 i = 0; i < m; ++i) V[i] = w[i] / beta;
  while (!solved && it < itmax && status == 0) {
    I64 k = it++;
    const double *vk = V + (size_t)k * m;
    const double *in = vk;
    if (right) { jo_ilu0_apply(P, tmp, vk); in = tmp; }
    jo_spmv(n, bs, rowptr, colidx, nz, in, w, 1.0, 0.0);
    if (left) { jo_ilu0_apply(P, tmp, w); memcpy(w, tmp, sizeof(double) * m); }
    for (I64 i = 0; i <= k; ++i) {
      const double *vi = V + (size_t)i * m;
      h[i] = dotp(m, vi, w);
      for (I64 l = 0; l < m; ++l) w[l] -= h[i] * vi[l];
    }
    double hbis = sqrt(dotp(m, w, w));
    for (I64 i = 0; i < k; ++i) {
      double a = cs[i] * h[i] + sn[i] * h[i + 1];
      h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
      h[i] = a;
    }
    double rr = hypot(h[k], hbis);
    cs[k] = rr == 0.0 ? 1.0 : h[k] / rr;
    sn[k] = rr == 0.0 ? 0.0 : hbis / rr;
    h[k] = rr;
    for (I64 i = 0; i <= k; ++i) R[(size_t)k * ld + i] = h[i];
    z[k + 1] = -sn[k] * z[k];
    z[k] = cs[k] * z[k];
    double rnorm = fabs(z[k + 1]);
    if (hist && it < hist_cap) hist[it] = rnorm;
    solved = rnorm <= eps;
    if (!solved) {
      if (hbis == 0.0) { status = 2; break; }
      double *vn = V + (size_t)(k + 1) * m;
      for (I64 l = 0; l < m; ++l) vn[l] = w[l] / hbis;
    }
  }
  I64 mm = it;
  if (status == 2) mm = it;  /* the last column was completed before the breakdown test */
  for (I64 j = mm - 1; j >= 0; --j) {
    double acc = z[j];
    for (I64 l = j + 1; l < mm; ++l) acc -= R[(size_t)l * ld + j] * h[l];
    h[j] = acc / R[(size_t)j * ld + j];  /* h reused as y */
  }
  for (I64 i = 0; i < m; ++i) tmp[i] = 0.0;
  for (I64 j = 0; j < mm; ++j) {
    const double *vj = V + (size_t)j * m;
    for (I64 l = 0; l < m; ++l) tmp[l] = h[j] * vj[l] + tmp[l];
  }
  if (right) jo_ilu0_apply(P, x, tmp); else memcpy(x, tmp, sizeof(double) * m);
  if (solved) status = 0; else if (status == 0 && it >= itmax) status = 1;
  *iters_out = it;
  free(V); free(w); free(tmp); free(R); free(cs); free(sn); free(z); free(h);
  return status;
}

/* ------------------------------------------------------------------------------------------ */
/* Partition helpers (src/partitioning.jl)                                                    */
/* ------------------------------------------------------------------------------------------ */

/* partition_linear (partitioning.jl:12-18): p[i] = ceil(i/(n/m)) */
int jo_partition_linear(I64 m, I64 n, I64 *p) {
  for (I64 i = 1; i <= n; ++i) p[i - 1] = (I64)ceil((double)i / ((double)n / (double)m));
  return 0;
}

/* compress_partition (partitioning.jl:92-99) */
int jo_compress_partition(I64 n, const I64 *p, I64 *out) {
  I64 *up = (I64 *)malloc((size_t)(n > 0 ? n : 1) * sizeof(I64));
  memcpy(up, p, (size_t)n * sizeof(I64));
  qsort(up, (size_t)n, sizeof(I64), cmp_i64);
  I64 m = 0;
  for (I64 i = 0; i < n; ++i) if (i == 0 || up[i] != up[i - 1]) up[m++] = up[i];
  for (I64 i = 0; i < n; ++i) {
    I64 lo = 0, hi = m; /* searchsortedfirst */
    while (lo < hi) { I64 mid = (lo + hi) / 2; if (up[mid] < p[i]) lo = mid + 1; else hi = mid; }
    out[i] = lo + 1;
  }
  free(up);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A.9 distributed helpers (ext/JutulPartitionedArraysExt)                                    */
/* ------------------------------------------------------------------------------------------ */

/* partition_boundary (utils.jl:32-56): ghosts of part `ip` = cells outside sharing a face with an owned
 * cell, in face-scan order, unique!d.  Returns count; bnd must have room for nc entries. */
I64 jo_partition_boundary(const I64 *N, I64 nf, I64 nc, const I64 *p, I64 ip, I64 *bnd) {
  char *seen = (char *)calloc((size_t)nc, 1);
  I64 n = 0;
  for (I64 i = 0; i < nf; ++i) {
    I64 l = N[2 * i], r = N[2 * i + 1];
    int il = p[l - 1] == ip, ir = p[r - 1] == ip;
    if (il && !ir) { if (!seen[r - 1]) { seen[r - 1] = 1; bnd[n++] = r; } }
    else if (ir && !il) { if (!seen[l - 1]) { seen[l - 1] = 1; bnd[n++] = l; } }
  }
  free(seen);
  return n;
}

/* remap_global_indices, order = :default (utils.jl:9-30): rank i owns offset_i+1..offset_i+count_i in
 * the order of findall(p .== i). */
int jo_remap_global_indices(I64 nc, const I64 *p, I64 np, I64 *remapped, I64 *counts) {
  for (I64 i = 0; i < np; ++i) counts[i] = 0;
  for (I64 c = 0; c < nc; ++c) counts[p[c] - 1]++;
  I64 *off = (I64 *)calloc((size_t)np, sizeof(I64));
  for (I64 i = 1; i < np; ++i) off[i] = off[i - 1] + counts[i - 1];
  for (I64 c = 0; c < nc; ++c) remapped[c] = ++off[p[c] - 1];
  free(off);
  return 0;
}

/* unit_diagonalize!(r, J::StaticSparsityMatrixCSR, n_self) (linalg.jl:18-35): ghost rows -> -I, r -> 0 */
int jo_unit_diagonalize(I64 n, I64 n_self, int bs, const I64 *rowptr, const I64 *colidx, double *nz, double *r) {
  int NN = bs * bs;
  for (I64 i = n_self * bs; i < n * bs; ++i) r[i] = 0.0;
  for (I64 row = n_self + 1; row <= n; ++row)
    for (I64 k = rowptr[row - 1]; k <= rowptr[row] - 1; ++k) {
      for (int t = 0; t < NN; ++t) nz[(k - 1) * NN + t] = 0.0;
      if (colidx[k - 1] == row)
        for (int e = 0; e < bs; ++e) nz[(k - 1) * NN + e * bs + e] = -1.0; /* -one(SMatrix) = -I */
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* law construction helper for ctypes                                                         */
/* ------------------------------------------------------------------------------------------ */
jo_law *jo_law_create(int kind, double dt, const double *params /* rho0[2], comp[2], mu[2], p_ref */) {
  jo_law *L = (jo_law *)calloc(1, sizeof(jo_law));
  L->kind = kind; L->N = (kind == 2) ? 2 : 1; L->dt = dt;
  for (int i = 0; i < 2; ++i) { L->rho0[i] = params ? params[i] : 1.0; L->comp[i] = params ? params[2 + i] : 0.0; L->mu[i] = params ? params[4 + i] : 1.0; }
  L->p_ref = params ? params[6] : 0.0;
  return L;
}
void jo_law_set_dt(jo_law *L, double dt) { L->dt = dt; }
void jo_law_free(jo_law *L) { free(L); }
