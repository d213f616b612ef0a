// ILU(0) factor + apply (a-11, a-12) and block-Jacobi ILU(0) (a-13) for gfx950.
//
// Reference: ilu0_csr / ilu0_csr! / ldiv! (StaticCSR/ilu0.jl:108-236) are strictly serial over rows; the
// reference's only parallelism is block-Jacobi over a row partition, one block per thread
// (par_ilu0.jl:47-90, precond/ilu.jl:37-60).  The same two modes exist here:
//
//  * LDS mode (block-Jacobi, the performance path): each block of the partition is owned by ONE workgroup.
//    The block's slice of the vector lives in LDS for the whole solve; rows are processed level by level
//    (in-block dependency levels, computed once on the host since the pattern is static) with a
//    __syncthreads() between levels.  Storage is laid out in processing order: L rows in forward-level
//    order, U rows + inverted pivots in backward-level order, column ids block-local.  One launch per apply,
//    every factor byte is read once, coalesced per level.
//  * GLOBAL mode (serial-ILU(0) semantics, or blocks too large for LDS): rows sorted by global level,
//    one launch per level, vector in HBM.  Mathematically identical to the reference's serial
//    factorisation in the same ordering (ILU(0) only depends on the orientation of the pattern's edges,
//    and a level order is a topological order of that orientation).
//
// In both modes each row performs the reference's IKJ update sequence (ascending k, `L_ik = A_ik*inv(D_k)`,
// inverted pivots stored at the end), so factors agree with the oracle to rounding.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <numeric>

#include "jh_internal.hpp"

namespace jh {
double *const *halo_push_targets(jh_tpfa d);
}
using namespace jh;

struct jh_ilu_s {
  jh_context ctx = nullptr;
  jh_csr A = nullptr;
  std::shared_ptr<Pattern> pat;
  int bs = 1;
  int64_t n = 0;
  int kind = 0;        // 0 ILU(0); 1 damped Jacobi (w*inv(A_ii)); 2 SPAI(0) (A_ii / sum of squared row entries)
  double jacobi_w = 1.0;
  bool lds_mode = false;
  int64_t nparts = 1, max_block_rows = 0, max_levels = 0;
  int threads = 256;
  size_t lds_bytes = 0;
  int max_blk_l = 0, max_blk_u = 0;  // largest per-block strict-L / strict-U entry counts (LDS factor kernel)
  size_t factor_lds_bytes = 0;       // 0: block does not fit -> global-memory factor kernels
  // symbolic data (host)
  std::vector<int32_t> rowmap;            // ilu row -> device row of A
  std::vector<int32_t> blk_ptr;           // execution blocks (LDS mode: partition blocks; GLOBAL: {0, n})
  std::vector<int32_t> flev_off, flev_ptr; // per block: offsets into flev_ptr; flev_ptr: ilu row starts
  std::vector<int32_t> blev_off, blev_ptr; // backward levels over U-order positions
  std::vector<int32_t> l_ptr, l_col, l_map, u_ptr, u_col, u_map, d_map, u_row, upos_of;
  std::vector<uint16_t> l_lev, u_lev;  // in-block level of every ilu row (fwd order) / U-order position (bwd order)
  DevBuf<uint16_t> d_l_lev, d_u_lev;
  // 16-bit copies of the block-local metadata for the chunked apply (LDS mode: < 8192 rows per block)
  DevBuf<uint16_t> d_l_col16, d_u_col16, d_u_row16, d_rowmap16;
  bool rowmap_local = false;  // block b's rows are the device rows [blk_ptr[b], blk_ptr[b+1]): rowmap16[device row] = ilu position inside the block
  DevBuf<int32_t> d_rowmap, d_blk_ptr, d_flev_off, d_flev_ptr, d_blev_off, d_blev_ptr, d_l_ptr, d_l_col, d_l_map, d_u_ptr,
      d_u_col, d_u_map, d_d_map, d_u_row, d_upos_of;
  DevBuf<double> l_val, u_val, dinv, xg;
  // halo send rows per execution block (LDS mode, rank-local subdomain): the apply writes them into the send buffer too
  std::vector<int32_t> send_ptr, send_local, send_slot;
  DevBuf<int32_t> d_send_ptr, d_send_local, d_send_slot;
  uint64_t halo_epoch = 0;
  bool factored = false;
  // ---- chunk-jagged layout (the performance path of the LDS mode, see ilu_apply_jds_kernel) ----------------------------------
  bool jag = false;      // l_val / u_val / dinv are stored in chunk-jagged order
  int jag_ku = 4;        // diagonals held in registers per sweep (4 or 8)
  bool jag_vr = false;   // rows of more than 8 entries are chains of lanes (virtual rows)
  int64_t jl_nent = 0, ju_nent = 0, j_nslots = 0;  // jagged entry counts (+ padding), 64 * chunks
  std::vector<int32_t> chunk_ptr;                  // per block: first chunk; block b has ceil(rows/64) chunks per sweep
  std::vector<int32_t> jl_map, ju_map, jd_map;     // A slot of every jagged L / U entry and of every bwd lane's pivot (-1: no row)
  DevBuf<int32_t> d_chunk_ptr, d_jl_map, d_ju_map, d_jd_map;
  DevBuf<int4> d_jf_desc, d_jb_desc;               // per chunk: entry base, counts 0-3, counts 4-7, lvlo | lvhi << 16
  DevBuf<uint32_t> d_jf_row, d_jb_row;             // per chunk lane: block-local row | level << 16 (0xffff....: no row)
  DevBuf<uint16_t> d_jl_col, d_ju_col;             // block-local column of every jagged entry
  // the jagged position of every old-order L / U entry and pivot (factor kernels that still produce the row-major order)
  std::vector<int32_t> jl_of_old, ju_of_old, jd_of_old;
  DevBuf<int32_t> d_jl_of_old, d_ju_of_old, d_jd_of_old;
  DevBuf<double> jl_val, ju_val, jdinv;  // factor values in chunk-jagged order
  // pivot-only factorisation (ilu_factor_diag_kernel): no elimination step of any block updates an off-diagonal entry
  bool diag_only = false;
  int max_chunks = 0;                              // chunks of the largest block
  DevBuf<int32_t> d_jt_map, d_jf_diag;             // per jagged L entry (i,k): A slot of (k,i) or -1; per forward lane: A slot of its pivot
  DevBuf<uint16_t> d_jf_bslot;                     // per forward lane: the row's backward chunk lane inside the block
  // "D-ILU" storage of a pivot-only factorisation: there L = I + L_A inv(D~) and U = D~ + U_A hold A's own in-block entries, so
  // M = L U = (D~ + L_A) inv(D~) (D~ + U_A) and the sweeps can run on A's entries and the inverted pivots alone:
  //   forward   g^_i = inv(D~_i) (b_i - sum_{k<i} A_ik g^_k)        (g^ = inv(D~) g in LDS)
  //   backward  y_i  = g^_i - inv(D~_i) s_i,  s_i = sum_{k>i} A_ik y_k   (y in LDS, as in the LU form)
  // With A's entries in both triangles the in-block part of the product A*y that follows every apply of the right-preconditioned
  // Krylov loop costs one more, dependency-free pass over the L chunks (ilu_apply_jds_kernel<..., MUL>):
  //   (A y)_i = A_ii y_i + s_i + sum_{k<i} A_ik y_k + sum_{k outside the block} A_ik y_k
  //             backward lane  third pass             ilu_eprod_kernel ("E" entries)
  bool uscaled = false;                            // D-ILU storage in use (l_val / u_val = A's entries in jagged order)
  DevBuf<double> jkap;                             // per backward slot: A_ii
  DevBuf<double> jdinv_f;                          // inverted pivots once more, in forward-lane order
  DevBuf<uint16_t> d_jb_dev, d_jf_dev;             // per backward / forward slot: device row of the lane's row relative to the block
                                                   // (| 0x8000: the row has entries outside its block)
  int64_t e_rows = 0, e_nent = 0;                  // rows with out-of-block entries, and those entries
  DevBuf<int32_t> d_e_row, d_e_ptr, d_e_col, d_e_slot;  // device row; CSR-like pointers; device column and slot of A of every entry
  DevBuf<double> e_val;                            // their values, copied from A at the start of every solve (ilu_eprod_refresh)
  // program-driven factorisation (ilu_factor_prog_kernel): per block the entry ranges and a 16-bit instruction stream
  bool prog = false;
  bool prog_rowmajor = false;  // the programs address the row-major factor arrays (rows too long for the jagged layout)
  std::vector<int32_t> blk_lbase, blk_ubase, blk_prog;  // [nb + 1] first jagged L / U entry and first program word of every block
  std::vector<int32_t> blk_dbase;                       // [nb + 1] first pivot slot of every block
  DevBuf<int32_t> d_blk_lbase, d_blk_ubase, d_blk_prog, d_blk_dbase;
  DevBuf<uint16_t> d_prog;
  size_t prog_lds_bytes = 0;
  bool prog_rows = false;  // the program is in the rows form (ilu_factor_rows_kernel)
  int prog_max_vals = 0, prog_max_words = 0;
};

namespace {

// ---- small dense blocks (column-major BS x BS) ------------------------------------------------------------------
template <int BS> struct Blk { double a[BS * BS]; };

template <int BS> __device__ __forceinline__ Blk<BS> blk_load(const double *p) {
  Blk<BS> r;
#pragma unroll
  for (int i = 0; i < BS * BS; ++i) r.a[i] = p[i];
  return r;
}
template <int BS> __device__ __forceinline__ void blk_store(double *p, const Blk<BS> &b) {
#pragma unroll
  for (int i = 0; i < BS * BS; ++i) p[i] = b.a[i];
}
// 2x2 blocks are 32-byte aligned in every value array (hipMalloc'ed bases, 32-byte entries): two 16-byte accesses instead of four
template <int BS> __device__ __forceinline__ Blk<BS> blk_load_al(const double *p) {
  if (BS == 2) {
    Blk<BS> r;
    const double2 lo = reinterpret_cast<const double2 *>(p)[0], hi = reinterpret_cast<const double2 *>(p)[1];
    r.a[0] = lo.x; r.a[1] = lo.y; r.a[2 % (BS * BS)] = hi.x; r.a[3 % (BS * BS)] = hi.y;
    return r;
  }
  return blk_load<BS>(p);
}
template <int BS> __device__ __forceinline__ void blk_store_al(double *p, const Blk<BS> &b) {
  if (BS == 2) {
    reinterpret_cast<double2 *>(p)[0] = make_double2(b.a[0], b.a[1]);
    reinterpret_cast<double2 *>(p)[1] = make_double2(b.a[2 % (BS * BS)], b.a[3 % (BS * BS)]);
    return;
  }
  blk_store<BS>(p, b);
}
template <int BS> __device__ __forceinline__ Blk<BS> blk_mul(const Blk<BS> &A, const Blk<BS> &B) {
  Blk<BS> C;
#pragma unroll
  for (int j = 0; j < BS; ++j)
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < BS; ++k) s += A.a[k * BS + i] * B.a[j * BS + k];
      C.a[j * BS + i] = s;
    }
  return C;
}
template <int BS> __device__ __forceinline__ void blk_sub(Blk<BS> &A, const Blk<BS> &B) {
#pragma unroll
  for (int i = 0; i < BS * BS; ++i) A.a[i] -= B.a[i];
}
template <int BS> __device__ __forceinline__ bool blk_nonzero(const Blk<BS> &A) {
  bool nz = false;
#pragma unroll
  for (int i = 0; i < BS * BS; ++i) nz |= (A.a[i] != 0.0);
  return nz;
}
template <int BS> __device__ __forceinline__ Blk<BS> blk_inv(const Blk<BS> &A);
template <> __device__ __forceinline__ Blk<1> blk_inv<1>(const Blk<1> &A) {
  Blk<1> r; r.a[0] = 1.0 / A.a[0]; return r;
}
template <> __device__ __forceinline__ Blk<2> blk_inv<2>(const Blk<2> &A) {
  const double a = A.a[0], c = A.a[1], b = A.a[2], d = A.a[3];
  const double idet = 1.0 / (a * d - b * c);
  Blk<2> r; r.a[0] = d * idet; r.a[1] = -c * idet; r.a[2] = -b * idet; r.a[3] = a * idet;
  return r;
}
template <> __device__ __forceinline__ Blk<3> blk_inv<3>(const Blk<3> &A) {
  double m[3][3];
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) m[i][j] = A.a[j * 3 + i];
  const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  const double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  const double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  const double idet = 1.0 / (m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02);
  double inv[3][3];
  inv[0][0] = c00 * idet; inv[1][0] = c01 * idet; inv[2][0] = c02 * idet;
  inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * idet;
  inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * idet;
  inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * idet;
  inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * idet;
  inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * idet;
  inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * idet;
  Blk<3> r;
  for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) r.a[j * 3 + i] = inv[i][j];
  return r;
}

struct IluDev {
  const int32_t *rowmap, *blk_ptr, *flev_off, *flev_ptr, *blev_off, *blev_ptr;
  const int32_t *l_ptr, *l_col, *l_map, *u_ptr, *u_col, *u_map, *d_map, *u_row, *upos_of;
  const uint16_t *l_lev, *u_lev, *l_col16, *u_col16, *u_row16, *rowmap16;
  double *l_val, *u_val, *dinv;
  // fused halo pack (may be null): block b copies local rows send_local[j] to send_buf[send_slot[j]], j in [send_ptr[b], send_ptr[b+1])
  const int32_t *send_ptr, *send_local, *send_slot;
  double *send_buf;
  double *const *send_dst;  // push halo: per send slot the address in the neighbour's landing buffer (overrides send_buf)
  // chunk-jagged layout
  const int32_t *chunk_ptr;
  const int4 *jf_desc, *jb_desc;
  const uint32_t *jf_row, *jb_row;
  const uint16_t *jl_col, *ju_col;
  // D-ILU storage / fused product (jh_ilu_s::uscaled)
  double *kap, *dinv_f;
  const uint16_t *jb_dev, *jf_dev;
};

// ---- numeric factorisation of one row (ilu0_factor!, ilu0.jl:108-144) -----------------------------------------------
// t: ilu row (absolute), b0: first ilu row of its block.  All rows of earlier levels are final.
template <int BS>
__device__ __forceinline__ void factor_row(const IluDev &F, int t, int b0) {
  constexpr int BB = BS * BS;
  const int lt = t - b0;
  const int ipos = F.upos_of[t];
  const int ls = F.l_ptr[t], le = F.l_ptr[t + 1];
  if (ls == le) return;
  const int us = F.u_ptr[ipos], ue = F.u_ptr[ipos + 1];
  Blk<BS> dii = blk_load<BS>(F.dinv + (size_t)ipos * BB);
  for (int p = ls; p < le; ++p) {
    const int k = F.l_col[p];
    const int kpos = F.upos_of[b0 + k];
    const Blk<BS> piv = blk_load<BS>(F.dinv + (size_t)kpos * BB);
    const Blk<BS> lik = blk_mul<BS>(blk_load<BS>(F.l_val + (size_t)p * BB), blk_inv<BS>(piv));  // nz_l*inv(A_kk)
    blk_store<BS>(F.l_val + (size_t)p * BB, lik);
    if (!blk_nonzero<BS>(lik)) continue;
    const int ks = F.u_ptr[kpos], ke = F.u_ptr[kpos + 1];
    for (int p2 = p + 1; p2 < le; ++p2) {  // remaining L entries of row i
      const int j = F.l_col[p2];
      for (int q = ks; q < ke; ++q)
        if (F.u_col[q] == j) {
          Blk<BS> v = blk_load<BS>(F.l_val + (size_t)p2 * BB);
          blk_sub<BS>(v, blk_mul<BS>(lik, blk_load<BS>(F.u_val + (size_t)q * BB)));
          blk_store<BS>(F.l_val + (size_t)p2 * BB, v);
          break;
        }
    }
    for (int q = ks; q < ke; ++q)  // D[i] -= A_ik*U[k,i]
      if (F.u_col[q] == lt) { blk_sub<BS>(dii, blk_mul<BS>(lik, blk_load<BS>(F.u_val + (size_t)q * BB))); break; }
    for (int qi = us; qi < ue; ++qi) {  // U part of row i
      const int j = F.u_col[qi];
      for (int q = ks; q < ke; ++q)
        if (F.u_col[q] == j) {
          Blk<BS> v = blk_load<BS>(F.u_val + (size_t)qi * BB);
          blk_sub<BS>(v, blk_mul<BS>(lik, blk_load<BS>(F.u_val + (size_t)q * BB)));
          blk_store<BS>(F.u_val + (size_t)qi * BB, v);
          break;
        }
    }
  }
  blk_store<BS>(F.dinv + (size_t)ipos * BB, dii);
}

// gather A's values through the maps (update_values!, ilu0.jl:83-98)
__global__ void ilu_load_kernel(double *dst, const double *aval, const int32_t *map, int64_t cnt, int bb) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cnt * bb; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = i / bb;
    int e = (int)(i - k * bb);
    dst[i] = aval[(size_t)map[k] * bb + e];
  }
}
template <int BS>
__global__ void ilu_invert_kernel(double *dinv, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  blk_store<BS>(dinv + (size_t)i * BS * BS, blk_inv<BS>(blk_load<BS>(dinv + (size_t)i * BS * BS)));
}

// LDS-resident numeric factorisation: the whole block (values gathered straight from A through the maps, plus
// 16-bit block-local metadata) is staged in LDS, factorised level by level there (every hop of the dependent
// lookup chain l_col -> upos -> u_ptr -> u_col -> u_val costs an LDS access instead of an L2/HBM access), and
// written back once, coalesced.  Same IKJ arithmetic as factor_row above.
// dv holds A_ii for rows that are not finished yet and inv(D_ii) for finished ones (every row inverts its pivot when it is
// done, so the elimination multiplies by the stored inverse instead of dividing once per L entry; same arithmetic:
// nz_l * inv(A_kk)).
template <int BS>
__device__ __forceinline__ void factor_row_lds(int lt, double *lv, double *uv, double *dv, const uint16_t *lc, const uint16_t *uc,
                                               const uint16_t *lp, const uint16_t *up, const uint16_t *upos) {
  constexpr int BB = BS * BS;
  const int ipos = upos[lt];
  const int ls = lp[lt], le = lp[lt + 1];
  Blk<BS> dii = blk_load<BS>(dv + (size_t)ipos * BB);
  if (ls != le) {
    const int us = up[ipos], ue = up[ipos + 1];
    for (int p = ls; p < le; ++p) {
      const int k = lc[p];
      const int kpos = upos[k];
      const Blk<BS> lik = blk_mul<BS>(blk_load<BS>(lv + (size_t)p * BB), blk_load<BS>(dv + (size_t)kpos * BB));
      blk_store<BS>(lv + (size_t)p * BB, lik);
      if (!blk_nonzero<BS>(lik)) continue;
      const int ks = up[kpos], ke = up[kpos + 1];
      for (int p2 = p + 1; p2 < le; ++p2) {
        const int j = lc[p2];
        for (int q = ks; q < ke; ++q)
          if (uc[q] == j) {
            Blk<BS> v = blk_load<BS>(lv + (size_t)p2 * BB);
            blk_sub<BS>(v, blk_mul<BS>(lik, blk_load<BS>(uv + (size_t)q * BB)));
            blk_store<BS>(lv + (size_t)p2 * BB, v);
            break;
          }
      }
      for (int q = ks; q < ke; ++q)
        if (uc[q] == lt) { blk_sub<BS>(dii, blk_mul<BS>(lik, blk_load<BS>(uv + (size_t)q * BB))); break; }
      for (int qi = us; qi < ue; ++qi) {
        const int j = uc[qi];
        for (int q = ks; q < ke; ++q)
          if (uc[q] == j) {
            Blk<BS> v = blk_load<BS>(uv + (size_t)qi * BB);
            blk_sub<BS>(v, blk_mul<BS>(lik, blk_load<BS>(uv + (size_t)q * BB)));
            blk_store<BS>(uv + (size_t)qi * BB, v);
            break;
          }
      }
    }
  }
  blk_store<BS>(dv + (size_t)ipos * BB, blk_inv<BS>(dii));
}

template <int BS>
__global__ void ilu_factor_lds_kernel(IluDev F, const double *__restrict__ aval, int maxnl, int maxnu, int maxnr, int maxlev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BB = BS * BS;
  double *lv = reinterpret_cast<double *>(smem);
  double *uv = lv + (size_t)maxnl * BB;
  double *dv = uv + (size_t)maxnu * BB;
  uint16_t *lc = reinterpret_cast<uint16_t *>(dv + (size_t)maxnr * BB);
  uint16_t *uc = lc + maxnl;
  uint16_t *lp = uc + maxnu;
  uint16_t *up = lp + (maxnr + 1);
  uint16_t *upos = up + (maxnr + 1);
  uint16_t *levp = upos + maxnr;  // level starts (block-local rows) + end sentinel: no global load inside the level loop
  (void)maxlev;
  const int b = blockIdx.x;
  const int b0 = F.blk_ptr[b], b1 = F.blk_ptr[b + 1];
  const int nr = b1 - b0;
  const int lev0 = F.flev_off[b], nlev = F.flev_off[b + 1] - 1 - lev0;
  for (int l = threadIdx.x; l <= nlev; l += blockDim.x) levp[l] = (uint16_t)(F.flev_ptr[lev0 + l] - b0);
  const int l0 = F.l_ptr[b0], nl = F.l_ptr[b1] - l0;
  const int u0 = F.u_ptr[b0], nu = F.u_ptr[b1] - u0;
  const int T = blockDim.x, tid = threadIdx.x;
  for (int j = tid; j < nl; j += T) {
    lc[j] = (uint16_t)F.l_col[l0 + j];
    const double *src = aval + (size_t)F.l_map[l0 + j] * BB;
#pragma unroll
    for (int e = 0; e < BB; ++e) lv[(size_t)j * BB + e] = src[e];
  }
  for (int j = tid; j < nu; j += T) {
    uc[j] = (uint16_t)F.u_col[u0 + j];
    const double *src = aval + (size_t)F.u_map[u0 + j] * BB;
#pragma unroll
    for (int e = 0; e < BB; ++e) uv[(size_t)j * BB + e] = src[e];
  }
  for (int t = tid; t <= nr; t += T) {
    lp[t] = (uint16_t)(F.l_ptr[b0 + t] - l0);
    up[t] = (uint16_t)(F.u_ptr[b0 + t] - u0);
    if (t < nr) {
      upos[t] = (uint16_t)(F.upos_of[b0 + t] - b0);
      const double *src = aval + (size_t)F.d_map[b0 + t] * BB;
#pragma unroll
      for (int e = 0; e < BB; ++e) dv[(size_t)t * BB + e] = src[e];
    }
  }
  __syncthreads();
  for (int lev = 0; lev < nlev; ++lev) {  // level 0 rows have no L entries: they only invert their pivot
    const int s = levp[lev], e = levp[lev + 1];
    for (int t = s + tid; t < e; t += T) factor_row_lds<BS>(t, lv, uv, dv, lc, uc, lp, up, upos);
    __syncthreads();
  }
  for (int j = tid; j < nl * BB; j += T) F.l_val[(size_t)l0 * BB + j] = lv[j];
  for (int j = tid; j < nu * BB; j += T) F.u_val[(size_t)u0 * BB + j] = uv[j];
  for (int j = tid; j < nr * BB; j += T) F.dinv[(size_t)b0 * BB + j] = dv[j];  // already inverted
}

// ---- program-driven numeric factorisation into the chunk-jagged layout ------------------------------------------------------
// ilu_factor_lds_kernel above finds every update pair of the IKJ elimination (ilu0.jl:108-144) by searching U rows in LDS --
// counters (profiles/r02a): 87% of the wave cycles are waits, 165M scalar + 97M vector instructions per launch, 1.13 ms.  The
// pattern is static, so the searches are done ONCE on the host: every row gets a short program over a unified block-local
// value array  vals = [ L entries | U entries | pivots ]  (jagged order, so the final store is contiguous):
//     per L entry (ascending column k):  lidx, kd, nupd, nupd x (tgt, src)
//         l = vals[lidx] * vals[kd]       (kd: inverted pivot of row k);   vals[lidx] = l
//         vals[tgt] -= l * vals[src]      (src: U entry (k, j); tgt: this row's entry / pivot in column j)
//     then the row's pivot is inverted.
// One workgroup per block: values gathered from A through the jagged maps, programs copied to LDS, rows of one dependency
// level in parallel, barrier, next level; L, U and inverted pivots leave in the order the apply reads them.
// one-byte diagonal count J of a chunk descriptor (chunk-jagged layout, see ilu_apply_jds_kernel)
template <int J>
__device__ __forceinline__ int jd_count(const int4 &d) {
  const unsigned w = J < 4 ? (unsigned)d.y : (unsigned)d.z;
  return (int)((w >> (8 * (J & 3))) & 0xffu);
}
// A chunk descriptor through the scalar cache: the kernels also store through global pointers, so the compiler cannot prove that a
// descriptor is not clobbered and would fetch it with a vector load -- whose result the wavefront then has to wait for before it
// can branch on the counts (one exposed memory latency per chunk step).  Constant address space: s_load_dwordx4.
__device__ __forceinline__ int4 jds_desc(const int4 *desc, int i) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef v4i __attribute__((address_space(4))) const *cptr;
  const v4i d = *((cptr)(unsigned long long)desc + i);
  return make_int4(d.x, d.y, d.z, d.w);
}
// Orders the LDS traffic of a one-wavefront workgroup: the LDS operations of a wavefront execute in order, so only the compiler
// has to be kept from moving them.  (__syncthreads() here drains vmcnt as well: every level step would wait for the prefetched
// entries of the NEXT chunk.)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ILU(0) refactorisation when no elimination step updates an off-diagonal entry -- the pattern inside every block is
// triangle-free: Cartesian and corner-point grids, the tet lattice.  Then U keeps A's entries, L_ik = A_ik inv(u_kk), and only
// the pivots u_ii = A_ii - sum_k L_ik A_ki follow the elimination order (ilu0.jl:108-144 with every process_partial_row! update
// landing on the diagonal): the recurrence of a forward triangular sweep.  One thread per row, addressed like the apply's
// forward sweep (chunk, lane): its L entries, their partners A_ki and its own diagonal are loaded up front (coalesced through the
// chunk-jagged maps), the inverted pivots live in LDS by block-local row, the levels are walked with one barrier each.  Same
// operations in the same order as ilu_factor_prog_kernel, hence the same bits.
template <int BS, int KU, bool SC>
__global__ __launch_bounds__(1024) void ilu_factor_diag_kernel(IluDev F, const double *__restrict__ aval, const int32_t *__restrict__ ubase,
                                                               const int32_t *__restrict__ jl_map, const int32_t *__restrict__ jt_map,
                                                               const int32_t *__restrict__ ju_map, const int32_t *__restrict__ jf_diag,
                                                               const uint16_t *__restrict__ jf_bslot) {
  extern __shared__ __attribute__((aligned(16))) double dv[];  // inverted pivots by block-local row
  constexpr int BB = BS * BS;
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  const int c0 = F.chunk_ptr[b], nch = F.chunk_ptr[b + 1] - c0;
  const int ch = tid >> 6;
  const bool has_chunk = ch < nch;
  // this thread's row
  int4 D = make_int4(0, 0, 0, 0);
  unsigned word = 0xffff0000u;
  if (has_chunk) { D = F.jf_desc[c0 + ch]; word = F.jf_row[(size_t)(c0 + ch) * 64 + lane]; }
  const int lt = (int)(word & 0xffffu), lev = (int)(word >> 16);
  const bool has_row = lev != 0xffff;
  const size_t fslot = (size_t)(c0 + ch) * 64 + lane;
  Blk<BS> acc, aii;
  int bslot = 0;
  if (has_row) { acc = blk_load_al<BS>(aval + (size_t)jf_diag[fslot] * BB); bslot = (int)jf_bslot[fslot]; }
  if (SC) aii = acc;
  int kcol[KU], pos[KU];
  bool act[KU];
  Blk<BS> av[KU], bv[KU];
  {
    int off = D.x;
#define JH_FD(J)                                                                                       \
    if (J < KU) {                                                                                      \
      const int cnt = jd_count<(J < 8 ? J : 0)>(D);                                                    \
      act[J < KU ? J : 0] = has_row && lane < cnt;                                                     \
      pos[J < KU ? J : 0] = off + lane;                                                                \
      if (act[J < KU ? J : 0]) {                                                                       \
        kcol[J < KU ? J : 0] = (int)F.jl_col[off + lane];                                              \
        av[J < KU ? J : 0] = blk_load_al<BS>(aval + (size_t)jl_map[off + lane] * BB);                     \
        const int mt = jt_map[off + lane];                                                             \
        if (mt >= 0) bv[J < KU ? J : 0] = blk_load_al<BS>(aval + (size_t)mt * BB);                        \
        else { _Pragma("unroll") for (int i = 0; i < BB; ++i) bv[J < KU ? J : 0].a[i] = 0.0; }         \
      }                                                                                                \
      off += cnt;                                                                                      \
    }
    JH_FD(0) JH_FD(1) JH_FD(2) JH_FD(3) JH_FD(4) JH_FD(5) JH_FD(6) JH_FD(7)
#undef JH_FD
  }
  // U holds A's entries (jagged order), four at a time per thread; SC (D-ILU storage): so does L
  {
    const int u0 = ubase[b], nu = ubase[b + 1] - u0;
    for (int j0 = tid; j0 < nu; j0 += 4 * T) {
      int m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = j0 + u * T; m[u] = j < nu ? ju_map[u0 + j] : -1; }
      Blk<BS> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (m[u] >= 0) v[u] = blk_load_al<BS>(aval + (size_t)m[u] * BB);
        else { _Pragma("unroll") for (int e = 0; e < BB; ++e) v[u].a[e] = 0.0; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * T;
        if (j < nu) blk_store_al<BS>(F.u_val + (size_t)(u0 + j) * BB, v[u]);
      }
    }
  }
  if (SC) {
#pragma unroll
    for (int j = 0; j < KU; ++j)
      if (act[j]) blk_store_al<BS>(F.l_val + (size_t)pos[j] * BB, av[j]);
  }
  const int lev0 = F.flev_off[b], nlev = F.flev_off[b + 1] - 1 - lev0;
  for (int lv = 0; lv < nlev; ++lv) {
    if (has_row && lev == lv) {
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        if (act[j]) {
          const Blk<BS> lik = blk_mul<BS>(av[j], blk_load<BS>(dv + (size_t)kcol[j] * BB));  // nz_l * inv(A_kk)
          if (!SC) blk_store_al<BS>(F.l_val + (size_t)pos[j] * BB, lik);
          if (blk_nonzero<BS>(lik)) blk_sub<BS>(acc, blk_mul<BS>(lik, bv[j]));
        }
      }
      const Blk<BS> di = blk_inv<BS>(acc);
      blk_store<BS>(dv + (size_t)lt * BB, di);
      blk_store_al<BS>(F.dinv + ((size_t)c0 * 64 + bslot) * BB, di);
      if (SC) {
        blk_store_al<BS>(F.dinv_f + fslot * BB, di);
        blk_store_al<BS>(F.kap + ((size_t)c0 * 64 + bslot) * BB, aii);
      }
    }
    __syncthreads();
  }
}

// The same refactorisation with ONE WAVEFRONT per block, shaped like the forward sweep of ilu_apply_jds_kernel: the block is walked
// chunk by chunk (64 rows in level order); the operands of chunk c+1 -- descriptor, row words, the A values behind the L maps,
// their partners and the pivots -- are in flight while chunk c walks ITS levels (a few, not the block's 36-51) out of registers
// and LDS, and nothing waits for a whole block: a workgroup per block holds nine wavefronts at one barrier per level and, with
// 2x2 blocks (~120 VGPRs), leaves ONE workgroup resident per CU.  Here a CU holds as many blocks as it has wavefront slots.
// Arithmetic, its order and the stores are those of ilu_factor_diag_kernel, hence the same bits.
// Measured (profiles/README.md, round 3): 2x2 blocks 1.05 -> 0.53 ms, scalar rows of 7 entries (KU = 8) 0.62 -> 0.54 ms; scalar
// KU = 4: 0.41 vs 0.40 ms -- the chunk-by-chunk walk spreads the touches of an A line (the row's own L entries, its U entries as
// partners of later rows, the U copy) over the lifetime of the block, and with ~28 blocks in flight per CU the XCD's L2 no longer
// holds them: 2.16 GB of HBM traffic per launch instead of 1.13, i.e. the kernel runs at the HBM rate.  (Staging the block's
// contiguous A range in LDS first removes the re-reads but leaves five wavefronts per CU: 0.64 ms.)  Default: this kernel for
// blocks and KU = 8, the workgroup-per-block kernel for scalar KU = 4.
template <int BS, int KU>
struct FWRow {
  unsigned word;
  int bslot;
  Blk<BS> acc;
  int kcol[KU];
  bool act[KU];
  Blk<BS> av[KU], bv[KU];
};
template <int BS, int KU, bool SC, bool PF>
__global__ __launch_bounds__(64) void ilu_factor_wave_kernel(IluDev F, const double *__restrict__ aval, const int32_t *__restrict__ ubase,
                                                             const int32_t *__restrict__ jl_map, const int32_t *__restrict__ jt_map,
                                                             const int32_t *__restrict__ ju_map, const int32_t *__restrict__ jf_diag,
                                                             const uint16_t *__restrict__ jf_bslot) {
  extern __shared__ __attribute__((aligned(16))) double dv[];  // inverted pivots by block-local row
  constexpr int BB = BS * BS;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int c0 = F.chunk_ptr[b], nch = F.chunk_ptr[b + 1] - c0;
  auto aload = [&](int slot) { return blk_load_al<BS>(aval + (size_t)slot * BB); };  // A value behind a map entry
  auto load = [&](const int4 &D, int chunk, FWRow<BS, KU> &R) {
    const size_t fslot = (size_t)chunk * 64 + lane;
    R.word = F.jf_row[fslot];
    const bool has_row = (R.word >> 16) != 0xffffu;
    R.bslot = 0;
    if (has_row) { R.acc = aload(jf_diag[fslot]); R.bslot = (int)jf_bslot[fslot]; }
    int off = D.x;
#define JH_FW(J)                                                                                       \
    if (J < KU) {                                                                                      \
      const int cnt = jd_count<(J < 8 ? J : 0)>(D);                                                    \
      const bool a = has_row && lane < cnt;                                                            \
      R.act[J < KU ? J : 0] = a;                                                                       \
      if (a) {                                                                                         \
        R.kcol[J < KU ? J : 0] = (int)F.jl_col[off + lane];                                            \
        R.av[J < KU ? J : 0] = aload(jl_map[off + lane]);                                              \
        const int mt = jt_map[off + lane];                                                             \
        if (mt >= 0) R.bv[J < KU ? J : 0] = aload(mt);                                                 \
        else { _Pragma("unroll") for (int i = 0; i < BB; ++i) R.bv[J < KU ? J : 0].a[i] = 0.0; }       \
      }                                                                                                \
      off += cnt;                                                                                      \
    }
    JH_FW(0) JH_FW(1) JH_FW(2) JH_FW(3) JH_FW(4) JH_FW(5) JH_FW(6) JH_FW(7)
#undef JH_FW
  };
  const int4 *desc = F.jf_desc;
  int4 dc = jds_desc(desc, c0), dn = jds_desc(desc, c0 + 1), dnn;  // (the descriptor arrays are padded by four entries)
  FWRow<BS, KU> cur, nxt;
  load(dc, c0, cur);
  const int u0 = ubase[b], nu = ubase[b + 1] - u0;
  // U holds A's entries (jagged order of the backward sweep), four at a time per lane.  (Copying the entries of backward chunk
  // nch-1-c -- roughly the rows of forward chunk c -- inside the chunk loop instead: 0.606 vs 0.591 ms on 2x2 blocks, dropped.)
  auto ucopy = [&](int ustart, int uend) {
    for (int j0 = ustart + lane; j0 < uend; j0 += 4 * 64) {
      int m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = j0 + u * 64; m[u] = j < uend ? ju_map[j] : -1; }
      Blk<BS> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (m[u] >= 0) v[u] = aload(m[u]);
        else { _Pragma("unroll") for (int e = 0; e < BB; ++e) v[u].a[e] = 0.0; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * 64;
        if (j < uend) blk_store_al<BS>(F.u_val + (size_t)j * BB, v[u]);
      }
    }
  };
  for (int c = 0; c < nch; ++c) {
    dnn = jds_desc(desc, c0 + c + 2);
    if (PF && c + 1 < nch) load(dn, c0 + c + 1, nxt);
    const int lt = (int)(cur.word & 0xffffu), lev = (int)(cur.word >> 16);
    const bool has_row = lev != 0xffff;
    const int lvlo = dc.w & 0xffff, lvhi = (int)((unsigned)dc.w >> 16);
    int pos[KU];
    {
      int off = dc.x;
#define JH_FP(J) if (J < KU) { pos[J < KU ? J : 0] = off + lane; off += jd_count<(J < 8 ? J : 0)>(dc); }
      JH_FP(0) JH_FP(1) JH_FP(2) JH_FP(3) JH_FP(4) JH_FP(5) JH_FP(6) JH_FP(7)
#undef JH_FP
    }
    Blk<BS> aii;
    if (SC) {
      aii = cur.acc;
#pragma unroll
      for (int j = 0; j < KU; ++j)
        if (cur.act[j]) blk_store_al<BS>(F.l_val + (size_t)pos[j] * BB, cur.av[j]);
    }
    for (int lv = lvlo; lv <= lvhi; ++lv) {
      if (has_row && lev == lv) {
#pragma unroll
        for (int j = 0; j < KU; ++j) {
          if (cur.act[j]) {
            const Blk<BS> lik = blk_mul<BS>(cur.av[j], blk_load<BS>(dv + (size_t)cur.kcol[j] * BB));  // nz_l * inv(A_kk)
            if (!SC) blk_store_al<BS>(F.l_val + (size_t)pos[j] * BB, lik);
            if (blk_nonzero<BS>(lik)) blk_sub<BS>(cur.acc, blk_mul<BS>(lik, cur.bv[j]));
          }
        }
        const Blk<BS> di = blk_inv<BS>(cur.acc);
        blk_store<BS>(dv + (size_t)lt * BB, di);
        blk_store_al<BS>(F.dinv + ((size_t)c0 * 64 + cur.bslot) * BB, di);
        if (SC) {
          blk_store_al<BS>(F.dinv_f + ((size_t)(c0 + c) * 64 + lane) * BB, di);
          blk_store_al<BS>(F.kap + ((size_t)c0 * 64 + cur.bslot) * BB, aii);
        }
      }
      wave_lds_fence();  // one wavefront: orders this level's LDS writes before the next level's reads
    }
    dc = dn; dn = dnn;
    if (PF) cur = nxt;
    else if (c + 1 < nch) load(dc, c0 + c + 1, cur);
  }
  ucopy(u0, u0 + nu);
}

// Development build (-DJH_APPLY_TIMING, tools/apply_timing.py): shader-clock stamps of the phases of every block's wavefront.
#ifdef JH_APPLY_TIMING
__device__ unsigned long long jh_apply_t[8 * 65536];
#define JH_T(k) do { if (threadIdx.x == 0 && blockIdx.x < 65536) jh_apply_t[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define JH_T(k) do { } while (0)
#endif
#ifdef JH_APPLY_TIMING
#define JH_TV(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 65536) jh_apply_t[blockIdx.x * 8 + (k)] = (v); } while (0)
#else
#define JH_TV(k, v) do { } while (0)
#endif
// gather of a block's factor entries through a map into LDS, four entries per thread at a time: the map loads, then the dependent
// value loads, are in flight together (one entry per thread and iteration leaves the block waiting on two memory latencies per
// 128 entries; deeper than four does not help: measured)
template <int BS>
__device__ __forceinline__ void prog_gather(double *vals, const double *__restrict__ aval, const int32_t *__restrict__ map, int first,
                                            int count, int vbase, int tid, int T) {
  constexpr int BB = BS * BS;
  for (int j0 = tid; j0 < count; j0 += 4 * T) {
    int m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = j0 + u * T; m[u] = j < count ? map[first + j] : -1; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * T;
      if (j < count) {
#pragma unroll
        for (int e = 0; e < BB; ++e)  // (-2: the unit pivot of a lane that does not finish its row, see the layout's virtual rows)
          vals[(size_t)(vbase + j) * BB + e] = m[u] >= 0 ? aval[(size_t)m[u] * BB + e] : ((m[u] == -2 && e % (BS + 1) == 0) ? 1.0 : 0.0);
      }
    }
  }
}

// WPR (wave per row; long rows, i.e. the row-major programs of polyhedral cells): a block of ~160 rows of ~15 entries has ~90
// dependency levels of one or two rows each, so a thread per row leaves the elimination serial (3.1 ms at 2M cells: 0.3 ms per
// block of LDS round trips); the update pairs of ONE multiplier are independent, so a wavefront takes a row and its lanes the
// pairs.  Same operations per entry, same order of the multipliers, hence the same bits.
template <int BS, bool WPR = false>
__global__ void ilu_factor_prog_kernel(IluDev F, const double *__restrict__ aval, const int32_t *__restrict__ lbase,
                                       const int32_t *__restrict__ ubase, const int32_t *__restrict__ pbase,
                                       const int32_t *__restrict__ jl_map, const int32_t *__restrict__ ju_map,
                                       const int32_t *__restrict__ jd_map, const uint16_t *__restrict__ prog, int max_vals,
                                       const int32_t *__restrict__ dbase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BB = BS * BS;
  double *vals = reinterpret_cast<double *>(smem);
  uint16_t *pw = reinterpret_cast<uint16_t *>(vals + (size_t)max_vals * BB);
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int b0 = F.blk_ptr[b], nr = F.blk_ptr[b + 1] - b0;
  const int l0 = lbase[b], nl = lbase[b + 1] - l0;
  const int u0 = ubase[b], nu = ubase[b + 1] - u0;
  const int d0 = dbase[b], nd = dbase[b + 1] - d0;  // pivot slots: backward chunk lanes (jagged layout) or U-order positions (row-major)
  const int p0 = pbase[b], np = pbase[b + 1] - p0;
  auto gather = [&](const int32_t *map, int first, int count, int vbase) { prog_gather<BS>(vals, aval, map, first, count, vbase, tid, T); };
  gather(jl_map, l0, nl, 0);
  gather(ju_map, u0, nu, nl);
  gather(jd_map, d0, nd, nl + nu);
  for (int j = tid; j < np; j += T) pw[j] = prog[p0 + j];
  __syncthreads();
  // program layout: [row offsets (nr + 1)] [pivot index of every row (nr)] [instructions]
  const uint16_t *rptr = pw, *rdiag = pw + nr + 1, *code = pw + 2 * nr + 1;
  const int lev0 = F.flev_off[b], nlev = F.flev_off[b + 1] - 1 - lev0;
  for (int lev = 0; lev < nlev; ++lev) {
    const int s = F.flev_ptr[lev0 + lev] - b0, e = F.flev_ptr[lev0 + lev + 1] - b0;
    if (WPR) {
      const int lane = tid & 63, wv = tid >> 6, nwv = T >> 6;
      for (int t = s + wv; t < e; t += nwv) {  // wave-uniform
        int pc = rptr[t];
        const int pe = rptr[t + 1];
        while (pc < pe) {
          const int lidx = code[pc], kd = code[pc + 1], nupd = code[pc + 2];
          pc += 3;
          // (every lane forms the multiplier from the same LDS words; lane 0 stores it)
          const Blk<BS> lik = blk_mul<BS>(blk_load<BS>(vals + (size_t)lidx * BB), blk_load<BS>(vals + (size_t)kd * BB));  // nz_l * inv(A_kk)
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) blk_store<BS>(vals + (size_t)lidx * BB, lik);
          if (blk_nonzero<BS>(lik)) {
            for (int u = lane; u < nupd; u += 64) {  // distinct targets: the pairs of one multiplier are independent
              const int tgt = code[pc + 2 * u], src = code[pc + 2 * u + 1];
              Blk<BS> v = blk_load<BS>(vals + (size_t)tgt * BB);
              blk_sub<BS>(v, blk_mul<BS>(lik, blk_load<BS>(vals + (size_t)src * BB)));
              blk_store<BS>(vals + (size_t)tgt * BB, v);
            }
          }
          pc += 2 * nupd;
          // the next multiplier of this row may be one of the targets just updated (by another lane)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane == 0) {
          const int di = rdiag[t];
          blk_store<BS>(vals + (size_t)di * BB, blk_inv<BS>(blk_load<BS>(vals + (size_t)di * BB)));
        }
      }
      __syncthreads();
      continue;
    }
    for (int t = s + tid; t < e; t += T) {
      int pc = rptr[t];
      const int pe = rptr[t + 1];
      while (pc < pe) {
        const int lidx = code[pc], kd = code[pc + 1], nupd = code[pc + 2];
        pc += 3;
        const Blk<BS> lik = blk_mul<BS>(blk_load<BS>(vals + (size_t)lidx * BB), blk_load<BS>(vals + (size_t)kd * BB));  // nz_l * inv(A_kk)
        blk_store<BS>(vals + (size_t)lidx * BB, lik);
        if (blk_nonzero<BS>(lik)) {
          for (int u = 0; u < nupd; ++u) {
            const int tgt = code[pc + 2 * u], src = code[pc + 2 * u + 1];
            Blk<BS> v = blk_load<BS>(vals + (size_t)tgt * BB);
            blk_sub<BS>(v, blk_mul<BS>(lik, blk_load<BS>(vals + (size_t)src * BB)));
            blk_store<BS>(vals + (size_t)tgt * BB, v);
          }
        }
        pc += 2 * nupd;
      }
      const int di = rdiag[t];
      blk_store<BS>(vals + (size_t)di * BB, blk_inv<BS>(blk_load<BS>(vals + (size_t)di * BB)));
    }
    __syncthreads();
  }
  for (int j = tid; j < nl * BB; j += T) F.l_val[(size_t)l0 * BB + j] = vals[j];
  for (int j = tid; j < nu * BB; j += T) F.u_val[(size_t)u0 * BB + j] = vals[(size_t)nl * BB + j];
  for (int j = tid; j < nd * BB; j += T) F.dinv[(size_t)d0 * BB + j] = vals[(size_t)(nl + nu) * BB + j];
}

// ---- wave per row, the row in registers (long rows: the virtual-row layout of polyhedral cells) ----------------------------------
// The wave-per-row variant above walks a row's program one multiplier at a time: header -> multiplier and pivot -> pair codes ->
// target and source values -> store -> fence, five dependent LDS round trips and ~80 instructions per multiplier on the only
// wavefront that is busy in a level of one or two rows (measured with the shader clock: 4200 cycles per row, 1.70 ms at 2M
// polyhedral cells, 0.04 of the HBM roofline).  The "rows form" of the program (prog_rows_form) turns the row around: lane e
// keeps entry e of the row in a register (two registers for rows of 65..128 entries) for the whole elimination, and the program
// is a dense table [multiplier][entry] of source indices (0xffff: no update), read PROG_CH multipliers at a time -- the sources
// (U entries of rows of earlier levels) and pivot inverses are final by then, so a chunk's operands come in two LDS latencies
// and the steps themselves touch no memory: the multiplier is a lane broadcast (v_readlane) from the register of the lane that
// owns it, the update one multiply-subtract on the lanes whose table entry is set.  Same operations on every entry in the same
// order as the other program kernels, hence the same bits.
template <int BS> struct ProgRows { static constexpr int CH = BS == 1 ? 8 : (BS == 2 ? 4 : 2); };  // multipliers per chunk (registers)
constexpr int PROG_ROWS_ENT = 128;  // entries of a row that take part in its elimination: two registers per lane

__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// one row: NE registers per lane; gp: the block's program (global memory)
template <int BS, int NE>
__device__ __forceinline__ void prog_row_eliminate(double *vals, const uint16_t *__restrict__ gp, int o, int nent, int nsteps, int dpos, int lane) {
  constexpr int BB = BS * BS, CH = ProgRows<BS>::CH;
  const int hb = o + nent, tb = (hb + 2 * nsteps + 3) & ~3;
  uint16_t idx[NE];
  Blk<BS> v[NE];
#pragma unroll
  for (int r = 0; r < NE; ++r) idx[r] = lane + 64 * r < nent ? gp[o + lane + 64 * r] : (uint16_t)0xffffu;
#pragma unroll
  for (int r = 0; r < NE; ++r) {
#pragma unroll
    for (int i = 0; i < BB; ++i) v[r].a[i] = 0.0;
    if (idx[r] != 0xffffu) v[r] = blk_load<BS>(vals + (size_t)idx[r] * BB);
  }
  for (int c0 = 0; c0 < nsteps; c0 += CH) {
    // the chunk's static operands: lane q < CH holds the position and the pivot inverse of multiplier c0 + q, every lane the
    // source indices of its entries
    int hm = 0;
    Blk<BS> dk;
#pragma unroll
    for (int i = 0; i < BB; ++i) dk.a[i] = 0.0;
    uint32_t code[NE][CH / 2];
#pragma unroll
    for (int r = 0; r < NE; ++r) {
      const uint32_t *tp = reinterpret_cast<const uint32_t *>(gp + tb + ((size_t)(c0 / CH) * nent + lane + 64 * r) * CH);
#pragma unroll
      for (int q = 0; q < CH / 2; ++q) code[r][q] = lane + 64 * r < nent ? tp[q] : 0xffffffffu;
    }
    if (lane < CH && c0 + lane < nsteps) {
      hm = gp[hb + 2 * (c0 + lane)];
      dk = blk_load<BS>(vals + (size_t)gp[hb + 2 * (c0 + lane) + 1] * BB);
    }
    // (rows of two registers per lane take the chunk in two halves: the same number of source registers)
    constexpr int HC = NE > 1 ? CH / 2 : CH;
#pragma unroll
    for (int h0 = 0; h0 < CH; h0 += HC) {
      Blk<BS> src[NE][HC];
#pragma unroll
      for (int r = 0; r < NE; ++r)
#pragma unroll
        for (int q = 0; q < HC; ++q) {
          const uint32_t sidx = (code[r][(h0 + q) >> 1] >> (16 * ((h0 + q) & 1))) & 0xffffu;
#pragma unroll
          for (int i = 0; i < BB; ++i) src[r][q].a[i] = 0.0;
          if (sidx != 0xffffu) src[r][q] = blk_load<BS>(vals + (size_t)sidx * BB);
        }
#pragma unroll
      for (int q = h0; q < h0 + HC; ++q) {
        if (c0 + q < nsteps) {  // wave-uniform
          const int m = __builtin_amdgcn_readlane(hm, q);  // entry position of the multiplier
          Blk<BS> lv, dks;
#pragma unroll
          for (int i = 0; i < BB; ++i) {
            dks.a[i] = readlane_f64(dk.a[i], q);
            lv.a[i] = readlane_f64(NE > 1 && m >= 64 ? v[NE - 1].a[i] : v[0].a[i], m & 63);
          }
          const Blk<BS> lik = blk_mul<BS>(lv, dks);  // nz_l * inv(A_kk)
#pragma unroll
          for (int r = 0; r < NE; ++r)
            if (lane + 64 * r == m) v[r] = lik;
          if (blk_nonzero<BS>(lik)) {
#pragma unroll
            for (int r = 0; r < NE; ++r)
              if (((code[r][q >> 1] >> (16 * (q & 1))) & 0xffffu) != 0xffffu) blk_sub<BS>(v[r], blk_mul<BS>(lik, src[r][q - h0]));
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NE; ++r) {
    if (lane + 64 * r == dpos) v[r] = blk_inv<BS>(v[r]);
    if (idx[r] != 0xffffu) blk_store<BS>(vals + (size_t)idx[r] * BB, v[r]);
  }
}

template <int BS>
__global__ void __launch_bounds__(512) ilu_factor_rows_kernel(IluDev F, const double *__restrict__ aval, const int32_t *__restrict__ lbase,
                                       const int32_t *__restrict__ ubase, const int32_t *__restrict__ pbase,
                                       const int32_t *__restrict__ jl_map, const int32_t *__restrict__ ju_map,
                                       const int32_t *__restrict__ jd_map, const uint16_t *__restrict__ prog, int max_vals,
                                       const int32_t *__restrict__ dbase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BB = BS * BS;
  double *vals = reinterpret_cast<double *>(smem);
  uint16_t *hw = reinterpret_cast<uint16_t *>(vals + (size_t)max_vals * BB);
  const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int nr = F.blk_ptr[b + 1] - F.blk_ptr[b];
  const int l0 = lbase[b], nl = lbase[b + 1] - l0;
  const int u0 = ubase[b], nu = ubase[b + 1] - u0;
  const int d0 = dbase[b], nd = dbase[b + 1] - d0;
  // the block's program: [levels] [level row pointers (levels + 1)] [per row: offset, entries, multipliers, position of the pivot]
  // -- the head, kept in LDS -- then the rows, read from global memory where they are needed (an L2-resident stream: with the
  // rows in LDS two blocks fit a CU, with the head only seven).  A row: [entry indices] [multiplier position, pivot index of its
  // row] x multipliers, then (from a multiple of four words) the source table, chunk by chunk: [entry][multiplier of the chunk]
  const uint16_t *__restrict__ gp = prog + pbase[b];
  const int nlev = (int)gp[0], hlen = 2 + nlev + 4 * nr;
  JH_T(0);
  prog_gather<BS>(vals, aval, jl_map, l0, nl, 0, tid, T);
  prog_gather<BS>(vals, aval, ju_map, u0, nu, nl, tid, T);
  prog_gather<BS>(vals, aval, jd_map, d0, nd, nl + nu, tid, T);
  JH_T(1);
  for (int j = tid; j < hlen; j += T) hw[j] = gp[j];
  __syncthreads();
  JH_T(2);
#ifdef JH_APPLY_TIMING
  unsigned long long t_rows = 0, n_rows = 0;
#endif
  const uint16_t *levp = hw + 1, *rows = hw + 2 + nlev;
  const int lane = tid & 63, wv = tid >> 6, nwv = T >> 6;
  // level 0: no multipliers, a thread per row inverts the pivot (its only entry)
  for (int t = (int)levp[0] + tid; t < (int)levp[1]; t += T) {
    const int di = (int)gp[rows[4 * t]];
    blk_store<BS>(vals + (size_t)di * BB, blk_inv<BS>(blk_load<BS>(vals + (size_t)di * BB)));
  }
  __syncthreads();
  for (int lev = 1; lev < nlev; ++lev) {
    const int s = (int)levp[lev], e = (int)levp[lev + 1];
    for (int t = s + wv; t < e; t += nwv) {  // wave-uniform
#ifdef JH_APPLY_TIMING
      const unsigned long long tr0 = __builtin_readcyclecounter();
#endif
      // (wave-uniform words: in scalar registers, so that the multiplier steps branch instead of masking)
      const int o = __builtin_amdgcn_readfirstlane((int)rows[4 * t]), nent = __builtin_amdgcn_readfirstlane((int)rows[4 * t + 1]);
      const int nsteps = __builtin_amdgcn_readfirstlane((int)rows[4 * t + 2]), dpos = __builtin_amdgcn_readfirstlane((int)rows[4 * t + 3]);
      if (nent <= 64) prog_row_eliminate<BS, 1>(vals, gp, o, nent, nsteps, dpos, lane);
      else prog_row_eliminate<BS, 2>(vals, gp, o, nent, nsteps, dpos, lane);
#ifdef JH_APPLY_TIMING
      t_rows += __builtin_readcyclecounter() - tr0; ++n_rows;
#endif
    }
    __syncthreads();
  }
  JH_T(3);
#ifdef JH_APPLY_TIMING
  JH_TV(6, t_rows); JH_TV(7, n_rows);
#endif
  for (int j = tid; j < nl * BB; j += T) F.l_val[(size_t)l0 * BB + j] = vals[j];
  for (int j = tid; j < nu * BB; j += T) F.u_val[(size_t)u0 * BB + j] = vals[(size_t)nl * BB + j];
  for (int j = tid; j < nd * BB; j += T) F.dinv[(size_t)d0 * BB + j] = vals[(size_t)(nl + nu) * BB + j];
  JH_T(4); JH_T(5);
}

// The rows form of one block's program (see ilu_factor_rows_kernel) from its instruction form; false when it does not fit the
// 16-bit words (the caller keeps the instruction form for the whole matrix then).
static bool prog_rows_form(const std::vector<uint16_t> &prog, int nrb, int nvals, const int32_t *levp, int nlev, int32_t row0, int CH,
                           std::vector<uint16_t> &out) {
  out.clear();
  if (nvals >= 0xffff || prog.empty() || nlev < 1) return false;
  const uint16_t *rptr = prog.data(), *rdiag = rptr + nrb + 1, *code = rptr + 2 * nrb + 1;
  std::vector<uint16_t> &W = out;
  W.assign(2 + (size_t)nlev + 4 * (size_t)nrb, 0);
  W[0] = (uint16_t)nlev;
  for (int l = 0; l <= nlev; ++l) W[1 + l] = (uint16_t)(levp[l] - row0);
  std::vector<uint16_t> ent;
  struct Step { uint16_t lidx, kd; int first, n; };
  std::vector<Step> steps;
  for (int lt = 0; lt < nrb; ++lt) {
    while (W.size() & 3) W.push_back(0);
    if (W.size() >= 0xffff) return false;
    ent.assign(1, rdiag[lt]);
    steps.clear();
    for (int pc = rptr[lt], pe = rptr[lt + 1]; pc < pe;) {
      const int nupd = code[pc + 2];
      steps.push_back({code[pc], code[pc + 1], pc + 3, nupd});
      ent.push_back(code[pc]);
      for (int u = 0; u < nupd; ++u) ent.push_back(code[pc + 3 + 2 * u]);
      pc += 3 + 2 * nupd;
    }
    std::sort(ent.begin(), ent.end());
    ent.erase(std::unique(ent.begin(), ent.end()), ent.end());
    const int nent = (int)ent.size(), nsteps = (int)steps.size();
    if (nent > PROG_ROWS_ENT || nsteps >= 0xffff) return false;
    auto pos = [&](uint16_t x) { return (uint16_t)(std::lower_bound(ent.begin(), ent.end(), x) - ent.begin()); };
    uint16_t *hr = W.data() + 2 + nlev + 4 * (size_t)lt;
    hr[0] = (uint16_t)W.size(); hr[1] = (uint16_t)nent; hr[2] = (uint16_t)nsteps; hr[3] = pos(rdiag[lt]);
    W.insert(W.end(), ent.begin(), ent.end());
    for (const Step &st : steps) { W.push_back(pos(st.lidx)); W.push_back(st.kd); }
    while (W.size() & 3) W.push_back(0);
    const size_t tb = W.size();
    const int nch = (nsteps + CH - 1) / CH;
    W.resize(tb + (size_t)nch * nent * CH, (uint16_t)0xffffu);
    for (int sidx = 0; sidx < nsteps; ++sidx) {
      const Step &st = steps[sidx];
      for (int u = 0; u < st.n; ++u)
        W[tb + ((size_t)(sidx / CH) * nent + pos(code[st.first + 2 * u])) * CH + sidx % CH] = code[st.first + 2 * u + 1];
    }
  }
  while (W.size() & 3) W.push_back(0);
  return W.size() < 0xffff;
}

// launches the program-driven refactorisation: wave or thread per row
template <int BS>
static void launch_factor_prog(jh_ilu_s *M, const IluDev &F, const double *aval, const int32_t *lmap, const int32_t *umap,
                               const int32_t *dmap, bool wpr, int threads, int64_t nb, hipStream_t s) {
  auto go = [&](auto kern) {
    if (M->prog_lds_bytes > 64 * 1024)
      JH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M->prog_lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(threads), M->prog_lds_bytes, s, F, aval, M->d_blk_lbase.p, M->d_blk_ubase.p,
                       M->d_blk_prog.p, lmap, umap, dmap, M->d_prog.p, M->prog_max_vals, M->d_blk_dbase.p);
  };
  if (wpr) go(ilu_factor_prog_kernel<BS, true>);
  else go(ilu_factor_prog_kernel<BS, false>);
}

// LDS mode: one workgroup per block, levels separated by __syncthreads()
template <int BS>
__global__ void ilu_factor_blocks_kernel(IluDev F) {
  const int b = blockIdx.x;
  const int b0 = F.blk_ptr[b];
  const int l0 = F.flev_off[b], l1 = F.flev_off[b + 1] - 1;  // levels [l0, l1); flev_ptr[l1] is the end sentinel
  for (int lev = l0 + 1; lev < l1; ++lev) {                  // level 0 has no L entries
    const int s = F.flev_ptr[lev], e = F.flev_ptr[lev + 1];
    for (int t = s + threadIdx.x; t < e; t += blockDim.x) factor_row<BS>(F, t, b0);
    __syncthreads();
  }
}
// GLOBAL mode: one launch per level
template <int BS>
__global__ void ilu_factor_level_kernel(IluDev F, int s, int e) {
  int t = s + blockIdx.x * blockDim.x + threadIdx.x;
  if (t < e) factor_row<BS>(F, t, 0);
}

// ---- triangular solves (invert_row!, forward/backward_substitute!, ilu0.jl:156-187) -----------------------------------
template <int BS>
__device__ __forceinline__ void fwd_row(const IluDev &F, int t, int lt, double *xs) {
  double v[BS];
#pragma unroll
  for (int e = 0; e < BS; ++e) v[e] = xs[lt * BS + e];
  for (int j = F.l_ptr[t]; j < F.l_ptr[t + 1]; ++j) {
    const int k = F.l_col[j];
    if (BS == 1) {
      v[0] -= F.l_val[j] * xs[k];
    } else {
      const double *A = F.l_val + (size_t)j * BS * BS;
#pragma unroll
      for (int e = 0; e < BS; ++e) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < BS; ++d) s += A[d * BS + e] * xs[k * BS + d];
        v[e] -= s;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < BS; ++e) xs[lt * BS + e] = v[e];
}
template <int BS>
__device__ __forceinline__ void bwd_row(const IluDev &F, int pos, int lt, double *xs) {
  double v[BS];
#pragma unroll
  for (int e = 0; e < BS; ++e) v[e] = xs[lt * BS + e];
  for (int j = F.u_ptr[pos]; j < F.u_ptr[pos + 1]; ++j) {
    const int k = F.u_col[j];
    if (BS == 1) {
      v[0] -= F.u_val[j] * xs[k];
    } else {
      const double *A = F.u_val + (size_t)j * BS * BS;
#pragma unroll
      for (int e = 0; e < BS; ++e) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < BS; ++d) s += A[d * BS + e] * xs[k * BS + d];
        v[e] -= s;
      }
    }
  }
  const double *D = F.dinv + (size_t)pos * BS * BS;
  if (BS == 1) {
    xs[lt] = D[0] * v[0];
  } else {
    double o[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < BS; ++d) s += D[d * BS + e] * v[d];
      o[e] = s;
    }
#pragma unroll
    for (int e = 0; e < BS; ++e) xs[lt * BS + e] = o[e];
  }
}

// ---- software-pipelined LDS apply -------------------------------------------------------------------------------
// A plain level loop is latency-bound: every level issues two dependent global loads (row pointers, then
// entries) before it can touch LDS (0.66 ms at 10M cells).  Nothing but the LDS vector depends on earlier levels, so the factor data of
// level l+1 (entries) and l+2 (row pointers) is fetched into registers while level l computes: the per-level
// critical path shrinks to LDS latency + one barrier.  Rows longer than PFW entries take a slow tail loop.
// (scalar matrices: 12 since round 3 -- only rows beyond the jagged layouts still take these kernels, i.e. polyhedral cells with ~6
// strict-L / strict-U entries per row on average, whose tails were global loads inside the level loop.  2M-cell polyhedral grid,
// width 4 / 8 / 12: solve 8.03 / 6.29 / 6.11 ms, 99.7 / 120.7 / 123.5 Newton it/s; 116 VGPRs, no scratch.  -DJH_PFW_SCALAR=n
// through JH_EXTRA_FLAGS for experiments.)
#ifndef JH_PFW_SCALAR
#define JH_PFW_SCALAR 12
#endif
template <int BS> struct PfWidth { static constexpr int value = BS == 1 ? JH_PFW_SCALAR : 4; };
#define PFW (PfWidth<BS>::value)

template <int BS>
struct RowPF {
  int lt;   // local row id (index into xs), -1 = lane idle
  int s, e; // entry range
  int col[PFW];
  double val[PFW * BS * BS];
  double dinv[BS * BS];
};

// C16: read the 16-bit copies of the block-local metadata (chunked apply)
template <int BS, bool BWD, bool C16 = false>
__device__ __forceinline__ void pf_ptrs(const IluDev &F, RowPF<BS> &R, int idx, int end, int b0) {
  // idx: ilu row (fwd) or U-order position (bwd)
  if (idx < end) {
    if (BWD) {
      R.lt = C16 ? (int)F.u_row16[idx] : F.u_row[idx];
      R.s = F.u_ptr[idx];
      R.e = F.u_ptr[idx + 1];
#pragma unroll
      for (int i = 0; i < BS * BS; ++i) R.dinv[i] = F.dinv[(size_t)idx * BS * BS + i];
    } else {
      R.lt = idx - b0;
      R.s = F.l_ptr[idx];
      R.e = F.l_ptr[idx + 1];
    }
  } else {
    R.lt = -1;
    R.s = R.e = 0;
  }
}
template <int BS, bool BWD, bool C16 = false>
__device__ __forceinline__ void pf_entries(const IluDev &F, RowPF<BS> &R) {
  const int32_t *cols = BWD ? F.u_col : F.l_col;
  const uint16_t *cols16 = BWD ? F.u_col16 : F.l_col16;
  const double *vals = BWD ? F.u_val : F.l_val;
#pragma unroll
  for (int j = 0; j < PFW; ++j) {
    if (R.s + j < R.e) {
      R.col[j] = C16 ? (int)cols16[R.s + j] : cols[R.s + j];
#pragma unroll
      for (int i = 0; i < BS * BS; ++i) R.val[j * BS * BS + i] = vals[(size_t)(R.s + j) * BS * BS + i];
    }
  }
}
template <int BS, bool BWD, bool C16 = false>
__device__ __forceinline__ void pf_compute(const IluDev &F, const RowPF<BS> &R, double *xs) {
  if (R.lt < 0) return;
  const int32_t *cols = BWD ? F.u_col : F.l_col;
  const uint16_t *cols16 = BWD ? F.u_col16 : F.l_col16;
  const double *vals = BWD ? F.u_val : F.l_val;
  double v[BS];
#pragma unroll
  for (int e = 0; e < BS; ++e) v[e] = xs[R.lt * BS + e];
#pragma unroll
  for (int j = 0; j < PFW; ++j) {
    if (R.s + j < R.e) {
      const int k = R.col[j];
      if (BS == 1) {
        v[0] -= R.val[j] * xs[k];
      } else {
#pragma unroll
        for (int e = 0; e < BS; ++e) {
          double sum = 0.0;
#pragma unroll
          for (int d = 0; d < BS; ++d) sum += R.val[j * BS * BS + d * BS + e] * xs[k * BS + d];
          v[e] -= sum;
        }
      }
    }
  }
  for (int j = R.s + PFW; j < R.e; ++j) {  // slow tail for long rows
    const int k = C16 ? (int)cols16[j] : cols[j];
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      double sum = 0.0;
#pragma unroll
      for (int d = 0; d < BS; ++d) sum += vals[(size_t)j * BS * BS + d * BS + e] * xs[k * BS + d];
      v[e] -= sum;
    }
  }
  if (BWD) {
    if (BS == 1) {
      xs[R.lt] = R.dinv[0] * v[0];
    } else {
      double o[BS];
#pragma unroll
      for (int e = 0; e < BS; ++e) {
        double sum = 0.0;
#pragma unroll
        for (int d = 0; d < BS; ++d) sum += R.dinv[d * BS + e] * v[d];
        o[e] = sum;
      }
#pragma unroll
      for (int e = 0; e < BS; ++e) xs[R.lt * BS + e] = o[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < BS; ++e) xs[R.lt * BS + e] = v[e];
  }
}

template <int BS, bool BWD>
__device__ __forceinline__ void pf_sweep(const IluDev &F, double *xs, int b0, int lev_begin, int lev_end) {
  const int32_t *lptr = BWD ? F.blev_ptr : F.flev_ptr;
  const int T = blockDim.x, tid = threadIdx.x;
  if (lev_begin >= lev_end) return;
  RowPF<BS> cur, nxt;
  // prologue: pointers+entries of the first level, pointers of the second
  int s0 = lptr[lev_begin], e0 = lptr[lev_begin + 1];
  pf_ptrs<BS, BWD>(F, cur, s0 + tid, e0, b0);
  pf_entries<BS, BWD>(F, cur);
  int s1 = e0, e1 = e0;
  if (lev_begin + 1 < lev_end) { e1 = lptr[lev_begin + 2]; }
  pf_ptrs<BS, BWD>(F, nxt, s1 + tid, e1, b0);
  for (int lev = lev_begin; lev < lev_end; ++lev) {
    // issue the loads of the next level's entries and the level-after-next's pointers before computing
    RowPF<BS> nn;
    pf_entries<BS, BWD>(F, nxt);
    int s2 = e1, e2 = e1;
    if (lev + 2 < lev_end) e2 = lptr[lev + 3];
    pf_ptrs<BS, BWD>(F, nn, s2 + tid, e2, b0);
    pf_compute<BS, BWD>(F, cur, xs);
    for (int idx = s0 + tid + T; idx < e0; idx += T) {  // levels wider than the workgroup: not prefetched
      RowPF<BS> r;
      pf_ptrs<BS, BWD>(F, r, idx, e0, b0);
      pf_entries<BS, BWD>(F, r);
      pf_compute<BS, BWD>(F, r, xs);
    }
    __syncthreads();
    cur = nxt;
    nxt = nn;
    s0 = s1; e0 = e1;
    s1 = s2; e1 = e2;
  }
}

template <int BS>
__global__ void ilu_apply_blocks_pf_kernel(IluDev F, const double *__restrict__ bvec, double *__restrict__ xvec) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int b = blockIdx.x;
  const int b0 = F.blk_ptr[b], b1 = F.blk_ptr[b + 1];
  const int nr = b1 - b0;
  for (int t = threadIdx.x; t < nr; t += blockDim.x) {
    const int dev = F.rowmap[b0 + t];
#pragma unroll
    for (int e = 0; e < BS; ++e) xs[t * BS + e] = bvec[(size_t)dev * BS + e];
  }
  __syncthreads();
  pf_sweep<BS, false>(F, xs, b0, F.flev_off[b] + 1, F.flev_off[b + 1] - 1);  // level 0 rows have no L entries
  pf_sweep<BS, true>(F, xs, b0, F.blev_off[b], F.blev_off[b + 1] - 1);
  for (int t = threadIdx.x; t < nr; t += blockDim.x) {
    const int dev = F.rowmap[b0 + t];
#pragma unroll
    for (int e = 0; e < BS; ++e) xvec[(size_t)dev * BS + e] = xs[t * BS + e];
  }
}

// ---- chunked variant: one wavefront per block, 64 rows in flight regardless of level width --------------------------
// The tail of a block's level structure is narrow (a few rows per level), so prefetching "the next level" keeps
// only a few rows in flight and every level pays a full memory latency.  Here the wavefront walks the block's rows
// (already sorted by level) in chunks of 64: each lane owns one row of the chunk, the factor data of chunk c+1 and
// the row pointers of chunk c+2 are in flight while chunk c is processed level by level out of registers and LDS.
template <int BS, bool BWD>
__device__ __forceinline__ void chunk_sweep(const IluDev &F, double *xs, int b0, int b1, int skip_level0) {
  const int lane = threadIdx.x;
  const uint16_t *levs = BWD ? F.u_lev : F.l_lev;
  RowPF<BS> cur, nxt, nn;
  int lv_cur = -1, lv_nxt = -1, lv_nn = -1;
  int base = b0;
  // prologue
  pf_ptrs<BS, BWD, true>(F, cur, base + lane, b1, b0);
  if (base + lane < b1) lv_cur = levs[base + lane];
  pf_entries<BS, BWD, true>(F, cur);
  pf_ptrs<BS, BWD, true>(F, nxt, base + 64 + lane, b1, b0);
  if (base + 64 + lane < b1) lv_nxt = levs[base + 64 + lane];
  for (; base < b1; base += 64) {
    pf_entries<BS, BWD, true>(F, nxt);
    pf_ptrs<BS, BWD, true>(F, nn, base + 128 + lane, b1, b0);
    lv_nn = (base + 128 + lane < b1) ? levs[base + 128 + lane] : -1;
    // levels present in this chunk: rows are sorted by level, so [level of lane 0, level of the last valid lane]
    const int nvalid = min(64, b1 - base);
    const int lv_lo = __shfl(lv_cur, 0, 64);
    const int lv_hi = __shfl(lv_cur, nvalid - 1, 64);
    for (int lv = max(lv_lo, skip_level0); lv <= lv_hi; ++lv) {
      if (lv_cur == lv) pf_compute<BS, BWD, true>(F, cur, xs);
      wave_lds_fence();  // (one wavefront per block: __syncthreads() would also wait for the next chunk's prefetched entries)
    }
    cur = nxt; lv_cur = lv_nxt;
    nxt = nn; lv_nxt = lv_nn;
  }
}

// GM = 0: input vector read as is.  GM = 1 / 2: the BiCGStab vector updates that produce the input are fused into the
// gather (the kernel is latency-, not bandwidth-bound, so the extra streams ride along):
//   1:  s = r - alpha*q            (alpha = rho/<c,q>)             input := s, s stored
//   2:  p = r + beta*(p - omega*q) (beta = (rho'/rho)(alpha/omega)) input := p, p stored
// Start of a one-wavefront-per-block apply with a fused BiCGStab update (GM 1 / 2): the pending second reduction stage (PendSum),
// the previous iteration's record (workgroup 0) and the coefficients -- GM 1: ca = alpha; GM 2: ca = beta, cb = omega.
// Two halves so that a kernel can issue other loads between the partials' loads and their use.
// false: the launch is a no-op (speculative iteration after the solve has converged); the answer is uniform over the wavefront.
template <int GM>
struct PendRegs {
  static constexpr int CNT = GM == 2 ? 2 : 1;  // GM 1: <c, A y>; GM 2: (rho', ||r||^2)
  double v[CNT][PEND_MAX / 64];
  double rho, rho_next, cv, ts, tt, done;  // the device scalars of the update, requested together with the partials
  XrRegs xr;                               // several ranks: the peers' sums (first poll issued with everything else)
};
template <int GM>
__device__ __forceinline__ void apply_prologue_loads(const IluGather &G, PendRegs<GM> &R) {
  if (GM == 0) return;
  // (before workgroup 0 stores anything to the scalars: behind that store these loads would be a dependent memory latency of
  // their own at the head of every wavefront)
  R.rho = G.sc[G.rho_slot];
  R.cv = G.sc[G.cv_slot];
  R.rho_next = GM == 2 ? G.sc[G.rho_next_slot] : 0.0;
  R.ts = GM == 2 ? G.sc[G.ts_slot] : 0.0;
  R.tt = GM == 2 ? G.sc[G.ts_slot + 1] : 0.0;
  R.done = G.done ? *G.done : 0.0;
  if (!G.pend.part) return;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < PEND_MAX / 64; ++j) {
    const int i = lane + 64 * j;
#pragma unroll
    for (int k = 0; k < PendRegs<GM>::CNT; ++k) R.v[k][j] = (i < G.pend.nparts) ? G.pend.part[(size_t)k * G.pend.stride + i] : 0.0;
  }
  if (G.pend.mail.self) xr_load(G.pend.mail, R.xr);
}
template <int GM>
__device__ __forceinline__ bool apply_prologue(const IluGather &G, PendRegs<GM> &R, double &ca, double &cb) {
  if (GM == 0) return true;
  double p[2] = {0.0, 0.0};
  const bool pend = G.pend.part != nullptr;
  if (pend) {  // the sums of pend_sum_wave, from loads issued earlier
#pragma unroll
    for (int k = 0; k < PendRegs<GM>::CNT; ++k) {
      double a = R.v[k][0];
#pragma unroll
      for (int j = 1; j < PEND_MAX / 64; ++j) a += R.v[k][j];
      // (more than PEND_MAX partials -- the CSR tile product of a large block matrix: the later batches in sequence, same order as
      // pend_sum_wave)
      for (int base = PEND_MAX; base < G.pend.nparts; base += PEND_MAX) {
        const int lane = threadIdx.x & 63;
        double w[PEND_MAX / 64];
#pragma unroll
        for (int j = 0; j < PEND_MAX / 64; ++j) {
          const int i = base + lane + 64 * j;
          w[j] = (i < G.pend.nparts) ? G.pend.part[(size_t)k * G.pend.stride + i] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < PEND_MAX / 64; ++j) a += w[j];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
      p[k] = a;
    }
    if (G.pend.mail.self) {  // the sums over the ranks (krylov.jl:51-105), finished by every wavefront as well
      if (blockIdx.x == 0) xr_push(G.pend.mail, p[0], p[1]);
      xr_sum(G.pend.mail, R.xr, blockIdx.x == 0, p[0], p[1]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // for the kernels behind this one
      G.sc_rw[G.pend.out_slot] = p[0];
      if (GM == 2) G.sc_rw[G.pend.out_slot + 1] = p[1];
    }
  }
  double dn = R.done;
  // every wavefront holds (rho', ||r||^2) of the previous iteration: all of them see its outcome, not only workgroup 0, and the
  // speculative apply past convergence leaves p and y alone
  if (GM == 2 && pend && G.pub_rec && sqrt(p[1]) <= G.pub_eps) dn = 1.0;
  if (GM == 2 && G.pub_rec && blockIdx.x == 0) {  // workgroup 0 publishes the previous iteration's record and knows its outcome
    if (threadIdx.x == 0) {
      publish_record(G.sc_rw, G.pub_pair, G.pub_eps, G.pub_rec, G.pub_seq);  // raises the done flag on convergence
      dn = (dn != 0.0 || G.sc_rw[S_DONE] != 0.0) ? 1.0 : 0.0;
    }
    dn = __shfl(dn, 0, 64);
  }
  if (dn != 0.0) return false;  // speculative launch after the Krylov solve has converged
  if (GM == 1) {
    ca = R.rho / (pend ? p[0] : R.cv);  // alpha
  } else if (GM == 2) {
    const double rho = R.rho, rho_next = pend ? p[0] : R.rho_next;
    const double alpha = rho / R.cv;
    cb = R.tt == 0.0 ? 0.0 : R.ts / R.tt;  // omega (0/0 guard, see bicg_omega)
    ca = (rho_next / rho) * (alpha / cb);  // beta
  }
  return true;
}
template <int BS, int GM>
__global__ __launch_bounds__(64) void ilu_apply_chunked_kernel(IluDev F, const double *__restrict__ bvec, double *__restrict__ xvec,
                                                              IluGather G) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  double ca = 0.0, cb = 0.0;
  PendRegs<GM> pr;
  apply_prologue_loads<GM>(G, pr);
  if (!apply_prologue<GM>(G, pr, ca, cb)) return;
  const int b = blockIdx.x;
  const int b0 = F.blk_ptr[b], b1 = F.blk_ptr[b + 1];
  const int nr = b1 - b0;
  // the block's rows are the device rows [b0, b1): they are read coalesced in device order and dropped at their ilu position
  // (rowmap16[dev] = position inside the block), so the vector loads do not wait for the map
  for (int t = threadIdx.x; t < nr; t += 64) {
    const int dev = b0 + t;
    const int pos = (int)F.rowmap16[dev];
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      const size_t o = (size_t)dev * BS + e;
      double v;
      if (GM == 0) {
        v = bvec[o];
      } else if (GM == 1) {
        v = (dev < G.n_owned_rows) ? G.r[o] - ca * G.q[o] : 0.0;
        G.out[o] = v;
      } else {
        const double pa = G.out[o] - cb * G.q[o];
        v = (dev < G.n_owned_rows) ? G.r[o] + ca * pa : 0.0;
        G.out[o] = v;
      }
      xs[pos * BS + e] = v;
    }
  }
  __syncthreads();
  chunk_sweep<BS, false>(F, xs, b0, b1, 1);  // forward: level-0 rows have no L entries
  chunk_sweep<BS, true>(F, xs, b0, b1, 0);   // backward: every row is scaled by its inverted pivot
  __syncthreads();
  for (int t = threadIdx.x; t < nr; t += 64) {
    const int dev = b0 + t;
    const int pos = (int)F.rowmap16[dev];
#pragma unroll
    for (int e = 0; e < BS; ++e) xvec[(size_t)dev * BS + e] = xs[pos * BS + e];
  }
  if (F.send_ptr) {  // rows that neighbouring ranks hold as ghosts go straight into the halo send buffer (no pack kernel)
    for (int j = F.send_ptr[b] + (int)threadIdx.x; j < F.send_ptr[b + 1]; j += 64) {
      const int t = F.send_local[j];
      if (F.send_dst) {  // xGMI store into the receiving rank's landing buffer
        double *dp = F.send_dst[F.send_slot[j]];
#pragma unroll
        for (int e = 0; e < BS; ++e) __hip_atomic_store(dp + e, xs[t * BS + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        const size_t o = (size_t)F.send_slot[j] * BS;
#pragma unroll
        for (int e = 0; e < BS; ++e) F.send_buf[o + e] = xs[t * BS + e];
      }
    }
  }
}

// ---- chunk-jagged apply: the same one-wavefront-per-block sweep with the factor stored for it ---------------------------------
// Counters on the row-major chunked kernel above (profiles/r02a: SQ_ACTIVE_INST_ANY x 7 waves/SIMD = 82% of the issue slots,
// 3 500 instructions per wavefront for 512 rows) say it is bound by instruction issue, not by bytes: per-row pointer pairs,
// per-entry predicated scattered loads, level bookkeeping.  Here the factor is stored the way the sweep consumes it:
//   * a sweep (forward L / backward U) walks the block in chunks of 64 rows; inside a chunk the lanes are sorted by entry
//     count and the entries are stored diagonal by diagonal (all first entries, all second entries, ...): lane l reads entry j
//     at base_j + l -- unconditional, fully coalesced loads, no row pointers (8 bytes per row and sweep gone), no padding;
//   * per chunk one 16-byte descriptor (entry base, eight one-byte diagonal counts, first / last level), read with scalar
//     loads two chunks ahead; per lane one 32-bit word (block-local row | level << 16);
//   * the entries and rows of chunk c+1 are in flight while chunk c is processed level by level out of registers and LDS;
//     a lane's own right-hand side is read when its chunk starts (nobody else writes it).
// Same arithmetic as fwd_row / bwd_row (ascending columns, inverted pivots), hence the same bits as the kernels above.
template <int BS, int KU>
struct JRow {
  unsigned word;         // row | level << 16
  int col[KU];
  double val[KU * BS * BS];
  double dinv[BS * BS];  // backward sweep only
  double kap[BS * BS];   // backward sweep with the fused product only: A_ii inv(D~_i)
  int dev;               // column-scaled U: device row of the lane's row relative to the block (| 0x8000: has out-of-block entries)
};
template <int BS, int KU, bool BWD, bool SC, bool MUL>
__device__ __forceinline__ void jds_load(const IluDev &F, const int4 &D, int chunk, int lane, JRow<BS, KU> &R) {
  constexpr int BB = BS * BS;
  const uint16_t *cols = BWD ? F.ju_col : F.jl_col;
  const double *vals = BWD ? F.u_val : F.l_val;
  R.word = (BWD ? F.jb_row : F.jf_row)[(size_t)chunk * 64 + lane];
  int off = D.x;
#define JH_JL(J)                                                                                          \
  if (J < KU && jd_count<(J < 8 ? J : 0)>(D) > 0) { /* wave-uniform: the diagonal exists in this chunk */ \
    R.col[J < KU ? J : 0] = (int)cols[off + lane]; /* lanes past the count read the next diagonal's */     \
    _Pragma("unroll") for (int i = 0; i < BB; ++i) R.val[(J < KU ? J : 0) * BB + i] = vals[(size_t)(off + lane) * BB + i]; \
    off += jd_count<(J < 8 ? J : 0)>(D);                                                                   \
  } else if (J < KU) { /* no such diagonal: a valid LDS index and a finite value for the unconditional reads of the sweep */ \
    R.col[J < KU ? J : 0] = 0;                                                                             \
    _Pragma("unroll") for (int i = 0; i < BB; ++i) R.val[(J < KU ? J : 0) * BB + i] = 0.0;                \
  }
  JH_JL(0) JH_JL(1) JH_JL(2) JH_JL(3) JH_JL(4) JH_JL(5) JH_JL(6) JH_JL(7)
#undef JH_JL
  if (BWD || SC) {
    const double *dsrc = BWD ? F.dinv : F.dinv_f;
#pragma unroll
    for (int i = 0; i < BB; ++i) R.dinv[i] = dsrc[((size_t)chunk * 64 + lane) * BB + i];
  }
  if (BWD && MUL) {
#pragma unroll
    for (int i = 0; i < BB; ++i) R.kap[i] = F.kap[((size_t)chunk * 64 + lane) * BB + i];
    R.dev = (int)F.jb_dev[(size_t)chunk * 64 + lane];
  }
}
// SC (D-ILU storage, jh_ilu_s::uscaled): the values are A's own entries; forward g^_i = inv(D~_i) (b_i - sum A_ik g^_k), backward
// y_i = g^_i - inv(D~_i) s_i with s_i = sum A_ik y_k.  MUL: the backward lane also stores A_ii y_i + s_i, the part of (A y)_i it
// knows, to qq (= q + first row of the block) at the row's device position.
// First chunk of a sweep in flight: its two descriptors and the lanes' entries.  Issued at the very start of the kernel for BOTH
// sweeps -- the sweeps' first loads then overlap the gather phase (and each other) instead of being two exposed descriptor ->
// entries latency chains per wavefront (~3 us each; at 1.25M rows per GPU a block's whole sweep is ~10 us).
template <int BS, int KU>
struct JPre {
  int4 dc, dn;
  JRow<BS, KU> cur;
};
template <int BS, int KU, bool BWD, bool SC, bool MUL>
__device__ __forceinline__ void jds_prefetch(const IluDev &F, int c0, int lane, JPre<BS, KU> &P) {
  const int4 *desc = BWD ? F.jb_desc : F.jf_desc;
  P.dc = jds_desc(desc, c0);
  P.dn = jds_desc(desc, c0 + 1);  // (the descriptor arrays are padded by four entries)
  jds_load<BS, KU, BWD, SC, MUL>(F, P.dc, c0, lane, P.cur);
}
// The levels of ONE chunk: the lane's row (cur) waits for its level, reads its operands from the block's vector in LDS, writes
// its result there.
// VR: rows of more than 8 entries are chains of lanes (virtual rows, see the layout): a lane then reads its row's entry of the vector
// when its own level comes up -- an earlier lane of the chain may have written a partial result there -- instead of at the start
// of the chunk.
template <int BS, int KU, bool BWD, bool SC, bool MUL, bool VR = false>
__device__ __forceinline__ void jds_chunk(double *xs, int lane, const int4 &dc, const JRow<BS, KU> &cur, double *qq) {
  constexpr int BB = BS * BS;
  const int lt = (int)(cur.word & 0xffffu), lev = (int)(cur.word >> 16);
  const int lvlo = dc.w & 0xffff, lvhi = (int)((unsigned)dc.w >> 16);
  double v[BS], x0[BS];
#pragma unroll
  for (int e = 0; e < BS; ++e) { x0[e] = VR ? 0.0 : xs[lt * BS + e]; v[e] = (SC && BWD) ? 0.0 : x0[e]; }  // this row's entry: only the row itself ever writes it
  // Entries the lane does not have (past a diagonal's count; diagonals the chunk does not have) become a zero coefficient times the
  // lane's own entry of the vector, ONCE per chunk: the level loop -- one VALU-bound step per dependency level, all wavefronts of
  // a SIMD in it together -- then is four multiply-subtracts, not four multiply-subtract-selects (v - 0 * x == v up to the sign
  // of a zero)
  int kc[KU];
  double cv[KU * BB];
#define JH_JM(J)                                                                                            \
  if (J < KU) {                                                                                             \
    const bool a = lane < jd_count<(J < 8 ? J : 0)>(dc);                                                   \
    kc[J < KU ? J : 0] = a ? cur.col[J < KU ? J : 0] : lt; /* (not the stray index: LDS past the block's rows may hold NaN) */ \
    _Pragma("unroll") for (int i = 0; i < BB; ++i) cv[(J < KU ? J : 0) * BB + i] = a ? cur.val[(J < KU ? J : 0) * BB + i] : 0.0; \
  }
  JH_JM(0) JH_JM(1) JH_JM(2) JH_JM(3) JH_JM(4) JH_JM(5) JH_JM(6) JH_JM(7)
#undef JH_JM
  // forward, LU form: level-0 rows have no L entries, x = b; D-ILU form: they are scaled by their pivot like every row
  for (int lv = (BWD || SC) ? lvlo : max(lvlo, 1); lv <= lvhi; ++lv) {
    if (lev == lv) {
      // all operands of the row first -- the LDS reads in flight together, one wait -- then the updates in column order: per level
      // one LDS round trip instead of one per entry behind a branch each
      // (skipping the reads of diagonals no row of the level has -- a ballot and a branch each -- was measured slower: 16.4k instead
      // of 11.0k cycles per forward sweep of a 256-row block)
      double xk[KU][BS];
      if (VR) {
#pragma unroll
        for (int e = 0; e < BS; ++e) { x0[e] = xs[lt * BS + e]; v[e] = x0[e]; }
      }
#pragma unroll
      for (int j = 0; j < KU; ++j) {
#pragma unroll
        for (int d = 0; d < BS; ++d) xk[j][d] = xs[kc[j] * BS + d];
      }
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        if (BS == 1) {
          v[0] = v[0] - cv[j] * xk[j][0];
        } else {
#pragma unroll
          for (int e = 0; e < BS; ++e) {
            double sum = 0.0;
#pragma unroll
            for (int d = 0; d < BS; ++d) sum += cv[j * BB + d * BS + e] * xk[j][d];
            v[e] = v[e] - sum;
          }
        }
      }
      if (BWD || SC) {
        double o[BS];  // inv(D~_i) v
#pragma unroll
        for (int e = 0; e < BS; ++e) {
          double sum = 0.0;
#pragma unroll
          for (int d = 0; d < BS; ++d) sum += cur.dinv[d * BS + e] * v[d];
          o[e] = sum;
        }
        if (SC && BWD) {  // v = -s_i: y_i = g^_i + inv(D~_i) v
#pragma unroll
          for (int e = 0; e < BS; ++e) o[e] += x0[e];
        }
#pragma unroll
        for (int e = 0; e < BS; ++e) xs[lt * BS + e] = o[e];
        if (MUL) {  // A_ii y_i + s_i
          const int dev = cur.dev & 0x7fff;
#pragma unroll
          for (int e = 0; e < BS; ++e) {
            double sum = -v[e];
#pragma unroll
            for (int d = 0; d < BS; ++d) sum += cur.kap[d * BS + e] * o[d];
            qq[dev * BS + e] = sum;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < BS; ++e) xs[lt * BS + e] = v[e];
      }
    }
    wave_lds_fence();  // this level's LDS writes before the next level's reads
  }
}
// The chunk loop: the entries of chunk c+1 are requested before chunk c walks its levels.  Measured alternatives (round 4, in-kernel
// clock stamps of tools/apply_timing.py and A/B runs of two builds on one box): the loop unrolled over two or three register sets in
// rotating roles with a fixed number of loads per step -- so that the wait in front of a chunk is s_waitcnt vmcnt(n) with younger
// chunks in flight instead of the full drain the copy "A = B" below implies -- was SLOWER at every size (10M rows 187 vs 184 us per
// launch, Cartesian 10M 222-242 vs 208 us, 1.25M rows 30.5 vs 29.9 us), two chunks ahead slower still: what a level step waits for
// is the LDS round trip of its own dependency chain, not the entry stream.
template <int BS, int KU, bool BWD, bool SC, bool MUL, bool VR = false>
__device__ __forceinline__ void jds_sweep(const IluDev &F, double *xs, int c0, int nch, int lane, const JPre<BS, KU> &P, double *qq = nullptr) {
  const int4 *desc = BWD ? F.jb_desc : F.jf_desc;
  JRow<BS, KU> A = P.cur, B;
  int4 dA = P.dc, dB = P.dn, dnext;
  for (int c = 0; c < nch; ++c) {
    dnext = jds_desc(desc, c0 + c + 2);  // (the descriptor arrays are padded by four entries)
    if (c + 1 < nch) jds_load<BS, KU, BWD, SC, MUL>(F, dB, c0 + c + 1, lane, B);
    jds_chunk<BS, KU, BWD, SC, MUL, VR>(xs, lane, dA, A, qq);
    dA = dB; dB = dnext;
    A = B;
  }
}
// Third pass of the fused product (MUL): q_i += sum_{k<i} L_ik w_k over the forward chunks -- no dependencies, all lanes of a
// chunk at once -- and the fused dot of the rows that are complete (no entry outside the block; the others are finished by
// ilu_eprod_kernel).  d0 += q.dw ; DOT == 2: d1 += q.q
template <int BS, int KU, int DOT>
__device__ __forceinline__ void jds_mul_pass(const IluDev &F, const double *xs, int c0, int nch, int lane, double *qq, const double *dwq,
                                             int dot_dev_end, double &d0, double &d1) {
  constexpr int BB = BS * BS;
  const int4 *desc = F.jf_desc;
  int4 dc = desc[c0], dn = desc[c0 + 1], dnn;
  JRow<BS, KU> cur, nxt;
  jds_load<BS, KU, false, false, false>(F, dc, c0, lane, cur);
  int devw = (int)F.jf_dev[(size_t)c0 * 64 + lane], devn = 0;
  for (int c = 0; c < nch; ++c) {
    dnn = desc[c0 + c + 2];
    if (c + 1 < nch) { jds_load<BS, KU, false, false, false>(F, dn, c0 + c + 1, lane, nxt); devn = (int)F.jf_dev[(size_t)(c0 + c + 1) * 64 + lane]; }
    const bool has_row = (cur.word >> 16) != 0xffffu;
    const int dev = devw & 0x7fff;
    double t[BS], wv[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      // the partial product was stored by another lane of this wavefront (backward sweep, ordered by the workgroup-scope fence)
      t[e] = has_row ? __hip_atomic_load(qq + dev * BS + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0.0;
      wv[e] = (DOT && has_row) ? dwq[dev * BS + e] : 0.0;
    }
#define JH_JM(J)                                                                                              \
    if (J < KU && jd_count<(J < 8 ? J : 0)>(dc) > 0) {                                                       \
      const int k = cur.col[J < KU ? J : 0];                                                                 \
      const bool a = lane < jd_count<(J < 8 ? J : 0)>(dc);                                                   \
      _Pragma("unroll") for (int e = 0; e < BS; ++e) {                                                       \
        double sum = 0.0;                                                                                     \
        _Pragma("unroll") for (int d = 0; d < BS; ++d) sum += cur.val[(J < KU ? J : 0) * BB + d * BS + e] * xs[k * BS + d]; \
        t[e] = a ? t[e] + sum : t[e];                                                                         \
      }                                                                                                       \
    }
    JH_JM(0) JH_JM(1) JH_JM(2) JH_JM(3) JH_JM(4) JH_JM(5) JH_JM(6) JH_JM(7)
#undef JH_JM
    if (has_row) {
#pragma unroll
      for (int e = 0; e < BS; ++e) qq[dev * BS + e] = t[e];
      if (DOT && !(devw & 0x8000) && dev < dot_dev_end) {
#pragma unroll
        for (int e = 0; e < BS; ++e) { d0 += t[e] * wv[e]; if (DOT == 2) d1 += t[e] * t[e]; }
      }
    }
    dc = dn; dn = dnn;
    cur = nxt;
    devw = devn;
  }
}

// MUL: 0 = apply only; 1 / 2 / 3 = fused product without a dot / with <q, dw> / with <q, dw>, <q, q> (jh_ilu_s::uscaled)
struct IluMul {
  double *q = nullptr;          // A * x (in-block part; rows with out-of-block entries are finished by ilu_eprod_kernel)
  const double *dw = nullptr;   // dot weights
  double *part = nullptr;       // one partial per block (+ pstride: second dot)
  size_t pstride = 0;
  int dot_rows = 0x7fffffff;    // device rows >= dot_rows (ghost rows of a rank-local subdomain) do not contribute to the dot
};
template <int BS, int GM, int KU, bool SC, int MUL, bool VR = false>
__global__ __launch_bounds__(64) void ilu_apply_jds_kernel(IluDev F, const double *__restrict__ bvec, double *__restrict__ xvec, IluGather G, IluMul Q) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int b = blockIdx.x, lane = threadIdx.x;
  JH_T(0);
  const int b0 = F.blk_ptr[b], b1 = F.blk_ptr[b + 1];
  const int nr = b1 - b0;
  const int c0 = F.chunk_ptr[b], nch = F.chunk_ptr[b + 1] - c0;
  // everything that does not depend on a value computed here is requested first: the dot partials, the first chunk of both sweeps,
  // the first batch of the vector rows
  double ca = 0.0, cb = 0.0;
  PendRegs<GM> pr;
  apply_prologue_loads<GM>(G, pr);
  // (the backward chunk costs a second set of entry registers for the whole forward sweep: only where that leaves the occupancy
  // alone -- scalar rows of at most 4 entries per sweep; 2x2 blocks would go from 92 to 137 VGPRs)
#ifndef JH_EARLY_BWD_KU
#define JH_EARLY_BWD_KU 4
#endif
  constexpr bool EARLY_BWD = BS == 1 && KU <= JH_EARLY_BWD_KU;
  JPre<BS, KU> pf, pb;
  jds_prefetch<BS, KU, false, SC, false>(F, c0, lane, pf);
  if (EARLY_BWD) jds_prefetch<BS, KU, true, SC, (MUL != 0)>(F, c0, lane, pb);
  // The block's rows are the device rows [b0, b1): read coalesced in device order, dropped at their ilu position.  GB row groups of
  // 64 per batch, all loads of a batch in flight together (a group at a time is two dependent memory latencies per 64 rows, in
  // series: eight of them for a 256-row block before the first sweep can start).
  constexpr int GB = BS == 1 ? 4 : 2;
  struct GRow { int pos; double a[BS], b[BS], c[BS]; };
  GRow gr[GB];
  auto gather_load = [&](int t0) {
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int t = t0 + g * 64 + lane;
      if (t < nr) {
        const int dev = b0 + t;
        gr[g].pos = (int)F.rowmap16[dev];
#pragma unroll
        for (int e = 0; e < BS; ++e) {
          const size_t o = (size_t)dev * BS + e;
          if (GM == 0) { gr[g].a[e] = bvec[o]; }
          else { gr[g].a[e] = G.r[o]; gr[g].b[e] = G.q[o]; if (GM == 2) gr[g].c[e] = G.out[o]; }
        }
      }
    }
  };
  auto gather_finish = [&](int t0) {
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int t = t0 + g * 64 + lane;
      if (t < nr) {
        const int dev = b0 + t;
#pragma unroll
        for (int e = 0; e < BS; ++e) {
          const size_t o = (size_t)dev * BS + e;
          double v;
          if (GM == 0) {
            v = (dev < G.n_owned_rows) ? gr[g].a[e] : 0.0;  // (ghost input zeroed when the caller says where the ghosts start)
          } else if (GM == 1) {
            v = (dev < G.n_owned_rows) ? gr[g].a[e] - ca * gr[g].b[e] : 0.0;
            G.out[o] = v;
          } else {
            const double pa = gr[g].c[e] - cb * gr[g].b[e];
            v = (dev < G.n_owned_rows) ? gr[g].a[e] + ca * pa : 0.0;
            G.out[o] = v;
          }
          xs[gr[g].pos * BS + e] = v;
        }
      }
    }
  };
  if (GM != 2) gather_load(0);  // (GM 2: after the partial sums, whose registers it then reuses: 6 instead of 5 wavefronts per SIMD)
  if (!apply_prologue<GM>(G, pr, ca, cb)) return;
  if (GM == 0 && MUL && G.done && *G.done != 0.0) return;
  JH_T(1);
  if (GM == 2) gather_load(0);
  gather_finish(0);
  for (int t0 = 64 * GB; t0 < nr; t0 += 64 * GB) { gather_load(t0); gather_finish(t0); }
  JH_T(2);
  __syncthreads();
  jds_sweep<BS, KU, false, SC, false, VR>(F, xs, c0, nch, lane, pf);
  JH_T(3);
  if (!EARLY_BWD) jds_prefetch<BS, KU, true, SC, (MUL != 0)>(F, c0, lane, pb);
  jds_sweep<BS, KU, true, SC, (MUL != 0), VR>(F, xs, c0, nch, lane, pb, MUL ? Q.q + (size_t)b0 * BS : nullptr);
  JH_T(4);
  __syncthreads();
  if (MUL) {
    // the partial products of the backward sweep were stored by lanes of this wavefront: a workgroup-scope fence orders them
    // before the third pass reads them back (an agent-scope fence writes back the XCD's L2: 1.7 ms per launch, measured)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double d0 = 0.0, d1 = 0.0;
    constexpr int DOT = MUL - 1;
    jds_mul_pass<BS, KU, DOT>(F, xs, c0, nch, lane, Q.q + (size_t)b0 * BS, DOT ? Q.dw + (size_t)b0 * BS : nullptr, Q.dot_rows - b0, d0, d1);
    if (DOT) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); if (DOT == 2) d1 += __shfl_down(d1, off, 64); }
      if (lane == 0) { Q.part[b] = d0; if (DOT == 2) Q.part[Q.pstride + b] = d1; }
    }
  }
  for (int t0 = 0; t0 < nr; t0 += 64 * GB) {  // (batched like the gather: the row map of GB groups in flight together)
    int pos[GB];
#pragma unroll
    for (int g = 0; g < GB; ++g) { const int t = t0 + g * 64 + lane; pos[g] = t < nr ? (int)F.rowmap16[b0 + t] : 0; }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int t = t0 + g * 64 + lane;
      if (t < nr) {
#pragma unroll
        for (int e = 0; e < BS; ++e) xvec[(size_t)(b0 + t) * BS + e] = xs[pos[g] * BS + e];
      }
    }
  }
  if (F.send_ptr) {  // rows that neighbouring ranks hold as ghosts go straight into the halo send buffer (no pack kernel)
    for (int j = F.send_ptr[b] + lane; j < F.send_ptr[b + 1]; j += 64) {
      const int t = F.send_local[j];
      double yv[BS];
#pragma unroll
      for (int e = 0; e < BS; ++e) yv[e] = xs[t * BS + e];
      if (F.send_dst) {  // xGMI store into the receiving rank's landing buffer
        double *dp = F.send_dst[F.send_slot[j]];
#pragma unroll
        for (int e = 0; e < BS; ++e) __hip_atomic_store(dp + e, yv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        const size_t o = (size_t)F.send_slot[j] * BS;
#pragma unroll
        for (int e = 0; e < BS; ++e) F.send_buf[o + e] = yv[e];
      }
    }
  }
  JH_T(5);
}
#ifdef JH_APPLY_TIMING
// rows-form factor kernel, means over the blocks of the last launch: [0] gather, [1] program copy, [2] level loop, [3] stores,
// [4] cycles wave 0 spent inside rows, [5] rows wave 0 took, [6] launch span
extern "C" int32_t jh_debug_factor_times(int64_t nblocks, double *out7) {
  std::vector<unsigned long long> h((size_t)8 * 65536);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(jh_apply_t), h.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
  const int64_t nb = std::min<int64_t>(nblocks, 65536);
  for (int k = 0; k < 7; ++k) out7[k] = 0.0;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int64_t b = 0; b < nb; ++b) {
    const unsigned long long *t = h.data() + b * 8;
    for (int k = 0; k < 4; ++k) out7[k] += (double)(t[k + 1] - t[k]) / nb;
    out7[4] += (double)t[6] / nb; out7[5] += (double)t[7] / nb;
    t0 = std::min(t0, t[0]); t1 = std::max(t1, t[5]);
  }
  out7[6] = (double)(t1 - t0);
  return 0;
}
// mean cycles of the phases over the blocks of the LAST launch: [0] prologue (pointers, partial sums, first loads), [1] gather,
// [2] forward sweep, [3] backward sweep, [4] scatter; [5] first start -> last end of the launch; [6] mean wavefront lifetime
extern "C" int32_t jh_debug_apply_times(int64_t nblocks, double *out7) {
  std::vector<unsigned long long> h((size_t)8 * 65536);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(jh_apply_t), h.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
  const int64_t nb = std::min<int64_t>(nblocks, 65536);
  for (int k = 0; k < 7; ++k) out7[k] = 0.0;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int64_t b = 0; b < nb; ++b) {
    const unsigned long long *t = h.data() + b * 8;
    for (int k = 0; k < 5; ++k) out7[k] += (double)(t[k + 1] - t[k]) / nb;
    out7[6] += (double)(t[5] - t[0]) / nb;
    t0 = std::min(t0, t[0]);
    t1 = std::max(t1, t[5]);
  }
  out7[5] = (double)(t1 - t0);
  return 0;
}
#endif

// The part of the product the blocks cannot form: q_i += sum over the entries of row i OUTSIDE its block of A_ik y_k, one thread
// per such row (13 % of the entries, a third of the rows on the bisection blocks of a tet grid), and those rows' share of the
// fused dot.  Partials go behind the part_off per-block partials of the apply kernel.
template <int BS, int DOT>
__global__ __launch_bounds__(256) void ilu_eprod_kernel(const int32_t *__restrict__ erow, const int32_t *__restrict__ eptr,
                                                        const int32_t *__restrict__ ecol, const double *__restrict__ eval,
                                                        const double *__restrict__ y, double *__restrict__ q,
                                                        int nrows, const double *__restrict__ dw, int dot_rows, double *__restrict__ part,
                                                        size_t pstride, int part_off, const double *done) {
  if (done && *done != 0.0) return;
  constexpr int BB = BS * BS;
  __shared__ double red[8];
  double d0 = 0.0, d1 = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) {
    const int row = erow[i];
    const int k0 = eptr[i], k1 = eptr[i + 1];
    double acc[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) acc[e] = q[(size_t)row * BS + e];
    for (int k = k0; k < k1; ++k) {
      const int c = ecol[k];
      const double *Ab = eval + (size_t)k * BB;
#pragma unroll
      for (int e = 0; e < BS; ++e) {
        double sum = 0.0;
#pragma unroll
        for (int d = 0; d < BS; ++d) sum += Ab[d * BS + e] * y[(size_t)c * BS + d];
        acc[e] += sum;
      }
    }
#pragma unroll
    for (int e = 0; e < BS; ++e) q[(size_t)row * BS + e] = acc[e];
    if (DOT && row < dot_rows) {
#pragma unroll
      for (int e = 0; e < BS; ++e) { d0 += acc[e] * dw[(size_t)row * BS + e]; if (DOT == 2) d1 += acc[e] * acc[e]; }
    }
  }
  if (DOT) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); if (DOT == 2) d1 += __shfl_down(d1, off, 64); }
    if (lane == 0) { red[w] = d0; red[4 + w] = d1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      part[part_off + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
      if (DOT == 2) part[pstride + part_off + blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
    }
  }
}

// old (row-major) factor arrays -> chunk-jagged arrays, after a factor kernel that still writes the row-major order
__global__ void ilu_jag_permute_kernel(double *__restrict__ dst, const double *__restrict__ src, const int32_t *__restrict__ pos, int64_t cnt, int bb) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cnt * bb; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / bb;
    const int e = (int)(i - k * bb);
    const int32_t p = pos[k];
    if (p >= 0) dst[(size_t)p * bb + e] = src[i];
  }
}

// GLOBAL mode kernels: x lives in HBM in ilu order
template <int BS>
__global__ void ilu_gather_kernel(double *xg, const double *bvec, const int32_t *rowmap, int64_t n, bool scatter) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int dev = rowmap[t];
#pragma unroll
  for (int e = 0; e < BS; ++e) {
    if (scatter) const_cast<double *>(bvec)[(size_t)dev * BS + e] = xg[t * BS + e];
    else xg[t * BS + e] = bvec[(size_t)dev * BS + e];
  }
}
template <int BS>
__global__ void ilu_fwd_level_kernel(IluDev F, double *xg, int s, int e) {
  int t = s + blockIdx.x * blockDim.x + threadIdx.x;
  if (t < e) fwd_row<BS>(F, t, t, xg);
}
template <int BS>
__global__ void ilu_bwd_level_kernel(IluDev F, double *xg, int s, int e) {
  int pos = s + blockIdx.x * blockDim.x + threadIdx.x;
  if (pos < e) bwd_row<BS>(F, pos, F.u_row[pos], xg);
}

// ---- DiagonalPreconditioner family (precond/diagonal.jl, jacobi.jl:5-18, spai.jl:40-60) --------------------------------
template <int BS>
__global__ void diag_precond_kernel(const int32_t *rowptr, const int32_t *diag, const double *val, double *D, int64_t n, int kind, double w) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int BB = BS * BS;
  const Blk<BS> Aii = blk_load<BS>(val + (size_t)diag[i] * BB);
  Blk<BS> out;
  if (kind == 1) {  // diagonal_precond(A, i, jac) = w*inv(A_ii)
    out = blk_inv<BS>(Aii);
#pragma unroll
    for (int e = 0; e < BB; ++e) out.a[e] *= w;
  } else {          // SPAI(0): inv(norm_sum)*A_ii with norm_sum = sum of v^2 over all scalar entries of the row
    double ns = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
#pragma unroll
      for (int e = 0; e < BB; ++e) { const double v = val[(size_t)k * BB + e]; ns += v * v; }
    const double inv = 1.0 / ns;
#pragma unroll
    for (int e = 0; e < BB; ++e) out.a[e] = inv * Aii.a[e];
  }
  blk_store<BS>(D + (size_t)i * BB, out);
}
template <int BS>
__global__ void diag_apply_kernel(const double *D, const double *b, double *x, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v[BS], o[BS];
#pragma unroll
  for (int e = 0; e < BS; ++e) v[e] = b[i * BS + e];
#pragma unroll
  for (int e = 0; e < BS; ++e) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < BS; ++d) s += D[(size_t)i * BS * BS + d * BS + e] * v[d];
    o[e] = s;
  }
#pragma unroll
  for (int e = 0; e < BS; ++e) x[i * BS + e] = o[e];
}

IluDev dev_view(jh_ilu M) {
  IluDev F;
  F.rowmap = M->d_rowmap.p; F.blk_ptr = M->d_blk_ptr.p;
  F.flev_off = M->d_flev_off.p; F.flev_ptr = M->d_flev_ptr.p;
  F.blev_off = M->d_blev_off.p; F.blev_ptr = M->d_blev_ptr.p;
  F.l_ptr = M->d_l_ptr.p; F.l_col = M->d_l_col.p; F.l_map = M->d_l_map.p;
  F.u_ptr = M->d_u_ptr.p; F.u_col = M->d_u_col.p; F.u_map = M->d_u_map.p;
  F.d_map = M->d_d_map.p; F.u_row = M->d_u_row.p; F.upos_of = M->d_upos_of.p;
  F.l_lev = M->d_l_lev.p; F.u_lev = M->d_u_lev.p;
  F.l_col16 = M->d_l_col16.p; F.u_col16 = M->d_u_col16.p; F.u_row16 = M->d_u_row16.p; F.rowmap16 = M->d_rowmap16.p;
  F.l_val = M->l_val.p; F.u_val = M->u_val.p; F.dinv = M->dinv.p;
  F.send_ptr = nullptr; F.send_local = nullptr; F.send_slot = nullptr; F.send_buf = nullptr; F.send_dst = nullptr;
  F.chunk_ptr = M->d_chunk_ptr.p; F.jf_desc = M->d_jf_desc.p; F.jb_desc = M->d_jb_desc.p;
  F.jf_row = M->d_jf_row.p; F.jb_row = M->d_jb_row.p; F.jl_col = M->d_jl_col.p; F.ju_col = M->d_ju_col.p;
  F.kap = M->jkap.p; F.dinv_f = M->jdinv_f.p; F.jb_dev = M->d_jb_dev.p; F.jf_dev = M->d_jf_dev.p;
  return F;
}

constexpr size_t LDS_CAP_BYTES = 64 * 1024;

}  // namespace

// --------------------------------------------------------------------------------------------------------------
// symbolic phase (host): fixed_block / diagonal_block restated for the device layout (ilu0.jl:13-81)
// --------------------------------------------------------------------------------------------------------------
extern "C" int32_t jh_ilu0_create(jh_csr A, const int64_t *partition, int64_t nparts, jh_ilu *out) {
  return guard([&] {
    if (!A || !out) JH_THROW("null argument");
    const Pattern &P = *A->pat;
    const int64_t n = P.n;
    for (int64_t i = 0; i < n; ++i)
      if (P.diag[i] < 0) JH_THROW("Diagonal must be present in sparsity pattern.");
    auto M = std::make_unique<jh_ilu_s>();
    M->ctx = A->ctx; M->A = A; M->pat = A->pat; M->bs = P.bs; M->n = n;
    jh::DeviceScope dev(M->ctx);  // (planning contexts: the symbolic phase and the layouts, nothing allocated or uploaded)
    const bool timing = M->ctx->opt.setup_timing != 0;
    auto tlast = std::chrono::steady_clock::now();
    double mblast = timing ? setup_faulted_mb() : 0.0;
    auto lap = [&](const char *what) {
      if (!timing) return;
      auto nw = std::chrono::steady_clock::now();
      const double mb = setup_faulted_mb();
      fprintf(stderr, "[jutul_hip setup] ilu0: %-22s %.3f s  %6.0f MB first-touched\n", what, std::chrono::duration<double>(nw - tlast).count(), mb - mblast);
      tlast = nw;
      mblast = mb;
    };
    // part id per device row
    std::vector<int32_t> part;
    assign_prefaulted(part, (size_t)n, (int32_t)0);
    int64_t np = 1;
    if (partition) {
      np = 0;
      for (int64_t i = 0; i < n; ++i) {
        int64_t h = P.perm.empty() ? i : P.perm[i];
        if (partition[h] < 1) JH_THROW("partition ids must be >= 1 (par_ilu0.jl:49)");
        part[i] = (int32_t)(partition[h] - 1);
        np = std::max<int64_t>(np, partition[h]);
      }
    } else if (nparts == -1) {
      if (P.block_ptr.size() < 2) JH_THROW("discretisation has no device blocks (use JH_REORDER_BLOCKS or a partition)");
      np = (int64_t)P.block_ptr.size() - 1;
      parallel_ranges(np, 64, [&](int64_t p0, int64_t p1) {
        for (int64_t b = p0; b < p1; ++b)
          for (int32_t i = P.block_ptr[b]; i < P.block_ptr[b + 1]; ++i) part[i] = (int32_t)b;
      });
    } else if (nparts > 1) {
      JH_THROW("nparts > 1 requires an explicit partition vector (generate_lookup, partitioning.jl:25-27)");
    }
    M->nparts = np;
    const bool device_blocks = !partition && nparts == -1;  // the parts are the device blocks: consecutive device rows, in order
    std::vector<int64_t> psize(np, 0);
    if (device_blocks) for (int64_t b = 0; b < np; ++b) psize[b] = P.block_ptr[b + 1] - P.block_ptr[b];
    else for (int64_t i = 0; i < n; ++i) psize[part[i]]++;
    int64_t maxrows = 0;
    for (auto s : psize) { if (s == 0) JH_THROW("empty block in partition (partitioning.jl:47)"); maxrows = std::max(maxrows, s); }
    M->max_block_rows = maxrows;
    M->lds_mode = (np > 1) && ((size_t)maxrows * P.bs * sizeof(double) <= LDS_CAP_BYTES);

    // The parts are independent from here on: everything below runs part by part (block by block) on all host cores.
    // rows of every part in ascending device-row order (counting sort)
    std::vector<int64_t> pstart(np + 1, 0);
    for (int64_t b = 0; b < np; ++b) pstart[b + 1] = pstart[b] + psize[b];
    std::vector<int32_t> prow;
    resize_parallel(prow, (size_t)n);
    if (device_blocks) {  // (the counting sort below is the identity here)
      parallel_ranges(n, 1 << 18, [&](int64_t b, int64_t e) { std::iota(prow.begin() + b, prow.begin() + e, (int32_t)b); });
    } else {
      std::vector<int64_t> cur(pstart.begin(), pstart.end() - 1);
      for (int64_t i = 0; i < n; ++i) prow[cur[part[i]]++] = (int32_t)i;
    }
    // dependency levels inside each part (forward: strict-lower entries; backward: strict-upper entries)
    std::vector<int32_t> flev, blev;
    assign_prefaulted(flev, (size_t)n, (int32_t)0);
    assign_prefaulted(blev, (size_t)n, (int32_t)0);
    parallel_ranges(np, 16, [&](int64_t p0, int64_t p1) {
      for (int64_t p = p0; p < p1; ++p) {
        for (int64_t r = pstart[p]; r < pstart[p + 1]; ++r) {
          const int32_t i = prow[r];
          int32_t lv = 0;
          for (int32_t k = P.rowptr[i]; k < P.diag[i]; ++k) {
            const int32_t c = P.col[k];
            if (part[c] == p) lv = std::max(lv, flev[c] + 1);
          }
          flev[i] = lv;
        }
        for (int64_t r = pstart[p + 1] - 1; r >= pstart[p]; --r) {
          const int32_t i = prow[r];
          int32_t lv = 0;
          for (int32_t k = P.diag[i] + 1; k < P.rowptr[i + 1]; ++k) {
            const int32_t c = P.col[k];
            if (part[c] == p) lv = std::max(lv, blev[c] + 1);
          }
          blev[i] = lv;
        }
      }
    });
    lap("levels");
    // execution blocks: LDS mode one per part, GLOBAL mode a single one; brow = their rows in ascending device-row order
    const bool lds = M->lds_mode;
    if (lds) {
      M->blk_ptr.assign(np + 1, 0);
      for (int64_t b = 0; b < np; ++b) M->blk_ptr[b + 1] = (int32_t)pstart[b + 1];
    } else {
      M->blk_ptr = {0, (int32_t)n};
      std::iota(prow.begin(), prow.end(), 0);
    }
    const std::vector<int32_t> &brow = prow;
    const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
    // levels per block (every level 0 .. max occurs: a row of level l depends on one of level l - 1)
    std::vector<int32_t> nflev(nb), nblev(nb);
    parallel_ranges(nb, 16, [&](int64_t b0, int64_t b1) {
      for (int64_t b = b0; b < b1; ++b) {
        int32_t mf = 0, mb = 0;
        for (int32_t r = M->blk_ptr[b]; r < M->blk_ptr[b + 1]; ++r) { mf = std::max(mf, flev[brow[r]]); mb = std::max(mb, blev[brow[r]]); }
        nflev[b] = mf + 1;
        nblev[b] = mb + 1;
      }
    });
    // level pointers: block b owns flev_ptr[flev_off[b] .. flev_off[b+1]) = level starts + one end sentinel,
    // i.e. level l of block b spans ilu rows [flev_ptr[flev_off[b]+l], flev_ptr[flev_off[b]+l+1]); blev_* likewise over U-order
    // positions
    int64_t maxlev = 0;
    M->flev_off.assign(nb + 1, 0);
    M->blev_off.assign(nb + 1, 0);
    for (int64_t b = 0; b < nb; ++b) {
      M->flev_off[b + 1] = M->flev_off[b] + nflev[b] + 1;
      M->blev_off[b + 1] = M->blev_off[b] + nblev[b] + 1;
      maxlev = std::max<int64_t>(maxlev, std::max(nflev[b], nblev[b]));
    }
    M->flev_ptr.assign(M->flev_off[nb], 0);
    M->blev_ptr.assign(M->blev_off[nb], 0);
    M->max_levels = maxlev;
    // ilu ordering: rows of a block by (flev, device row); U-order: its ilu rows by (blev, ilu row) -- two counting sorts
    std::vector<int32_t> order, ilu_of, uord;
    resize_parallel(order, (size_t)n); resize_parallel(ilu_of, (size_t)n); resize_parallel(uord, (size_t)n);
    resize_parallel(M->upos_of, (size_t)n);
    resize_parallel(M->u_row, (size_t)n);
    parallel_ranges(nb, 16, [&](int64_t b0, int64_t b1) {
      std::vector<int32_t> cnt;
      for (int64_t b = b0; b < b1; ++b) {
        const int32_t r0 = M->blk_ptr[b], r1 = M->blk_ptr[b + 1];
        int32_t *fp = M->flev_ptr.data() + M->flev_off[b], *bp = M->blev_ptr.data() + M->blev_off[b];
        cnt.assign((size_t)nflev[b] + 1, 0);
        for (int32_t r = r0; r < r1; ++r) cnt[flev[brow[r]] + 1]++;
        for (int32_t l = 0; l < nflev[b]; ++l) cnt[l + 1] += cnt[l];
        for (int32_t l = 0; l <= nflev[b]; ++l) fp[l] = r0 + cnt[l];  // (the last one is the end sentinel r1)
        for (int32_t r = r0; r < r1; ++r) {
          const int32_t i = brow[r], t = r0 + cnt[flev[i]]++;
          order[t] = i;
          ilu_of[i] = t;
        }
        cnt.assign((size_t)nblev[b] + 1, 0);
        for (int32_t t = r0; t < r1; ++t) cnt[blev[order[t]] + 1]++;
        for (int32_t l = 0; l < nblev[b]; ++l) cnt[l + 1] += cnt[l];
        for (int32_t l = 0; l <= nblev[b]; ++l) bp[l] = r0 + cnt[l];
        for (int32_t t = r0; t < r1; ++t) {
          const int32_t pos = r0 + cnt[blev[order[t]]]++;
          uord[pos] = t;
          M->upos_of[t] = pos;
          M->u_row[pos] = t - r0;  // local id of the row inside its execution block
        }
      }
    });
    M->rowmap = order;
    lap("row order, level pointers");
    // L (forward order) and U (backward order) storage: count, prefix, fill
    assign_prefaulted(M->l_ptr, (size_t)n + 1, (int32_t)0);
    assign_prefaulted(M->u_ptr, (size_t)n + 1, (int32_t)0);
    resize_parallel(M->d_map, (size_t)n);
    parallel_ranges(n, 1 << 16, [&](int64_t t0, int64_t t1) {
      for (int64_t t = t0; t < t1; ++t) {
        const int32_t i = order[t], j = order[uord[t]];
        int32_t nl = 0, nu = 0;
        for (int32_t k = P.rowptr[i]; k < P.diag[i]; ++k)  // (shadow slots of a multigraph pattern hold no value: Pattern::shadow_slots)
          nl += part[P.col[k]] == part[i] && !P.is_shadow(k);
        for (int32_t k = P.diag[j] + 1; k < P.rowptr[j + 1]; ++k) nu += part[P.col[k]] == part[j] && !P.is_shadow(k);
        M->l_ptr[t + 1] = nl;
        M->u_ptr[t + 1] = nu;
      }
    });
    {
      int64_t sl = 0, su = 0;
      for (int64_t t = 0; t < n; ++t) { sl += M->l_ptr[t + 1]; su += M->u_ptr[t + 1]; }
      if (sl > INT32_MAX || su > INT32_MAX) JH_THROW("ILU(0): more than 2^31 - 1 entries in a factor");
    }
    for (int64_t t = 0; t < n; ++t) { M->l_ptr[t + 1] += M->l_ptr[t]; M->u_ptr[t + 1] += M->u_ptr[t]; }
    resize_parallel(M->l_col, (size_t)M->l_ptr[n]);
    resize_parallel(M->l_map, (size_t)M->l_ptr[n]);
    resize_parallel(M->u_col, (size_t)M->u_ptr[n]);
    resize_parallel(M->u_map, (size_t)M->u_ptr[n]);
    parallel_ranges(n, 1 << 16, [&](int64_t t0, int64_t t1) {
      for (int64_t t = t0; t < t1; ++t) {
        {
          const int32_t i = order[t], b0 = M->blk_ptr[lds ? part[i] : 0];
          int32_t w = M->l_ptr[t];
          for (int32_t k = P.rowptr[i]; k < P.diag[i]; ++k)
            if (part[P.col[k]] == part[i] && !P.is_shadow(k)) { M->l_col[w] = ilu_of[P.col[k]] - b0; M->l_map[w++] = k; }
        }
        {
          const int64_t pos = t;
          const int32_t i = order[uord[pos]], b0 = M->blk_ptr[lds ? part[i] : 0];
          int32_t w = M->u_ptr[pos];
          M->d_map[pos] = P.diag[i];
          for (int32_t k = P.diag[i] + 1; k < P.rowptr[i + 1]; ++k)
            if (part[P.col[k]] == part[i] && !P.is_shadow(k)) { M->u_col[w] = ilu_of[P.col[k]] - b0; M->u_map[w++] = k; }
        }
      }
    });
    lap("L / U entries");
    if (lds) {
      int64_t mxl = 0, mxu = 0;
      for (int64_t b = 0; b < nb; ++b) {
        mxl = std::max<int64_t>(mxl, M->l_ptr[M->blk_ptr[b + 1]] - M->l_ptr[M->blk_ptr[b]]);
        mxu = std::max<int64_t>(mxu, M->u_ptr[M->blk_ptr[b + 1]] - M->u_ptr[M->blk_ptr[b]]);
      }
      const size_t bbv = (size_t)P.bs * P.bs;
      size_t bytes = sizeof(double) * bbv * (size_t)(mxl + mxu + maxrows) +
                     sizeof(uint16_t) * (size_t)(mxl + mxu + 3 * maxrows + 2 + maxlev + 2);
      bytes = (bytes + 15) & ~(size_t)15;
      // (not beyond 64 KB: with the 160 KB opt-in the LDS row kernel took 151 ms on 512-row blocks of a 2M-cell polyhedral grid,
      // the global-memory row kernels 33 ms; long rows take the program-driven kernel instead, below)
      if (mxl < 65536 && mxu < 65536 && maxrows < 65536 && bytes <= LDS_CAP_BYTES && !M->ctx->opt.ilu_factor_global) {
        M->max_blk_l = (int)mxl;
        M->max_blk_u = (int)mxu;
        M->factor_lds_bytes = bytes;
      }
    }
    resize_parallel(M->l_lev, (size_t)n);
    resize_parallel(M->u_lev, (size_t)n);
    if (maxlev >= 65536) JH_THROW("more than 65535 dependency levels");
    parallel_ranges(n, 1 << 16, [&](int64_t b, int64_t e) {
      for (int64_t t = b; t < e; ++t) { M->l_lev[t] = (uint16_t)flev[order[t]]; M->u_lev[t] = (uint16_t)blev[order[uord[t]]]; }
    });
    // fused halo pack: (block, local row) of every send row of the discretisation's halo plan
    if (lds && A->disc && A->disc->halo.active && A->disc->halo.n_send > 0) {
      const auto &H = A->disc->halo;
      std::vector<std::vector<std::pair<int32_t, int32_t>>> per(nb);
      std::vector<int32_t> blk_of(n);
      for (int64_t b = 0; b < nb; ++b)
        for (int32_t t = M->blk_ptr[b]; t < M->blk_ptr[b + 1]; ++t) blk_of[t] = (int32_t)b;
      for (int64_t i = 0; i < H.n_send; ++i) {
        const int32_t t = ilu_of[H.send_idx_host[i]];
        const int32_t b = blk_of[t];
        per[b].push_back({t - M->blk_ptr[b], (int32_t)i});
      }
      M->send_ptr.assign(nb + 1, 0);
      for (int64_t b = 0; b < nb; ++b) {
        M->send_ptr[b + 1] = M->send_ptr[b] + (int32_t)per[b].size();
        for (auto &pr : per[b]) { M->send_local.push_back(pr.first); M->send_slot.push_back(pr.second); }
      }
      M->halo_epoch = H.epoch;
    }
    lap("sizes, halo lists");
    // one wavefront per small block keeps many blocks resident per CU (the level loop is latency-bound)
    M->threads = maxrows <= 1024 ? 64 : (maxrows <= 2048 ? 128 : 256);
    { const int t = (int)M->ctx->opt.ilu_threads; if (t == 64 || t == 128 || t == 256 || t == 512) M->threads = t; }
    // ---- chunk-jagged layout (ilu_apply_jds_kernel): per sweep and 64-lane chunk, lanes sorted by entry count, entries stored
    // diagonal by diagonal.  Needs block-local 16-bit ids; a lane carries at most 8 entries.
    // VIRTUAL ROWS (round 4): a row with more than 8 strict-L (strict-U) entries -- polyhedral / PEBI cells -- becomes a chain of
    // lanes of at most 8 entries each, in column order.  Every lane of the chain reads the row's entry of the block vector when its
    // own level comes up, subtracts its entries and writes the partial result back; the chain's last lane is the row's result
    // (backward: times the inverted pivot; the other lanes carry a unit pivot).  A lane's level = 1 + the latest of its operands'
    // final levels and of its predecessor in the chain, so the arithmetic is the row's left-to-right sum (mat.jl / ilu0.jl:156-187
    // order): the same bits as the row-major sweep.  Rows of at most 8 entries are chains of one lane with their old level.
    struct VRow { int32_t p, first, level; uint8_t n, final; };
    int maxcnt = 0;
    for (int64_t t = 0; t < n; ++t) maxcnt = std::max({maxcnt, M->l_ptr[t + 1] - M->l_ptr[t], M->u_ptr[t + 1] - M->u_ptr[t]});
    const bool VR = maxcnt > 8;
    // lanes of block b in sweep order, sorted by level (stable); done: scratch of the block's row count
    auto build_vrows = [&](int64_t b, int sweep, std::vector<VRow> &out, std::vector<int32_t> &done) {
      const int32_t b0 = M->blk_ptr[b], b1 = M->blk_ptr[b + 1];
      const std::vector<int32_t> &ptr = sweep ? M->u_ptr : M->l_ptr, &ocol = sweep ? M->u_col : M->l_col;
      out.clear();
      if (!VR) {
        for (int32_t p = b0; p < b1; ++p)
          out.push_back(VRow{p, ptr[p], sweep ? (int32_t)M->u_lev[p] : (int32_t)M->l_lev[p], (uint8_t)(ptr[p + 1] - ptr[p]), 1});
        return;
      }
      done.assign((size_t)(b1 - b0), 0);
      for (int32_t p = b0; p < b1; ++p) {  // sweep order: the operands of a row come before it
        const int32_t lt = sweep ? M->u_row[p] : p - b0, cnt = ptr[p + 1] - ptr[p];
        if (cnt == 0) { out.push_back(VRow{p, ptr[p], 0, 0, 1}); done[lt] = 0; continue; }
        int32_t lprev = -1;
        for (int32_t f = 0; f < cnt; f += 8) {
          const int32_t m = std::min(8, cnt - f);
          int32_t lv = lprev + 1;
          for (int32_t e = 0; e < m; ++e) lv = std::max(lv, done[ocol[ptr[p] + f + e]] + 1);
          out.push_back(VRow{p, ptr[p] + f, lv, (uint8_t)m, (uint8_t)(f + m == cnt)});
          lprev = lv;
        }
        done[lt] = lprev;
      }
      std::stable_sort(out.begin(), out.end(), [](const VRow &a, const VRow &c) { return a.level < c.level; });
    };
    if (lds && M->threads == 64 && maxrows < 65536 && maxlev < 0xffff && M->ctx->opt.ilu_jagged) {
      bool jag_ok = true;
      {
        M->jag = true;
        M->jag_vr = VR;
        M->jag_ku = maxcnt <= 4 ? 4 : 8;
        M->chunk_ptr.assign(nb + 1, 0);
        if (!VR) {
          for (int64_t b = 0; b < nb; ++b) M->chunk_ptr[b + 1] = (M->blk_ptr[b + 1] - M->blk_ptr[b] + 63) / 64;
        } else {  // the longer of the two sweeps' lane lists decides (the shorter one is padded with empty lanes)
          parallel_ranges(nb, 16, [&](int64_t bb0, int64_t bb1) {
            std::vector<VRow> v;
            std::vector<int32_t> done;
            for (int64_t b = bb0; b < bb1; ++b) {
              size_t m = 0;
              for (int sweep = 0; sweep < 2; ++sweep) { build_vrows(b, sweep, v, done); m = std::max(m, v.size()); }
              M->chunk_ptr[b + 1] = (int32_t)((m + 63) / 64);
            }
          });
        }
        for (int64_t b = 0; b < nb; ++b) M->chunk_ptr[b + 1] += M->chunk_ptr[b];
        const int64_t nchunks = M->chunk_ptr[nb];
        M->j_nslots = nchunks * 64;
        std::vector<int4> fdesc(nchunks + 4, make_int4(0, 0, 0, 0)), bdesc(nchunks + 4, make_int4(0, 0, 0, 0));  // + 4: read ahead
        std::vector<uint32_t> frow, brow;
        assign_prefaulted(frow, (size_t)M->j_nslots, 0xffff0000u);
        assign_prefaulted(brow, (size_t)M->j_nslots, 0xffff0000u);
        // the jagged order permutes the entries inside a block: block b keeps the ranges [l_ptr[b0], l_ptr[b1]) / [u_ptr[b0], u_ptr[b1]),
        // so the blocks can be laid out independently -- on all host cores
        const size_t nlent = M->l_col.size(), nuent = M->u_col.size();
        std::vector<uint16_t> jlc, juc;  // + 64: lanes past a diagonal's count load too
        assign_prefaulted(jlc, nlent + 64, (uint16_t)0);
        assign_prefaulted(juc, nuent + 64, (uint16_t)0);
        assign_prefaulted(M->jl_map, nlent, (int32_t)0); assign_prefaulted(M->ju_map, nuent, (int32_t)0);
        assign_prefaulted(M->jd_map, (size_t)M->j_nslots, (int32_t)-1);
        assign_prefaulted(M->jl_of_old, nlent, (int32_t)-1); assign_prefaulted(M->ju_of_old, nuent, (int32_t)-1); assign_prefaulted(M->jd_of_old, (size_t)n, (int32_t)-1);
        M->blk_lbase.assign(nb + 1, 0); M->blk_ubase.assign(nb + 1, 0); M->blk_prog.assign(nb + 1, 0);
        for (int64_t b = 0; b <= nb; ++b) { M->blk_lbase[b] = M->l_ptr[M->blk_ptr[b]]; M->blk_ubase[b] = M->u_ptr[M->blk_ptr[b]]; }
        lap("  jagged: arrays");
        std::vector<std::vector<uint16_t>> progs(nb);  // factorisation program of every block (ilu_factor_prog_kernel)
        // long rows: the rows form of the programs (ilu_factor_rows_kernel; ilu_factor_wave_per_row = 2 keeps the instruction form).
        // Scalar matrices only: the entries of a block of 2x2 cells take 80 KB of LDS, two blocks per CU either way, and the
        // instruction form is the faster one there (1.50 against 1.67 ms at 1M polyhedral two-phase cells).
        const bool want_rows = VR && P.bs == 1 && M->ctx->opt.ilu_factor_wave_per_row == 1;
        std::vector<std::vector<uint16_t>> progs_rows(want_rows ? nb : 0);
        std::vector<char> rows_ok(nb, 0);
        std::vector<int> blk_vals(nb, 0);
        std::vector<char> blk_ok(nb, 1), blk_diag(nb, 1);  // blk_diag: every update of the block targets a pivot
        std::vector<int32_t> jt_map, jf_diag;
        assign_prefaulted(jt_map, nlent, (int32_t)-1);
        assign_prefaulted(jf_diag, (size_t)M->j_nslots, (int32_t)-1);
        std::vector<uint16_t> jf_bslot;
        assign_prefaulted(jf_bslot, (size_t)M->j_nslots, (uint16_t)0);
        // The programs are what the program-driven kernels read; the pivot-only kernels (every update of every block lands on a
        // pivot: triangle-free block patterns -- the lattice, Cartesian and most TPFA grids) never do.  Where the static conditions of
        // those kernels hold, a block's program is therefore generated (its size and whether it is pivot-only are needed) but not
        // kept -- 98M words = 200 MB at 10M cells, twice with the concatenated copy, plus the upload and the HBM -- until a block turns
        // out not to be pivot-only; the few blocks that were processed before that are generated again afterwards.
        int mxc_pre = 0;
        for (int64_t b = 0; b < nb; ++b) mxc_pre = std::max(mxc_pre, M->chunk_ptr[b + 1] - M->chunk_ptr[b]);
        const bool expect_diag = !VR && mxc_pre * 64 <= 1024 && sizeof(double) * (size_t)M->max_block_rows * M->bs * M->bs <= 64 * 1024 &&
                                 M->ctx->opt.ilu_diag_factor != 0;
        std::atomic<bool> keep_programs{!expect_diag};
        std::vector<size_t> prog_words(nb, 0);   // length of every block's (instruction-form) program, kept or not
        auto run_blocks = [&](const std::vector<int64_t> *list) {
        parallel_ranges(list ? (int64_t)list->size() : nb, 16, [&](int64_t bb0, int64_t bb1) {
          int lanes[64];
          std::vector<uint16_t> code, tprog;
          std::vector<VRow> vrows[2];
          std::vector<int32_t> done;
          for (int64_t bi = bb0; bi < bb1; ++bi) {
            const int64_t b = list ? (*list)[(size_t)bi] : bi;
            const int32_t b0 = M->blk_ptr[b], b1 = M->blk_ptr[b + 1], nrb = b1 - b0;
            int32_t wpos[2] = {M->blk_lbase[b], M->blk_ubase[b]};  // running jagged position of the forward / backward sweep
            for (int sweep = 0; sweep < 2; ++sweep) build_vrows(b, sweep, vrows[sweep], done);
            for (int32_t c = 0; c < M->chunk_ptr[b + 1] - M->chunk_ptr[b]; ++c) {
              const int64_t ch = M->chunk_ptr[b] + c;
              for (int sweep = 0; sweep < 2; ++sweep) {
                const std::vector<VRow> &vr = vrows[sweep];
                const int32_t v0 = 64 * c, nrc = std::max(0, std::min<int32_t>(64, (int32_t)vr.size() - v0));
                std::iota(lanes, lanes + nrc, 0);
                std::stable_sort(lanes, lanes + nrc, [&](int a, int c2) { return vr[v0 + a].n > vr[v0 + c2].n; });
                std::vector<uint16_t> &jc = sweep ? juc : jlc;
                std::vector<int32_t> &jm = sweep ? M->ju_map : M->jl_map;
                std::vector<int32_t> &jold = sweep ? M->ju_of_old : M->jl_of_old;
                const std::vector<int32_t> &ocol = sweep ? M->u_col : M->l_col, &omap = sweep ? M->u_map : M->l_map;
                int32_t &w = wpos[sweep];
                int4 D;
                D.x = (int)w;
                unsigned cw[2] = {0, 0};
                for (int j = 0; j < 8; ++j) {
                  int cn = 0;
                  for (int l = 0; l < nrc; ++l) {
                    const VRow &v = vr[v0 + lanes[l]];
                    if ((int)v.n <= j) break;  // sorted by count
                    const int32_t old = v.first + j;
                    jold[old] = w;
                    jc[w] = (uint16_t)ocol[old];
                    jm[w] = omap[old];
                    ++w;
                    ++cn;
                  }
                  cw[j >> 2] |= (unsigned)cn << (8 * (j & 3));
                }
                D.y = (int)cw[0]; D.z = (int)cw[1];
                int lo = 0xffff, hi = 0;
                for (int l = 0; l < nrc; ++l) {
                  const VRow &v = vr[v0 + lanes[l]];
                  const int32_t p = v.p;  // ilu row (forward) or U-order position (backward)
                  const int lv = v.level;
                  if (lv >= 0xffff) { blk_ok[b] = 0; continue; }
                  const int lt = sweep ? M->u_row[p] : p - b0;
                  (sweep ? brow : frow)[ch * 64 + l] = (uint32_t)lt | ((uint32_t)lv << 16);
                  lo = std::min(lo, lv); hi = std::max(hi, lv);
                  if (sweep) {
                    // the pivot slot of a row = the backward lane that finishes it; the other lanes of a chain carry a unit pivot (-2)
                    if (v.final) { M->jd_map[ch * 64 + l] = M->d_map[p]; M->jd_of_old[p] = (int32_t)(ch * 64 + l); }
                    else M->jd_map[ch * 64 + l] = -2;
                  }
                }
                D.w = (int)((unsigned)lo | ((unsigned)hi << 16));
                (sweep ? bdesc : fdesc)[ch] = D;
              }
            }
            // ---- factorisation program: the update pairs of the IKJ elimination (ilu0.jl:108-144), found once here ----------
            const int32_t l0 = M->blk_lbase[b], nl = M->blk_lbase[b + 1] - l0, u0 = M->blk_ubase[b], nu = M->blk_ubase[b + 1] - u0;
            const int32_t dslot0 = M->chunk_ptr[b] * 64, nd = (M->chunk_ptr[b + 1] - M->chunk_ptr[b]) * 64;
            blk_vals[b] = nl + nu + nd;
            if (nl + nu + nd >= 65536) { blk_ok[b] = 0; continue; }
            std::vector<uint16_t> &prog = tprog;
            prog.assign(2 * (size_t)nrb + 1, 0);  // row offsets (nrb + 1), pivot indices (nrb)
            code.clear();
            auto didx = [&](int32_t t) { return (uint16_t)(nl + nu + (M->jd_of_old[M->upos_of[t]] - dslot0)); };  // t: ilu row
            for (int32_t c = 0; c < M->chunk_ptr[b + 1] - M->chunk_ptr[b] && !VR; ++c)  // pivot slot / backward lane of every forward lane (pivot-only kernels)
              for (int l = 0; l < 64; ++l) {
                const size_t fs = (size_t)(M->chunk_ptr[b] + c) * 64 + l;
                if ((frow[fs] >> 16) == 0xffffu) continue;
                const int32_t ip = M->upos_of[b0 + (int32_t)(frow[fs] & 0xffffu)];
                jf_diag[fs] = M->d_map[ip];
                jf_bslot[fs] = (uint16_t)(M->jd_of_old[ip] - dslot0);
              }
            for (int32_t lt = 0; lt < nrb && blk_ok[b]; ++lt) {
              const int32_t t = b0 + lt, ipos = M->upos_of[t];
              prog[lt] = (uint16_t)code.size();
              prog[nrb + 1 + lt] = didx(t);
              const int32_t ls = M->l_ptr[t], le = M->l_ptr[t + 1], us = M->u_ptr[ipos], ue = M->u_ptr[ipos + 1];
              for (int32_t p = ls; p < le; ++p) {
                const int32_t k = M->l_col[p], kpos = M->upos_of[b0 + k];
                code.push_back((uint16_t)(M->jl_of_old[p] - l0));
                code.push_back(didx(b0 + k));
                const size_t cnt_at = code.size();
                code.push_back(0);
                uint16_t nupd = 0;
                for (int32_t q = M->u_ptr[kpos]; q < M->u_ptr[kpos + 1]; ++q) {  // U row k, ascending columns
                  const int32_t j = M->u_col[q];
                  int32_t tgt = -1;
                  if (j == lt) {
                    tgt = didx(t);
                    jt_map[M->jl_of_old[p]] = M->u_map[q];  // the A slot of (k, i), partner of the L entry (i, k)
                  } else if (j < lt) {  // a later strict-L entry of this row (process_partial_row! on rem_l_pos, ilu0.jl:100-106)
                    for (int32_t p2 = p + 1; p2 < le; ++p2) if (M->l_col[p2] == j) { tgt = M->jl_of_old[p2] - l0; break; }
                  } else {
                    for (int32_t qi = us; qi < ue; ++qi) if (M->u_col[qi] == j) { tgt = nl + (M->ju_of_old[qi] - u0); break; }
                  }
                  if (tgt < 0) continue;  // (k, j) has no counterpart in row i: dropped fill, ILU(0)
                  if (j != lt) blk_diag[b] = 0;
                  code.push_back((uint16_t)tgt);
                  code.push_back((uint16_t)(nl + (M->ju_of_old[q] - u0)));
                  ++nupd;
                }
                code[cnt_at] = nupd;
              }
              if (code.size() >= 65536) blk_ok[b] = 0;
            }
            prog[nrb] = (uint16_t)code.size();
            prog.insert(prog.end(), code.begin(), code.end());
            prog_words[b] = prog.size();
            if (!blk_ok[b] || !blk_diag[b]) keep_programs.store(true, std::memory_order_relaxed);
            if (list || keep_programs.load(std::memory_order_relaxed)) progs[b] = prog;
            if (want_rows && blk_ok[b])
              rows_ok[b] = prog_rows_form(prog, nrb, nl + nu + nd, M->flev_ptr.data() + M->flev_off[b],
                                          M->flev_off[b + 1] - M->flev_off[b] - 1, b0, P.bs == 1 ? 8 : (P.bs == 2 ? 4 : 2), progs_rows[b]);
          }
        });
        };
        run_blocks(nullptr);
        lap("  jagged: blocks (layout + programs)");
        {
          bool ok = true;
          int max_vals = 0, max_words = 0;
          size_t total = 0;
          bool rows = want_rows;
          size_t rows_total = 0;  // (32-bit program offsets: beyond them the instruction form, half the size, is kept)
          for (int64_t b = 0; b < nb && rows; ++b) { rows = rows_ok[b] != 0; rows_total += progs_rows[b].size(); }
          rows = rows && rows_total < (size_t)INT32_MAX;
          if (rows) progs.swap(progs_rows);
          M->prog_rows = rows;
          for (int64_t b = 0; b < nb; ++b) {
            ok = ok && blk_ok[b];
            max_vals = std::max(max_vals, blk_vals[b]);
            // LDS words: the whole program, or (rows form: the rows are read from global memory) its head
            max_words = std::max<int>(max_words, rows ? 2 + (M->flev_off[b + 1] - M->flev_off[b] - 1) + 4 * (M->blk_ptr[b + 1] - M->blk_ptr[b])
                                                      : (int)prog_words[b]);
            M->blk_prog[b] = (int32_t)total;
            total += rows ? progs[b].size() : prog_words[b];
          }
          M->blk_prog[nb] = (int32_t)total;
          {
            bool all_diag = ok && !VR;  // (the pivot-only kernels take one lane per row)
            int mxc = 0;
            for (int64_t b = 0; b < nb; ++b) { all_diag = all_diag && blk_diag[b]; mxc = std::max(mxc, M->chunk_ptr[b + 1] - M->chunk_ptr[b]); }
            M->max_chunks = mxc;
            // the sweep kernel keeps the block's inverted pivots in dynamic LDS (8 * rows * bs^2 bytes): blocks beyond the 64 KB a launch
            // gets without an opt-in (e.g. > 910 rows of 3x3 blocks from a caller's partition) take the program kernel
            M->diag_only = all_diag && mxc * 64 <= 1024 && sizeof(double) * (size_t)M->max_block_rows * M->bs * M->bs <= 64 * 1024 &&
                           M->ctx->opt.ilu_diag_factor != 0;
            if (M->diag_only) {
              hipStream_t sd = M->ctx->stream;
              M->d_jt_map.upload(jt_map, sd); M->d_jf_diag.upload(jf_diag, sd); M->d_jf_bslot.upload(jf_bslot, sd);
              M->d_blk_ubase.upload(M->blk_ubase, sd);
            }
            // D-ILU storage + fused product: OPT-IN (JH_FUSED_PRODUCT=1).  Measured on MI355X, 10M cells (profiles/r03_fused_product_*):
            // apply + third pass 0.334 ms, out-of-block launch 0.108 ms = 0.442 ms against 0.202 + 0.176 = 0.379 ms for apply +
            // jagged SpMV.  The factor / matrix ENTRIES read twice today (0.41 GB) are saved, but the per-row streams the fusion
            // adds -- A_ii, the second copy of the pivots, the partial product written and read back, dot weights and q touched
            // again by the out-of-block rows at sector granularity -- cost as much: 1.8 GB either way.
            M->uscaled = M->diag_only && M->blk_ptr.size() > 1 && M->ctx->opt.fused_product != 0;
            if (M->uscaled) {
              // device row (relative to its block) of every forward / backward lane, flagged when the row has entries outside the block;
              // the list of those entries ("E": what ilu_eprod_kernel adds after the apply)
              std::vector<uint16_t> jb_dev(M->j_nslots, 0), jf_dev(M->j_nslots, 0);
              std::vector<char> has_e(n, 0);
              std::vector<int32_t> ecount(n + 1, 0);
              parallel_ranges(nb, 16, [&](int64_t bb0, int64_t bb1) {
                for (int64_t b = bb0; b < bb1; ++b) {
                  const int32_t b0 = M->blk_ptr[b], b1 = M->blk_ptr[b + 1];
                  for (int32_t i = b0; i < b1; ++i) {  // device rows of the block (rowmap_local: the block's rows are [b0, b1))
                    int32_t ce = 0;
                    for (int32_t k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) ce += (P.col[k] < b0 || P.col[k] >= b1) && !P.is_shadow(k);
                    ecount[i + 1] = ce;
                    has_e[i] = ce > 0;
                  }
                  for (int32_t c = M->chunk_ptr[b]; c < M->chunk_ptr[b + 1]; ++c)
                    for (int l = 0; l < 64; ++l) {
                      const size_t sl = (size_t)c * 64 + l;
                      for (int sweep = 0; sweep < 2; ++sweep) {
                        const uint32_t wd = (sweep ? brow : frow)[sl];
                        if ((wd >> 16) == 0xffffu) continue;
                        const int32_t dev = M->rowmap[b0 + (int32_t)(wd & 0xffffu)];  // ilu row -> device row
                        (sweep ? jb_dev : jf_dev)[sl] = (uint16_t)((dev - b0) | (has_e[dev] ? 0x8000 : 0));
                      }
                    }
                }
              });
              bool local = true;  // (checked again below for every block; the lists assume it)
              for (int64_t b = 0; b < nb && local; ++b)
                for (int32_t t = M->blk_ptr[b]; t < M->blk_ptr[b + 1]; ++t)
                  if (M->rowmap[t] < M->blk_ptr[b] || M->rowmap[t] >= M->blk_ptr[b + 1]) { local = false; break; }
              if (!local || maxrows >= 0x8000) {
                M->uscaled = false;
              } else {
                std::vector<int32_t> e_row, e_ptr(1, 0), e_col, e_slot;
                std::vector<int32_t> blk_of_dev(n);
                for (int64_t b = 0; b < nb; ++b)
                  for (int32_t i = M->blk_ptr[b]; i < M->blk_ptr[b + 1]; ++i) blk_of_dev[i] = (int32_t)b;
                for (int64_t i = 0; i < n; ++i) {
                  if (!has_e[i]) continue;
                  const int32_t b0 = M->blk_ptr[blk_of_dev[i]], b1 = M->blk_ptr[blk_of_dev[i] + 1];
                  e_row.push_back((int32_t)i);
                  for (int32_t k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k)
                    if ((P.col[k] < b0 || P.col[k] >= b1) && !P.is_shadow(k)) { e_col.push_back(P.col[k]); e_slot.push_back(k); }
                  e_ptr.push_back((int32_t)e_col.size());
                }
                M->e_rows = (int64_t)e_row.size();
                M->e_nent = (int64_t)e_col.size();
                hipStream_t su = M->ctx->stream;
                M->d_jb_dev.upload(jb_dev, su); M->d_jf_dev.upload(jf_dev, su);
                if (e_row.empty()) { e_row.push_back(0); e_col.push_back(0); e_slot.push_back(0); }
                M->d_e_row.upload(e_row, su); M->d_e_ptr.upload(e_ptr, su); M->d_e_col.upload(e_col, su); M->d_e_slot.upload(e_slot, su);
                M->jkap.alloc((size_t)M->j_nslots * P.bs * P.bs);
                M->jdinv_f.alloc((size_t)M->j_nslots * P.bs * P.bs);
                M->jkap.zero(su);
                M->jdinv_f.zero(su);
              }
            }
            if (timing) fprintf(stderr, "[jutul_hip setup] ilu0: pivot-only factorisation %s (largest block: %d chunks), fused product %s (%lld rows with %lld out-of-block entries)\n",
                                M->diag_only ? "on" : "off", mxc, M->uscaled ? "on" : "off", (long long)M->e_rows, (long long)M->e_nent);
          }
          size_t bytes = sizeof(double) * P.bs * P.bs * (size_t)max_vals + sizeof(uint16_t) * (size_t)max_words;
          bytes = (bytes + 15) & ~(size_t)15;
          if (ok && total < (size_t)INT32_MAX && bytes <= 160 * 1024 - 512 && M->ctx->opt.ilu_prog) {
            M->prog = true;
            M->prog_lds_bytes = bytes; M->prog_max_vals = max_vals; M->prog_max_words = max_words;
            if (timing) fprintf(stderr, "[jutul_hip setup] ilu0: factor programs: LDS %zu B per block (values %d, words %d), %zu words total, %lld rows in the largest block%s\n",
                                bytes, max_vals, max_words, total, (long long)maxrows, M->diag_only ? " (pivot-only kernels: not kept, not uploaded)" : "");
            if (!M->diag_only) {  // (jh_ilu0_factor takes the pivot-only kernels whenever diag_only holds: nothing reads the programs then)
              if (!rows) {  // blocks whose program was not kept while every block so far had been pivot-only
                std::vector<int64_t> again;
                for (int64_t b = 0; b < nb; ++b) if (prog_words[b] > 0 && progs[b].empty()) again.push_back(b);
                if (!again.empty()) run_blocks(&again);
              }
              std::vector<uint16_t> prog(std::max<size_t>(total, 1), 0);
              parallel_ranges(nb, 64, [&](int64_t bb0, int64_t bb1) {
                for (int64_t b = bb0; b < bb1; ++b)
                  if (!progs[b].empty()) std::copy(progs[b].begin(), progs[b].end(), prog.begin() + M->blk_prog[b]);
              });
              hipStream_t sp = M->ctx->stream;
              M->d_blk_lbase.upload(M->blk_lbase, sp); M->d_blk_ubase.upload(M->blk_ubase, sp); M->d_blk_prog.upload(M->blk_prog, sp);
              M->blk_dbase.assign(nb + 1, 0);
              for (int64_t b = 0; b <= nb; ++b) M->blk_dbase[b] = M->chunk_ptr[b] * 64;
              M->d_blk_dbase.upload(M->blk_dbase, sp);
              M->d_prog.upload(prog, sp);
            }
          }
        }
        lap("  jagged: program concat");
        M->jl_nent = (int64_t)nlent; M->ju_nent = (int64_t)nuent;
        hipStream_t sj = M->ctx->stream;
        M->d_chunk_ptr.upload(M->chunk_ptr, sj);
        M->d_jf_desc.upload(fdesc, sj); M->d_jb_desc.upload(bdesc, sj);
        M->d_jf_row.upload(frow, sj); M->d_jb_row.upload(brow, sj);
        M->d_jl_col.upload(jlc, sj); M->d_ju_col.upload(juc, sj);
        M->d_jl_map.upload(M->jl_map, sj); M->d_ju_map.upload(M->ju_map, sj); M->d_jd_map.upload(M->jd_map, sj);
        // (old position -> jagged position: what ilu_to_jagged permutes by after a row-major factorisation; the program-driven and
        // pivot-only kernels write the jagged arrays themselves)
        if (!M->prog) { M->d_jl_of_old.upload(M->jl_of_old, sj); M->d_ju_of_old.upload(M->ju_of_old, sj); M->d_jd_of_old.upload(M->jd_of_old, sj); }
        const size_t bbj = (size_t)P.bs * P.bs;
        M->jl_val.alloc((size_t)(M->jl_nent + 64) * bbj); M->ju_val.alloc((size_t)(M->ju_nent + 64) * bbj); M->jdinv.alloc((size_t)M->j_nslots * bbj);
        M->jl_val.zero(sj);
        M->ju_val.zero(sj);
        M->jdinv.zero(sj);
        // chains of lanes need the unit pivots the program kernel writes: without a program (a block beyond its 16-bit indices or
        // the LDS) long rows keep the row-major sweeps
        if (VR && !M->prog) { M->jag = false; M->jag_vr = false; jag_ok = false; }
      }
      if (!jag_ok && M->ctx->opt.ilu_prog) {
        // Rows with more than 8 strict-L / strict-U entries (polyhedral / PEBI cells): no jagged layout, the triangular sweeps keep
        // the row-major kernels -- but the refactorisation still runs program-driven (ilu_factor_prog_kernel over the row-major
        // arrays: value index = old entry position, pivot slot = U-order position).  The per-row factor kernels it replaces
        // search the pattern on the device: 33 ms instead of ~1 ms on a 2M-cell polyhedral grid.
        M->blk_lbase.assign(nb + 1, 0); M->blk_ubase.assign(nb + 1, 0); M->blk_prog.assign(nb + 1, 0); M->blk_dbase.assign(nb + 1, 0);
        for (int64_t b = 0; b <= nb; ++b) { M->blk_lbase[b] = M->l_ptr[M->blk_ptr[b]]; M->blk_ubase[b] = M->u_ptr[M->blk_ptr[b]]; M->blk_dbase[b] = M->blk_ptr[b]; }
        std::vector<std::vector<uint16_t>> progs(nb);
        std::vector<int> blk_vals(nb, 0);
        std::vector<char> blk_ok(nb, 1);
        parallel_ranges(nb, 16, [&](int64_t bb0, int64_t bb1) {
          std::vector<uint16_t> code;
          for (int64_t b = bb0; b < bb1; ++b) {
            const int32_t b0 = M->blk_ptr[b], nrb = M->blk_ptr[b + 1] - b0;
            const int32_t l0 = M->blk_lbase[b], nl = M->blk_lbase[b + 1] - l0, u0 = M->blk_ubase[b], nu = M->blk_ubase[b + 1] - u0;
            blk_vals[b] = nl + nu + nrb;
            if (nl + nu + nrb >= 65536) { blk_ok[b] = 0; continue; }
            std::vector<uint16_t> &prog = progs[b];
            prog.assign(2 * (size_t)nrb + 1, 0);  // row offsets (nrb + 1), pivot indices (nrb)
            code.clear();
            auto didx = [&](int32_t t) { return (uint16_t)(nl + nu + (M->upos_of[t] - b0)); };  // t: ilu row
            for (int32_t lt = 0; lt < nrb && blk_ok[b]; ++lt) {
              const int32_t t = b0 + lt, ipos = M->upos_of[t];
              prog[lt] = (uint16_t)code.size();
              prog[nrb + 1 + lt] = didx(t);
              const int32_t ls = M->l_ptr[t], le = M->l_ptr[t + 1], us = M->u_ptr[ipos], ue = M->u_ptr[ipos + 1];
              for (int32_t pp = ls; pp < le; ++pp) {
                const int32_t k = M->l_col[pp], kpos = M->upos_of[b0 + k];
                code.push_back((uint16_t)(pp - l0));
                code.push_back(didx(b0 + k));
                const size_t cnt_at = code.size();
                code.push_back(0);
                uint16_t nupd = 0;
                for (int32_t q = M->u_ptr[kpos]; q < M->u_ptr[kpos + 1]; ++q) {  // U row k, ascending columns
                  const int32_t j = M->u_col[q];
                  int32_t tgt = -1;
                  if (j == lt) {
                    tgt = didx(t);
                  } else if (j < lt) {  // a later strict-L entry of this row (process_partial_row! on rem_l_pos, ilu0.jl:100-106)
                    for (int32_t p2 = pp + 1; p2 < le; ++p2) if (M->l_col[p2] == j) { tgt = p2 - l0; break; }
                  } else {
                    for (int32_t qi = us; qi < ue; ++qi) if (M->u_col[qi] == j) { tgt = nl + (qi - u0); break; }
                  }
                  if (tgt < 0) continue;  // (k, j) has no counterpart in row i: dropped fill, ILU(0)
                  code.push_back((uint16_t)tgt);
                  code.push_back((uint16_t)(nl + (q - u0)));
                  ++nupd;
                }
                code[cnt_at] = nupd;
              }
              if (code.size() >= 65536) blk_ok[b] = 0;
            }
            prog[nrb] = (uint16_t)code.size();
            prog.insert(prog.end(), code.begin(), code.end());
          }
        });
        bool ok = true;
        int max_vals = 0, max_words = 0;
        size_t total = 0;
        for (int64_t b = 0; b < nb; ++b) {
          ok = ok && blk_ok[b];
          max_vals = std::max(max_vals, blk_vals[b]);
          max_words = std::max<int>(max_words, (int)progs[b].size());
          M->blk_prog[b] = (int32_t)total;
          total += progs[b].size();
        }
        M->blk_prog[nb] = (int32_t)total;
        size_t bytes = sizeof(double) * P.bs * P.bs * (size_t)max_vals + sizeof(uint16_t) * (size_t)max_words;
        bytes = (bytes + 15) & ~(size_t)15;
        if (timing) fprintf(stderr, "[jutul_hip setup] ilu0: long rows (%d entries): row-major factor programs %s, LDS %zu B per block\n", maxcnt,
                            (ok && bytes <= 160 * 1024 - 512) ? "on" : "off", bytes);
        if (ok && total < (size_t)INT32_MAX && bytes <= 160 * 1024 - 512) {
          std::vector<uint16_t> prog(std::max<size_t>(total, 1), 0);
          for (int64_t b = 0; b < nb; ++b)
            if (!progs[b].empty()) std::copy(progs[b].begin(), progs[b].end(), prog.begin() + M->blk_prog[b]);
          M->prog = true;
          M->prog_rowmajor = true;
          M->prog_lds_bytes = bytes; M->prog_max_vals = max_vals; M->prog_max_words = max_words;
          hipStream_t sp = M->ctx->stream;
          M->d_blk_lbase.upload(M->blk_lbase, sp); M->d_blk_ubase.upload(M->blk_ubase, sp); M->d_blk_prog.upload(M->blk_prog, sp);
          M->d_blk_dbase.upload(M->blk_dbase, sp);
          M->d_prog.upload(prog, sp);
        }
      }
    }
    lap("jagged layout + programs");
    // upload
    hipStream_t s = M->ctx->stream;
    // Option ilu_lean_upload (default 0, to be validated on a GPU: JH_OPTIONS=ilu_lean_upload=1 over the GPU suite): with the
    // chunk-jagged layout AND a factorisation that writes it directly (pivot-only / program kernels) the device reads chunk_ptr, the
    // j* tables, rowmap16, flev_off / flev_ptr and blk_ptr only -- the row-major structure arrays (~0.7 GB at 10M cells) stay on the host.
    bool lean = false;
    if (lds) {  // 16-bit copies for the chunked apply
      std::vector<uint16_t> rm(n);
      std::atomic<bool> local{true};
      parallel_ranges(nb, 64, [&](int64_t bb0, int64_t bb1) {
        for (int64_t b = bb0; b < bb1; ++b)
          for (int32_t t = M->blk_ptr[b]; t < M->blk_ptr[b + 1]; ++t) {
            const int32_t off = M->rowmap[t] - M->blk_ptr[b];  // device row of ilu row t, relative to the block
            if (off < 0 || off >= M->blk_ptr[b + 1] - M->blk_ptr[b]) { local.store(false, std::memory_order_relaxed); continue; }
            rm[M->rowmap[t]] = (uint16_t)(t - M->blk_ptr[b]);  // device row -> position inside the block
          }
      });
      M->rowmap_local = local.load();
      if (!M->rowmap_local) M->jag = false;  // the jagged kernels read and write the vector in device order
      lean = M->jag && M->prog && M->ctx->opt.ilu_lean_upload != 0;
      if (!lean) {
        std::vector<uint16_t> lc(M->l_col.size()), uc(M->u_col.size()), ur(M->u_row.size());
        parallel_ranges((int64_t)lc.size(), 1 << 18, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) lc[i] = (uint16_t)M->l_col[i]; });
        parallel_ranges((int64_t)uc.size(), 1 << 18, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) uc[i] = (uint16_t)M->u_col[i]; });
        parallel_ranges((int64_t)ur.size(), 1 << 18, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) ur[i] = (uint16_t)M->u_row[i]; });
        M->d_l_col16.upload(lc, s); M->d_u_col16.upload(uc, s); M->d_u_row16.upload(ur, s);
      }
      M->d_rowmap16.upload(rm, s);
    } else {
      M->jag = false;
    }
    if (!M->send_ptr.empty()) {
      M->d_send_ptr.upload(M->send_ptr, s); M->d_send_local.upload(M->send_local, s); M->d_send_slot.upload(M->send_slot, s);
    }
    if (!lean) {
      M->d_l_lev.upload(M->l_lev, s); M->d_u_lev.upload(M->u_lev, s);
      M->d_rowmap.upload(M->rowmap, s);
    }
    M->d_blk_ptr.upload(M->blk_ptr, s);
    M->d_flev_off.upload(M->flev_off, s); M->d_flev_ptr.upload(M->flev_ptr, s);
    if (!lean) {
      M->d_blev_off.upload(M->blev_off, s); M->d_blev_ptr.upload(M->blev_ptr, s);
      M->d_l_ptr.upload(M->l_ptr, s); M->d_l_col.upload(M->l_col, s); M->d_l_map.upload(M->l_map, s);
      M->d_u_ptr.upload(M->u_ptr, s); M->d_u_col.upload(M->u_col, s); M->d_u_map.upload(M->u_map, s);
      M->d_d_map.upload(M->d_map, s); M->d_u_row.upload(M->u_row, s); M->d_upos_of.upload(M->upos_of, s);
    }
    const int bb = P.bs * P.bs;
    if (!(M->jag && M->prog)) {  // row-major factor arrays: only for the kernels that still use them
      M->l_val.alloc(std::max<size_t>(M->l_col.size(), 1) * bb);
      M->u_val.alloc(std::max<size_t>(M->u_col.size(), 1) * bb);
      M->dinv.alloc((size_t)n * bb);
    }
    if (!lds) M->xg.alloc((size_t)n * P.bs);
    M->lds_bytes = lds ? (size_t)maxrows * P.bs * sizeof(double) : 0;
    jh::stream_sync(s);
    lap("upload");
    *out = M.release();
  });
}

extern "C" int32_t jh_ilu0_destroy(jh_ilu M) {
  return guard([&] { delete M; });
}

extern "C" int32_t jh_ilu0_info(jh_ilu M, int64_t *nblocks, int64_t *max_block_rows, int64_t *max_levels) {
  return guard([&] {
    if (!M) JH_THROW("null handle");
    if (nblocks) *nblocks = M->nparts;
    if (max_block_rows) *max_block_rows = M->max_block_rows;
    if (max_levels) *max_levels = M->max_levels;
  });
}

// stats[0] strict-lower entries kept, [1] strict-upper entries kept, [2] execution blocks, [3] bit 0 = LDS mode, bit 1 = chunk-jagged
// layout, bit 2 = program-driven factorisation available, bit 3 = pivot-only factorisation in use
extern "C" int32_t jh_ilu0_stats(jh_ilu M, int64_t *stats4) {
  return guard([&] {
    if (!M || !stats4) JH_THROW("null argument");
    stats4[0] = (int64_t)M->l_col.size();
    stats4[1] = (int64_t)M->u_col.size();
    stats4[2] = (int64_t)M->blk_ptr.size() - 1;
    stats4[3] = (M->lds_mode ? 1 : 0) | (M->jag ? 2 : 0) | (((M->jag && M->prog) || M->prog_rowmajor) ? 4 : 0) | (M->jag && M->prog && M->diag_only ? 8 : 0) |
                ((M->jag && M->uscaled && M->ctx->opt.fused_product) ? 16 : 0) | (M->jag && M->prog && M->prog_rows ? 32 : 0);
  });
}

extern "C" int32_t jh_diag_precond_create(jh_csr A, int32_t kind, double w, jh_ilu *out) {
  return guard([&] {
    if (!A || !out) JH_THROW("null argument");
    if (kind != 1 && kind != 2) JH_THROW("kind must be 1 (Jacobi) or 2 (SPAI0)");
    for (int64_t i = 0; i < A->pat->n; ++i)
      if (A->pat->diag[i] < 0) JH_THROW("Diagonal must be present in sparsity pattern.");
    auto M = std::make_unique<jh_ilu_s>();
    M->ctx = A->ctx; M->A = A; M->pat = A->pat; M->bs = A->pat->bs; M->n = A->pat->n;
    M->kind = kind;
    M->jacobi_w = w;
    jh::select_device(M->ctx);
    M->dinv.alloc((size_t)M->n * M->bs * M->bs);
    *out = M.release();
  });
}

namespace jh {
// row-major factor arrays -> chunk-jagged ones (factor kernels that still produce the row-major order)
static void ilu_to_jagged(jh_ilu M) {
  if (M->prog) JH_THROW("internal error: ilu_to_jagged on a preconditioner whose kernels write the jagged arrays themselves");
  hipStream_t s = M->ctx->stream;
  const int bb = M->bs * M->bs;
  auto g = [](int64_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096))); };
  if (!M->l_col.empty())
    hipLaunchKernelGGL(ilu_jag_permute_kernel, g((int64_t)M->l_col.size() * bb), dim3(256), 0, s, M->jl_val.p, M->l_val.p, M->d_jl_of_old.p, (int64_t)M->l_col.size(), bb);
  if (!M->u_col.empty())
    hipLaunchKernelGGL(ilu_jag_permute_kernel, g((int64_t)M->u_col.size() * bb), dim3(256), 0, s, M->ju_val.p, M->u_val.p, M->d_ju_of_old.p, (int64_t)M->u_col.size(), bb);
  hipLaunchKernelGGL(ilu_jag_permute_kernel, g(M->n * bb), dim3(256), 0, s, M->jdinv.p, M->dinv.p, M->d_jd_of_old.p, M->n, bb);
}
// launches the chunk-jagged apply (GM as in ilu_apply_chunked_kernel); Q.q != nullptr: with the fused product (mul = 1 + dots)
static void ilu_apply_jagged(jh_ilu M, IluDev F, const double *b, double *x, const IluGather &G, const IluMul &Q = IluMul(), int mul = 0) {
  F.l_val = M->jl_val.p; F.u_val = M->ju_val.p; F.dinv = M->jdinv.p;
  const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
  hipStream_t s = M->ctx->stream;
  if (mul && !M->uscaled) JH_THROW("fused product needs column-scaled factors");
#define JH_J(BSV, GMV, KUV, SCV, MULV) hipLaunchKernelGGL((ilu_apply_jds_kernel<BSV, GMV, KUV, SCV, MULV>), dim3((unsigned)nb), dim3(64), M->lds_bytes, s, F, b, x, G, Q)
#define JH_JV(BSV, GMV) hipLaunchKernelGGL((ilu_apply_jds_kernel<BSV, GMV, 8, false, 0, true>), dim3((unsigned)nb), dim3(64), M->lds_bytes, s, F, b, x, G, Q)
#define JH_JM(BSV, GMV, KUV)                                                                          \
  do {                                                                                                 \
    if (M->jag_vr) JH_JV(BSV, GMV);                                                                    \
    else if (!M->uscaled) JH_J(BSV, GMV, KUV, false, 0);                                               \
    else if (mul == 0) JH_J(BSV, GMV, KUV, true, 0);                                                   \
    else if (mul == 1) JH_J(BSV, GMV, KUV, true, 1);                                                   \
    else if (mul == 2) JH_J(BSV, GMV, KUV, true, 2);                                                   \
    else JH_J(BSV, GMV, KUV, true, 3);                                                                 \
  } while (0)
#define JH_JK(BSV, GMV) do { if (M->jag_ku == 4) JH_JM(BSV, GMV, 4); else JH_JM(BSV, GMV, 8); } while (0)
  switch (M->bs * 10 + G.mode) {
    case 10: JH_JK(1, 0); break; case 11: JH_JK(1, 1); break; case 12: JH_JK(1, 2); break;
    case 20: JH_JK(2, 0); break; case 21: JH_JK(2, 1); break; case 22: JH_JK(2, 2); break;
    case 30: JH_JK(3, 0); break; case 31: JH_JK(3, 1); break; case 32: JH_JK(3, 2); break;
    default: JH_THROW("bad jagged apply mode");
  }
#undef JH_JK
#undef JH_JM
#undef JH_JV
#undef JH_J
}
void ilu_factor(jh_ilu M) {
  jh_context ctx = M->ctx;
  hipStream_t s = ctx->stream;
  if (M->kind != 0) {
    const Pattern &P = *M->pat;
    dim3 g((unsigned)((M->n + 255) / 256));
    switch (M->bs) {
      case 1: hipLaunchKernelGGL(diag_precond_kernel<1>, g, dim3(256), 0, s, P.d_rowptr.p, P.d_diag.p, M->A->val.p, M->dinv.p, M->n, M->kind, M->jacobi_w); break;
      case 2: hipLaunchKernelGGL(diag_precond_kernel<2>, g, dim3(256), 0, s, P.d_rowptr.p, P.d_diag.p, M->A->val.p, M->dinv.p, M->n, M->kind, M->jacobi_w); break;
      case 3: hipLaunchKernelGGL(diag_precond_kernel<3>, g, dim3(256), 0, s, P.d_rowptr.p, P.d_diag.p, M->A->val.p, M->dinv.p, M->n, M->kind, M->jacobi_w); break;
    }
    M->factored = true;
    return;
  }
  const int bb = M->bs * M->bs;
  const double *aval = M->A->val.p;
  if (M->jag && M->prog) {  // straight into the chunk-jagged arrays
    IluDev F = dev_view(M);
    F.l_val = M->jl_val.p; F.u_val = M->ju_val.p; F.dinv = M->jdinv.p;
    const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
    if (M->diag_only) {  // every elimination update lands on a pivot: the sweep-shaped kernel
      const int dthreads = 64 * M->max_chunks;
      const size_t dlds = sizeof(double) * (size_t)M->max_block_rows * M->bs * M->bs;
#define JH_DIAG(BSV, KUV, SCV) hipLaunchKernelGGL((ilu_factor_diag_kernel<BSV, KUV, SCV>), dim3((unsigned)nb), dim3(dthreads), dlds, s, F, aval, \
                                                   M->d_blk_ubase.p, M->d_jl_map.p, M->d_jt_map.p, M->d_ju_map.p, M->d_jf_diag.p, M->d_jf_bslot.p)
#define JH_DIAGK(BSV) do { if (M->uscaled) { if (M->jag_ku == 4) JH_DIAG(BSV, 4, true); else JH_DIAG(BSV, 8, true); } \
                           else { if (M->jag_ku == 4) JH_DIAG(BSV, 4, false); else JH_DIAG(BSV, 8, false); } } while (0)
      // One wavefront per block (ilu_factor_wave_kernel) or one workgroup per block (ilu_factor_diag_kernel).  JH_ILU_FACTOR_WAVE:
      // unset = by block size and row length (see the kernel), 0 = never, 1 = always, 2 / 3 = always, with / without the prefetch of
      // the next chunk's operands
      const int wave_env = (int)ctx->opt.ilu_factor_kernel;  // (per call: the tests switch it)
      const int wave = wave_env >= 0 ? wave_env : ((M->bs > 1 || M->jag_ku > 4) ? 1 : 0);
      if (wave) {
#define JH_FWL(BSV, KUV, SCV, PFV) hipLaunchKernelGGL((ilu_factor_wave_kernel<BSV, KUV, SCV, PFV>), dim3((unsigned)nb), dim3(64), dlds, s, F, aval, \
                                                      M->d_blk_ubase.p, M->d_jl_map.p, M->d_jt_map.p, M->d_ju_map.p, M->d_jf_diag.p, M->d_jf_bslot.p)
#define JH_FWK(BSV, PFV) do { if (M->uscaled) { if (M->jag_ku == 4) JH_FWL(BSV, 4, true, PFV); else JH_FWL(BSV, 8, true, PFV); } \
                              else { if (M->jag_ku == 4) JH_FWL(BSV, 4, false, PFV); else JH_FWL(BSV, 8, false, PFV); } } while (0)
        // prefetch: scalar matrices only by default (blocks: the registers of a second chunk cost more wavefronts per SIMD than
        // the prefetch hides: 0.532 vs 0.532 ms measured, 2 instead of 3 wavefronts per SIMD)
        switch (M->bs) {
          case 1: if (wave != 3) JH_FWK(1, true); else JH_FWK(1, false); break;
          case 2: if (wave == 2) JH_FWK(2, true); else JH_FWK(2, false); break;
          case 3: if (wave == 2) JH_FWK(3, true); else JH_FWK(3, false); break;
        }
#undef JH_FWK
#undef JH_FWL
      } else {
        switch (M->bs) { case 1: JH_DIAGK(1); break; case 2: JH_DIAGK(2); break; case 3: JH_DIAGK(3); break; }
      }
#undef JH_DIAGK
#undef JH_DIAG
      JH_HIP(hipGetLastError());  // a refused launch must not leave stale factors marked as fresh
      M->factored = true;
      return;
    }
    const int pthreads = ctx->opt.ilu_factor_threads ? (int)ctx->opt.ilu_factor_threads : 512;
    // long rows (chains of lanes in the layout): a wavefront per row, its lanes on the update pairs of one multiplier
    const bool wprj = M->jag_vr && ctx->opt.ilu_factor_wave_per_row != 0;
    if (M->prog_rows) {
      // (four wavefronts: seven blocks of ~20 KB of LDS share a CU; eight would leave room for four blocks)
      const int rthreads = ctx->opt.ilu_factor_threads ? std::min((int)ctx->opt.ilu_factor_threads, 512) : 256;
      auto go = [&](auto kern) {
        if (M->prog_lds_bytes > 64 * 1024)
          JH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M->prog_lds_bytes));
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(rthreads), M->prog_lds_bytes, s, F, aval, M->d_blk_lbase.p, M->d_blk_ubase.p,
                           M->d_blk_prog.p, M->d_jl_map.p, M->d_ju_map.p, M->d_jd_map.p, M->d_prog.p, M->prog_max_vals, M->d_blk_dbase.p);
      };
      go(ilu_factor_rows_kernel<1>);  // (scalar matrices only: see want_rows)
    } else
    switch (M->bs) {
      case 1: launch_factor_prog<1>(M, F, aval, M->d_jl_map.p, M->d_ju_map.p, M->d_jd_map.p, wprj, pthreads, nb, s); break;
      case 2: launch_factor_prog<2>(M, F, aval, M->d_jl_map.p, M->d_ju_map.p, M->d_jd_map.p, wprj, pthreads, nb, s); break;
      case 3: launch_factor_prog<3>(M, F, aval, M->d_jl_map.p, M->d_ju_map.p, M->d_jd_map.p, wprj, pthreads, nb, s); break;
    }
    JH_HIP(hipGetLastError());
    M->factored = true;
    return;
  }
  if (M->prog_rowmajor && M->lds_mode) {  // long rows: the program-driven refactorisation over the row-major arrays
    IluDev F = dev_view(M);
    const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
    const int pthreads = ctx->opt.ilu_factor_threads ? (int)ctx->opt.ilu_factor_threads : 512;
    const bool wpr = ctx->opt.ilu_factor_wave_per_row != 0;  // wave per row (long rows); 0: thread per row
    switch (M->bs) {
      case 1: launch_factor_prog<1>(M, F, aval, M->d_l_map.p, M->d_u_map.p, M->d_d_map.p, wpr, pthreads, nb, s); break;
      case 2: launch_factor_prog<2>(M, F, aval, M->d_l_map.p, M->d_u_map.p, M->d_d_map.p, wpr, pthreads, nb, s); break;
      case 3: launch_factor_prog<3>(M, F, aval, M->d_l_map.p, M->d_u_map.p, M->d_d_map.p, wpr, pthreads, nb, s); break;
    }
    JH_HIP(hipGetLastError());
    M->factored = true;
    return;
  }
  if (M->lds_mode && M->factor_lds_bytes) {
    IluDev F = dev_view(M);
    const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
    const int mr = (int)M->max_block_rows;
    // the bulk load/store phases want many lanes (memory-level parallelism); the level loop only needs a few
    const int fthreads = ctx->opt.ilu_factor_threads ? (int)ctx->opt.ilu_factor_threads : 512;
    if (M->factor_lds_bytes > 64 * 1024) {
      const int fb = (int)M->factor_lds_bytes;
      switch (M->bs) {
        case 1: JH_HIP(hipFuncSetAttribute((const void *)ilu_factor_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, fb)); break;
        case 2: JH_HIP(hipFuncSetAttribute((const void *)ilu_factor_lds_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, fb)); break;
        case 3: JH_HIP(hipFuncSetAttribute((const void *)ilu_factor_lds_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, fb)); break;
      }
    }
    switch (M->bs) {
      case 1: hipLaunchKernelGGL(ilu_factor_lds_kernel<1>, dim3((unsigned)nb), dim3(fthreads), M->factor_lds_bytes, s, F, aval, M->max_blk_l, M->max_blk_u, mr, (int)M->max_levels); break;
      case 2: hipLaunchKernelGGL(ilu_factor_lds_kernel<2>, dim3((unsigned)nb), dim3(fthreads), M->factor_lds_bytes, s, F, aval, M->max_blk_l, M->max_blk_u, mr, (int)M->max_levels); break;
      case 3: hipLaunchKernelGGL(ilu_factor_lds_kernel<3>, dim3((unsigned)nb), dim3(fthreads), M->factor_lds_bytes, s, F, aval, M->max_blk_l, M->max_blk_u, mr, (int)M->max_levels); break;
    }
    JH_HIP(hipGetLastError());
    if (M->jag) ilu_to_jagged(M);
    M->factored = true;
    return;
  }
  auto grid_for = [](int64_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096))); };
  if (!M->l_col.empty())
    hipLaunchKernelGGL(ilu_load_kernel, grid_for((int64_t)M->l_col.size() * bb), dim3(256), 0, s, M->l_val.p, aval, M->d_l_map.p, (int64_t)M->l_col.size(), bb);
  if (!M->u_col.empty())
    hipLaunchKernelGGL(ilu_load_kernel, grid_for((int64_t)M->u_col.size() * bb), dim3(256), 0, s, M->u_val.p, aval, M->d_u_map.p, (int64_t)M->u_col.size(), bb);
  hipLaunchKernelGGL(ilu_load_kernel, grid_for(M->n * bb), dim3(256), 0, s, M->dinv.p, aval, M->d_d_map.p, M->n, bb);
  IluDev F = dev_view(M);
  const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
  if (M->lds_mode) {
    switch (M->bs) {
      case 1: hipLaunchKernelGGL(ilu_factor_blocks_kernel<1>, dim3((unsigned)nb), dim3(M->threads), 0, s, F); break;
      case 2: hipLaunchKernelGGL(ilu_factor_blocks_kernel<2>, dim3((unsigned)nb), dim3(M->threads), 0, s, F); break;
      case 3: hipLaunchKernelGGL(ilu_factor_blocks_kernel<3>, dim3((unsigned)nb), dim3(M->threads), 0, s, F); break;
    }
  } else {
    const int nlev = M->flev_off[1] - M->flev_off[0] - 1;
    for (int lev = 1; lev < nlev; ++lev) {
      int st = M->flev_ptr[lev], en = M->flev_ptr[lev + 1];
      dim3 g((unsigned)((en - st + 127) / 128));
      switch (M->bs) {
        case 1: hipLaunchKernelGGL(ilu_factor_level_kernel<1>, g, dim3(128), 0, s, F, st, en); break;
        case 2: hipLaunchKernelGGL(ilu_factor_level_kernel<2>, g, dim3(128), 0, s, F, st, en); break;
        case 3: hipLaunchKernelGGL(ilu_factor_level_kernel<3>, g, dim3(128), 0, s, F, st, en); break;
      }
    }
  }
  dim3 gi((unsigned)((M->n + 255) / 256));
  switch (M->bs) {
    case 1: hipLaunchKernelGGL(ilu_invert_kernel<1>, gi, dim3(256), 0, s, M->dinv.p, M->n); break;
    case 2: hipLaunchKernelGGL(ilu_invert_kernel<2>, gi, dim3(256), 0, s, M->dinv.p, M->n); break;
    case 3: hipLaunchKernelGGL(ilu_invert_kernel<3>, gi, dim3(256), 0, s, M->dinv.p, M->n); break;
  }
  if (M->jag) ilu_to_jagged(M);
  M->factored = true;
}

void ilu_apply(jh_ilu M, const double *b, double *x) {
  if (!M->factored) JH_THROW("ILU(0) applied before jh_ilu0_factor");
  hipStream_t s = M->ctx->stream;
  if (M->kind != 0) {
    dim3 g((unsigned)((M->n + 255) / 256));
    switch (M->bs) {
      case 1: hipLaunchKernelGGL(diag_apply_kernel<1>, g, dim3(256), 0, s, M->dinv.p, b, x, M->n); break;
      case 2: hipLaunchKernelGGL(diag_apply_kernel<2>, g, dim3(256), 0, s, M->dinv.p, b, x, M->n); break;
      case 3: hipLaunchKernelGGL(diag_apply_kernel<3>, g, dim3(256), 0, s, M->dinv.p, b, x, M->n); break;
    }
    return;
  }
  IluDev F = dev_view(M);
  const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
  if (M->jag) { ilu_apply_jagged(M, F, b, x, IluGather()); return; }
  if (M->lds_mode) {
    switch (M->bs) {
      case 1: hipLaunchKernelGGL(ilu_apply_blocks_pf_kernel<1>, dim3((unsigned)nb), dim3(M->threads), M->lds_bytes, s, F, b, x); break;
      case 2: hipLaunchKernelGGL(ilu_apply_blocks_pf_kernel<2>, dim3((unsigned)nb), dim3(M->threads), M->lds_bytes, s, F, b, x); break;
      case 3: hipLaunchKernelGGL(ilu_apply_blocks_pf_kernel<3>, dim3((unsigned)nb), dim3(M->threads), M->lds_bytes, s, F, b, x); break;
    }
    return;
  }
  dim3 gn((unsigned)((M->n + 255) / 256));
  double *xg = M->xg.p;
#define JH_BS_SWITCH(KERNEL, GRID, BLOCK, ...)                                                    \
  switch (M->bs) {                                                                                \
    case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, BLOCK, 0, s, __VA_ARGS__); break;                 \
    case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, BLOCK, 0, s, __VA_ARGS__); break;                 \
    case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, BLOCK, 0, s, __VA_ARGS__); break;                 \
  }
  JH_BS_SWITCH(ilu_gather_kernel, gn, dim3(256), xg, b, M->d_rowmap.p, M->n, false);
  const int nfl = M->flev_off[1] - M->flev_off[0] - 1;
  for (int lev = 1; lev < nfl; ++lev) {
    int st = M->flev_ptr[lev], en = M->flev_ptr[lev + 1];
    dim3 g((unsigned)((en - st + 127) / 128));
    JH_BS_SWITCH(ilu_fwd_level_kernel, g, dim3(128), F, xg, st, en);
  }
  const int nbl = M->blev_off[1] - M->blev_off[0] - 1;
  for (int lev = 0; lev < nbl; ++lev) {
    int st = M->blev_ptr[lev], en = M->blev_ptr[lev + 1];
    dim3 g((unsigned)((en - st + 127) / 128));
    JH_BS_SWITCH(ilu_bwd_level_kernel, g, dim3(128), F, xg, st, en);
  }
  JH_BS_SWITCH(ilu_gather_kernel, gn, dim3(256), xg, (const double *)x, M->d_rowmap.p, M->n, true);
#undef JH_BS_SWITCH
}
bool ilu_can_fuse_gather(jh_ilu M) {
  return M && M->kind == 0 && M->lds_mode && M->threads == 64 && M->rowmap_local && M->ctx->opt.fuse_gather;
}
// x = M^-1 b with the ghost rows of b taken as zero (rows >= n_owned_rows; parray_preconditioner_apply!, linalg.jl:78-88) and the
// rows neighbouring ranks hold as ghosts stored into the halo send buffer / landing buffers by the same launch -- the first
// apply of a solve, which has no vector update to fuse.  Only on the chunk-jagged layout (ilu_can_pack_halo && M->jag).
bool ilu_can_pack_halo(jh_ilu M);
bool ilu_can_apply_pack(jh_ilu M) { return ilu_can_pack_halo(M) && M->jag && M->kind == 0; }
void ilu_apply_pack(jh_ilu M, const double *b, double *x, int n_owned_rows) {
  if (!M->factored) JH_THROW("ILU(0) applied before jh_ilu0_factor");
  if (!ilu_can_apply_pack(M)) JH_THROW("packed apply needs the chunk-jagged layout and a matching halo plan");
  IluDev F = dev_view(M);
  F.send_ptr = M->d_send_ptr.p; F.send_local = M->d_send_local.p; F.send_slot = M->d_send_slot.p;
  F.send_buf = M->A->disc->halo.d_send_buf.p;
  F.send_dst = halo_push_targets(M->A->disc);
  IluGather G;
  G.n_owned_rows = n_owned_rows;
  ilu_apply_jagged(M, F, b, x, G);
}
// x = M^-1 * (fused vector update), see IluGather / ilu_apply_chunked_kernel
// true if ilu_apply_fused(..., pack = true) can fill the halo send buffer of the matrix' discretisation
bool ilu_can_pack_halo(jh_ilu M) {
  return M && M->ctx->opt.fused_pack && !M->send_ptr.empty() && M->A->disc && M->A->disc->halo.active && M->A->disc->halo.epoch == M->halo_epoch;
}
void ilu_apply_fused(jh_ilu M, const IluGather &G, double *x, bool pack) {
  if (!M->factored) JH_THROW("ILU(0) applied before jh_ilu0_factor");
  hipStream_t s = M->ctx->stream;
  IluDev F = dev_view(M);
  if (pack) {
    if (!ilu_can_pack_halo(M)) JH_THROW("fused halo pack requested without a matching halo plan");
    F.send_ptr = M->d_send_ptr.p; F.send_local = M->d_send_local.p; F.send_slot = M->d_send_slot.p;
    F.send_buf = M->A->disc->halo.d_send_buf.p;
    F.send_dst = halo_push_targets(M->A->disc);  // non-null when the push halo is enabled: the exchange that follows is a push
  }
  if (M->jag) { ilu_apply_jagged(M, F, nullptr, x, G); return; }
  const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
#define JH_FUSED(BSV, GMV) hipLaunchKernelGGL((ilu_apply_chunked_kernel<BSV, GMV>), dim3((unsigned)nb), dim3(64), M->lds_bytes, s, F, (const double *)nullptr, x, G)
  switch (M->bs * 10 + G.mode) {
    case 11: JH_FUSED(1, 1); break;
    case 12: JH_FUSED(1, 2); break;
    case 21: JH_FUSED(2, 1); break;
    case 22: JH_FUSED(2, 2); break;
    case 31: JH_FUSED(3, 1); break;
    case 32: JH_FUSED(3, 2); break;
    default: JH_THROW("bad fused gather mode");
  }
#undef JH_FUSED
}
// The product that follows every preconditioner apply of the right-preconditioned Krylov loop, formed inside the apply
// (jh_ilu_s::uscaled): x = M^-1 (b, or the fused vector update G), q = A x, optional fused dot (SpmvDot contract: mode 1
// sum(q .* w), mode 2 sum(q .* w), sum(q .* q)).  Two launches: the apply with its third pass (in-block entries, one dot partial
// per block) and ilu_eprod_kernel (out-of-block entries, partials behind them).  Returns the number of partials; the caller
// runs the second reduction stage.  done: the launches are no-ops once *done != 0.
bool ilu_can_fuse_product(jh_ilu M) {
  // (the option is read per solve: the parity tests compare both paths in one process)
  return M && M->ctx->opt.fused_product && M->kind == 0 && M->jag && M->uscaled && M->factored;
}
// the out-of-block entries' values in the order ilu_eprod_kernel reads them: once per solve (the matrix is constant inside one)
void ilu_eprod_refresh(jh_ilu M) {
  if (!M->uscaled || M->e_nent == 0) return;
  const int bb = M->bs * M->bs;
  if (M->e_val.n < (size_t)M->e_nent * bb) M->e_val.alloc((size_t)M->e_nent * bb);
  k_gather_blocks(M->ctx->stream, M->e_val.p, M->A->val.p, M->d_e_slot.p, M->e_nent, bb, false);
}
static int eprod_grid(jh_ilu M) { return (int)std::max<int64_t>(1, std::min<int64_t>((M->e_rows + 255) / 256, 2048)); }
void ilu_apply_mul(jh_ilu M, const IluGather &G, const double *b, double *x, double *q, const SpmvDot *dot, const double *done, bool pack) {
  if (!ilu_can_fuse_product(M)) JH_THROW("fused preconditioner apply + product is not available for this factorisation");
  jh_context ctx = M->ctx;
  const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
  const int mode = dot ? dot->mode : 0;
  if (mode) ensure_partials(ctx, (size_t)nb + eprod_grid(M));
  IluDev F = dev_view(M);
  if (pack) {  // as in ilu_apply_fused: the rows neighbouring ranks hold as ghosts leave with the apply
    if (!ilu_can_pack_halo(M)) JH_THROW("fused halo pack requested without a matching halo plan");
    F.send_ptr = M->d_send_ptr.p; F.send_local = M->d_send_local.p; F.send_slot = M->d_send_slot.p;
    F.send_buf = M->A->disc->halo.d_send_buf.p;
    F.send_dst = halo_push_targets(M->A->disc);
  }
  IluMul Q;
  Q.q = q; Q.dw = dot ? dot->w : nullptr; Q.part = ctx->partials.p; Q.pstride = ctx->partial_stride;
  if (dot && dot->n_rows > 0) Q.dot_rows = (int)dot->n_rows;
  IluGather G2 = G;
  if (G2.mode == 0) G2.done = done;
  ilu_apply_jagged(M, F, b, x, G2, Q, 1 + mode);
}
// second launch of the fused product: the out-of-block entries; returns the number of dot partials of both launches
int ilu_eprod(jh_ilu M, const double *x, double *q, const SpmvDot *dot, const double *done) {
  jh_context ctx = M->ctx;
  hipStream_t s = ctx->stream;
  const int64_t nb = (int64_t)M->blk_ptr.size() - 1;
  const int mode = dot ? dot->mode : 0;
  const int ge = eprod_grid(M);
  const double *dw = dot ? dot->w : nullptr;
  const int drows = dot ? (int)dot->n_rows : 0;
#define JH_E(BSV, DV) hipLaunchKernelGGL((ilu_eprod_kernel<BSV, DV>), dim3(ge), dim3(256), 0, s, M->d_e_row.p, M->d_e_ptr.p, M->d_e_col.p, \
                                         M->e_val.p, x, q, (int)M->e_rows, dw, drows, ctx->partials.p, ctx->partial_stride, (int)nb, done)
#define JH_ED(BSV) do { if (mode == 0) JH_E(BSV, 0); else if (mode == 1) JH_E(BSV, 1); else JH_E(BSV, 2); } while (0)
  switch (M->bs) { case 1: JH_ED(1); break; case 2: JH_ED(2); break; case 3: JH_ED(3); break; }
#undef JH_ED
#undef JH_E
  return (int)nb + ge;
}
}  // namespace jh

extern "C" int32_t jh_ilu0_factor(jh_ilu M) {
  return guard([&] {
    if (!M) JH_THROW("null handle");
    jh::select_device(M->ctx);
    jh::ilu_factor(M);
    JH_HIP(hipGetLastError());
  });
}

extern "C" int32_t jh_ilu0_apply(jh_ilu M, jh_vec b, jh_vec x) {
  return guard([&] {
    if (!M || !b || !x) JH_THROW("null handle");
    if (b->len != M->n * M->bs || x->len != b->len) JH_THROW("dimension mismatch in ilu apply");
    jh::select_device(M->ctx);
    if (b == x && !M->lds_mode) { /* in-place is fine: gather copies first */ }
    jh::ilu_apply(M, b->d.p, x->d.p);
    JH_HIP(hipGetLastError());
  });
}

extern "C" int32_t jh_ilu0_apply_mul(jh_ilu M, jh_vec b, jh_vec x, jh_vec q) {
  return guard([&] {
    if (!M || !b || !x || !q) JH_THROW("null handle");
    if (b->len != M->n * M->bs || x->len != b->len || q->len != b->len) JH_THROW("dimension mismatch in ilu apply + product");
    if (b == x || x == q || b == q) JH_THROW("b, x and q must not alias");
    if (!M->factored) JH_THROW("ILU(0) applied before jh_ilu0_factor");
    jh::select_device(M->ctx);
    if (jh::ilu_can_fuse_product(M)) {
      jh::ilu_eprod_refresh(M);
      jh::ilu_apply_mul(M, jh::IluGather(), b->d.p, x->d.p, q->d.p, nullptr, nullptr, false);
      jh::ilu_eprod(M, x->d.p, q->d.p, nullptr, nullptr);
    } else {  // patterns with triangles inside a block, several ranks, diagonal preconditioners: two operators
      jh::ilu_apply(M, b->d.p, x->d.p);
      jh::k_spmv(M->ctx, *M->pat, M->A->val.p, x->d.p, q->d.p, 1.0, 0.0);
    }
    JH_HIP(hipGetLastError());
  });
}

extern "C" int32_t jh_ilu0_get_factor(jh_ilu M, double *lu) {
  return guard([&] {
    if (!M || !lu) JH_THROW("null argument");
    if (!M->factored) JH_THROW("not factored");
    if (M->kind != 0) JH_THROW("jh_ilu0_get_factor is for ILU(0) handles");
    const Pattern &P = *M->pat;
    const int bb = M->bs * M->bs;
    auto hslot = [&](int32_t k) -> int64_t { return P.nz_hslot.empty() ? k : P.nz_hslot[k]; };
    if (M->jag) {  // chunk-jagged storage: every entry knows the slot of A it came from
      std::vector<double> l(M->jl_val.n), u(M->ju_val.n), d(M->jdinv.n);
      JH_HIP(hipStreamSynchronize(M->ctx->stream));
      jh::copy_d2h(l.data(), M->jl_val.p, l.size() * sizeof(double), M->ctx->stream);
      jh::copy_d2h(u.data(), M->ju_val.p, u.size() * sizeof(double), M->ctx->stream);
      jh::copy_d2h(d.data(), M->jdinv.p, d.size() * sizeof(double), M->ctx->stream);
      if (M->uscaled) {  // D-ILU storage: l holds A's entries; the factor's L_ik = A_ik inv(D~_k) (ilu0.jl:108-144)
        const int64_t nbk = (int64_t)M->blk_ptr.size() - 1;
        for (int64_t bk = 0; bk < nbk; ++bk) {
          const int32_t r0 = M->blk_ptr[bk];
          for (int32_t t = r0; t < M->blk_ptr[bk + 1]; ++t)
            for (int32_t pp = M->l_ptr[t]; pp < M->l_ptr[t + 1]; ++pp) {
              const int32_t j = M->jl_of_old[pp];
              const double *dk = d.data() + (size_t)M->jd_of_old[M->upos_of[r0 + M->l_col[pp]]] * bb;
              double tmp[9];
              for (int cc = 0; cc < M->bs; ++cc)
                for (int rr = 0; rr < M->bs; ++rr) {
                  double sum = 0.0;
                  for (int kk = 0; kk < M->bs; ++kk) sum += l[(size_t)j * bb + kk * M->bs + rr] * dk[cc * M->bs + kk];
                  tmp[cc * M->bs + rr] = sum;
                }
              for (int e = 0; e < bb; ++e) l[(size_t)j * bb + e] = tmp[e];
            }
        }
      }
      for (size_t j = 0; j < M->jl_map.size(); ++j)
        for (int e = 0; e < bb; ++e) lu[hslot(M->jl_map[j]) * bb + e] = l[j * bb + e];
      for (size_t j = 0; j < M->ju_map.size(); ++j)
        for (int e = 0; e < bb; ++e) lu[hslot(M->ju_map[j]) * bb + e] = u[j * bb + e];
      for (size_t j = 0; j < M->jd_map.size(); ++j)
        if (M->jd_map[j] >= 0)
          for (int e = 0; e < bb; ++e) lu[hslot(M->jd_map[j]) * bb + e] = d[j * bb + e];
      return;
    }
    std::vector<double> l(M->l_val.n), u(M->u_val.n), d(M->dinv.n);
    JH_HIP(hipStreamSynchronize(M->ctx->stream));
    jh::copy_d2h(l.data(), M->l_val.p, l.size() * sizeof(double), M->ctx->stream);
    jh::copy_d2h(u.data(), M->u_val.p, u.size() * sizeof(double), M->ctx->stream);
    jh::copy_d2h(d.data(), M->dinv.p, d.size() * sizeof(double), M->ctx->stream);
    for (size_t j = 0; j < M->l_map.size(); ++j)
      for (int e = 0; e < bb; ++e) lu[hslot(M->l_map[j]) * bb + e] = l[j * bb + e];
    for (size_t j = 0; j < M->u_map.size(); ++j)
      for (int e = 0; e < bb; ++e) lu[hslot(M->u_map[j]) * bb + e] = u[j * bb + e];
    for (int64_t pos = 0; pos < M->n; ++pos)
      for (int e = 0; e < bb; ++e) lu[hslot(M->d_map[pos]) * bb + e] = d[pos * bb + e];
  });
}
