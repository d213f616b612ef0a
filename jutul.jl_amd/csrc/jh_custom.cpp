// Runtime-defined conservation laws (SURVEY a-7 / 8f-3): the reference lets a user write the physics of an equation as
// ordinary Julia code that is evaluated on ForwardDiff duals (update_equation_in_entity!, equations.jl:578-594,
// ad/generic.jl:53-96).  Here the user hands over HIP source for two device functions written against the same forward-mode
// dual type the built-in laws use; it is compiled for the device with hiprtc into the tile assembly kernel below, so a new
// law needs no rebuild of the library and no hand-derived Jacobian.
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstring>

#include "jh_internal.hpp"

namespace jh {
void k_set_diag_data(hipStream_t s, double *nzdata, const int32_t *diag, const int32_t *perm, const double *cell_data, int64_t n,
                     bool use_const, double cval);

namespace {
struct Rtc {
  void *h = nullptr;
  hiprtcResult (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char **) = nullptr;
  hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*GetCode)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*DestroyProgram)(hiprtcProgram *) = nullptr;
};
Rtc &rtc() {
  static Rtc R;
  if (R.h) return R;
  for (const char *nm : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
    R.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (R.h) break;
  }
  if (!R.h) JH_THROW(std::string("cannot dlopen libhiprtc: ") + dlerror());
#define JH_SYM(field, name)                 \
  *(void **)(&R.field) = dlsym(R.h, name);  \
  if (!R.field) JH_THROW(std::string("libhiprtc lacks ") + name);
  JH_SYM(CreateProgram, "hiprtcCreateProgram");
  JH_SYM(CompileProgram, "hiprtcCompileProgram");
  JH_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
  JH_SYM(GetProgramLog, "hiprtcGetProgramLog");
  JH_SYM(GetCodeSize, "hiprtcGetCodeSize");
  JH_SYM(GetCode, "hiprtcGetCode");
  JH_SYM(DestroyProgram, "hiprtcDestroyProgram");
#undef JH_SYM
  return R;
}

// The device program: dual numbers + the operators of flux.jl:335-405 (PRELUDE), the user's two functions, the tile kernel
// (same tile structure, LDS staging and coalesced nzval store as assemble_tile_kernel in jh_assembly.hip, generic in JH_N).
const char *PRELUDE = R"JHSRC(
#define JH_NP (2 * JH_N)
struct D { double v; double d[JH_NP]; };   // value + partials w.r.t. (self primaries 0..N-1, other primaries N..2N-1)
__device__ inline D dconst(double v) { D r; r.v = v; for (int i = 0; i < JH_NP; ++i) r.d[i] = 0.0; return r; }
__device__ inline D dvar(double v, int i) { D r = dconst(v); r.d[i] = 1.0; return r; }
__device__ inline D operator+(D a, D b) { D r; r.v = a.v + b.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ inline D operator-(D a, D b) { D r; r.v = a.v - b.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ inline D operator-(D a) { D r; r.v = -a.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = -a.d[i]; return r; }
__device__ inline D operator*(D a, D b) { D r; r.v = a.v * b.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ inline D operator/(D a, D b) { D r; r.v = a.v / b.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
__device__ inline D operator+(D a, double s) { a.v += s; return a; }
__device__ inline D operator+(double s, D a) { a.v += s; return a; }
__device__ inline D operator-(D a, double s) { a.v -= s; return a; }
__device__ inline D operator-(double s, D a) { return dconst(s) - a; }
__device__ inline D operator*(double s, D a) { D r; r.v = s * a.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = s * a.d[i]; return r; }
__device__ inline D operator*(D a, double s) { return s * a; }
__device__ inline D operator/(D a, double s) { D r; r.v = a.v / s; for (int i = 0; i < JH_NP; ++i) r.d[i] = a.d[i] / s; return r; }
__device__ inline D dexp(D a) { double e = exp(a.v); D r; r.v = e; for (int i = 0; i < JH_NP; ++i) r.d[i] = e * a.d[i]; return r; }
__device__ inline D dlog(D a) { D r; r.v = log(a.v); for (int i = 0; i < JH_NP; ++i) r.d[i] = a.d[i] / a.v; return r; }
__device__ inline D dsqrt(D a) { double s = sqrt(a.v); D r; r.v = s; for (int i = 0; i < JH_NP; ++i) r.d[i] = 0.5 * a.d[i] / s; return r; }
__device__ inline D dpow(D a, double p) { double w = pow(a.v, p - 1.0); D r; r.v = w * a.v; for (int i = 0; i < JH_NP; ++i) r.d[i] = p * w * a.d[i]; return r; }
// flux.jl:335-405
__device__ inline D face_average(D a, D b) { return 0.5 * (a + b); }
__device__ inline D two_point_potential_drop(D p_self, D p_other, double gdz, D rho_self, D rho_other) {
  return (p_self - p_other) + gdz * face_average(rho_self, rho_other);
}
__device__ inline D upwind(D potential_drop, D v_self, D v_other) { return potential_drop.v < 0.0 ? v_other : v_self; }
)JHSRC";

const char *KERNEL = R"JHSRC(
#define JH_THREADS 256
#define JH_ROWS 256
#define JH_TNNZ (JH_N == 1 ? 1024 : 512)
#define JH_NN (JH_N * JH_N)
extern "C" __global__ __launch_bounds__(JH_THREADS) void jh_custom_assemble(
    const int *__restrict__ tile_row, int ntiles, const int *__restrict__ rowptr, const int *__restrict__ col,
    const int *__restrict__ diag, const double *__restrict__ Tnz, const double *__restrict__ gnz, const double *__restrict__ X,
    const double *__restrict__ X0, double *__restrict__ nz, double *__restrict__ r, double dt, const double *__restrict__ par) {
  __shared__ double qv[JH_TNNZ * JH_N];
  __shared__ double dsv[JH_TNNZ * JH_NN];
  __shared__ double xs[JH_ROWS * JH_N];
  __shared__ int rp[JH_ROWS + 1];
  __shared__ unsigned char rowof[JH_TNNZ];
  const int chunk = (ntiles + 7) / 8;
  const int t = (blockIdx.x % 8) * chunk + blockIdx.x / 8;  // XCD b % 8 sweeps one contiguous eighth of the tiles
  if (t >= ntiles) return;
  const int r0 = tile_row[t], r1 = tile_row[t + 1], nrows = r1 - r0;
  const int base = rowptr[r0], cnt = rowptr[r1] - base, tid = threadIdx.x;
  for (int i = tid; i <= nrows; i += JH_THREADS) rp[i] = rowptr[r0 + i] - base;
  for (int i = tid; i < nrows * JH_N; i += JH_THREADS) xs[i] = X[(size_t)r0 * JH_N + i];
  __syncthreads();
  if (tid < nrows)
    for (int j = rp[tid]; j < rp[tid + 1]; ++j) rowof[j] = (unsigned char)tid;
  __syncthreads();
  constexpr int KPT = JH_TNNZ / JH_THREADS;
  double off[KPT][JH_NN];
  bool isdiag[KPT];
#pragma unroll
  for (int kk = 0; kk < KPT; ++kk) {
    const int k = tid + kk * JH_THREADS;
    isdiag[kk] = false;
    if (k >= cnt) continue;
    const int lr = rowof[k], c = col[base + k];
    if (c == r0 + lr) {
      for (int e = 0; e < JH_N; ++e) qv[k * JH_N + e] = 0.0;
      for (int i = 0; i < JH_NN; ++i) dsv[k * JH_NN + i] = 0.0;
      isdiag[kk] = true;
      continue;
    }
    D xself[JH_N], xother[JH_N], q[JH_N];
    const bool inl = (c >= r0 && c < r1);
    for (int e = 0; e < JH_N; ++e) {
      xself[e] = dvar(xs[lr * JH_N + e], e);
      xother[e] = dvar(inl ? xs[(c - r0) * JH_N + e] : X[(size_t)c * JH_N + e], JH_N + e);
    }
    jh_flux(xself, xother, Tnz[base + k], gnz ? gnz[base + k] : 0.0, par, q);
    for (int e = 0; e < JH_N; ++e) {
      qv[k * JH_N + e] = q[e].v;
      for (int d = 0; d < JH_N; ++d) {
        dsv[k * JH_NN + d * JH_N + e] = q[e].d[d];       // column-major (e, d): d q_e / d x_self_d
        off[kk][d * JH_N + e] = q[e].d[JH_N + d];         // d q_e / d x_other_d
      }
    }
  }
  __syncthreads();
  if (tid < nrows) {
    const int row = r0 + tid, dk = diag[row] - base;
    const double vol = Tnz[base + dk];  // the diagonal slot carries the accumulation coefficient of the row
    D x[JH_N], x0[JH_N], M[JH_N], M0[JH_N];
    for (int e = 0; e < JH_N; ++e) { x[e] = dvar(xs[tid * JH_N + e], e); x0[e] = dconst(X0[(size_t)row * JH_N + e]); }
    jh_mass(x, par, M);
    jh_mass(x0, par, M0);
    double ar[JH_N], ap[JH_NN];
    for (int e = 0; e < JH_N; ++e) {  // update_accumulation! (conservation.jl:558-568): (M - M0)/dt
      ar[e] = (vol * M[e].v - vol * M0[e].v) / dt;
      for (int d = 0; d < JH_N; ++d) ap[d * JH_N + e] = (vol * M[e].d[d]) / dt;
    }
    for (int j = rp[tid]; j < rp[tid + 1]; ++j) {
      for (int e = 0; e < JH_N; ++e) ar[e] = ar[e] + qv[j * JH_N + e];
      for (int i = 0; i < JH_NN; ++i) ap[i] = ap[i] + dsv[j * JH_NN + i];
    }
    for (int e = 0; e < JH_N; ++e) r[(size_t)row * JH_N + e] = ar[e];
    for (int i = 0; i < JH_NN; ++i) dsv[dk * JH_NN + i] = ap[i];
  }
  __syncthreads();
  // off-diagonal blocks into their (free by now) dsv slots next to the parked diagonal blocks; the tile's nzval range then
  // leaves the CU as one linear copy (a lane storing its own block writes 8 bytes every JH_NN*8: partial lines per instruction)
#pragma unroll
  for (int kk = 0; kk < KPT; ++kk) {
    const int k = tid + kk * JH_THREADS;
    if (k < cnt && !isdiag[kk])
      for (int i = 0; i < JH_NN; ++i) dsv[k * JH_NN + i] = off[kk][i];
  }
  __syncthreads();
  double *dst = nz + (size_t)base * JH_NN;
  for (int i = tid; i < cnt * JH_NN; i += JH_THREADS) dst[i] = dsv[i];
}
)JHSRC";
}  // namespace

void custom_compile(jh_law L, const char *user_source) {
  Rtc &R = rtc();
  std::string src = "#define JH_N " + std::to_string(L->N) + "\n" + PRELUDE + "\n" + user_source + "\n" + KERNEL;
  hiprtcProgram prog;
  if (R.CreateProgram(&prog, src.c_str(), "jh_custom_law.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) JH_THROW("hiprtcCreateProgram failed");
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  hiprtcResult rc = R.CompileProgram(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    R.GetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) R.GetProgramLog(prog, &log[0]);
    R.DestroyProgram(&prog);
    JH_THROW("custom law does not compile:\n" + log);
  }
  size_t n = 0;
  R.GetCodeSize(prog, &n);
  std::vector<char> code(n);
  R.GetCode(prog, code.data());
  R.DestroyProgram(&prog);
  JH_HIP(hipModuleLoadData(&L->custom_module, code.data()));
  JH_HIP(hipModuleGetFunction(&L->custom_kernel, L->custom_module, "jh_custom_assemble"));
}

void custom_launch(jh_law L, double dt, jh_csr A, jh_vec r) {
  const Pattern &P = *A->pat;
  if (dt <= 0.0) JH_THROW("custom laws are time dependent: dt > 0");
  const int *tile_row = P.d_tile_row.p, *rowptr = P.d_rowptr.p, *col = P.d_col.p, *diag = P.d_diag.p;
  int ntiles = P.ntiles;
  const double *Tnz = L->Tnz.p, *gnz = L->has_gdz ? L->gnz.p : nullptr, *X = L->X.p, *X0 = L->X0.p, *par = L->custom_par.p;
  double *nz = A->val.p, *rr = r->d.p;
  void *args[] = {&tile_row, &ntiles, &rowptr, &col, &diag, &Tnz, &gnz, &X, &X0, &nz, &rr, &dt, &par};
  const int chunk = (P.ntiles + NUM_XCD - 1) / NUM_XCD;
  JH_HIP(hipModuleLaunchKernel(L->custom_kernel, chunk * NUM_XCD, 1, 1, TILE_THREADS, 1, 1, 0, L->ctx->stream, args, nullptr));
}

}  // namespace jh

using namespace jh;

extern "C" int32_t jh_law_create_custom(jh_tpfa d, const char *source, const double *params, int32_t n_params, jh_law *out) {
  return guard([&] {
    if (!d || !source || !out) JH_THROW("null argument");
    if (d->N < 1 || d->N > 3) JH_THROW("custom laws support 1..3 equations per cell");
    if (n_params < 0 || (n_params > 0 && !params)) JH_THROW("bad parameter array");
    jh::select_device(d->ctx);
    auto L = std::make_unique<jh_law_s>();
    L->ctx = d->ctx;
    L->disc = d;
    L->kind = JH_LAW_CUSTOM;
    L->N = d->N;
    L->X.alloc((size_t)d->nc * L->N);
    L->X0.alloc((size_t)d->nc * L->N);
    L->Tnz.alloc(d->nnzb);
    hipStream_t s = d->ctx->stream;
    JH_HIP(hipMemsetAsync(L->X.p, 0, L->X.n * sizeof(double), s));
    JH_HIP(hipMemsetAsync(L->X0.p, 0, L->X0.n * sizeof(double), s));
    JH_HIP(hipMemsetAsync(L->Tnz.p, 0, L->Tnz.n * sizeof(double), s));
    k_set_diag_data(s, L->Tnz.p, d->pat->d_diag.p, nullptr, nullptr, d->nc, true, 1.0);
    std::vector<double> p(params, params + n_params);
    if (p.empty()) p.push_back(0.0);
    L->custom_par.upload(p, s);
    JH_HIP(hipStreamSynchronize(s));
    custom_compile(L.get(), source);
    *out = L.release();
  });
}
