/*
 * jutul_hip.h -- C ABI of libjutul_hip.so: the MI355X (gfx950) implementation of Jutul.jl's per-Newton
 * hot path (TPFA residual/Jacobian assembly into (block-)CSR, CSR SpMV, ILU(0) factor/apply, BiCGStab,
 * ghost-cell halo exchange) behind Jutul's own operator surface.
 *
 * Jutul.jl has no FFI: its extension points are Julia dispatch seams.  Every entry point below names the
 * reference seam it replaces (file:line relative to the Jutul.jl source tree); the Julia-side binding
 * (`ccall`) a maintainer would add is shown in INTEGRATION.md and julia/JutulHIP.jl.
 *
 * Conventions
 *  - every function returns int32: 0 = OK, < 0 = error; jh_last_error() gives the message (thread-local).
 *    The Julia glue turns non-zero into `error(...)`, preserving the reference's exception semantics
 *    (e.g. failure_cuts_timestep, simulator.jl:509-518).
 *  - host index arrays cross the boundary as the reference stores them: Int64, 1-BASED (context.jl:76-78).
 *    They are converted once to 0-based int32 on the device.
 *  - host floating-point arrays are Float64 in the reference's layouts ([N, nc] column-major for block /
 *    entity-major vectors; flat jac_buffer for Jacobian values, linsolve/default.jl:166-226).
 *  - handles are opaque; a handle is not re-entrant; all work on one context is ordered on one HIP
 *    stream; calls that return host-visible data synchronise that stream.
 *  - device-resident vectors are `jh_vec` handles so the Krylov loop never leaves HBM.
 */
#ifndef JUTUL_HIP_H
#define JUTUL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jh_context_s *jh_context; /* JutulContext (core_types.jl:86-88; contexts/csr.jl:3-23)        */
typedef struct jh_tpfa_s *jh_tpfa;       /* TwoPointPotentialFlowHardCoded + pattern + positions (a-1..a-4) */
typedef struct jh_csr_s *jh_csr;         /* StaticSparsityMatrixCSR on the device (StaticCSR/mat.jl:1-8)    */
typedef struct jh_vec_s *jh_vec;         /* r_buffer / dx_buffer-like device vector (default.jl:188-226)    */
typedef struct jh_law_s *jh_law;         /* ConservationLaw + ConservationLawTPFAStorage (conservation.jl:101-135) */
typedef struct jh_ilu_s *jh_ilu;         /* ILUFactorCSR / ParallelILUFactorCSR (ilu0.jl:190-203, par_ilu0.jl:2-5) */
typedef struct jh_subdomain_s *jh_subdomain; /* one rank's share of a partitioned grid (dd/subdomains.jl:77-182)            */
typedef struct jh_krylov_s *jh_krylov;   /* GenericKrylov(:bicgstab) workspace (linsolve/krylov.jl:27-58)   */

/* ---- errors / context ------------------------------------------------------------------------------- */
int32_t jh_last_error(char *buf, int64_t cap);
int32_t jh_version(void);
/* device_id: HIP device ordinal.  Mirrors SingleCUDAContext's role (contexts/cuda.jl). */
int32_t jh_context_create(int32_t device_id, jh_context *out);
/* A PLANNING context: no device behind it.  The set-up entry points run their host phases on it -- jh_tpfa_create(_weighted)
 * and every jh_tpfa_get_* table getter (a-1 .. a-4: the tables setup_equation_storage builds on the host,
 * conservation/conservation.jl:101-216), jh_csr_create / jh_csr_create_from_pattern (pattern only), jh_spmv_info (jagged layout),
 * jh_ilu0_create + jh_ilu0_info / jh_ilu0_stats (the symbolic phase of ilu0_csr, StaticCSR/ilu0.jl:13-81), jh_halo_create + jh_halo_info
 * (a rank's halo plan: index tables and their validation, ext/JutulPartitionedArraysExt/utils.jl:91-148) -- and allocate or upload
 * nothing; every entry point that computes refuses it ("this context has no device").  NOT a CPU fallback: it lets a maintainer
 * (and this repository's CPU-only tests) check the set-up tables against Jutul's own and time the set-up on a box without a GPU. */
int32_t jh_context_create_host(jh_context *out);
/* Planning contexts with option plan_checksum = 1: a running checksum of every table a real context would have uploaded so far
 * (pattern, tiles, jagged layouts, ILU(0) levels / maps / programs; in upload order).  Equal checksums = the device would see the
 * same tables: how this repository shows that work on the set-up code leaves them bit for bit as they were. */
int32_t jh_context_plan_checksum(jh_context ctx, int64_t *out);
int32_t jh_context_destroy(jh_context ctx);
int32_t jh_synchronize(jh_context ctx); /* synchronize(ctx), context.jl:72 */
/* Named integer options of a context -- the role of the keyword arguments of the reference's contexts and solver set-up
 * (ParallelCSRContext(nthreads; matrix_layout, minbatch), contexts/csr.jl:3-23; ILUZeroPreconditioner / GenericKrylov kwargs,
 * precond/ilu.jl:1-20, linsolve/krylov.jl:27-58): kernel and path choices a caller may make, read per handle.  Options that
 * shape a layout are read when the object is created (set them before jh_tpfa_create / jh_ilu0_create / the first solve).
 * Unknown key: error.  The environment variable JH_OPTIONS="key=value,key=value" seeds every new context (A/B runs).
 *   consumer_reduce (1)   BiCGStab: the kernel that needs a fused dot product sums its partials itself (no reduction launch)
 *   spmv_jagged (1), spmv_col_bits (0 = by size | 16 | 32), spmv_waves_per_xcd (0 = resident),
 *   spmv_waves (0 = auto | 4 | 8 | 16 wavefronts per workgroup), spmv_window (1), spmv_pipe (1)
 *   spmv_nontemporal (-1 = by working set | 0 | 1)  matrix stream of the jagged SpMV (32-bit columns) with non-temporal loads:
 *                         off when matrix, factors and vectors fit the 256 MB Infinity Cache (read when the layout is built)
 *   sync_loop (0)         1: no speculative Krylov iteration
 *   halo_overlap (0), fused_pack (1), comm_timeout_ms (600000; 0 = wait like a collective)
 *   fused_product (0), fuse_gather (1)
 *   ilu_jagged (1), ilu_threads (0), ilu_factor_kernel (-1 = by pattern | 0 workgroup | 1 wavefront per block),
 *   ilu_factor_threads (0 = 512; rows-form programs 256), ilu_diag_factor (1), ilu_prog (1), ilu_factor_global (0),
 *   ilu_lean_upload (0; 1 = with the chunk-jagged layout and the pivot-only / program factor kernels the row-major structure arrays are
 *                    not uploaded -- written in a session without a GPU, to be validated before it becomes the default),
 *   ilu_factor_wave_per_row (long rows; read when the preconditioner is created: 1 = rows-form programs (scalar matrices) | 2 = instruction-form
 *                            programs, wavefront per row | 0 = thread per row)
 *   asm_pipe (1), asm_pipe2_wgs (0), block_order (0 bisection | 1 onion), block_weights (1: use the face weights of
 *   jh_tpfa_create_weighted | 0: ignore them)
 *   setup_heap (0; process-wide, glibc: while 1, large blocks come from the heap and freed heap memory is kept, so the set-up's later
 *   tables reuse mapped pages -- half the page faults of a first set-up, none in a second one; 0 restores the thresholds and trims)
 *   read_sync (0), setup_timing (0), plan_checksum (0; planning contexts), jds_keep (0), upload_bounce (1; process-wide: 0 copies caller arrays straight from their
 *   pageable memory instead of through the page-locked bounce buffer)
 *   xrank_consumer (-1 = when jh_comm_set_exclusive declared it | 0 | 1)  several ranks: the dot products of the Krylov loop are
 *                         summed over the ranks inside the consuming kernels and the push-halo hand-shake runs inside the product
 *                         kernel -- five launches per BiCGStab iteration as on one rank */
int32_t jh_context_set_option(jh_context ctx, const char *key, int64_t value);
/* Restricts every kernel of the context to the compute units [first_cu, first_cu + n_cus) of the device's CU-mask bit order
 * (hipExtStreamCreateWithCUMask; consecutive bits alternate over the XCDs, so a contiguous range takes the same share of every
 * XCD); n_cus <= 0 removes the mask.  For several ranks on ONE device -- the role of DebugPArrayBackend's ranks inside one
 * process (src/ext/partitionedarrays_ext.jl:37-39) with real concurrency: with disjoint ranges no rank's waiting kernel can keep
 * another rank's kernel off the chip (tools/micro/cu_mask_probe.hip).  Call before any other object of the context is created. */
int32_t jh_context_set_cu_mask(jh_context ctx, int32_t first_cu, int32_t n_cus);
int32_t jh_context_get_option(jh_context ctx, const char *key, int64_t *value);
/* GPU timer on the context stream (HIP events) -- feeds report[:equations_time] & co (simulator.jl:427-433) */
int32_t jh_timer_start(jh_context ctx);
int32_t jh_timer_stop_ms(jh_context ctx, double *ms);

/* ---- a-1..a-4: connectivity, pattern, positions ------------------------------------------------------- */
/* reorder modes for the DEVICE cell numbering (host numbering is never changed; all host-facing arrays
 * are in host numbering). */
#define JH_REORDER_NONE 0   /* device order == host order: ILU(0) identical to the reference's ordering     */
#define JH_REORDER_BLOCKS 1 /* compact graph blocks, BFS inside/between blocks (locality + block-Jacobi)    */
/* TwoPointPotentialFlowHardCoded(N, nc) (flux.jl:172-190) via get_facepos (utils.jl:813-874) and
 * get_connection (flux.jl:128-142); CSR pattern per declare_pattern (conservation.jl:486-505) +
 * static_sparsity_sparse (StaticCSR/mat.jl:73-76).
 * N: 2 x nf neighborship, column-major, 1-based; several faces between one cell pair are accepted with the reference's
 * semantics (one pattern entry per pair, residual / diagonal sum all faces, the off-diagonal holds the last face's derivative:
 * sparse() merge + plain assignment per half-face, conservation.jl:486-505, ad.jl:74-76).  block_n: equations = primary
 * variables per cell.
 * partition (may be NULL): nc entries, 1-based part id per cell; cells of one part become contiguous on the
 * device (used as the block-Jacobi ILU(0) partition, precond/ilu.jl:37-60).  block_rows: target rows per
 * automatically grown block when partition == NULL and reorder == JH_REORDER_BLOCKS (0 = default: 512, or 256
 * below 2M cells).
 * n_owned: for a rank-local subdomain whose cells are [owned..., ghosts...] (ext/JutulPartitionedArraysExt/
 * utils.jl:178-184) the number of owned cells; ghosts stay the last device rows.  <= 0 or nc: no ghosts. */
int32_t jh_tpfa_create(jh_context ctx, int64_t nc, int64_t nf, const int64_t *N, int32_t block_n, int32_t reorder,
                       const int64_t *partition, int64_t block_rows, int64_t n_owned, jh_tpfa *out);
/* The same with |coupling| per face (nf doubles: the transmissibilities; NULL = unweighted): with JH_REORDER_BLOCKS and no
 * partition given, the graph bisection that forms the device blocks -- the block-Jacobi ILU(0) partition -- minimises the cut
 * WEIGHT, i.e. drops weak couplings first.  This is what the reference does: ILUZeroPreconditioner partitions the |A|-weighted
 * graph of the matrix (precond/ilu.jl:37-60 -> generate_lookup -> generate_metis_graph, partitioning.jl:20-51,64-78).  Worth
 * 10-25 % of the BiCGStab iterations on the bench grids (profiles/r05_*weights*).  The weights are only used here: transmissibilities
 * for the assembly are handed over with jh_law_set_data. */
int32_t jh_tpfa_create_weighted(jh_context ctx, int64_t nc, int64_t nf, const int64_t *N, const double *face_weights,
                                int32_t block_n, int32_t reorder, const int64_t *partition, int64_t block_rows, int64_t n_owned,
                                jh_tpfa *out);
int32_t jh_tpfa_destroy(jh_tpfa d);
int32_t jh_tpfa_sizes(jh_tpfa d, int64_t *nc, int64_t *nf, int64_t *nhf, int64_t *nnzb, int32_t *block_n);
/* conn_pos / conn_data (flux.jl:161-190): face_pos[nc+1]; self/other/face/face_sign [nhf] */
int32_t jh_tpfa_get_conn(jh_tpfa d, int64_t *face_pos, int64_t *self, int64_t *other, int64_t *face, int64_t *face_sign);
/* block pattern: rowptr[nc+1], colidx[nnzb], ascending columns, diagonal present */
int32_t jh_tpfa_get_pattern(jh_tpfa d, int64_t *rowptr, int64_t *colidx);
/* jacobian_positions (conservation.jl:143-216, equations.jl:95-113): pos_acc [N*N, nc], pos_flux [N*N, nhf],
 * partial index fastest ((e-1)*np + d), BlockMajorLayout flat index into jac_buffer (1-based). */
int32_t jh_tpfa_get_positions(jh_tpfa d, int64_t *pos_acc, int64_t *pos_flux);
/* JutulMatrixLayout (core_types.jl:101-165).  The device always stores N x N blocks (BlockMajorLayout); the scalar
 * layouts are provided at the boundary: EquationMajor row(cell,e) = (e-1)*nc + cell, EntityMajor row = N*(cell-1) + e
 * (alignment_linear_index, equations.jl:132-138), scalar CSR with ascending columns (models.jl:585-611). */
#define JH_LAYOUT_EQUATION_MAJOR 0
#define JH_LAYOUT_ENTITY_MAJOR 1
#define JH_LAYOUT_BLOCK_MAJOR 2
/* scalar pattern of a layout: rowptr[nc*N + 1], colidx[nnzb*N*N] (block layout: the block pattern) */
int32_t jh_tpfa_get_pattern_layout(jh_tpfa d, int32_t layout, int64_t *rowptr, int64_t *colidx);
/* jacobian_positions for a layout (find_jac_position, equations.jl:4-117): same shapes as jh_tpfa_get_positions */
int32_t jh_tpfa_get_positions_layout(jh_tpfa d, int32_t layout, int64_t *pos_acc, int64_t *pos_flux);
/* device cell order: perm[i] = host cell (1-based) stored at device position i; block_ptr (nblocks+1,
 * 0-based device rows) of the contiguous blocks.  Pass NULL to skip an output. */
int32_t jh_tpfa_get_ordering(jh_tpfa d, int64_t *perm, int64_t *nblocks, int64_t *block_ptr, int64_t block_ptr_cap);
/* Rank-local subdomains (n_owned < nc, JH_REORDER_BLOCKS): the device order is [interior blocks | boundary blocks |
 * ghost blocks]; interior = no cell of the block has a ghost neighbour, so those rows neither feed nor need the ghost
 * exchange (consistent!, ext/JutulPartitionedArraysExt/linalg.jl:46) and the solver overlaps it with them.  Returns the
 * number of leading interior device rows / blocks / SpMV tiles, -1 when the discretisation is not split. */
int32_t jh_tpfa_get_split(jh_tpfa d, int64_t *interior_rows, int64_t *interior_blocks, int64_t *interior_tiles);

/* ---- device vectors ------------------------------------------------------------------------------------- */
/* A vector of nc*block_n doubles attached to a discretisation (cell-ordered: upload/download apply the device
 * permutation) or a plain vector of n doubles attached to a matrix created from a raw pattern. */
int32_t jh_vec_create(jh_tpfa d, jh_vec *out);
int32_t jh_vec_create_for(jh_csr A, jh_vec *out);
int32_t jh_vec_destroy(jh_vec v);
int32_t jh_vec_upload(jh_vec v, const double *host);  /* host layout [N, nc] (block/entity major) */
int32_t jh_vec_download(jh_vec v, double *host);
int32_t jh_vec_fill(jh_vec v, double value);
int32_t jh_vec_copy(jh_vec dst, jh_vec src);
int32_t jh_vec_axpby(jh_vec y, double a, jh_vec x, double b); /* y = a*x + b*y */
int32_t jh_vec_dot(jh_vec a, jh_vec b, double *out);
int32_t jh_vec_length(jh_vec v, int64_t *n);

/* ---- Jacobian / sparse operators (a-10) -------------------------------------------------------------------- */
/* LinearizedSystem.jac + jac_buffer for a TPFA discretisation (linsolve/default.jl:113-129,166-186) */
int32_t jh_csr_create(jh_tpfa d, jh_csr *out);
/* StaticSparsityMatrixCSR(m, n, rowptr, cols, nzval) (StaticCSR/mat.jl:79-83): square n x n, block size bs,
 * 1-based rowptr/colidx with ascending columns; nz may be NULL. */
int32_t jh_csr_create_from_pattern(jh_context ctx, int64_t n, int32_t bs, const int64_t *rowptr, const int64_t *colidx,
                                   const double *nz, jh_csr *out);
int32_t jh_csr_destroy(jh_csr A);
int32_t jh_csr_sizes(jh_csr A, int64_t *n, int64_t *nnzb, int32_t *bs);
/* nonzeros(A) in the HOST pattern order (flat jac_buffer: block k, entry (e,d) at (k-1)N^2 + N(d-1) + e) */
int32_t jh_csr_set_values(jh_csr A, const double *nz);
int32_t jh_csr_get_values(jh_csr A, double *nz);
/* nonzeros(A) of the scalar CSR a layout would build (same values, that layout's slot order) */
int32_t jh_csr_get_values_layout(jh_csr A, int32_t layout, double *nz);
int32_t jh_csr_set_values_layout(jh_csr A, int32_t layout, const double *nz);
/* vectors in a layout's order: equation-major x[(e-1)*nc + cell], entity/block-major x[N*(cell-1) + e]
 * (views at models.jl:1105-1172, utils.jl:65-85) */
int32_t jh_vec_upload_layout(jh_vec v, int32_t layout, const double *host);
int32_t jh_vec_download_layout(jh_vec v, int32_t layout, double *host);
/* mul!(y, A, x, alpha, beta) (StaticCSR/mat.jl:24-39; block: linsolve/block_cpu.jl:1-17) */
int32_t jh_spmv(jh_csr A, jh_vec x, jh_vec y, double alpha, double beta);
/* The same product through the layout the Krylov loop multiplies with: a jagged-slice copy of the values (64-row slices, rows
 * sorted by length, entries stored diagonal by diagonal; refreshed at the start of every jh_bicgstab / jh_newton_step solve).
 * Same accumulation order, hence the same bits as jh_spmv.  Scalar matrices with at most 8 entries per row; error otherwise.
 * Exposed so that the parity tests can check the layout on its own. */
int32_t jh_spmv_jagged(jh_csr A, jh_vec x, jh_vec y, double alpha, double beta);
/* Which product the Krylov loop uses for this matrix, for the host to report: out6 = [0] 1 = jagged-slice kernels (0 = CSR tile
 * kernel: block size > 1 or a row longer than the jagged layout holds), [1] 1 = 16-bit column codes (matrices of 3M rows and more),
 * [2] longest row (block entries), [3] 64-row slices, [4] CSR tiles, [5] entries of the far-column lists (incl. padding). */
int32_t jh_spmv_info(jh_csr A, int64_t *out6);
/* krylov_scale_system! / apply_scaling_to_linearized_system! (linsolve/krylov.jl:194, default.jl:325-385):
 * kind 0 :none, 1 :diagonal (J <- diag(1/|A_ii[j,j]|) J, r likewise), 2 :dt (J <- dt J, r <- dt r).  In place. */
int32_t jh_scale_system(jh_csr A, jh_vec r, int32_t kind, double dt);
/* unit_diagonalize!(r, J, n_self) (ext/JutulPartitionedArraysExt/linalg.jl:18-35): rows of ghost cells -> -I,
 * r_ghost -> 0.  ghost rows are the device rows >= n_owned of a distributed discretisation. */
int32_t jh_unit_diagonalize(jh_csr A, jh_vec r, int64_t n_owned);

/* ---- a-5/a-6/a-8/a-9/a-15: conservation law --------------------------------------------------------------- */
#define JH_LAW_POISSON 0      /* VariablePoissonEquation(TimeDependent) (variable_poisson.jl:90-133), N = 1 */
#define JH_LAW_COMPRESSIBLE 1 /* single-phase slightly compressible, N = 1 (build-defined from flux.jl:335-375) */
#define JH_LAW_TWOPHASE 2     /* immiscible two-phase (p, S_w), N = 2, SPU upwind (flux.jl:382-405)            */
#define JH_LAW_CUSTOM 3       /* physics supplied as source at run time (jh_law_create_custom)                 */
/* params: rho0[2], compressibility[2], viscosity[2], p_ref (7 doubles; NULL = ones/zeros) */
int32_t jh_law_create(jh_tpfa d, int32_t kind, const double *params, jh_law *out);
/* Runtime-defined law (the generic-AD path for user equations, equations.jl:578-594, ad/generic.jl:53-96): `source` is HIP
 * device code defining, for N = block_n of the discretisation (1..3) and the dual type D (value + 2N partials; operators
 * + - * /, dexp dlog dsqrt dpow, face_average, two_point_potential_drop, upwind of flux.jl:335-405 are provided):
 *   __device__ void jh_flux(const D *self, const D *other, double T, double gdz, const double *par, D *q);
 *       q[e] = flux of equation e out of `self` towards `other` across a face with transmissibility T (must be antisymmetric
 *       in (self, other), conservation.jl:397-415);
 *   __device__ void jh_mass(const D *x, const double *par, D *M);
 *       M[e] = conserved quantity per unit volume; the kernel forms vol*(M(x) - M(x0))/dt (conservation.jl:558-568).
 * It is compiled with hiprtc into the tile assembly kernel; the Jacobian comes from the duals.  params: n_params doubles
 * readable as par[]. All other jh_law_* / jh_assemble / jh_newton_step calls work on the returned handle. */
int32_t jh_law_create_custom(jh_tpfa d, const char *source, const double *params, int32_t n_params, jh_law *out);
int32_t jh_law_destroy(jh_law L);
#define JH_FACE_TRANS 0 /* T_f [nf]   (compute_face_trans, finite-volume.jl:224-233)  */
#define JH_FACE_GDZ 1   /* gdz_f [nf] (compute_face_gdz, finite-volume.jl:304-313)    */
#define JH_CELL_VOLUME 2 /* accumulation coefficient vol*phi [nc] (1 for the Poisson law) */
int32_t jh_law_set_data(jh_law L, int32_t which, const double *host);
int32_t jh_law_set_state(jh_law L, const double *X);   /* primary variables [N, nc]                      */
int32_t jh_law_set_state0(jh_law L, const double *X0); /* previous-step state (state0)                   */
int32_t jh_law_get_state(jh_law L, double *X);
/* One primary variable as Jutul stores it -- state[k] / state0[k], a contiguous Float64 array over the cells (host numbering):
 * which 0 = state, 1 = state0; e = 0-based index of the variable.  get_output_state (models.jl:1048-1058) copies state0[k] at
 * every report step: this call is the device -> host transfer behind it.  With the target registered (jh_host_register) the
 * copy is one DMA straight into the host array at PCIe speed. */
int32_t jh_law_get_variable(jh_law L, int32_t which, int32_t e, double *out);
/* hipHostRegister / hipHostUnregister of a host array the caller keeps alive (Jutul's state0[k] arrays live as long as the
 * simulator storage): page-locks it so that uploads / downloads are direct DMA transfers.  Every other caller-owned array that
 * crosses this interface is ordinary (pageable) memory: the library copies it through its own page-locked bounce buffer and the
 * call returns when the copy is complete -- the caller may free or reuse the array at once.
 * A range that is page-locked already (by another component) is success and stays that component's: jh_host_unregister only
 * releases what jh_host_register itself pinned, and is a no-op for anything else. */
int32_t jh_host_register(void *ptr, int64_t bytes);
int32_t jh_host_unregister(void *ptr);
int32_t jh_law_update_state0(jh_law L);                /* state0 <- state (update_after_step!, models.jl:983-1011) */
int32_t jh_law_reset_state(jh_law L);                  /* state <- state0 (timestep cut)                 */
/* forces: PoissonSource-style cell sources added to the residual (models.jl:889-901; variable_poisson.jl:78-84).
 * cells 1-based [n], values [N, n]. */
int32_t jh_law_set_sources(jh_law L, int64_t n, const int64_t *cells, const double *values);
/* (forces are re-applied in every Newton iteration, models.jl:889-901: handing over the list of the previous call again is
 * recognised and costs neither an upload nor a synchronisation) */
/* update_equation! (conservation.jl:572-626) + update_linearized_system_equation! (conservation.jl:298-430)
 * fused: one pass over the CSR-ordered half-faces; writes every nzval slot of A and r. dt <= 0 selects the
 * stationary Poisson variant. */
int32_t jh_assemble(jh_law L, double dt, jh_csr A, jh_vec r);
/* convergence_criterion (equations.jl:619-629): err[e] = max_cells |r[e, cell]| over the first n_owned cells
 * (n_owned <= 0: all) */
int32_t jh_convergence(jh_law L, jh_vec r, int64_t n_owned, double *err);
/* update_primary_variables! (models.jl:928-953, variables/utils.jl:110-174): X += clamp-chain(w*dx) with
 * per-variable scale / abs_max / rel_max / minimum / maximum (5*N doubles, NaN = unset; NULL = no limits). */
int32_t jh_update_primary(jh_law L, jh_vec dx, double w, const double *limits);
/* increment_norm (models.jl:955-965) on the device: out[2e] = sum |dx[e, cell]|, out[2e + 1] = max |dx[e, cell]| over the first
 * n_owned cells (<= 0: all) for every primary variable e -- what update_primary_variables! reports, without dx leaving HBM.  The
 * sums propagate NaN / Inf, so `check = true` (check_increment) is isfinite(out[2e]). */
int32_t jh_increment_norm(jh_law L, jh_vec dx, int64_t n_owned, double *out);
/* variable_change_report(state[k], state0[k], pvar) (models.jl:1023-1038) on the device, called by update_after_step!
 * (models.jl:983-1011) BEFORE state0 <- state: out[4e..4e+3] = sum |x - x0|, max |x - x0|, sum |x|, max |x| of primary
 * variable e over the first n_owned cells (<= 0: all). */
int32_t jh_law_change_report(jh_law L, int64_t n_owned, double *out);
/* The limits jh_newton_step applies in its update (the variables' minimum / maximum / absolute / relative increment limits,
 * variables/utils.jl:110-174; e.g. Saturations: abs_max 0.2, minimum 0, maximum 1): same 5*N layout, NULL = none (default). */
int32_t jh_law_set_update_limits(jh_law L, const double *limits);

/* ---- a-11..a-13: ILU(0) ----------------------------------------------------------------------------------- */
/* ilu0_csr(A) (StaticCSR/ilu0.jl:213-221) when partition == NULL and nparts <= 1; ilu0_csr(A, partition)
 * (par_ilu0.jl:47-55) otherwise.  partition: n entries, 1-based, HOST row numbering; or NULL with nparts == -1
 * to use the discretisation's own device blocks (jh_tpfa_create).  Symbolic phase only. */
int32_t jh_ilu0_create(jh_csr A, const int64_t *partition, int64_t nparts, jh_ilu *out);
int32_t jh_ilu0_destroy(jh_ilu M);
/* ilu0_csr!(LU, A) (ilu0.jl:223-231): numeric factorisation from A's current values */
int32_t jh_ilu0_factor(jh_ilu M);
/* ldiv!(x, LU, b) (ilu0.jl:233-236; apply!, precond/ilu.jl:62-94) */
int32_t jh_ilu0_apply(jh_ilu M, jh_vec b, jh_vec x);
/* x = M^-1 b and q = A x in one call: the pair every iteration of the right-preconditioned Krylov loop evaluates twice
 * (ldiv! then mul!, StaticCSR/ilu0.jl:156-187 + mat.jl:24-61).  With pivot-only factors on one rank (jh_ilu0_stats bit 4) the
 * in-block part of the product is formed inside the apply kernel: the factors then hold A's own in-block entries (U = D~ + U_A,
 * L D~ = L_A), so (A x)_i = A_ii x_i + (g_i - D~_i x_i) + sum_{k<i} L_ik D~_k x_k + sum_{k outside the block} A_ik x_k with g the
 * forward-sweep result -- one pass over the factors instead of factors + matrix; a second, small launch adds the out-of-block
 * entries.  Otherwise (patterns with triangles inside a block, rank-local subdomains, diagonal preconditioners) the two
 * operators run one after the other: same result to rounding. */
int32_t jh_ilu0_apply_mul(jh_ilu M, jh_vec b, jh_vec x, jh_vec q);
/* factor values scattered to A's HOST pattern: L multipliers below the diagonal, inv(U_ii) on it, U above */
int32_t jh_ilu0_get_factor(jh_ilu M, double *lu);
int32_t jh_ilu0_info(jh_ilu M, int64_t *nblocks, int64_t *max_block_rows, int64_t *max_levels);
/* stats4: [0] strict-lower block entries kept in L, [1] strict-upper kept in U (fixed_block, ilu0.jl:13-54),
 * [2] execution blocks, [3] kernel selection: bit 0 = LDS (block-Jacobi) kernels (0: level-per-launch kernels), bit 1 = chunk-jagged
 * layout, bit 2 = program-driven refactorisation available, bit 3 = pivot-only refactorisation in use (no elimination step
 * updates an off-diagonal entry: triangle-free block patterns), bit 4 = the product A*x is fused into the apply of the Krylov loop
 * (jh_ilu0_apply_mul), bit 5 = the programs are in the rows form (long rows of scalar matrices: a wavefront per row, the row in
 * registers) */
int32_t jh_ilu0_stats(jh_ilu M, int64_t *stats4);

/* DiagonalPreconditioner family (precond/diagonal.jl): kind 1 = JacobiPreconditioner(w) D_i = w*inv(A_ii)
 * (precond/jacobi.jl:5-18), kind 2 = SPAI0Preconditioner D_i = A_ii / sum(row entries squared) (precond/spai.jl:40-60).
 * The handle is used like an ILU(0) one: jh_ilu0_factor = update_preconditioner!, jh_ilu0_apply = apply!, usable as
 * the preconditioner of jh_bicgstab / jh_newton_step. */
int32_t jh_diag_precond_create(jh_csr A, int32_t kind, double w, jh_ilu *out);

/* ---- a-14: Krylov ---------------------------------------------------------------------------------------------- */
#define JH_SIDE_NONE 0
#define JH_SIDE_LEFT 1  /* M = prec (distributed path, ext/JutulPartitionedArraysExt/krylov.jl:60)        */
#define JH_SIDE_RIGHT 2 /* N = prec (IterativeSolverConfig default precond_side = :right, linsolve/utils.jl:25) */
int32_t jh_krylov_create(jh_csr A, jh_krylov *out);
int32_t jh_krylov_destroy(jh_krylov K);
/* linear_solve!(sys, GenericKrylov(:bicgstab), ...) core (linsolve/krylov.jl:71-182) = Krylov.jl bicgstab!
 * with x0 = 0, c = r0, stop on ||r|| <= atol + rtol*||r0||, history = true.
 * status: 0 solved, 1 itmax reached, 2 breakdown.  hist: ||r_k||, k = 0..iters (at most hist_cap entries). */
int32_t jh_bicgstab(jh_krylov K, jh_ilu M, int32_t side, jh_vec b, jh_vec x, double rtol, double atol, int64_t itmax,
                    int64_t *iters, int32_t *status, double *hist, int64_t hist_cap);
/* GenericKrylov(:gmres) (linsolve/krylov.jl:214-218; ext/JutulPartitionedArraysExt/krylov.jl:126-148) = Krylov.jl gmres!:
 * MGS Arnoldi + Givens rotations, restart = false (the basis grows with the iteration count), x0 = 0, same stopping
 * rule, status and history conventions as jh_bicgstab. */
int32_t jh_gmres(jh_krylov K, jh_ilu M, int32_t side, jh_vec b, jh_vec x, double rtol, double atol, int64_t itmax,
                 int64_t *iters, int32_t *status, double *hist, int64_t hist_cap);
/* IterativeSolverConfig.min_iterations (linsolve/krylov.jl:120-131, 199-206).  n > 1: like the reference, the solvers
 * run with atol = rtol = 1e-20 and stop as soon as ||r_k|| <= atol + rtol*||r_0|| holds at an iteration k >= n; that exit
 * is reported as status 3 ("user-requested exit": Krylov.jl leaves stats.solved false).  n <= 1: plain stopping rule. */
int32_t jh_krylov_set_min_iterations(jh_krylov K, int64_t min_iterations);
/* Which path the last BiCGStab solve of this workspace took (for the host to report and for tests to assert): out6 = [0] 1 = the
 * dot products were finished by the kernels that consume them (no reduction launches), [1] 1 = over several ranks as well,
 * [2] 1 = products out of the jagged-slice copy, [3] 1 = products fused into the preconditioner apply, [4] 1 = the ghost
 * exchanges of the loop were push halos, [5] 1 = with their hand-shake inside the product kernel (no finish launch). */
int32_t jh_krylov_last_path(jh_krylov K, int64_t *out6);
/* PrecondWrapper-style instrumentation (linsolve/krylov.jl:5-25): accumulated HIP-event time [ms] and launch
 * count of [0] the SpMV and [1] the preconditioner apply inside jh_bicgstab / jh_newton_step.  Reads the
 * totals (ms2/count2 may be NULL), then optionally resets them and enables/disables further profiling.
 * enable = n > 1 times the launches of every n-th Krylov iteration only (an event pair costs ~4.5 us of stream time,
 * which is visible on small per-GPU problems); enable = 1 times every launch. */
int32_t jh_krylov_profile(jh_krylov K, int32_t enable, int32_t reset, double *ms2, int64_t *count2);
/* update_dx_from_vector! (linsolve/default.jl:444-446): dx = -x */
int32_t jh_vec_negate_into(jh_vec dx, jh_vec x);

/* ---- Newton convenience (perform_step!, simulator.jl:392-455) -------------------------------------------------- */
typedef struct {
  double assembly_ms;      /* equations_time + linear_system_time */
  double convergence_ms;   /* convergence_time */
  double precond_ms;       /* update_preconditioner! */
  double linear_solve_ms;  /* linear_solve_time (excl. precond update) */
  double update_ms;        /* update_time */
  double error[2];         /* max|r| per equation before the solve */
  double lin_res0, lin_res; /* first / last Krylov residual norm */
  int64_t linear_iterations;
  int32_t linear_status;
  int32_t converged;       /* all(error < tol) (and solve skipped, check_before_solve = true) */
} jh_newton_report;
/* One Newton iteration: assemble -> convergence check -> ILU factor + BiCGStab + dx = -x + primary update.
 * force_solve: 0 solve unless converged, 1 always solve, -1 never solve (assembly + convergence only).
 * tol: per-equation residual tolerance (1e-3 default in the reference). */
int32_t jh_newton_step(jh_law L, jh_csr A, jh_ilu M, jh_krylov K, jh_vec r, jh_vec dx, double dt, double tol,
                       int32_t force_solve, double rtol, double atol, int64_t itmax, int32_t side,
                       jh_newton_report *rep);

/* ---- a-16/a-17: domain decomposition, halo exchange (RCCL over xGMI) ------------------------------------------- */
/* One process per GPU.  The host side (torch.distributed / MPI / Julia Distributed) only has to broadcast the
 * 128-byte unique id; all data-path collectives run inside the library on the context stream. */
int32_t jh_comm_unique_id(char *id128);
int32_t jh_comm_init(jh_context ctx, int32_t nranks, int32_t rank, const char *id128);
int32_t jh_comm_finalize(jh_context ctx);
/* Mailbox all-reduce for the ranks of ONE node: the Krylov loop's scalar reductions (mpi_scalar_allreduce,
 * ext/.../utils.jl:232-234; 1-2 doubles, three per BiCGStab iteration) bypass ncclAllReduce and go through peer-mapped
 * uncached device memory (hipIpc): one single-wavefront kernel per reduction, summation in rank order (identical bits on
 * every rank); two slot sets alternate with the parity of a device-resident count of EXECUTED reductions.  Protocol: each rank calls jh_comm_ipc_export (64-byte handle of its mailbox), the host all-gathers the
 * handles (MPI.Allgather / torch.distributed), each rank calls jh_comm_ipc_attach (maps the peers and self-tests 64
 * reductions with known answers, time-out guarded: *ok = 0 on any failure), the host ANDs the *ok of all ranks and calls
 * jh_comm_ipc_enable.  Without enable, ncclAllReduce is used.  jh_comm_init_ipc_only creates a communicator without RCCL
 * (scalar reductions only): for tests with several processes on one GPU. */
/* Teardown: peers write into and poll each other's mailboxes / landing buffers, so the host must synchronise the ranks
 * (barrier) after the last solve and before any rank calls jh_comm_finalize / jh_tpfa_destroy. */
int32_t jh_comm_init_ipc_only(jh_context ctx, int32_t nranks, int32_t rank);
/* What carries this rank's data right now, for the host to report and check (simulate_parray builds all ranks itself,
 * ext/JutulPartitionedArraysExt/interface.jl:2-97; a host that launched N processes verifies that N ranks run).
 * out8: [0] ranks of the communicator (1 without one), [1] this rank, [2] ranks RCCL counts in its communicator
 * (ncclCommCount; 0 = no RCCL communicator), [3] 1 = scalar all-reduces through the mailboxes (3 = and finished by the consuming kernels, jh_comm_set_exclusive), [4] 1 = ghost exchanges through
 * the host callback, [5] 1 = in-process backend, [6] waits that timed out so far, [7] time limit of a wait in seconds.
 * Waits inside the mailbox all-reduce / push halo are time-limited (JH_COMM_TIMEOUT_S, default 600; 0 = unbounded): when a
 * peer never arrives, the running solve fails with a jh_last_error message naming this rank, the missing peer and the
 * epoch, instead of hanging every rank of the node. */
int32_t jh_comm_info(jh_context ctx, int64_t *out8);
/* out6: [0] 1 = the ghost exchanges of the Krylov loop are push halos, [1] 1 = receives land directly in the ghost rows,
 * [2] cells sent, [3] cells received, [4] neighbour ranks, [5] owned cells */
int32_t jh_halo_info(jh_tpfa d, int64_t *out6);
/* Host-language halo backend: instead of ncclSend/ncclRecv the library stages the packed send buffer to the host and calls
 * fn(user, send, n_send, recv, n_recv, block_n), which must fill recv (both buffers hold block_n doubles per cell, the
 * neighbours' segments in the order of jh_halo_create) and return 0 -- e.g. MPI.Isend/Irecv from Julia, or gloo.  Used
 * with jh_comm_init or jh_comm_init_ipc_only; fn == NULL restores the RCCL exchange. */
typedef int32_t (*jh_halo_callback)(void *user, const double *send, int64_t n_send, double *recv, int64_t n_recv, int32_t block_n);
int32_t jh_comm_set_halo_callback(jh_context ctx, jh_halo_callback fn, void *user);
int32_t jh_comm_ipc_export(jh_context ctx, char *handle64);
int32_t jh_comm_ipc_attach(jh_context ctx, const char *handles, int32_t *ok);
int32_t jh_comm_ipc_enable(jh_context ctx, int32_t enable);
/* exclusive = 1: no two ranks of the communicator share compute units (one process per GPU -- the deployment -- or CU-masked
 * contexts on one GPU).  With mailboxes enabled the BiCGStab loop then all-reduces its dot
 * products (ext/JutulPartitionedArraysExt/krylov.jl:51-105) inside the kernels that consume them: workgroup 0 stores the local
 * sums into the peers' mailboxes, every wavefront collects the peers' sums and adds them in rank order (identical bits on all
 * ranks), and the push halo's signal / wait / copy (consistent!, linalg.jl:37-55) runs inside the product kernel.  Where ranks
 * share compute units a chip-filling kernel waiting for a peer would keep that peer off the chip: leave it 0 (default).
 * The library checks the declaration against what it has seen: jh_comm_ipc_attach exchanges the PCI bus ids of the ranks' devices
 * (in-process ranks: their device numbers), and exclusive = 1 is an ERROR when two ranks sit on one device and this context has
 * no CU mask (jh_context_set_cu_mask).  jh_comm_devices_distinct reports the observation: 1 every rank on a device of its own,
 * 0 at least two ranks on one device, -1 not observed (mailboxes never attached) -- a host that has no better knowledge passes
 * `distinct == 1` on as `exclusive`. */
int32_t jh_comm_devices_distinct(jh_context ctx, int32_t *distinct);
int32_t jh_comm_set_exclusive(jh_context ctx, int32_t exclusive);
/* Self-test of that path, collective (every rank calls it, after jh_comm_ipc_enable(ctx, 1)): 16 launches of 256 one-wavefront
 * workgroups in which every wavefront collects the peers' sums, waits limited to 5 s; *ok = 1 if every wavefront held the sums in
 * rank order.  Pass the AND of all ranks' answers (and of "every rank has compute units of its own") to jh_comm_set_exclusive --
 * the same all-or-none negotiation as for the mailboxes and the push halo; a failure leaves the reduction-launch path usable. */
int32_t jh_comm_xrank_selftest(jh_context ctx, int32_t *ok);
/* In-process multi-rank backend (the analogue of DebugPArrayBackend / JuliaPArrayBackend,
 * src/ext/partitionedarrays_ext.jl:37-39): ranks are host threads of one process exchanging through host memory;
 * same pack/unpack kernels, halo plans and reduction placement as the RCCL path.  For tests on one GPU. */
int32_t jh_comm_local_group_create(int32_t nranks, void **group);
int32_t jh_comm_local_group_destroy(void *group);
int32_t jh_comm_init_local(jh_context ctx, void *group, int32_t rank);
/* Halo plan of a rank-local discretisation whose cells are [owned..., ghosts...] in HOST numbering
 * (ext/JutulPartitionedArraysExt/utils.jl:9-56,178-184).  For each neighbour rank: the local (1-based) owned
 * cells to send and the local ghost cells to receive, in matching order on both sides. */
int32_t jh_halo_create(jh_tpfa d, int64_t n_owned, int32_t n_nbr, const int32_t *nbr_rank, const int64_t *send_ptr,
                       const int64_t *send_cells, const int64_t *recv_ptr, const int64_t *recv_cells);
/* Push halo for the exchanges INSIDE the Krylov loop (consistent! before every mul!, ext/.../linalg.jl:46), ranks of one
 * node, needs attached mailboxes: the kernel that produces the vector stores each boundary row straight into the landing
 * buffers (peer-mapped uncached memory) of the ranks holding it as a ghost; a finish kernel signals, waits for the
 * neighbours and copies the landed values into the ghost rows.  No collective-library call is left in the loop.  Set-up:
 * jh_halo_ipc_export (64-byte handle of this rank's landing buffer) -> host gathers, per neighbour i, its handle, the
 * first cell of this rank's segment in i's receive order (nbr_offset) and i's total receive count (nbr_stride) ->
 * jh_halo_ipc_attach -> jh_halo_ipc_selftest (one time-limited exchange of a vector whose ghost values the host knows) ->
 * jh_halo_ipc_enable with the AND over all ranks.  All other exchanges (state, right-hand side, solution) and the fallback
 * stay on RCCL / the callback backend. */
int32_t jh_halo_ipc_export(jh_tpfa d, char *handle64);
int32_t jh_halo_ipc_attach(jh_tpfa d, const char *nbr_handles, const int64_t *nbr_offset, const int64_t *nbr_stride, int32_t *ok);
int32_t jh_halo_ipc_selftest(jh_tpfa d, jh_vec v, const double *expected_ghosts, int32_t *ok);
int32_t jh_halo_ipc_enable(jh_tpfa d, int32_t enable);
/* partition(N, nparts) in the role of the reference's MetisPartitioner (partitioning.jl:29-51, generate_metis_graph :64-78;
 * Metis itself is third-party): recursive bisection of the cell graph, every cut grown breadth-first from a pseudo-peripheral
 * cell and refined by Fiduccia-Mattheyses moves; face_weights (may be NULL) weight the edges like the reference's |A|-weighted
 * graph.  out[c] in 1..nparts, every part non-empty, part sizes within `imbalance` (e.g. 0.03; < 0: default) of nc / nparts per
 * bisection.  Host integer work, no device needed.  A partition vector is an input of jh_tpfa_create / jh_ilu0_create and of
 * the host-side subdomain logic: any partitioner can be used instead. */
int32_t jh_partition_graph(int64_t nc, int64_t nf, const int64_t *N, const double *face_weights, int64_t nparts, double imbalance,
                           int64_t *out);
/* Recursive coordinate bisection of nc points (X: point-major, dim = 1..3 coordinates each) into nparts compact parts,
 * out[c] in 1..nparts -- the build-side partitioner when centroids are at hand (role of partitioning.jl:29-51). */
int32_t jh_partition_rcb(int64_t nc, int32_t dim, const double *X, int64_t nparts, int64_t *out);
/* The rank-local subdomain PArraySimulator builds from a partition vector (ext/JutulPartitionedArraysExt/interface.jl:38-63,
 * submap_cells with buffer = 0, dd/subdomains.jl:77-182): local cells = [owned, ascending global id ..., ghosts ...], the
 * faces with both cells local, the local neighbourship and the halo plan.  rank is 1-based like the partition ids.
 * ghost_order 0: ghosts by ascending global id (the reference's order); 1: by (owning rank, global id), so that every
 * neighbour's ghosts are consecutive local cells (jh_halo_create then receives straight into the vectors).
 * sizes[5] = n_owned, n_local, local faces, neighbours, total send cells (total receive cells = n_local - n_owned).
 * jh_subdomain_get copies out (any pointer may be NULL): cells[n_local] and faces[...] are 1-based global ids, N_local the
 * local pairs [l0, r0, l1, r1, ...] (1-based), neighbors 0-based ranks ascending, send/recv_ptr[neighbours + 1] offsets into
 * send/recv_idx (1-based local cells).  Host work on all cores, no device needed. */
int32_t jh_subdomain_create(int64_t nc, int64_t nf, const int64_t *N, const int64_t *partition, int64_t rank, int32_t ghost_order,
                            jh_subdomain *out);
int32_t jh_subdomain_sizes(jh_subdomain s, int64_t *sizes);
int32_t jh_subdomain_get(jh_subdomain s, int64_t *cells, int64_t *faces, int64_t *N_local, int32_t *neighbors, int64_t *send_ptr,
                         int64_t *send_idx, int64_t *recv_ptr, int64_t *recv_idx);
int32_t jh_subdomain_destroy(jh_subdomain s);
/* consistent!(v) (ext/.../linalg.jl:46, krylov.jl:54,75; interface.jl:200): owner values -> ghosts */
int32_t jh_halo_exchange(jh_tpfa d, jh_vec v);
int32_t jh_halo_exchange_state(jh_law L);
/* mpi_scalar_allreduce (ext/.../utils.jl:232-234): op 0 sum, 1 max; n doubles in place on the host */
int32_t jh_allreduce(jh_context ctx, double *values, int32_t n, int32_t op);

#ifdef __cplusplus
}
#endif
#endif
