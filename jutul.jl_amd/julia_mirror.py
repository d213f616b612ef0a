"""Call-for-call Python twin of julia/JutulHIP.jl (Julia is absent from the build image).

Every method below carries the name of the Julia function it mirrors and issues exactly the `@jh :entry_point` calls that
function lists, in the same order, through ctypes.  Two users:
  * tests/test_gpu_julia_sequence.py runs it on a recording library, checks each method's entry points against what it
    parses out of the .jl file, and compares the Newton steps with the oracle;
  * bench.py --path seams times THIS sequence (what `simulate!` would drive through the binding) beside the fused
    jh_newton_step line.
State residency follows the .jl file: after the first upload the device copy of the primary variables IS the state
(models.jl:928-953, conservation.jl:549-555: the reference never copies them either); `device_state_valid`,
`device_state0_valid` and `host_state_stale` are the same three flags.
"""
import contextlib
import ctypes as C

import numpy as np

from . import _lib
from ._lib import JutulHIPError, check, f64, i64, pf, pi

H = C.c_void_p


class SourceAccumulator:
    """SourceAccumulator of the .jl file: what get_diagonal_entries hands to apply_forces! (models.jl:889-901).  Forces do
    `d[c] += v` (1-based cell); touched cells are remembered, so the flush is O(#sources)."""

    def __init__(self, N, nc):
        self.N, self.nc = N, nc
        self.slot, self.cells, self.vals = {}, [], []

    def add(self, cell, value, e=1):
        j = self.slot.get(cell)
        if j is None:
            j = self.slot[cell] = len(self.cells)
            self.cells.append(cell)
            self.vals.extend([0.0] * self.N)
        self.vals[j * self.N + e - 1] += value

    def reset(self):  # reset_sources! (conservation.jl:660-664)
        self.slot, self.cells, self.vals = {}, [], []


class SourceRows:
    """SourceRows of the .jl file: get_diagonal_entries of a MEMBER equation -- its rows of the group's accumulator."""

    def __init__(self, acc, offset, ne):
        self.acc, self.offset, self.ne = acc, offset, ne

    def add(self, cell, value, e=1):
        assert 1 <= e <= self.ne
        self.acc.add(cell, value, self.offset + e)


class HIPEquationMember:
    """HIPEquationMember of the .jl file: a later conservation law of a model with several (models.jl:549-572): rows
    offset+1 .. offset+ne of the block the first law's storage (the group) assembles."""

    def __init__(self, offset, ne):
        self.group, self.offset, self.ne = None, int(offset), int(ne)


class HIPConservationLawStorage:
    def __init__(self):
        self.disc, self.law, self.jac, self.r, self.dx = H(), H(), H(), H(), H()
        self.nc = self.N = self.n_owned = 0
        self.dt = float("nan")
        self.X = None
        self.sources = None
        self.src_cells, self.src_vals = [], []
        self.err = None
        self.device_state_valid = self.device_state0_valid = self.host_state_stale = self.host_state0_stale = self.registered = False
        self.assemblies, self.err_of_assembly = 0, -1


class JuliaMirror:
    """lib: the loaded library or a recorder wrapping it (anything with the jh_* entry points and, optionally, a
    `julia(name)` context manager that labels the calls of one Julia function)."""

    def __init__(self, lib=None):
        self.L = lib if lib is not None else _lib.load()
        self.copy_output = True  # Jutul's own get_output_state copies state0[k] (models.jl:1054)

    def _fn(self, name):
        j = getattr(self.L, "julia", None)
        return j(name) if j is not None else contextlib.nullcontext()

    # ---- setup_equation_storage (conservation.jl:137) -----------------------------------------------------------------------------
    def setup_equation_storage(self, ctx, N, nc, ne=1, law_kind=0, params=None, face_trans=None, face_gdz=None, cell_volumes=None,
                               block_rows=0, n_owned=0, law_source=None):
        L, s = self.L, HIPConservationLawStorage()
        with self._fn("setup_equation_storage"):
            N = np.asarray(N, dtype=np.int64)
            Nf = i64(np.asfortranarray(N).T.reshape(-1))
            check(L.jh_tpfa_create_weighted(ctx.h, nc, N.shape[1], pi(Nf), pf(f64(face_trans)), ne, 1, None, int(block_rows), int(n_owned),
                                            C.byref(s.disc)))
            par = f64(params) if params is not None else None
            if law_source is None:
                check(L.jh_law_create(s.disc, int(law_kind), pf(par), C.byref(s.law)))
            else:
                check(L.jh_law_create_custom(s.disc, law_source.encode(), pf(par), 0 if par is None else par.size, C.byref(s.law)))
            check(L.jh_law_set_data(s.law, 0, pf(f64(face_trans))))
            if face_gdz is not None:
                check(L.jh_law_set_data(s.law, 1, pf(f64(face_gdz))))
            if cell_volumes is not None:
                check(L.jh_law_set_data(s.law, 2, pf(f64(cell_volumes))))
            check(L.jh_csr_create(s.disc, C.byref(s.jac)))
            check(L.jh_vec_create(s.disc, C.byref(s.r)))
            check(L.jh_vec_create(s.disc, C.byref(s.dx)))
        s.nc, s.N = int(nc), int(ne)
        s.n_owned = int(n_owned) if n_owned > 0 else int(nc)
        s.X = np.zeros(nc * ne)
        s.sources = SourceAccumulator(ne, nc)
        s.err = np.zeros(ne)
        return s

    def setup_equation_storages(self, ctx, N, nc, equations, **kw):
        """Jutul's loop over model.equations (setup_storage_equations!, models.jl:549-572) for a model with several conservation
        laws on Cells: `equations` = components of every law, in equation order.  The first call of the .jl file's
        setup_equation_storage builds the group (block size = all components), every later one returns a member; the members are
        linked when hip_equation_storage first runs (setup_linearized_system!)."""
        group = self.setup_equation_storage(ctx, N, nc, ne=int(sum(equations)), **kw)
        group.ne_own = int(equations[0])
        out, off = [group], int(equations[0])
        for ne in equations[1:]:
            with self._fn("setup_equation_storage"):      # (the member branch issues no entry point)
                out.append(HIPEquationMember(off, ne))
            off += int(ne)
        for m in out[1:]:                                 # hip_equation_storage
            m.group = group
        return out

    def adopt(self, disc, law, lsys):
        """Storage over handles that already exist (jutul_amd objects): bench.py builds the problem once and times either path."""
        s = HIPConservationLawStorage()
        s.disc, s.law, s.jac, s.r, s.dx = disc.h, law.h, lsys.jac.h, lsys.r.h, lsys.dx.h
        s.nc, s.N, s.n_owned = disc.nc, law.N, disc.n_owned
        s.ne_own = s.N
        s.X = np.zeros(s.nc * s.N)
        s.sources = SourceAccumulator(s.N, s.nc)
        s.err = np.zeros(s.N)
        return s

    # ---- update_equation! (conservation.jl:572; state_pair :549-555) -----------------------------------------------------------
    def update_equation(self, s, state, state0, dt):
        if isinstance(s, HIPEquationMember):     # update_equation!(m::HIPEquationMember, ...) = nothing
            return
        with self._fn("update_equation!"):
            if not s.device_state_valid:
                check(self.L.jh_law_set_state(s.law, pf(f64(np.asarray(state).reshape(-1)))))
                s.device_state_valid, s.host_state_stale = True, False
            if not s.device_state0_valid:
                check(self.L.jh_law_set_state0(s.law, pf(f64(np.asarray(state0).reshape(-1)))))
                s.device_state0_valid = True
                s.host_state0_stale = False
            s.sources.reset()
            s.dt = float(dt)

    def get_diagonal_entries(self, s):
        if isinstance(s, HIPEquationMember):
            return SourceRows(s.group.sources, s.offset, s.ne)
        return s.sources

    # ---- update_linearized_system_equation! (conservation.jl:298) ------------------------------------------------------------
    def update_linearized_system_equation(self, s, nz=None, r=None):
        if isinstance(s, HIPEquationMember):     # the group's kernel has assembled the member's rows
            return
        L, a = self.L, s.sources
        with self._fn("update_linearized_system_equation!"):
            if a.cells != s.src_cells or a.vals != s.src_vals:
                cells, vals = i64(a.cells), f64(a.vals)
                check(L.jh_law_set_sources(s.law, cells.size, pi(cells) if cells.size else None, pf(vals) if vals.size else None))
                s.src_cells, s.src_vals = list(a.cells), list(a.vals)
            check(L.jh_assemble(s.law, s.dt, s.jac, s.r))
            s.assemblies += 1
            if nz is not None:
                check(L.jh_csr_get_values(s.jac, pf(nz)))
            if r is not None:
                check(L.jh_vec_download(s.r, pf(r)))

    # ---- host <-> device state hand-over ----------------------------------------------------------------------------------------
    def invalidate_device_state(self, s, state=True, state0=True):
        if state:
            s.device_state_valid = False
        if state0:
            s.device_state0_valid = False

    def sync_host_state(self, s, host_state):
        with self._fn("sync_host_state!"):
            if s.host_state_stale:
                check(self.L.jh_law_get_state(s.law, pf(s.X)))
                host_state[:] = s.X.reshape(host_state.shape)
                s.host_state_stale = False
        return host_state

    def get_output_state(self, s, host_state0_vars):
        """get_output_state (models.jl:1048-1058): the regular device -> host transfer, once per report step.
        host_state0_vars: storage.state0[k] for every primary variable k (contiguous Float64 arrays over the cells)."""
        with self._fn("get_output_state"):
            if s.host_state0_stale:
                for e, v0 in enumerate(host_state0_vars):
                    assert v0.dtype == np.float64 and v0.flags.c_contiguous and v0.size == s.nc
                    if not s.registered:
                        check(self.L.jh_host_register(v0.ctypes.data_as(C.c_void_p), v0.nbytes))
                    check(self.L.jh_law_get_variable(s.law, 1, e, pf(v0)))
                s.registered = True
                s.host_state0_stale = False
        return {e: v.copy() for e, v in enumerate(host_state0_vars)} if self.copy_output else None

    def unregister(self, host_state0_vars):
        for v0 in host_state0_vars:
            self.L.jh_host_unregister(v0.ctypes.data_as(C.c_void_p))

    # ---- convergence_criterion (equations.jl:619-629) ---------------------------------------------------------------------------
    def reduce_errors(self, s):
        """reduce_errors!: max |r_e| of every component of the block, once per assembly"""
        with self._fn("reduce_errors!"):
            if s.err_of_assembly != s.assemblies:
                check(self.L.jh_convergence(s.law, s.r, s.n_owned, pf(s.err)))
                s.err_of_assembly = s.assemblies
        return s.err

    def convergence_criterion(self, s):
        """the group reports the rows of its own law (all of them for a single-equation model), a member reports its rows"""
        with self._fn("convergence_criterion"):
            if isinstance(s, HIPEquationMember):
                return self.reduce_errors(s.group)[s.offset:s.offset + s.ne].copy()
            return self.reduce_errors(s)[:getattr(s, "ne_own", s.N)].copy()

    # ---- update_preconditioner! (precond/ilu.jl:37) -------------------------------------------------------------------------------
    def update_preconditioner(self, prec, s):
        with self._fn("update_preconditioner!"):
            if not prec.get("handle"):
                with self._fn("hip_preconditioner_create!"):   # HIPILUZero (kind None) / HIPJacobi (1, w) / HIPSPAI0 (2)
                    h = H()
                    if prec.get("kind"):
                        check(self.L.jh_diag_precond_create(s.jac, int(prec["kind"]), float(prec.get("w", 1.0)), C.byref(h)))
                    else:
                        check(self.L.jh_ilu0_create(s.jac, None, -1, C.byref(h)))
                    prec["handle"] = h
            check(self.L.jh_ilu0_factor(prec["handle"]))

    # ---- linear_solve! (linsolve/krylov.jl:71-182) ------------------------------------------------------------------------------
    def linear_solve(self, s, krylov, rtol=1e-3, atol=1e-12, max_iterations=100, min_iterations=1, side=2, download_increment=None):
        """krylov: dict(preconditioner=dict(), storage=None, solver='bicgstab').  Returns (ok, iterations, residual history)."""
        L = self.L
        with self._fn("linear_solve!"):
            self.update_preconditioner(krylov["preconditioner"], s)
            if krylov.get("storage") is None:
                h = H()
                check(L.jh_krylov_create(s.jac, C.byref(h)))
                krylov["storage"] = h
            ws = krylov["storage"]
            check(L.jh_krylov_set_min_iterations(ws, int(min_iterations)))
            iters, status = C.c_int64(), C.c_int32()
            hist = np.zeros(max_iterations + 2)
            solve = L.jh_gmres if krylov.get("solver") == "gmres" else L.jh_bicgstab
            check(solve(ws, krylov["preconditioner"]["handle"], side, s.r, s.dx, float(rtol), float(atol), int(max_iterations),
                        C.byref(iters), C.byref(status), pf(hist), hist.size))
            check(L.jh_vec_negate_into(s.dx, s.dx))
            n = iters.value
            solved = status.value == 0
            manual = min_iterations > 1
            bad = (manual and n == max_iterations) or (not manual and not solved)
            if bad and n > 0 and hist[n] / hist[0] > 1.0:
                raise JutulHIPError(f"Bad linear solve: final residual {hist[n]}, rel. value {hist[n] / hist[0]}")
            if download_increment is not None:  # hip_download_increment(model) == true
                check(L.jh_vec_download(s.dx, pf(download_increment)))
        return solved, n, hist[: n + 1].copy()

    # ---- update_primary_variables! (models.jl:928-965) -------------------------------------------------------------------------
    def update_primary_variables(self, s, relaxation=1.0, check_increment=False, limits=None):
        norms = np.zeros(2 * s.N)
        with self._fn("update_primary_variables!"):
            check(self.L.jh_increment_norm(s.law, s.dx, s.n_owned, pf(norms)))
            if check_increment and not np.isfinite(norms).all():
                raise JutulHIPError("Primary variables recieved invalid updates.")
            lim = f64(limits).reshape(-1) if limits is not None else None
            check(self.L.jh_update_primary(s.law, s.dx, float(relaxation), pf(lim)))
            s.host_state_stale = True
        return [dict(sum=norms[2 * e], max=norms[2 * e + 1]) for e in range(s.N)]

    # ---- update_after_step! (models.jl:983-1011) / reset_state_to_previous_state! (:1068-1073) ---------------------------------
    def update_after_step(self, s):
        rep4 = np.zeros(4 * s.N)
        with self._fn("update_after_step!"):
            check(self.L.jh_law_change_report(s.law, s.nc, pf(rep4)))   # all local cells, like length(X) in the reference
            check(self.L.jh_law_update_state0(s.law))
            s.host_state0_stale = True
        return [dict(dx=dict(sum=rep4[4 * e], max=rep4[4 * e + 1]), x=dict(sum=rep4[4 * e + 2], max=rep4[4 * e + 3]), n=s.nc)
                for e in range(s.N)]

    def reset_state_to_previous_state(self, s, host_state=None, host_state0=None):
        with self._fn("reset_state_to_previous_state!"):
            if s.device_state0_valid:
                check(self.L.jh_law_reset_state(s.law))
                s.device_state_valid, s.host_state_stale = True, True
            else:
                host_state[:] = host_state0
                s.device_state_valid = False

    def destroy(self, s, krylov=None):
        lib = _lib.load()
        if krylov is not None:
            if krylov.get("storage") is not None:
                lib.jh_krylov_destroy(krylov["storage"])
            if krylov["preconditioner"].get("handle"):
                lib.jh_ilu0_destroy(krylov["preconditioner"]["handle"])
        for h, d in ((s.dx, "jh_vec_destroy"), (s.r, "jh_vec_destroy"), (s.jac, "jh_csr_destroy"), (s.law, "jh_law_destroy"),
                     (s.disc, "jh_tpfa_destroy")):
            getattr(lib, d)(h)
