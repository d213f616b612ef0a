"""The Julia binding (jutul.jl_amd/julia/JutulHIP.jl) cannot run in the build image (no Julia).  This test keeps it honest:
it parses the `@jh :entry_point` calls of every function of the .jl file and then performs ONE perform_step! (simulator.jl:392-455)
through ctypes with a Python mirror of those functions -- asserting that each mirror issues exactly the entry points its Julia
twin lists, in the same order (branches the default path does not take are named explicitly) -- and compares the Newton update
with the oracle's.  If the .jl file and the mirror diverge, or a seam is not wired (data upload, state0, forces, convergence,
primary update, dx download), this fails."""
import contextlib
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "jutul.jl_amd", "julia", "JutulHIP.jl")


def parse_julia_calls():
    """{function name: [entry points in source order]} for block (`function f(...) ... end`) and one-line definitions."""
    out, cur = {}, None
    for line in open(JL).read().splitlines():
        m = re.match(r"^function\s+([\w.!]+)\s*\(", line)
        if m:
            cur = m.group(1).split(".")[-1]
            out.setdefault(cur, [])
        one = re.match(r"^([\w.!]+)\(.*\)\s*=\s*@jh\s+:(\w+)", line)
        if one and cur is None:
            out.setdefault(one.group(1).split(".")[-1], []).append(one.group(2))
            continue
        if cur is not None:
            out[cur] += re.findall(r"@jh\s+:(\w+)", line)
        if line.startswith("end") and cur is not None:
            cur = None
    return out


class Recorder:
    """Stands in for the loaded library: logs (current Julia function, entry point) for every call."""

    def __init__(self, lib):
        self._lib, self.log, self.ctx = lib, [], None

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("jh_") or name == "jh_last_error":
            return fn

        def wrapped(*a):
            self.log.append((self.ctx, name))
            return fn(*a)
        return wrapped

    @contextlib.contextmanager
    def julia(self, fname):
        prev, self.ctx = self.ctx, fname
        try:
            yield
        finally:
            self.ctx = prev


def is_subsequence(sub, full):
    it = iter(full)
    return all(any(x == y for y in it) for x in sub)


# entry points of a Julia function that the DEFAULT perform_step! does not reach (named so that nothing is skipped silently)
NOT_ON_DEFAULT_PATH = {
    "setup_equation_storage": {"jh_law_create_custom"},                 # physics as device source (generic-AD path)
    "update_linearized_system_equation!": {"jh_csr_get_values", "jh_vec_download"},  # host copies for callers that insist
    "linear_solve!": {"jh_scale_system", "jh_vec_dot", "jh_gmres"},     # :diagonal/:dt scaling, relaxed tolerance, GMRES
}


@pytest.mark.gpu
def test_julia_binding_sequence_for_one_newton_step(oracle):
    import jutul_amd as ja
    from jutul_amd import _lib
    from jutul_amd._lib import check, f64, i64, pf, pi
    jl = parse_julia_calls()
    for f in ("setup_equation_storage", "update_equation!", "update_linearized_system_equation!", "convergence_criterion",
              "update_preconditioner!", "linear_solve!", "update_primary_variables!", "update_after_step!",
              "reset_state_to_previous_state!", "setup_distributed!", "set_halo!", "post_update_linearized_system!"):
        assert jl.get(f), f"JutulHIP.jl no longer defines {f} with @jh calls"
    L = Recorder(_lib.load())
    H = C.c_void_p

    # ---- problem: compressible single-phase law with gravity, sources on two cells ---------------------------------------------
    g = ja.tet_lattice_mesh(7, 6, 5)
    nc, nf, N = g["nc"], g["nf"], g["N"]
    rng = np.random.default_rng(5)
    T = g["T"] / g["T"].mean()
    vol = g["volumes"]
    gdz = 0.01 * rng.standard_normal(nf)
    par = f64([1.0, 1.0, 1e-2, 1e-2, 1.0, 1.0, 1.0])
    state0 = 1.0 + 0.1 * rng.random(nc)
    state = state0 + 0.01 * rng.standard_normal(nc)
    dt, tol = 0.7, 1e-3
    forces = {3: 0.5, nc - 2: -0.5}                     # cell -> value, what apply_forces_to_equation! adds (d[c] += f.value)

    ctx = ja.HIPContext(0)
    # -- setup_equation_storage (conservation.jl:137)
    with L.julia("setup_equation_storage"):
        disc, law, jac, r, dx = H(), H(), H(), H(), H()
        Nf = i64(np.asfortranarray(N).T.reshape(-1))
        check(L.jh_tpfa_create(ctx.h, nc, nf, pi(Nf), 1, 1, None, 0, 0, C.byref(disc)))
        check(L.jh_law_create(disc, 1, pf(par), C.byref(law)))
        check(L.jh_law_set_data(law, 0, pf(f64(T))))
        check(L.jh_law_set_data(law, 1, pf(f64(gdz))))
        check(L.jh_law_set_data(law, 2, pf(f64(vol))))
        check(L.jh_csr_create(disc, C.byref(jac)))
        check(L.jh_vec_create(disc, C.byref(r)))
        check(L.jh_vec_create(disc, C.byref(dx)))
    sources = np.zeros(nc)                              # HIPConservationLawStorage.sources
    # -- update_equation! (conservation.jl:572)
    with L.julia("update_equation!"):
        check(L.jh_law_set_state(law, pf(f64(state))))
        check(L.jh_law_set_state0(law, pf(f64(state0))))
        sources[:] = 0.0
    # -- apply_forces! (models.jl:889-901): d = get_diagonal_entries(eq, eq_s); d[c] += f.value  (host accumulator, no ccall)
    for c, v in forces.items():
        sources[c - 1] += v
    # -- update_linearized_system_equation! (conservation.jl:298)
    with L.julia("update_linearized_system_equation!"):
        cells = i64(np.flatnonzero(sources) + 1)
        check(L.jh_law_set_sources(law, cells.size, pi(cells), pf(f64(sources[cells - 1]))))
        check(L.jh_assemble(law, dt, jac, r))
    # -- check_convergence -> convergence_criterion (equations.jl:619-629)
    with L.julia("convergence_criterion"):
        err = np.zeros(1)
        check(L.jh_convergence(law, r, nc, pf(err)))
    converged = bool(err[0] < tol)
    # -- linear_solve! (krylov.jl:71-182), default GenericKrylov(:bicgstab, preconditioner = HIPILUZero())
    ilu, ks = H(), H()
    with L.julia("linear_solve!"):
        with L.julia("update_preconditioner!"):
            check(L.jh_ilu0_create(jac, None, -1, C.byref(ilu)))
            check(L.jh_ilu0_factor(ilu))
        check(L.jh_krylov_create(jac, C.byref(ks)))
        check(L.jh_krylov_set_min_iterations(ks, 1))
        iters, status = C.c_int64(), C.c_int32()
        hist = np.zeros(302)
        check(L.jh_bicgstab(ks, ilu, 2, r, dx, 1e-12, 1e-14, 300, C.byref(iters), C.byref(status), pf(hist), hist.size))
        check(L.jh_vec_negate_into(dx, dx))
        dx_buffer = np.zeros(nc)
        check(L.jh_vec_download(dx, pf(dx_buffer)))
    assert status.value == 0
    # -- update_primary_variables! (models.jl:928-953)
    with L.julia("update_primary_variables!"):
        check(L.jh_update_primary(law, dx, 1.0, None))
        X = np.zeros(nc)
        check(L.jh_law_get_state(law, pf(X)))
    # -- update_after_step! (models.jl:983-1011)
    with L.julia("update_after_step!"):
        check(L.jh_law_update_state0(law))

    # ---- 1. the mirror issued what the Julia functions list, in order ------------------------------------------------------------
    by_fn = {}
    for fn, name in L.log:
        by_fn.setdefault(fn, []).append(name)
    for fn, called in by_fn.items():
        listed = jl[fn]
        assert is_subsequence(called, listed), (fn, called, listed)
        missing = set(listed) - set(called)
        assert missing <= NOT_ON_DEFAULT_PATH.get(fn, set()), f"{fn}: JutulHIP.jl also calls {sorted(missing)}; the mirror does not"
    assert set(by_fn) == {"setup_equation_storage", "update_equation!", "update_linearized_system_equation!", "convergence_criterion",
                          "update_preconditioner!", "linear_solve!", "update_primary_variables!", "update_after_step!"}
    # every entry point the .jl file names exists in the library with the header's name
    lib = _lib.load()
    for fn, names in jl.items():
        for nm in names:
            assert hasattr(lib, nm), (fn, nm)

    # ---- 2. the step equals the oracle's Newton step --------------------------------------------------------------------------------
    osys = oracle.TPFASystem(N, nc)
    olaw = oracle.Law("compressible", dt, rho0=(1.0, 1.0), comp=(1e-2, 1e-2), mu=(1.0, 1.0), p_ref=1.0)
    fc = np.array(sorted(forces))
    nz_o, r_o = osys.assemble(olaw, state, state0, vol, T, gdz=gdz, src_cells=fc, src_values=[forces[int(c)] for c in fc])
    # read the device Jacobian / residual back for the comparison
    nzv = np.zeros(osys.nnzb)
    check(lib.jh_csr_get_values(jac, pf(nzv)))
    rv = np.zeros(nc)
    check(lib.jh_vec_download(r, pf(rv)))
    assert np.allclose(rv, r_o, rtol=1e-12, atol=1e-13)
    assert np.allclose(nzv, nz_o, rtol=1e-12, atol=1e-14)
    assert abs(err[0] - np.abs(r_o).max()) <= 1e-12 * np.abs(r_o).max() and not converged
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    J = sp.csr_matrix((nz_o, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
    x_ref = spl.spsolve(J.tocsc(), r_o)
    assert np.allclose(dx_buffer, -x_ref, rtol=1e-6, atol=1e-8)       # linear_solve! writes dx = -x
    assert np.allclose(X, state - x_ref, rtol=1e-6, atol=1e-8)         # update_primary_variables!: state += dx
    # update_after_step!: state0 <- state on the device: the next assembly sees a vanishing accumulation term
    check(lib.jh_law_set_sources(law, 0, None, None))
    check(lib.jh_assemble(law, dt, jac, r))
    check(lib.jh_vec_download(r, pf(rv)))
    _, r_flux = osys.assemble(olaw, X, X, vol, T, gdz=gdz)
    assert np.allclose(rv, r_flux, rtol=1e-11, atol=1e-12)
    for h, d in ((ks, "jh_krylov_destroy"), (ilu, "jh_ilu0_destroy"), (dx, "jh_vec_destroy"), (r, "jh_vec_destroy"),
                 (jac, "jh_csr_destroy"), (law, "jh_law_destroy"), (disc, "jh_tpfa_destroy")):
        getattr(lib, d)(h)


def test_julia_binding_names_match_header():
    """Every @jh entry point of JutulHIP.jl is declared in include/jutul_hip.h (same spelling)."""
    hdr = open(os.path.join(ROOT, "include", "jutul_hip.h")).read()
    declared = set(re.findall(r"int32_t\s+(jh_\w+)\s*\(", hdr))
    used = {nm for names in parse_julia_calls().values() for nm in names}
    used |= set(re.findall(r"ccall\(\(:(jh_\w+)", open(JL).read()))
    assert used and used <= declared, sorted(used - declared)
