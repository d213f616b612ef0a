"""The Julia binding (jutul.jl_amd/julia/JutulHIP.jl) cannot run in the build image (no Julia).  This test keeps it honest:
it parses the `@jh :entry_point` calls of every function of the .jl file and then drives several perform_step!s
(simulator.jl:392-455) through its Python twin (jutul.jl_amd/julia_mirror.py; the object bench.py --path seams times) --
asserting that every invocation issues exactly the entry points its Julia twin lists, in the same order (branches the
sequence does not take are named explicitly), that the state stays resident on the device (one upload, no per-iteration
copies), and that the Newton updates equal the oracle's.  If the .jl file and the mirror diverge, or a seam is not wired,
this fails."""
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
    """Stands in for the loaded library: logs (Julia function, invocation number, entry point) for every call."""

    def __init__(self, lib):
        self._lib, self.log, self.ctx, self.inv, self._n = lib, [], None, 0, 0

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("jh_") or name == "jh_last_error":
            return fn

        def wrapped(*a):
            self.log.append((self.ctx, self.inv, name))
            return fn(*a)
        return wrapped

    @contextlib.contextmanager
    def julia(self, fname):
        prev = (self.ctx, self.inv)
        self._n += 1
        self.ctx, self.inv = fname, self._n
        try:
            yield
        finally:
            self.ctx, self.inv = prev

    def calls(self, fname):
        """entry points of every invocation of one Julia function, in order: [[...], [...], ...]"""
        out = {}
        for fn, inv, name in self.log:
            if fn == fname:
                out.setdefault(inv, []).append(name)
        return [out[k] for k in sorted(out)]


def is_subsequence(sub, full):
    it = iter(full)
    return all(any(x == y for y in it) for x in sub)


# entry points of a Julia function that this test's perform_step! sequence does not reach (named so that nothing is skipped silently)
NOT_REACHED = {
    "setup_equation_storage": {"jh_law_create_custom"},                 # physics as device source (generic-AD path)
    "update_linearized_system_equation!": {"jh_csr_get_values", "jh_vec_download"},  # host copies for callers that insist
    "linear_solve!": {"jh_scale_system", "jh_vec_dot", "jh_gmres", "jh_vec_download"},  # scaling, relaxed tolerance, GMRES, hip_download_increment
}
STEP_FUNCTIONS = ("setup_equation_storage", "update_equation!", "update_linearized_system_equation!", "reduce_errors!",
                  "update_preconditioner!", "linear_solve!", "update_primary_variables!", "update_after_step!",
                  "reset_state_to_previous_state!", "sync_host_state!", "get_output_state")


@pytest.mark.gpu
def test_julia_binding_sequence_device_resident_state(oracle):
    """Three Newton iterations over two time steps + a time-step cut + a host-side edit of the state, through the Python twin of
    JutulHIP.jl (jutul.jl_amd/julia_mirror.py, the same object bench.py --path seams times): every invocation issues what its
    Julia function lists, the state is uploaded ONCE, nothing but scalars comes back per iteration, and the trajectory equals
    the oracle's."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import jutul_amd as ja
    from jutul_amd import _lib
    from jutul_amd.julia_mirror import JuliaMirror
    jl = parse_julia_calls()
    for f in STEP_FUNCTIONS + ("setup_distributed!", "set_halo!", "post_update_linearized_system!"):
        assert jl.get(f), f"JutulHIP.jl no longer defines {f} with @jh calls"
    rec = Recorder(_lib.load())
    m = JuliaMirror(rec)

    # ---- problem: compressible single-phase law with gravity, sources on two cells ---------------------------------------------
    g = ja.tet_lattice_mesh(7, 6, 5)
    nc, nf, N = g["nc"], g["nf"], g["N"]
    rng = np.random.default_rng(5)
    T = g["T"] / g["T"].mean()
    vol = g["volumes"]
    gdz = 0.01 * rng.standard_normal(nf)
    par = [1.0, 1.0, 1e-2, 1e-2, 1.0, 1.0, 1.0]
    state0 = 1.0 + 0.1 * rng.random(nc)           # storage.state0 (host)
    state = state0 + 0.01 * rng.standard_normal(nc)  # storage.state (host)
    dt, tol = 0.7, 1e-3
    forces = {3: 0.5, nc - 2: -0.5}               # cell -> value, what apply_forces_to_equation! adds (d[c] += f.value)
    ctx = ja.HIPContext(0)
    s = m.setup_equation_storage(ctx, N, nc, ne=1, law_kind=1, params=par, face_trans=T, face_gdz=gdz, cell_volumes=vol)
    krylov = dict(preconditioner={}, storage=None, solver="bicgstab")

    osys = oracle.TPFASystem(N, nc)
    olaw = oracle.Law("compressible", dt, rho0=(1.0, 1.0), comp=(1e-2, 1e-2), mu=(1.0, 1.0), p_ref=1.0)
    fc = np.array(sorted(forces))
    fv = [forces[int(c)] for c in fc]

    def oracle_newton(x, x0):
        nz_o, r_o = osys.assemble(olaw, x, x0, vol, T, gdz=gdz, src_cells=fc, src_values=fv)
        J = sp.csr_matrix((nz_o, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
        return nz_o, r_o, x - spl.spsolve(J.tocsc(), r_o)

    def perform_step(check_parity=None):
        m.update_equation(s, state, state0, dt)                   # update_state_dependents! -> update_equation!
        d = m.get_diagonal_entries(s)                             # apply_forces! (models.jl:889-901)
        for c, v in forces.items():
            d.add(c, v)
        nz = np.zeros(osys.nnzb) if check_parity else None
        r = np.zeros(nc) if check_parity else None
        m.update_linearized_system_equation(s, nz, r)             # update_linearized_system!
        err = m.convergence_criterion(s)                          # check_convergence
        ok, its, hist = m.linear_solve(s, krylov, rtol=1e-12, atol=1e-14, max_iterations=300)   # solve_and_update!
        assert ok
        norms = m.update_primary_variables(s, check_increment=True)
        return err, norms, nz, r

    # ---- time step 1, Newton iteration 1 and 2 -----------------------------------------------------------------------------------
    x_ref, x0_ref = state.copy(), state0.copy()
    nz_o, r_o, x1 = oracle_newton(x_ref, x0_ref)
    err, norms, nz, r = perform_step(check_parity=True)
    assert np.allclose(r, r_o, rtol=1e-12, atol=1e-13) and np.allclose(nz, nz_o, rtol=1e-12, atol=1e-14)
    assert abs(err[0] - np.abs(r_o).max()) <= 1e-12 * np.abs(r_o).max() and err[0] > tol
    dx_ref = x1 - x_ref
    assert abs(norms[0]["sum"] - np.abs(dx_ref).sum()) <= 1e-6 * np.abs(dx_ref).sum()      # increment_norm (models.jl:955-965)
    assert abs(norms[0]["max"] - np.abs(dx_ref).max()) <= 1e-6 * np.abs(dx_ref).max()
    _, _, x2 = oracle_newton(x1, x0_ref)
    perform_step()
    # the host arrays were never refreshed: the device copy is the state
    assert s.host_state_stale and np.array_equal(state, x_ref)
    rep = m.update_after_step(s)                                                              # end of the ministep
    assert abs(rep[0]["dx"]["max"] - np.abs(x2 - x0_ref).max()) <= 1e-6 * np.abs(x2 - x0_ref).max()    # variable_change_report
    assert abs(rep[0]["x"]["sum"] - np.abs(x2).sum()) <= 1e-9 * np.abs(x2).sum()
    out = m.get_output_state(s, [state0])                                                     # report step: the one download,
    assert np.allclose(out[0], x2, rtol=1e-7, atol=1e-9) and np.array_equal(out[0], state0)   # straight into storage.state0[k]
    assert s.host_state_stale and np.array_equal(state, x_ref)    # the Dual-valued storage.state is refreshed on demand only
    m.sync_host_state(s, state)
    assert np.array_equal(state, state0)

    # ---- time step 2: one iteration, then a time-step cut (reset_state_to_previous_state!) -----------------------------------
    _, _, x3 = oracle_newton(x2, x2)
    perform_step()
    m.reset_state_to_previous_state(s, state, state0)
    m.sync_host_state(s, state)
    assert np.allclose(state, x2, rtol=1e-7, atol=1e-9)          # back at the start of the step, on the device and on the host
    perform_step()
    m.sync_host_state(s, state)
    assert np.allclose(state, x3, rtol=1e-6, atol=1e-8)
    # ---- a host-side writer (reset_variables!) invalidates the device copy: next update_equation! uploads again --------------
    state[:] = x_ref
    state0[:] = x0_ref
    m.invalidate_device_state(s)
    perform_step()
    m.sync_host_state(s, state)
    assert np.allclose(state, x1, rtol=1e-6, atol=1e-8)

    # ---- 1. every invocation issued what its Julia function lists, in order ---------------------------------------------------
    seen = set()
    for fn in {f for f, _, _ in rec.log}:
        listed = jl[fn]
        for called in rec.calls(fn):
            assert is_subsequence(called, listed), (fn, called, listed)
            seen |= set(called)
    for fn in STEP_FUNCTIONS:
        reached = {n for c in rec.calls(fn) for n in c}
        assert reached, f"the sequence never entered {fn}"
        missing = set(jl[fn]) - reached
        assert missing <= NOT_REACHED.get(fn, set()), f"{fn}: JutulHIP.jl also calls {sorted(missing)}; the mirror does not"
    lib = _lib.load()
    for fn, names in jl.items():
        for nm in names:
            assert hasattr(lib, nm), (fn, nm)
    # ---- 2. residency: 6 perform_step!s, 2 state uploads (first use + after the host edit), sources uploaded once ------------
    ue = rec.calls("update_equation!")
    assert [len(c) for c in ue] == [2, 0, 0, 0, 2] or [len(c) for c in ue if c] == [2, 2], ue
    assert sum("jh_law_set_sources" in c for c in rec.calls("update_linearized_system_equation!")) == 1
    per_iteration = {n for f, _, n in rec.log if f in ("update_primary_variables!", "linear_solve!", "update_after_step!")}
    assert not per_iteration & {"jh_vec_download", "jh_law_get_state", "jh_law_set_state", "jh_law_set_state0"}
    m.unregister([state0])
    m.destroy(s, krylov)


def test_two_equation_bookkeeping_of_the_twin_needs_no_device():
    """CPU part of the above: SourceRows writes the member's forces into its component of the group's accumulator, and the .jl file
    defines the member methods as no-ops (no entry point may hide in them)."""
    from jutul_amd.julia_mirror import HIPEquationMember, SourceAccumulator, SourceRows
    acc = SourceAccumulator(3, 10)
    SourceRows(acc, 1, 2).add(4, 0.5)           # second equation, first component -> row 2 of the block
    SourceRows(acc, 1, 2).add(4, 0.25, e=2)
    acc.add(4, 1.0)
    acc.add(9, -1.0, 3)
    assert acc.cells == [4, 9] and acc.vals == [1.0, 0.5, 0.25, 0.0, 0.0, -1.0]
    src = open(JL).read()
    for line in ("update_equation!(m::HIPEquationMember, law::ConservationLaw, storage, model, dt) = nothing",
                 "update_linearized_system_equation!(nz, r, model, law::ConservationLaw, m::HIPEquationMember) = nothing",
                 "get_diagonal_entries(eq::ConservationLaw, m::HIPEquationMember) = SourceRows(m.group.sources, m.offset, m.ne)"):
        assert line in src, line
    jl = parse_julia_calls()
    assert jl["reduce_errors!"] == ["jh_convergence"] and not jl.get("convergence_criterion")
    assert HIPEquationMember(1, 1).group is None


def test_julia_binding_names_match_header():
    """Every @jh entry point of JutulHIP.jl is declared in include/jutul_hip.h (same spelling)."""
    hdr = open(os.path.join(ROOT, "include", "jutul_hip.h")).read()
    declared = set(re.findall(r"int32_t\s+(jh_\w+)\s*\(", hdr))
    used = {nm for names in parse_julia_calls().values() for nm in names}
    used |= set(re.findall(r"ccall\(\(:(jh_\w+)", open(JL).read()))
    assert used and used <= declared, sorted(used - declared)
