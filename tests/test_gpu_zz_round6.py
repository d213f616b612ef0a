"""GPU tests WRITTEN IN A ROUND WITHOUT A GPU (round 6: the lease was closed from outside the build).  They sort last on purpose:
`pytest -x` reaches them only after every test that has run on a GPU before.  What they cover:
  * a model with two conservation laws through the twin of JutulHIP.jl (one device block, per-equation forces and convergence);
  * the ORACLE PIN of the consumer-side cross-rank reduction (granule all-reduce + folded halo hand-shake) -- 2 CU-masked processes,
    residual history against the oracle-driven N-rank emulation (tests/_dd_emulation.py);
  * the liveness promise: ranks sharing an unmasked device are seen by the library and jh_comm_set_exclusive(1) is refused.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_julia_sequence import Recorder, is_subsequence, parse_julia_calls

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(world, port, worker, extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    env.pop("NCCL_DEBUG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", worker)]
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_consumer_side_allreduce_history_is_the_oracle_history(world):
    """xrank_worker.py with JH_TEST_ORACLE_PIN=1: rank 0 gathers every rank's matrix, right-hand side, halo plan and device
    ordering and compares the granule path's residual history with the oracle emulation: first six residuals 1e-8, ten 1e-7,
    iterations +-1."""
    r = _torchrun(world, 29700 + world, "xrank_worker.py", {"JH_TEST_KIND": "poisson", "JH_TEST_ORACLE_PIN": "1"})
    assert r.returncode == 0 and f"XRANK_OK {world} poisson" in r.stdout and f"XRANK_ORACLE_OK {world}" in r.stdout, \
        r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_exclusive_promise_is_refused_for_ranks_sharing_an_unmasked_device():
    r = _torchrun(2, 29711, "ipc_allreduce_worker.py", {"JH_TEST_EXCLUSIVE_REFUSAL": "1"})
    assert r.returncode == 0 and "EXCLUSIVE_REFUSED_OK 2" in r.stdout and "IPC_ALLREDUCE_OK 2" in r.stdout, \
        r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_julia_binding_two_conservation_laws_share_one_block(oracle):
    """A model with TWO conservation laws on Cells (water and oil mass balance, one component each: Jutul's loops over
    model.equations, models.jl:549-572,774-783,830-883,889-901) through the twin of JutulHIP.jl: the first law's storage is the
    group that assembles the 2x2 block system, the second law's storage is a member that names its row.  Forces are applied per
    equation through get_diagonal_entries (apply_forces!), convergence is reported per equation from ONE device reduction per
    assembly, and the Newton trajectory equals the oracle's block system (immiscible two-phase law)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import jutul_amd as ja
    from jutul_amd import _lib
    from jutul_amd.julia_mirror import HIPEquationMember, JuliaMirror
    jl = parse_julia_calls()
    rec = Recorder(_lib.load())
    m = JuliaMirror(rec)
    g = ja.tet_lattice_mesh(6, 5, 4)
    nc, nf, N = g["nc"], g["nf"], g["N"]
    rng = np.random.default_rng(11)
    T = g["T"] / g["T"].mean()
    vol = g["volumes"]
    gdz = 0.01 * rng.standard_normal(nf)
    par = [1.0, 0.8, 1e-2, 2e-2, 1.0, 2.0, 1.0]      # rho0[2], compressibility[2], viscosity[2], p_ref
    X0 = np.stack([1.0 + 0.1 * rng.random(nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)    # [p, S_w] per cell
    X = X0.copy()
    dt = 0.4
    ctx = ja.HIPContext(0)
    group, member = m.setup_equation_storages(ctx, N, nc, [1, 1], law_kind=2, params=par, face_trans=T, face_gdz=gdz, cell_volumes=vol)
    assert isinstance(member, HIPEquationMember) and member.group is group and (member.offset, member.ne) == (1, 1) and group.N == 2
    eqs = [group, member]                             # storage[:equations] in equation order
    krylov = dict(preconditioner={}, storage=None, solver="bicgstab")
    # forces per equation: d[c] += v on the equation's own diagonal entries
    forces = [{2: 0.02, nc - 1: -0.015}, {2: 0.0, nc - 1: -0.02, 7: 0.01}]
    src_cells = sorted(set(forces[0]) | set(forces[1]))
    src_vals = np.array([[forces[0].get(c, 0.0), forces[1].get(c, 0.0)] for c in src_cells]).reshape(-1)
    osys = oracle.TPFASystem(N, nc, 2)
    olaw = oracle.Law("twophase", dt, rho0=(1.0, 0.8), comp=(1e-2, 2e-2), mu=(1.0, 2.0), p_ref=1.0)

    def oracle_newton(x, x0):
        nz_o, r_o = osys.assemble(olaw, x, x0, vol, T, gdz, src_cells, src_vals)
        A = sp.bsr_matrix((nz_o.reshape(-1, 2, 2).transpose(0, 2, 1), osys.colidx - 1, osys.rowptr - 1), shape=(2 * nc, 2 * nc))
        return nz_o, r_o, x - spl.spsolve(A.tocsc(), r_o)

    def perform_step(parity=False):
        for s in eqs:
            m.update_equation(s, X, X0, dt)                          # update_equations! over all equations
        for s, f in zip(eqs, forces):                                # apply_forces! over all equations
            d = m.get_diagonal_entries(s)
            for c, v in f.items():
                d.add(c, v)
        nz = np.zeros(osys.nnzb * 4) if parity else None
        r = np.zeros(2 * nc) if parity else None
        m.update_linearized_system_equation(eqs[0], nz, r)           # update_linearized_system! over all equations
        m.update_linearized_system_equation(eqs[1])
        errs = [m.convergence_criterion(s) for s in eqs]             # check_convergence over all equations
        ok, its, _ = m.linear_solve(group, krylov, rtol=1e-10, atol=1e-13, max_iterations=400)
        assert ok
        m.update_primary_variables(group, check_increment=True)
        return errs, nz, r

    nz_o, r_o, x1 = oracle_newton(X, X0)
    errs, nz, r = perform_step(parity=True)
    assert np.allclose(r, r_o, rtol=1e-12, atol=1e-13) and np.allclose(nz, nz_o, rtol=1e-12, atol=1e-14)
    rb = np.abs(r_o.reshape(nc, 2))
    assert len(errs[0]) == 1 and len(errs[1]) == 1               # every equation reports its own component
    assert abs(errs[0][0] - rb[:, 0].max()) <= 1e-12 * rb[:, 0].max() and abs(errs[1][0] - rb[:, 1].max()) <= 1e-12 * rb[:, 1].max()
    _, _, x2 = oracle_newton(x1, X0)
    perform_step()
    host = np.zeros(2 * nc)
    group.host_state_stale = True
    m.sync_host_state(group, host)
    assert np.allclose(host, x2, rtol=1e-6, atol=1e-8)
    # one reduction per assembly although two equations asked; the member's functions issue nothing
    red = rec.calls("reduce_errors!")            # (invocations that issued an entry point: one per assembly, not one per equation)
    assert [c for c in red] == [["jh_convergence"], ["jh_convergence"]], red
    assert sum("jh_assemble" in c for c in rec.calls("update_linearized_system_equation!")) == 2
    # the second equation's sources reached the device as the second component of the block
    assert sum("jh_law_set_sources" in c for c in rec.calls("update_linearized_system_equation!")) == 1
    for fn in {f for f, _, _ in rec.log}:
        for called in rec.calls(fn):
            assert is_subsequence(called, jl[fn]), (fn, called, jl[fn])
    m.destroy(group, krylov)
