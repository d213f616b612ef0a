"""GPU tests of the domain-decomposed path on ONE GPU: ranks are host threads joined by the in-process
LocalCommGroup (the DebugPArrayBackend analogue); same kernels / halo plans / reductions as the RCCL path.
Property pinned (the reference has no test here): N-rank result == 1-rank result to solver tolerance."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


def run_ranks(nranks, fn):
    out, err = [None] * nranks, [None] * nranks

    def wrap(r):
        try:
            out[r] = fn(r)
        except Exception as e:  # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=wrap, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for e in err:
        if e is not None:
            raise e
    return out


def problem(ja, dims=(10, 9, 8), kind="poisson", seed=0):
    g = ja.tet_lattice_mesh(*dims)
    rng = np.random.default_rng(seed)
    nblk = 2 if kind == "twophase" else 1
    nc = g["nc"]
    if nblk == 1:
        X0 = 1.0 + 0.1 * rng.random(nc)
    else:
        X0 = np.stack([1.0 + 0.1 * rng.random(nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)
    return g, g["T"] / g["T"].mean(), X0, nblk


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("nranks", [2, 3])
@pytest.mark.parametrize("kind", ["poisson", "twophase"])
def test_distributed_newton_matches_single_rank(ja, kind, nranks, side):
    from jutul_amd import dd
    g, T, X0, nblk = problem(ja, kind=kind)
    nc = g["nc"]
    par = dict(rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
    dt = 0.5
    q = 1.0 if nblk == 1 else 0.02  # keep the two-phase saturations inside (0, 1) without update limits
    src = ([1, nc], np.array([[q] * nblk, [-q] * nblk]))

    def make_sim(law):
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                              relative_tolerance=1e-11, max_iterations=300, precond_side=side)
        return ja.Simulator(law, ks, tolerance=1e-9)

    # single rank reference with the same side (left = the reference's MPI path, ext/.../krylov.jl:60; right = Jutul's
    # single-process default, which on the device runs the fused kernels also in the distributed case)
    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, block_n=nblk, reorder="blocks", block_rows=256)
    law0 = ja.ConservationLaw(disc0, kind, **par)
    law0.set_face_trans(T)
    law0.set_volumes(g["volumes"])
    law0.set_state(X0)
    law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    sim0 = make_sim(law0)
    ok0, its0, _ = sim0.solve_ministep(dt)
    assert ok0
    X_ref = law0.get_state().reshape(nc, nblk)

    part = dd.partition_rcb(g["cell_centroids"], nranks)
    group = ja.LocalCommGroup(nranks)

    def rank_fn(r):
        ctx = ja.HIPContext(0)
        ctx.comm_init_local(group, r)
        disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, kind=kind, block_n=nblk,
                                               sources=src, block_rows=256, law_params=par)
        sim = make_sim(law)
        ok, its, rep = sim.solve_ministep(dt)
        X = law.get_state().reshape(-1, nblk)
        ctx.comm_finalize()
        return ok, its, sub, X

    res = run_ranks(nranks, rank_fn)
    X = np.zeros_like(X_ref)
    for ok, its, sub, Xl in res:
        assert ok
        own = sub["cells"][: sub["n_owned"]] - 1
        X[own] = Xl[: sub["n_owned"]]
        # ghosts carry the owner's values after the final consistent!(x) + update
        gh = sub["cells"][sub["n_owned"]:] - 1
        assert np.allclose(Xl[sub["n_owned"]:], X_ref[gh], rtol=1e-7, atol=1e-9)
    assert np.abs(X - X_ref).max() <= 1e-7 * np.abs(X_ref).max()
    assert all(r[1] == res[0][1] for r in res)  # every rank takes the same number of Newton iterations


def test_halo_exchange_and_unit_diagonalize(ja):
    """consistent!(v) fills every ghost with its owner's value; unit_diagonalize! turns ghost rows into -I."""
    from jutul_amd import dd
    g, T, X0, _ = problem(ja, dims=(6, 5, 4))
    nc = g["nc"]
    nranks = 3
    part = dd.partition_rcb(g["cell_centroids"], nranks)
    group = ja.LocalCommGroup(nranks)
    glob = np.arange(1, nc + 1, dtype=np.float64) * 1.5

    def rank_fn(r):
        ctx = ja.HIPContext(0)
        ctx.comm_init_local(group, r)
        disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, block_rows=64)
        loc = glob[sub["cells"] - 1].copy()
        loc[sub["n_owned"]:] = -777.0  # stale ghosts
        v = ja.DeviceVector(disc, loc)
        ja._lib.check(ja._lib.load().jh_halo_exchange(disc.h, v.h))
        out = v.download()
        lsys = ja.LinearizedSystem(disc)
        law.update_equation_and_linearized_system(0.5, lsys.jac, lsys.r)
        ja._lib.check(ja._lib.load().jh_unit_diagonalize(lsys.jac.h, lsys.r.h, sub["n_owned"]))
        rowptr, colidx = disc.pattern()
        nz, rr = lsys.jac.nzval, lsys.r.download()
        ctx.comm_finalize()
        return sub, out, rowptr, colidx, nz, rr

    for sub, out, rowptr, colidx, nz, rr in run_ranks(nranks, rank_fn):
        assert np.array_equal(out, glob[sub["cells"] - 1])
        no = sub["n_owned"]
        assert np.all(rr[no:] == 0.0)
        for row in range(no, sub["n_local"]):
            cols = colidx[rowptr[row] - 1: rowptr[row + 1] - 1]
            vals = nz[rowptr[row] - 1: rowptr[row + 1] - 1]
            assert np.array_equal(vals, np.where(cols == row + 1, -1.0, 0.0))


def test_rccl_single_rank_communicator(ja):
    """The RCCL code path with a 1-rank communicator (dlopen librccl, ncclCommInitRank, in-stream all-reduce)."""
    from jutul_amd import dd
    g, T, X0, _ = problem(ja, dims=(6, 5, 4))
    nc = g["nc"]
    ctx = ja.HIPContext(0)
    ctx.comm_init(1, 0, ja.HIPContext.comm_unique_id())
    assert np.allclose(ctx.allreduce([1.5, -2.0], "sum"), [1.5, -2.0])
    part = np.ones(nc, dtype=np.int64)
    disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, 0, T, g["volumes"], X0, sources=([1, nc], [1.0, -1.0]))
    assert sub["n_owned"] == nc and len(sub["neighbors"]) == 0
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                                             relative_tolerance=1e-10, precond_side="left"), tolerance=1e-8)
    ok, its, rep = sim.solve_ministep(0.5)
    assert ok and its == 2
    ctx.comm_finalize()


def test_ghost_cells_do_not_break_lds_blocks(ja):
    """A rank-local subdomain with many ghosts must still get LDS-sized ILU blocks (ghost rows are -I rows)."""
    from jutul_amd import dd
    g, T, X0, _ = problem(ja, dims=(24, 20, 16))
    part = dd.partition_rcb(g["cell_centroids"], 2)
    ctx = ja.HIPContext(0)
    disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, 0, T, g["volumes"], X0, block_rows=128)
    assert sub["n_local"] - sub["n_owned"] > 500
    perm, bp = disc.ordering()
    assert np.diff(bp).max() <= 160
    assert np.all(perm[sub["n_owned"]:] > sub["n_owned"])  # ghosts stay the last device rows
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(0.5, lsys.jac, lsys.r)
    prec = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    assert prec.info()["lds_mode"]


def test_interior_boundary_split_of_a_subdomain(ja):
    """Device order of a rank-local subdomain: interior blocks (no ghost neighbour) first, then boundary blocks, then ghosts.
    Every cell that is sent to a neighbour lies in a boundary block; interior rows reference no ghost column."""
    from jutul_amd import dd
    g, T, X0, _ = problem(ja, dims=(24, 20, 16))
    part = dd.partition_rcb(g["cell_centroids"], 4)
    ctx = ja.HIPContext(0)
    disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, 1, T, g["volumes"], X0, block_rows=128)
    n_owned, n_local = sub["n_owned"], sub["n_local"]
    rows_i, blocks_i, tiles_i = disc.split()
    perm, bp = disc.ordering()
    assert 0 < rows_i < n_owned and 0 < blocks_i < len(bp) - 1 and tiles_i > 0
    assert bp[blocks_i] == rows_i
    iperm = np.empty(n_local, dtype=np.int64)
    iperm[perm - 1] = np.arange(n_local)
    N = sub["N"] - 1
    ghost_nbr = np.zeros(n_local, dtype=bool)          # host cells with a ghost neighbour
    ghost_nbr[N[0][N[1] >= n_owned]] = True
    ghost_nbr[N[1][N[0] >= n_owned]] = True
    dev_has_ghost = ghost_nbr[perm - 1]
    assert not dev_has_ghost[:rows_i].any()
    # every boundary block really touches a ghost (otherwise it would have been classified interior)
    owned_blocks = np.searchsorted(bp, n_owned)
    for b in range(blocks_i, owned_blocks):
        assert dev_has_ghost[bp[b]:bp[b + 1]].any()
    for cells in sub["send"]:
        assert np.all(iperm[np.asarray(cells) - 1] >= rows_i)
    assert np.all(perm[n_owned:] > n_owned)
    # a single-process discretisation is not split
    disc1 = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], reorder="blocks")
    assert disc1.split() == (-1, -1, -1)


def test_overlapped_halo_exchange_path():
    """The opt-in overlapped ghost exchange (option halo_overlap = 1: second stream, split interior/boundary SpMV) must give the
    same distributed Newton results; the switch is read once per process, hence the subprocess."""
    import os, subprocess, sys
    env = dict(os.environ, JH_OPTIONS="halo_overlap=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_distributed.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_distributed_newton_matches_single_rank or test_eight_ranks_many_neighbours"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_eight_ranks_many_neighbours(ja):
    """8 ranks (the node size the bench targets), RCB parts with up to 7 neighbours each: plan symmetry, ghost consistency and
    the distributed right-preconditioned Newton step == single rank."""
    from jutul_amd import dd
    g, T, X0, _ = problem(ja, dims=(20, 18, 16), seed=3)
    nc = g["nc"]
    nranks = 8
    part = dd.partition_rcb(g["cell_centroids"], nranks)
    subs = [dd.local_subdomain(g["N"], part, r + 1, ghost_order="owner") for r in range(nranks)]
    assert max(len(s["neighbors"]) for s in subs) >= 3
    for r, s in enumerate(subs):  # what r sends to q is what q expects from r, in the same global order
        for q, snd in zip(s["neighbors"], s["send"]):
            sq = subs[int(q)]
            j = list(sq["neighbors"]).index(r)
            assert np.array_equal(s["cells"][snd - 1], sq["cells"][sq["recv"][j] - 1])
    dt = 0.5
    src = ([1, nc], np.array([[1.0], [-1.0]]))

    def make_sim(law):
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                              relative_tolerance=1e-11, max_iterations=300, precond_side="right")
        return ja.Simulator(law, ks, tolerance=1e-9)

    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, reorder="blocks", block_rows=256)
    law0 = ja.ConservationLaw(disc0, "poisson")
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    ok0, _, _ = make_sim(law0).solve_ministep(dt)
    assert ok0
    X_ref = law0.get_state()
    group = ja.LocalCommGroup(nranks)

    def rank_fn(r):
        ctx = ja.HIPContext(0)
        ctx.comm_init_local(group, r)
        disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, sources=src, block_rows=256,
                                               ghost_order="owner")
        ok, its, rep = make_sim(law).solve_ministep(dt)
        X = law.get_state()
        ctx.comm_finalize()
        return ok, sub, X

    X = np.zeros(nc)
    for ok, sub, Xl in run_ranks(nranks, rank_fn):
        assert ok
        X[sub["cells"][: sub["n_owned"]] - 1] = Xl[: sub["n_owned"]]
        assert np.allclose(Xl[sub["n_owned"]:], X_ref[sub["cells"][sub["n_owned"]:] - 1], rtol=1e-7, atol=1e-9)
    assert np.abs(X - X_ref).max() <= 1e-7 * np.abs(X_ref).max()


def test_distributed_gmres_matches_single_rank(ja):
    """GMRES is the second Krylov method of the reference's distributed path (ext/.../krylov.jl:126-148)."""
    from jutul_amd import dd
    g, T, X0, _ = problem(ja, dims=(9, 8, 7), seed=5)
    nc = g["nc"]
    src = ([1, nc], np.array([[1.0], [-1.0]]))
    dt = 0.5

    def run_one(ctx, disc, law):
        lsys = ja.LinearizedSystem(disc)
        law.synchronize_ghosts() if disc.n_owned < disc.nc else None
        law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
        if disc.n_owned < disc.nc:
            ja._lib.check(ja._lib.load().jh_unit_diagonalize(lsys.jac.h, lsys.r.h, disc.n_owned))
        ks = ja.GenericKrylov("gmres", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-11,
                              max_iterations=150, precond_side="left")
        out = ja.linear_solve(lsys, ks)
        return out, lsys.dx.download()

    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, reorder="blocks", block_rows=128)
    law0 = ja.ConservationLaw(disc0, "poisson")
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    out0, dx_ref = run_one(ctx0, disc0, law0)
    assert out0["ok"]
    nranks = 3
    part = dd.partition_rcb(g["cell_centroids"], nranks)
    group = ja.LocalCommGroup(nranks)

    def rank_fn(r):
        ctx = ja.HIPContext(0)
        ctx.comm_init_local(group, r)
        disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, sources=src, block_rows=128)
        out, dx = run_one(ctx, disc, law)
        ctx.comm_finalize()
        return out, sub, dx

    dx = np.zeros(nc)
    its = set()
    for out, sub, dxl in run_ranks(nranks, rank_fn):
        assert out["ok"]
        its.add(out["iterations"])
        dx[sub["cells"][: sub["n_owned"]] - 1] = dxl[: sub["n_owned"]]
    assert len(its) == 1
    assert np.abs(dx - dx_ref).max() <= 1e-7 * np.abs(dx_ref).max()


def test_custom_law_on_two_ranks_matches_builtin_single_rank(ja):
    """A runtime-defined law (jh_law_create_custom) goes through the same distributed machinery: the built-in compressible law
    restated as user source on 2 ranks == the built-in law on one rank."""
    from jutul_amd import dd
    src_code = r"""
__device__ D rho(D p, const double *par) { return par[0] * dexp(par[1] * (p - par[3])); }
__device__ void jh_flux(const D *self, const D *other, double T, double gdz, const double *par, D *q) {
  D rs = rho(self[0], par), ro = rho(other[0], par);
  q[0] = (T / par[2]) * (face_average(rs, ro) * two_point_potential_drop(self[0], other[0], gdz, rs, ro));
}
__device__ void jh_mass(const D *x, const double *par, D *M) { M[0] = rho(x[0], par); }
"""
    g, T, X0, _ = problem(ja, dims=(9, 8, 7), seed=5)
    nc = g["nc"]
    dt = 0.5
    src = ([1, nc], np.array([[0.6], [-0.6]]))

    def make_sim(law):
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-11,
                              max_iterations=300)
        return ja.Simulator(law, ks, tolerance=1e-9)

    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, reorder="blocks", block_rows=128)
    law0 = ja.ConservationLaw(disc0, "compressible", rho0=(1.2, 1.0), compressibility=(0.05, 0.0), viscosity=(0.8, 1.0), p_ref=1.0)
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    ok0, its0, _ = make_sim(law0).solve_ministep(dt)
    assert ok0 and its0 >= 2
    X_ref = law0.get_state()
    part = dd.partition_rcb(g["cell_centroids"], 2)
    group = ja.LocalCommGroup(2)

    def rank_fn(r):
        ctx = ja.HIPContext(0)
        ctx.comm_init_local(group, r)
        disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, kind="custom", sources=src, block_rows=128,
                                               law_params=dict(source=src_code, params=[1.2, 0.05, 0.8, 1.0]))
        ok, its, _ = make_sim(law).solve_ministep(dt)
        X = law.get_state()
        ctx.comm_finalize()
        return ok, its, sub, X

    X = np.zeros(nc)
    for ok, its, sub, Xl in run_ranks(2, rank_fn):
        assert ok and its == its0
        X[sub["cells"][: sub["n_owned"]] - 1] = Xl[: sub["n_owned"]]
    assert np.abs(X - X_ref).max() <= 1e-7 * np.abs(X_ref).max()


# ---- the distributed preconditioner's SEMANTICS against the oracle (not only the converged answer) ----------------------------
from tests._dd_emulation import emulated_bicgstab as _emulated_bicgstab, oracle_rank  # noqa: E402


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("nranks", [3, 8])
def test_distributed_preconditioner_history_matches_oracle_emulation(ja, nranks, side):
    """parray_preconditioner_apply! (ext/JutulPartitionedArraysExt/linalg.jl:1-35,78-101) pinned against the oracle: a converged
    answer is the same for any preconditioner, the residual HISTORY is not.  Device: nranks in-process ranks, rtol 1e-8; oracle:
    the N-rank emulation above with the device's matrices, block partition and elimination order.  First 10 residuals to 1e-8,
    iteration counts +-1, owned solution to 1e-6 of its scale."""
    from jutul_amd import dd
    from jutul_amd._lib import load as L, check as ck_rc
    from oracle import oracle as o
    from tests import _ilu_checks as ck
    g, T, X0, _ = problem(ja, dims=(10, 9, 8))
    nc = g["nc"]
    dt, rtol = 0.5, 1e-8
    src = ([1, nc], np.array([[1.0], [-1.0]]))
    part = dd.partition_rcb(g["cell_centroids"], nranks)
    group = ja.LocalCommGroup(nranks)

    def rank_fn(r):
        ctx = ja.HIPContext(0)
        ctx.comm_init_local(group, r)
        disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, sources=src, block_rows=64)
        lsys = ja.LinearizedSystem(disc)
        law.synchronize_ghosts()
        law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
        ck_rc(L().jh_unit_diagonalize(lsys.jac.h, lsys.r.h, disc.n_owned))
        nz, b = lsys.jac.nzval, lsys.r.download()
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=rtol,
                              absolute_tolerance=0.0, max_iterations=200, precond_side=side)
        out = ja.linear_solve(lsys, ks)
        perm, bp = disc.ordering()
        x = -lsys.dx.download()
        ctx.comm_finalize()
        return dict(out=out, sub=sub, nz=nz, b=b, perm=perm, bp=bp, x=x)

    res = run_ranks(nranks, rank_fn)
    ranks = [oracle_rank(o, ck, rr["sub"], rr["nz"], rr["b"], rr["perm"], rr["bp"]) for rr in res]
    hist_o, its_o, x_o = _emulated_bicgstab(o, ranks, side, rtol)
    for k, rr in enumerate(res):
        out = rr["out"]
        assert out["ok"]
        assert abs(out["iterations"] - its_o) <= 1, (out["iterations"], its_o)
        m = min(11, len(hist_o), len(out["residuals"]))
        assert m >= 6
        assert np.allclose(out["residuals"][:m], hist_o[:m], rtol=1e-8, atol=0.0), (side, nranks, out["residuals"][:m], hist_o[:m])
        no = ranks[k]["no"]
        assert np.abs(rr["x"][:no] - x_o[k][:no]).max() <= 1e-6 * np.abs(x_o[k][:no]).max()
