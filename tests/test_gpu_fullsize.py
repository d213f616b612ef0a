"""Size-independent properties at BASELINE.json's full sizes (the oracle would take minutes there):
10M-cell single-phase grid (configs[1]/[4] family) and a 5M-cell two-phase grid (configs[3]): row sums, conservation,
linearity of the residual, SpMV adjoint identity, ILU-preconditioned solve residual check with the GPU SpMV; and the
1M-cell case decomposed over 8 ranks (configs[2]) against the single-rank Newton update."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


def test_10M_cells_single_phase_properties(ja):
    from bench import dims_for_cells
    ctx = ja.HIPContext(0)
    g = ja.tet_lattice_mesh(*dims_for_cells(10_000_000))
    nc = g["nc"]
    assert nc > 9_900_000
    T = g["T"] / g["T"].mean()
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks")
    perm, bp = disc.ordering()
    assert np.array_equal(np.bincount(perm, minlength=nc + 1)[1:], np.ones(nc, dtype=np.int64))  # a permutation
    assert bp[-1] == nc and np.diff(bp).max() <= 640
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(T)
    law.set_volumes(g["volumes"])
    rng = np.random.default_rng(0)
    U = 1.0 + 0.1 * rng.random(nc)
    dt = 5.0
    lsys = ja.LinearizedSystem(disc)
    law.set_state(U)
    law.set_state0(np.zeros(nc))
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    r1 = lsys.r.download()
    # (i) conservation: fluxes cancel pairwise -> sum(r) == sum(vol*U/dt)
    assert np.isclose(r1.sum(), (g["volumes"] * U).sum() / dt, rtol=1e-9)
    # (ii) row sums of the Jacobian: J*1 == vol/dt
    ones = ja.DeviceVector(disc, np.ones(nc))
    y = ja.mul_(ja.DeviceVector(disc), lsys.jac, ones).download()
    assert np.abs(y - g["volumes"] / dt).max() < 1e-9 * np.abs(y).max() + 1e-12
    # (iii) linearity of the residual for the Poisson law: r(U) == J*U for state0 = 0 (no sources)
    JU = ja.mul_(ja.DeviceVector(disc), lsys.jac, ja.DeviceVector(disc, U)).download()
    assert np.abs(JU - r1).max() <= 1e-11 * np.abs(r1).max()
    # (iv) symmetry of the TPFA Jacobian: <x, J y> == <J x, y>
    xv, yv = rng.standard_normal(nc), rng.standard_normal(nc)
    dx, dy = ja.DeviceVector(disc, xv), ja.DeviceVector(disc, yv)
    Jy = ja.mul_(ja.DeviceVector(disc), lsys.jac, dy)
    Jx = ja.mul_(ja.DeviceVector(disc), lsys.jac, dx)
    a, b = dx.dot(Jy), Jx.dot(dy)
    assert abs(a - b) <= 1e-9 * max(abs(a), abs(b))
    # (v) one Newton step solves the linear problem: the second assembly is converged, true residual checked with SpMV
    law.set_state0(U)
    law.set_sources([1, nc], [1.0, -1.0])
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                          relative_tolerance=1e-8, max_iterations=300)
    sim = ja.Simulator(law, ks, tolerance=1e-5)  # ||r||_2 drops by 1e-8 from ~6e2: max|r| lands near 1e-6
    rep = sim.perform_step(dt, 1)
    assert rep.linear_status == 0 and rep.lin_res <= 1.0001e-8 * rep.lin_res0
    Jdx = ja.mul_(ja.DeviceVector(disc), sim.lsys.jac, sim.lsys.dx)  # J*dx + r == 0
    assert Jdx.axpby(1.0, sim.lsys.r, 1.0).dot(Jdx) ** 0.5 <= 2e-8 * sim.lsys.r.dot(sim.lsys.r) ** 0.5
    rep2 = sim.perform_step(dt, 2)
    assert rep2.converged == 1


def test_5M_cells_two_phase_block_properties(ja):
    from bench import dims_for_cells
    ctx = ja.HIPContext(0)
    g = ja.tet_lattice_mesh(*dims_for_cells(5_000_000))
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=2, reorder="blocks")
    par = dict(rho0=(1.0, 0.8), compressibility=(1e-3, 2e-3), viscosity=(1.0, 2.0), p_ref=1.0)
    law = ja.ConservationLaw(disc, "twophase", **par)
    law.set_face_trans(T)
    law.set_volumes(g["volumes"])
    rng = np.random.default_rng(1)
    X0 = np.stack([1.0 + 0.05 * rng.random(nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)
    law.set_state(X0)
    law.set_state0(X0)
    dt = 0.5
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    r = lsys.r.download().reshape(nc, 2)
    # state == state0 -> accumulation vanishes; fluxes cancel pairwise per phase -> sum(r_e) == 0 (mass conservation)
    scale = np.abs(r).sum(0)
    assert np.all(np.abs(r.sum(0)) <= 1e-9 * scale)
    # Jacobian vs directional finite difference of the residual (block 2x2 assembly + block SpMV)
    d = rng.standard_normal(nc * 2) * np.tile([1.0, 0.1], nc)
    Jd = ja.mul_(ja.DeviceVector(disc), lsys.jac, ja.DeviceVector(disc, d)).download()
    h = 1e-6
    r2 = ja.DeviceVector(disc)
    jac2 = ja.StaticSparsityMatrixCSR(disc)
    law.set_state(X0 + h * d)
    law.update_equation_and_linearized_system(dt, jac2, r2)
    rp = r2.download()
    law.set_state(X0 - h * d)
    law.update_equation_and_linearized_system(dt, jac2, r2)
    rm = r2.download()
    fd = (rp - rm) / (2 * h)
    # SPU upwinding has kinks where the potential difference changes sign: a central difference straddling a kink is
    # O(1) off for that row, so require agreement on all but a vanishing fraction of rows
    bad = np.abs(fd - Jd) > 1e-5 * np.abs(Jd).max()
    assert bad.mean() < 2e-4
    assert np.linalg.norm((fd - Jd)[~bad]) <= 1e-6 * np.linalg.norm(Jd)
    # block-ILU(0) BiCGStab on the 2x2 system, residual verified with the block SpMV
    law.set_state(X0)
    src = np.zeros((2, 2))
    src[0], src[1] = [0.01, 0.01], [-0.01, -0.01]
    law.set_sources([1, nc], src.reshape(-1))
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                          relative_tolerance=1e-8, max_iterations=300)
    sim = ja.Simulator(law, ks, tolerance=1e-7)
    rep = sim.perform_step(dt, 1)
    assert rep.linear_status == 0
    Jdx = ja.mul_(ja.DeviceVector(disc), sim.lsys.jac, sim.lsys.dx)
    res = Jdx.axpby(1.0, sim.lsys.r, 1.0)
    assert res.dot(res) ** 0.5 <= 2e-8 * sim.lsys.r.dot(sim.lsys.r) ** 0.5
    ok, its, rep = sim.solve_ministep(dt)
    assert ok and its <= 8


def test_1M_cells_eight_ranks_match_single_rank(ja):
    """BASELINE configs[2]: the 1M-cell single-phase case domain-decomposed over 8 ranks (here 8 in-process ranks on one GPU,
    the DebugPArrayBackend analogue) with ghost halo exchange and block-Jacobi ILU(0) gives the single-rank Newton update."""
    import threading
    from bench import dims_for_cells
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(*dims_for_cells(1_000_000))
    nc = g["nc"]
    assert nc > 990_000
    T = g["T"] / g["T"].mean()
    X0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
    dt, nranks = 5.0, 8
    src = ([1, nc], np.array([[1.0], [-1.0]]))

    def make_sim(law):
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-10,
                              max_iterations=400, precond_side="right")
        return ja.Simulator(law, ks, tolerance=1e-8)

    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, reorder="blocks")
    law0 = ja.ConservationLaw(disc0, "poisson")
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    ok0, its0, _ = make_sim(law0).solve_ministep(dt)
    assert ok0
    X_ref = law0.get_state()
    part = dd.partition_rcb(g["cell_centroids"], nranks)
    group = ja.LocalCommGroup(nranks)
    out, err = [None] * nranks, []

    def rank_fn(r):
        try:
            ctx = ja.HIPContext(0)
            ctx.comm_init_local(group, r)
            disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, sources=src, block_rows=0,
                                                   ghost_order="owner")
            assert disc.split()[0] > 0
            ok, its, rep = make_sim(law).solve_ministep(dt)
            out[r] = (ok, its, sub, law.get_state())
            ctx.comm_finalize()
        except Exception as e:  # a failing rank would leave the others waiting in a collective
            err.append(e)
            raise

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not err, err
    X = np.zeros(nc)
    for ok, its, sub, Xl in out:
        assert ok and its == its0
        X[sub["cells"][: sub["n_owned"]] - 1] = Xl[: sub["n_owned"]]
        assert np.allclose(Xl[sub["n_owned"]:], X_ref[sub["cells"][sub["n_owned"]:] - 1], rtol=1e-7, atol=1e-9)
    assert np.abs(X - X_ref).max() <= 1e-7 * np.abs(X_ref).max()


def _newton_update_over_ranks_matches_single_rank(ja, cells, kind, nranks, dt):
    """One Newton update (perform_step!: ghost exchange of the primary variables, assembly, ghost rows -> -I, block-Jacobi ILU(0),
    distributed BiCGStab at rtol 1e-10, update, simulator.jl:392-455 through ext/JutulPartitionedArraysExt) of the case decomposed
    over `nranks` in-process ranks == the single-rank update to 1e-7 of the state's scale, owned cells and ghost copies."""
    import threading
    from bench import dims_for_cells
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(*dims_for_cells(cells))
    nc = g["nc"]
    assert nc > 0.99 * cells
    T = g["T"] / g["T"].mean()
    rng = np.random.default_rng(3)
    bs = 2 if kind == "twophase" else 1
    if bs == 1:
        X0 = 1.0 + 0.1 * rng.random(nc)
        src = ([1, nc], np.array([[1.0], [-1.0]]))
        par = {}
    else:
        X0 = np.stack([1.0 + 0.05 * rng.random(nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)
        src = ([1, nc], np.array([[1.0, 0.2], [-1.0, -0.2]]))
        par = dict(rho0=(1.0, 0.8), compressibility=(1e-3, 2e-3), viscosity=(1.0, 2.0), p_ref=1.0)

    def make_sim(law):
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-10,
                              max_iterations=600, precond_side="right")
        return ja.Simulator(law, ks, tolerance=1e-8)

    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, block_n=bs, reorder="blocks")
    law0 = ja.ConservationLaw(disc0, kind, **par)
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    rep0 = make_sim(law0).perform_step(dt, 1)
    assert rep0.linear_status == 0 and rep0.linear_iterations > 3
    X_ref = law0.get_state().reshape(nc, bs)
    del law0, disc0
    part = dd.partition_rcb(g["cell_centroids"], nranks)
    group = ja.LocalCommGroup(nranks)
    out, err = [None] * nranks, []

    def rank_fn(r):
        try:
            ctx = ja.HIPContext(0)
            ctx.comm_init_local(group, r)
            disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, r, T, g["volumes"], X0, kind=kind, block_n=bs, sources=src,
                                                   block_rows=0, law_params=par, ghost_order="owner")
            assert disc.split()[0] > 0
            rep = make_sim(law).perform_step(dt, 1)
            law.synchronize_ghosts()   # consistent!(state): the ghost copies of the updated primary variables
            out[r] = (rep.linear_status, sub["cells"].copy(), sub["n_owned"], law.get_state().reshape(-1, bs))
            ctx.comm_finalize()
        except Exception as e:  # a failing rank would leave the others waiting in a collective
            err.append(e)
            raise

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(900) for t in th]
    assert not err, err
    X = np.zeros((nc, bs))
    scale = np.abs(X_ref).max(axis=0)
    for status, cells_r, n_owned, Xl in out:
        assert status == 0
        X[cells_r[:n_owned] - 1] = Xl[:n_owned]
        assert np.all(np.abs(Xl[n_owned:] - X_ref[cells_r[n_owned:] - 1]) <= 1e-7 * scale)
    assert np.all(np.abs(X - X_ref) <= 1e-7 * scale)
    assert np.abs(X_ref - X0.reshape(nc, bs)).max() > 1e-4       # the update is not a no-op


def test_5M_cells_two_phase_eight_ranks_match_single_rank(ja):
    """BASELINE configs[3] at full size: the 5M-cell two-phase case (2x2 block-CSR assembly, block-ILU(0)) decomposed over 8 ranks."""
    _newton_update_over_ranks_matches_single_rank(ja, 5_000_000, "twophase", 8, 0.5)


def test_10M_cells_eight_ranks_match_single_rank(ja):
    """BASELINE configs[4] at full size: the 10M-cell grid decomposed over 8 ranks (1.25M owned cells each)."""
    _newton_update_over_ranks_matches_single_rank(ja, 10_000_000, "poisson", 8, 5.0)


def test_1M_cells_nonlinear_timesteps_match_the_oracle_newton_sequence(ja, oracle):
    """bench.py's nonlinear line (--law compressible --compressibility 0.5 --timesteps) at 1M cells: three implicit time steps
    through Simulator.solve_timestep (assemble -> converged? -> ILU(0) refactor + BiCGStab -> update -> ..., simulator.jl:304-617)
    against the oracle's Newton loop on the same grid with the same control flow (min_nonlinear_iterations = 1, tolerance on
    max|r|): the SAME number of Newton iterations in every time step and the same states to 1e-7.  Linear solves at rtol 1e-8 on
    both sides, so that the iteration counts are decided by the nonlinear convergence and not by the last digit of an inexact
    solve (the device eliminates in its own block order: its BiCGStab iterates differ from the oracle's, Krylov.jl is unpinned)."""
    from bench import dims_for_cells
    from jutul_amd import dd
    ctx = ja.HIPContext(0)
    g = ja.tet_lattice_mesh(*dims_for_cells(1_000_000))
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    vol = g["volumes"]
    par = dict(rho0=(1.0, 1.0), compressibility=(0.5, 0.5), viscosity=(1.0, 1.0), p_ref=1.0)
    X0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
    src_c, src_v = [1, nc], [1.0, -1.0]
    dt, tol, nsteps = 5.0, 1e-7, 3
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks")
    law = ja.ConservationLaw(disc, "compressible", **par)
    law.set_face_trans(T); law.set_volumes(vol); law.set_state(X0); law.set_state0(X0); law.set_sources(src_c, src_v)
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-8,
                          max_iterations=300)
    sim = ja.Simulator(law, ks, tolerance=tol)
    gpu_its, gpu_states = [], []
    for _ in range(nsteps):
        gpu_its.append(sim.solve_timestep(dt))
        assert len(sim.last_ministeps) == 1 and sim.last_ministeps[0]["success"]      # no time-step cut on this problem
        gpu_states.append(law.get_state())
    # the oracle: same loop, its own BiCGStab with a block-Jacobi ILU(0) over 8 coordinate blocks
    osys = oracle.TPFASystem(g["N"], nc)
    olaw = oracle.Law("compressible", dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    part = dd.partition_rcb(g["cell_centroids"], 8)
    X = X0.copy()
    Fo = None
    for step in range(nsteps):
        Xp = X.copy()
        for it in range(1, 17):
            nz, r = osys.assemble(olaw, X, Xp, vol, T, src_cells=src_c, src_values=src_v)
            if it > 1 and np.abs(r).max() < tol:
                break
            if Fo is None:
                Fo = oracle.ILU0(nc, 1, osys.rowptr, osys.colidx, nz, partition=part)
            else:
                Fo.refactor(nz)
            x, st = oracle.bicgstab(nc, 1, osys.rowptr, osys.colidx, nz, r, prec=Fo, side="right", rtol=1e-8, atol=1e-30, itmax=300)
            assert st["solved"]
            X = X - x
        assert it == gpu_its[step], (step, it, gpu_its)
        assert np.abs(gpu_states[step] - X).max() <= 1e-7 * np.abs(X).max()
    assert max(gpu_its) >= 3        # genuinely nonlinear: more than one Newton update per time step


def test_1M_cells_two_phase_newton_sequence_matches_the_oracle(ja, oracle):
    """configs[3]'s law (immiscible two-phase, 2x2 blocks, SPU upwinding) at 1M cells as a Newton SEQUENCE against the oracle (the
    scalar twin is the test above): two implicit time steps through Simulator.solve_timestep with Jutul's saturation update
    limits (absolute increment 0.2, bounds [0, 1]: choose_increment, variables/utils.jl:110-174) against the oracle's loop --
    assemble -> converged? -> block-ILU(0) refactor + BiCGStab -> limited update -- with the same control flow: equal Newton
    iteration counts per time step, states to 1e-6 of each variable's scale.  Linear solves at rtol 1e-8 on both sides."""
    from bench import LAW_PAR, apply_update, dims_for_cells, initial_state, source_values, update_limits
    from jutul_amd import dd
    ctx = ja.HIPContext(0)
    g = ja.tet_lattice_mesh(*dims_for_cells(1_000_000))
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    vol = g["volumes"]
    par = LAW_PAR["twophase"]
    X0 = initial_state(np, "twophase", nc)
    src_c, src_v = [1, nc], np.asarray(source_values(np, "twophase", [1.0, -1.0]))
    lim = update_limits(np, "twophase")
    dt, tol, nsteps = 0.5, 1e-7, 2   # (the oracle's 2x2 solves at rtol 1e-8 are the slow part: ~70 s per time step)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=2, reorder="blocks")
    law = ja.ConservationLaw(disc, "twophase", **par)
    law.set_face_trans(T); law.set_volumes(vol); law.set_state(X0); law.set_state0(X0); law.set_sources(src_c, src_v)
    law.set_update_limits(lim)
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-8,
                          max_iterations=400)
    sim = ja.Simulator(law, ks, tolerance=tol)
    gpu_its, gpu_states = [], []
    for _ in range(nsteps):
        gpu_its.append(sim.solve_timestep(dt))
        assert len(sim.last_ministeps) == 1 and sim.last_ministeps[0]["success"]
        gpu_states.append(law.get_state())
    osys = oracle.TPFASystem(g["N"], nc, nblk=2)
    olaw = oracle.Law("twophase", dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    part = dd.partition_rcb(g["cell_centroids"], 8)
    X = X0.copy()
    Fo = None
    for step in range(nsteps):
        Xp = X.copy()
        for it in range(1, 17):
            nz, r = osys.assemble(olaw, X, Xp, vol, T, src_cells=src_c, src_values=src_v)
            if it > 1 and np.abs(r).max() < tol:
                break
            if Fo is None:
                Fo = oracle.ILU0(nc, 2, osys.rowptr, osys.colidx, nz, partition=part)
            else:
                Fo.refactor(nz)
            x, st = oracle.bicgstab(nc, 2, osys.rowptr, osys.colidx, nz, r, prec=Fo, side="right", rtol=1e-8, atol=1e-30, itmax=400)
            assert st["solved"]
            X = apply_update(np, X, -x, lim)
        assert it == gpu_its[step], (step, it, gpu_its)
        G, O = gpu_states[step].reshape(-1, 2), X.reshape(-1, 2)
        for e in range(2):
            assert np.abs(G[:, e] - O[:, e]).max() <= 1e-6 * np.abs(O[:, e]).max(), (step, e)
    assert max(gpu_its) >= 2


def test_3M_cells_oracle_parity_where_the_defaults_switch(ja, oracle):
    """Oracle parity at a size where the library's defaults change code path: >= 3M rows -> 16-bit column codes in the jagged
    SpMV (no environment override), >= 2M cells -> 512-row bisection blocks, ~100 tiles per XCD chunk.  Assembly, jh_spmv,
    jh_spmv_jagged and one block-Jacobi ILU(0) factor + apply on the device blocks, each against the C oracle with a PER-ENTRY
    bound |a - b| <= c * eps * sum |terms| (not scaled by the largest entry of the array): with a 400x transmissibility
    contrast the smallest entries are pinned as tightly as the largest."""
    import os
    import scipy.sparse as sp
    from bench import dims_for_cells
    assert "JH_OPTIONS" not in os.environ
    eps = np.finfo(np.float64).eps
    ctx = ja.HIPContext(0)
    assert ctx.get_option("spmv_col_bits") == 0 and ctx.get_option("spmv_jagged") == 1
    g = ja.tet_lattice_mesh(*dims_for_cells(3_200_000))
    nc, nf, N = g["nc"], g["nf"], g["N"]
    assert nc >= 3_000_000
    rng = np.random.default_rng(11)
    T = g["T"] / g["T"].mean() * np.exp(rng.uniform(-3.0, 3.0, nf))      # 400x contrast between neighbouring faces
    vol = g["volumes"]
    U, U0 = 1.0 + rng.random(nc), 1.0 + rng.random(nc)
    dt = 0.8
    src_c, src_v = np.array([7, nc - 5]), np.array([0.75, -0.25])
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks")     # library defaults: 512-row bisection blocks
    perm, bp = disc.ordering()
    assert 400 < nc / (len(bp) - 1) <= 576 and np.diff(bp).max() <= 576
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(T)
    law.set_volumes(vol)
    law.set_state(U)
    law.set_state0(U0)
    law.set_sources(src_c, src_v)
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    info = lsys.jac.spmv_info()
    assert info["jagged"] and info["col16"] and info["longest_row"] == 5, info
    # ---- assembly: r_i = sum_f T_f (u_i - u_j) + vol_i (u_i - u0_i) / dt - src_i ; J_ii = sum_f T_f + vol_i / dt ; J_ij = -T_f
    osys = oracle.TPFASystem(N, nc)
    olaw = oracle.Law("poisson", dt)
    nz_o, r_o = osys.assemble(olaw, U, U0, vol, T, src_cells=src_c, src_values=src_v)
    rp, ci = osys.rowptr - 1, osys.colidx - 1
    A = sp.csr_matrix((nz_o, ci, rp), shape=(nc, nc))
    Aabs = abs(A)
    src_abs = np.zeros(nc)
    src_abs[src_c - 1] = np.abs(src_v)
    Toff = Aabs - sp.diags(Aabs.diagonal())                         # |T_f| on the off-diagonal
    term_r = Toff @ np.abs(U) + np.asarray(Toff.sum(axis=1)).ravel() * np.abs(U) + vol * (np.abs(U) + np.abs(U0)) / dt + src_abs
    r = lsys.r.download()
    assert np.all(np.abs(r - r_o) <= 8 * eps * term_r), float((np.abs(r - r_o) / term_r).max() / eps)
    nz = lsys.jac.nzval
    assert np.all(np.abs(nz - nz_o) <= 8 * eps * np.abs(nz_o)), float((np.abs(nz - nz_o) / np.abs(nz_o)).max() / eps)
    assert np.abs(nz_o).min() < 1e-3 * np.abs(nz_o).max()           # the contrast is really there
    # ---- SpMV: CSR tile kernel and the jagged kernel with its default 16-bit codes; bound (row length + 1) * eps * (|A| |x|)_i
    x = rng.standard_normal(nc)
    y_o = oracle.spmv(nc, 1, osys.rowptr, osys.colidx, nz_o, x)
    bound = 6 * eps * (Aabs @ np.abs(x))
    lsys.jac.nzval = nz_o                                           # the same matrix bits on both sides
    dxv = ja.DeviceVector(disc, x)
    y_csr = ja.mul_(ja.DeviceVector(disc), lsys.jac, dxv).download()
    y_jag = ja.mul_(ja.DeviceVector(disc), lsys.jac, dxv, jagged=True).download()
    assert np.all(np.abs(y_csr - y_o) <= bound) and np.all(np.abs(y_jag - y_o) <= bound)
    assert np.array_equal(y_csr, y_jag)                             # same accumulation order -> same bits
    # ---- block-Jacobi ILU(0) over the 512-row device blocks: componentwise backward error against the ORACLE's factors
    F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    fi = F.info()
    assert fi["nblocks"] == len(bp) - 1 and fi["jagged"]
    p0 = perm - 1
    part = np.zeros(nc, dtype=np.int64)
    part[p0] = np.repeat(np.arange(1, len(bp)), np.diff(bp))
    Ap = A[p0][:, p0].tocsr()                                       # the device's elimination order (see the small-size test)
    Ap.sort_indices()
    Fo = oracle.ILU0(nc, 1, Ap.indptr + 1, Ap.indices + 1, Ap.data, partition=part[p0])
    lu = Fo.export(Ap.data.size)                                    # L multipliers | inv(U_ii) | U, zero outside the blocks
    rows = np.repeat(np.arange(nc), np.diff(Ap.indptr))
    lower, upper, dia = Ap.indices < rows, Ap.indices > rows, Ap.indices == rows
    Lm = sp.csr_matrix((lu[lower], (rows[lower], Ap.indices[lower])), shape=(nc, nc)) + sp.identity(nc, format="csr")
    Um = sp.csr_matrix((lu[upper], (rows[upper], Ap.indices[upper])), shape=(nc, nc)) + sp.diags(1.0 / lu[dia])
    b = rng.standard_normal(nc)
    xh = F.apply(lsys.jac.new_vector(), lsys.jac.new_vector(b)).download()[p0]
    bp_ = b[p0]
    resid = np.abs(Lm @ (Um @ xh) - bp_)
    scale = abs(Lm) @ (abs(Um) @ np.abs(xh)) + np.abs(bp_)
    nlev = fi["max_levels"]
    assert np.all(resid <= 4 * nlev * eps * scale), float((resid / scale).max() / eps)
    x_o = Fo.apply(bp_)
    assert np.abs(xh - x_o).max() <= 1e-10 * np.abs(x_o).max()
    kept = int(lower.sum() - (lu[lower] == 0).sum()), int(upper.sum() - (lu[upper] == 0).sum())
    assert (fi["l_entries"], fi["u_entries"]) == kept, (fi, kept)


@pytest.mark.parametrize("cells", [3_200_000, 400_000])
def test_consumer_side_reduction_stage_is_deterministic_and_matches_the_separate_launch(ja, cells):
    """Default (option consumer_reduce = 1): the second stage of the fused dot products of BiCGStab runs inside the kernel that
    needs the scalar next -- every wavefront of the consumer sums the producer's partials itself, in one fixed order (PendSum,
    csrc/jh_internal.hpp) -- instead of a one-workgroup launch.  Ordered sums, no atomics: four long solves of the same system
    must give bit-identical residual histories and solutions, and they must follow the history of the separate-launch path
    (consumer_reduce = 0; a different but fixed summation tree) to rounding.  3.2M cells: 16-bit column codes, 512-row blocks,
    a chip-filling launch; 0.4M cells: 32-bit columns, 256-row blocks, fewer workgroups than the chip holds."""
    from bench import dims_for_cells
    ctx = ja.HIPContext(0)
    assert ctx.get_option("consumer_reduce") == 1
    g = ja.tet_lattice_mesh(*dims_for_cells(cells))
    nc = g["nc"]
    rng = np.random.default_rng(5)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks")
    law = ja.ConservationLaw(disc, "compressible", rho0=(1.0, 1.0), compressibility=(1e-1, 1e-1), viscosity=(1.0, 1.0), p_ref=1.0)
    law.set_face_trans(g["T"] / g["T"].mean())
    law.set_volumes(g["volumes"])
    U = 1.0 + 0.1 * rng.random(nc)
    law.set_state(U)
    law.set_state0(U)
    law.set_sources([1, nc], [1.0, -1.0])
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(5.0, lsys.jac, lsys.r)
    prec = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-9, max_iterations=400)

    def solve():
        out = ja.linear_solve(lsys, ks, update_preconditioner=False)
        return out["iterations"], out["residuals"].copy(), lsys.dx.download()
    runs = [solve() for _ in range(4)]
    assert runs[0][0] > 8
    for it, hist, dx in runs[1:]:
        assert it == runs[0][0] and np.array_equal(hist, runs[0][1]) and np.array_equal(dx, runs[0][2])
    for waves in (4, 16):   # other workgroup shapes of the product (other partial counts): same solve to rounding
        ctx.set_option("spmv_waves", waves)
        it, hist, dx = solve()
        assert abs(it - runs[0][0]) <= max(3, 0.15 * it)
        assert np.allclose(hist[:12], runs[0][1][:12], rtol=1e-9)
    ctx.set_option("spmv_waves", 0)
    ctx.set_option("consumer_reduce", 0)
    it0, hist0, dx0 = solve()
    ctx.set_option("consumer_reduce", 1)
    k = 12
    assert np.allclose(runs[0][1][:k], hist0[:k], rtol=1e-9), (runs[0][1][:k], hist0[:k])
    assert abs(it0 - runs[0][0]) <= max(3, 0.15 * it0)
    assert np.abs(dx0 - runs[0][2]).max() <= 1e-6 * np.abs(dx0).max()
    # an iteration limit that ends the solve inside the loop: the last iteration's sums are formed by the one-wavefront kernel,
    # in the same order as a consuming kernel forms them -> the same residual history up to there
    ks2 = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-30, max_iterations=7)
    out = ja.linear_solve(lsys, ks2, update_preconditioner=False)
    assert out["iterations"] == 7 and not out["ok"] and np.array_equal(out["residuals"][:8], runs[0][1][:8])
