"""Parity on genuinely unstructured grid families (the Kuhn lattice of the bench is bipartite, triangle-free and has <= 4 faces
per cell: the best case of every layout heuristic):
  * Delaunay tets of graded random points (what the reference would read through ext/JutulGmshExt/interface.jl): odd cycles in
    the dual graph -> the ILU(0) elimination updates off-diagonal entries (program-driven factor kernel), varying cell sizes;
  * the polyhedral median dual of such a mesh: ~15 faces per cell, rows of up to ~50 entries -> beyond the jagged layouts
    (CSR tile SpMV, row-major ILU kernels).
Everything against the oracle through the C ABI, for the scalar compressible law and the 2x2-block two-phase law: tables
bit-exact, assembly / SpMV 1e-12 per row scale, block-Jacobi ILU(0) FACTOR VALUES entry by entry against the size of each
entry's own terms and the triangular solve as a componentwise backward error (tests/_ilu_checks.py), the Newton update through the
true residual with the oracle's Jacobian."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


def run_family(ja, oracle, g, kind, expect):
    import scipy.sparse as sp
    from tests import _ilu_checks as ck
    ctx = ja.HIPContext(0)
    nc, nf, N = g["nc"], g["nf"], g["N"]
    bs = 2 if kind == "twophase" else 1
    bb = bs * bs
    rng = np.random.default_rng(4)
    T = g["T"] / g["T"].mean()
    vol = g["volumes"] / g["volumes"].mean()
    gdz = ja.compute_face_gdz(N, g["cell_centroids"][2]) * 1e-3
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, block_n=bs, reorder="blocks")
    # ---- a-1..a-4: bit exact ---------------------------------------------------------------------------------------------------
    h = oracle.half_face_map(N, nc)
    c = disc.conn
    assert np.array_equal(c["face_pos"], h["face_pos"]) and np.array_equal(c["other"], h["other"])
    assert np.array_equal(c["face"], h["faces"]) and np.array_equal(c["face_sign"], h["face_sign"])
    rowptr, colidx = disc.pattern()
    orp, oci = oracle.csr_pattern(nc, h)
    assert np.array_equal(rowptr, orp) and np.array_equal(colidx, oci)
    assert np.diff(rowptr).max() == expect["longest_row"]
    # ---- a-5..a-9: assembly ----------------------------------------------------------------------------------------------------
    par = dict(rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
    law = ja.ConservationLaw(disc, kind, **par)
    if bs == 1:
        X, X0 = rng.uniform(1.0, 2.0, nc), rng.uniform(1.0, 2.0, nc)
    else:   # (pressure, saturation) per cell
        X = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
        X0 = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
    law.set_face_trans(T)
    law.set_volumes(vol)
    law.set_face_gdz(gdz)
    law.set_state(X)
    law.set_state0(X0)
    src_c = [5, nc - 3]
    src_v = np.array([0.3, -0.3]) if bs == 1 else np.array([0.3, 0.1, -0.3, -0.1])
    law.set_sources(src_c, src_v)
    lsys = ja.LinearizedSystem(disc)
    dt = 0.5
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    osys = oracle.TPFASystem(N, nc, bs)
    olaw = oracle.Law(kind, dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    nz_o, r_o = osys.assemble(olaw, X, X0, vol, T, gdz, src_c, src_v)
    blocks = nz_o.reshape(-1, bs, bs).transpose(0, 2, 1)             # column-major blocks -> [row, col]
    A = sp.bsr_matrix((blocks, oci - 1, orp - 1), shape=(nc * bs, nc * bs)).tocsr()
    # |J| (|x| <= 2) + |r|: the size of a row's terms, per CELL (the largest of its equations)
    row_scale = (np.asarray(abs(A).sum(axis=1)).ravel() * 2.5 + np.abs(r_o)).reshape(nc, bs).max(axis=1)
    assert np.all(np.abs(lsys.r.download() - r_o) <= 1e-12 * np.repeat(row_scale, bs))
    nz = lsys.jac.nzval
    assert np.all(np.abs(nz - nz_o) <= 1e-12 * np.repeat(row_scale, np.diff(orp) * bb))
    # ---- a-10: SpMV ------------------------------------------------------------------------------------------------------------
    info = lsys.jac.spmv_info()
    assert info["jagged"] == expect["jagged_spmv"] and info["longest_row"] == expect["longest_row"], info
    x = rng.standard_normal(nc * bs)
    lsys.jac.nzval = nz_o
    y_o = oracle.spmv(nc, bs, orp, oci, nz_o, x)
    bound = 1e-13 * (abs(A) @ np.abs(x)) * max(8, expect["longest_row"]) * bs
    xv = ja.DeviceVector(disc, x)
    y = ja.mul_(ja.DeviceVector(disc), lsys.jac, xv).download()
    assert np.all(np.abs(y - y_o) <= bound)
    if info["jagged"]:
        assert np.array_equal(ja.mul_(ja.DeviceVector(disc), lsys.jac, xv, jagged=True).download(), y)
    # ---- a-11..a-13: block-Jacobi ILU(0) on the device blocks vs the oracle in the device's elimination order: the FACTOR VALUES
    # entry by entry against the size of each entry's own terms, the triangular solve as a componentwise backward error with the
    # oracle's factors (cell volumes over 8 decades: nothing here is scaled by the largest entry of an array) -----------------
    F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    fi = F.info()
    assert fi["factor_kernel"] == expect["factor_kernel"] and fi["jagged"] == expect["jagged_ilu"], fi
    perm, bp = disc.ordering()
    p0 = perm - 1
    rp_p, ci_p, nz_p, part_p, slot = ck.device_order_problem(nc, bs, orp, oci, nz_o, perm, bp)
    Fo, lu_o, Lm, Um = ck.oracle_factors(oracle, nc, bs, rp_p, ci_p, nz_p, part_p)
    lu_dev = F.factor_values().reshape(-1, bb)[slot].reshape(-1)
    w_f = ck.check_factor_values(lu_dev, lu_o, Lm, Um, nc, bs, rp_p, ci_p, c=64.0 * max(1, fi["max_levels"] // 16))
    b = rng.standard_normal(nc * bs)
    pe = (p0[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
    xh = F.apply(lsys.jac.new_vector(), lsys.jac.new_vector(b)).download()[pe]
    w_s = ck.check_triangular_solve(xh, b[pe], Lm, Um, levels=fi["max_levels"])
    # ---- a-14: BiCGStab + primary update through the Newton step -----------------------------------------------------------------
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-10,
                          max_iterations=600)
    law.set_state(X)
    sim = ja.Simulator(law, ks)
    rep = sim.perform_step(dt, 1)
    assert rep.linear_status == 0 and rep.linear_iterations > 3
    # the Newton update solves J dx = -r: true residual with the ORACLE's Jacobian (conditioning-independent; the coefficient
    # contrast of these grids is ~1e8, so a bound on the error itself would be a statement about the grid)
    dxv = law.get_state() - X
    assert np.linalg.norm(A @ dxv + r_o) <= 5e-10 * np.linalg.norm(r_o)
    return dict(its=int(rep.linear_iterations), blocks=fi["nblocks"], levels=fi["max_levels"], factor_eps=w_f, solve_eps=w_s,
                kept=(fi["l_entries"] + fi["u_entries"]) / (oci.size - nc))


@pytest.mark.parametrize("kind", ["compressible", "twophase"])
def test_delaunay_tet_mesh_parity(ja, oracle, kind):
    g = ja.delaunay_tet_mesh(32000, grading=2.0)                     # ~210k tets, cell volumes over 8 decades
    assert g["nc"] > 200_000
    out = run_family(ja, oracle, g, kind,
                     dict(longest_row=5, jagged_spmv=kind != "twophase", jagged_ilu=True, factor_kernel="program"))
    assert out["kept"] > 0.6


@pytest.mark.parametrize("kind", ["compressible", "twophase"])
def test_polyhedral_dual_mesh_parity(ja, oracle, kind):
    g = ja.polyhedral_dual_mesh(30000, grading=1.5)                  # 30k cells, ~15 faces each
    deg = np.bincount(g["N"].reshape(-1), minlength=g["nc"] + 1)[1:]
    assert deg.max() > 16 and deg.mean() > 12
    run_family(ja, oracle, g, kind,
               dict(longest_row=int(deg.max()) + 1, jagged_spmv=False, jagged_ilu=True,   # ILU: chains of lanes
                    factor_kernel="program, rows form" if kind == "compressible" else "program"))


@pytest.mark.parametrize("kind", ["compressible", "twophase"])
def test_cartesian_mesh_parity(ja, oracle, kind):
    g = ja.cartesian_mesh(60, 60, 60)                                # 216k hexahedra, the reference's CartesianMesh: 6 faces per cell
    run_family(ja, oracle, g, kind,
               dict(longest_row=7, jagged_spmv=kind != "twophase", jagged_ilu=True, factor_kernel="pivot-only"))


@pytest.mark.parametrize("kind", ["compressible", "twophase"])
def test_long_row_factor_program_forms_agree_bitwise(ja, kind):
    """ilu_factor_wave_per_row: thread per row (0), wavefront per row over the rows-form programs (1, default; scalar matrices)
    and over the instruction-form programs (2) do the same operations on every entry in the same order -> identical factor bits"""
    g = ja.polyhedral_dual_mesh(12000, grading=1.5)
    bs = 2 if kind == "twophase" else 1
    rng = np.random.default_rng(9)
    got = {}
    for mode in (0, 1, 2):
        ctx = ja.HIPContext(0, ilu_factor_wave_per_row=mode)
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], block_n=bs, reorder="blocks")
        lsys = ja.LinearizedSystem(disc)
        rowptr, colidx = disc.pattern()
        if "nz" not in got:   # a diagonally dominant random matrix on the pattern
            rows = np.repeat(np.arange(g["nc"]), np.diff(rowptr))
            blk = rng.uniform(-1.0, 1.0, (colidx.size, bs, bs))
            blk[colidx - 1 == rows] += 40.0 * np.eye(bs)
            got["nz"] = blk.reshape(-1)
        lsys.jac.nzval = got["nz"]
        F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
        assert F.info()["factor_kernel"] == ("program, rows form" if mode == 1 and bs == 1 else "program"), F.info()
        got[mode] = F.factor_values().copy()
    assert np.isfinite(got[1]).all() and np.abs(got[1]).max() > 0
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[2], got[1])


@pytest.mark.parametrize("reach,block_rows,form", [(110, 256, "program, rows form"), (260, 512, "program")])
def test_factor_rows_form_with_rows_of_two_registers(ja, reach, block_rows, form):
    """rows of 65..128 entries in one block (two registers per lane in ilu_factor_rows_kernel): hub cells joined to their ~100
    graph-nearest cells on top of a polyhedral mesh; bitwise against the thread-per-row instruction form.  Second case: more than
    128 in-block entries -- the rows form does not hold such a row, the whole matrix keeps the instruction form."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cg
    g = ja.polyhedral_dual_mesh(6000, grading=1.2)
    nc, N = g["nc"], np.asarray(g["N"])
    A = sp.coo_matrix((np.ones(N.shape[1]), (N[0] - 1, N[1] - 1)), shape=(nc, nc))
    A = (A + A.T).tocsr()
    extra = []
    for hub in (nc // 7, nc // 2, (4 * nc) // 5):
        order = cg.breadth_first_order(A, hub, directed=False, return_predecessors=False)[1:reach]
        have = set(A.indices[A.indptr[hub]:A.indptr[hub + 1]])
        extra += [(hub + 1, int(c) + 1) for c in order if int(c) not in have]
    N2 = np.concatenate([N, np.array(extra, dtype=N.dtype).T], axis=1)
    rng = np.random.default_rng(3)
    got = {}
    for mode in (0, 1):
        ctx = ja.HIPContext(0, ilu_factor_wave_per_row=mode)
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, N2, nc, reorder="blocks", block_rows=block_rows)
        lsys = ja.LinearizedSystem(disc)
        rowptr, colidx = disc.pattern()
        assert np.diff(rowptr).max() >= reach - 10
        if "nz" not in got:
            rows = np.repeat(np.arange(nc), np.diff(rowptr))
            nz = rng.uniform(-1.0, 1.0, colidx.size)
            nz[colidx - 1 == rows] += 1.5 * reach + 50.0
            got["nz"] = nz
        lsys.jac.nzval = got["nz"]
        # the hub rows keep most of their entries inside their block: more than 64 take part in the elimination
        perm, bp = disc.ordering()
        blk_of = np.repeat(np.arange(len(bp) - 1), np.diff(bp))
        dev_of = np.empty(nc, dtype=np.int64); dev_of[perm - 1] = np.arange(nc)
        inblock = [int(np.sum(blk_of[dev_of[colidx[rowptr[h] - 1:rowptr[h + 1] - 1] - 1]] == blk_of[dev_of[h]])) for h in (nc // 7, nc // 2, (4 * nc) // 5)]
        got["inblock"] = inblock
        F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
        fi = F.info()
        assert fi["factor_kernel"] == (form if mode == 1 else "program"), (fi, got.get("inblock"))
        got[mode] = F.factor_values().copy()
    assert max(got["inblock"]) > (64 if reach < 129 else 128), got["inblock"]
    assert np.isfinite(got[1]).all() and np.array_equal(got[0], got[1])


def test_bicgstab_history_on_a_graded_grid_is_the_oracle_history(ja):
    """The solve of the bench (Poisson law, dt = 5, rtol 1e-3, right-preconditioned BiCGStab, block-Jacobi ILU(0) on the device's
    weighted blocks) on a graded Delaunay grid -- transmissibilities over 8 decades, volumes over 8 -- against the oracle's
    BiCGStab with the oracle's ILU(0) in the device's elimination order and block partition: the first residuals of the history to
    1e-9 of their value, the first 25 to 1e-6 (the slowly converging solve amplifies the rounding of the two dot-product orders),
    equal iteration counts at rtol 1e-3.  (Krylov.jl is not in the reference tree: the oracle restates the published algorithm;
    this pins that the device runs the same recurrence on the same preconditioner, not only that both converge.)"""
    from oracle import oracle as o
    from tests import _ilu_checks as ck
    g = ja.delaunay_tet_mesh(45000, grading=2.0, scramble=True)
    nc, N = g["nc"], g["N"]
    T = g["T"] / g["T"].mean()
    assert T.max() / T.min() > 1e6
    ctx = ja.HIPContext(0)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", face_weights=T)
    law = ja.ConservationLaw(disc, "poisson")
    U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
    law.set_face_trans(T); law.set_volumes(g["volumes"]); law.set_state(U0); law.set_state0(U0); law.set_sources([1, nc], [1.0, -1.0])
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(5.0, lsys.jac, lsys.r)
    nz, r = lsys.jac.nzval, lsys.r.download()
    for rtol, itmax in ((1e-3, 100), (1e-8, 300)):
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=rtol,
                              absolute_tolerance=0.0, max_iterations=itmax)
        out = ja.linear_solve(lsys, ks)
        osys = o.TPFASystem(N, nc)
        perm, bp = disc.ordering()
        rp_p, ci_p, nz_p, part_p, _ = ck.device_order_problem(nc, 1, osys.rowptr, osys.colidx, nz, perm, bp)
        F = o.ILU0(nc, 1, rp_p, ci_p, nz_p, partition=part_p)
        xo, st = o.bicgstab(nc, 1, rp_p, ci_p, nz_p, r[perm - 1], prec=F, side="right", rtol=rtol, atol=0.0, itmax=itmax)
        # (the tight solve may run into itmax on this grid -- on both sides alike; what is compared is the recurrence)
        assert out["ok"] == bool(st["solved"]), (rtol, out["ok"], st["solved"], out["iterations"], st["iterations"])
        if rtol == 1e-3:
            assert out["ok"] and out["iterations"] == st["iterations"], (out["iterations"], st["iterations"])
        m = min(len(out["residuals"]), len(st["residuals"]), 25)   # (later, rounding differences of the two dot-product orders have grown)
        assert m >= 4
        assert np.allclose(out["residuals"][:6], st["residuals"][:6], rtol=1e-9, atol=0.0), (out["residuals"][:6], st["residuals"][:6])
        assert np.allclose(out["residuals"][:m], st["residuals"][:m], rtol=1e-6, atol=0.0), (out["residuals"][:m], st["residuals"][:m])
        if out["ok"]:
            x = np.empty(nc); x[perm - 1] = xo
            assert np.abs(-lsys.dx.download() - x).max() <= 1e-5 * np.abs(x).max()
