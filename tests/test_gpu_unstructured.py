"""Parity on genuinely unstructured grid families (the Kuhn lattice of the bench is bipartite, triangle-free and has <= 4 faces
per cell: the best case of every layout heuristic):
  * Delaunay tets of graded random points (what the reference would read through ext/JutulGmshExt/interface.jl): odd cycles in
    the dual graph -> the ILU(0) elimination updates off-diagonal entries (program-driven factor kernel), varying cell sizes;
  * the polyhedral median dual of such a mesh: ~15 faces per cell, rows of up to ~50 entries -> beyond the jagged layouts
    (CSR tile SpMV, row-major ILU kernels).
Everything against the oracle through the C ABI: tables bit-exact, assembly / SpMV 1e-12 per row scale, block-Jacobi ILU(0)
apply 1e-10, BiCGStab solution 1e-7."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


def run_family(ja, oracle, g, kind, expect):
    import scipy.sparse as sp
    ctx = ja.HIPContext(0)
    nc, nf, N = g["nc"], g["nf"], g["N"]
    rng = np.random.default_rng(4)
    T = g["T"] / g["T"].mean()
    vol = g["volumes"] / g["volumes"].mean()
    gdz = ja.compute_face_gdz(N, g["cell_centroids"][2]) * 1e-3
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks")
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
    X, X0 = rng.uniform(1.0, 2.0, nc), rng.uniform(1.0, 2.0, nc)
    law.set_face_trans(T)
    law.set_volumes(vol)
    law.set_face_gdz(gdz)
    law.set_state(X)
    law.set_state0(X0)
    src_c, src_v = [5, nc - 3], [0.3, -0.3]
    law.set_sources(src_c, src_v)
    lsys = ja.LinearizedSystem(disc)
    dt = 0.5
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    osys = oracle.TPFASystem(N, nc)
    olaw = oracle.Law(kind, dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    nz_o, r_o = osys.assemble(olaw, X, X0, vol, T, gdz, src_c, src_v)
    A = sp.csr_matrix((nz_o, oci - 1, orp - 1), shape=(nc, nc))
    row_scale = np.asarray(abs(A).sum(axis=1)).ravel() * 2.5 + np.abs(r_o)      # |J| (|x| <= 2) + |r|: the size of a row's terms
    assert np.all(np.abs(lsys.r.download() - r_o) <= 1e-12 * row_scale)
    nz = lsys.jac.nzval
    assert np.all(np.abs(nz - nz_o) <= 1e-12 * np.repeat(row_scale, np.diff(orp)))
    # ---- a-10: SpMV ------------------------------------------------------------------------------------------------------------
    info = lsys.jac.spmv_info()
    assert info["jagged"] == expect["jagged_spmv"] and info["longest_row"] == expect["longest_row"], info
    x = rng.standard_normal(nc)
    lsys.jac.nzval = nz_o
    y_o = oracle.spmv(nc, 1, orp, oci, nz_o, x)
    bound = 1e-13 * (abs(A) @ np.abs(x)) * max(8, expect["longest_row"])
    xv = ja.DeviceVector(disc, x)
    y = ja.mul_(ja.DeviceVector(disc), lsys.jac, xv).download()
    assert np.all(np.abs(y - y_o) <= bound)
    if info["jagged"]:
        assert np.array_equal(ja.mul_(ja.DeviceVector(disc), lsys.jac, xv, jagged=True).download(), y)
    # ---- a-11..a-13: block-Jacobi ILU(0) on the device blocks vs the oracle in the device's elimination order ----------------
    F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    fi = F.info()
    assert fi["factor_kernel"] == expect["factor_kernel"] and fi["jagged"] == expect["jagged_ilu"], fi
    perm, bp = disc.ordering()
    p0 = perm - 1
    part = np.zeros(nc, dtype=np.int64)
    part[p0] = np.repeat(np.arange(1, len(bp)), np.diff(bp))
    Ap = A[p0][:, p0].tocsr()
    Ap.sort_indices()
    Fo = oracle.ILU0(nc, 1, Ap.indptr + 1, Ap.indices + 1, Ap.data, partition=part[p0])
    b = rng.standard_normal(nc)
    xh = F.apply(lsys.jac.new_vector(), lsys.jac.new_vector(b)).download()[p0]
    x_o = Fo.apply(b[p0])
    assert np.abs(xh - x_o).max() <= 1e-10 * np.abs(x_o).max()
    # ---- a-14: BiCGStab + primary update through the Newton step -----------------------------------------------------------------
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-10,
                          max_iterations=400)
    law.set_state(X)
    sim = ja.Simulator(law, ks)
    rep = sim.perform_step(dt, 1)
    assert rep.linear_status == 0 and rep.linear_iterations > 3
    # the Newton update solves J dx = -r: true residual with the ORACLE's Jacobian (conditioning-independent; the coefficient
    # contrast of these grids is ~1e8, so a bound on the error itself would be a statement about the grid)
    dxv = law.get_state() - X
    assert np.linalg.norm(A @ dxv + r_o) <= 5e-10 * np.linalg.norm(r_o)
    return dict(its=int(rep.linear_iterations), blocks=fi["nblocks"], levels=fi["max_levels"],
                kept=(fi["l_entries"] + fi["u_entries"]) / (A.nnz - nc))


def test_delaunay_tet_mesh_parity(ja, oracle):
    g = ja.delaunay_tet_mesh(32000, grading=2.0)                     # ~210k tets, cell volumes over 8 decades
    assert g["nc"] > 200_000
    out = run_family(ja, oracle, g, "compressible",
                     dict(longest_row=5, jagged_spmv=True, jagged_ilu=True, factor_kernel="program"))
    assert out["kept"] > 0.6


def test_polyhedral_dual_mesh_parity(ja, oracle):
    g = ja.polyhedral_dual_mesh(30000, grading=1.5)                  # 30k cells, ~15 faces each
    deg = np.bincount(g["N"].reshape(-1), minlength=g["nc"] + 1)[1:]
    assert deg.max() > 16 and deg.mean() > 12
    run_family(ja, oracle, g, "compressible",
               dict(longest_row=int(deg.max()) + 1, jagged_spmv=False, jagged_ilu=False, factor_kernel="program"))


def test_cartesian_mesh_parity(ja, oracle):
    g = ja.cartesian_mesh(60, 60, 60)                                # 216k hexahedra, the reference's CartesianMesh: 6 faces per cell
    run_family(ja, oracle, g, "compressible",
               dict(longest_row=7, jagged_spmv=True, jagged_ilu=True, factor_kernel="pivot-only"))
