"""Pins the CPU oracle (oracle/jutul_oracle.c) against the reference's own known-answer tests, fixtures and
mathematical identities.  CPU only."""
import numpy as np
import pytest


def dense_from_csr(n, bs, rowptr, colidx, nz):
    A = np.zeros((n * bs, n * bs))
    nzb = nz.reshape(-1, bs, bs)  # each block column-major -> transpose
    for r in range(n):
        for p in range(rowptr[r] - 1, rowptr[r + 1] - 1):
            c = colidx[p] - 1
            A[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs] = nzb[p].T
    return A


# --- KAT 5/6/11: connectivity ------------------------------------------------------------------------------
def test_cartesian_face_order_matches_pico_fixture(oracle, golden):
    """cart.jl:197-225 MRST face order == neighbors of data/testgrids/pico.mat (3x3x1), test/mesh.jl:88-94."""
    geo = oracle.cartesian_geometry((3, 3, 1), (3.0, 3.0, 1.0))  # pico cells are unit cubes
    assert np.array_equal(geo["N"], golden["pico"]["N"])
    assert np.array_equal(geo["N"], golden["kat"]["pico_N"])
    assert np.allclose(geo["areas"], golden["pico"]["areas"])
    assert np.allclose(geo["volumes"], golden["pico"]["volumes"])
    assert np.allclose(geo["cell_centroids"], golden["pico"]["cell_centroids"])
    assert np.allclose(geo["face_centroids"], golden["pico"]["face_centroids"])
    assert np.allclose(geo["normals"], golden["pico"]["normals"])


def test_facepos_against_mrst_cell_faces(oracle, golden):
    """get_facepos (utils.jl:813-874) on the pico neighborship vs MRST's own cells.faces restricted to
    interior faces (sorted ascending as utils.jl:836-838 requires)."""
    p = golden["pico"]
    N = p["N"]
    nc = 9
    faces, facepos = oracle.get_facepos(N, nc)
    interior = p["interior"]
    new_id = np.cumsum(interior)  # old face (1-based) -> interior id
    fp, cf = p["cells_facePos"], p["cells_faces"]
    for c in range(nc):
        mine = faces[facepos[c] - 1: facepos[c + 1] - 1]
        mrst = [new_id[f - 1] for f in cf[fp[c] - 1: fp[c + 1] - 1] if interior[f - 1]]
        assert list(mine) == sorted(mrst)
    assert facepos[0] == 1 and facepos[-1] == 2 * N.shape[1] + 1


def test_half_face_map_signs_and_others(oracle):
    geo = oracle.cartesian_geometry((4, 3, 2))
    N, nc = geo["N"], geo["nc"]
    h = oracle.half_face_map(N, nc)
    for k in range(h["faces"].size):
        f, s, o, sg = h["faces"][k], h["self"][k], h["other"][k], h["face_sign"][k]
        if sg == 1:
            assert N[0, f - 1] == s and N[1, f - 1] == o
        else:
            assert sg == -1 and N[1, f - 1] == s and N[0, f - 1] == o
    # each face appears exactly twice
    assert np.array_equal(np.bincount(h["faces"])[1:], 2 * np.ones(geo["nf"], dtype=np.int64))


# --- KAT 1/2: transmissibilities -----------------------------------------------------------------------------
def test_trans_poisson_3x1(oracle):
    """CartesianMesh((3,1),(1.0,1.0)), coefficient 1 => T_hf = 6, T_f = 3 (SURVEY 8c KAT 1)."""
    geo = oracle.cartesian_geometry((3, 1), (1.0, 1.0))
    h = oracle.half_face_map(geo["N"], geo["nc"])
    Thf = oracle.half_face_trans(geo, np.ones((1, 3)), h)
    assert np.allclose(Thf, 6.0, rtol=1e-14)
    Tf = oracle.face_trans(Thf, h["faces"], geo["nf"])
    assert np.allclose(Tf, 3.0, rtol=1e-14)


def test_boundary_trans_unit_cube_is_one(oracle):
    """compute_boundary_trans == 1 on the unit-perm 2x2x2 unit mesh (test/utils.jl:294-298) -- the boundary
    half-trans uses the same half_face_trans formula (finite-volume.jl:220-222)."""
    geo = oracle.cartesian_geometry((2, 2, 2))
    A, n = geo["boundary_areas"], geo["boundary_normals"]
    C = geo["boundary_centroids"] - geo["cell_centroids"][:, geo["boundary_neighbors"] - 1]
    T = A * np.einsum("ij,ij->j", C, n) / np.einsum("ij,ij->j", C, C)
    assert np.all(T == 1.0)
    # interior: uniform 2x2x2, dx=0.5: T_hf = 0.25*0.25/0.0625 = 1, T_f = 0.5
    h = oracle.half_face_map(geo["N"], geo["nc"])
    Thf = oracle.half_face_trans(geo, np.ones((1, 8)), h)
    assert np.allclose(Thf, 1.0)
    assert np.allclose(oracle.face_trans(Thf, h["faces"], geo["nf"]), 0.5)


def test_trans_anisotropic_and_tensor(oracle):
    geo = oracle.cartesian_geometry((3, 2, 2), (3.0, 2.0, 2.0))
    h = oracle.half_face_map(geo["N"], geo["nc"])
    nc = geo["nc"]
    K3 = np.tile(np.array([[1.0], [2.0], [3.0]]), (1, nc))
    T3 = oracle.half_face_trans(geo, K3, h)
    K6 = np.tile(np.array([[1.0], [0.0], [0.0], [2.0], [0.0], [3.0]]), (1, nc))
    T6 = oracle.half_face_trans(geo, K6, h)
    assert np.array_equal(T3, T6)
    # x-faces see Kxx=1, y-faces Kyy=2, z-faces Kzz=3 ; dx=dy=dz=1 -> T_hf = K*1*0.5/0.25 = 2K
    nfx = 2 * 2 * 2
    nfy = 3 * 1 * 2
    for k, f in enumerate(h["faces"]):
        Kd = 1.0 if f <= nfx else (2.0 if f <= nfx + nfy else 3.0)
        assert np.isclose(T3[k], 2 * Kd)


def test_face_gdz(oracle):
    geo = oracle.cartesian_geometry((2, 2, 3))
    z = geo["cell_centroids"][2]
    gdz = oracle.face_gdz(geo["N"], z, g=10.0)
    N = geo["N"]
    assert np.allclose(gdz, -10.0 * (z[N[1] - 1] - z[N[0] - 1]))


# --- pattern / positions -------------------------------------------------------------------------------------
def test_pattern_is_sorted_with_diagonal(oracle):
    geo = oracle.cartesian_geometry((4, 3, 2))
    h = oracle.half_face_map(geo["N"], geo["nc"])
    rowptr, colidx = oracle.csr_pattern(geo["nc"], h)
    assert rowptr[-1] - 1 == geo["nc"] + 2 * geo["nf"]
    for r in range(geo["nc"]):
        cols = colidx[rowptr[r] - 1: rowptr[r + 1] - 1]
        assert np.all(np.diff(cols) > 0)
        assert (r + 1) in cols


@pytest.mark.parametrize("N", [1, 2, 3])
def test_block_positions(oracle, N):
    """find_jac_position BlockMajorLayout: ix = (pos-1)N^2 + N(d-1) + e (equations.jl:95-113); flux entries
    go to (row=other, col=self) (conservation.jl:202-212)."""
    geo = oracle.cartesian_geometry((3, 3))
    nc = geo["nc"]
    h = oracle.half_face_map(geo["N"], nc)
    rowptr, colidx = oracle.csr_pattern(nc, h)
    pa, pf = oracle.align(nc, N, 2, rowptr, colidx, h)
    for c in range(1, nc + 1):
        pos = oracle.lib().jo_find_sparse_position_csr(rowptr.ctypes.data_as(oracle.I64P),
                                                       colidx.ctypes.data_as(oracle.I64P), c, c)
        for e in range(1, N + 1):
            for d in range(1, N + 1):
                assert pa[(e - 1) * N + d - 1, c - 1] == (pos - 1) * N * N + N * (d - 1) + e
    for k in range(h["faces"].size):
        s, o = h["self"][k], h["other"][k]
        cols = colidx[rowptr[o - 1] - 1: rowptr[o] - 1]
        pos = rowptr[o - 1] + int(np.where(cols == s)[0][0])
        for e in range(1, N + 1):
            for d in range(1, N + 1):
                assert pf[(e - 1) * N + d - 1, k] == (pos - 1) * N * N + N * (d - 1) + e
    # all slots written exactly once per assembly
    allpos = np.concatenate([pa.ravel(), pf.ravel()])
    assert np.array_equal(np.sort(allpos), np.arange(1, (rowptr[-1] - 1) * N * N + 1))


@pytest.mark.parametrize("layout", [0, 1])
def test_scalar_layout_positions(oracle, layout):
    """EquationMajor row=(e-1)nc+cell, EntityMajor row=N(cell-1)+e (equations.jl:132-138)."""
    N = 2
    geo = oracle.cartesian_geometry((3, 2))
    nc = geo["nc"]
    h = oracle.half_face_map(geo["N"], nc)
    rowptr, colidx = oracle.csr_pattern(nc, h)
    srp, sci = oracle.csr_pattern_scalar(nc, N, layout, rowptr, colidx)
    for r in range(nc * N):
        assert np.all(np.diff(sci[srp[r] - 1: srp[r + 1] - 1]) > 0)
    pa, pf = oracle.align(nc, N, layout, rowptr, colidx, h, srp, sci)

    def rc(i, e):
        return (e - 1) * nc + i if layout == 0 else N * (i - 1) + e

    for k in range(h["faces"].size):
        s, o = h["self"][k], h["other"][k]
        for e in range(1, N + 1):
            for d in range(1, N + 1):
                p = pf[(e - 1) * N + d - 1, k]
                row = int(np.searchsorted(srp, p, side="right"))
                assert row == rc(o, e) and sci[p - 1] == rc(s, d)
    allpos = np.concatenate([pa.ravel(), pf.ravel()])
    assert np.array_equal(np.sort(allpos), np.arange(1, srp[-1]))


def test_layout_vectors(oracle, golden):
    """Block vs equation-major ordering of the reference's 2-cell / 7-dof state (test/adjoints/utils.jl:57-69)
    pins alignment_linear_index (equations.jl:132-138): eq-major nc*(e-1)+c, entity/block-major n*(c-1)+e."""
    k = golden["kat"]
    eq, blk = k["layout_eq"], k["layout_block"]
    nc, ndof = 2, 7
    ali = oracle.lib().jo_alignment_linear_index
    for c in range(1, nc + 1):
        for e in range(1, ndof + 1):
            i_eq = ali(c, e, nc, ndof, 0)
            i_blk = ali(c, e, nc, ndof, 2)
            assert i_eq == nc * (e - 1) + c and i_blk == ndof * (c - 1) + e
            assert eq[i_eq - 1] == blk[i_blk - 1]


# --- KAT 1: Poisson 3x1 known answer ---------------------------------------------------------------------------
def test_poisson_3x1_known_answer(oracle, golden):
    """test/test_systems/variable_poisson.jl:5-39: U - U[1] == [0, 1/3, 2/3]."""
    geo = oracle.cartesian_geometry((3, 1), (1.0, 1.0))
    nc = geo["nc"]
    sysm = oracle.TPFASystem(geo["N"], nc)
    Thf = oracle.half_face_trans(geo, np.ones((1, nc)), sysm.hfm)
    Tf = oracle.face_trans(Thf, sysm.hfm["faces"], geo["nf"])
    law = oracle.Law("poisson", dt=0.0)  # stationary variant with 1e-10 regulariser
    U = np.ones(nc)  # state0 U = 1.0
    for it in range(3):
        nz, r = sysm.assemble(law, U, U, np.ones(nc), Tf, src_cells=[1, nc], src_values=[1.0, -1.0])
        if np.max(np.abs(r)) < 1e-12 and it > 0:
            break
        A = dense_from_csr(nc, 1, sysm.rowptr, sysm.colidx, nz)
        U = U + np.linalg.solve(A, -r)
    assert np.allclose(U - U[0], golden["kat"]["poisson_3x1"], rtol=1e-8, atol=1e-8)


def test_residual_is_linear_and_jacobian_matches_fd(oracle):
    rng = np.random.default_rng(0)
    geo = oracle.cartesian_geometry((4, 3, 2))
    nc = geo["nc"]
    for kind, nblk in [("poisson", 1), ("compressible", 1), ("twophase", 2)]:
        sysm = oracle.TPFASystem(geo["N"], nc, nblk)
        Tf = rng.uniform(0.5, 2.0, geo["nf"])
        gdz = oracle.face_gdz(geo["N"], geo["cell_centroids"][2], g=9.81)
        vol = rng.uniform(0.5, 1.5, nc)
        law = oracle.Law(kind, dt=0.7, rho0=(1.0, 0.8), comp=(1e-2, 2e-2), mu=(1.0, 2.0), p_ref=1.0)
        if nblk == 1:
            X = rng.uniform(1.0, 2.0, nc)
            X0 = rng.uniform(1.0, 2.0, nc)
        else:
            X = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
            X0 = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
        nz, r = sysm.assemble(law, X, X0, vol, Tf, gdz)
        J = dense_from_csr(nc, nblk, sysm.rowptr, sysm.colidx, nz)
        Jfd = np.zeros_like(J)
        for j in range(nc * nblk):
            h = 1e-6
            Xp, Xm = X.copy(), X.copy()
            Xp[j] += h
            Xm[j] -= h
            _, rp = sysm.assemble(law, Xp, X0, vol, Tf, gdz)
            _, rm = sysm.assemble(law, Xm, X0, vol, Tf, gdz)
            Jfd[:, j] = (rp - rm) / (2 * h)
        assert np.allclose(J, Jfd, rtol=2e-6, atol=2e-6), kind
        # mass conservation: fluxes cancel pairwise => sum of residual == sum of accumulation
        acc, hf = oracle.update_equation(law, nc, sysm.hfm, X, X0, vol, Tf, gdz)
        assert np.allclose(r.reshape(nc, nblk).sum(0), acc[:, :, 0].sum(0), rtol=1e-10, atol=1e-10)


def test_poisson_row_sums(oracle):
    """J*1 == 1/dt*vol for the Poisson law (flux derivatives cancel in each row)."""
    geo = oracle.cartesian_geometry((5, 4))
    nc = geo["nc"]
    sysm = oracle.TPFASystem(geo["N"], nc)
    rng = np.random.default_rng(1)
    Tf = rng.uniform(0.5, 2.0, geo["nf"])
    U = rng.standard_normal(nc)
    law = oracle.Law("poisson", dt=0.25)
    nz, r = sysm.assemble(law, U, U, np.ones(nc), Tf)
    y = oracle.spmv(nc, 1, sysm.rowptr, sysm.colidx, nz, np.ones(nc))
    assert np.allclose(y, 4.0, rtol=1e-12)


# --- KAT 10: heat 2-D as a periodic TPFA Poisson law -------------------------------------------------------------
def periodic_heat_neighbors(nx, ny):
    """SimpleHeatSystem (heat_2d.jl:7-49) periodic 5-pt stencil expressed as a neighborship: every cell has a
    face to its right and its upper periodic neighbour."""
    N = []
    for j in range(ny):
        for i in range(nx):
            c = j * nx + i + 1
            N.append((c, j * nx + (i + 1) % nx + 1))
    for j in range(ny):
        for i in range(nx):
            c = j * nx + i + 1
            N.append((c, ((j + 1) % ny) * nx + i + 1))
    return np.array(N, dtype=np.int64).T.copy()


def test_heat_identity(oracle):
    """((I/dt) - Lap_h) T1 = T0/dt with the periodic 5-pt stencil (heat_2d.jl:32-51)."""
    nx = ny = 8
    nc = nx * ny
    h = 1.0 / nx
    N = periodic_heat_neighbors(nx, ny)
    sysm = oracle.TPFASystem(N, nc)
    Tf = np.full(N.shape[1], 1.0 / h ** 2)
    T0 = np.zeros(nc)
    T0[(nx // 2) * nx + nx // 2] = 100.0
    law = oracle.Law("poisson", dt=1.0)
    nz, r = sysm.assemble(law, T0, T0, np.ones(nc), Tf)
    A = dense_from_csr(nc, 1, sysm.rowptr, sysm.colidx, nz)
    T1 = T0 + np.linalg.solve(A, -r)
    # independent 5-pt stencil
    Tm = T1.reshape(ny, nx)
    lap = (np.roll(Tm, 1, 1) + np.roll(Tm, -1, 1) + np.roll(Tm, 1, 0) + np.roll(Tm, -1, 0) - 4 * Tm) / h ** 2
    assert np.allclose((Tm - T0.reshape(ny, nx)) / 1.0 - lap, 0.0, atol=1e-9)
    assert np.isclose(T1.sum(), T0.sum())


# --- sparse kernels ------------------------------------------------------------------------------------------------
def random_system(oracle, dims, bs, seed=0):
    rng = np.random.default_rng(seed)
    geo = oracle.cartesian_geometry(dims)
    nc = geo["nc"]
    h = oracle.half_face_map(geo["N"], nc)
    rowptr, colidx = oracle.csr_pattern(nc, h)
    nnzb = rowptr[-1] - 1
    nz = rng.standard_normal((nnzb, bs, bs)) * 0.3
    for r in range(nc):  # diagonally dominant blocks
        for p in range(rowptr[r] - 1, rowptr[r + 1] - 1):
            if colidx[p] == r + 1:
                nz[p] += 4.0 * np.eye(bs)
    return nc, rowptr, colidx, nz.reshape(-1)


@pytest.mark.parametrize("bs", [1, 2])
def test_spmv_vs_dense(oracle, bs):
    nc, rowptr, colidx, nz = random_system(oracle, (4, 3, 3), bs)
    A = dense_from_csr(nc, bs, rowptr, colidx, nz)
    rng = np.random.default_rng(2)
    x, y0 = rng.standard_normal(nc * bs), rng.standard_normal(nc * bs)
    assert np.allclose(oracle.spmv(nc, bs, rowptr, colidx, nz, x), A @ x, rtol=1e-13)
    assert np.allclose(oracle.spmv(nc, bs, rowptr, colidx, nz, x, y0, alpha=-2.0, beta=1.0), y0 - 2 * A @ x)
    assert np.allclose(oracle.spmv(nc, bs, rowptr, colidx, nz, x, y0, alpha=0.5, beta=3.0), 3 * y0 + 0.5 * A @ x)


def split_lu(n, bs, rowptr, colidx, lu):
    L = np.eye(n * bs)
    U = np.zeros((n * bs, n * bs))
    b = lu.reshape(-1, bs, bs)
    for r in range(n):
        for p in range(rowptr[r] - 1, rowptr[r + 1] - 1):
            c = colidx[p] - 1
            blk = b[p].T
            if c < r:
                L[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs] = blk
            elif c == r:
                U[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs] = np.linalg.inv(blk)  # stored inverted
            else:
                U[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs] = blk
    return L, U


@pytest.mark.parametrize("bs", [1, 2])
def test_ilu0_is_exact_on_pattern(oracle, bs):
    """ILU(0) definition: (L*U)[i,j] == A[i,j] for (i,j) in the pattern (StaticCSR/ilu0.jl:108-144)."""
    nc, rowptr, colidx, nz = random_system(oracle, (4, 4, 2), bs, seed=3)
    F = oracle.ILU0(nc, bs, rowptr, colidx, nz)
    lu = F.export(nz.size)
    L, U = split_lu(nc, bs, rowptr, colidx, lu)
    A = dense_from_csr(nc, bs, rowptr, colidx, nz)
    P = dense_from_csr(nc, bs, rowptr, colidx, np.ones_like(nz)) != 0
    assert np.allclose((L @ U)[P], A[P], rtol=1e-12, atol=1e-12)
    b = np.random.default_rng(4).standard_normal(nc * bs)
    assert np.allclose(F.apply(b), np.linalg.solve(U, np.linalg.solve(L, b)), rtol=1e-11)
    # refactor with new values reuses the maps (ilu0_csr!)
    F.refactor(2 * nz)
    assert np.allclose(F.apply(b), 0.5 * np.linalg.solve(U, np.linalg.solve(L, b)), rtol=1e-11)


def test_ilu0_tridiagonal_is_exact_lu(oracle):
    """1-D chain: no fill-in, so ILU(0) == LU and the apply is an exact solve."""
    geo = oracle.cartesian_geometry((12,), (1.0,))
    nc = geo["nc"]
    h = oracle.half_face_map(geo["N"], nc)
    rowptr, colidx = oracle.csr_pattern(nc, h)
    nz = np.where(colidx == np.repeat(np.arange(1, nc + 1), np.diff(rowptr)), 2.5, -1.0)
    F = oracle.ILU0(nc, 1, rowptr, colidx, nz)
    b = np.arange(1.0, nc + 1)
    A = dense_from_csr(nc, 1, rowptr, colidx, nz)
    assert np.allclose(F.apply(b), np.linalg.solve(A, b), rtol=1e-12)


@pytest.mark.parametrize("bs", [1, 2])
def test_block_jacobi_ilu0(oracle, bs):
    """ilu0_csr(A, partition) (par_ilu0.jl:47-90) == independent ILU(0) of each diagonal block."""
    nc, rowptr, colidx, nz = random_system(oracle, (6, 4), bs, seed=5)
    part = oracle.partition_linear(3, nc)
    F = oracle.ILU0(nc, bs, rowptr, colidx, nz, partition=part)
    A = dense_from_csr(nc, bs, rowptr, colidx, nz)
    b = np.random.default_rng(6).standard_normal(nc * bs)
    x = F.apply(b)
    for blk in range(1, 4):
        rows = np.where(part == blk)[0]
        dof = (rows[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        Ab = A[np.ix_(dof, dof)]
        # reference sub-ILU through the serial oracle on the extracted block
        sub_rowptr = [1]
        sub_col, sub_nz = [], []
        nzb = nz.reshape(-1, bs * bs)
        loc = {g: i + 1 for i, g in enumerate(rows + 1)}
        for r in rows:
            for p in range(rowptr[r] - 1, rowptr[r + 1] - 1):
                if colidx[p] in loc:
                    sub_col.append(loc[colidx[p]])
                    sub_nz.append(nzb[p])
            sub_rowptr.append(len(sub_col) + 1)
        Fs = oracle.ILU0(len(rows), bs, np.array(sub_rowptr), np.array(sub_col), np.array(sub_nz).reshape(-1))
        assert np.allclose(x[dof], Fs.apply(b[dof]), rtol=1e-12)
        assert Ab.shape[0] == dof.size


@pytest.mark.parametrize("side", ["right", "left"])
def test_bicgstab_converges(oracle, side):
    nc, rowptr, colidx, nz = random_system(oracle, (8, 8, 4), 1, seed=7)
    A = dense_from_csr(nc, 1, rowptr, colidx, nz)
    b = np.random.default_rng(8).standard_normal(nc)
    F = oracle.ILU0(nc, 1, rowptr, colidx, nz)
    x, st = oracle.bicgstab(nc, 1, rowptr, colidx, nz, b, prec=F, side=side, rtol=1e-8, atol=1e-14, itmax=100)
    assert st["solved"] and st["iterations"] < 30
    assert len(st["residuals"]) == st["iterations"] + 1
    assert np.linalg.norm(A @ x - b) <= 1e-6 * np.linalg.norm(b)
    x0, st0 = oracle.bicgstab(nc, 1, rowptr, colidx, nz, b, prec=None, rtol=1e-8, atol=1e-14, itmax=200)
    assert st0["solved"] and st0["iterations"] >= st["iterations"]


# --- partitions / distributed helpers ----------------------------------------------------------------------------------
def test_partition_kats(oracle, golden):
    k = golden["kat"]
    assert np.array_equal(oracle.compress_partition(k["compress_in"]), k["compress_out"])  # test/partitioning.jl:8
    assert np.array_equal(oracle.compress_partition(np.arange(1, 11)), np.arange(1, 11))  # :9
    for npart in range(1, 11):  # test/partitioning.jl:23-28 validity for np = 1..10
        p = oracle.partition_linear(npart, 100)
        assert p.min() == 1 and p.max() == npart and np.all(np.bincount(p)[1:] > 0)
    assert np.array_equal(oracle.partition_linear(3, 7), np.ceil(np.arange(1, 8) / (7 / 3)).astype(np.int64))


def test_partition_boundary_and_remap(oracle):
    geo = oracle.cartesian_geometry((4, 2))
    N, nc = geo["N"], geo["nc"]
    p = np.array([1, 1, 2, 2, 1, 1, 2, 2])
    b1 = oracle.partition_boundary(N, p, 1)
    b2 = oracle.partition_boundary(N, p, 2)
    assert sorted(b1) == [3, 7] and sorted(b2) == [2, 6]
    rem, counts = oracle.remap_global_indices(p, 2)
    assert list(counts) == [4, 4]
    assert list(rem) == [1, 2, 5, 6, 3, 4, 7, 8]  # owned-first contiguous per rank, findall order


def test_unit_diagonalize(oracle):
    nc, rowptr, colidx, nz = random_system(oracle, (3, 3), 2, seed=9)
    r = np.ones(nc * 2)
    nz2, r2 = oracle.unit_diagonalize(nc, 6, 2, rowptr, colidx, nz, r)
    A = dense_from_csr(nc, 2, rowptr, colidx, nz2)
    assert np.array_equal(A[12:, :][:, 12:], -np.eye(6)) and np.all(A[12:, :12] == 0)
    assert np.all(r2[12:] == 0) and np.all(r2[:12] == 1)
    A0 = dense_from_csr(nc, 2, rowptr, colidx, nz)
    assert np.array_equal(A[:12], A0[:12])


@pytest.mark.parametrize("side", ["right", "left"])
def test_gmres_converges_and_residual_history_is_true_residual(oracle, side):
    """GMRES (Krylov.jl gmres!, third party): the Givens-recurrence residual equals the true (preconditioned) residual
    norm, the history is non-increasing, and without restarts it converges in <= n iterations."""
    nc, rowptr, colidx, nz = random_system(oracle, (6, 5, 4), 1, seed=17)
    A = dense_from_csr(nc, 1, rowptr, colidx, nz)
    b = np.random.default_rng(18).standard_normal(nc)
    F = oracle.ILU0(nc, 1, rowptr, colidx, nz)
    x, st = oracle.gmres(nc, 1, rowptr, colidx, nz, b, prec=F, side=side, rtol=1e-10, atol=1e-14, itmax=60)
    assert st["solved"] and st["iterations"] < 25
    res = st["residuals"]
    assert np.all(np.diff(res) <= 1e-12 * res[0])
    r = b - A @ x
    rn = np.linalg.norm(F.apply(r)) if side == "left" else np.linalg.norm(r)
    assert abs(rn - res[-1]) <= 1e-8 * res[0]
    xb, stb = oracle.bicgstab(nc, 1, rowptr, colidx, nz, b, prec=F, side=side, rtol=1e-10, atol=1e-14, itmax=100)
    assert np.allclose(x, xb, rtol=1e-6, atol=1e-8)
    x0, st0 = oracle.gmres(nc, 1, rowptr, colidx, nz, b, prec=None, rtol=1e-10, atol=1e-14, itmax=nc)
    assert st0["solved"] and st0["iterations"] <= nc


def test_multigraph_two_cells_two_faces_hand_computed(oracle):
    """Two cells joined by TWO faces (T1, T2), Poisson law, hand-computed from the reference's loops: the pattern has one
    off-diagonal entry per cell pair (sparse() merges duplicates, conservation.jl:486-505); residual and diagonal sum both
    half-faces (fill_conservation_eq!, conservation.jl:373-430); the off-diagonal slot is assigned per half-face in ascending
    face order (update_jacobian_inner!, ad.jl:74-76), so it ends up holding -T2 only."""
    N = np.array([[1, 2], [2, 1]], dtype=np.int64)           # columns = faces: N[:,0] = (1,2), N[:,1] = (2,1)
    nc, T1, T2, dt = 2, 3.0, 5.0, 0.5
    U, U0, vol = np.array([2.0, 1.0]), np.array([1.5, 1.25]), np.array([1.0, 1.0])
    osys = oracle.TPFASystem(N, nc)
    assert list(osys.rowptr) == [1, 3, 5] and list(osys.colidx) == [1, 2, 1, 2]
    nz, r = osys.assemble(oracle.Law("poisson", dt), U, U0, vol, np.array([T1, T2]))
    q = (T1 + T2) * (U[0] - U[1])
    assert np.allclose(r, [(U[0] - U0[0]) / dt + q, (U[1] - U0[1]) / dt - q], rtol=1e-14)
    assert np.allclose(nz, [1 / dt + T1 + T2, -T2, -T2, 1 / dt + T1 + T2], rtol=1e-14)


def test_bench_cartesian_mesh_matches_the_oracle_geometry(oracle):
    """meshgen.cartesian_mesh (bench / GPU parity input) against cartesian_geometry + half_face_trans + face_trans of the oracle:
    same neighbourship, volumes, centroids and (with its per-cell permeability) the same face transmissibilities."""
    from jutul_amd import cartesian_mesh
    dims, size = (5, 4, 3), (2.0, 1.0, 3.0)
    g = cartesian_mesh(*dims, size=size, scramble=False)
    geo = oracle.cartesian_geometry(dims, size)
    assert np.array_equal(g["N"], geo["N"]) and g["nc"] == geo["nc"] and g["nf"] == geo["nf"]
    assert np.allclose(g["volumes"], geo["volumes"], rtol=1e-14) and np.allclose(g["cell_centroids"], geo["cell_centroids"], rtol=1e-14)
    h = oracle.half_face_map(geo["N"], geo["nc"])
    Tf = oracle.face_trans(oracle.half_face_trans(geo, g["perm_k"][None, :], h), h["faces"], geo["nf"])
    assert np.allclose(g["T"], Tf, rtol=1e-13)
    gs = cartesian_mesh(*dims, size=size)                     # scrambled numbering: the same grid under a permutation
    key = lambda N, T: sorted(zip(np.minimum(N[0], N[1]).tolist(), np.maximum(N[0], N[1]).tolist()))
    assert len(set(key(gs["N"], gs["T"]))) == g["nf"] and np.isclose(np.sort(gs["T"]).sum(), np.sort(g["T"]).sum(), rtol=1e-12)


def test_config0_simple_heat_50x50_newton_loop(oracle):
    """BASELINE configs[0] on the CPU, at its own size: SimpleHeat on the 50 x 50 periodic grid (heat_2d.jl:7-49; initial box of
    docs/src/index.md:38-45) through the oracle's restatement of the WHOLE per-step path -- assemble -> converged? -> ILU(0) +
    BiCGStab -> update -> assemble -> converged (simulator.jl:392-455: a linear problem takes two assemblies and one solve) -- for
    five implicit steps.  Checked against the independent 5-point stencil identity ((I/dt) - Lap_h) T1 = T0/dt, conservation of
    the total, the maximum principle, and decay towards the mean."""
    nx = ny = 50
    nc = nx * ny
    h = 1.0 / nx
    N = periodic_heat_neighbors(nx, ny)
    sysm = oracle.TPFASystem(N, nc)
    Tf = np.full(N.shape[1], 1.0 / h ** 2)
    T = np.zeros((ny, nx))
    T[20:30, 20:30] = 100.0
    T = T.reshape(-1)
    vol = np.ones(nc)
    dt = 1e-4
    law = oracle.Law("poisson", dt=dt)
    spread = []
    for step in range(5):
        T0 = T.copy()
        solves = 0
        for it in range(3):
            nz, r = sysm.assemble(law, T, T0, vol, Tf)
            if np.abs(r).max() < 1e-9 * 100.0 / dt and it > 0:      # (|r| starts at ~T/dt = 1e6; the solve leaves ~1e-12 of that)
                break
            F = oracle.ILU0(nc, 1, sysm.rowptr, sysm.colidx, nz)
            x, info = oracle.bicgstab(nc, 1, sysm.rowptr, sysm.colidx, nz, r, prec=F, side="right", rtol=1e-12, atol=0.0, itmax=200)
            assert info["solved"] and 1 <= info["iterations"] < 60
            T = T - x
            solves += 1
        assert solves == 1 and it == 1                      # two assemblies, one solve
        Tm, T0m = T.reshape(ny, nx), T0.reshape(ny, nx)
        lap = (np.roll(Tm, 1, 1) + np.roll(Tm, -1, 1) + np.roll(Tm, 1, 0) + np.roll(Tm, -1, 0) - 4 * Tm) / h ** 2
        assert np.abs((Tm - T0m) / dt - lap).max() <= 1e-7 * 100.0 / dt
        assert np.isclose(T.sum(), T0.sum(), rtol=1e-11)    # periodic domain: the total is conserved
        assert T.min() >= -1e-9 and T.max() <= T0.max() + 1e-9
        spread.append(T.max() - T.min())
    assert all(b < a for a, b in zip(spread, spread[1:]))
