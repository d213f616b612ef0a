"""GPU parity tests: the HIP path (through the C ABI of libjutul_hip.so) against the CPU oracle on the same
seeded inputs.  Integer tables bit-exact; floating point within the tolerances written in each test
(fp64; reductions are re-associated on the GPU, hence relative 1e-12 .. 1e-10, not bit equality)."""
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12  # fp64 assembly / SpMV / ILU tolerance (relative to the largest entry of the compared array)


@contextlib.contextmanager
def options(ctx, **kv):
    """Context options (jh_context_set_option) for the duration of a block; the context fixture is shared between tests."""
    old = {k: ctx.get_option(k) for k in kv}
    try:
        for k, v in kv.items():
            ctx.set_option(k, v)
        yield
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)


def relerr(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


@pytest.fixture(scope="module")
def ctx(ja):
    return ja.HIPContext(0)


def tet_case(ja, dims=(5, 4, 3), seed=0):
    g = ja.tet_lattice_mesh(*dims)
    rng = np.random.default_rng(seed)
    g["Tn"] = g["T"] / g["T"].mean()
    g["gdz"] = 9.81 * rng.standard_normal(g["nf"]) * 0.01
    return g, rng


# ---- a-1..a-4: integer tables, bit exact -----------------------------------------------------------------------------
@pytest.mark.parametrize("reorder", ["none", "blocks"])
@pytest.mark.parametrize("nblk", [1, 2])
def test_connectivity_pattern_positions_bit_exact(ja, ctx, oracle, reorder, nblk):
    g, _ = tet_case(ja)
    nc = g["nc"]
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=nblk, reorder=reorder, block_rows=64)
    h = oracle.half_face_map(g["N"], nc)
    c = disc.conn
    assert np.array_equal(c["face_pos"], h["face_pos"])
    assert np.array_equal(c["self"], h["self"]) and np.array_equal(c["other"], h["other"])
    assert np.array_equal(c["face"], h["faces"]) and np.array_equal(c["face_sign"], h["face_sign"])
    rowptr, colidx = disc.pattern()
    orp, oci = oracle.csr_pattern(nc, h)
    assert np.array_equal(rowptr, orp) and np.array_equal(colidx, oci)
    pa, pf = disc.jacobian_positions()
    opa, opf = oracle.align(nc, nblk, 2, orp, oci, h)
    assert np.array_equal(pa, opa) and np.array_equal(pf, opf)
    perm, bp = disc.ordering()
    assert np.array_equal(np.sort(perm), np.arange(1, nc + 1))
    if reorder == "blocks":
        assert bp[0] == 0 and bp[-1] == nc and np.all(np.diff(bp) > 0) and np.diff(bp).max() <= 80


def test_cartesian_and_pico_connectivity(ja, ctx, oracle, golden):
    N = golden["pico"]["N"]
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, 9)
    h = oracle.half_face_map(N, 9)
    assert np.array_equal(disc.conn["face"], h["faces"]) and np.array_equal(disc.conn["other"], h["other"])
    assert np.array_equal(ja.cartesian_neighbors((3, 3, 1)), N)


def test_rejects_bad_input(ja, ctx):
    with pytest.raises(ja.JutulHIPError):
        ja.TwoPointPotentialFlowHardCoded(ctx, np.array([[1], [5]]), 3)  # neighbour out of range
    d = ja.TwoPointPotentialFlowHardCoded(ctx, np.array([[1, 1], [2, 2]]), 2)  # duplicate pair: a multigraph, accepted like the reference does
    assert d.nnzb == 4 and d.nhf == 4
    with pytest.raises(ja.JutulHIPError):
        ja.TwoPointPotentialFlowHardCoded(ctx, np.array([[1], [1]]), 2)  # self loop


# ---- a-5/a-6: assembly -------------------------------------------------------------------------------------------------
def assemble_both(ja, ctx, oracle, g, rng, kind, reorder, dt=0.7, with_sources=True, gravity=True):
    nc = g["nc"]
    nblk = 2 if kind == "twophase" else 1
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=nblk, reorder=reorder, block_rows=64)
    par = dict(rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
    law = ja.ConservationLaw(disc, kind, **par)
    if nblk == 1:
        X, X0 = rng.uniform(1.0, 2.0, nc), rng.uniform(1.0, 2.0, nc)
    else:
        X = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
        X0 = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
    vol = g["volumes"]
    law.set_face_trans(g["Tn"])
    law.set_volumes(vol)
    use_g = gravity and kind != "poisson"
    if use_g:
        law.set_face_gdz(g["gdz"])
    law.set_state(X)
    law.set_state0(X0)
    src_c = [3, nc, 3] if with_sources else []
    src_v = rng.standard_normal(len(src_c) * nblk)
    if with_sources:
        law.set_sources(src_c, src_v)
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    osys = oracle.TPFASystem(g["N"], nc, nblk)
    olaw = oracle.Law(kind, dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    nz_o, r_o = osys.assemble(olaw, X, X0, vol, g["Tn"], g["gdz"] if use_g else None, src_c if with_sources else None,
                              src_v if with_sources else None)
    return lsys, law, disc, osys, nz_o, r_o


@pytest.mark.parametrize("reorder", ["none", "blocks"])
@pytest.mark.parametrize("kind", ["poisson", "compressible", "twophase"])
def test_assembly_parity(ja, ctx, oracle, kind, reorder):
    g, rng = tet_case(ja, (6, 5, 4), seed=1)
    lsys, law, disc, osys, nz_o, r_o = assemble_both(ja, ctx, oracle, g, rng, kind, reorder)
    assert relerr(lsys.r.download(), r_o) < RTOL
    assert relerr(lsys.jac.nzval, nz_o) < RTOL
    # and per entry against the scale of its OWN row (sum of the row's block magnitudes), not the largest entry of the array:
    # with the 100x permeability contrast of the grid the small entries are pinned as tightly as the large ones
    nn = law.N * law.N
    blk = np.abs(nz_o).reshape(-1, nn)
    row_of = np.repeat(np.arange(g["nc"]), np.diff(osys.rowptr))
    row_scale = np.zeros(g["nc"])
    np.add.at(row_scale, row_of, blk.sum(axis=1))
    assert np.all(np.abs(lsys.jac.nzval - nz_o).reshape(-1, nn).max(axis=1) <= 1e-13 * row_scale[row_of])
    rs = np.repeat(row_scale * 2.5, law.N) + np.abs(r_o)
    assert np.all(np.abs(lsys.r.download() - r_o) <= 1e-13 * rs)
    err = law.convergence_criterion(lsys.r)
    assert np.allclose(err, oracle.convergence(law.N, g["nc"], r_o), rtol=1e-12)
    assert np.array_equal(law.get_state().shape, (g["nc"] * law.N,))


def test_assembly_stationary_poisson_regulariser(ja, ctx, oracle):
    g, rng = tet_case(ja, (3, 3, 2), seed=2)
    lsys, law, disc, osys, nz_o, r_o = assemble_both(ja, ctx, oracle, g, rng, "poisson", "blocks", dt=0.0)
    assert relerr(lsys.r.download(), r_o) < RTOL and relerr(lsys.jac.nzval, nz_o) < RTOL
    rowptr, colidx = disc.pattern()
    d0 = rowptr[0] - 1 + int(np.where(colidx[rowptr[0] - 1: rowptr[1] - 1] == 1)[0][0])
    others = g["Tn"][osys.hfm["faces"][osys.hfm["face_pos"][0] - 1: osys.hfm["face_pos"][1] - 1] - 1].sum()
    assert np.isclose(lsys.jac.nzval[d0], others + 1e-10, rtol=1e-14)


def test_assembly_larger_tiles(ja, ctx, oracle):
    """> 1 tile per XCD chunk, rows straddling tiles, boundary rows."""
    g, rng = tet_case(ja, (16, 12, 10), seed=3)
    for kind in ["poisson", "twophase"]:
        lsys, law, disc, osys, nz_o, r_o = assemble_both(ja, ctx, oracle, g, rng, kind, "blocks")
        assert relerr(lsys.r.download(), r_o) < RTOL and relerr(lsys.jac.nzval, nz_o) < RTOL


def test_heat_2d_periodic_config1(ja, ctx, oracle):
    """BASELINE config 1: SimpleHeat on a 50x50 periodic Cartesian grid (heat_2d.jl:7-49) expressed as the TPFA
    Poisson law on the periodic neighborship; exact check ((I/dt) - Lap_h) T1 = T0/dt."""
    nx = ny = 50
    nc = nx * ny
    h = 1.0 / nx
    N = []
    for j in range(ny):
        for i in range(nx):
            N.append((j * nx + i + 1, j * nx + (i + 1) % nx + 1))
    for j in range(ny):
        for i in range(nx):
            N.append((j * nx + i + 1, ((j + 1) % ny) * nx + i + 1))
    N = np.array(N, dtype=np.int64).T.copy()
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(np.full(N.shape[1], 1.0 / h ** 2))
    T0 = np.zeros((ny, nx))
    T0[20:30, 20:30] = 100.0  # docs/src/index.md:38-45 box
    law.set_state(T0.reshape(-1))
    law.set_state0(T0.reshape(-1))
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(),
                                             relative_tolerance=1e-12, max_iterations=200))
    its = sim.solve_timestep(1.0)
    T1 = law.get_state().reshape(ny, nx)
    lap = (np.roll(T1, 1, 1) + np.roll(T1, -1, 1) + np.roll(T1, 1, 0) + np.roll(T1, -1, 0) - 4 * T1) / h ** 2
    assert np.abs((T1 - T0) / 1.0 - lap).max() < 1e-6 * 100.0 / h ** 2 * 1e-3
    assert np.isclose(T1.sum(), T0.sum(), rtol=1e-7)
    assert its == 2  # linear problem: 2 assemblies + 1 solve per ministep


# ---- a-10: SpMV --------------------------------------------------------------------------------------------------------
def random_csr(oracle, dims, bs, seed):
    rng = np.random.default_rng(seed)
    geo = oracle.cartesian_geometry(dims)
    nc = geo["nc"]
    h = oracle.half_face_map(geo["N"], nc)
    rowptr, colidx = oracle.csr_pattern(nc, h)
    nnzb = rowptr[-1] - 1
    nz = rng.standard_normal((nnzb, bs, bs)) * 0.3
    rows = np.repeat(np.arange(1, nc + 1), np.diff(rowptr))
    nz[colidx == rows] += 4.0 * np.eye(bs)
    return nc, rowptr, colidx, nz.reshape(-1), rng


@pytest.mark.parametrize("bs", [1, 2, 3])
def test_spmv_parity(ja, ctx, oracle, bs):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (9, 8, 7), bs, seed=4)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=bs, rowptr=rowptr, colidx=colidx, nzval=nz)
    assert np.array_equal(A.nzval, nz)
    x, y0 = rng.standard_normal(nc * bs), rng.standard_normal(nc * bs)
    xd, yd = A.new_vector(x), A.new_vector(y0)
    for alpha, beta in [(1.0, 0.0), (-2.0, 1.0), (0.5, 3.0)]:
        yd.upload(y0)
        ja.mul_(yd, A, xd, alpha, beta)
        ref = oracle.spmv(nc, bs, rowptr, colidx, nz, x, y0, alpha, beta)
        assert relerr(yd.download(), ref) < RTOL
    assert abs(xd.dot(yd) - x @ yd.download()) < 1e-10 * abs(x @ yd.download()) + 1e-12


@pytest.mark.parametrize("dims,reorder", [((9, 8, 7), "blocks"), ((5, 1, 1), "none"), ((33, 17, 1), "blocks"), ((4, 4, 4), "none")])
def test_spmv_jagged_slice_layout_matches_csr_bitwise(ja, ctx, oracle, dims, reorder):
    """The Krylov loop multiplies out of a jagged-slice copy of the Jacobian (64-row slices, rows sorted by length, entries stored
    diagonal by diagonal): same left-to-right accumulation (mat.jl:41-61) => the same bits as the CSR tile kernel, and the oracle's
    product to rounding.  Ragged last slice, rows of every length 1..7, alpha/beta forms."""
    N = ja.cartesian_neighbors(dims)
    nc = int(np.prod(dims))
    rng = np.random.default_rng(11)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder=reorder, block_rows=64)
    A = ja.StaticSparsityMatrixCSR(disc)
    rowptr, colidx = disc.pattern()
    nz = rng.standard_normal(A.nnzb)
    A.nzval = nz
    x, y0 = rng.standard_normal(nc), rng.standard_normal(nc)
    for alpha, beta in [(1.0, 0.0), (-2.0, 1.0), (0.5, 3.0)]:
        ya = ja.mul_(ja.DeviceVector(disc, y0), A, ja.DeviceVector(disc, x), alpha, beta).download()
        yb = ja.mul_(ja.DeviceVector(disc, y0), A, ja.DeviceVector(disc, x), alpha, beta, jagged=True).download()
        assert np.array_equal(ya, yb)
        assert relerr(yb, oracle.spmv(nc, 1, rowptr, colidx, nz, x, y0, alpha, beta)) < RTOL
    # new values after the first product: the copy is refreshed, not cached
    nz2 = rng.standard_normal(A.nnzb)
    A.nzval = nz2
    yb = ja.mul_(ja.DeviceVector(disc), A, ja.DeviceVector(disc, x), jagged=True).download()
    assert relerr(yb, oracle.spmv(nc, 1, rowptr, colidx, nz2, x)) < RTOL


@pytest.mark.parametrize("case", ["lattice", "far", "tiny"])
def test_spmv_jagged_16bit_column_codes(ja, oracle, case):
    """Large matrices store the jagged-slice columns as 16-bit codes (window origin of the slice + code, or an index into the
    slice's list of far columns): the same bits as the CSR kernel, also with columns far outside every window, on the first
    slices (window origin negative), the last ragged slice, and a matrix smaller than one window."""
    ctx = ja.HIPContext(0, spmv_col_bits=16)   # (the default switches at 3M rows; read when the layout is built)
    rng = np.random.default_rng(12)
    if case == "lattice":
        g = ja.tet_lattice_mesh(13, 11, 9)
        N, nc, reorder = g["N"], g["nc"], "blocks"
    elif case == "tiny":
        N, nc, reorder = ja.cartesian_neighbors((7, 3, 1)), 21, "none"
    else:   # a ring plus random long-range pairs: most rows reach 60 000+ rows away, host order kept
        nc, reorder = 150_001, "none"
        ring = np.stack([np.arange(1, nc + 1), np.roll(np.arange(1, nc + 1), -1)])
        perm = rng.permutation(nc)
        a, b = perm[: nc // 2] + 1, perm[nc // 2: 2 * (nc // 2)] + 1
        far = np.stack([a, b])[:, np.abs(a - b) > 1]
        N = np.concatenate([ring[:, :-1], far], axis=1)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder=reorder, block_rows=128)
    A = ja.StaticSparsityMatrixCSR(disc)
    rowptr, colidx = disc.pattern()
    nz = rng.standard_normal(A.nnzb)
    A.nzval = nz
    x, y0 = rng.standard_normal(nc), rng.standard_normal(nc)
    for alpha, beta in [(1.0, 0.0), (-2.0, 1.0)]:
        ya = ja.mul_(ja.DeviceVector(disc, y0), A, ja.DeviceVector(disc, x), alpha, beta).download()
        yb = ja.mul_(ja.DeviceVector(disc, y0), A, ja.DeviceVector(disc, x), alpha, beta, jagged=True).download()
        assert np.array_equal(ya, yb)
        assert relerr(yb, oracle.spmv(nc, 1, rowptr, colidx, nz, x, y0, alpha, beta)) < RTOL
    if case == "far":
        assert (np.abs(colidx - np.repeat(np.arange(1, nc + 1), np.diff(rowptr))) > 57344).mean() > 0.05   # the far list is exercised


def test_spmv_jagged_rejects_blocks_and_long_rows(ja, ctx, oracle):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (4, 4, 3), 2, seed=4)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=2, rowptr=rowptr, colidx=colidx, nzval=nz)
    with pytest.raises(ja.JutulHIPError, match="jagged"):
        ja.mul_(A.new_vector(), A, A.new_vector(), jagged=True)


def test_spmv_long_rows_and_empty_offdiag(ja, ctx, oracle):
    """Arrow matrix: one dense row (> TILE_NNZ entries -> long-row path), diagonal elsewhere."""
    n = 3000
    rp, ci, nz = [1], [], []
    rng = np.random.default_rng(5)
    for r in range(1, n + 1):
        cols = list(range(1, n + 1)) if r == 7 else sorted({r, 7})
        ci += cols
        nz += list(rng.standard_normal(len(cols)))
        rp.append(len(ci) + 1)
    rp, ci, nz = np.array(rp), np.array(ci), np.array(nz)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=n, bs=1, rowptr=rp, colidx=ci, nzval=nz)
    x = rng.standard_normal(n)
    y = ja.mul_(A.new_vector(), A, A.new_vector(x))
    assert relerr(y.download(), oracle.spmv(n, 1, rp, ci, nz, x)) < 1e-11


def test_spmv_dimension_mismatch_raises(ja, ctx, oracle):
    nc, rowptr, colidx, nz, _ = random_csr(oracle, (3, 3), 1, seed=6)
    nc2, rp2, ci2, nz2, _ = random_csr(oracle, (4, 3), 1, seed=6)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=1, rowptr=rowptr, colidx=colidx, nzval=nz)
    B = ja.StaticSparsityMatrixCSR(context=ctx, n=nc2, bs=1, rowptr=rp2, colidx=ci2, nzval=nz2)
    with pytest.raises(ja.JutulHIPError):
        ja.mul_(A.new_vector(), A, B.new_vector())


# ---- a-11..a-13: ILU(0) ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("mode", ["serial", "partition_linear", "partition_scattered"])
def test_ilu0_factor_and_apply_parity(ja, ctx, oracle, bs, mode):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (8, 7, 5), bs, seed=7)
    part = None
    if mode == "partition_linear":
        part = oracle.partition_linear(6, nc)
    elif mode == "partition_scattered":
        part = rng.integers(1, 5, nc)
    from tests import _ilu_checks as ck
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=bs, rowptr=rowptr, colidx=colidx, nzval=nz)
    F = ja.ilu0_csr(A, part)
    info = F.info()
    # factor values entry by entry against the size of each entry's own terms, the solve as a componentwise backward error with
    # the oracle's factors (tests/_ilu_checks.py) -- nothing scaled by the largest entry of an array
    Fo, lu_o, Lm, Um = ck.oracle_factors(oracle, nc, bs, rowptr, colidx, nz, part)
    ck.check_factor_values(F.factor_values(), lu_o, Lm, Um, nc, bs, rowptr, colidx)
    b = rng.standard_normal(nc * bs)
    x = ja.ldiv_(A.new_vector(), F, A.new_vector(b)).download()
    ck.check_triangular_solve(x, b, Lm, Um, levels=info["max_levels"])
    assert relerr(x, Fo.apply(b)) < 1e-10
    # ilu0_csr!: refactor after the values change (a power of two: the factors scale exactly, the solve by its inverse)
    A.nzval = 2.0 * nz
    F.update_preconditioner(A)
    x2 = ja.ldiv_(A.new_vector(), F, A.new_vector(b)).download()
    ck.check_triangular_solve(2.0 * x2, b, Lm, Um, levels=info["max_levels"])
    assert info["nblocks"] == (1 if part is None else int(part.max()))


def test_ilu0_tridiagonal_exact(ja, ctx, oracle):
    geo = oracle.cartesian_geometry((200,), (1.0,))
    nc = geo["nc"]
    h = oracle.half_face_map(geo["N"], nc)
    rowptr, colidx = oracle.csr_pattern(nc, h)
    rows = np.repeat(np.arange(1, nc + 1), np.diff(rowptr))
    nz = np.where(colidx == rows, 2.5, -1.0)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=1, rowptr=rowptr, colidx=colidx, nzval=nz)
    F = ja.ilu0_csr(A)
    b = np.arange(1.0, nc + 1)
    x = ja.ldiv_(A.new_vector(), F, A.new_vector(b)).download()
    import scipy.sparse as sp
    M = sp.csr_matrix((nz, colidx - 1, rowptr - 1), shape=(nc, nc))
    assert np.linalg.norm(M @ x - b) < 1e-9 * np.linalg.norm(b)
    assert F.info()["max_levels"] == nc


def test_ilu0_device_blocks_on_tpfa_jacobian(ja, ctx, oracle):
    """Block-Jacobi ILU(0) over the discretisation's own device blocks == oracle with the same partition."""
    g, rng = tet_case(ja, (8, 6, 5), seed=8)
    lsys, law, disc, osys, nz_o, r_o = assemble_both(ja, ctx, oracle, g, rng, "poisson", "blocks")
    perm, bp = disc.ordering()
    part = np.zeros(g["nc"], dtype=np.int64)
    for b in range(len(bp) - 1):
        part[perm[bp[b]: bp[b + 1]] - 1] = b + 1
    F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    # the device ordering is a different elimination order inside each block: compare against the oracle on the
    # permuted system (same orientation), i.e. the mathematical definition rather than the host ordering
    nc = g["nc"]
    import scipy.sparse as sp
    A = sp.csr_matrix((nz_o, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
    p0 = perm - 1
    Ap = A[p0][:, p0].tocsr()
    Ap.sort_indices()
    part_p = part[p0]
    Fo = oracle.ILU0(nc, 1, Ap.indptr + 1, Ap.indices + 1, Ap.data, partition=part_p)
    b = rng.standard_normal(nc)
    x = F.apply(lsys.jac.new_vector(), lsys.jac.new_vector(b)).download()
    assert relerr(x[p0], Fo.apply(b[p0])) < 1e-10
    assert F.info()["nblocks"] == len(bp) - 1


@pytest.mark.parametrize("kind", ["poisson", "twophase"])
def test_preconditioned_operator_fused_product(ja, ctx, oracle, kind):
    """jh_ilu0_apply_mul: x = M^-1 b and q = A x formed in one pass over the column-scaled pivot-only factors (the product
    inside the preconditioner apply of the right-preconditioned Krylov loop) == oracle spmv(ilu_apply(b)) at 1e-10, for scalar
    and 2x2-block matrices; the standalone apply on the same factors, the factor export and BiCGStab through the fused
    operator agree with the unfused path (option fused_product = 0 at solve time) to rounding."""
    import scipy.sparse as sp
    g, rng = tet_case(ja, (9, 8, 6), seed=12)
    lsys, law, disc, osys, nz_o, r_o = assemble_both(ja, ctx, oracle, g, rng, kind, "blocks")
    nc, bs = g["nc"], law.N
    ctx.set_option("fused_product", 1)   # opt-in (measured slower than apply + jagged SpMV, see jh_ilu.hip): read when the factors are laid out
    try:
        F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
        _fused_product_checks(ja, ctx, oracle, kind, F, g, rng, lsys, law, disc, osys, nz_o)
    finally:
        ctx.set_option("fused_product", 0)


def _fused_product_checks(ja, ctx, oracle, kind, F, g, rng, lsys, law, disc, osys, nz_o):
    import scipy.sparse as sp
    nc, bs = g["nc"], law.N
    info = F.info()
    assert info["factor_kernel"] == "pivot-only" and info["fused_product"], info
    perm, bp = disc.ordering()
    p0 = perm - 1
    part = np.zeros(nc, dtype=np.int64)
    part[p0] = np.repeat(np.arange(1, len(bp)), np.diff(bp))
    b = rng.standard_normal(nc * bs)
    x, q = F.apply_mul(lsys.jac.new_vector(), lsys.jac.new_vector(), lsys.jac.new_vector(b))
    x, q = x.download(), q.download()
    # oracle in the device's elimination order (block rows permuted, blocks intact)
    if bs == 1:
        A = sp.csr_matrix((nz_o, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
        Ap = A[p0][:, p0].tocsr()
        Ap.sort_indices()
        Fo = oracle.ILU0(nc, 1, Ap.indptr + 1, Ap.indices + 1, Ap.data, partition=part[p0])
        x_o = np.zeros(nc)
        x_o[p0] = Fo.apply(b[p0])
    else:
        blocks = nz_o.reshape(-1, bs, bs).transpose(0, 2, 1)          # column-major blocks -> [row, col]
        A = sp.bsr_matrix((blocks, osys.colidx - 1, osys.rowptr - 1), shape=(nc * bs, nc * bs)).tocsr()
        pat = sp.csr_matrix((np.arange(1, osys.colidx.size + 1), osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
        Pp = pat[p0][:, p0].tocsr()
        Pp.sort_indices()
        nz_p = nz_o.reshape(-1, bs * bs)[Pp.data - 1].reshape(-1)
        Fo = oracle.ILU0(nc, bs, Pp.indptr + 1, Pp.indices + 1, nz_p, partition=part[p0])
        x_o = np.zeros(nc * bs)
        x_o.reshape(nc, bs)[p0] = Fo.apply(b.reshape(nc, bs)[p0].reshape(-1)).reshape(nc, bs)
    q_o = oracle.spmv(nc, bs, osys.rowptr, osys.colidx, nz_o, x_o)
    assert relerr(x, x_o) < 1e-10 and relerr(q, q_o) < 1e-10
    # the product of the SAME x through the CSR kernel: pins the fused arithmetic itself at SpMV precision
    q_csr = ja.mul_(lsys.jac.new_vector(), lsys.jac, lsys.jac.new_vector(x)).download()
    assert relerr(q, q_csr) < 1e-12
    # standalone apply on the column-scaled factors, factor export
    assert relerr(F.apply(lsys.jac.new_vector(), lsys.jac.new_vector(b)).download(), x_o) < 1e-10
    assert np.count_nonzero(F.factor_values()) > 0
    # BiCGStab through the fused operator: same iterates as the two-operator path (the option is read per solve)
    sol = {}
    for off in (False, True):
        with options(ctx, fused_product=0 if off else 1):
            ks = ja.GenericKrylov("bicgstab", preconditioner=F, relative_tolerance=1e-9, max_iterations=300)
            out = ja.linear_solve(lsys, ks, update_preconditioner=False)
            sol[off] = (out["iterations"], out["residuals"], lsys.dx.download())
    assert abs(sol[False][0] - sol[True][0]) <= 1 and sol[False][0] > 3
    assert relerr(sol[False][2], sol[True][2]) < 1e-7
    k = min(len(sol[False][1]), len(sol[True][1]), 6)
    assert np.allclose(sol[False][1][:k], sol[True][1][:k], rtol=1e-6)


def test_device_blocks_come_from_graph_bisection(ja, ctx):
    """JH_REORDER_BLOCKS: the device blocks (= block-Jacobi ILU(0) partition, the job Metis does for the reference,
    precond/ilu.jl:37-60) are a valid partition into round(nc / block_rows) connected-looking compact blocks, none above the
    size cap (block_rows + 1/8), ghost cells of a rank-local subdomain stay the last device rows, and the cut is the one
    of compact blocks (well below what blocks grown along the rim of the assigned region give: 18 % on this lattice)."""
    g = ja.tet_lattice_mesh(22, 20, 18)
    nc, N = g["nc"], g["N"]
    rows = 256
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", block_rows=rows)
    perm, bp = disc.ordering()
    assert np.array_equal(np.sort(perm), np.arange(1, nc + 1))
    sizes = np.diff(bp)
    assert bp[0] == 0 and bp[-1] == nc and sizes.min() >= 1 and sizes.max() <= rows + rows // 8
    assert len(sizes) == (nc + rows // 2) // rows
    blk = np.zeros(nc, dtype=np.int64)
    blk[perm - 1] = np.repeat(np.arange(len(sizes)), sizes)
    cut = (blk[N[0] - 1] != blk[N[1] - 1]).mean()
    assert cut < 0.165, cut
    # the blocks are connected (bisection of a connected lattice with breadth-first grown sides) up to the few cells the
    # refinement moves across a cut without their neighbours
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    same = blk[N[0] - 1] == blk[N[1] - 1]
    G = sp.coo_matrix((np.ones(int(same.sum())), (N[0][same] - 1, N[1][same] - 1)), shape=(nc, nc))
    ncomp, _ = csg.connected_components(G, directed=False)
    assert ncomp <= len(sizes) + max(2, len(sizes) // 16), (ncomp, len(sizes))
    # rank-local subdomain: owned cells in blocks, ghosts last
    from jutul_amd import dd
    part = dd.partition_rcb(g["cell_centroids"], 2)
    sub = dd.local_subdomain(N, part, 1)
    d2 = ja.TwoPointPotentialFlowHardCoded(ctx, sub["N"], sub["n_local"], reorder="blocks", block_rows=rows, n_owned=sub["n_owned"])
    perm2, bp2 = d2.ordering()
    assert np.array_equal(np.sort(perm2), np.arange(1, sub["n_local"] + 1))
    assert perm2[sub["n_owned"]:].min() > sub["n_owned"] and perm2[: sub["n_owned"]].max() <= sub["n_owned"]


def test_device_blocks_cut_weak_couplings_first(ja, oracle):
    """jh_tpfa_create_weighted: with the face transmissibilities as edge weights the device blocks (= block-Jacobi ILU(0)
    partition) are what the reference's Metis partition of the |A|-weighted graph is (precond/ilu.jl:37-60 -> generate_metis_graph,
    partitioning.jl:64-78): a valid partition under the same size cap whose cut WEIGHT is well below the unweighted one's (the cut
    face COUNT may rise); the host-facing tables do not change; option block_weights = 0 reproduces the unweighted blocks; and a
    BiCGStab solve of the same system takes no more iterations with the weighted blocks."""
    g = ja.tet_lattice_mesh(22, 20, 18)
    nc, N = g["nc"], g["N"]
    T = g["T"] / g["T"].mean()
    assert T.max() / T.min() > 20          # the Kuhn tets' transmissibilities span two decades: there is something to choose
    rows = 256

    def blocks(ctx, **kw):
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", block_rows=rows, **kw)
        perm, bp = disc.ordering()
        assert np.array_equal(np.sort(perm), np.arange(1, nc + 1))
        sizes = np.diff(bp)
        assert sizes.min() >= 1 and sizes.max() <= rows + rows // 8 and len(sizes) == (nc + rows // 2) // rows
        blk = np.zeros(nc, dtype=np.int64)
        blk[perm - 1] = np.repeat(np.arange(len(sizes)), sizes)
        cut = blk[N[0] - 1] != blk[N[1] - 1]
        return disc, perm, bp, cut.mean(), T[cut].sum() / T.sum()

    ctx = ja.HIPContext(0)
    du, pu, bu, cut_u, w_u = blocks(ctx)
    dw, pw, bw, cut_w, w_w = blocks(ctx, face_weights=T)
    assert w_w < 0.75 * w_u, (w_w, w_u)     # (lattice: 0.164 -> 0.099)
    assert abs(w_u - cut_u) < 0.03           # unweighted: the cut takes faces as they come
    for a, b in zip(du.pattern(), dw.pattern()):
        assert np.array_equal(a, b)          # host numbering untouched
    # weights in arbitrary units (SI transmissibilities ~1e-12) give the same blocks: they are scaled to mean 1 inside
    _, ps, bs_, _, _ = blocks(ctx, face_weights=1e-12 * T)
    assert np.array_equal(ps, pw) and np.array_equal(bs_, bw)
    ctx0 = ja.HIPContext(0, block_weights=0)
    _, p0, b0, _, _ = blocks(ctx0, face_weights=T)
    assert np.array_equal(p0, pu) and np.array_equal(b0, bu)
    # the same linear system, both partitions: the weighted blocks need no more iterations (lattice at this size: 16 vs 19)
    its = []
    for disc in (du, dw):
        law = ja.ConservationLaw(disc, "poisson")
        law.set_face_trans(T); law.set_volumes(g["volumes"])
        U = 1.0 + 0.1 * np.sin(np.arange(nc) * 0.001)     # smooth state: a residual the preconditioner has to work for
        law.set_state(U); law.set_state0(U); law.set_sources([1, nc], [1.0, -1.0])
        lsys = ja.LinearizedSystem(disc)
        law.update_equation_and_linearized_system(50.0, lsys.jac, lsys.r)
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-8, max_iterations=300)
        out = ja.linear_solve(lsys, ks)
        assert out["ok"]
        its.append(out["iterations"])
    assert its[1] <= its[0], its


@pytest.mark.parametrize("wave", [None, "0", "1", "2", "3"])
@pytest.mark.parametrize("bs", [1, 2, 3])
@pytest.mark.parametrize("grid", ["bipartite", "bipartite7", "triangles"])
def test_ilu0_factor_kernels_pivot_only_and_program(ja, ctx, oracle, grid, bs, wave):
    """The refactorisation picks its kernel from the pattern: on a triangle-free pattern (Cartesian / tet-lattice grids) no
    elimination step updates an off-diagonal entry and the sweep-shaped pivot-only kernel runs; a pattern with triangles (a
    Cartesian grid with one diagonal per cell pair) needs the general program-driven kernel.  Both give the oracle's ILU(0)
    (ilu0.jl:108-144) on the device-ordered matrix to 1e-11, scalar and 2x2 blocks, after a refactorisation with new values."""
    import scipy.sparse as sp
    # the pivot-only refactorisation has two kernels (one workgroup / one wavefront per block; the default picks by block size and
    # row length): option ilu_factor_kernel = 0 / 1 forces one, 2 / 3 the wavefront kernel with / without its operand prefetch
    if wave is not None and grid == "triangles":
        pytest.skip("the switch only concerns the pivot-only kernels")
    with options(ctx, ilu_factor_kernel=-1 if wave is None else int(wave)):
        _factor_kernel_case(ja, ctx, oracle, grid, bs, wave)


def _factor_kernel_case(ja, ctx, oracle, grid, bs, wave):
    import scipy.sparse as sp
    dims = (13, 11, 1)
    N = ja.cartesian_neighbors(dims)
    nc = int(np.prod(dims))
    if grid == "triangles":
        idx = np.arange(1, nc + 1).reshape(dims[1], dims[0])
        diag = np.stack([idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()])
        N = np.concatenate([N, diag], axis=1)
    elif grid == "bipartite7":   # degree-7 bipartite graph: up to 7 strict-lower / strict-upper entries in a row (8-diagonal kernels)
        h = 600
        nc = 2 * h
        a = np.arange(h)
        N = np.concatenate([np.stack([a + 1, (a + o) % h + h + 1]) for o in (0, 1, 5, 17, 40, 111, 290)], axis=1)
    rng = np.random.default_rng(31 + bs)
    # forced kernels: blocks of several 64-row chunks (the chunk pipeline of the wavefront kernel, several waves in the other)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, block_n=bs, reorder="blocks", block_rows=256 if (wave is not None and grid == "bipartite7") else 64)
    A = ja.StaticSparsityMatrixCSR(disc)
    rowptr, colidx = disc.pattern()
    perm, bp = disc.ordering()
    part = np.zeros(nc, dtype=np.int64)
    for b in range(len(bp) - 1):
        part[perm[bp[b]: bp[b + 1]] - 1] = b + 1
    F = ja.ILUZeroPreconditioner(partition="blocks")
    for trial in range(2):
        nzb = rng.standard_normal((A.nnzb, bs, bs)) * 0.2
        rows = np.repeat(np.arange(nc), np.diff(rowptr))
        dg = colidx - 1 == rows
        nzb[dg] += 4.0 * np.eye(bs)
        A.nzval = nzb.transpose(0, 2, 1).reshape(-1)   # blocks column-major in the flat buffer
        F.update_preconditioner(A)
        info = F.info()
        assert info["jagged"] and info["factor_kernel"] == ("program" if grid == "triangles" else "pivot-only"), info
        # the oracle in the device's elimination order: factor VALUES entry by entry (L multipliers, inverted pivots, U) and the
        # triangular solve as a componentwise backward error with the oracle's factors (tests/_ilu_checks.py)
        from tests import _ilu_checks as ck
        nz_flat = nzb.transpose(0, 2, 1).reshape(-1)
        rp_p, ci_p, nz_p, part_p, slot = ck.device_order_problem(nc, bs, rowptr, colidx, nz_flat, perm, bp)
        Fo, lu_o, Lm, Um = ck.oracle_factors(oracle, nc, bs, rp_p, ci_p, nz_p, part_p)
        ck.check_factor_values(F.factor_values().reshape(-1, bs * bs)[slot].reshape(-1), lu_o, Lm, Um, nc, bs, rp_p, ci_p)
        p0 = perm - 1
        pe = (p0[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        b = rng.standard_normal(nc * bs)
        x = F.apply(A.new_vector(), A.new_vector(b)).download()
        ck.check_triangular_solve(x[pe], b[pe], Lm, Um, levels=info["max_levels"])
        assert relerr(x[pe], Fo.apply(b[pe])) < 1e-10, (grid, bs, trial)


# ---- a-14: BiCGStab ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("side", ["right", "left"])
@pytest.mark.parametrize("bs", [1, 2])
def test_bicgstab_parity(ja, ctx, oracle, side, bs):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (12, 10, 8), bs, seed=9)
    b = rng.standard_normal(nc * bs)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=bs, rowptr=rowptr, colidx=colidx, nzval=nz)

    class Sys:  # minimal LinearizedSystem over a raw matrix
        pass
    s = Sys()
    s.disc = type("D", (), {"ctx": ctx})()
    s.jac, s.r, s.dx, s._x = A, A.new_vector(b), A.new_vector(), A.new_vector()
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(), relative_tolerance=1e-9,
                          absolute_tolerance=1e-14, max_iterations=100, precond_side=side)
    out = ja.linear_solve(s, ks)
    Fo = oracle.ILU0(nc, bs, rowptr, colidx, nz)
    xo, st = oracle.bicgstab(nc, bs, rowptr, colidx, nz, b, prec=Fo, side=side, rtol=1e-9, atol=1e-14, itmax=100)
    assert out["ok"] and st["solved"]
    assert abs(out["iterations"] - st["iterations"]) <= 1
    dx = s.dx.download()
    assert relerr(-dx, xo) < 1e-7  # dx = -x
    assert len(out["residuals"]) == out["iterations"] + 1
    assert np.allclose(out["residuals"][0], st["residuals"][0], rtol=1e-10)
    # one ILU(0) block, the oracle's elimination order: not only the converged answer but the residual HISTORY agrees while the
    # residual is far above the rounding floor
    m = min(11, len(out["residuals"]), len(st["residuals"]))
    assert m >= 3 and np.allclose(out["residuals"][:m], st["residuals"][:m], rtol=1e-7), (out["residuals"][:m], st["residuals"][:m])
    r = oracle.spmv(nc, bs, rowptr, colidx, nz, -dx) - b
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(b)


def test_bicgstab_itmax_and_history(ja, ctx, oracle):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (10, 10, 10), 1, seed=10)
    b = rng.standard_normal(nc)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=1, rowptr=rowptr, colidx=colidx, nzval=nz)
    s = type("S", (), {})()
    s.disc = type("D", (), {"ctx": ctx})()
    s.jac, s.r, s.dx, s._x = A, A.new_vector(b), A.new_vector(), A.new_vector()
    ks = ja.GenericKrylov("bicgstab", preconditioner=None, relative_tolerance=1e-14, max_iterations=3)
    out = ja.linear_solve(s, ks)
    assert not out["ok"] and out["iterations"] == 3 and out["status"] == 1 and len(out["residuals"]) == 4
    assert out["residuals"][-1] < out["residuals"][0]


# ---- Newton loop / KATs ----------------------------------------------------------------------------------------------------------
def test_poisson_3x1_known_answer_on_gpu(ja, ctx, oracle, golden):
    """test/test_systems/variable_poisson.jl:5-39 through the HIP path: U - U[1] == [0, 1/3, 2/3]."""
    geo = oracle.cartesian_geometry((3, 1), (1.0, 1.0))
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, geo["N"], 3)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(np.full(2, 3.0))
    law.set_state(np.ones(3))
    law.set_state0(np.ones(3))
    law.set_sources([1, 3], [1.0, -1.0])
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(),
                                             relative_tolerance=1e-12, max_iterations=50), tolerance=1e-8)
    ok, its, rep = sim.solve_ministep(0.0)  # stationary variant
    assert ok
    U = law.get_state()
    assert np.allclose(U - U[0], golden["kat"]["poisson_3x1"], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("kind", ["compressible", "twophase"])
def test_nonlinear_newton_matches_oracle_newton(ja, ctx, oracle, kind):
    """Full Newton loop on the GPU vs an oracle Newton loop with direct solves: same converged state."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    g, rng = tet_case(ja, (5, 4, 3), seed=11)
    nc = g["nc"]
    nblk = 2 if kind == "twophase" else 1
    par = dict(rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=nblk, reorder="blocks", block_rows=64)
    law = ja.ConservationLaw(disc, kind, **par)
    if nblk == 1:
        X0 = rng.uniform(1.0, 2.0, nc)
    else:
        X0 = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)
    law.set_face_trans(g["Tn"])
    law.set_volumes(g["volumes"])
    law.set_face_gdz(g["gdz"])
    law.set_state(X0)
    law.set_state0(X0)
    dt = 0.05
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                                             relative_tolerance=1e-10, max_iterations=200), tolerance=1e-9)
    ok, its, rep = sim.solve_ministep(dt)
    assert ok
    X_gpu = law.get_state()
    osys = oracle.TPFASystem(g["N"], nc, nblk)
    olaw = oracle.Law(kind, dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    X = X0.copy()
    for it in range(30):
        nz, r = osys.assemble(olaw, X, X0, g["volumes"], g["Tn"], g["gdz"])
        if np.abs(r).max() < 1e-9 and it > 0:
            break
        bsr = sp.bsr_matrix((nz.reshape(-1, nblk, nblk).transpose(0, 2, 1), osys.colidx - 1, osys.rowptr - 1),
                            shape=(nc * nblk, nc * nblk))
        X = X + spl.spsolve(bsr.tocsc(), -r)
    assert relerr(X_gpu, X) < 1e-8
    assert its <= 10


def test_timestep_cut_and_state_restore(ja, ctx, oracle):
    g, rng = tet_case(ja, (3, 3, 2), seed=12)
    nc = g["nc"]
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(g["Tn"])
    X0 = rng.uniform(1, 2, nc)
    law.set_state(X0 + 1.0)
    law.set_state0(X0)
    law.reset_state()
    assert np.array_equal(law.get_state(), X0)
    law.set_state(X0 + 1.0)
    law.update_state0()
    law.set_state(X0)
    law.reset_state()
    assert np.array_equal(law.get_state(), X0 + 1.0)


def test_update_primary_limits(ja, ctx, oracle):
    """choose_increment chain (variables/utils.jl:146-174): scale -> abs -> rel -> lower -> upper."""
    g, rng = tet_case(ja, (2, 2, 2), seed=13)
    nc = g["nc"]
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc)
    law = ja.ConservationLaw(disc, "poisson")
    v = rng.uniform(0.5, 2.0, nc)
    dx = rng.standard_normal(nc) * 3
    law.set_state(v)
    lim = np.array([2.0, 1.5, 0.5, 0.4, 2.2])  # scale, abs_max, rel_max, min, max
    law.update_primary_variables(ja.DeviceVector(disc, dx), w=0.5, limits=lim)
    dv = 0.5 * dx * 2.0
    dv = np.sign(dv) * np.minimum(np.abs(dv), 1.5)
    dv = np.sign(dv) * np.minimum(np.abs(dv), 0.5 * np.abs(v))
    dv = np.maximum(dv, 0.4 - v)
    dv = np.minimum(dv, 2.2 - v)
    assert np.allclose(law.get_state(), v + dv, rtol=1e-15)
    nan = np.nan
    law.set_state(v)
    law.update_primary_variables(ja.DeviceVector(disc, dx), w=1.0, limits=np.array([nan, nan, nan, nan, nan]))
    assert np.allclose(law.get_state(), v + dx, rtol=1e-15)


def test_newton_step_applies_the_laws_update_limits(ja, ctx, oracle):
    """Simulator.perform_step (jh_newton_step) applies the limits stored on the law (jh_law_set_update_limits: the variables'
    minimum / maximum / increment limits of update_primary_variables!, variables/utils.jl:110-174) -- both on the fused path and
    on the general one; without limits the plain Newton update."""
    g, rng = tet_case(ja, (5, 4, 4), seed=21)
    nc = g["nc"]
    nan = np.nan
    lim = np.array([[nan, 0.004, nan, nan, nan], [nan, 0.05, nan, 0.0, 1.0]])
    out = {}
    for mode in ("none", "fused", "general"):
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=2, reorder="blocks")
        law = ja.ConservationLaw(disc, "twophase", rho0=(1.0, 0.8), compressibility=(1e-3, 2e-3), viscosity=(1.0, 2.0), p_ref=1.0)
        law.set_face_trans(g["Tn"]); law.set_volumes(g["volumes"])
        U0 = np.stack([1.0 + 0.05 * rng.random(nc), rng.uniform(0.02, 0.98, nc)]).T.reshape(-1) if "U0" not in out else out["U0"]
        out["U0"] = U0
        law.set_state(U0); law.set_state0(U0)
        law.set_sources([1, nc], [0.5, 0.5, -0.5, -0.5])
        if mode != "none":
            law.set_update_limits(lim)
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-10,
                              max_iterations=200, **({"true_residual": True} if mode == "general" else {}))
        sim = ja.Simulator(law, ks)
        assert sim._needs_general_path() == (mode == "general")
        sim.perform_step(0.5, 1)
        out[mode] = law.get_state().reshape(nc, 2)
        out[mode + "_dx"] = sim.lsys.dx.download().reshape(nc, 2)
    V = out["U0"].reshape(nc, 2)
    assert np.allclose(out["none"], V + out["none_dx"], rtol=1e-13, atol=1e-14)
    assert np.abs(out["none_dx"][:, 1]).max() > 0.05          # the limits below do bind
    for mode in ("fused", "general"):
        d = out[mode + "_dx"].copy()
        d[:, 0] = np.sign(d[:, 0]) * np.minimum(np.abs(d[:, 0]), 0.004)
        d[:, 1] = np.sign(d[:, 1]) * np.minimum(np.abs(d[:, 1]), 0.05)
        d[:, 1] = np.minimum(np.maximum(d[:, 1], 0.0 - V[:, 1]), 1.0 - V[:, 1])
        assert np.allclose(out[mode], V + d, rtol=1e-13, atol=1e-14), mode
        assert out[mode][:, 1].min() >= 0.0 and out[mode][:, 1].max() <= 1.0
    assert np.allclose(out["fused"], out["general"], rtol=1e-6, atol=1e-8)


# ---- size-independent properties at a larger size ---------------------------------------------------------------------------------
def test_medium_tet_mesh_end_to_end_properties(ja, ctx, oracle):
    """~100k cells: (i) J*1 == vol/dt row sums, (ii) conservation: sum(r) == sum(acc), (iii) block-Jacobi ILU
    BiCGStab reaches rtol, true residual checked with the GPU SpMV, (iv) linearity: one Newton step solves it."""
    g, rng = tet_case(ja, (28, 25, 24), seed=14)
    nc = g["nc"]
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=2048)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(g["Tn"])
    law.set_volumes(g["volumes"])
    U0 = 1.0 + 0.1 * rng.random(nc)
    law.set_state(U0)
    law.set_state0(np.zeros(nc))
    dt = 2.0
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    ones = ja.DeviceVector(disc, np.ones(nc))
    y = ja.mul_(ja.DeviceVector(disc), lsys.jac, ones).download()
    assert relerr(y, g["volumes"] / dt) < 1e-9
    r = lsys.r.download()
    assert np.isclose(r.sum(), (g["volumes"] * U0 / dt).sum(), rtol=1e-10)
    law.set_state0(U0)
    law.set_sources([1, nc], [1.0, -1.0])
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                                             relative_tolerance=1e-10, max_iterations=300), tolerance=1e-6)
    rep = sim.perform_step(dt, 1)
    assert rep.linear_status == 0 and rep.lin_res <= 1e-10 * rep.lin_res0 * 1.0001 + 1e-12
    rep2 = sim.perform_step(dt, 2)
    assert rep2.converged == 1 and max(rep2.error[:1]) < 1e-6
    # cross-check the solution against the oracle's BiCGStab on the same system (host ordering, serial ILU)
    osys = oracle.TPFASystem(g["N"], nc)
    nz_o, r_o = osys.assemble(oracle.Law("poisson", dt), U0, U0, g["volumes"], g["Tn"], src_cells=[1, nc],
                              src_values=[1.0, -1.0])
    Fo = oracle.ILU0(nc, 1, osys.rowptr, osys.colidx, nz_o)
    xo, st = oracle.bicgstab(nc, 1, osys.rowptr, osys.colidx, nz_o, r_o, prec=Fo, rtol=1e-10, atol=1e-30, itmax=300)
    assert st["solved"]
    assert relerr(law.get_state() - U0, -xo) < 1e-5


# ---- scalar matrix layouts at the boundary (a-3/a-4) ------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["equation_major", "entity_major"])
def test_scalar_layouts_pattern_positions_values(ja, ctx, oracle, layout):
    """EquationMajor / EntityMajor (equations.jl:132-138, models.jl:585-611): pattern + jacobian positions bit-exact
    vs the oracle, Jacobian values and residual delivered in that layout == the oracle's fill through ITS positions."""
    g, rng = tet_case(ja, (5, 4, 3), seed=21)
    nc, N = g["nc"], 2
    lay = oracle.LAYOUT[layout]
    lsys, law, disc, osys, nz_blk, r_blk = assemble_both(ja, ctx, oracle, g, rng, "twophase", "blocks")
    srp, sci = oracle.csr_pattern_scalar(nc, N, lay, osys.rowptr, osys.colidx)
    rp, ci = disc.pattern_layout(layout)
    assert np.array_equal(rp, srp) and np.array_equal(ci, sci)
    opa, opf = oracle.align(nc, N, lay, osys.rowptr, osys.colidx, osys.hfm, srp, sci)
    pa, pf = disc.jacobian_positions_layout(layout)
    assert np.array_equal(pa, opa) and np.array_equal(pf, opf)
    # values: scatter the block-major oracle Jacobian through both position tables and compare slot by slot
    nz_scalar = np.zeros(nz_blk.size)
    bpa, bpf = osys.pos_acc, osys.pos_flux
    nz_scalar[opa.reshape(-1) - 1] = nz_blk[bpa.reshape(-1) - 1]
    nz_scalar[opf.reshape(-1) - 1] = nz_blk[bpf.reshape(-1) - 1]
    assert relerr(lsys.jac.nzval_layout(layout), nz_scalar) < RTOL
    r_dev = lsys.r.download_layout(layout)
    r_ref = r_blk.reshape(nc, N).T.reshape(-1) if layout == "equation_major" else r_blk
    assert relerr(r_dev, r_ref) < RTOL
    # round trip: set values / vectors in the layout, read back block-major
    A2 = ja.StaticSparsityMatrixCSR(disc)
    A2.set_nzval_layout(nz_scalar, layout)
    assert relerr(A2.nzval, nz_blk) < 1e-15
    v = ja.DeviceVector(disc).upload_layout(r_ref, layout)
    assert np.array_equal(v.download(), r_blk)
    # block layout passes through
    assert np.array_equal(disc.pattern_layout("block_major")[1], osys.colidx)


# ---- DiagonalPreconditioner family ------------------------------------------------------------------------------------
@pytest.mark.parametrize("bs", [1, 2])
def test_jacobi_and_spai0_preconditioners(ja, ctx, oracle, bs):
    """JacobiPreconditioner w*inv(A_ii) (precond/jacobi.jl:14-17) and SPAI(0) A_ii/sum(v^2) (precond/spai.jl:40-60) vs numpy;
    both usable as BiCGStab preconditioners."""
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (7, 6, 5), bs, seed=31)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=bs, rowptr=rowptr, colidx=colidx, nzval=nz)
    blocks = nz.reshape(-1, bs, bs).transpose(0, 2, 1)  # [nnzb, e, d]
    rows = np.repeat(np.arange(nc), np.diff(rowptr))
    dmask = colidx - 1 == rows
    Aii = blocks[dmask]
    y = rng.standard_normal(nc * bs)
    w = 2.0 / 3.0
    xj = ja.JacobiPreconditioner(w).update_preconditioner(A).apply(A.new_vector(), A.new_vector(y)).download()
    ref = np.einsum("ied,id->ie", w * np.linalg.inv(Aii), y.reshape(nc, bs)).reshape(-1)
    assert relerr(xj, ref) < 1e-13
    ns = np.zeros(nc)
    np.add.at(ns, rows, (blocks ** 2).sum((1, 2)))
    xs = ja.SPAI0Preconditioner().update_preconditioner(A).apply(A.new_vector(), A.new_vector(y)).download()
    ref = np.einsum("ied,id->ie", Aii / ns[:, None, None], y.reshape(nc, bs)).reshape(-1)
    assert relerr(xs, ref) < 1e-13
    b = rng.standard_normal(nc * bs)
    for prec in (ja.JacobiPreconditioner(), ja.SPAI0Preconditioner()):
        s = type("S", (), {})()
        s.disc = type("D", (), {"ctx": ctx})()
        s.jac, s.r, s.dx, s._x = A, A.new_vector(b), A.new_vector(), A.new_vector()
        out = ja.linear_solve(s, ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-9, max_iterations=200))
        assert out["ok"]
        r = oracle.spmv(nc, bs, rowptr, colidx, nz, -s.dx.download()) - b
        assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(b)


@pytest.mark.parametrize("bs", [1, 2])
def test_system_scaling(ja, ctx, oracle, bs):
    """:diagonal and :dt scaling of the linearized system (linsolve/default.jl:325-385): J <- diag(F) J, r <- F.*r."""
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (6, 5, 4), bs, seed=41)
    rows = np.repeat(np.arange(nc), np.diff(rowptr))
    blocks = nz.reshape(-1, bs, bs).transpose(0, 2, 1)  # [nnzb, e, d]
    r = rng.standard_normal(nc * bs)
    for scaling, dt in (("diagonal", 1.0), ("dt", 3.5)):
        A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=bs, rowptr=rowptr, colidx=colidx, nzval=nz)
        s = type("S", (), {})()
        s.jac, s.r = A, A.new_vector(r)
        ja.scale_system(s, scaling, dt)
        if scaling == "dt":
            F = np.full((nc, bs), dt)
        else:
            Aii = blocks[colidx - 1 == rows]
            F = 1.0 / np.abs(np.einsum("iee->ie", Aii))
        ref_blocks = blocks * F[rows][:, :, None]
        assert relerr(A.nzval, ref_blocks.transpose(0, 2, 1).reshape(-1)) < 1e-15
        assert relerr(s.r.download(), (r.reshape(nc, bs) * F).reshape(-1)) < 1e-15


# ---- edge cases -----------------------------------------------------------------------------------------------------------------------
def test_single_cell_no_faces_scalar_kat(ja, ctx):
    """ScalarTestSystem analogue (test/test_systems/scalar.jl:12-40): one cell, no faces, dX/dt = 1 -> X = 1 after dt = 1 and
    0.5 after a half step (empty neighborship: get_facepos branch utils.jl:862-866, flux.jl:185-188)."""
    N = np.zeros((2, 0), dtype=np.int64)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, 1)
    assert disc.nnzb == 1 and disc.nhf == 0
    assert list(disc.conn["face_pos"]) == [1, 1]
    law = ja.ConservationLaw(disc, "poisson")
    law.set_state([0.0])
    law.set_state0([0.0])
    law.set_sources([1], [-1.0])  # r = (X - X0)/dt - 1
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(), relative_tolerance=1e-12))
    assert sim.solve_timestep(0.5) == 2
    assert np.isclose(law.get_state()[0], 0.5, rtol=1e-14)
    assert sim.solve_timestep(0.5) == 2
    assert np.isclose(law.get_state()[0], 1.0, rtol=1e-14)


def test_ragged_grid_with_isolated_cells(ja, ctx, oracle):
    """Cells without any face (isolated) mixed with connected ones: empty rows segments, diagonal-only CSR rows."""
    N = np.array([[1, 2, 5], [2, 3, 6]], dtype=np.int64)  # cells 4, 7, 8 are isolated
    nc = 8
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", block_rows=3)
    h = oracle.half_face_map(N, nc)
    assert np.array_equal(disc.conn["face_pos"], h["face_pos"]) and np.array_equal(disc.conn["other"], h["other"])
    rp, ci = disc.pattern()
    orp, oci = oracle.csr_pattern(nc, h)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
    rng = np.random.default_rng(5)
    law = ja.ConservationLaw(disc, "poisson")
    T, U, U0 = rng.uniform(1, 2, 3), rng.standard_normal(nc), rng.standard_normal(nc)
    law.set_face_trans(T)
    law.set_state(U)
    law.set_state0(U0)
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(0.3, lsys.jac, lsys.r)
    osys = oracle.TPFASystem(N, nc)
    nz_o, r_o = osys.assemble(oracle.Law("poisson", 0.3), U, U0, np.ones(nc), T)
    assert relerr(lsys.r.download(), r_o) < RTOL and relerr(lsys.jac.nzval, nz_o) < RTOL
    F = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
    x = F.apply(lsys.jac.new_vector(), lsys.r).download()
    assert np.all(np.isfinite(x))


def test_argument_validation(ja, ctx, oracle):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (4, 3), 1, seed=51)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=1, rowptr=rowptr, colidx=colidx, nzval=nz)
    with pytest.raises(ja.JutulHIPError):  # partition ids must be >= 1 (par_ilu0.jl:49)
        ja.ilu0_csr(A, np.zeros(nc, dtype=np.int64))
    with pytest.raises(ja.JutulHIPError):  # empty block (partitioning.jl:47)
        p = np.ones(nc, dtype=np.int64)
        p[0] = 3
        ja.ilu0_csr(A, p)
    with pytest.raises(ja.JutulHIPError):  # unsorted columns
        ja.StaticSparsityMatrixCSR(context=ctx, n=2, bs=1, rowptr=[1, 3, 4], colidx=[2, 1, 2], nzval=[1.0, 2.0, 3.0])
    g, _ = tet_case(ja, (2, 2, 2))
    disc1 = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"])
    with pytest.raises(ja.JutulHIPError):  # two-phase law needs block_n = 2
        ja.ConservationLaw(disc1, "twophase")
    law = ja.ConservationLaw(disc1, "compressible")
    lsys = ja.LinearizedSystem(disc1)
    with pytest.raises(ja.JutulHIPError):  # dt must be positive for the compressible law
        law.update_equation_and_linearized_system(0.0, lsys.jac, lsys.r)
    F = ja.ILUZeroPreconditioner()
    with pytest.raises(Exception):  # apply before update_preconditioner!
        F.apply(lsys.dx, lsys.r)


def test_bitwise_run_to_run_determinism(ja, ctx, oracle):
    """Owner-computes assembly (no atomics) + ordered two-stage reductions: two identical Newton steps give identical bits
    (the reference's thread-safety-by-construction, SURVEY section 5 'race detection')."""
    g, rng = tet_case(ja, (12, 10, 9), seed=61)
    nc = g["nc"]
    outs = []
    for _ in range(2):
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=256)
        law = ja.ConservationLaw(disc, "compressible", rho0=(1.0, 1.0), compressibility=(1e-2, 0.0), viscosity=(1.0, 1.0), p_ref=1.0)
        law.set_face_trans(g["Tn"])
        law.set_volumes(g["volumes"])
        law.set_face_gdz(g["gdz"])
        X0 = 1.0 + 0.1 * np.random.default_rng(7).random(nc)
        law.set_state(X0)
        law.set_state0(X0)
        law.set_sources([1, nc], [0.3, -0.3])
        sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                                                 relative_tolerance=1e-8, max_iterations=200), tolerance=1e-8)
        ok, its, rep = sim.solve_ministep(0.4)
        assert ok
        outs.append((law.get_state(), sim.lsys.jac.nzval, sim.lsys.r.download(), its, rep.linear_iterations))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2]) and outs[0][3:] == outs[1][3:]


# ---- randomized unstructured graphs (not mesh-like) -----------------------------------------------------------------------------------
def random_graph(rng, nc, avg_deg):
    """Random connected simple graph as a neighborship: a random spanning tree plus random extra edges."""
    edges = set()
    order = rng.permutation(nc)
    for i in range(1, nc):
        a, b = int(order[i]), int(order[rng.integers(0, i)])
        edges.add((min(a, b), max(a, b)))
    while len(edges) < nc * avg_deg // 2:
        a, b = (int(v) for v in rng.integers(0, nc, 2))
        if a != b:
            edges.add((min(a, b), max(a, b)))
    E = np.array(sorted(edges), dtype=np.int64)
    flip = rng.random(len(E)) < 0.5  # random left/right orientation
    E[flip] = E[flip][:, ::-1]
    return (E[rng.permutation(len(E))].T + 1).copy()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_graphs_full_chain_parity(ja, ctx, oracle, seed):
    """Irregular degrees (1 .. ~25), random orientation and face order: tables bit-exact, assembly / SpMV / ILU(0) /
    BiCGStab within the fp64 tolerances, for scalar and 2x2-block laws, both device orderings."""
    rng = np.random.default_rng(100 + seed)
    nc = int(rng.integers(300, 1500))
    N = random_graph(rng, nc, avg_deg=int(rng.integers(3, 9)))
    nf = N.shape[1]
    kind = ["poisson", "compressible", "twophase", "twophase"][seed]
    nblk = 2 if kind == "twophase" else 1
    reorder = "blocks" if seed % 2 else "none"
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, block_n=nblk, reorder=reorder, block_rows=int(rng.integers(16, 200)))
    osys = oracle.TPFASystem(N, nc, nblk)
    c = disc.conn
    assert np.array_equal(c["face_pos"], osys.hfm["face_pos"]) and np.array_equal(c["face"], osys.hfm["faces"])
    assert np.array_equal(c["other"], osys.hfm["other"]) and np.array_equal(c["face_sign"], osys.hfm["face_sign"])
    rp, ci = disc.pattern()
    assert np.array_equal(rp, osys.rowptr) and np.array_equal(ci, osys.colidx)
    pa, pf = disc.jacobian_positions()
    assert np.array_equal(pa, osys.pos_acc) and np.array_equal(pf, osys.pos_flux)
    par = dict(rho0=(1.0, 0.7), compressibility=(2e-2, 1e-2), viscosity=(1.0, 3.0), p_ref=1.0)
    law = ja.ConservationLaw(disc, kind, **par)
    T, vol, gdz = rng.uniform(0.2, 3.0, nf), rng.uniform(0.5, 2.0, nc), rng.standard_normal(nf) * 0.05
    if nblk == 1:
        X, X0 = rng.uniform(1, 2, nc), rng.uniform(1, 2, nc)
    else:
        X = np.stack([rng.uniform(1, 2, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
        X0 = np.stack([rng.uniform(1, 2, nc), rng.uniform(0.2, 0.8, nc)]).T.reshape(-1)
    law.set_face_trans(T)
    law.set_volumes(vol)
    if kind != "poisson":
        law.set_face_gdz(gdz)
    law.set_state(X)
    law.set_state0(X0)
    dt = 0.37
    lsys = ja.LinearizedSystem(disc)
    law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    olaw = oracle.Law(kind, dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    nz_o, r_o = osys.assemble(olaw, X, X0, vol, T, gdz if kind != "poisson" else None)
    assert relerr(lsys.r.download(), r_o) < RTOL and relerr(lsys.jac.nzval, nz_o) < RTOL
    x = rng.standard_normal(nc * nblk)
    y = ja.mul_(ja.DeviceVector(disc), lsys.jac, ja.DeviceVector(disc, x), 1.5, 0.0).download()
    assert relerr(y, oracle.spmv(nc, nblk, osys.rowptr, osys.colidx, nz_o, x, alpha=1.5)) < RTOL
    if reorder == "none":  # same elimination order as the reference: factors and solves comparable entry by entry
        F = ja.ilu0_csr(lsys.jac)
        Fo = oracle.ILU0(nc, nblk, osys.rowptr, osys.colidx, nz_o)
        lu_o = Fo.export(nz_o.size)
        assert relerr(F.factor_values(), lu_o) < 1e-10
        b = rng.standard_normal(nc * nblk)
        assert relerr(F.apply(lsys.jac.new_vector(), lsys.jac.new_vector(b)).download(), Fo.apply(b)) < 1e-9
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(), relative_tolerance=1e-9, max_iterations=300)
        out = ja.linear_solve(lsys, ks)
        xo, st = oracle.bicgstab(nc, nblk, osys.rowptr, osys.colidx, nz_o, r_o, prec=Fo, rtol=1e-9, atol=1e-12, itmax=300)
        assert out["ok"] and st["solved"] and abs(out["iterations"] - st["iterations"]) <= 1
        assert relerr(-lsys.dx.download(), xo) < 1e-6
    else:
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-9,
                              max_iterations=400)
        out = ja.linear_solve(lsys, ks)
        assert out["ok"]
        res = oracle.spmv(nc, nblk, osys.rowptr, osys.colidx, nz_o, -lsys.dx.download()) - r_o
        assert np.linalg.norm(res) <= 1e-7 * np.linalg.norm(r_o)


def test_time_dependent_poisson_2x2_schedule(ja, ctx, oracle):
    """test/test_systems/variable_poisson.jl:71-115: time-dependent Poisson on a 2x2 mesh, dt = [0.1, 0.9, 10, 100],
    K = face trans of k = 1, sources +-1: the report steps are taken with the Newton / ministep control of SURVEY A.8 and the
    states equal backward-Euler steps computed with dense solves."""
    geo = oracle.cartesian_geometry((2, 2), (1.0, 1.0))
    nc = 4
    h = oracle.half_face_map(geo["N"], nc)
    Tf = oracle.face_trans(oracle.half_face_trans(geo, np.ones((1, nc)), h), h["faces"], geo["nf"])
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, geo["N"], nc)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(Tf)
    U = np.ones(nc)
    law.set_state(U)
    law.set_state0(U)
    law.set_sources([1, nc], [1.0, -1.0])
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(), relative_tolerance=1e-12))
    dts = [0.1, 0.9, 10.0, 100.0]
    its = sim.simulate(dts)
    assert its == [2, 2, 2, 2]  # linear law: 2 assemblies + 1 solve per ministep, no cuts
    # dense backward Euler: (I/dt + L) U1 = U0/dt - src
    L = np.zeros((nc, nc))
    for f in range(geo["nf"]):
        a, b = geo["N"][0, f] - 1, geo["N"][1, f] - 1
        L[a, a] += Tf[f]; L[b, b] += Tf[f]; L[a, b] -= Tf[f]; L[b, a] -= Tf[f]
    src = np.array([1.0, 0, 0, -1.0])
    for dt in dts:
        U = np.linalg.solve(np.eye(nc) / dt + L, U / dt - src)
    assert np.allclose(law.get_state(), U, rtol=1e-9, atol=1e-9)
    assert np.isclose(sum(dts), 111.0)


def test_krylov_tolerance_options(ja, ctx, oracle):
    """true_residual and the Newton-history relaxed tolerance of linear_solve! (linsolve/krylov.jl:96-118) change only the
    (atol, rtol) handed to BiCGStab: check them through the residual the solve stops at."""
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (10, 9, 8), 1, seed=71)
    b = rng.standard_normal(nc)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=1, rowptr=rowptr, colidx=colidx, nzval=nz)

    def solve(sub=None, **cfg):
        s = type("S", (), {})()
        s.disc = type("D", (), {"ctx": ctx})()
        s.jac, s.r, s.dx, s._x = A, A.new_vector(b), A.new_vector(), A.new_vector()
        ks = solve.ks if sub not in (None, 1) else ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(), **cfg)
        solve.ks = ks
        return ja.linear_solve(s, ks, subiteration=sub), ks

    base, _ = solve(relative_tolerance=1e-6, precond_side="left")
    tr, _ = solve(relative_tolerance=1e-6, precond_side="left", true_residual=True)
    # true_residual: stop on ||r_prec|| <= atol + rtol*||b|| (unpreconditioned norm of b) instead of a relative reduction
    assert tr["residuals"][-1] <= 1e-12 + 1e-6 * np.linalg.norm(b) * 1.0001
    assert tr["iterations"] <= base["iterations"] + 1
    first, ks = solve(sub=1, relative_tolerance=1e-8, nonlinear_relative_tolerance=1e-2, relaxed_relative_tolerance=0.1)
    assert ks.r_norm is not None and np.isclose(ks.r_norm, np.linalg.norm(b))
    second, _ = solve(sub=2)  # same ||r||: rtol = max(min(r0*1e-2/r_k, 0.1), 1e-8) = 1e-2
    assert second["iterations"] < first["iterations"]
    assert second["residuals"][-1] <= 1e-12 + 1e-2 * second["residuals"][0] * 1.0001


# ---- GMRES -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("side", ["right", "left"])
@pytest.mark.parametrize("bs", [1, 2])
def test_gmres_parity(ja, ctx, oracle, side, bs):
    nc, rowptr, colidx, nz, rng = random_csr(oracle, (11, 9, 8), bs, seed=81)
    b = rng.standard_normal(nc * bs)
    A = ja.StaticSparsityMatrixCSR(context=ctx, n=nc, bs=bs, rowptr=rowptr, colidx=colidx, nzval=nz)
    s = type("S", (), {})()
    s.disc = type("D", (), {"ctx": ctx})()
    s.jac, s.r, s.dx, s._x = A, A.new_vector(b), A.new_vector(), A.new_vector()
    ks = ja.GenericKrylov("gmres", preconditioner=ja.ILUZeroPreconditioner(), relative_tolerance=1e-9, absolute_tolerance=1e-14,
                          max_iterations=80, precond_side=side)
    out = ja.linear_solve(s, ks)
    Fo = oracle.ILU0(nc, bs, rowptr, colidx, nz)
    xo, st = oracle.gmres(nc, bs, rowptr, colidx, nz, b, prec=Fo, side=side, rtol=1e-9, atol=1e-14, itmax=80)
    assert out["ok"] and st["solved"] and abs(out["iterations"] - st["iterations"]) <= 1
    assert relerr(-s.dx.download(), xo) < 1e-7
    n = min(len(out["residuals"]), len(st["residuals"]))
    assert np.allclose(out["residuals"][: n - 1], st["residuals"][: n - 1], rtol=1e-6, atol=1e-12 * st["residuals"][0])
    r = oracle.spmv(nc, bs, rowptr, colidx, nz, -s.dx.download()) - b
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(b)
    # itmax reached
    ks2 = ja.GenericKrylov("gmres", preconditioner=None, relative_tolerance=1e-14, max_iterations=4)
    out2 = ja.linear_solve(s, ks2)
    assert not out2["ok"] and out2["iterations"] == 4 and out2["status"] == 1


def test_no_device_memory_leak_over_handle_lifetimes(ja, ctx, oracle):
    """Handles own their device memory: repeated create / use / destroy cycles do not grow the device footprint."""
    import gc
    import torch
    g, rng = tet_case(ja, (14, 12, 10), seed=91)
    nc = g["nc"]

    def cycle():
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=256)
        law = ja.ConservationLaw(disc, "poisson")
        law.set_face_trans(g["Tn"])
        law.set_state(np.ones(nc))
        law.set_state0(np.ones(nc))
        law.set_sources([1, nc], [1.0, -1.0])
        sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks")))
        sim.solve_timestep(1.0)
        for obj in (sim.linear_solver.preconditioner, sim.linear_solver, sim.lsys.jac, sim.lsys.r, sim.lsys.dx, sim.lsys._x, law, disc):
            obj.close()

    cycle()
    gc.collect()
    ctx.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(10):
        cycle()
    gc.collect()
    ctx.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20  # < 8 MiB drift (allocator granularity), one cycle allocates > 100 MiB


def test_min_iterations_callback_termination(ja):
    """IterativeSolverConfig.min_iterations (krylov.jl:120-131, 199-206): the solve may stop only at an iteration k >= min_iterations
    with ||r_k|| <= atol + rtol*||r_0||; the exit is Krylov.jl's "user-requested exit" (stats.solved == false)."""
    g = ja.tet_lattice_mesh(6, 5, 4)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    ctx = ja.HIPContext(0)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=64)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(T); law.set_volumes(g["volumes"])
    U = 1.0 + 0.1 * np.random.default_rng(1).random(nc)
    law.set_state(U); law.set_state0(np.zeros(nc))
    def solve(solver, min_it):
        lsys = ja.LinearizedSystem(disc)
        law.update_equation_and_linearized_system(1.0, lsys.jac, lsys.r)
        ks = ja.GenericKrylov(solver, preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-2,
                              max_iterations=80, min_iterations=min_it)
        return ja.linear_solve(lsys, ks)

    for solver in ("bicgstab", "gmres"):
        base = solve(solver, 1)
        n1 = base["iterations"]
        assert base["ok"] and base["status"] == 0 and 3 <= n1 <= 60
        # below the natural count the callback changes nothing but the reported status
        low = solve(solver, n1 - 1)
        assert low["iterations"] == n1 and low["status"] == 3 and not low["ok"]
        np.testing.assert_allclose(low["residuals"], base["residuals"], rtol=1e-12)
        # above it the solver keeps iterating until min_iterations and the residual keeps falling
        high = solve(solver, n1 + 6)
        assert high["iterations"] == n1 + 6 and high["status"] == 3
        assert high["residuals"][n1 + 6] < base["residuals"][n1]
        np.testing.assert_allclose(high["residuals"][: n1 + 1], base["residuals"], rtol=1e-9)


def test_simulate_output_restart_and_distributed_consolidation(ja, tmp_path):
    """simulate! with output_path / restart (simulator.jl:150-260, 680-705): a run interrupted after step 2 and restarted gives
    the states of the uninterrupted run; two in-process ranks write proc_<r> folders that consolidate to the same states."""
    import threading
    from jutul_amd import dd, io
    g = ja.tet_lattice_mesh(7, 6, 5)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    X0 = 1.0 + 0.1 * np.random.default_rng(5).random(nc)
    par = dict(rho0=(1.0, 1.0), compressibility=(5e-2, 5e-2), viscosity=(1.0, 1.0), p_ref=1.0)
    src = ([1, nc], np.array([[0.5], [-0.5]]))
    steps = [0.2, 0.4, 0.4, 0.8]

    def make(ctx, disc=None, law=None):
        if law is None:
            disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=64)
            law = ja.ConservationLaw(disc, "compressible", **par)
            law.set_face_trans(T); law.set_volumes(g["volumes"]); law.set_state(X0); law.set_state0(X0)
            law.set_sources(src[0], src[1].reshape(-1))
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-10,
                              max_iterations=200)
        return ja.Simulator(law, ks, tolerance=1e-9)

    ctx = ja.HIPContext(0)
    full = str(tmp_path / "full")
    its = make(ctx).simulate(steps, output_path=full)
    assert len(its) == 4 and io.valid_restart_indices(full) == [1, 2, 3, 4]
    ref = [io.read_restart(full, s)[0]["Pressure"] for s in (1, 2, 3, 4)]
    assert np.abs(ref[3] - ref[0]).max() > 1e-4  # the state really evolves
    # interrupted after two steps, then restart=True continues with step 3
    part = str(tmp_path / "part")
    make(ctx).simulate(steps[:2], output_path=part)
    its2 = make(ctx).simulate(steps, output_path=part, restart=True)
    assert len(its2) == 2
    for s in (1, 2, 3, 4):
        np.testing.assert_allclose(io.read_restart(part, s)[0]["Pressure"], ref[s - 1], rtol=1e-9)
    assert make(ctx).simulate(steps, output_path=part, restart=True) == []  # nothing left to do
    # two ranks
    dist_path = str(tmp_path / "dist")
    part2 = dd.partition_rcb(g["cell_centroids"], 2)
    group = ja.LocalCommGroup(2)
    err = []

    def rank_fn(r):
        try:
            c = ja.HIPContext(0)
            c.comm_init_local(group, r)
            disc, law, sub = dd.setup_rank_problem(c, g["N"], part2, r, T, g["volumes"], X0, kind="compressible", sources=src,
                                                   block_rows=64, law_params=par)
            make(c, disc, law).simulate(steps, output_path=dist_path, rank=r, cells_global=sub["cells"], n_total=nc)
            c.comm_finalize()
        except Exception as e:  # noqa: BLE001
            err.append(e)
            raise
    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not err, err
    io.consolidate_distributed_results_on_disk(dist_path, 2, [1, 2, 3, 4])
    for s in (1, 2, 3, 4):
        np.testing.assert_allclose(io.read_restart(dist_path, s)[0]["Pressure"], ref[s - 1], rtol=1e-7)


# ---- runtime-defined laws (SURVEY a-7 / 8f-3) ------------------------------------------------------------------------------------
COMPRESSIBLE_SRC = r"""
// par: rho0, c, mu, p_ref -- the built-in single-phase law restated as user code
__device__ D rho(D p, const double *par) { return par[0] * dexp(par[1] * (p - par[3])); }
__device__ void jh_flux(const D *self, const D *other, double T, double gdz, const double *par, D *q) {
  D rs = rho(self[0], par), ro = rho(other[0], par);
  D dphi = two_point_potential_drop(self[0], other[0], gdz, rs, ro);
  q[0] = (T / par[2]) * (face_average(rs, ro) * dphi);
}
__device__ void jh_mass(const D *x, const double *par, D *M) { M[0] = rho(x[0], par); }
"""

REACTION_SRC = r"""
// two coupled species: nonlinear diffusion of u with a v-dependent mobility, linear diffusion of v, cubic storage of u
__device__ void jh_flux(const D *s, const D *o, double T, double gdz, const double *par, D *q) {
  D mob = par[0] + face_average(s[1], o[1]) * face_average(s[1], o[1]);
  q[0] = T * (mob * (s[0] - o[0]));
  q[1] = (par[1] * T) * (s[1] - o[1]) + gdz * upwind(s[1] - o[1], s[1], o[1]);
}
__device__ void jh_mass(const D *x, const double *par, D *M) {
  M[0] = x[0] + par[2] * (x[0] * x[0] * x[0]);
  M[1] = dsqrt(x[1]) * x[0];
}
"""


def test_custom_law_reproduces_the_builtin_compressible_law(ja, oracle):
    g = ja.tet_lattice_mesh(6, 5, 4)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    rng = np.random.default_rng(11)
    gdz = 0.05 * rng.standard_normal(g["nf"])
    P, P0 = 1.0 + 0.2 * rng.random(nc), 1.0 + 0.2 * rng.random(nc)
    ctx = ja.HIPContext(0)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=64)
    out = []
    for law in (ja.ConservationLaw(disc, "compressible", rho0=(1.3, 1.0), compressibility=(0.07, 0.0), viscosity=(0.9, 1.0), p_ref=1.1),
                ja.ConservationLaw(disc, "custom", source=COMPRESSIBLE_SRC, params=[1.3, 0.07, 0.9, 1.1])):
        law.set_face_trans(T); law.set_face_gdz(gdz); law.set_volumes(g["volumes"]); law.set_state(P); law.set_state0(P0)
        law.set_sources([3], [0.25])
        lsys = ja.LinearizedSystem(disc)
        law.update_equation_and_linearized_system(0.7, lsys.jac, lsys.r)
        out.append((lsys.r.download(), np.array(lsys.jac.nzval)))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=1e-13, atol=1e-15)
    # independent checker: the oracle's cell-based generic-AD path (GenericAutoDiffCache, equations.jl:578-594, ad/generic.jl:53-96)
    osys = oracle.TPFASystem(g["N"], nc)
    nz_o, r_o = oracle.generic_cell_assemble("compressible", 1, nc, osys.hfm, osys.rowptr, osys.colidx, P, P0, g["volumes"], T, gdz, 0.7,
                                             [1.3, 0.07, 0.9, 1.1], [3], [0.25])
    np.testing.assert_allclose(out[1][0], r_o, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out[1][1], nz_o, rtol=1e-12, atol=1e-14)
    # and a full Newton step converges with it
    law.set_state(P0)
    ok, its, _ = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"),
                                                    relative_tolerance=1e-10), tolerance=1e-9).solve_ministep(0.7)
    assert ok and 2 <= its <= 8


def test_custom_two_equation_law_jacobian_matches_finite_differences(ja, oracle):
    """A law that is not built in (2 equations per cell): residual and dual-generated 2x2-block Jacobian == the oracle's
    cell-based generic-AD path slot by slot, and == central differences of the residual; conservation of the flux part;
    compile errors are reported."""
    g = ja.tet_lattice_mesh(4, 4, 3)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    rng = np.random.default_rng(12)
    gdz = 0.1 * rng.standard_normal(g["nf"])
    X = np.stack([1.0 + 0.3 * rng.random(nc), 0.5 + 0.4 * rng.random(nc)], axis=1).reshape(-1)
    X0 = X * (1.0 + 0.05 * rng.standard_normal(X.size))
    ctx = ja.HIPContext(0)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=2, reorder="blocks", block_rows=64)
    law = ja.ConservationLaw(disc, "custom", source=REACTION_SRC, params=[0.4, 0.7, 0.3])
    law.set_face_trans(T); law.set_face_gdz(gdz); law.set_volumes(g["volumes"]); law.set_state0(X0)
    lsys = ja.LinearizedSystem(disc)
    dt = 0.9

    def residual(x):
        law.set_state(x)
        law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
        return lsys.r.download()

    r0 = residual(X)
    A = lsys.jac  # Jacobian at X
    osys = oracle.TPFASystem(g["N"], nc, nblk=2)
    nz_o, r_o = oracle.generic_cell_assemble("reaction", 2, nc, osys.hfm, osys.rowptr, osys.colidx, X, X0, g["volumes"], T, gdz, dt,
                                             [0.4, 0.7, 0.3])
    np.testing.assert_allclose(r0, r_o, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(np.array(A.nzval), nz_o, rtol=1e-12, atol=1e-13)
    nz_f, r_f = oracle.fvm_face_assemble("reaction", 2, nc, g["N"], osys.rowptr, osys.colidx, X, X0, g["volumes"], T, gdz, dt, [0.4, 0.7, 0.3])
    np.testing.assert_allclose(np.array(A.nzval), nz_f, rtol=1e-12, atol=1e-13)  # face-based PotentialFlow{:fvm} form
    y = np.zeros(2 * nc)
    v = rng.standard_normal(2 * nc)
    Jv = ja.mul_(ja.DeviceVector(disc), A, ja.DeviceVector(disc, v)).download()
    h = 1e-6
    fd = (residual(X + h * v) - residual(X - h * v)) / (2 * h)
    # the SPU upwind switch of equation 2 is only piecewise smooth: rows next to a switching face may differ
    bad = np.abs(Jv - fd) > 1e-6 * (1 + np.abs(fd))
    assert bad.mean() < 0.02, bad.mean()
    # flux antisymmetry => the flux part of each equation sums to zero: sum(r) == sum(vol*(M - M0))/dt
    u, w, u0, w0 = X[0::2], X[1::2], X0[0::2], X0[1::2]
    M1, M10 = u + 0.3 * u ** 3, u0 + 0.3 * u0 ** 3
    assert np.isclose(r0[0::2].sum(), (g["volumes"] * (M1 - M10)).sum() / dt, rtol=1e-10)
    with pytest.raises(ja.JutulHIPError, match="does not compile"):
        ja.ConservationLaw(disc, "custom", source="__device__ void jh_flux() { syntax error }", params=[])


# ---- multigraph neighbourships (two or more faces between one cell pair) ---------------------------------------------------------
@pytest.mark.parametrize("reorder,law", [("none", "poisson"), ("blocks", "compressible")])
def test_multigraph_neighbourship_matches_reference_semantics(ja, ctx, oracle, reorder, law):
    """The reference accepts a neighbourship with parallel faces: its pattern holds one entry per cell pair (sparse() merges
    duplicates, conservation.jl:486-505), residual and diagonal sum every half-face, and the off-diagonal slot is ASSIGNED per
    half-face in ascending face order (ad.jl:74-76), i.e. the last parallel face wins.  Connectivity tables, pattern and
    positions bit-exact; residual / Jacobian / SpMV / ILU(0) / BiCGStab against the oracle, which restates exactly that."""
    dims = (4, 3, 2) if law == "compressible" else (6, 5, 1)  # the 2-D case keeps every row within the jagged SpMV's 8 entries
    N0 = ja.cartesian_neighbors(dims)
    nc = int(np.prod(dims))
    rng = np.random.default_rng(21)
    pick = rng.choice(N0.shape[1], size=7, replace=False)
    extra = N0[:, pick].copy()
    extra[:, ::2] = extra[::-1, ::2]            # some parallel faces with the opposite orientation
    N = np.concatenate([N0, extra, N0[:, pick[:2]]], axis=1)  # two pairs even get a third face
    nf = N.shape[1]
    T, gdz, vol = 0.5 + rng.random(nf), 0.05 * rng.standard_normal(nf), 0.5 + rng.random(nc)
    X, X0 = 1.0 + 0.2 * rng.random(nc), 1.0 + 0.2 * rng.random(nc)
    dt = 0.6
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder=reorder, block_rows=8)
    osys = oracle.TPFASystem(N, nc)
    # a-1 .. a-4: bit-exact
    c = disc.conn
    for k, ok in (("face_pos", "face_pos"), ("self", "self"), ("other", "other"), ("face", "faces"), ("face_sign", "face_sign")):
        assert np.array_equal(c[k], osys.hfm[ok]), k
    rp, ci = disc.pattern()
    assert np.array_equal(rp, osys.rowptr) and np.array_equal(ci, osys.colidx) and disc.nnzb == osys.nnzb < nc + 2 * nf
    pa, pfl = disc.jacobian_positions()
    assert np.array_equal(pa.reshape(-1), np.asarray(osys.pos_acc).reshape(-1)) and np.array_equal(pfl.reshape(-1), np.asarray(osys.pos_flux).reshape(-1))
    # a-5 / a-6
    par = dict(rho0=(1.2, 1.0), compressibility=(0.05, 0.0), viscosity=(0.8, 1.0), p_ref=1.0)
    L = ja.ConservationLaw(disc, law, **par)
    L.set_face_trans(T); L.set_volumes(vol); L.set_state(X); L.set_state0(X0)
    olaw = oracle.Law(law, dt, rho0=par["rho0"], comp=par["compressibility"], mu=par["viscosity"], p_ref=par["p_ref"])
    g = None
    if law == "compressible":
        L.set_face_gdz(gdz)
        g = gdz
    L.set_sources([2], [0.3])
    lsys = ja.LinearizedSystem(disc)
    L.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
    nz_o, r_o = osys.assemble(olaw, X, X0, vol, T, gdz=g, src_cells=[2], src_values=[0.3])
    np.testing.assert_allclose(lsys.r.download(), r_o, rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(lsys.jac.nzval, nz_o, rtol=RTOL, atol=1e-14)
    # a-10: both SpMV layouts on the device pattern with its shadow slots
    x = rng.standard_normal(nc)
    ref = oracle.spmv(nc, 1, osys.rowptr, osys.colidx, nz_o, x)
    for jag in (False, True):
        try:
            y = ja.mul_(ja.DeviceVector(disc), lsys.jac, ja.DeviceVector(disc, x), jagged=jag).download()
        except ja.JutulHIPError as e:  # a cell with more than 8 entries (7-point stencil + parallel faces): CSR tile kernel only
            assert jag and "jagged" in str(e)
            continue
        assert relerr(y, ref) < RTOL
    # values round trip through the host pattern
    lsys.jac.nzval = nz_o * 2.0
    np.testing.assert_array_equal(lsys.jac.nzval, nz_o * 2.0)
    lsys.jac.nzval = nz_o
    # a-11 .. a-14 on the merged pattern (device order == host order with reorder="none": identical ILU(0))
    if reorder == "none":
        F = ja.ilu0_csr(lsys.jac)
        Fo = oracle.ILU0(nc, 1, osys.rowptr, osys.colidx, nz_o)
        np.testing.assert_allclose(F.factor_values(), Fo.export(nz_o.size), rtol=1e-11, atol=1e-13)
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks" if reorder == "blocks" else None),
                          relative_tolerance=1e-11, max_iterations=200)
    out = ja.linear_solve(lsys, ks)
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    J = sp.csr_matrix((nz_o, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
    assert out["ok"] and np.allclose(lsys.dx.download(), -spl.spsolve(J.tocsc(), r_o), rtol=1e-7, atol=1e-9)


# ---- Simulator.perform_step honours every GenericKrylov / IterativeSolverConfig option (round-1 advisor finding) ------------------
@pytest.mark.parametrize("opts", [dict(solver="gmres"), dict(scaling="diagonal"), dict(scaling="dt"), dict(true_residual=True),
                                  dict(nonlinear_relative_tolerance=1e-2)])
def test_simulator_routes_non_default_solver_options(ja, ctx, oracle, opts):
    """jh_newton_step is the fast path for the reference's default solver set-up only; GMRES, :diagonal / :dt scaling,
    true_residual and the Newton-history relaxed tolerance (linsolve/krylov.jl:71-182) must not be silently replaced by plain
    BiCGStab: perform_step then runs assemble -> convergence -> linear_solve() -> update, and the ministep converges to the
    same state as with the default solver."""
    g = ja.tet_lattice_mesh(5, 4, 3)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    rng = np.random.default_rng(17)
    P0 = 1.0 + 0.2 * rng.random(nc)
    par = dict(rho0=(1.2, 1.0), compressibility=(0.05, 0.0), viscosity=(0.8, 1.0), p_ref=1.0)

    def run(**kw):
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", block_rows=64)
        law = ja.ConservationLaw(disc, "compressible", **par)
        law.set_face_trans(T); law.set_volumes(g["volumes"]); law.set_state(P0); law.set_state0(P0)
        law.set_sources([1, nc], [0.4, -0.4])
        solver = kw.pop("solver", "bicgstab")
        scaling = kw.pop("scaling", "none")
        ks = ja.GenericKrylov(solver, preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), scaling=scaling,
                              relative_tolerance=1e-10, max_iterations=300, **kw)
        sim = ja.Simulator(law, ks, tolerance=1e-8)
        general = sim._needs_general_path()
        ok, its, rep = sim.solve_ministep(0.8)
        assert ok
        return law.get_state(), general, rep

    ref, general_ref, _ = run()
    got, general, rep = run(**dict(opts))
    assert not general_ref and general
    # (the relaxed tolerance solves the later Newton iterations more loosely: same fixed point to the Newton tolerance)
    assert np.allclose(got, ref, rtol=1e-6 if "nonlinear_relative_tolerance" in opts else 1e-8, atol=1e-10)
    assert rep.converged == 1


def test_halo_plan_must_agree_with_the_discretisation_on_n_owned(ja):
    """jh_halo_create rejects a plan whose owned / ghost split differs from the one the discretisation was ordered for (dots, norms
    and unit_diagonalize! take the owned cells to be the leading device rows; round-1 advisor finding)."""
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(4, 4, 3)
    part = dd.partition_rcb(g["cell_centroids"], 2)
    sub = dd.local_subdomain(g["N"], part, 1)
    c = ja.HIPContext(0)
    disc = ja.TwoPointPotentialFlowHardCoded(c, sub["N"], sub["n_local"], reorder="blocks", block_rows=32)  # n_owned not given: ghosts get mixed in
    with pytest.raises(ja.JutulHIPError, match="n_owned"):
        disc.set_halo(sub["n_owned"], sub["neighbors"], sub["send"], sub["recv"])
    good = ja.TwoPointPotentialFlowHardCoded(c, sub["N"], sub["n_local"], reorder="blocks", block_rows=32, n_owned=sub["n_owned"])
    good.set_halo(sub["n_owned"], sub["neighbors"], sub["send"], sub["recv"])
    assert good.halo_info()["n_owned"] == sub["n_owned"]


def test_device_side_reports_and_per_variable_download(ja, ctx):
    """increment_norm (models.jl:955-965), variable_change_report (models.jl:1023-1038) and the per-variable state download behind
    get_output_state (models.jl:1048-1058) for a 2-variable law on a reordered grid: against numpy on the host arrays."""
    import ctypes as C
    from jutul_amd import _lib
    from jutul_amd._lib import check, pf
    g, rng = tet_case(ja, (7, 6, 5), seed=21)
    nc = g["nc"]
    n_own = nc - 37                     # a rank-local model: owned cells first, the last 37 host cells are ghosts (last device rows)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=2, reorder="blocks", block_rows=64, n_owned=n_own)
    law = ja.ConservationLaw(disc, "twophase", rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
    X = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.copy()      # [nc, 2] = the reference's [N, nc] column-major
    X0 = np.stack([rng.uniform(1.0, 2.0, nc), rng.uniform(0.2, 0.8, nc)]).T.copy()
    law.set_state(X)
    law.set_state0(X0)
    dxh = rng.standard_normal((nc, 2))
    dx = ja.DeviceVector(disc, dxh)
    for e, (s_, m_) in enumerate(law.increment_norm(dx)):
        assert abs(s_ - np.abs(dxh[:, e]).sum()) <= 1e-12 * np.abs(dxh[:, e]).sum() and m_ == np.abs(dxh[:, e]).max()
    for e, (s_, m_) in enumerate(law.increment_norm(dx, n_owned=n_own)):                   # owned cells only
        assert abs(s_ - np.abs(dxh[:n_own, e]).sum()) <= 1e-12 * np.abs(dxh[:, e]).sum() and m_ == np.abs(dxh[:n_own, e]).max()
    for e, rep in enumerate(law.change_report()):
        d = np.abs(X[:, e] - X0[:, e])
        assert abs(rep["dx"][0] - d.sum()) <= 1e-12 * d.sum() and rep["dx"][1] == d.max()
        assert abs(rep["x"][0] - np.abs(X[:, e]).sum()) <= 1e-12 * np.abs(X[:, e]).sum() and rep["x"][1] == np.abs(X[:, e]).max()
    dxh[5, 1] = np.nan                                                                    # check_increment: a non-finite entry shows in the sum
    dx.upload(dxh)
    assert not np.isfinite(law.increment_norm(dx)[1][0]) and np.isfinite(law.increment_norm(dx)[0][0])
    assert np.isnan(law.increment_norm(dx)[1][1])      # ... and in the maximum (Julia's max propagates NaN, models.jl:955-965)
    for lane in (0, 63, 64, nc - 1):                    # wherever it sits in its wavefront / workgroup
        dxh[:, 1] = rng.standard_normal(nc)
        dxh[lane, 1] = np.nan
        dx.upload(dxh)
        assert np.isnan(law.increment_norm(dx)[1][1]) and np.isfinite(law.increment_norm(dx)[0][1])
    L = _lib.load()
    out = np.empty(nc)
    check(L.jh_host_register(out.ctypes.data_as(C.c_void_p), out.nbytes))                # page-locked target, as JutulHIP.jl does for state0[k]
    try:
        for which, ref in ((0, X), (1, X0)):
            for e in range(2):
                check(L.jh_law_get_variable(law.h, which, e, pf(out)))
                assert np.array_equal(out, ref[:, e])
    finally:
        check(L.jh_host_unregister(out.ctypes.data_as(C.c_void_p)))


def test_context_options_api(ja):
    """jh_context_set_option / jh_context_get_option: named integer options per context (the keyword arguments of the reference's
    contexts and solver set-up, contexts/csr.jl:3-23); unknown keys are errors; JH_OPTIONS seeds new contexts."""
    import os
    c = ja.HIPContext(0, consumer_reduce=0, spmv_col_bits=16)
    assert c.get_option("consumer_reduce") == 0 and c.get_option("spmv_col_bits") == 16 and c.get_option("fuse_gather") == 1
    c.set_option("consumer_reduce", 1)
    assert c.get_option("consumer_reduce") == 1
    with pytest.raises(ja.JutulHIPError, match="unknown option"):
        c.set_option("no_such_option", 1)
    with pytest.raises(ja.JutulHIPError, match="unknown option"):
        c.get_option("no_such_option")
    assert ja.HIPContext(0).get_option("consumer_reduce") == 1          # per context, not per process
    os.environ["JH_OPTIONS"] = "sync_loop=1,comm_timeout_ms=2500"
    try:
        d = ja.HIPContext(0)
        assert d.get_option("sync_loop") == 1 and d.get_option("comm_timeout_ms") == 2500
        os.environ["JH_OPTIONS"] = "bogus=1"
        with pytest.raises(ja.JutulHIPError, match="unknown option"):
            ja.HIPContext(0)
    finally:
        os.environ.pop("JH_OPTIONS", None)


def test_option_values_and_context_knobs_are_validated(ja):
    """What goes into launch geometry or shared state unvalidated is a hang or a race waiting for a typo: ilu_factor_threads only
    takes the shapes its kernels are written for, JH_OPTIONS only integers, a CU mask only compute units the device has, the
    communicator knobs need a communicator; jh_host_unregister releases only what jh_host_register pinned itself."""
    import ctypes as C
    import os
    from jutul_amd._lib import load, check
    c = ja.HIPContext(0)
    for bad in (1, 32, 96, 1000):
        with pytest.raises(ja.JutulHIPError, match="ilu_factor_threads"):
            c.set_option("ilu_factor_threads", bad)
    for good in (0, 64, 128, 256, 512):
        c.set_option("ilu_factor_threads", good)
    os.environ["JH_OPTIONS"] = "sync_loop=on"
    try:
        with pytest.raises(ja.JutulHIPError, match="not an integer"):
            ja.HIPContext(0)
    finally:
        os.environ.pop("JH_OPTIONS", None)
    with pytest.raises(ja.JutulHIPError, match="compute units"):
        c.set_cu_mask(200, 100)
    c.set_cu_mask(64, 64)      # a masked context computes the same numbers
    g, rng = tet_case(ja, (8, 7, 6), seed=2)
    disc = ja.TwoPointPotentialFlowHardCoded(c, g["N"], g["nc"], reorder="blocks", block_rows=64)
    lsys = ja.LinearizedSystem(disc)
    nz = rng.standard_normal(disc.pattern()[1].size)
    lsys.jac.nzval = nz
    x = rng.standard_normal(g["nc"])
    y_masked = ja.mul_(ja.DeviceVector(disc), lsys.jac, ja.DeviceVector(disc, x)).download()
    c2 = ja.HIPContext(0)
    disc2 = ja.TwoPointPotentialFlowHardCoded(c2, g["N"], g["nc"], reorder="blocks", block_rows=64)
    l2 = ja.LinearizedSystem(disc2)
    l2.jac.nzval = nz
    assert np.array_equal(y_masked, ja.mul_(ja.DeviceVector(disc2), l2.jac, ja.DeviceVector(disc2, x)).download())
    c.set_cu_mask(0, 0)        # mask removed
    with pytest.raises(ja.JutulHIPError, match="no communicator"):
        c.comm_set_exclusive(True)
    # page-locking: a second registration of the same range is success and not ours; unregistering twice is harmless
    L = load()
    a = np.zeros(1 << 16)
    p = a.ctypes.data_as(C.c_void_p)
    check(L.jh_host_register(p, a.nbytes))
    check(L.jh_host_register(p, a.nbytes))      # already page-locked: success
    check(L.jh_host_unregister(p))
    check(L.jh_host_unregister(p))              # not registered by the library any more: no-op
    b = np.ones(8)
    check(L.jh_host_unregister(b.ctypes.data_as(C.c_void_p)))   # never registered: no-op, not an error


@pytest.mark.parametrize("nt", [0, 1])
def test_jagged_spmv_stream_policy_same_bits(ja, oracle, nt):
    """spmv_nontemporal: the matrix stream of the jagged SpMV through plain or non-temporal loads (chosen by working set against the
    Infinity Cache when left at -1) -- a cache policy, the product's bits do not change"""
    g, rng = tet_case(ja, (14, 12, 10), seed=5)
    ctx = ja.HIPContext(0, spmv_nontemporal=nt)
    assert ctx.get_option("spmv_nontemporal") == nt
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], reorder="blocks")
    lsys = ja.LinearizedSystem(disc)
    rowptr, colidx = disc.pattern()
    nz = rng.standard_normal(colidx.size)
    lsys.jac.nzval = nz
    x = rng.standard_normal(g["nc"])
    xv = ja.DeviceVector(disc, x)
    y = ja.mul_(ja.DeviceVector(disc), lsys.jac, xv).download()
    yj = ja.mul_(ja.DeviceVector(disc), lsys.jac, xv, jagged=True).download()
    assert np.array_equal(y, yj)
    assert relerr(y, oracle.spmv(g["nc"], 1, rowptr, colidx, nz, x)) < RTOL
