"""Oracle-internal cross-checks of the generic-AD formulations (SURVEY a-7): (i) the cell-based GenericAutoDiffCache path
(equations.jl:578-594, ad/generic.jl:53-96) and (ii) the face-based PotentialFlow{:fvm} assembly (fvm_assembly.jl:175-283) must
give the residual and Jacobian of the half-face path (a-5 / a-6, conservation.jl:298-430) -- three formulations, two dual-number
implementations, flux antisymmetry used by only one of them.  The device tests compare runtime-defined laws with (i)."""
import numpy as np
import pytest

RT = 1e-12


def mesh(oracle, dims, seed):
    rng = np.random.default_rng(seed)
    geo = oracle.cartesian_geometry(dims)
    N, nc, nf = geo["N"], geo["nc"], geo["N"].shape[1]
    return N, nc, nf, rng


@pytest.mark.parametrize("dims", [(5, 4, 3), (7, 1, 1), (3, 3, 1)])
def test_cell_based_generic_ad_equals_half_face_path_compressible(oracle, dims):
    N, nc, nf, rng = mesh(oracle, dims, 3)
    T, gdz, vol = 0.5 + rng.random(nf), 0.05 * rng.standard_normal(nf), 0.5 + rng.random(nc)
    P, P0 = 1.0 + 0.2 * rng.random(nc), 1.0 + 0.2 * rng.random(nc)
    dt = 0.7
    osys = oracle.TPFASystem(N, nc)
    law = oracle.Law("compressible", dt, rho0=(1.3, 1.0), comp=(0.07, 0.0), mu=(0.9, 1.0), p_ref=1.1)
    nz_h, r_h = osys.assemble(law, P, P0, vol, T, gdz=gdz, src_cells=[2, nc], src_values=[0.25, -0.5])
    par = [1.3, 0.07, 0.9, 1.1]
    nz_c, r_c = oracle.generic_cell_assemble("compressible", 1, nc, osys.hfm, osys.rowptr, osys.colidx, P, P0, vol, T, gdz, dt, par,
                                             [2, nc], [0.25, -0.5])
    nz_f, r_f = oracle.fvm_face_assemble("compressible", 1, nc, N, osys.rowptr, osys.colidx, P, P0, vol, T, gdz, dt, par, [2, nc], [0.25, -0.5])
    for nz, r in ((nz_c, r_c), (nz_f, r_f)):
        np.testing.assert_allclose(r, r_h, rtol=RT, atol=1e-14)
        np.testing.assert_allclose(nz, nz_h, rtol=RT, atol=1e-14)


def test_reaction_law_three_ways_and_finite_differences(oracle):
    """The two-equation user law of the device tests: cell-based == face-based; Jacobian == central differences of the residual
    away from the upwind kinks; flux part conservative."""
    N, nc, nf, rng = mesh(oracle, (4, 4, 3), 12)
    T, gdz, vol = 0.5 + rng.random(nf), 0.1 * rng.standard_normal(nf), 0.5 + rng.random(nc)
    X = np.stack([1.0 + 0.3 * rng.random(nc), 0.5 + 0.4 * rng.random(nc)], axis=1).reshape(-1)
    X0 = X * (1.0 + 0.05 * rng.standard_normal(X.size))
    par, dt = [0.4, 0.7, 0.3], 0.9
    osys = oracle.TPFASystem(N, nc, nblk=2)
    args = (2, nc, osys.hfm, osys.rowptr, osys.colidx)

    def res(x):
        return oracle.generic_cell_assemble("reaction", *args, x, X0, vol, T, gdz, dt, par)

    nz_c, r_c = res(X)
    nz_f, r_f = oracle.fvm_face_assemble("reaction", 2, nc, N, osys.rowptr, osys.colidx, X, X0, vol, T, gdz, dt, par)
    np.testing.assert_allclose(r_f, r_c, rtol=RT, atol=1e-14)
    np.testing.assert_allclose(nz_f, nz_c, rtol=RT, atol=1e-13)
    v = rng.standard_normal(2 * nc)
    Jv = oracle.spmv(nc, 2, osys.rowptr, osys.colidx, nz_c, v)
    h = 1e-6
    fd = (res(X + h * v)[1] - res(X - h * v)[1]) / (2 * h)
    bad = np.abs(Jv - fd) > 1e-6 * (1 + np.abs(fd))
    assert bad.mean() < 0.02
    u, u0 = X[0::2], X0[0::2]
    assert np.isclose(r_c[0::2].sum(), (vol * ((u + 0.3 * u ** 3) - (u0 + 0.3 * u0 ** 3))).sum() / dt, rtol=1e-10)
