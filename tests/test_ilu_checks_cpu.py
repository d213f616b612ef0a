"""The componentwise ILU(0) checks the GPU tests use (tests/_ilu_checks.py), exercised on the CPU: the oracle passes its own
bounds, a perturbation of one entry by a few hundred eps of its own terms does not."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import _ilu_checks as ck


def _problem(oracle, bs, seed):
    rng = np.random.default_rng(seed)
    geo = oracle.cartesian_geometry((7, 6, 5), (1.0, 1.0, 1.0))
    n = geo["nc"]
    h = oracle.half_face_map(geo["N"], n)
    rowptr, colidx = oracle.csr_pattern(n, h)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    blk = rng.standard_normal((colidx.size, bs, bs)) * np.exp(rng.uniform(-3, 3, colidx.size))[:, None, None] * 0.2
    dg = colidx - 1 == rows
    blk[dg] += 6.0 * np.eye(bs) * np.exp(rng.uniform(-1, 1, dg.sum()))[:, None, None]
    nz = blk.transpose(0, 2, 1).reshape(-1)          # blocks column-major in the flat buffer
    part = oracle.partition_linear(5, n)
    return n, rowptr, colidx, nz, part, rng


@pytest.mark.parametrize("bs", [1, 2])
def test_oracle_passes_its_own_componentwise_bounds(oracle, bs):
    n, rowptr, colidx, nz, part, rng = _problem(oracle, bs, 3 + bs)
    Fo, lu, L, U = ck.oracle_factors(oracle, n, bs, rowptr, colidx, nz, part)
    # L U reproduces A on the kept pattern (ILU(0) identity): the scalar expansion is the right one
    A = sp.bsr_matrix((nz.reshape(-1, bs, bs).transpose(0, 2, 1), colidx - 1, rowptr - 1), shape=(n * bs, n * bs)).tocsr()
    keep = np.repeat(part[np.repeat(np.arange(n), np.diff(rowptr))] == part[colidx - 1], bs * bs)
    P = (L @ U).tocsr()
    kept = sp.bsr_matrix((np.where(keep.reshape(-1, bs, bs), nz.reshape(-1, bs, bs).transpose(0, 2, 1), 0.0), colidx - 1, rowptr - 1),
                         shape=(n * bs, n * bs)).tocsr()
    D = (P - kept).multiply(kept != 0)
    assert abs(D).max() <= 1e-12 * abs(A).max()
    assert ck.check_factor_values(lu.copy(), lu, L, U, n, bs, rowptr, colidx) == 0.0
    b = rng.standard_normal(n * bs)
    w = ck.check_triangular_solve(Fo.apply(b), b, L, U, levels=8)
    assert w < 32
    # one entry off by 1000 eps of its own terms is caught; a solve with one component off by 1e-9 too
    bad = lu.copy()
    k = np.flatnonzero(bad)[17]
    bad[k] *= 1.0 + 1e-9
    with pytest.raises(AssertionError):
        ck.check_factor_values(bad, lu, L, U, n, bs, rowptr, colidx)
    xb = Fo.apply(b)
    xb[5] *= 1.0 + 1e-9
    with pytest.raises(AssertionError):
        ck.check_triangular_solve(xb, b, L, U, levels=8)


def test_device_order_problem_is_a_symmetric_permutation(oracle):
    n, rowptr, colidx, nz, part, rng = _problem(oracle, 2, 9)
    perm = rng.permutation(n) + 1
    rp, ci, nzp, partp, slot = ck.device_order_problem(n, 2, rowptr, colidx, nz, perm, np.array([0, 50, 120, n]))
    A = sp.bsr_matrix((nz.reshape(-1, 2, 2).transpose(0, 2, 1), colidx - 1, rowptr - 1), shape=(2 * n, 2 * n)).toarray()
    Ap = sp.bsr_matrix((nzp.reshape(-1, 2, 2).transpose(0, 2, 1), ci - 1, rp - 1), shape=(2 * n, 2 * n)).toarray()
    pe = ((perm - 1)[:, None] * 2 + np.arange(2)[None, :]).reshape(-1)
    assert np.array_equal(Ap, A[pe][:, pe])
    assert np.array_equal(nz.reshape(-1, 4)[slot].reshape(-1), nzp) and partp.size == n
