"""Componentwise checks of a block-Jacobi ILU(0) (ilu0_csr / ldiv!, StaticCSR/ilu0.jl:108-187, par_ilu0.jl:75-80) against the
oracle's factors -- no bound is scaled by the largest entry of an array:

  factor values   |f - f_o|_ij <= c eps (|L||U|)_ij-type bounds, entry by entry (the size of the terms the elimination adds up for
                  that entry, from the ORACLE's factors), for L multipliers, inverted pivots and U entries;
  triangular solve  componentwise backward error |L U x - b| <= c levels eps (|L||U||x| + |b|) with the ORACLE's L and U.

Works for N x N blocks through the scalar expansion of the block factors."""
import numpy as np
import scipy.sparse as sp

EPS = np.finfo(np.float64).eps


def _expand(rowptr0, colidx0, blocks, n, bs):
    """block CSR (blocks [nnz, bs, bs] as [row, col]) -> scalar CSR"""
    return sp.bsr_matrix((blocks, colidx0, rowptr0), shape=(n * bs, n * bs)).tocsr()


def oracle_factors(oracle, n, bs, rowptr1, colidx1, nz_flat, partition):
    """The oracle's block-Jacobi ILU(0) of the matrix (1-based CSR pattern, blocks column-major in nz_flat) as scalar sparse
    matrices: returns (Fo, lu_flat, L, U) with L unit lower, U upper incl. the pivots (inverted back from the stored inverses)."""
    Fo = oracle.ILU0(n, bs, rowptr1, colidx1, nz_flat, partition=partition)
    lu = Fo.export(nz_flat.size)
    rp, ci = rowptr1 - 1, colidx1 - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    blk = lu.reshape(-1, bs, bs).transpose(0, 2, 1).copy()           # [row, col] inside a block
    lower, upper, dia = ci < rows, ci > rows, ci == rows
    piv = np.linalg.inv(blk[dia])                                    # the factor stores inv(U_ii)
    Lb = np.where(lower[:, None, None], blk, 0.0)
    Ub = np.where(upper[:, None, None], blk, 0.0)
    Ub[dia] = piv
    L = _expand(rp, ci, Lb, n, bs) + sp.identity(n * bs, format="csr")
    U = _expand(rp, ci, Ub, n, bs)
    L.eliminate_zeros()
    U.eliminate_zeros()
    return Fo, lu, L.tocsr(), U.tocsr()


def check_factor_values(lu_dev, lu_o, L, U, n, bs, rowptr1, colidx1, c=64.0):
    """lu_dev / lu_o: flat factor exports in the SAME slot order (L multipliers, inv(U_ii), U; zero outside the blocks).
    Entry bounds from the oracle's factors: T = |L||U| on the pattern is what the elimination sums for an entry of A;
    U_ij: c eps T_ij; L_ik = (...) inv(U_kk): c eps T_ik * |inv(U_kk)| (block norms); inv(U_ii): c eps |inv| T_ii |inv|."""
    rp, ci = rowptr1 - 1, colidx1 - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    T = (abs(L) @ abs(U)).tocsr()
    # block-wise magnitude of T at the pattern's slots
    Tb = np.zeros(ci.size)
    Tcoo = T.tocoo()
    key = (Tcoo.row // bs).astype(np.int64) * n + (Tcoo.col // bs)
    slot_key = rows.astype(np.int64) * n + ci
    order = np.argsort(slot_key)
    pos = np.searchsorted(slot_key[order], key)
    ok = (pos < slot_key.size) & (slot_key[order][np.minimum(pos, slot_key.size - 1)] == key)
    np.add.at(Tb, order[pos[ok]], Tcoo.data[ok])                     # sum of the block's |terms|
    blk_o = lu_o.reshape(-1, bs * bs)
    blk_d = lu_dev.reshape(-1, bs * bs)
    inside = np.abs(blk_o).sum(axis=1) != 0                          # entries outside every block are not part of the factor
    dia = ci == rows
    pivnorm = np.zeros(n)
    pivnorm[rows[dia]] = np.abs(blk_o[dia]).sum(axis=1)              # |inv(U_kk)|
    lower, upper = ci < rows, ci > rows
    bound = np.zeros(ci.size)
    bound[upper] = Tb[upper]
    bound[lower] = Tb[lower] * pivnorm[ci[lower]]
    bound[dia] = pivnorm[rows[dia]] ** 2 * Tb[dia]
    err = np.abs(blk_d - blk_o).max(axis=1)
    m = inside
    worst = float((err[m] / np.maximum(bound[m], 1e-300)).max() / EPS) if m.any() else 0.0
    assert np.all(err[m] <= c * EPS * bound[m]), f"factor entry off by {worst:.1f} eps of its own terms (allowed {c})"
    # and nothing is written outside the blocks
    assert not np.any(blk_d[~inside]), "factor values outside the block-Jacobi blocks"
    return worst


def check_triangular_solve(x_dev, b, L, U, levels, c=4.0):
    """componentwise backward error of x_dev ~ (L U)^-1 b with the oracle's factors"""
    resid = np.abs(L @ (U @ x_dev) - b)
    scale = abs(L) @ (abs(U) @ np.abs(x_dev)) + np.abs(b)
    worst = float((resid / np.maximum(scale, 1e-300)).max() / EPS)
    assert np.all(resid <= c * max(levels, 1) * EPS * scale), f"triangular solve: backward error {worst:.1f} eps (allowed {c * levels})"
    return worst


def device_order_problem(n, bs, rowptr1, colidx1, nz_flat, perm1, block_ptr):
    """The matrix in the device's elimination order (block rows permuted, blocks intact) and the block-Jacobi partition of the
    device blocks: returns (rowptr1_p, colidx1_p, nz_p, part_p, slot_of_p) with slot_of_p[k] = host slot of permuted slot k."""
    p0 = perm1 - 1
    pat = sp.csr_matrix((np.arange(1, colidx1.size + 1), colidx1 - 1, rowptr1 - 1), shape=(n, n))
    Pp = pat[p0][:, p0].tocsr()
    Pp.sort_indices()
    slot = Pp.data - 1
    nz_p = nz_flat.reshape(-1, bs * bs)[slot].reshape(-1)
    part = np.repeat(np.arange(1, len(block_ptr)), np.diff(block_ptr))
    return Pp.indptr + 1, Pp.indices + 1, nz_p, part, slot
