"""Synthetic grids for tests and benchmarks (build-side; the reference's mesh generators -- src/meshes, Gmsh
extension -- are out of scope, the kernels consume only N (2 x nf), T_f and volumes).

* cartesian_neighbors: interior neighborship of CartesianMesh in MRST face order (meshes/cart.jl:197-225).
* tet_lattice_mesh: "tet-like" unstructured connectivity: every hexahedron of an nx x ny x nz lattice is
  split into 6 Kuhn tetrahedra => nc = 6 nx ny nz cells with <= 4 faces each, nf ~ 2 nc.  Node coordinates are
  jittered, permeability is log-uniform per cell, transmissibilities follow the reference's TPFA formulas
  (discretization/finite-volume.jl:220-233) and the cell numbering is scrambled by a seeded permutation so that the
  input ordering is genuinely unstructured.
"""
import itertools

import numpy as np


def cartesian_neighbors(dims):
    """N (2 x nf, 1-based) of CartesianMesh(dims): x-faces (z, y, x<nx), y-faces (y<ny, z, x), z-faces (z<nz, y, x)."""
    d3 = tuple(dims) + (1,) * (3 - len(dims))
    nx, ny, nz = d3
    idx = np.arange(1, nx * ny * nz + 1, dtype=np.int64).reshape(nz, ny, nx)  # [z, y, x]
    parts = []
    if nx > 1:
        parts.append(np.stack([idx[:, :, :-1].reshape(-1), idx[:, :, 1:].reshape(-1)]))  # loops z, y, x
    if ny > 1:
        l = np.transpose(idx[:, :-1, :], (1, 0, 2)).reshape(-1)  # loops y, z, x
        r = np.transpose(idx[:, 1:, :], (1, 0, 2)).reshape(-1)
        parts.append(np.stack([l, r]))
    if nz > 1:
        parts.append(np.stack([idx[:-1, :, :].reshape(-1), idx[1:, :, :].reshape(-1)]))  # loops z, y, x
    if not parts:
        return np.zeros((2, 0), dtype=np.int64)
    return np.ascontiguousarray(np.concatenate(parts, axis=1))


_PERMS = list(itertools.permutations(range(3)))  # 6 Kuhn tetrahedra per hexahedron
_PIDX = {p: i for i, p in enumerate(_PERMS)}


def tet_lattice_mesh(nx, ny, nz, *, jitter=0.2, seed_perm=20260928, seed_jitter=1, seed_perm_k=2, scramble=True,
                     k_range=(1e-14, 1e-12), size=None, dtype=np.float64):
    """Returns dict(N, nc, nf, T, volumes, cell_centroids, perm_k, areas, ...).  All index arrays 1-based."""
    nh = nx * ny * nz
    nc = 6 * nh
    if size is None:
        size = (float(nx), float(ny), float(nz))
    hx = np.array([size[0] / nx, size[1] / ny, size[2] / nz])
    # ---- nodes -------------------------------------------------------------------------------------------
    gx, gy, gz = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    nodes = np.stack([gx, gy, gz], axis=-1).astype(dtype) * hx  # [nx+1, ny+1, nz+1, 3]
    rng = np.random.default_rng(seed_jitter)
    nodes = nodes + rng.uniform(-jitter, jitter, nodes.shape) * hx

    def node_id(i, j, k):
        return (i * (ny + 1) + j) * (nz + 1) + k

    P = nodes.reshape(-1, 3)
    hi, hj, hk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    hi, hj, hk = hi.reshape(-1), hj.reshape(-1), hk.reshape(-1)
    hex_id = (hk * ny + hj) * nx + hi  # x fastest
    order = np.argsort(hex_id)
    hi, hj, hk = hi[order], hj[order], hk[order]  # arrays indexed by hex id
    base = np.stack([hi, hj, hk], axis=1)  # [nh, 3]

    def vert(off):  # node ids of base + off for every hex
        return node_id(hi + off[0], hj + off[1], hk + off[2])

    # vertices of the 6 tets of every hex: v0, v1 = v0+e_a, v2 = v1+e_b, v3 = v0+(1,1,1)
    tet_v = np.zeros((nh, 6, 4), dtype=np.int64)
    for pi_, (a, b, c) in enumerate(_PERMS):
        e = np.eye(3, dtype=np.int64)
        tet_v[:, pi_, 0] = vert((0, 0, 0))
        tet_v[:, pi_, 1] = vert(tuple(e[a]))
        tet_v[:, pi_, 2] = vert(tuple(e[a] + e[b]))
        tet_v[:, pi_, 3] = vert((1, 1, 1))
    cell_of = (np.arange(nh)[:, None] * 6 + np.arange(6)[None, :])  # 0-based cell ids [nh, 6]
    V = P[tet_v]  # [nh, 6, 4, 3]
    cc = V.mean(axis=2).reshape(nc, 3)
    d1, d2, d3 = V[:, :, 1] - V[:, :, 0], V[:, :, 2] - V[:, :, 0], V[:, :, 3] - V[:, :, 0]
    vol = np.abs(np.einsum("hpi,hpi->hp", np.cross(d1, d2), d3)).reshape(nc) / 6.0
    # ---- faces ----------------------------------------------------------------------------------------------
    left, right, fnodes = [], [], []
    for pi_, p in enumerate(_PERMS):  # interior faces of a hex: adjacent transpositions
        a, b, c = p
        for q, shared in (((b, a, c), (0, 2, 3)), ((a, c, b), (0, 1, 3))):
            qi = _PIDX[q]
            if qi > pi_:
                left.append(cell_of[:, pi_])
                right.append(cell_of[:, qi])
                fnodes.append(tet_v[:, pi_][:, list(shared)])
    dims = (nx, ny, nz)
    stride = (1, nx, nx * ny)
    for ax in range(3):  # faces between neighbouring hexes: tet (a,b,c) of h  <->  tet (b,c,a) of h + e_a
        ok = base[:, ax] < dims[ax] - 1
        hsel = np.nonzero(ok)[0]
        for pi_, p in enumerate(_PERMS):
            if p[0] != ax:
                continue
            qi = _PIDX[(p[1], p[2], p[0])]
            left.append(cell_of[hsel, pi_])
            right.append(cell_of[hsel + stride[ax], qi])
            fnodes.append(tet_v[hsel, pi_][:, [1, 2, 3]])
    l = np.concatenate(left)
    r = np.concatenate(right)
    fn = np.concatenate(fnodes)
    nf = l.size
    F = P[fn]  # [nf, 3, 3]
    fc = F.mean(axis=1)
    nrm = np.cross(F[:, 1] - F[:, 0], F[:, 2] - F[:, 0])
    area = 0.5 * np.linalg.norm(nrm, axis=1)
    nhat = nrm / (2.0 * area[:, None])
    flip = np.einsum("fi,fi->f", nhat, cc[r] - cc[l]) < 0
    nhat[flip] *= -1.0  # normals point from N[1,f] to N[2,f]
    # ---- transmissibilities (half_face_trans + harmonic average, finite-volume.jl:220-233) -----------------------
    rk = np.random.default_rng(seed_perm_k)
    K = np.exp(rk.uniform(np.log(k_range[0]), np.log(k_range[1]), nc))
    Cl, Cr = fc - cc[l], fc - cc[r]
    Tl = area * K[l] * np.einsum("fi,fi->f", Cl, nhat) / np.einsum("fi,fi->f", Cl, Cl)
    Tr = area * K[r] * np.einsum("fi,fi->f", Cr, -nhat) / np.einsum("fi,fi->f", Cr, Cr)
    if np.any(Tl <= 0) or np.any(Tr <= 0):
        raise ValueError("jitter too large: non-positive half-face transmissibility")
    T = 1.0 / (1.0 / Tl + 1.0 / Tr)
    # ---- scramble the cell numbering ------------------------------------------------------------------------------
    if scramble:
        rp = np.random.default_rng(seed_perm)
        new_of_old = rp.permutation(nc)
    else:
        new_of_old = np.arange(nc)
    N = np.stack([new_of_old[l], new_of_old[r]]).astype(np.int64) + 1
    inv = np.empty(nc, dtype=np.int64)
    inv[new_of_old] = np.arange(nc)
    return dict(N=np.ascontiguousarray(N), nc=nc, nf=nf, T=T, volumes=vol[inv], cell_centroids=cc[inv].T.copy(),
                perm_k=K[inv], areas=area, normals=nhat.T.copy(), face_centroids=fc.T.copy(), dim=3)


def cartesian_mesh(nx, ny, nz, *, size=None, seed_perm=20260928, seed_perm_k=2, scramble=True, k_range=(1e-14, 1e-12)):
    """CartesianMesh((nx, ny, nz), size) as the kernels see it (the reference's most common grid: meshes/cart.jl): neighbourship in
    MRST face order, half-face transmissibilities A K / (h/2) of the regular hexahedra (finite-volume.jl:220-222 with C along the
    face normal) and their harmonic average (:224-233), log-uniform permeability per cell, optionally scrambled numbering.
    Six faces per interior cell: rows of 7 entries, triangle-free pattern."""
    nc = nx * ny * nz
    if size is None:
        size = (float(nx), float(ny), float(nz))
    h = np.array([size[0] / nx, size[1] / ny, size[2] / nz])
    N0 = cartesian_neighbors((nx, ny, nz)) - 1  # 0-based, natural numbering
    nfx, nfy = (nx - 1) * ny * nz, nx * (ny - 1) * nz
    nf = N0.shape[1]
    axis = np.zeros(nf, dtype=np.int64)
    axis[nfx:nfx + nfy] = 1
    axis[nfx + nfy:] = 2
    area = np.array([h[1] * h[2], h[0] * h[2], h[0] * h[1]])[axis]
    K = np.exp(np.random.default_rng(seed_perm_k).uniform(np.log(k_range[0]), np.log(k_range[1]), nc))
    half = 0.5 * h[axis]
    Tl, Tr = area * K[N0[0]] / half, area * K[N0[1]] / half
    T = 1.0 / (1.0 / Tl + 1.0 / Tr)
    new_of_old = _scramble(nc, seed_perm, scramble)
    N = np.stack([new_of_old[N0[0]], new_of_old[N0[1]]]).astype(np.int64) + 1
    inv = np.empty(nc, dtype=np.int64)
    inv[new_of_old] = np.arange(nc)
    k, j, i = np.unravel_index(np.arange(nc), (nz, ny, nx))
    cc = (np.stack([i, j, k]).astype(np.float64) + 0.5) * h[:, None]
    return dict(N=np.ascontiguousarray(N), nc=nc, nf=nf, T=T, volumes=np.full(nc, float(np.prod(h))), cell_centroids=cc[:, inv].copy(),
                perm_k=K[inv], areas=area, dim=3)


def _graded_points(n_points, seed, grading):
    """Seeded points in the unit cube; grading > 1 warps them towards the corner at the origin (x -> x**grading per axis), so
    that the local spacing varies by ~grading * n^(1/3)."""
    rng = np.random.default_rng(seed)
    P = rng.random((int(n_points), 3))
    return P ** float(grading) if grading != 1.0 else P


def _scramble(nc, seed, scramble):
    return np.random.default_rng(seed).permutation(nc) if scramble else np.arange(nc)


def delaunay_tet_mesh(n_points, *, seed=7, grading=1.0, seed_perm=20260928, seed_perm_k=2, scramble=True, k_range=(1e-14, 1e-12)):
    """A genuinely unstructured tetrahedral grid (the kind the reference reads through its Gmsh extension,
    ext/JutulGmshExt/interface.jl): the Delaunay triangulation of seeded random (optionally graded) points.  Cells = tets
    (~6.7 per point), <= 4 faces per cell, but -- unlike the Kuhn lattice -- the dual graph has odd cycles (5 tets round an
    edge are common), the vertex valence varies and so do the cell sizes.  TPFA transmissibilities follow
    finite-volume.jl:220-233; the grid is not K-orthogonal, so half-face transmissibilities are taken by magnitude and floored
    (synthetic data: the kernels need positive coefficients, not a convergent discretisation).  Same keys as
    tet_lattice_mesh."""
    from scipy.spatial import Delaunay
    P = _graded_points(n_points, seed, grading)
    tri = Delaunay(P)
    tets = tri.simplices.astype(np.int64)            # [nc, 4]
    nbr = tri.neighbors.astype(np.int64)             # [nc, 4]: neighbour opposite vertex j, -1 on the hull
    nc = tets.shape[0]
    V = P[tets]                                       # [nc, 4, 3]
    cc = V.mean(axis=1)
    vol = np.abs(np.einsum("ci,ci->c", np.cross(V[:, 1] - V[:, 0], V[:, 2] - V[:, 0]), V[:, 3] - V[:, 0])) / 6.0
    vol = np.maximum(vol, 1e-9 * np.median(vol))      # hull slivers
    ci, cj = np.nonzero(nbr >= 0)
    other = nbr[ci, cj]
    keep = ci < other                                 # every interior face once
    l, r, opp = ci[keep], other[keep], cj[keep]
    nf = l.size
    fidx = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])[opp]      # the three vertices of the face opposite vertex `opp`
    F = P[np.take_along_axis(tets[l], fidx, axis=1)]  # [nf, 3, 3]
    fc = F.mean(axis=1)
    nrm = np.cross(F[:, 1] - F[:, 0], F[:, 2] - F[:, 0])
    area = 0.5 * np.linalg.norm(nrm, axis=1)
    area = np.maximum(area, 1e-12 * np.median(area))
    nhat = nrm / np.maximum(np.linalg.norm(nrm, axis=1), 1e-300)[:, None]
    flip = np.einsum("fi,fi->f", nhat, cc[r] - cc[l]) < 0
    nhat[flip] *= -1.0
    K = np.exp(np.random.default_rng(seed_perm_k).uniform(np.log(k_range[0]), np.log(k_range[1]), nc))
    Cl, Cr = fc - cc[l], fc - cc[r]
    Tl = area * K[l] * np.abs(np.einsum("fi,fi->f", Cl, nhat)) / np.maximum(np.einsum("fi,fi->f", Cl, Cl), 1e-300)
    Tr = area * K[r] * np.abs(np.einsum("fi,fi->f", Cr, nhat)) / np.maximum(np.einsum("fi,fi->f", Cr, Cr), 1e-300)
    floor = 1e-6 * np.median(np.concatenate([Tl, Tr]))
    Tl, Tr = np.maximum(Tl, floor), np.maximum(Tr, floor)
    T = 1.0 / (1.0 / Tl + 1.0 / Tr)
    new_of_old = _scramble(nc, seed_perm, scramble)
    N = np.stack([new_of_old[l], new_of_old[r]]).astype(np.int64) + 1
    inv = np.empty(nc, dtype=np.int64)
    inv[new_of_old] = np.arange(nc)
    return dict(N=np.ascontiguousarray(N), nc=nc, nf=nf, T=T, volumes=vol[inv], cell_centroids=cc[inv].T.copy(), perm_k=K[inv],
                areas=area, normals=nhat.T.copy(), face_centroids=fc.T.copy(), dim=3, points=n_points)


def polyhedral_dual_mesh(n_points, *, seed=7, grading=1.0, seed_perm=20260928, seed_perm_k=2, scramble=True, k_range=(1e-14, 1e-12)):
    """A polyhedral grid with MANY faces per cell: the median dual of the Delaunay tet mesh of seeded points (one polyhedral
    cell per point, one face per Delaunay edge -- the connectivity of a Voronoi / PEBI grid): ~15 faces per cell on average,
    up to 30 and more, i.e. Jacobian rows far beyond the <= 8 entries of tet / hex grids.  Cell volume = a quarter of the
    attached tets, face area from the dual 'diamond' of its edge (sum over the tets round the edge of V_t / (2 |e|)),
    T_f = harmonic K * A_f / |e|.  Same keys as tet_lattice_mesh."""
    from scipy.spatial import Delaunay
    P = _graded_points(n_points, seed, grading)
    tets = Delaunay(P).simplices.astype(np.int64)
    nc = P.shape[0]
    V = P[tets]
    tvol = np.abs(np.einsum("ci,ci->c", np.cross(V[:, 1] - V[:, 0], V[:, 2] - V[:, 0]), V[:, 3] - V[:, 0])) / 6.0
    vol = np.bincount(tets.reshape(-1), weights=np.repeat(tvol / 4.0, 4), minlength=nc)
    vol = np.maximum(vol, 1e-9 * np.median(vol))
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
    a = tets[:, pairs[:, 0]].reshape(-1)
    b = tets[:, pairs[:, 1]].reshape(-1)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    # unique edges in (lo, hi) order with the tet volumes round each edge summed: a COO -> CSR conversion (counting sort in C)
    # instead of np.unique over 6 * ntet keys (50 s at 2M tets)
    import scipy.sparse as sp
    E = sp.coo_matrix((np.repeat(tvol, 6), (lo, hi)), shape=(nc, nc)).tocsr()
    E.sum_duplicates()
    E.sort_indices()
    l, r = np.repeat(np.arange(nc, dtype=np.int64), np.diff(E.indptr)), E.indices.astype(np.int64)
    nf = l.size
    elen = np.linalg.norm(P[r] - P[l], axis=1)
    area = E.data / (2.0 * np.maximum(elen, 1e-300))
    area = np.maximum(area, 1e-12 * np.median(area))
    K = np.exp(np.random.default_rng(seed_perm_k).uniform(np.log(k_range[0]), np.log(k_range[1]), nc))
    T = 2.0 * K[l] * K[r] / (K[l] + K[r]) * area / np.maximum(elen, 1e-300)
    T = np.maximum(T, 1e-6 * np.median(T))
    new_of_old = _scramble(nc, seed_perm, scramble)
    N = np.stack([new_of_old[l], new_of_old[r]]).astype(np.int64) + 1
    inv = np.empty(nc, dtype=np.int64)
    inv[new_of_old] = np.arange(nc)
    return dict(N=np.ascontiguousarray(N), nc=nc, nf=nf, T=T, volumes=vol[inv], cell_centroids=P[inv].T.copy(), perm_k=K[inv],
                areas=area, dim=3, points=n_points)
