"""ctypes wrapper around oracle/libjutul_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (jutul.jl_amd) never does.  All index arrays hold the reference's 1-based Int64 values.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

I64P = C.POINTER(C.c_int64)
F64P = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libjutul_oracle.so")
    src = os.path.join(_HERE, "jutul_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjutul_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libjutul_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.jo_ilu0_csr.restype = C.c_void_p
        _LIB.jo_law_create.restype = C.c_void_p
        _LIB.jo_partition_boundary.restype = C.c_int64
        _LIB.jo_alignment_linear_index.restype = C.c_int64
        _LIB.jo_find_sparse_position_csr.restype = C.c_int64
    return _LIB


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pi(a):
    return a.ctypes.data_as(I64P) if a is not None else None


def _pf(a):
    return a.ctypes.data_as(F64P) if a is not None else None


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


# ---------------------------------------------------------------------------------------------------
# Cartesian mesh geometry: tpfv_geometry(::CartesianMesh) (src/meshes/cart.jl:146-302), MRST face order
# ---------------------------------------------------------------------------------------------------
def cartesian_geometry(dims, size=None):
    """dims=(nx,[ny,[nz]]) cells, size=physical extent per dim (cart.jl:30-70: deltas = size/dims).
    Returns dict with N [2,nf] (1-based), areas, volumes, normals [d,nf], cell_centroids [d,nc],
    face_centroids [d,nf], plus boundary_* arrays."""
    d = len(dims)
    dims3 = tuple(dims) + (1,) * (3 - d)
    if size is None:
        size = (1.0,) * d
    size3 = tuple(size) + (1.0,) * (3 - d)
    nx, ny, nz = dims3
    delta = [size3[i] / dims3[i] for i in range(3)]
    nc = nx * ny * nz

    def cidx(x, y, z):  # 1-based ijk -> 1-based linear, cart.jl:113-117
        return (z - 1) * nx * ny + (y - 1) * nx + x

    V = np.zeros(nc)
    cc = np.zeros((d, nc))
    for x in range(1, nx + 1):
        for y in range(1, ny + 1):
            for z in range(1, nz + 1):
                c = cidx(x, y, z)
                V[c - 1] = delta[0] * delta[1] * delta[2]
                pos = (x, y, z)
                for i in range(d):
                    cc[i, c - 1] = (pos[i] - 1) * delta[i] + delta[i] / 2
    nf = (nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1)
    N = np.zeros((2, nf), dtype=np.int64)
    areas = np.zeros(nf)
    fn = np.zeros((d, nf))
    fc = np.zeros((d, nf))
    pos = 0

    def add_face(x, y, z, D):
        nonlocal pos
        idx = cidx(x, y, z)
        N[0, pos] = idx
        N[1, pos] = cidx(x + (D == 1), y + (D == 2), z + (D == 3))
        A = 1.0
        for i in range(3):
            if i != D - 1:
                A *= delta[i]
        areas[pos] = A
        if D - 1 < d:
            fn[D - 1, pos] = 1.0
        fc[:, pos] = cc[:, idx - 1]
        if D - 1 < d:
            fc[D - 1, pos] += delta[D - 1] / 2.0
        pos += 1

    for z in range(1, nz + 1):
        for y in range(1, ny + 1):
            for x in range(1, nx):
                add_face(x, y, z, 1)
    for y in range(1, ny):
        for z in range(1, nz + 1):
            for x in range(1, nx + 1):
                add_face(x, y, z, 2)
    for z in range(1, nz):
        for y in range(1, ny + 1):
            for x in range(1, nx + 1):
                add_face(x, y, z, 3)
    assert pos == nf
    # boundary faces (cart.jl:228-283)
    bN, bA, bn, bc = [], [], [], []

    def add_bnd(x, y, z, D, is_start):
        idx = cidx(x, y, z)
        A = 1.0
        for i in range(3):
            if i != D - 1:
                A *= delta[i]
        sgn = -1.0 if is_start else 1.0
        n = np.zeros(d)
        n[D - 1] = sgn
        cen = cc[:, idx - 1].copy()
        cen[D - 1] += sgn * delta[D - 1] / 2.0
        bN.append(idx), bA.append(A), bn.append(n), bc.append(cen)

    for y in range(1, ny + 1):
        for z in range(1, nz + 1):
            for (x, st) in [(1, True), (nx, False)]:
                add_bnd(x, y, z, 1, st)
    if d > 1:
        for x in range(1, nx + 1):
            for z in range(1, nz + 1):
                for (y, st) in [(1, True), (ny, False)]:
                    add_bnd(x, y, z, 2, st)
        if d > 2:
            for x in range(1, nx + 1):
                for y in range(1, ny + 1):
                    for (z, st) in [(1, True), (nz, False)]:
                        add_bnd(x, y, z, 3, st)
    return dict(N=N, areas=areas, volumes=V, normals=fn, cell_centroids=cc, face_centroids=fc, nc=nc, nf=nf,
                dim=d, boundary_neighbors=np.array(bN, dtype=np.int64), boundary_areas=np.array(bA),
                boundary_normals=np.array(bn).T.copy(), boundary_centroids=np.array(bc).T.copy())


# ---------------------------------------------------------------------------------------------------
# connectivity / transmissibility / pattern
# ---------------------------------------------------------------------------------------------------
def get_facepos(N, nc):
    N = _i(np.asfortranarray(N).T.reshape(-1))  # column-major 2 x nf flattened
    nf = N.size // 2
    faces = np.zeros(2 * nf, dtype=np.int64)
    facepos = np.zeros(nc + 1, dtype=np.int64)
    _ck(lib().jo_get_facepos(_pi(N), C.c_int64(nf), C.c_int64(nc), _pi(faces), _pi(facepos)), "get_facepos")
    return faces, facepos


def half_face_map(N, nc):
    """(cells, faces, face_pos, face_sign) as in domains.jl:101-122 plus `self`."""
    faces, facepos = get_facepos(N, nc)
    Nf = _i(np.asfortranarray(N).T.reshape(-1))
    nhf = faces.size
    self_ = np.zeros(nhf, dtype=np.int64)
    other = np.zeros(nhf, dtype=np.int64)
    sign = np.zeros(nhf, dtype=np.int64)
    _ck(lib().jo_half_face_map(_pi(Nf), C.c_int64(nhf // 2), C.c_int64(nc), _pi(faces), _pi(facepos), _pi(self_),
                               _pi(other), _pi(sign)), "half_face_map")
    return dict(self=self_, cells=other, other=other, faces=faces, face_pos=facepos, face_sign=sign)


def half_face_trans(geo, perm, hfm):
    cc = _f(np.asfortranarray(geo["cell_centroids"]).T.reshape(-1))
    fc = _f(np.asfortranarray(geo["face_centroids"]).T.reshape(-1))
    fn = _f(np.asfortranarray(geo["normals"]).T.reshape(-1))
    areas = _f(geo["areas"])
    perm = np.atleast_2d(np.asarray(perm, dtype=np.float64))
    if perm.shape[1] != geo["nc"]:
        perm = perm.T
    npm = perm.shape[0]
    permf = _f(np.asfortranarray(perm).T.reshape(-1))
    T = np.zeros(hfm["faces"].size)
    _ck(lib().jo_half_face_trans(C.c_int64(geo["nc"]), C.c_int(geo["dim"]), _pf(cc), _pf(fc), _pf(fn), _pf(areas),
                                 _pf(permf), C.c_int(npm), _pi(hfm["faces"]), _pi(hfm["face_pos"]),
                                 _pi(hfm["face_sign"]), _pf(T)), "half_face_trans")
    return T


def face_trans(T_hf, faces, nf):
    T = np.zeros(nf)
    _ck(lib().jo_face_trans(C.c_int64(nf), C.c_int64(T_hf.size), _pf(_f(T_hf)), _pi(_i(faces)), _pf(T)), "face_trans")
    return T


def face_gdz(N, z, g=9.80665):
    Nf = _i(np.asfortranarray(N).T.reshape(-1))
    nf = Nf.size // 2
    out = np.zeros(nf)
    _ck(lib().jo_face_gdz(_pi(Nf), C.c_int64(nf), _pf(_f(z)), C.c_double(g), _pf(out)), "face_gdz")
    return out


def csr_pattern(nc, hfm):
    rowptr = np.zeros(nc + 1, dtype=np.int64)
    _ck(lib().jo_csr_pattern(C.c_int64(nc), _pi(hfm["face_pos"]), _pi(hfm["other"]), _pi(rowptr), None), "pattern")
    colidx = np.zeros(rowptr[-1] - 1, dtype=np.int64)
    _ck(lib().jo_csr_pattern(C.c_int64(nc), _pi(hfm["face_pos"]), _pi(hfm["other"]), _pi(rowptr), _pi(colidx)),
        "pattern")
    return rowptr, colidx


def csr_pattern_scalar(nc, N, layout, b_rowptr, b_colidx):
    rowptr = np.zeros(nc * N + 1, dtype=np.int64)
    colidx = np.zeros((b_rowptr[-1] - 1) * N * N, dtype=np.int64)
    _ck(lib().jo_csr_pattern_scalar(C.c_int64(nc), C.c_int(N), C.c_int(layout), _pi(b_rowptr), _pi(b_colidx),
                                    _pi(rowptr), _pi(colidx)), "pattern_scalar")
    return rowptr, colidx


LAYOUT = dict(equation_major=0, entity_major=1, block_major=2)


def align(nc, N, layout, rowptr, colidx, hfm, s_rowptr=None, s_colidx=None, active=None):
    nhf = hfm["faces"].size
    pos_acc = np.zeros(N * N * nc, dtype=np.int64)
    pos_flux = np.zeros(N * N * nhf, dtype=np.int64)
    act = _i(active) if active is not None else None
    _ck(lib().jo_align(C.c_int64(nc), C.c_int64(nhf), C.c_int(N), C.c_int(layout), _pi(rowptr), _pi(colidx),
                       _pi(s_rowptr), _pi(s_colidx), _pi(hfm["self"]), _pi(hfm["other"]), _pi(act), _pi(pos_acc),
                       _pi(pos_flux)), "align")
    return pos_acc.reshape(nc, N * N).T.copy(), pos_flux.reshape(nhf, N * N).T.copy()


# ---------------------------------------------------------------------------------------------------
# law / assembly
# ---------------------------------------------------------------------------------------------------
KIND = dict(poisson=0, compressible=1, twophase=2)


class Law:
    def __init__(self, kind, dt, rho0=(1.0, 1.0), comp=(0.0, 0.0), mu=(1.0, 1.0), p_ref=0.0):
        self.kind = KIND[kind] if isinstance(kind, str) else kind
        self.N = 2 if self.kind == 2 else 1
        par = _f(list(rho0) + list(comp) + list(mu) + [p_ref])
        self.h = C.c_void_p(lib().jo_law_create(C.c_int(self.kind), C.c_double(dt), _pf(par)))
        self.dt = dt

    def set_dt(self, dt):
        self.dt = dt
        lib().jo_law_set_dt(self.h, C.c_double(dt))

    def __del__(self):
        try:
            lib().jo_law_free(self.h)
        except Exception:
            pass


def _zero(a):
    lib().jo_zero(C.c_int64(a.size), a.ctypes.data_as(C.POINTER(C.c_double)))
    return a


def update_equation(law, nc, hfm, X, X0, vol, Tf, gdz=None, out=None):
    """a-5.  Returns (acc [nc, N, 1+N], hf [nhf, N, 1+N]) Dual arrays.  out = (acc, hf) of an earlier call: the caches are kept
    between Newton iterations like the reference's ConservationLawTPFAStorage (conservation.jl:101-135), zeroed on all cores."""
    N = law.N
    nhf = hfm["faces"].size
    if out is None:
        acc = np.zeros((nc, N, 1 + N))
        hf = np.zeros((nhf, N, 1 + N))
    else:
        acc, hf = _zero(out[0]), _zero(out[1])
    X, X0, vol, Tf = _f(X), _f(X0), _f(vol), _f(Tf)
    g = _f(gdz) if gdz is not None else None
    _ck(lib().jo_update_equation(law.h, C.c_int64(nc), C.c_int64(nhf), _pi(hfm["face_pos"]), _pi(hfm["self"]),
                                 _pi(hfm["other"]), _pi(hfm["faces"]), _pi(hfm["face_sign"]), _pf(X), _pf(X0),
                                 _pf(vol), _pf(Tf), _pf(g), _pf(acc), _pf(hf)), "update_equation")
    return acc, hf


def apply_sources(acc, cells, values):
    N = acc.shape[1]
    cells = _i(cells)
    values = _f(np.asarray(values, dtype=np.float64).reshape(cells.size, N))
    _ck(lib().jo_apply_sources(C.c_int(N), C.c_int64(cells.size), _pi(cells), _pf(values), _pf(acc)), "sources")


def fill_conservation_eq(nc, N, hfm, acc, hf, pos_acc, pos_flux, nnz_flat, equation_major=False, out=None, flat_pos=None):
    """a-6.  Returns (nz flat [nnz_flat], r flat [N*nc]) in the requested residual layout.  out = (nz, r) buffers to reuse
    (jac_buffer / r_buffer live as long as the LinearizedSystem, linsolve/default.jl:166-226); flat_pos = the flattened
    position tables of an earlier call."""
    if out is None:
        nz = np.zeros(nnz_flat)
        r = np.zeros(N * nc)
    else:
        nz, r = _zero(out[0]), _zero(out[1])
    if flat_pos is None:
        pa = _i(pos_acc.T.reshape(-1))
        pf = _i(pos_flux.T.reshape(-1))
    else:
        pa, pf = flat_pos
    se, sc = (nc, 1) if equation_major else (1, N)
    _ck(lib().jo_fill_conservation_eq(C.c_int(N), C.c_int64(nc), _pi(hfm["face_pos"]), _pf(acc), _pf(hf), _pi(pa),
                                      _pi(pf), _pf(nz), _pf(r), C.c_int64(se), C.c_int64(sc)), "fill")
    return nz, r


GEN_LAW = {"compressible": 1, "reaction": 2}


def generic_cell_assemble(law, N, nc, hfm, rowptr, colidx, X, X0, vol, Tf, gdz, dt, par, src_cells=(), src_values=()):
    """a-7 (i): cell-based GenericAutoDiffCache path (equations.jl:578-594, ad/generic.jl:53-96) for the user laws of the device
    tests.  Returns (nz flat block-major, r [N, nc] flat)."""
    nz = np.zeros(int(rowptr[-1] - 1) * N * N)
    r = np.zeros(N * nc)
    sc, sv = _i(src_cells), _f(np.asarray(src_values, dtype=np.float64).reshape(-1))
    g = _f(gdz) if gdz is not None else None
    _ck(lib().jo_generic_cell_assemble(C.c_int(GEN_LAW[law]), C.c_int(N), C.c_int64(nc), _pi(hfm["face_pos"]), _pi(hfm["other"]),
                                       _pi(hfm["faces"]), _pi(hfm["face_sign"]), _pi(_i(rowptr)), _pi(_i(colidx)), _pf(_f(X)), _pf(_f(X0)),
                                       _pf(_f(vol)), _pf(_f(Tf)), _pf(g), C.c_double(dt), _pf(_f(par)), C.c_int64(sc.size), _pi(sc), _pf(sv),
                                       _pf(nz), _pf(r)), "generic_cell_assemble")
    return nz, r


def fvm_face_assemble(law, N, nc, Nlr, rowptr, colidx, X, X0, vol, Tf, gdz, dt, par, src_cells=(), src_values=()):
    """a-7 (ii): face-based PotentialFlow{:fvm} assembly (conservation/fvm_assembly.jl:175-283)."""
    Nlr = np.asarray(Nlr, dtype=np.int64)
    nf = Nlr.shape[1]
    nz = np.zeros(int(rowptr[-1] - 1) * N * N)
    r = np.zeros(N * nc)
    sc, sv = _i(src_cells), _f(np.asarray(src_values, dtype=np.float64).reshape(-1))
    g = _f(gdz) if gdz is not None else None
    Nf = _i(np.asfortranarray(Nlr).T.reshape(-1))
    _ck(lib().jo_fvm_face_assemble(C.c_int(GEN_LAW[law]), C.c_int(N), C.c_int64(nc), C.c_int64(nf), _pi(Nf), _pi(_i(rowptr)), _pi(_i(colidx)),
                                   _pf(_f(X)), _pf(_f(X0)), _pf(_f(vol)), _pf(_f(Tf)), _pf(g), C.c_double(dt), _pf(_f(par)),
                                   C.c_int64(sc.size), _pi(sc), _pf(sv), _pf(nz), _pf(r)), "fvm_face_assemble")
    return nz, r


def convergence(N, nc, r, equation_major=False):
    e = np.zeros(N)
    se, sc = (nc, 1) if equation_major else (1, N)
    lib().jo_convergence(C.c_int(N), C.c_int64(nc), _pf(_f(r)), C.c_int64(se), C.c_int64(sc), _pf(e))
    return e


# ---------------------------------------------------------------------------------------------------
# sparse kernels
# ---------------------------------------------------------------------------------------------------
def spmv(n, bs, rowptr, colidx, nz, x, y=None, alpha=1.0, beta=0.0):
    y = np.zeros(n * bs) if y is None else _f(y).copy()
    _ck(lib().jo_spmv(C.c_int64(n), C.c_int(bs), _pi(rowptr), _pi(colidx), _pf(_f(nz)), _pf(_f(x)), _pf(y),
                      C.c_double(alpha), C.c_double(beta)), "spmv")
    return y


class ILU0:
    """ilu0_csr(A[, partition]) / ilu0_csr! / ldiv! (StaticCSR/ilu0.jl, par_ilu0.jl)."""

    def __init__(self, n, bs, rowptr, colidx, nz, partition=None):
        self.n, self.bs = n, bs
        self.rowptr, self.colidx = _i(rowptr), _i(colidx)
        part = _i(partition) if partition is not None else None
        h = lib().jo_ilu0_csr(C.c_int64(n), C.c_int(bs), _pi(self.rowptr), _pi(self.colidx), _pf(_f(nz)), _pi(part))
        if not h:
            raise RuntimeError("ilu0_csr failed (missing diagonal or bad partition)")
        self.h = C.c_void_p(h)

    def refactor(self, nz):
        _ck(lib().jo_ilu0_refactor(self.h, _pf(_f(nz))), "ilu0_refactor")

    def apply(self, b):
        b = _f(b)
        x = np.zeros_like(b)
        _ck(lib().jo_ilu0_apply(self.h, _pf(x), _pf(b)), "ilu0_apply")
        return x

    def export(self, nnz_flat):
        lu = np.zeros(nnz_flat)
        _ck(lib().jo_ilu0_export(self.h, _pf(lu)), "ilu0_export")
        return lu

    def __del__(self):
        try:
            lib().jo_ilu0_free(self.h)
        except Exception:
            pass


SIDE = dict(none=0, left=1, right=2)


def bicgstab(n, bs, rowptr, colidx, nz, b, prec=None, side="right", rtol=1e-3, atol=1e-12, itmax=100):
    x = np.zeros(n * bs)
    iters = C.c_int64(0)
    hist = np.zeros(itmax + 2)
    st = lib().jo_bicgstab(C.c_int64(n), C.c_int(bs), _pi(_i(rowptr)), _pi(_i(colidx)), _pf(_f(nz)),
                           prec.h if prec is not None else None, C.c_int(SIDE[side] if prec is not None else 0),
                           _pf(_f(b)), _pf(x), C.c_double(rtol), C.c_double(atol), C.c_int64(itmax), C.byref(iters),
                           _pf(hist), C.c_int64(hist.size))
    if st < 0:
        raise MemoryError("oracle Krylov solver: workspace allocation failed")
    return x, dict(status=st, solved=(st == 0), iterations=iters.value, residuals=hist[: iters.value + 1].copy())


def gmres(n, bs, rowptr, colidx, nz, b, prec=None, side="right", rtol=1e-3, atol=1e-12, itmax=100):
    x = np.zeros(n * bs)
    iters = C.c_int64(0)
    hist = np.zeros(itmax + 2)
    st = lib().jo_gmres(C.c_int64(n), C.c_int(bs), _pi(_i(rowptr)), _pi(_i(colidx)), _pf(_f(nz)),
                        prec.h if prec is not None else None, C.c_int(SIDE[side] if prec is not None else 0),
                        _pf(_f(b)), _pf(x), C.c_double(rtol), C.c_double(atol), C.c_int64(itmax), C.byref(iters),
                        _pf(hist), C.c_int64(hist.size))
    if st < 0:
        raise MemoryError("oracle Krylov solver: workspace allocation failed")
    return x, dict(status=st, solved=(st == 0), iterations=iters.value, residuals=hist[: iters.value + 1].copy())


# ---------------------------------------------------------------------------------------------------
# partition / distributed helpers
# ---------------------------------------------------------------------------------------------------
def partition_linear(m, n):
    p = np.zeros(n, dtype=np.int64)
    lib().jo_partition_linear(C.c_int64(m), C.c_int64(n), _pi(p))
    return p


def compress_partition(p):
    p = _i(p)
    out = np.zeros_like(p)
    lib().jo_compress_partition(C.c_int64(p.size), _pi(p), _pi(out))
    return out


def partition_boundary(N, p, ip):
    Nf = _i(np.asfortranarray(N).T.reshape(-1))
    p = _i(p)
    bnd = np.zeros(p.size, dtype=np.int64)
    n = lib().jo_partition_boundary(_pi(Nf), C.c_int64(Nf.size // 2), C.c_int64(p.size), _pi(p), C.c_int64(ip),
                                    _pi(bnd))
    return bnd[:n].copy()


def remap_global_indices(p, np_):
    p = _i(p)
    rem = np.zeros_like(p)
    counts = np.zeros(np_, dtype=np.int64)
    lib().jo_remap_global_indices(C.c_int64(p.size), _pi(p), C.c_int64(np_), _pi(rem), _pi(counts))
    return rem, counts


def unit_diagonalize(n, n_self, bs, rowptr, colidx, nz, r):
    nz, r = _f(nz).copy(), _f(r).copy()
    lib().jo_unit_diagonalize(C.c_int64(n), C.c_int64(n_self), C.c_int(bs), _pi(_i(rowptr)), _pi(_i(colidx)),
                              _pf(nz), _pf(r))
    return nz, r


def touch_copy(a):
    """Copy of a float64 / int64 array whose pages are first touched by the OpenMP threads (static schedule), not by numpy."""
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 8
    out = np.empty_like(a)
    lib().jo_touch_copy(C.c_int64(a.size), out.ctypes.data_as(C.POINTER(C.c_double)), a.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def set_num_threads(n):
    lib().jo_set_num_threads(C.c_int(n))


def num_threads():
    return lib().jo_num_threads()


# ---------------------------------------------------------------------------------------------------
# convenience: full setup + one assembly, as the reference's Simulator setup would do it
# ---------------------------------------------------------------------------------------------------
class TPFASystem:
    """Holds a-1..a-4 tables for a neighborship N (2 x nf, 1-based) with block size N_blk (block-major)."""

    def __init__(self, N, nc, nblk=1):
        self.N, self.nc, self.nblk = np.asarray(N, dtype=np.int64), nc, nblk
        self.nf = self.N.shape[1]
        self.hfm = half_face_map(self.N, nc)
        self.rowptr, self.colidx = csr_pattern(nc, self.hfm)
        self.nnzb = int(self.rowptr[-1] - 1)
        self.pos_acc, self.pos_flux = align(nc, nblk, 2, self.rowptr, self.colidx, self.hfm)
        self._buf = None

    def assemble(self, law, X, X0, vol, Tf, gdz=None, src_cells=None, src_values=None, reuse=False):
        """reuse=True keeps the Dual caches, the position tables' flat copies and the jac / r buffers between calls (what the
        reference's storage does); the returned arrays are then overwritten by the next call."""
        if reuse and getattr(self, "_buf", None) is None:
            self._buf = dict(flat_pos=(_i(self.pos_acc.T.reshape(-1)), _i(self.pos_flux.T.reshape(-1))))
        b = self._buf if reuse else None
        acc, hf = update_equation(law, self.nc, self.hfm, X, X0, vol, Tf, gdz, out=b.get("duals") if b else None)
        if src_cells is not None and len(src_cells):
            apply_sources(acc, src_cells, src_values)
        nz, r = fill_conservation_eq(self.nc, self.nblk, self.hfm, acc, hf, self.pos_acc, self.pos_flux,
                                     self.nnzb * self.nblk * self.nblk, out=b.get("sys") if b else None,
                                     flat_pos=b["flat_pos"] if b else None)
        if b is not None:
            b["duals"], b["sys"] = (acc, hf), (nz, r)
        return nz, r


def submap_cells(N, indices, nc, buffer=0, excluded=()):
    """Literal (loop) restatement of submap_cells (dd/subdomains.jl:77-182) without gmap / active_global."""
    N = np.asarray(N, dtype=np.int64)
    faces, facepos = get_facepos(N, nc)
    nf = N.shape[1]
    cell_active = np.zeros(nc + 1, dtype=bool)
    cell_is_bnd = np.zeros(nc + 1, dtype=bool)
    face_active = np.zeros(nf + 1, dtype=bool)
    interior = np.zeros(nc + 1, dtype=bool)
    interior[np.asarray(indices)] = True
    for gc in indices:
        pb = False
        for fi in range(facepos[gc - 1], facepos[gc]):
            face = faces[fi - 1]
            l, r = N[0, face - 1], N[1, face - 1]
            if interior[l] and interior[r]:
                face_active[face] = True
            else:
                pb = True
        cell_active[gc] = True
        cell_is_bnd[gc] = pb and buffer == 0
    if buffer > 0:
        for gc in indices:
            for fi in range(facepos[gc - 1], facepos[gc]):
                face = faces[fi - 1]
                l, r = N[0, face - 1], N[1, face - 1]
                other = r if l == gc else l
                if other in excluded:
                    continue
                face_active[face] = True
                if not cell_active[other] and not interior[other]:
                    cell_active[other] = True
                    cell_is_bnd[other] = True
        cells = np.flatnonzero(cell_active)
        is_b = cell_is_bnd[cells]
    else:
        cells = np.asarray(indices, dtype=np.int64).copy()
        is_b = np.zeros(cells.size, dtype=bool)
    return dict(cells=cells, faces=np.flatnonzero(face_active), is_boundary=is_b)
