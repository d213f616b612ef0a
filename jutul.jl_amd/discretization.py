"""Setup-time discretisation helpers (a-2): TPFA transmissibilities and gravity terms, vectorised numpy restatements of
src/discretization/finite-volume.jl.  Float64, host side, run once per Simulator setup (not on the hot path)."""
import numpy as np


def half_face_map(N, nc):
    """half_face_map(N, nc) (domains.jl:101-122) via get_facepos (utils.jl:813-874): per cell the ascending list of
    incident interior faces; per half-face the neighbour cell and the sign (+1 iff N[1, face] == cell).  1-based."""
    N = np.asarray(N, dtype=np.int64)
    nf = N.shape[1]
    cells = np.concatenate([N[0], N[1]])
    faces = np.concatenate([np.arange(1, nf + 1), np.arange(1, nf + 1)])
    sign = np.concatenate([np.ones(nf, dtype=np.int64), -np.ones(nf, dtype=np.int64)])
    other = np.concatenate([N[1], N[0]])
    order = np.lexsort((faces, cells))  # by cell, then ascending face id (sort! at utils.jl:836-838)
    counts = np.bincount(cells, minlength=nc + 1)[1:]
    face_pos = np.concatenate([[1], 1 + np.cumsum(counts)]).astype(np.int64)
    return dict(cells=other[order], faces=faces[order], face_pos=face_pos, face_sign=sign[order], self=cells[order])


def expand_perm(perm, dim):
    """expand_perm (finite-volume.jl:156-218): [np, nc] -> [nc, dim, dim] symmetric tensors."""
    K = np.atleast_2d(np.asarray(perm, dtype=np.float64))
    npm, nc = K.shape
    out = np.zeros((nc, dim, dim))
    if dim == 1:
        out[:, 0, 0] = K[0]
    elif dim == 2:
        if npm == 1:
            out[:, 0, 0] = out[:, 1, 1] = K[0]
        elif npm == 2:
            out[:, 0, 0], out[:, 1, 1] = K[0], K[1]
        elif npm == 3:
            out[:, 0, 0], out[:, 0, 1], out[:, 1, 0], out[:, 1, 1] = K[0], K[1], K[1], K[2]
        else:
            raise ValueError("Permeability for two-dimensional grids must have 1/2/3 entries per cell")
    else:
        if npm == 1:
            out[:, 0, 0] = out[:, 1, 1] = out[:, 2, 2] = K[0]
        elif npm == 3:
            out[:, 0, 0], out[:, 1, 1], out[:, 2, 2] = K[0], K[1], K[2]
        elif npm == 6:
            out[:, 0, 0], out[:, 0, 1], out[:, 0, 2] = K[0], K[1], K[2]
            out[:, 1, 0], out[:, 1, 1], out[:, 1, 2] = K[1], K[3], K[4]
            out[:, 2, 0], out[:, 2, 1], out[:, 2, 2] = K[2], K[4], K[5]
        else:
            raise ValueError("Permeability for three-dimensional meshes must have 1/3/6 entries per cell")
    return out


def compute_half_face_trans(cell_centroids, face_centroids, face_normals, face_areas, perm, N):
    """compute_half_face_trans (finite-volume.jl:130-154,220-222): T_hf = A * dot(K*C, sgn*n) / dot(C, C), C = x_f - x_c.
    Arrays are [dim, n] like the reference's geometry; returns (T_hf, half_face_map)."""
    cc, fc, fn = (np.asarray(a, dtype=np.float64) for a in (cell_centroids, face_centroids, face_normals))
    dim, nc = cc.shape
    hf = half_face_map(N, nc)
    c, f = hf["self"] - 1, hf["faces"] - 1
    K = expand_perm(perm, dim)[c]  # [nhf, dim, dim]
    C = (fc[:, f] - cc[:, c]).T
    Nn = (fn[:, f] * hf["face_sign"]).T
    KC = np.einsum("hij,hj->hi", K, C)
    T_hf = np.asarray(face_areas, dtype=np.float64)[f] * np.einsum("hi,hi->h", KC, Nn) / np.einsum("hi,hi->h", C, C)
    return T_hf, hf


def compute_face_trans(T_hf, faces, nf):
    """compute_face_trans (finite-volume.jl:224-233): harmonic average accumulated in half-face order."""
    acc = np.zeros(nf)
    np.add.at(acc, np.asarray(faces) - 1, 1.0 / np.asarray(T_hf, dtype=np.float64))
    return 1.0 / acc


def compute_face_gdz(N, z, g=9.80665):
    """compute_face_gdz (finite-volume.jl:304-313): gdz_f = -g (z[r] - z[l])."""
    N = np.asarray(N, dtype=np.int64)
    z = np.asarray(z, dtype=np.float64)
    return -g * (z[N[1] - 1] - z[N[0] - 1])
