"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference for the MRST
fixture data file data/testgrids/pico.mat, which the reference's own test/mesh.jl:49-100 uses).
The .npz files hold DATA only (arrays from the fixture + known answers quoted from the reference's tests);
no reference source text is stored.
"""
import os
import numpy as np
import scipy.io as sio

HERE = os.path.dirname(os.path.abspath(__file__))


def pico():
    m = sio.loadmat("/root/reference/data/testgrids/pico.mat", squeeze_me=True, struct_as_record=False)
    G, rock = m["G"], m["rock"]
    nb = np.asarray(G.faces.neighbors, dtype=np.int64)  # [nfaces_all, 2], 0 = boundary
    interior = (nb[:, 0] > 0) & (nb[:, 1] > 0)
    np.savez(os.path.join(HERE, "pico.npz"),
             neighbors_all=nb,
             N=nb[interior].T.copy(),  # 2 x nf interior neighborship in file order (MRSTWrapMesh tpfv_geometry)
             interior=interior,
             areas=np.asarray(G.faces.areas, dtype=np.float64)[interior],
             normals=np.asarray(G.faces.normals, dtype=np.float64)[interior].T.copy(),
             face_centroids=np.asarray(G.faces.centroids, dtype=np.float64)[interior].T.copy(),
             cell_centroids=np.asarray(G.cells.centroids, dtype=np.float64).T.copy(),
             volumes=np.asarray(G.cells.volumes, dtype=np.float64),
             cells_facePos=np.asarray(G.cells.facePos, dtype=np.int64),
             cells_faces=np.asarray(G.cells.faces, dtype=np.int64)[:, 0],
             cartDims=np.asarray(G.cartDims, dtype=np.int64),
             perm=np.asarray(rock.perm, dtype=np.float64))


def known_answers():
    # values quoted from the reference's tests (file:line in the key comments of tests/test_oracle_kat.py)
    np.savez(os.path.join(HERE, "kat.npz"),
             poisson_3x1=np.array([0.0, 1 / 3, 2 / 3]),                   # test/test_systems/variable_poisson.jl:28-35
             compress_in=np.array([1, 3, 6, 5]), compress_out=np.array([1, 2, 4, 3]),  # test/partitioning.jl:8-9
             layout_eq=np.array([1.0, 2.0, 0.1, 0.2, 0.3, 0.4, 10, 20, 30, 40, 50, 60, 70, 80]),   # test/adjoints/utils.jl:57
             layout_block=np.array([1.0, 0.1, 0.3, 10, 30, 50, 70, 2.0, 0.2, 0.4, 20, 40, 60, 80]),  # :67
             pico_N=np.array([[1, 2, 4, 5, 7, 8, 1, 2, 3, 4, 5, 6], [2, 3, 5, 6, 8, 9, 4, 5, 6, 7, 8, 9]]))


if __name__ == "__main__":
    pico()
    known_answers()
    print("golden written")
