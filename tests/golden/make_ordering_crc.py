#!/usr/bin/env python3
"""Writes tests/golden/ordering_crc.json: checksums of the device ordering (perm, block_ptr) and figures of the ILU(0) symbolic phase
for a few seeded grids, taken from the library through a planning context.  The device ordering is NOT reference data (any valid
partition reproduces the reference's semantics); the file pins it so that work on the set-up code (threads, data structures) can
show that it leaves the blocks -- and with them every iteration count measured on the GPU -- exactly as they were.
Regenerate only when the ordering is changed on purpose:  python tests/golden/make_ordering_crc.py
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cases(ja):
    return {
        "lattice_40x39x38_scrambled_weighted": (lambda: ja.tet_lattice_mesh(40, 39, 38, scramble=True), True, 0, 1),
        "lattice_40x39x38_natural_unweighted": (lambda: ja.tet_lattice_mesh(40, 39, 38, scramble=False), False, 0, 1),
        "cartesian_70_scrambled_weighted_2x2": (lambda: ja.cartesian_mesh(70, 70, 70, scramble=True), True, 0, 2),
        "delaunay_60k_weighted": (lambda: ja.delaunay_tet_mesh(60000, grading=2.0), True, 0, 1),
        "polyhedral_40k_weighted": (lambda: ja.polyhedral_dual_mesh(40000, grading=1.5), True, 0, 1),
        "lattice_30_block_rows_128": (lambda: ja.tet_lattice_mesh(30, 30, 30, scramble=True), True, 128, 1),
    }


def figures(ja, ctx, make, weighted, block_rows, block_n):
    g = make()
    d = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], block_n=block_n, reorder="blocks", block_rows=block_rows,
                                          face_weights=g["T"] if weighted else None)
    perm, bp = d.ordering()
    A = ja.StaticSparsityMatrixCSR(d)
    fi = ja.ILUZeroPreconditioner(partition="blocks").symbolic(A).info()
    crc = zlib.crc32(np.ascontiguousarray(bp).tobytes(), zlib.crc32(np.ascontiguousarray(perm).tobytes()))
    return dict(nc=int(g["nc"]), ordering_crc=int(crc), nblocks=fi["nblocks"], max_levels=fi["max_levels"], l_entries=fi["l_entries"],
                jagged_slices=A.spmv_info()["slices"])


if __name__ == "__main__":
    import jutul_amd as ja
    ctx = ja.HIPContext("host")
    out = {name: figures(ja, ctx, *c) for name, c in cases(ja).items()}
    with open(os.path.join(ROOT, "tests", "golden", "ordering_crc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
