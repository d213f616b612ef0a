"""jh_subdomain_* / jh_partition_rcb (host code of libjutul_hip.so, no GPU) against the numpy statement of the reference's
subdomain construction (tests/_dd_numpy.py; interface.jl:38-63, dd/subdomains.jl:77-182)."""
import numpy as np
import pytest

import jutul_amd as ja
from jutul_amd import dd
from _dd_numpy import local_subdomain_numpy, partition_rcb_numpy


def same_sub(a, b):
    assert a["n_owned"] == b["n_owned"] and a["n_local"] == b["n_local"]
    for k in ("cells", "faces", "N", "neighbors"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert len(a["send"]) == len(b["send"]) == len(a["recv"]) == len(b["recv"])
    for x, y in zip(a["send"] + a["recv"], b["send"] + b["recv"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("order", ["global", "owner"])
@pytest.mark.parametrize("nparts", [2, 5, 8])
def test_subdomain_matches_numpy(order, nparts):
    g = ja.tet_lattice_mesh(9, 8, 7)
    part = dd.partition_rcb(g["cell_centroids"], nparts)
    for r in range(1, nparts + 1):
        same_sub(dd.local_subdomain(g["N"], part, r, ghost_order=order), local_subdomain_numpy(g["N"], part, r, ghost_order=order))


def test_subdomain_random_partition_and_edge_cases():
    g = ja.tet_lattice_mesh(6, 5, 4)
    rng = np.random.default_rng(0)
    part = rng.integers(1, 5, g["nc"])           # scattered parts: many neighbours, ghost-ghost faces
    for r in range(1, 5):
        for order in ("global", "owner"):
            same_sub(dd.local_subdomain(g["N"], part, r, ghost_order=order), local_subdomain_numpy(g["N"], part, r, ghost_order=order))
    one = np.ones(g["nc"], dtype=np.int64)        # a single part: no ghosts, no neighbours
    s = dd.local_subdomain(g["N"], one, 1)
    assert s["n_owned"] == s["n_local"] == g["nc"] and len(s["neighbors"]) == 0 and s["faces"].size == g["nf"]
    two = one.copy()
    two[0] = 3                                    # parts 1 and 3 are populated, part 2 is empty: legal, an empty subdomain
    empty = dd.local_subdomain(g["N"], two, 2)
    assert empty["n_owned"] == 0 and empty["n_local"] == 0 and empty["faces"].size == 0
    with pytest.raises(Exception, match="parts 1..1"):
        dd.local_subdomain(g["N"], one, 2)        # a rank beyond the partition's parts (0- vs 1-based mix-up) is an error
    with pytest.raises(Exception, match="non-finite"):
        dd.partition_rcb(np.full((3, 10), np.nan), 2)
    with pytest.raises(Exception):
        dd.local_subdomain(g["N"], one * 0, 1)    # partition ids are 1-based
    with pytest.raises(ValueError):
        dd.local_subdomain(g["N"], one, 1, ghost_order="x")


@pytest.mark.parametrize("nparts", [1, 2, 3, 7, 8])
def test_rcb_balanced_compact_and_like_numpy(nparts):
    g = ja.tet_lattice_mesh(10, 9, 8)
    X = g["cell_centroids"]
    p = dd.partition_rcb(X, nparts)
    q = partition_rcb_numpy(X, nparts)
    assert p.min() == 1 and p.max() == nparts
    assert np.abs(np.sort(np.bincount(p)[1:]) - np.sort(np.bincount(q)[1:])).max() <= 1   # the same part sizes (up to the rounding of halves)
    # the same cuts up to ties at the medians: the cut of the C partition is within a few percent of numpy's
    N = g["N"]
    cut = lambda v: int((v[N[0] - 1] != v[N[1] - 1]).sum())
    assert cut(p) <= 1.1 * cut(q) + 8
    # transposed input (3 x nc) is accepted as well
    assert np.array_equal(dd.partition_rcb(np.ascontiguousarray(np.asarray(X).T) if np.asarray(X).shape[0] != 3 else np.asarray(X).T, nparts), p)
