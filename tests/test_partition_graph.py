"""jh_partition_graph: the coordinate-free graph partitioner standing in for the reference's MetisPartitioner
(partitioning.jl:29-51).  The reference pins Metis only by validity properties (test/partitioning.jl:23-28: every part used,
lengths match); same here plus balance, cut quality against the coordinate bisection, determinism, weights, degenerate inputs.
Host code only (no GPU)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ja():
    import jutul_amd
    return jutul_amd


def cut(N, p, w=None):
    c = p[N[0] - 1] != p[N[1] - 1]
    return float(c.sum() if w is None else w[c].sum())


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 13])
def test_valid_balanced_and_deterministic(ja, k):
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(9, 8, 7)
    N, nc = g["N"], g["nc"]
    p = dd.partition_graph(N, nc, k)
    assert p.shape == (nc,) and p.min() == 1 and p.max() == k
    sizes = np.bincount(p)[1:]
    assert len(sizes) == k and sizes.min() > 0                     # every part used (test/partitioning.jl:23-28)
    assert sizes.max() <= np.ceil(1.10 * nc / k) and sizes.min() >= np.floor(0.90 * nc / k)
    assert np.array_equal(p, dd.partition_graph(N, nc, k))         # deterministic
    assert np.array_equal(dd.compress_partition(p), p)
    if k > 1:  # quality: within 1.6x of the coordinate bisection on a lattice, where planes are near-optimal cuts
        assert cut(N, p) <= 1.6 * cut(N, dd.partition_rcb(g["cell_centroids"], k))
        # parts are (nearly) connected: process_partition splits disconnected pieces off -- few new parts appear
        assert dd.process_partition(N, p).max() <= k + max(2, k // 2)


def test_face_weights_steer_the_cut(ja):
    """A 2 x n strip with heavy rungs: the unweighted bisection cuts 2 rail edges (weight 2); if one rail pair in the middle is made
    very heavy the weighted cut moves away from it."""
    from jutul_amd import dd
    n = 40
    rails = [(i, i + 1) for i in range(1, n)] + [(n + i, n + i + 1) for i in range(1, n)]
    rungs = [(i, n + i) for i in range(1, n + 1)]
    N = np.array(rails + rungs, dtype=np.int64).T
    w = np.ones(N.shape[1])
    p = dd.partition_graph(N, 2 * n, 2)
    assert cut(N, p) == 2 and abs(int((p == 1).sum()) - n) <= 2
    mid = [j for j, (a, b) in enumerate(rails) if a in (n // 2, n + n // 2)]
    w[mid] = 1000.0
    pw = dd.partition_graph(N, 2 * n, 2, face_weights=w)
    assert cut(N, pw, w) <= 6.0 and all(pw[N[0, j] - 1] == pw[N[1, j] - 1] for j in mid)  # never through the heavy pair


def test_degenerate_inputs(ja):
    from jutul_amd import dd
    N = ja.cartesian_neighbors((4, 1, 1))
    assert list(dd.partition_graph(N, 4, 4)) != [] and sorted(dd.partition_graph(N, 4, 4)) == [1, 2, 3, 4]
    with pytest.raises(ja.JutulHIPError):
        dd.partition_graph(N, 4, 5)
    # disconnected graph (two components) and an isolated cell
    N2 = np.array([[1, 2], [2, 3], [5, 6], [6, 7]], dtype=np.int64).T
    p = dd.partition_graph(N2, 8, 2)
    assert sorted(np.bincount(p)[1:]) == [4, 4]
    # usable as the partition of the subdomain logic: owned cells partition the grid
    g = ja.tet_lattice_mesh(4, 4, 3)
    part = dd.partition_graph(g["N"], g["nc"], 3)
    owned = np.concatenate([dd.local_subdomain(g["N"], part, r)["cells"][: dd.local_subdomain(g["N"], part, r)["n_owned"]] for r in (1, 2, 3)])
    assert np.array_equal(np.sort(owned), np.arange(1, g["nc"] + 1))


def test_partition_with_groups_like_the_reference_tests(ja):
    """test/partitioning.jl:13-53: every part used for np = 1..10 on a random symmetric graph; the 50-cell chain in 5 parts;
    groups stay together in all three modes (contracted, heavy weights with and without buffer)."""
    from jutul_amd import dd
    rng = np.random.default_rng(4)
    n = 100
    ij = np.argwhere(np.triu(rng.random((n, n)) < 0.2, 1)) + 1          # sprand(100, 100, 0.2) + I, symmetrised
    N = ij.T
    for k in range(1, 11):
        p = dd.partition(N, k, n=n)
        assert p.min() == 1 and p.max() == k and all((p == i).any() for i in range(1, k + 1))
    N = np.stack([np.arange(1, 50), np.arange(2, 51)])
    p = dd.partition(N, 5)
    assert p.size == 50 and p.min() == 1 and p.max() == 5 and all((p == i).any() for i in range(1, 6))
    grps = [[3, 4, 5], [17, 19, 18]]
    for kw in (dict(group_by_weights=False), dict(group_by_weights=True, buffer_group=True), dict(group_by_weights=True, buffer_group=False)):
        p = dd.partition(N, 5, groups=grps, **kw)
        assert p.size == 50 and p.min() == 1 and p.max() == 5
        for g in grps:
            assert len(set(p[np.array(g) - 1])) == 1, (kw, g, p[np.array(g) - 1])
    with pytest.raises(ValueError):
        dd.partition(N, 49, groups=[list(range(1, 11))])               # 41 contracted cells < 49 blocks


def test_partition_does_not_depend_on_the_thread_count(monkeypatch):
    """jh_partition_graph builds its adjacency and sweeps its top bisection jobs on all host cores: same parts for any thread count
    (a 355k-cell scrambled lattice: above the size from which the sweeps run on thread teams; weighted, unweighted, multigraph)."""
    import jutul_amd as ja
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(40, 39, 38, scramble=True)
    N, nc = g["N"], g["nc"]
    dup = np.concatenate([N, N[:, :100]], axis=1)
    res = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("JH_SETUP_THREADS", threads)
        res.append((dd.partition_graph(N, nc, 8, face_weights=g["T"]), dd.partition_graph(N, nc, 37), dd.partition_graph(dup, nc, 5)))
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert np.array_equal(a, b)
    sizes = np.bincount(res[0][1])[1:]
    assert sizes.size == 37 and sizes.min() > 0.85 * nc / 37 and sizes.max() < 1.15 * nc / 37


def test_sealed_and_non_finite_faces_have_a_small_positive_weight(ja):
    """Zero, NaN and Inf couplings count as 1e-3 of the mean weight (a sealed face is cheap to cut, not free: the reference's integer
    Metis weights have a floor of one unit, generate_metis_graph partitioning.jl:64-78) -- the partition equals the one of the
    same graph with those weights written out, and it is valid and balanced.  The device blocks (jh_tpfa_create_weighted) follow the
    same rule: a grid whose sealed faces are given as 0 and one where they are given as 1e-3 x mean get the same blocks."""
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(8, 7, 6)
    N, nc = g["N"], g["nc"]
    rng = np.random.default_rng(3)
    w = rng.random(N.shape[1]) ** 2 + 0.01
    bad = rng.choice(N.shape[1], N.shape[1] // 10, replace=False)
    w_bad = w.copy()
    w_bad[bad[0::3]] = 0.0
    w_bad[bad[1::3]] = np.nan
    w_bad[bad[2::3]] = np.inf
    finite = np.where(np.isfinite(w_bad), np.abs(w_bad), 0.0)
    w_floor = finite.copy()
    w_floor[bad] = 1e-3 * finite.sum() / N.shape[1]
    for k in (2, 7):
        p = dd.partition_graph(N, nc, k, face_weights=w_bad)
        sizes = np.bincount(p)[1:]
        assert len(sizes) == k and sizes.min() >= np.floor(0.9 * nc / k) and sizes.max() <= np.ceil(1.1 * nc / k)
        assert np.array_equal(p, dd.partition_graph(N, nc, k, face_weights=w_floor))
    ctx = ja.HIPContext("host")
    a = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", block_rows=64, face_weights=w_bad).ordering()
    b = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", block_rows=64, face_weights=w_floor).ordering()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
