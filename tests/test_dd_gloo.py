"""CPU multi-process test (gloo, world_size 2) of the domain-decomposition host logic: partitions, rank-local
subdomains with one ghost ring, halo plans and `consistent!` (HostExchange).  Compute is done with the CPU oracle
(tests may use it as the checker); the property pinned is the reference's distributed semantics:
N-rank distributed BiCGStab == serial solve."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import jutul_amd as ja
        from jutul_amd import dd
        from oracle import oracle as o
        o.set_num_threads(1)
        g = ja.tet_lattice_mesh(7, 6, 5)
        nc = g["nc"]
        T = g["T"] / g["T"].mean()
        rng = np.random.default_rng(0)
        U0 = 1.0 + 0.1 * rng.random(nc)
        dt = 0.5
        part = dd.partition_rcb(g["cell_centroids"], world)
        sub = dd.local_subdomain(g["N"], part, rank + 1)
        cells = sub["cells"] - 1
        no, nl = sub["n_owned"], sub["n_local"]
        # every owned cell of mine that is a ghost elsewhere is in a send list, and vice versa (plan symmetry)
        counts = torch.zeros(world, dtype=torch.int64)
        for s, snd in zip(sub["neighbors"], sub["send"]):
            counts[int(s)] = len(snd)
        allc = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, counts)
        for s, rcv in zip(sub["neighbors"], sub["recv"]):
            assert int(allc[int(s)][rank]) == len(rcv)
        # consistent!: ghosts receive the owner's value
        ex = dd.HostExchange(sub)
        v = (cells + 1).astype(np.float64)
        v[no:] = -1.0
        ex(v)
        assert np.array_equal(v, (cells + 1).astype(np.float64))
        # local system = assemble on owned+ghost, ghost rows -> -I (linalg.jl:18-35)
        osys = o.TPFASystem(sub["N"], nl)
        law = o.Law("poisson", dt)
        src_c = [i + 1 for i, c in enumerate(cells[:no]) if c in (0, nc - 1)]
        src_v = [1.0 if cells[i - 1] == 0 else -1.0 for i in src_c]
        nz, r = osys.assemble(law, U0[cells], U0[cells], g["volumes"][cells], T[sub["faces"] - 1], src_cells=src_c or None,
                              src_values=src_v or None)
        nz, r = o.unit_diagonalize(nl, no, 1, osys.rowptr, osys.colidx, nz, r)
        F = o.ILU0(nl, 1, osys.rowptr, osys.colidx, nz)

        def gdot(a, b):
            t = torch.tensor([float(a[:no] @ b[:no])], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t[0])

        def prec(x):  # parray_preconditioner_apply!: ghost input zeroed (linalg.jl:78-88)
            x = x.copy()
            x[no:] = 0.0
            return F.apply(x)

        def mul(x):  # distributed_mul!: consistent!(x) then local mul! (linalg.jl:37-55)
            ex(x)
            return o.spmv(nl, 1, osys.rowptr, osys.colidx, nz, x)

        # left-preconditioned BiCGStab on distributed vectors (ext/.../krylov.jl:51-105)
        b = r.copy()
        ex(b)
        x = np.zeros(nl)
        rr = prec(b)
        p, c = rr.copy(), rr.copy()
        rho = gdot(c, rr)
        r0 = np.sqrt(gdot(rr, rr))
        its = 0
        while np.sqrt(gdot(rr, rr)) > 1e-12 * r0 and its < 200:
            its += 1
            vv = prec(mul(p))
            alpha = rho / gdot(c, vv)
            s = rr - alpha * vv
            t = prec(mul(s))
            omega = gdot(t, s) / gdot(t, t)
            x = x + alpha * p + omega * s
            rr = s - omega * t
            rho_n = gdot(c, rr)
            p = rr + (rho_n / rho) * (alpha / omega) * (p - omega * vv)
            rho = rho_n
        ex(x)
        q.put((rank, cells[:no], x[:no], cells[no:], x[no:], its))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(("error", rank, traceback.format_exc()))


def test_two_rank_gloo_distributed_solve_matches_serial():
    import torch.multiprocessing as mp
    from oracle import oracle as o
    import jutul_amd as ja
    world = 2
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for r in res:
        assert r[0] != "error", r
    # serial reference
    g = ja.tet_lattice_mesh(7, 6, 5)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    U0 = 1.0 + 0.1 * np.random.default_rng(0).random(nc)
    osys = o.TPFASystem(g["N"], nc)
    nz, r = osys.assemble(o.Law("poisson", 0.5), U0, U0, g["volumes"], T, src_cells=[1, nc], src_values=[1.0, -1.0])
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    A = sp.csr_matrix((nz, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc))
    x_ref = spl.spsolve(A.tocsc(), r)
    x = np.zeros(nc)
    for rank, own, xo, gh, xg, its in res:
        x[own] = xo
        assert np.allclose(xg, x_ref[gh], rtol=1e-8, atol=1e-10)  # ghosts consistent after the final exchange
        assert its < 100
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()


def test_partition_and_subdomain_properties(oracle):
    import jutul_amd as ja
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(6, 5, 4)
    nc, N = g["nc"], g["N"]
    assert np.array_equal(dd.partition_linear(3, 7), oracle.partition_linear(3, 7))  # partitioning.jl:12-18
    assert np.array_equal(dd.compress_partition([1, 3, 6, 5]), [1, 2, 4, 3])         # test/partitioning.jl:8
    for npart in (1, 2, 3, 5, 8):
        p = dd.partition_rcb(g["cell_centroids"], npart)
        assert p.min() == 1 and p.max() == npart and np.all(np.bincount(p)[1:] > 0)  # test/partitioning.jl:13-19
        assert np.bincount(p)[1:].max() - np.bincount(p)[1:].min() <= npart
        rem, counts = dd.remap_global_indices(p, npart)
        orem, ocounts = oracle.remap_global_indices(p, npart)
        assert np.array_equal(rem, orem) and np.array_equal(counts, ocounts)
        owned_total = 0
        for r in range(1, npart + 1):
            sub = dd.local_subdomain(N, p, r)
            owned_total += sub["n_owned"]
            ghosts = sub["cells"][sub["n_owned"]:]
            assert np.array_equal(ghosts, np.sort(oracle.partition_boundary(N, p, r)))  # utils.jl:32-56
            assert np.array_equal(ghosts, dd.partition_boundary(N, p, r))
            # faces kept iff both cells local (subdomains.jl:108-119), local numbering consistent
            gl = sub["cells"]
            assert np.array_equal(gl[sub["N"][0] - 1], N[0][sub["faces"] - 1])
            assert np.array_equal(gl[sub["N"][1] - 1], N[1][sub["faces"] - 1])
            loc = np.zeros(nc + 1, dtype=bool)
            loc[gl] = True
            assert np.array_equal(np.flatnonzero(loc[N[0]] & loc[N[1]]) + 1, sub["faces"])
            for s, snd, rcv in zip(sub["neighbors"], sub["send"], sub["recv"]):
                assert np.all(p[gl[rcv - 1] - 1] == s + 1) and np.all(snd <= sub["n_owned"]) and np.all(rcv > sub["n_owned"])
            # ghosts grouped by owner: same ghost set, every neighbour's receive list is a run of consecutive local cells
            sub_o = dd.local_subdomain(N, p, r, ghost_order="owner")
            assert np.array_equal(np.sort(sub_o["cells"][sub_o["n_owned"]:]), ghosts) and sub_o["n_owned"] == sub["n_owned"]
            assert np.array_equal(sub_o["neighbors"], sub["neighbors"])
            nxt = sub_o["n_owned"] + 1
            for snd_o, snd, rcv in zip(sub_o["send"], sub["send"], sub_o["recv"]):
                assert np.array_equal(rcv, np.arange(nxt, nxt + rcv.size)) and np.array_equal(snd_o, snd)
                assert np.all(np.diff(sub_o["cells"][rcv - 1]) > 0)  # ascending global id inside an owner group
                nxt += rcv.size
        assert owned_total == nc


def test_partition_helper_kats():
    """Known answers of the reference's partition helpers (test/partitioning.jl:55-70, test/utils.jl:10-30)."""
    import jutul_amd as ja
    from jutul_amd import dd
    N5 = ja.cartesian_neighbors((5,))
    assert list(dd.process_partition(N5, [1, 1, 2, 1, 1])) == [1, 1, 2, 3, 3]          # test/partitioning.jl:60-61
    p = np.array([1, 1, 2, 3, 3])
    out = dd.process_partition(N5, p)
    assert list(out) == [1, 1, 2, 3, 3] and out is not p                                # :62-64 (copy)
    assert list(dd.process_partition(N5, p, weights=[1.0, 1.0, 1.0, 0.0])) == [1, 1, 2, 3, 4]  # :66-67
    for n in range(1, 60):                                                              # test/utils.jl:10-30
        for m in range(1, n + 1):
            assert dd.load_balanced_endpoint(0, n, m) == 0 and dd.load_balanced_endpoint(m, n, m) == n
            count = 0
            for i in range(1, m + 1):
                a, b = dd.load_balanced_interval(i, n, m)
                delta = b - a + 1
                assert delta in (n // m, -(-n // m)) and delta > 0
                count += delta
            assert count == n
    # cartesian_partition (partitioning.jl:184-236): 4x4 cell centroids into 2x2 coarse blocks
    x, y = np.meshgrid(np.arange(4) + 0.5, np.arange(4) + 0.5, indexing="xy")
    pts = np.stack([x.reshape(-1), y.reshape(-1)])
    pc = dd.cartesian_partition(pts, (2, 2))
    assert list(pc) == [1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 4, 4]
    assert list(dd.cartesian_partition(pts, 4)) == list(range(1, 17))
    flat = np.stack([x.reshape(-1), np.zeros(16)])                                       # degenerate dimension skipped
    assert dd.cartesian_partition(flat, (2, 3)).max() == 2


def test_submap_cells_and_discretization_mirror(oracle):
    """submap_cells (dd/subdomains.jl:77-182) buffer 0/1 vs the literal oracle restatement, the reference's
    extract_submesh KAT (test/mesh.jl:181-185: cells 1:3 of a 2x2x2 mesh -> 3 cells / 2 faces), and the host-side
    transmissibility helpers (finite-volume.jl:130-233,304-313) vs the C oracle."""
    import jutul_amd as ja
    from jutul_amd import dd
    N = ja.cartesian_neighbors((2, 2, 2))
    s0 = dd.submap_cells(N, [1, 2, 3], 8, buffer=0)
    assert list(s0["cells"]) == [1, 2, 3] and len(s0["faces"]) == 2 and not s0["is_boundary"].any()
    g = ja.tet_lattice_mesh(4, 3, 3)
    nc = g["nc"]
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(np.arange(1, nc + 1), 40, replace=False))
    for buf in (0, 1):
        a = dd.submap_cells(g["N"], idx, nc, buffer=buf)
        b = oracle.submap_cells(g["N"], idx, nc, buffer=buf)
        assert np.array_equal(a["cells"], b["cells"]) and np.array_equal(a["faces"], b["faces"])
        assert np.array_equal(a["is_boundary"], b["is_boundary"])
    a = dd.submap_cells(g["N"], idx, nc, buffer=1, excluded=[int(idx[0]) % nc + 1])
    b = oracle.submap_cells(g["N"], idx, nc, buffer=1, excluded=[int(idx[0]) % nc + 1])
    assert np.array_equal(a["cells"], b["cells"]) and np.array_equal(a["faces"], b["faces"])
    # transmissibilities
    perm = np.stack([g["perm_k"], 2 * g["perm_k"], 3 * g["perm_k"]])
    Thf, hf = ja.compute_half_face_trans(g["cell_centroids"], g["face_centroids"], g["normals"], g["areas"], perm, g["N"])
    oh = oracle.half_face_map(g["N"], nc)
    for k in ("faces", "face_pos", "face_sign"):
        assert np.array_equal(hf[k], oh[k])
    assert np.array_equal(hf["cells"], oh["other"])
    geo = dict(nc=nc, dim=3, cell_centroids=g["cell_centroids"], face_centroids=g["face_centroids"], normals=g["normals"],
               areas=g["areas"])
    Thf_o = oracle.half_face_trans(geo, perm, oh)
    assert np.allclose(Thf, Thf_o, rtol=1e-13)
    assert np.allclose(ja.compute_face_trans(Thf, hf["faces"], g["nf"]), oracle.face_trans(Thf_o, oh["faces"], g["nf"]), rtol=1e-13)
    z = g["cell_centroids"][2]
    assert np.array_equal(ja.compute_face_gdz(g["N"], z, 9.81), oracle.face_gdz(g["N"], z, 9.81))


@pytest.mark.parametrize("side", ["left", "right"])
def test_oracle_rank_emulation_solves_the_global_system(oracle, side):
    """tests/_dd_emulation.py (the oracle pin of the GPU's multi-rank Krylov loop, used by test_gpu_distributed.py and
    xrank_worker.py) checked on its own: three ranks built from oracle assemblies, a scrambled elimination order with two
    block-Jacobi blocks per rank, both preconditioning sides -- the owned parts of the emulated solve are the global solution."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import jutul_amd as ja
    from jutul_amd import dd
    from tests import _ilu_checks as ck
    from tests._dd_emulation import emulated_bicgstab, oracle_rank
    o = oracle
    g = ja.tet_lattice_mesh(7, 6, 5)
    nc = g["nc"]
    T = g["T"] / g["T"].mean()
    rng = np.random.default_rng(0)
    U0 = 1.0 + 0.1 * rng.random(nc)
    world = 3
    part = dd.partition_rcb(g["cell_centroids"], world)
    ranks = []
    subs = []
    for r in range(world):
        sub = dd.local_subdomain(g["N"], part, r + 1)
        cells = sub["cells"] - 1
        no, nl = sub["n_owned"], sub["n_local"]
        osys = o.TPFASystem(sub["N"], nl)
        src_c = [i + 1 for i, c in enumerate(cells[:no]) if c in (0, nc - 1)]
        src_v = [1.0 if cells[i - 1] == 0 else -1.0 for i in src_c]
        nz, b = osys.assemble(o.Law("poisson", 0.5), U0[cells], U0[cells], g["volumes"][cells], T[sub["faces"] - 1],
                              src_cells=src_c or None, src_values=src_v or None)
        nz, b = o.unit_diagonalize(nl, no, 1, osys.rowptr, osys.colidx, nz, b)
        perm = np.concatenate([rng.permutation(no), np.arange(no, nl)]) + 1     # ghosts stay last, as on the device
        bp = np.array([0, no // 2, no, nl])
        ranks.append(oracle_rank(o, ck, sub, nz, b, perm, bp))
        subs.append(sub)
    hist, its, xs = emulated_bicgstab(o, ranks, side, 1e-11)
    assert 3 < its < 100 and hist[-1] <= 1e-11 * hist[0] and np.all(np.diff(np.log(hist))[:3] < 0)
    osys = o.TPFASystem(g["N"], nc)
    nz, rhs = osys.assemble(o.Law("poisson", 0.5), U0, U0, g["volumes"], T, src_cells=[1, nc], src_values=[1.0, -1.0])
    x_ref = spl.spsolve(sp.csr_matrix((nz, osys.colidx - 1, osys.rowptr - 1), shape=(nc, nc)).tocsc(), rhs)
    x = np.zeros(nc)
    for sub, xl in zip(subs, xs):
        x[sub["cells"][: sub["n_owned"]] - 1] = xl[: sub["n_owned"]]
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
