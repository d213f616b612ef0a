"""CPU-only: the PRODUCT library's set-up tables (SURVEY 8 rows a-1 .. a-4, a-13's partition, a-11's symbolic phase) through a
planning context (jh_context_create_host) -- the same C++ that runs in front of the GPU kernels, checked bit for bit against the
oracle without a device.  Every compute entry point must refuse such a context (there is no CPU fallback)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ja():
    import __graft_entry__ as g
    g.build()
    import jutul_amd
    return jutul_amd


@pytest.fixture(scope="module")
def hctx(ja):
    return ja.HIPContext("host")


def check_tables(ja, hctx, oracle, N, nc, nblk, reorder, block_rows=64, weights=None):
    disc = ja.TwoPointPotentialFlowHardCoded(hctx, N, nc, block_n=nblk, reorder=reorder, block_rows=block_rows, face_weights=weights)
    h = oracle.half_face_map(N, nc)
    c = disc.conn
    assert np.array_equal(c["face_pos"], h["face_pos"])
    assert np.array_equal(c["self"], h["self"]) and np.array_equal(c["other"], h["other"])
    assert np.array_equal(c["face"], h["faces"]) and np.array_equal(c["face_sign"], h["face_sign"])
    rowptr, colidx = disc.pattern()
    orp, oci = oracle.csr_pattern(nc, h)
    assert np.array_equal(rowptr, orp) and np.array_equal(colidx, oci)
    pa, pf = disc.jacobian_positions()
    opa, opf = oracle.align(nc, nblk, 2, orp, oci, h)
    assert np.array_equal(pa, opa) and np.array_equal(pf, opf)
    perm, bp = disc.ordering()
    assert np.array_equal(np.sort(perm), np.arange(1, nc + 1))
    return disc, h, (orp, oci), (perm, bp)


@pytest.mark.parametrize("reorder", ["none", "blocks"])
@pytest.mark.parametrize("nblk", [1, 2])
def test_tables_bit_exact_on_a_tet_lattice(ja, hctx, oracle, reorder, nblk):
    g = ja.tet_lattice_mesh(5, 4, 3)
    disc, _, _, (perm, bp) = check_tables(ja, hctx, oracle, g["N"], g["nc"], nblk, reorder)
    if reorder == "blocks":
        assert bp[0] == 0 and bp[-1] == g["nc"] and np.all(np.diff(bp) > 0) and np.diff(bp).max() <= 80


@pytest.mark.parametrize("layout", ["equation_major", "entity_major"])
def test_scalar_layout_tables_bit_exact(ja, hctx, oracle, layout):
    g = ja.tet_lattice_mesh(5, 4, 3)
    nc, N = g["nc"], 2
    disc, h, (orp, oci), _ = check_tables(ja, hctx, oracle, g["N"], nc, N, "blocks")
    lay = oracle.LAYOUT[layout]
    srp, sci = oracle.csr_pattern_scalar(nc, N, lay, orp, oci)
    rp, ci = disc.pattern_layout(layout)
    assert np.array_equal(rp, srp) and np.array_equal(ci, sci)
    opa, opf = oracle.align(nc, N, lay, orp, oci, h, srp, sci)
    pa, pf = disc.jacobian_positions_layout(layout)
    assert np.array_equal(pa, opa) and np.array_equal(pf, opf)


def test_pico_fixture_random_graphs_and_multigraphs(ja, hctx, oracle, golden):
    N = golden["pico"]["N"]
    check_tables(ja, hctx, oracle, N, 9, 1, "none")
    rng = np.random.default_rng(5)
    for nc, nf in [(40, 90), (200, 700), (17, 16)]:
        a = rng.integers(1, nc + 1, size=3 * nf)
        b = rng.integers(1, nc + 1, size=3 * nf)
        keep = a != b
        key = np.minimum(a, b) * (nc + 1) + np.maximum(a, b)
        _, first = np.unique(key[keep], return_index=True)
        N = np.stack([a[keep][np.sort(first)], b[keep][np.sort(first)]])[:, :nf]
        for reorder in ("none", "blocks"):
            check_tables(ja, hctx, oracle, N, nc, 1, reorder, block_rows=16)
    d = ja.TwoPointPotentialFlowHardCoded(hctx, np.array([[1, 1], [2, 2]]), 2)  # duplicate pair: a multigraph, accepted like the reference does
    assert d.nnzb == 4 and d.nhf == 4
    with pytest.raises(ja.JutulHIPError):
        ja.TwoPointPotentialFlowHardCoded(hctx, np.array([[1], [5]]), 3)
    with pytest.raises(ja.JutulHIPError):
        ja.TwoPointPotentialFlowHardCoded(hctx, np.array([[1], [1]]), 2)


@pytest.mark.parametrize("family", ["delaunay", "polyhedral", "cartesian"])
def test_unstructured_families_tables_and_block_partition(ja, hctx, oracle, family):
    g = {"delaunay": lambda: ja.delaunay_tet_mesh(1500, grading=2.0),
         "polyhedral": lambda: ja.polyhedral_dual_mesh(1500, grading=1.5),
         "cartesian": lambda: ja.cartesian_mesh(14, 13, 12)}[family]()
    nc = g["nc"]
    disc, h, (orp, oci), (perm, bp) = check_tables(ja, hctx, oracle, g["N"], nc, 1, "blocks", block_rows=128, weights=g["T"])
    sizes = np.diff(bp)
    assert sizes.min() >= 1 and sizes.max() <= 144          # the bisection's size cap: block_rows * 9 / 8
    # ILU(0) symbolic phase on the device blocks: the entries it keeps are exactly the in-block couplings of the oracle's pattern
    A = ja.StaticSparsityMatrixCSR(disc)
    prec = ja.ILUZeroPreconditioner(partition="blocks").symbolic(A)
    fi = prec.info()
    blk_of_host = np.empty(nc, dtype=np.int64)
    blk_of_host[perm - 1] = np.repeat(np.arange(sizes.size), sizes)
    rows = np.repeat(np.arange(nc), np.diff(orp))
    cols = oci - 1
    same = blk_of_host[rows] == blk_of_host[cols]
    assert fi["nblocks"] == sizes.size and fi["max_block_rows"] == sizes.max()
    assert fi["l_entries"] + fi["u_entries"] == int(np.count_nonzero(same & (rows != cols)))
    assert fi["l_entries"] == fi["u_entries"]               # structurally symmetric pattern
    # dependency levels of the triangular sweeps (ilu0.jl:156-187 in the device's elimination order): the longest chain of in-block
    # lower (forward) / upper (backward) couplings, restated here from the ordering alone
    dev_of_host = np.empty(nc, dtype=np.int64)
    dev_of_host[perm - 1] = np.arange(nc)
    nbrs = [[] for _ in range(nc)]
    for r, c, keep in zip(dev_of_host[rows], dev_of_host[cols], same & (rows != cols)):
        if keep:
            nbrs[r].append(c)
    flev, blev = np.zeros(nc, dtype=np.int64), np.zeros(nc, dtype=np.int64)
    for i in range(nc):
        lower = [flev[j] for j in nbrs[i] if j < i]
        flev[i] = 1 + max(lower) if lower else 0
    for i in range(nc - 1, -1, -1):
        upper = [blev[j] for j in nbrs[i] if j > i]
        blev[i] = 1 + max(upper) if upper else 0
    assert fi["max_levels"] == 1 + max(flev.max(), blev.max())


def test_weighted_blocks_cut_less_coupling_weight(ja, hctx):
    """jh_tpfa_create_weighted (partitioning.jl:64-78: Metis on the |A|-weighted graph): the blocks cut less coupling WEIGHT than
    the unweighted ones on a grid whose transmissibilities vary."""
    g = ja.tet_lattice_mesh(16, 15, 14)
    nc, N, T = g["nc"], g["N"], g["T"]

    def cut_weight(weights):
        d = ja.TwoPointPotentialFlowHardCoded(hctx, N, nc, reorder="blocks", block_rows=128, face_weights=weights)
        perm, bp = d.ordering()
        blk = np.empty(nc, dtype=np.int64)
        blk[perm - 1] = np.repeat(np.arange(bp.size - 1), np.diff(bp))
        cut = blk[N[0] - 1] != blk[N[1] - 1]
        return T[cut].sum() / T.sum()

    assert cut_weight(T) < 0.9 * cut_weight(None)


def test_set_up_does_not_depend_on_the_thread_count(ja):
    """The set-up runs on all host cores; nothing it produces may depend on the thread timing or count -- the getters' tables and
    (plan_checksum) every table the device would be handed: pattern, tiles, jagged layouts, ILU(0) levels, maps and programs."""
    code = r"""
import sys, zlib, numpy as np
sys.path.insert(0, %r)
import jutul_amd as ja
ctx = ja.HIPContext("host", plan_checksum=1)
g = ja.tet_lattice_mesh(30, 29, 28, scramble=True)   # 146k cells: above the size from which the sweeps run on thread teams
d = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], reorder="blocks", block_rows=256, face_weights=g["T"])
perm, bp = d.ordering()
rp, ci = d.pattern()
pa, pf = d.jacobian_positions()
A = ja.StaticSparsityMatrixCSR(d)
fi = ja.ILUZeroPreconditioner(partition="blocks").symbolic(A).info()
h = 0
for a in (perm, bp, rp, ci, pa, pf):
    h = zlib.crc32(np.ascontiguousarray(a).tobytes(), h)
print(h, fi["nblocks"], fi["max_levels"], fi["l_entries"], A.spmv_info()["slices"], ctx.plan_checksum())
a, b = ja.tet_lattice_mesh(30, 29, 28, scramble=True), ja.tet_lattice_mesh(19, 18, 17)   # two pieces, no face between them
N2 = np.concatenate([a["N"], b["N"] + a["nc"], a["N"][:, :500]], axis=1)                  # ... and 500 duplicated faces (multigraph)
d = ja.TwoPointPotentialFlowHardCoded(ctx, N2, a["nc"] + b["nc"], reorder="blocks", face_weights=np.concatenate([a["T"], b["T"], a["T"][:500]]))
A = ja.StaticSparsityMatrixCSR(d)
ja.ILUZeroPreconditioner(partition="blocks").symbolic(A)
print("two pieces", zlib.crc32(np.ascontiguousarray(d.ordering()[0]).tobytes()), ctx.plan_checksum())
g = ja.polyhedral_dual_mesh(6000, grading=1.5)      # long rows: virtual rows, rows-form factorisation programs
for bn in (1, 2):
    d = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], block_n=bn, reorder="blocks", face_weights=g["T"])
    A = ja.StaticSparsityMatrixCSR(d)
    ja.ILUZeroPreconditioner(partition="blocks").symbolic(A)
    print(A.spmv_info()["longest_row"], ctx.plan_checksum())
""" % ROOT
    outs = []
    for threads in ("1", "3", "8"):
        env = dict(os.environ, JH_SETUP_THREADS=threads)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] and outs[0] == outs[1] == outs[2], outs


def test_planning_context_refuses_every_compute_path(ja, hctx):
    g = ja.tet_lattice_mesh(3, 3, 2)
    disc = ja.TwoPointPotentialFlowHardCoded(hctx, g["N"], g["nc"], reorder="blocks", block_rows=32)
    A = ja.StaticSparsityMatrixCSR(disc)
    prec = ja.ILUZeroPreconditioner(partition="blocks").symbolic(A)
    for call in (lambda: ja.ConservationLaw(disc, "poisson"),
                 lambda: ja.DeviceVector(disc),
                 lambda: A.new_vector(),
                 lambda: ja.LinearizedSystem(disc),
                 lambda: prec.update_preconditioner(A),
                 lambda: prec.factor_values(),
                 lambda: setattr(A, "nzval", np.ones(disc.nnzb)),
                 lambda: A.set_nzval_layout(np.ones(disc.nnzb), "block_major"),
                 lambda: ja.mul_(None, A, None) if False else ja.StaticSparsityMatrixCSR(context=hctx, n=2, bs=1, rowptr=[1, 2, 3], colidx=[1, 2], nzval=[1.0, 1.0]).nzval,
                 lambda: A.nzval,
                 lambda: hctx.synchronize(),
                 lambda: hctx.timer_start(),
                 lambda: hctx.set_cu_mask(0, 8),
                 lambda: hctx.comm_init_ipc_only(1, 0),
                 lambda: ja.GenericKrylov("bicgstab", preconditioner=prec)._workspace(A)):
        with pytest.raises(ja.JutulHIPError, match="no device|not factored"):
            call()


def test_device_ordering_is_the_pinned_one(ja, hctx):
    """tests/golden/ordering_crc.json (make_ordering_crc.py): the set-up code may be re-threaded or restructured, the blocks it cuts
    -- hence every iteration count measured on the GPU -- stay what they were."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_ordering_crc", os.path.join(ROOT, "tests", "golden", "make_ordering_crc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pinned = json.load(open(os.path.join(ROOT, "tests", "golden", "ordering_crc.json")))
    for name, case in mod.cases(ja).items():
        assert mod.figures(ja, hctx, *case) == pinned[name], name


def test_rank_local_set_up_and_halo_plan_without_a_device(ja, hctx):
    """a-16 / a-17, host part, in the product library: every rank's discretisation (ghosts last, owned-first numbering), its halo
    plan and the ILU(0) symbolic phase with the fused halo pack, for a 3-rank decomposition -- the plans of neighbouring ranks
    mirror each other, ghosts grouped by owner are received without an unpack pass, and a wrong plan is refused."""
    from jutul_amd import dd
    g = ja.tet_lattice_mesh(12, 11, 10)
    N, nc = g["N"], g["nc"]
    part = dd.partition_graph(N, nc, 3, face_weights=g["T"])
    subs = [dd.local_subdomain(N, part, r + 1, ghost_order="owner") for r in range(3)]
    discs = []
    for r, sub in enumerate(subs):
        d = ja.TwoPointPotentialFlowHardCoded(hctx, sub["N"], sub["n_local"], reorder="blocks", block_rows=64, n_owned=sub["n_owned"],
                                              face_weights=g["T"][sub["faces"] - 1])
        d.set_halo(sub["n_owned"], sub["neighbors"], sub["send"], sub["recv"])
        hi = d.halo_info()
        assert hi["n_owned"] == sub["n_owned"] and hi["n_nbr"] == len(sub["neighbors"]) and hi["direct_recv"]
        assert hi["n_send"] == sum(len(x) for x in sub["send"]) and hi["n_recv"] == sub["n_local"] - sub["n_owned"]
        perm, bp = d.ordering()
        assert np.array_equal(np.sort(perm[:sub["n_owned"]]), np.arange(1, sub["n_owned"] + 1))      # ghosts stay the last device rows
        interior_rows, interior_blocks, _ = d.split()      # [interior blocks | boundary blocks | ghosts]
        assert 0 <= interior_rows <= sub["n_owned"] and (interior_rows == 0 or bp[interior_blocks] == interior_rows)
        A = ja.StaticSparsityMatrixCSR(d)
        fi = ja.ILUZeroPreconditioner(partition="blocks").symbolic(A).info()
        assert fi["nblocks"] == bp.size - 1 and fi["jagged"]
        discs.append(d)
    # what rank a sends to rank b is, cell for cell in global numbering, what b receives from a
    for a, sa in enumerate(subs):
        for i, b in enumerate(sa["neighbors"]):
            sb = subs[int(b)]
            j = list(sb["neighbors"]).index(a)
            assert np.array_equal(sa["cells"][np.asarray(sa["send"][i]) - 1], sb["cells"][np.asarray(sb["recv"][j]) - 1])
    sub = subs[0]
    d = ja.TwoPointPotentialFlowHardCoded(hctx, sub["N"], sub["n_local"], reorder="blocks", block_rows=64, n_owned=sub["n_owned"])
    with pytest.raises(ja.JutulHIPError, match="not an owned cell"):
        d.set_halo(sub["n_owned"], sub["neighbors"], [np.asarray(s) * 0 + sub["n_local"] for s in sub["send"]], sub["recv"])
    with pytest.raises(ja.JutulHIPError, match="not a ghost cell"):
        d.set_halo(sub["n_owned"], sub["neighbors"], sub["send"], [np.asarray(r_) * 0 + 1 for r_ in sub["recv"]])
    d2 = ja.TwoPointPotentialFlowHardCoded(hctx, sub["N"], sub["n_local"], reorder="blocks", block_rows=64)   # n_owned forgotten
    with pytest.raises(ja.JutulHIPError, match="disagree on n_owned"):
        d2.set_halo(sub["n_owned"], sub["neighbors"], sub["send"], sub["recv"])
    with pytest.raises(ja.JutulHIPError, match="no device"):
        ja.DeviceVector(discs[0])


def test_setup_heap_option_changes_no_table(ja):
    """setup_heap (process-wide malloc policy of the host during the set-up) is about page faults only: same tables, and the option
    can be switched on and off around a set-up."""
    g = ja.tet_lattice_mesh(20, 19, 18, scramble=True)
    sums = []
    for heap in (0, 1):
        ctx = ja.HIPContext("host", plan_checksum=1, setup_heap=heap)
        assert ctx.get_option("setup_heap") == heap
        d = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], g["nc"], reorder="blocks", face_weights=g["T"])
        A = ja.StaticSparsityMatrixCSR(d)
        A.spmv_info()
        ja.ILUZeroPreconditioner(partition="blocks").symbolic(A)
        ctx.set_option("setup_heap", 0)
        assert ctx.get_option("setup_heap") == 0
        sums.append(ctx.plan_checksum())
    assert sums[0] == sums[1] != 0


def test_long_thin_graphs_do_not_pay_for_the_thread_teams(ja, monkeypatch):
    """A 1-D column has as many breadth-first levels as cells: the level-synchronous sweeps of the set-up (bfs_parallel) must fall
    back to the plain loop for small levels -- same blocks for any thread count, and no barrier per cell."""
    import time
    n = 300_000
    chain = np.stack([np.arange(1, n), np.arange(2, n + 1)])
    w, h = 6, 40_000                                                    # a 6 x 40000 strip: 40k levels of 6 cells
    idx = np.arange(w * h).reshape(h, w) + 1
    strip = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]), np.stack([idx[:-1].ravel(), idx[1:].ravel()])], axis=1)
    for name, N, nc in (("chain", chain, n), ("strip", strip, w * h)):
        sums = []
        for threads in ("1", "8"):
            monkeypatch.setenv("JH_SETUP_THREADS", threads)
            ctx = ja.HIPContext("host", plan_checksum=1)
            t0 = time.time()
            d = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks")
            dt = time.time() - t0
            perm, bp = d.ordering()
            assert np.array_equal(np.sort(perm), np.arange(1, nc + 1)) and np.diff(bp).max() <= 576
            sums.append(ctx.plan_checksum())
            assert dt < 30.0, (name, threads, dt)
        assert sums[0] == sums[1], name


def test_exclusive_ranks_on_one_device_are_refused(ja):
    """jh_comm_set_exclusive(1) is a liveness promise (every wavefront of the chip-filling kernels spins until all peers have
    pushed): the library checks it against the devices it has seen and refuses when two ranks share one without CU masks.  Here:
    two in-process ranks on planning contexts (the same "device"); the host's own check of the observation works too."""
    group = ja.LocalCommGroup(2)
    a, b = ja.HIPContext("host"), ja.HIPContext("host")
    a.comm_init_local(group, 0)
    assert a.comm_devices_distinct() is None        # rank 1 has not joined yet: nothing observed
    a.comm_set_exclusive(False)                     # (turning it off is always fine)
    b.comm_init_local(group, 1)
    assert a.comm_devices_distinct() is False and b.comm_devices_distinct() is False
    with pytest.raises(ja.JutulHIPError, match="same device.*CU mask"):
        a.comm_set_exclusive(True)
    with pytest.raises(ja.JutulHIPError, match="has no device"):
        a.allreduce([1.0])                          # a planning context never computes, communicator or not
