"""Domain decomposition host logic (a-16/a-17): partitions, rank-local subdomains with one ghost ring, halo plans.

Mirrors src/partitioning.jl and ext/JutulPartitionedArraysExt/utils.jl.  Integer work only; the data path
(pack -> RCCL send/recv -> unpack, all-reduces) runs inside libjutul_hip.so.  `HostExchange` performs the same
`consistent!` on host arrays through torch.distributed (gloo or nccl) -- used to distribute initial data and by
the CPU multi-process tests.
"""
import numpy as np


# ---- partitioners ------------------------------------------------------------------------------------------------
def partition_linear(m, n):
    """partition_linear(m, n) (partitioning.jl:12-18): p[i] = ceil(i / (n/m)), 1-based parts."""
    i = np.arange(1, n + 1, dtype=np.float64)
    return np.ceil(i / (n / m)).astype(np.int64)


def compress_partition(p):
    """compress_partition (partitioning.jl:92-99): renumber to 1..k preserving order of the ids."""
    p = np.asarray(p, dtype=np.int64)
    up = np.unique(p)
    return np.searchsorted(up, p).astype(np.int64) + 1


def load_balanced_endpoint(block_index, nvals, nblocks):
    """load_balanced_endpoint (partitioning.jl:315-323)."""
    width, remainder = divmod(nvals, nblocks)
    return min(min(block_index, remainder) + width * block_index, nvals)


def load_balanced_interval(b, n, m):
    """load_balanced_interval (partitioning.jl:330-334): 1-based inclusive (start, stop) of block b in 1..m."""
    return load_balanced_endpoint(b - 1, n, m) + 1, load_balanced_endpoint(b, n, m)


def cartesian_partition(pts, dims):
    """cartesian_partition(pts, dim) (partitioning.jl:184-236): bin points into a coarse Cartesian lattice over their
    bounding box; pts is [ndim, npts]; dims one int or one per dimension.  Degenerate dimensions are skipped exactly
    like the reference (their row of the index matrix stays 0), the result is compressed to 1..k."""
    pts = np.asarray(pts, dtype=np.float64)
    ndim, npts = pts.shape
    dims = [int(dims)] * ndim if np.isscalar(dims) else [int(d) for d in dims]
    if len(dims) != ndim:
        raise ValueError("Second argument must be one value or one per dimension (=number of rows in pts)")
    prow = np.zeros((ndim, npts), dtype=np.int64)
    for d in range(ndim):
        x = pts[d]
        x0, x1 = x.min(), x.max()
        if np.isclose(x0, x1):
            continue
        dx = (x1 - x0) / dims[d]
        prow[d] = np.clip(np.ceil((x - x0) / dx), 1, dims[d]).astype(np.int64)
    p = np.zeros(npts, dtype=np.int64)
    for d in range(ndim):
        p += prow[d] * int(np.prod(dims[:d]))
    return compress_partition(p)


def partition_graph(N, nc, nparts, face_weights=None, imbalance=0.03):
    """Graph partition without coordinates, in the role of the reference's MetisPartitioner (partitioning.jl:29-51): recursive
    bisection + Fiduccia-Mattheyses refinement inside libjutul_hip.so (jh_partition_graph; host code, no GPU needed).
    N: 2 x nf neighbourship (1-based); returns the 1-based part of every cell."""
    import ctypes as C
    from . import _lib
    N = np.asarray(N, dtype=np.int64)
    Nf = np.ascontiguousarray(np.asfortranarray(N).T.reshape(-1))
    out = np.zeros(int(nc), dtype=np.int64)
    w = None if face_weights is None else np.ascontiguousarray(face_weights, dtype=np.float64)
    _lib.check(_lib.load().jh_partition_graph(int(nc), int(N.shape[1]), _lib.pi(Nf), _lib.pf(w), int(nparts), float(imbalance), _lib.pi(out)))
    return out


def partition(N, num_coarse, weights=None, groups=None, n=None, group_by_weights=False, buffer_group=False, imbalance=0.03):
    """partition(N, num_coarse, weights; groups, n, group_by_weights, buffer_group) (partitioning.jl:239-308): partition a
    neighbourship with optional groups of cells that must not be divided (e.g. the cells a well perforates).
    group_by_weights=False: every group is contracted into one vertex before partitioning and expanded afterwards (always
    kept together); True: faces inside a group (buffer_group: touching a group) get 100x the largest weight instead, which
    makes cutting them very expensive.  The partitioner is jh_partition_graph in the role of the reference's Metis."""
    N = np.asarray(N, dtype=np.int64)
    assert N.shape[0] == 2
    nf = N.shape[1]
    w = np.ones(nf) if weights is None else np.asarray(weights, dtype=np.float64).copy()
    assert w.size == nf
    n = int(N.max()) if n is None else int(n)
    part = None
    if groups is not None and not group_by_weights:
        part = np.arange(1, n + 1)
        for i, grp in enumerate(groups):
            part[np.asarray(grp, dtype=np.int64) - 1] = n + 1 + i
        part = compress_partition(part)
        N = part[N - 1]
        n_inner = int(part.max())
    else:
        n_inner = n
    if groups is not None and group_by_weights:
        heavy = 100.0 * w.max()
        for grp in groups:
            inl, inr = np.isin(N[0], grp), np.isin(N[1], grp)
            w[(inl | inr) if buffer_group else (inl & inr)] = heavy
    if num_coarse > n_inner:
        raise ValueError("more coarse blocks than (contracted) cells")
    p = partition_graph(N, n_inner, num_coarse, face_weights=w, imbalance=imbalance)
    return p[part - 1] if part is not None else p


def process_partition(N, partition, weights=None):
    """process_partition (partitioning.jl:128-160): split every coarse block into its connected components; the first
    component (the one containing the block's lowest cell) keeps the id, the others get max+1, max+2, ...
    Faces whose weight is ~0 do not connect."""
    N = np.asarray(N, dtype=np.int64)
    part = np.asarray(partition, dtype=np.int64)
    new_p = part.copy()
    max_p = int(part.max())
    l, r = N[0] - 1, N[1] - 1
    ok = np.ones(l.size, dtype=bool) if weights is None else ~np.isclose(np.asarray(weights, dtype=np.float64), 0.0)
    same = ok & (part[l] == part[r])
    # union-find over the kept faces
    parent = np.arange(part.size)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for a, b in zip(l[same], r[same]):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)  # root = lowest cell: components ordered by their first vertex
    roots = np.array([find(i) for i in range(part.size)])
    for coarse in range(1, int(part.max()) + 1):
        cells = np.flatnonzero(part == coarse)
        if cells.size == 0:
            continue
        comp_roots = np.unique(roots[cells])  # ascending = order of each component's first cell
        for cr in comp_roots[1:]:
            max_p += 1
            new_p[cells[roots[cells] == cr]] = max_p
    return new_p


def partition_rcb(centroids, nparts):
    """Recursive coordinate bisection into `nparts` (any integer) compact parts; 1-based part ids.
    Build-side stand-in for MetisPartitioner (partitioning.jl:29-51; Metis.jl is an un-vendored C library):
    the partition vector is an INPUT to the hot path, any valid vector works.  Runs inside libjutul_hip.so on all host
    cores (jh_partition_rcb; no GPU needed)."""
    from . import _lib
    X = np.asarray(centroids, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    if X.shape[0] in (1, 2, 3) and X.shape[1] > 3:
        X = X.T
    X = np.ascontiguousarray(X)
    out = np.zeros(X.shape[0], dtype=np.int64)
    _lib.check(_lib.load().jh_partition_rcb(X.shape[0], X.shape[1], _lib.pf(X), int(nparts), _lib.pi(out)))
    return out


# ---- distributed numbering (ext/JutulPartitionedArraysExt/utils.jl) --------------------------------------------------------
def partition_boundary(N, p, rank):
    """partition_boundary (utils.jl:32-56) for one part (1-based `rank`): cells of other parts sharing a face with
    an owned cell.  Returned ascending (the reference delegates the final ghost order to PartitionedArrays;
    results are compared in global numbering)."""
    l, r = N[0] - 1, N[1] - 1
    il, ir = p[l] == rank, p[r] == rank
    return np.unique(np.concatenate([r[il & ~ir], l[ir & ~il]])) + 1


def remap_global_indices(p, nparts):
    """remap_global_indices, order = :default (utils.jl:9-30): owned-first contiguous numbering per rank."""
    p = np.asarray(p, dtype=np.int64)
    counts = np.bincount(p, minlength=nparts + 1)[1:]
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    order = np.argsort(p, kind="stable")
    rem = np.empty(p.size, dtype=np.int64)
    rem[order] = np.arange(1, p.size + 1)
    assert np.all(rem[order[offs]] == offs + 1)
    return rem, counts


def submap_cells(N, indices, nc=None, buffer=0, excluded=()):
    """submap_cells(gmap, N, indices; buffer) (dd/subdomains.jl:77-182) without a parent map: the local cell list,
    the kept faces and the boundary flags of a subdomain.  buffer = 0: cells = `indices` in the given order, faces
    kept iff both cells are listed, nothing flagged (the mode the distributed path uses).  buffer = 1: every face of a
    listed cell is kept, the neighbours are appended as boundary cells (cells come back ascending: `findall`)."""
    assert buffer in (0, 1)
    N = np.asarray(N, dtype=np.int64)
    idx = np.asarray(indices, dtype=np.int64)
    nc = int(N.max()) if nc is None else nc
    inside = np.zeros(nc + 1, dtype=bool)
    inside[idx] = True
    l, r = N[0], N[1]
    both = inside[l] & inside[r]
    if buffer == 0:
        return dict(cells=idx.copy(), faces=np.flatnonzero(both) + 1, is_boundary=np.zeros(idx.size, dtype=bool))
    excl = np.zeros(nc + 1, dtype=bool)
    excl[np.asarray(list(excluded), dtype=np.int64)] = True
    touch_l = inside[l] & ~excl[r]   # face seen from listed cell l: other = r
    touch_r = inside[r] & ~excl[l]
    face_active = both | touch_l | touch_r
    cell_active = inside.copy()
    cell_active[r[touch_l]] = True
    cell_active[l[touch_r]] = True
    cells = np.flatnonzero(cell_active)
    return dict(cells=cells, faces=np.flatnonzero(face_active) + 1, is_boundary=~inside[cells])


def local_subdomain(N, p, rank, ghost_order="global"):
    """Rank-local subdomain as PArraySimulator builds it (interface.jl:38-63 with submap_cells buffer = 0,
    dd/subdomains.jl:77-182): cells = [owned (findall order) ..., ghosts (ascending global id) ...]; faces kept iff
    both cells are local.  `rank` is 1-based.  All returned index arrays are 1-based.
    ghost_order="owner" sorts the ghosts by (owning rank, global id) instead: every neighbour's ghosts are then consecutive
    local cells and the device library receives them straight into the vectors (no unpack kernel).
    Built inside libjutul_hip.so on all host cores (jh_subdomain_*; no GPU needed)."""
    import ctypes as C
    from . import _lib
    if ghost_order not in ("global", "owner"):
        raise ValueError("ghost_order must be 'global' or 'owner'")
    L = _lib.load()
    N = np.asarray(N, dtype=np.int64)
    p = _lib.i64(p)
    Nf = np.ascontiguousarray(np.asfortranarray(N).T.reshape(-1))
    h = C.c_void_p()
    _lib.check(L.jh_subdomain_create(p.size, N.shape[1], _lib.pi(Nf), _lib.pi(p), int(rank), 1 if ghost_order == "owner" else 0, C.byref(h)))
    try:
        sz = np.zeros(5, dtype=np.int64)
        _lib.check(L.jh_subdomain_sizes(h, _lib.pi(sz)))
        n_owned, n_local, nfl, nn, nsend = (int(v) for v in sz)
        cells = np.empty(n_local, dtype=np.int64)
        faces = np.empty(nfl, dtype=np.int64)
        Nl = np.empty(2 * nfl, dtype=np.int64)
        nbr = np.empty(nn, dtype=np.int32)
        sp, rp = np.empty(nn + 1, dtype=np.int64), np.empty(nn + 1, dtype=np.int64)
        si, ri = np.empty(nsend, dtype=np.int64), np.empty(n_local - n_owned, dtype=np.int64)
        _lib.check(L.jh_subdomain_get(h, _lib.pi(cells), _lib.pi(faces), _lib.pi(Nl), _lib.pi32(nbr), _lib.pi(sp), _lib.pi(si),
                                      _lib.pi(rp), _lib.pi(ri)))
    finally:
        L.jh_subdomain_destroy(h)
    return dict(cells=cells, n_owned=n_owned, n_local=n_local, faces=faces, N=np.ascontiguousarray(Nl.reshape(nfl, 2).T),
                neighbors=nbr, send=[si[sp[k]:sp[k + 1]] for k in range(nn)], recv=[ri[rp[k]:rp[k + 1]] for k in range(nn)])


class HostExchange:
    """consistent!(v) on host arrays over torch.distributed (any backend): owner -> ghost copies."""

    def __init__(self, sub, block=1):
        self.sub, self.block = sub, block

    def __call__(self, v):
        import torch
        import torch.distributed as dist
        sub, b = self.sub, self.block
        v2 = v.reshape(-1, b)
        reqs, bufs = [], []
        for s, snd, rcv in zip(sub["neighbors"], sub["send"], sub["recv"]):
            out = torch.from_numpy(np.ascontiguousarray(v2[snd - 1]))
            inp = torch.empty((len(rcv), b), dtype=torch.float64)
            reqs.append(dist.isend(out, int(s)))
            reqs.append(dist.irecv(inp, int(s)))
            bufs.append((rcv, inp, out))
        for q in reqs:
            q.wait()
        for rcv, inp, _ in bufs:
            v2[rcv - 1] = inp.numpy()
        return v


def packed_exchange(sub):
    """Halo callback for HIPContext.comm_set_halo_callback over torch.distributed (any backend): exchanges the PACKED send /
    receive buffers of the device library, whose segments follow the neighbour order of the halo plan."""
    nsend = [len(c) for c in sub["send"]]
    nrecv = [len(c) for c in sub["recv"]]
    nbrs = [int(r) for r in sub["neighbors"]]

    def exchange(send, recv, bs):
        import torch
        import torch.distributed as dist
        reqs, bufs, so, ro = [], [], 0, 0
        for nb, ns, nr in zip(nbrs, nsend, nrecv):
            out = torch.from_numpy(send[so * bs:(so + ns) * bs].copy())
            inp = torch.empty(nr * bs, dtype=torch.float64)
            reqs += [dist.isend(out, nb), dist.irecv(inp, nb)]
            bufs.append((ro, nr, inp, out))
            so += ns
            ro += nr
        for q in reqs:
            q.wait()
        for ro, nr, inp, _ in bufs:
            recv[ro * bs:(ro + nr) * bs] = inp.numpy()
    return exchange


def setup_push_halo(disc, sub, rank, world, enable=True):
    """Collective set-up of the push halo (jh_halo_ipc_*) over torch.distributed (control plane, any backend): landing-buffer
    handles and halo plans are all-gathered, every rank maps its neighbours' buffers, one time-limited test exchange of the
    global cell ids is verified, and the push halo is enabled only if every rank passed.  Needs enabled mailboxes
    (HIPContext.comm_ipc_*).  Returns True if the Krylov-loop exchanges now use it."""
    import torch
    import torch.distributed as dist
    from . import DeviceVector
    nbrs = [int(q) for q in sub["neighbors"]]
    import sys
    import traceback
    try:
        handle = disc.halo_ipc_export()
    except Exception:  # noqa: BLE001
        traceback.print_exc(file=sys.stderr)  # not fatal: the exchange stays on the collective library
        handle = None
    info = [None] * world
    dist.all_gather_object(info, (handle, nbrs, [len(c) for c in sub["recv"]]))  # every rank takes part
    ok = False
    if all(i[0] is not None for i in info):
        try:
            hs, off, stride = [], [], []
            for q in nbrs:
                hq, nbr_q, nrecv_q = info[q]
                j = nbr_q.index(rank)
                hs.append(hq)
                off.append(int(sum(nrecv_q[:j])))
                stride.append(int(sum(nrecv_q)))
            ok = disc.halo_ipc_attach(hs, off, stride)
            if ok:  # ghosts must receive their owners' global cell ids
                N = disc.block_n
                gid = np.asarray(sub["cells"], dtype=np.float64)
                val = gid[:, None] + 0.25 * np.arange(N)[None, :]
                val[sub["n_owned"]:] = -1.0
                v = DeviceVector(disc)
                v.upload(val.reshape(-1))
                expect = np.concatenate([gid[np.asarray(c) - 1] for c in sub["recv"]] + [np.zeros(0)])
                ok = disc.halo_ipc_selftest(v, (expect[:, None] + 0.25 * np.arange(N)[None, :]).reshape(-1))
                if not ok:
                    print(f"[jutul_amd] rank {rank}: push-halo self-test failed, using the collective library", file=sys.stderr)
        except Exception:  # noqa: BLE001
            traceback.print_exc(file=sys.stderr)
            ok = False
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    use = bool(int(flag[0])) and enable
    disc.halo_ipc_enable(use)
    return use


def setup_rank_problem(ctx, N, part, rank, T, vol, X0, kind="poisson", block_n=1, sources=None, reorder="blocks",
                       block_rows=512, law_params=None, gdz=None, ghost_order="global"):
    """Builds this rank's discretisation + law the way PArraySimulator does per rank (interface.jl:38-63):
    submodel on [owned..., ghosts...], substate, halo plan.  `rank` is 0-based; `sources` = (cells 1-based global,
    values [n, block_n]).  Returns (disc, law, sub)."""
    from . import ConservationLaw, TwoPointPotentialFlowHardCoded
    sub = local_subdomain(N, part, rank + 1, ghost_order=ghost_order)
    cells = sub["cells"] - 1
    disc = TwoPointPotentialFlowHardCoded(ctx, sub["N"], sub["n_local"], block_n=block_n, reorder=reorder,
                                          block_rows=block_rows, n_owned=sub["n_owned"], face_weights=np.asarray(T)[sub["faces"] - 1])
    disc.set_halo(sub["n_owned"], sub["neighbors"], sub["send"], sub["recv"])
    law = ConservationLaw(disc, kind, **(law_params or {}))
    law.set_face_trans(np.asarray(T)[sub["faces"] - 1])
    law.set_volumes(np.asarray(vol)[cells])
    if gdz is not None:
        law.set_face_gdz(np.asarray(gdz)[sub["faces"] - 1])
    X0 = np.asarray(X0).reshape(-1, block_n)
    law.set_state(X0[cells].reshape(-1))
    law.set_state0(X0[cells].reshape(-1))
    if sources is not None:
        sc, sv = np.asarray(sources[0], dtype=np.int64), np.asarray(sources[1], dtype=np.float64).reshape(-1, block_n)
        g2l = np.full(np.asarray(part).size, 0, dtype=np.int64)
        g2l[cells[: sub["n_owned"]]] = np.arange(1, sub["n_owned"] + 1)
        keep = g2l[sc - 1] > 0  # forces only on the owning rank
        if keep.any():
            law.set_sources(g2l[sc[keep] - 1], sv[keep].reshape(-1))
    return disc, law, sub
