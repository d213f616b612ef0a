"""Device study: BiCGStab iterations per step of the bench problem with different device-block partitions handed to
jh_tpfa_create as `partition` -- the library's own weighted bisection vs a heavy-edge-matching multilevel prototype (numpy / Python
loops, host side): is a multilevel partitioner (Metis's scheme, partitioning.jl:29-51) worth building into the library?
usage: python tools/partition_study.py [cells = 1253160] [block rows = 256] [steps = 30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import jutul_amd as ja
from jutul_amd import dd
from bench import dims_for_cells

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1253160
brows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
g = ja.tet_lattice_mesh(*dims_for_cells(cells), scramble=True)
nc = g["nc"]; T = g["T"] / g["T"].mean(); vol = g["volumes"]; N = g["N"]; Nf = N - 1
U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
nb = max(1, (nc + brows // 2) // brows)


def hem(Nf, w, n, rng):
    W = sp.csr_matrix((np.r_[w, w], (np.r_[Nf[0], Nf[1]], np.r_[Nf[1], Nf[0]])), shape=(n, n)).tocsr()
    match = -np.ones(n, dtype=np.int64)
    indptr, indices, data = W.indptr, W.indices, W.data
    for v in rng.permutation(n):
        if match[v] >= 0:
            continue
        best, bw = -1, -1.0
        for k in range(indptr[v], indptr[v + 1]):
            u = indices[k]
            if match[u] < 0 and u != v and data[k] > bw:
                best, bw = u, data[k]
        if best >= 0:
            match[v] = best; match[best] = v
        else:
            match[v] = v
    rep = np.minimum(np.arange(n), match)
    uniq, agg = np.unique(rep, return_inverse=True)
    return agg, len(uniq)


def hem_partition(levels):
    rng = np.random.default_rng(0)
    aggs, Nc, wc, n = [], Nf, T, nc
    for _ in range(levels):
        agg, nco = hem(Nc, wc, n, rng)
        aggs.append(agg)
        a, b = agg[Nc[0]], agg[Nc[1]]
        keep = a != b
        key = np.minimum(a[keep], b[keep]) * nco + np.maximum(a[keep], b[keep])
        uk, inv = np.unique(key, return_inverse=True)
        wc = np.bincount(inv, weights=wc[keep]); Nc = np.stack([uk // nco, uk % nco]); n = nco
    part = dd.partition_graph(Nc + 1, n, nb, face_weights=wc)
    for agg in reversed(aggs):
        part = part[agg]
    return part


def run(label, **kw):
    ctx = ja.HIPContext(0)
    t0 = time.time()
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", block_rows=brows, **kw)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(T); law.set_volumes(vol); law.set_state(U0); law.set_state0(U0); law.set_sources([1, nc], [1.0, -1.0])
    prec = ja.ILUZeroPreconditioner(partition="blocks")
    sim = ja.Simulator(law, ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-3, max_iterations=100))
    its = []
    for s in range(steps + 3):
        if s == 3:
            ctx.synchronize(); t1 = time.time()
        rep = sim.perform_step(5.0, 1); law.update_state0(); its.append(int(rep.linear_iterations))
    ctx.synchronize(); el = time.time() - t1
    info = prec.info()
    part = kw.get("partition")
    if part is None:
        perm, bp = disc.ordering(); part = np.empty(nc, dtype=np.int64); part[perm - 1] = np.repeat(np.arange(1, len(bp)), np.diff(bp))
    cut = part[Nf[0]] != part[Nf[1]]
    print(f"{label:46s} its/step {np.mean(its[3:]):6.2f}  {steps / el:7.1f} Newton it/s  blocks {info['nblocks']} max {info['max_block_rows']} levels {info['max_levels']} "
          f"cut faces {cut.mean():.4f} cut weight {T[cut].sum() / T.sum():.4f}", flush=True)


run("library: unweighted bisection")
run("library: weighted bisection (default)", face_weights=T)
run("host: jh_partition_graph weighted -> partition", partition=dd.partition_graph(N, nc, nb, face_weights=T))
for L in (1, 2):
    t0 = time.time(); p = hem_partition(L)
    print(f"  (HEM x{L} prototype: {time.time() - t0:.0f} s on the host, largest block {np.bincount(p).max()})", flush=True)
    run(f"host: HEM x{L} + weighted bisection -> partition", partition=p)
