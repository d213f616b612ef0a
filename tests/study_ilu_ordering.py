"""CPU study with the oracle (test infrastructure, not collected by pytest): BiCGStab iterations of the block-Jacobi ILU(0) against
block partition and in-block elimination order over the bench step sequence.
usage: python tests/study_ilu_ordering.py [cells] [block rows] [dt]; results: profiles/r05_ordering_study.txt"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee, breadth_first_order
import jutul_amd as ja
from jutul_amd import dd
from oracle import oracle as o
from bench import dims_for_cells

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
brows = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = ja.tet_lattice_mesh(*dims_for_cells(cells), scramble=True)
nc = g["nc"]; T = g["T"] / g["T"].mean(); vol = g["volumes"]
U = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
osys = o.TPFASystem(g["N"], nc)
DT = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
law = o.Law("poisson", DT, rho0=(1.0, 1.0), comp=(1e-3, 1e-3), mu=(1.0, 1.0), p_ref=1.0)
nz, r = osys.assemble(law, U, U, vol, T, src_cells=[1, nc], src_values=[1.0, -1.0])
rp, ci = osys.rowptr - 1, osys.colidx - 1
A = sp.csr_matrix((nz, ci, rp), shape=(nc, nc))
Nf = g["N"] - 1
G = sp.csr_matrix((np.ones(Nf.shape[1]), (Nf[0], Nf[1])), shape=(nc, nc)); G = (G + G.T).tocsr()

def solve(perm, part_of_perm, label):
    """perm: new -> old; part_of_perm: 1-based block id of every NEW row.  Runs the bench's step sequence (U <- U - x, U0 <- U)
    with this preconditioner and reports the BiCGStab iterations of steps 1..10"""
    Ap = A[perm][:, perm].tocsr(); Ap.sort_indices()
    F = o.ILU0(nc, 1, Ap.indptr + 1, Ap.indices + 1, Ap.data, partition=part_of_perm)
    Uc = U.copy(); its = []
    for step in range(10):
        nzs, rs = osys.assemble(law, Uc, Uc, vol, T, src_cells=[1, nc], src_values=[1.0, -1.0])
        x, st = o.bicgstab(nc, 1, Ap.indptr + 1, Ap.indices + 1, Ap.data, rs[perm], prec=F, side="right", rtol=1e-3, atol=1e-12, itmax=300)
        xo = np.empty(nc); xo[perm] = x
        Uc = Uc - xo
        its.append(st["iterations"])
    print(f"{label:62s} its/step {its}  mean(4..10) {np.mean(its[3:]):6.2f}", flush=True)

# reference-shaped: natural numbering of the generator is lost (scrambled); 16 RCB blocks, rows in given order
part16 = dd.partition_rcb(g["cell_centroids"], 16)
order = np.argsort(part16, kind="stable")
solve(order, part16[order], "16 RCB blocks, scrambled order inside")
cen = g["cell_centroids"]
nb = max(1, nc // brows)
pg = dd.partition_graph(g["N"], nc, nb)
order = np.argsort(pg, kind="stable")
bp = np.concatenate([[0], np.cumsum(np.bincount(pg[order] - 1, minlength=nb))])
solve(order, pg[order], f"{nb} bisection blocks (~{brows}), scrambled order inside")

def inblock(order_fn, label):
    perm = np.empty(nc, dtype=np.int64)
    for b in range(nb):
        rows = order[bp[b]:bp[b + 1]]
        Gb = G[rows][:, rows].tocsr()
        perm[bp[b]:bp[b + 1]] = rows[order_fn(Gb, rows)]
    solve(perm, pg[order], f"{nb} bisection blocks, {label}")

def bfs_centre(Gb, rows):
    # pseudo-centre: BFS from 0 -> farthest u -> BFS from u -> middle of path ~ take the vertex at half depth; approximate by double sweep
    o1, _ = breadth_first_order(Gb, 0, directed=False)
    o2, pred = breadth_first_order(Gb, o1[-1], directed=False)
    # walk back from the far end half way
    v = o2[-1]; path = [v]
    while pred[v] >= 0: v = pred[v]; path.append(v)
    c = path[len(path) // 2]
    o3, _ = breadth_first_order(Gb, c, directed=False)
    rest = np.setdiff1d(np.arange(len(rows)), o3)
    return np.concatenate([o3, rest])
def rcm(Gb, rows): return reverse_cuthill_mckee(Gb, symmetric_mode=True)
def cm(Gb, rows): return reverse_cuthill_mckee(Gb, symmetric_mode=True)[::-1]
def bfs_rim(Gb, rows):
    o1, _ = breadth_first_order(Gb, 0, directed=False)
    o2, _ = breadth_first_order(Gb, o1[-1], directed=False)
    rest = np.setdiff1d(np.arange(len(rows)), o2)
    return np.concatenate([o2, rest])
def lex(Gb, rows):
    c = cen[:, rows] if cen.shape[0] == 3 else cen[rows].T; return np.lexsort((c[0], c[1], c[2]))
def rev_centre(Gb, rows): return bfs_centre(Gb, rows)[::-1]
inblock(bfs_centre, "BFS from the block centre (device today)")
inblock(rev_centre, "reverse BFS from the centre")
inblock(bfs_rim, "BFS from a pseudo-peripheral cell")
inblock(rcm, "RCM")
inblock(cm, "Cuthill-McKee")
inblock(lex, "lexicographic by centroid (z, y, x)")


# ---- round 5 addendum: the weighted blocks (jh_partition_graph with the transmissibilities) and weight-aware in-block orders ----
W = sp.csr_matrix((np.r_[T, T], (np.r_[Nf[0], Nf[1]], np.r_[Nf[1], Nf[0]])), shape=(nc, nc)).tocsr()
pg = dd.partition_graph(g["N"], nc, nb, face_weights=T)
order = np.argsort(pg, kind="stable")
bp = np.concatenate([[0], np.cumsum(np.bincount(pg[order] - 1, minlength=nb))])
def strong_dfs(Gb, rows):
    """greedy chain: continue with the strongest-coupled unvisited neighbour of the current row, else of the most recent rows"""
    Wb = W[rows][:, rows].tocsr()
    n = len(rows); seen = np.zeros(n, bool); out = []
    ip, ix, dv = Wb.indptr, Wb.indices, Wb.data
    for s0 in range(n):
        if seen[s0]: continue
        stack = [s0]; seen[s0] = True
        while stack:
            v = stack.pop(); out.append(v)
            nb_ = [(dv[k], ix[k]) for k in range(ip[v], ip[v + 1]) if not seen[ix[k]]]
            nb_.sort()                       # weakest first -> strongest is popped next
            for w_, u in nb_:
                seen[u] = True; stack.append(u)
    return np.array(out)
def strong_bfs(Gb, rows):
    """breadth-first from the block centre, neighbours taken strongest first"""
    base = bfs_centre(Gb, rows)
    Wb = W[rows][:, rows].tocsr()
    n = len(rows); seen = np.zeros(n, bool); out = []
    ip, ix, dv = Wb.indptr, Wb.indices, Wb.data
    for s0 in base:
        if seen[s0]: continue
        q = [s0]; seen[s0] = True; h = 0
        while h < len(q):
            v = q[h]; h += 1; out.append(v)
            nb_ = sorted(((dv[k], ix[k]) for k in range(ip[v], ip[v + 1]) if not seen[ix[k]]), reverse=True)
            for w_, u in nb_:
                seen[u] = True; q.append(u)
    return np.array(out)
print("--- weighted blocks ---")
inblock(bfs_centre, "weighted blocks, BFS from the centre (device today)")
inblock(rcm, "weighted blocks, RCM")
inblock(strong_bfs, "weighted blocks, centre BFS, strongest neighbour first")
inblock(strong_dfs, "weighted blocks, strongest-neighbour chains (DFS)")
