"""CPU study (no GPU): what do the couplings that block-Jacobi ILU(0) drops cost, and does the ORDER of the blocks matter if they are
kept?  The library cuts its device blocks on a planning context; the oracle factors (a) block-Jacobi ILU(0) as the device does,
(b) ONE global ILU(0) of the same matrix with the blocks in bisection-tree order, (c) in multicolour order (blocks of a colour do not
touch: the order a device kernel could sweep colour by colour), (d) in random order -- and runs the bench's step sequence.
usage: python tools/multicolour_ilu_oracle.py [nx,ny,nz = 40,40,40]         (result: profiles/r05_ordering_study.txt, section 9)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "40,40,40"]
exec(open(os.path.join(ROOT, "tools", "weight_variants_oracle.py")).read().split('run("unweighted", None)')[0])   # mesh, matrix, oracle, planning context
d = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", face_weights=T)
perm, bp = d.ordering(); p = perm - 1
nb = bp.size - 1
blk_dev = np.repeat(np.arange(nb), np.diff(bp))          # block of device row
blk = np.empty(nc, dtype=np.int64); blk[p] = blk_dev     # block of host cell
# block graph
a, b = blk[N[0]], blk[N[1]]
m = a != b
B = sp.coo_matrix((np.ones(2 * m.sum()), (np.r_[a[m], b[m]], np.r_[b[m], a[m]])), shape=(nb, nb)).tocsr()
# greedy colouring in block order
colour = -np.ones(nb, dtype=np.int64)
for v in range(nb):
    used = set(colour[B.indices[B.indptr[v]:B.indptr[v + 1]]])
    c = 0
    while c in used: c += 1
    colour[v] = c
print("blocks", nb, "colours", colour.max() + 1, np.bincount(colour))
def solve(label, order_rows, part):
    q = p[order_rows]                                     # host cell of new row
    Ad = A[q][:, q].tocsr(); Ad.sort_indices()
    rp, ci, nz = Ad.indptr.astype(np.int64) + 1, Ad.indices.astype(np.int64) + 1, Ad.data
    M = o.ILU0(nc, 1, rp, ci, nz, partition=None if part is None else part[order_rows] + 1)
    U = U0.copy(); its = []
    for s in range(9):
        r = (L @ U - src)
        x, st = o.bicgstab(nc, 1, rp, ci, nz, -r[q], prec=M, side="right", rtol=1e-3, itmax=100)
        dx = np.empty(nc); dx[q] = x
        U = U + dx; its.append(st["iterations"])
    print(f"{label:56s} its {its} mean(last 5) {np.mean(its[-5:]):.2f}", flush=True)
ident = np.arange(nc)
solve("block-Jacobi ILU(0), device order (today)", ident, blk_dev)
solve("GLOBAL ILU(0), device order (bisection-tree block order)", ident, None)
# multicolour block order: blocks sorted by (colour, block id), rows inside a block unchanged
border = np.lexsort((np.arange(nb), colour))
rank_of_block = np.empty(nb, dtype=np.int64); rank_of_block[border] = np.arange(nb)
rows_mc = np.lexsort((np.arange(nc), rank_of_block[blk_dev]))
solve("GLOBAL ILU(0), blocks in multicolour order", rows_mc, None)
rng = np.random.default_rng(0); rb = rng.permutation(nb); rr = np.empty(nb, dtype=np.int64); rr[rb] = np.arange(nb)
solve("GLOBAL ILU(0), blocks in random order", np.lexsort((np.arange(nc), rr[blk_dev])), None)
