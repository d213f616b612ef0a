"""CPU study (no GPU): which face weights should the device-block bisection be given?  The library cuts the blocks on a planning
context, the ORACLE factors block-Jacobi ILU(0) in the device's order and runs the bench sequence (Poisson law, dt = 5, BiCGStab
rtol 1e-3) -- the same iteration counts the device shows (unweighted 25.0 / T-weighted 18.0 here, 24.7 / 18.9 on the MI355X at 1.25M
cells).  Variants: T, T^2, sqrt(T), T scaled by the diagonals (AMG-style strength of connection).
usage: python tools/weight_variants_oracle.py [nx,ny,nz = 55,55,55]      (result: profiles/r05_ordering_study.txt, section 7)"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import jutul_amd as ja
from oracle import oracle as o
o.build()
dims = (55, 55, 55) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1].split(","))
g = ja.tet_lattice_mesh(*dims, scramble=True)
N, nc = g["N"] - 1, g["nc"]
T = g["T"] / g["T"].mean(); vol = g["volumes"]; dt = 5.0
U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
src = np.zeros(nc); src[0] = 1.0; src[-1] = -1.0
L = sp.coo_matrix((np.r_[T, T, -T, -T], (np.r_[N[0], N[1], N[0], N[1]], np.r_[N[0], N[1], N[1], N[0]])), shape=(nc, nc)).tocsr()
A = (L + sp.diags(vol / dt)).tocsr()
diag = A.diagonal()
ctx = ja.HIPContext("host")
def run(label, weights, steps=9):
    d = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks", face_weights=weights)
    perm, bp = d.ordering(); p = perm - 1
    nb = bp.size - 1
    Ad = A[p][:, p].tocsr(); Ad.sort_indices()
    part = np.repeat(np.arange(1, nb + 1), np.diff(bp))
    blk = np.empty(nc, dtype=np.int64); blk[p] = part
    cut = blk[N[0]] != blk[N[1]]
    rp, ci, nz = Ad.indptr.astype(np.int64) + 1, Ad.indices.astype(np.int64) + 1, Ad.data
    M = o.ILU0(nc, 1, rp, ci, nz, partition=part)
    U = U0.copy(); its = []
    for s in range(steps):
        r = (L @ U - src)          # accumulation term is zero at the start of a step (U == U_old)
        x, st = o.bicgstab(nc, 1, rp, ci, nz, -r[p], prec=M, side="right", rtol=1e-3, itmax=100)
        dx = np.empty(nc); dx[p] = x
        U = U + dx; its.append(st["iterations"])
    print(f"{label:34s} blocks {nb} cut faces {cut.mean():.4f} cut weight {T[cut].sum() / T.sum():.4f} its {its}  mean(last 5) {np.mean(its[-5:]):.2f}", flush=True)
run("unweighted", None)
run("T (default)", T)
run("T^2", T ** 2)
run("T^0.5", np.sqrt(T))
run("T / sqrt(a_ii a_jj)", T / np.sqrt(diag[N[0]] * diag[N[1]]))
run("T / min(a_ii, a_jj)", T / np.minimum(diag[N[0]], diag[N[1]]))
run("T / max(a_ii, a_jj)", T / np.maximum(diag[N[0]], diag[N[1]]))
