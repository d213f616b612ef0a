"""Fuzz of the set-up (no GPU): random grids -- 1-D / 2-D / 3-D Cartesian pieces, one or several, scrambled or not, with duplicated faces,
weights, ghost cells, block sizes and block widths -- built through a planning context with 1, 3 and 8 host threads; every table the
device would be handed (plan_checksum) must agree.  Sizes straddle the thresholds of the thread teams (131 072 cells) and of the
private subgraphs (65 536 / 40 000).       python tools/setup_fuzz.py [seed = 0] [seconds = 300]; JH_FUZZ_CASES=n stops after n cases, JH_FUZZ_PRINT=1
prints every case's checksum (A/B of two builds: JUTUL_HIP_LIB=<other .so>, same seed, diff the outputs)      (649 cases in session 4 of round 5)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def lattice_like(nc, deg, rng, pieces=1):
    # random geometric-ish graph: points on a line/grid connected to near neighbours in a random permutation space
    parts = []
    off = 0
    for p in range(pieces):
        m = nc // pieces
        dims = rng.integers(1, 4)
        if dims == 1: shape = (m,)
        elif dims == 2: a = int(rng.integers(2, max(3, int(m ** 0.5) * 2))); shape = (a, max(1, m // a))
        else: a = int(rng.integers(2, max(3, int(m ** (1/3)) * 2))); b = int(rng.integers(2, max(3, int(m ** (1/3)) * 2))); shape = (a, b, max(1, m // (a * b)))
        N = ja.cartesian_neighbors(shape) + off
        parts.append(N); off += int(np.prod(shape))
    N = np.concatenate(parts, axis=1)
    n = off
    if rng.random() < 0.5:  # scramble
        perm = rng.permutation(n) + 1
        N = perm[N - 1]
    if rng.random() < 0.3:  # multigraph: duplicate some faces
        k = int(N.shape[1] * 0.01) + 1
        N = np.concatenate([N, N[:, rng.integers(0, N.shape[1], k)]], axis=1)
    return N, n
t0 = time.time(); cases = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 300) and cases < int(os.environ.get("JH_FUZZ_CASES", "1000000")):
    nc_t = int(rng.choice([30000, 41000, 66000, 70000, 131072, 140000, 200000, 300000]))
    N, nc = lattice_like(nc_t, 0, rng, pieces=int(rng.choice([1, 1, 2, 5])))
    w = rng.random(N.shape[1]) ** 3 if rng.random() < 0.6 else None
    n_owned = nc if rng.random() < 0.7 else int(nc * 0.9)
    if n_owned < nc:  # ghosts must be the last cells: fine for any graph
        pass
    br = int(rng.choice([0, 64, 128, 512]))
    bn = int(rng.choice([1, 1, 2]))
    sums = []
    for th in ("1", "3", "8"):
        os.environ["JH_SETUP_THREADS"] = th
        ctx = ja.HIPContext("host", plan_checksum=1)
        d = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, block_n=bn, reorder="blocks", block_rows=br, n_owned=n_owned, face_weights=w)
        A = ja.StaticSparsityMatrixCSR(d); A.spmv_info()
        ja.ILUZeroPreconditioner(partition="blocks").symbolic(A)
        perm, bp = d.ordering()
        assert np.array_equal(np.sort(perm), np.arange(1, nc + 1))
        assert np.all(perm[n_owned:] > n_owned) if n_owned < nc else True
        sums.append(ctx.plan_checksum())
    assert sums[0] == sums[1] == sums[2], (nc, N.shape, br, bn, n_owned, sums)
    cases += 1
    if os.environ.get("JH_FUZZ_PRINT"):  # one line per case: two builds of the library run with the same seed must print the same lines
        print(cases, nc, N.shape[1], br, bn, n_owned, sums[0], flush=True)
print("fuzz cases", cases, "all thread counts agree", round(time.time() - t0), "s")
