"""Dev tool: BiCGStab iteration counts for different orderings / ILU modes (1M cells)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells
cells = int(os.environ.get("CELLS", "1000000")); dt = float(os.environ.get("DT", "5.0"))
ctx = ja.HIPContext(0)
for scramble in (False, True):
    g = ja.tet_lattice_mesh(*dims_for_cells(cells), scramble=scramble); nc = g["nc"]
    T = g["T"] / g["T"].mean(); U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
    for reorder, part, br in (("none", None, 0), ("blocks", "blocks", 512), ("blocks", "blocks", 4096), ("blocks", None, 512)):
        disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder=reorder, block_rows=br)
        law = ja.ConservationLaw(disc, "poisson")
        law.set_face_trans(T); law.set_volumes(g["volumes"]); law.set_sources([1, nc], [1.0, -1.0])
        law.set_state(U0); law.set_state0(U0)
        ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition=part), relative_tolerance=1e-3, max_iterations=300)
        sim = ja.Simulator(law, ks)
        its = []
        for _ in range(4):
            rep = sim.perform_step(dt, 1); law.update_state0(); its.append(rep.linear_iterations)
        print("scramble", scramble, "reorder", reorder, "partition", part, "block_rows", br, "its", its, ks.preconditioner.info()["max_levels"], flush=True)
