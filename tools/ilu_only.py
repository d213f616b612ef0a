"""Dev tool: runs a handful of ILU applies + SpMV + assembly on the 10M mesh (for PMC profiling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells
cells = int(os.environ.get("CELLS", "10000000")); b = int(os.environ.get("BLOCK", "512"))
nx, ny, nz = dims_for_cells(cells)
mesh = ja.tet_lattice_mesh(nx, ny, nz); nc = mesh["nc"]
ctx = ja.HIPContext(0)
disc = ja.TwoPointPotentialFlowHardCoded(ctx, mesh["N"], nc, reorder="blocks", block_rows=b)
law = ja.ConservationLaw(disc, "poisson")
law.set_face_trans(mesh["T"] / mesh["T"].mean()); law.set_volumes(mesh["volumes"])
U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
law.set_state(U0); law.set_state0(U0)
lsys = ja.LinearizedSystem(disc)
for _ in range(3):
    law.update_equation_and_linearized_system(5.0, lsys.jac, lsys.r)
prec = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
x = ja.DeviceVector(disc)
for _ in range(5):
    prec.apply(x, lsys.r)
    ja.mul_(lsys.dx, lsys.jac, x)
ctx.synchronize()
print("done", prec.info())
