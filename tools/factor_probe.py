"""Dev tool: time the ILU(0) refactorisation alone (JH_ILU_FACTOR_DEBUG variants give wrong factors but show the phase costs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells, LAW_PAR, initial_state, source_values
law_name = os.environ.get("LAW", "poisson"); cells = int(os.environ.get("CELLS", "10000000" if law_name == "poisson" else "5000000"))
g = ja.tet_lattice_mesh(*dims_for_cells(cells)); nc = g["nc"]; Nb = 2 if law_name == "twophase" else 1
ctx = ja.HIPContext(0)
disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=Nb, reorder="blocks")
law = ja.ConservationLaw(disc, law_name, **LAW_PAR[law_name])
law.set_face_trans(g["T"] / g["T"].mean()); law.set_volumes(g["volumes"]); U0 = initial_state(np, law_name, nc); law.set_state(U0); law.set_state0(U0)
prec = ja.ILUZeroPreconditioner(partition="blocks")
ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-3, max_iterations=100)
sim = ja.Simulator(law, ks)
law.update_equation_and_linearized_system(5.0 if law_name == "poisson" else 0.5, sim.lsys.jac, sim.lsys.r)
prec.update_preconditioner(sim.lsys.jac)
for _ in range(3): prec.update_preconditioner(sim.lsys.jac)
ctx.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): prec.update_preconditioner(sim.lsys.jac)
ctx.synchronize(); print(law_name, "dbg", os.environ.get("JH_ILU_FACTOR_DEBUG", "0"), "threads", os.environ.get("JH_ILU_FACTOR_THREADS", "512"), "factor ms", round((time.perf_counter() - t0) / n * 1e3, 4), flush=True)
