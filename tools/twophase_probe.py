"""Dev tool: 5M-cell two-phase (N = 2) timings: assembly, block SpMV, block ILU(0) apply/factor, BiCGStab."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells
cells = int(os.environ.get("CELLS", "5000000")); b = int(os.environ.get("BLOCK", "512"))
g = ja.tet_lattice_mesh(*dims_for_cells(cells)); nc = g["nc"]
ctx = ja.HIPContext(0)
disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, block_n=2, reorder="blocks", block_rows=b)
law = ja.ConservationLaw(disc, "twophase", rho0=(1.0, 0.8), compressibility=(1e-3, 2e-3), viscosity=(1.0, 2.0), p_ref=1.0)
law.set_face_trans(g["T"] / g["T"].mean()); law.set_volumes(g["volumes"])
rng = np.random.default_rng(1)
X0 = np.stack([1.0 + 0.05 * rng.random(nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)
law.set_state(X0); law.set_state0(X0)
law.set_sources([1, nc], np.array([[0.01, 0.01], [-0.01, -0.01]]).reshape(-1))
prec = ja.ILUZeroPreconditioner(partition="blocks")
ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-3, max_iterations=200)
sim = ja.Simulator(law, ks)
dt = 0.5
sim.perform_step(dt, 1)
ks.profile(True, True)
ctx.synchronize(); t0 = time.perf_counter()
reps = [sim.perform_step(dt, 1) for _ in range(3)]
ctx.synchronize(); el = time.perf_counter() - t0
pr = ks.profile(False, True)
nnzb, n = disc.nnzb, nc
info = prec.info()
asm = np.mean([r.assembly_ms for r in reps])
B_asm = (24 * 2 + 4 + 8 * 2 + 8 * 4) * n + (4 + 8 + 8 * 4) * disc.nhf
B_spmv = (8 * 4 + 4) * nnzb + (4 + 16 * 2) * n
B_ilu = (8 * 4 + 4) * (info["l_entries"] + info["u_entries"]) + (32 + 8 + 16 * 2) * n
print("its/s", round(3 / el, 2), "lin its", np.mean([r.linear_iterations for r in reps]), info)
print("assembly ms", round(asm, 3), "GB/s", round(B_asm / asm / 1e6, 1))
sp = pr["spmv_ms"] / pr["spmv_count"]; il = pr["precond_ms"] / pr["precond_count"]
print("spmv ms", round(sp, 3), "GB/s", round(B_spmv / sp / 1e6, 1))
print("ilu apply ms", round(il, 3), "GB/s", round(B_ilu / il / 1e6, 1))
print("factor ms", np.mean([r.precond_ms for r in reps]), "solve ms", np.mean([r.linear_solve_ms for r in reps]))
