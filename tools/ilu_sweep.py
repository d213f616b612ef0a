"""Dev tool: one mesh, sweep ILU block size / threads; prints its/s, BiCGStab its, apply ms."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells

cells = int(os.environ.get("CELLS", "10000000"))
dt = float(os.environ.get("DT", "5.0"))
blocks = [int(x) for x in os.environ.get("BLOCKS", "256,512,1024,2048").split(",")]
threads = [int(x) for x in os.environ.get("THREADS", "64,128,256").split(",")]
nx, ny, nz = dims_for_cells(cells)
mesh = ja.tet_lattice_mesh(nx, ny, nz)
nc = mesh["nc"]
T = mesh["T"] / mesh["T"].mean()
U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
ctx = ja.HIPContext(0)
print("block_rows threads its/s lin_its nblocks maxlev apply_ms spmv_ms factor_ms solve_ms", flush=True)
for b in blocks:
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, mesh["N"], nc, reorder="blocks", block_rows=b)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(T); law.set_volumes(mesh["volumes"]); law.set_sources([1, nc], [1.0, -1.0])
    for t in threads:
        os.environ["JH_ILU_THREADS"] = "64"; os.environ["JH_ILU_LPG"] = str(t)
        law.set_state(U0); law.set_state0(U0)
        prec = ja.ILUZeroPreconditioner(partition="blocks")
        ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-3, max_iterations=200)
        sim = ja.Simulator(law, ks)
        sim.perform_step(dt, 1); law.update_state0()
        ks.profile(True, True)
        ctx.synchronize(); t0 = time.perf_counter()
        reps = []
        for _ in range(3):
            reps.append(sim.perform_step(dt, 1)); law.update_state0()
        ctx.synchronize(); el = time.perf_counter() - t0
        pr = ks.profile(False, True)
        info = prec.info()
        print(b, t, round(3 / el, 2), np.mean([r.linear_iterations for r in reps]), info["nblocks"], info["max_levels"],
              round(pr["precond_ms"] / max(pr["precond_count"], 1), 4), round(pr["spmv_ms"] / max(pr["spmv_count"], 1), 4),
              round(np.mean([r.precond_ms for r in reps]), 3), round(np.mean([r.linear_solve_ms for r in reps]), 2), flush=True)
        del sim, ks, prec
