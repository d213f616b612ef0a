"""Dev tool: does a better block partition (graph bisection + FM, compact 512-cell blocks) pay in BiCGStab iterations / kernel
times?  Compares the library's onion-grown BFS blocks with blocks from dd.partition_graph (ordered inside from their centres)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from jutul_amd import dd
from bench import dims_for_cells
cells = int(os.environ.get("CELLS", "2000000")); rows = int(os.environ.get("ROWS", "512")); steps = int(os.environ.get("STEPS", "30"))
g = ja.tet_lattice_mesh(*dims_for_cells(cells)); nc = g["nc"]; N = g["N"]
T = g["T"] / g["T"].mean(); U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
def run(tag, **kw):
    ctx = ja.HIPContext(0)
    disc = ja.TwoPointPotentialFlowHardCoded(ctx, N, nc, reorder="blocks", **kw)
    law = ja.ConservationLaw(disc, "poisson")
    law.set_face_trans(T); law.set_volumes(g["volumes"]); law.set_state(U0); law.set_state0(U0); law.set_sources([1, nc], [1.0, -1.0])
    prec = ja.ILUZeroPreconditioner(partition="blocks")
    ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-3, max_iterations=100)
    sim = ja.Simulator(law, ks)
    for _ in range(5): sim.perform_step(5.0, 1); law.update_state0()
    ks.profile(enable=8, reset=True)
    ctx.synchronize(); t0 = time.perf_counter()
    reps = []
    for _ in range(steps): reps.append(sim.perform_step(5.0, 1)); law.update_state0()
    ctx.synchronize(); el = time.perf_counter() - t0
    pr = ks.profile(False, True); info = prec.info()
    perm, bp = disc.ordering()
    blk = np.zeros(nc, dtype=np.int64); blk[perm - 1] = np.repeat(np.arange(bp.size - 1), np.diff(bp))
    cut = (blk[N[0] - 1] != blk[N[1] - 1]).mean()
    print(f"{tag:10s} it/s {steps/el:7.2f} lin its {np.mean([r.linear_iterations for r in reps]):6.2f} blocks {info['nblocks']} rows<= {info['max_block_rows']} "
          f"levels<= {info['max_levels']} cut {cut:.3f} spmv {pr['spmv_ms']/max(1,pr['spmv_count']):.4f} ilu {pr['precond_ms']/max(1,pr['precond_count']):.4f} "
          f"fac {np.mean([r.precond_ms for r in reps]):.3f} asm {np.mean([r.assembly_ms for r in reps]):.3f}", flush=True)
run("onion", block_rows=rows)
t0 = time.time(); part = dd.partition_graph(N, nc, max(1, nc // rows), imbalance=0.05); print("partition_graph", round(time.time() - t0, 1), "s", flush=True)
run("bisect+fm", partition=part)
