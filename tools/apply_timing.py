"""Dev tool: where a block's wavefront spends its cycles inside ilu_apply_jds_kernel (library built with
JH_EXTRA_FLAGS=-DJH_APPLY_TIMING python jutul.jl_amd/build.py --force).  usage: python tools/apply_timing.py [cells]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
ctx = ja.HIPContext(0)
g = ja.tet_lattice_mesh(*dims_for_cells(cells))
nc = g["nc"]
disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks")
law = ja.ConservationLaw(disc, "poisson")
law.set_face_trans(g["T"] / g["T"].mean()); law.set_volumes(g["volumes"])
U = 1.0 + 0.1 * np.random.default_rng(0).random(nc)
law.set_state(U); law.set_state0(U); law.set_sources([1, nc], [1.0, -1.0])
lsys = ja.LinearizedSystem(disc)
law.update_equation_and_linearized_system(5.0, lsys.jac, lsys.r)
prec = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-30, max_iterations=12)
from jutul_amd import _L
lib = _L()
lib.jh_debug_apply_times.argtypes = [C.c_int64, C.POINTER(C.c_double)]
out = (C.c_double * 7)()
nb = prec.info()["nblocks"]
for rep in range(3):
    ja.linear_solve(lsys, ks, update_preconditioner=False)
    ctx.synchronize()
    assert lib.jh_debug_apply_times(nb, out) == 0
    names = ["prologue", "gather", "fwd sweep", "bwd sweep", "scatter", "launch span", "wave lifetime"]
    print(f"cells {nc} blocks {nb}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, out)), flush=True)
