"""Dev tool: where a block spends its cycles inside ilu_factor_rows_kernel (library built with
JH_EXTRA_FLAGS=-DJH_APPLY_TIMING python jutul.jl_amd/build.py --force).  usage: python tools/factor_timing.py [cells] [threads]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = ja.HIPContext(0, ilu_factor_threads=threads)
g = ja.polyhedral_dual_mesh(cells, grading=2.0)
nc = g["nc"]
disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks")
law = ja.ConservationLaw(disc, "poisson")
law.set_face_trans(g["T"] / g["T"].mean()); law.set_volumes(g["volumes"] / g["volumes"].mean())
U = 1.0 + 0.1 * np.random.default_rng(0).random(nc)
law.set_state(U); law.set_state0(U); law.set_sources([1, nc], [1.0, -1.0])
lsys = ja.LinearizedSystem(disc)
law.update_equation_and_linearized_system(5.0, lsys.jac, lsys.r)
prec = ja.ILUZeroPreconditioner(partition="blocks").update_preconditioner(lsys.jac)
from jutul_amd import _L
lib = _L()
lib.jh_debug_factor_times.argtypes = [C.c_int64, C.POINTER(C.c_double)]
out = (C.c_double * 7)()
fi = prec.info()
nb = fi["nblocks"]
for rep in range(3):
    prec.update_preconditioner(lsys.jac)
    ctx.synchronize()
    assert lib.jh_debug_factor_times(nb, out) == 0
    names = ["gather", "program copy", "level loop", "stores", "wave0 in rows", "wave0 rows", "launch span"]
    print(f"cells {nc} blocks {nb} levels {fi['max_levels']} threads {threads}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, out)), flush=True)
