"""Dev tool: where does the SpMV's time go?  Times the CSR tile kernel, the jagged-slice kernel and diagnostic variants of it
(no gather / no store / non-temporal stream loads; wrong results by construction) on the bench matrix.  Run under
rocprofv3 --kernel-trace --stats for per-kernel durations; prints host-timed averages as well."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jutul_amd as ja
from bench import dims_for_cells
cells = int(os.environ.get("CELLS", "10000000"))
g = ja.tet_lattice_mesh(*dims_for_cells(cells)); nc = g["nc"]
ctx = ja.HIPContext(0)
disc = ja.TwoPointPotentialFlowHardCoded(ctx, g["N"], nc, reorder="blocks")
law = ja.ConservationLaw(disc, "poisson")
law.set_face_trans(g["T"] / g["T"].mean()); law.set_volumes(g["volumes"])
U = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
law.set_state(U); law.set_state0(U)
A = ja.StaticSparsityMatrixCSR(disc); r = ja.DeviceVector(disc)
law.update_equation_and_linearized_system(5.0, A, r)
x = ja.DeviceVector(disc, np.random.default_rng(1).standard_normal(nc)); y = ja.DeviceVector(disc)
nnz = disc.nnzb
def timeit(label, fn, reps=40):
    for _ in range(3): fn()
    ctx.synchronize(); ctx.timer_start()
    for _ in range(reps): fn()
    ms = ctx.timer_stop_ms() / reps
    print(f"{label:28s} {ms*1e3:8.1f} us   {(12.0*nnz + 20.0*nc)/ms/1e6:8.1f} GB/s (12 nnz + 20 n)", flush=True)
timeit("csr tile kernel", lambda: ja.mul_(y, A, x))
os.environ["JH_OPTIONS"] = "jds_keep=1"
ja.mul_(y, A, x, jagged=True)
timeit("jagged", lambda: ja.mul_(y, A, x, jagged=True))
# streaming references on the same device: copy and dot of nnz-sized arrays
a = ja.DeviceVector(disc); b = ja.DeviceVector(disc)
timeit("vec copy 80MB (ref)", lambda: a.copy_from(b))
