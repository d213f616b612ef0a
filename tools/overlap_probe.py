"""Dev tool: the per-rank work of the 8-GPU run on ONE GPU, with a real RCCL point-to-point exchange.

Rank 0's subdomain of the 8-way RCB partition (its owned cells + ghost ring) is solved with a 1-rank RCCL communicator
whose halo plan sends to / receives from rank 0 itself: the ghosts receive the values of (arbitrary) owned boundary cells,
so the numbers are not the physical solution, but every launch, the RCCL send/recv kernel, the second stream and the
events are the ones of the multi-GPU path.  JH_OPTIONS=halo_overlap=1 enables the overlapped exchange (default: serial)."""
import os, sys, time
os.environ.pop("NCCL_DEBUG", None)  # the image exports NCCL_DEBUG=VERSION: keep stdout clean
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import jutul_amd as ja
from jutul_amd import dd
from bench import dims_for_cells

cells = int(os.environ.get("CELLS", "10000000")); nparts = int(os.environ.get("PARTS", "8"))
steps = int(os.environ.get("STEPS", "10"))
mesh = ja.tet_lattice_mesh(*dims_for_cells(cells)); nc = mesh["nc"]
T = mesh["T"] / mesh["T"].mean(); U0 = 1.0 + 0.1 * np.random.default_rng(3).random(nc)
part = dd.partition_rcb(mesh["cell_centroids"], nparts)
sub = dd.local_subdomain(mesh["N"], part, 1, ghost_order=os.environ.get("GHOST_ORDER", "owner"))
n_owned, n_local = sub["n_owned"], sub["n_local"]
ctx = ja.HIPContext(0)
ctx.comm_init(1, 0, ja.HIPContext.comm_unique_id())
disc = ja.TwoPointPotentialFlowHardCoded(ctx, sub["N"], n_local, reorder="blocks", n_owned=n_owned)
send = np.concatenate([np.asarray(c) for c in sub["send"]]); recv = np.concatenate([np.asarray(c) for c in sub["recv"]])
send = np.resize(send, recv.size)          # self exchange needs equal counts
disc.set_halo(n_owned, [0], [send], [recv])
if os.environ.get("PUSH") == "1":  # Krylov-loop exchanges by direct stores into the (own) landing buffer instead of RCCL send/recv
    assert ctx.comm_ipc_attach([ctx.comm_ipc_export()])
    ctx.comm_ipc_enable(True)
    assert disc.halo_ipc_attach([disc.halo_ipc_export()], [0], [recv.size])
    disc.halo_ipc_enable(True)
c0 = sub["cells"] - 1
law = ja.ConservationLaw(disc, "poisson")
law.set_face_trans(T[sub["faces"] - 1]); law.set_volumes(mesh["volumes"][c0]); law.set_state(U0[c0]); law.set_state0(U0[c0])
law.set_sources([1], [1.0])
prec = ja.ILUZeroPreconditioner(partition="blocks")
ks = ja.GenericKrylov("bicgstab", preconditioner=prec, relative_tolerance=1e-3, max_iterations=100, precond_side="right")
sim = ja.Simulator(law, ks)
def step():
    rep = sim.perform_step(5.0, 1); law.update_state0(); return rep
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = [step() for _ in range(steps)]
torch.cuda.synchronize(); el = time.perf_counter() - t0
its = [int(r.linear_iterations) for r in reps]
print(f"owned {n_owned} ghosts {n_local - n_owned} split {disc.split()} overlap {'on' if ctx.get_option('halo_overlap') else 'off'} push {os.environ.get('PUSH', '0')}: "
      f"{el / steps * 1e3:.3f} ms/step, {np.mean(its):.1f} its/step, {el / np.sum(its) * 1e6:.1f} us/iteration", flush=True)
ctx.comm_finalize()
