"""Worker of tests/test_gpu_ipc_allreduce.py::test_multi_process_newton: the distributed Newton step across PROCESSES that share
GPU 0.  Scalar reductions go through the mailboxes (hipIpc), ghost exchanges through the host-callback backend over gloo (RCCL
cannot put two ranks on one device); everything else -- pipelined BiCGStab loop, fused halo pack, ghost handling -- is the code
the multi-GPU run executes.  Rank 0 compares the gathered solution with a single-process solve."""
import os
import sys

os.environ.pop("NCCL_DEBUG", None)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
import jutul_amd as ja
from jutul_amd import dd

kind = os.environ.get("JH_TEST_KIND", "poisson")
nblk = 2 if kind == "twophase" else 1
g = ja.tet_lattice_mesh(14, 12, 10)
nc = g["nc"]
T = g["T"] / g["T"].mean()
rng = np.random.default_rng(7)
if nblk == 1:
    X0 = 1.0 + 0.1 * rng.random(nc)
else:
    X0 = np.stack([1.0 + 0.1 * rng.random(nc), rng.uniform(0.3, 0.7, nc)]).T.reshape(-1)
par = dict(rho0=(1.0, 0.8), compressibility=(1e-2, 2e-2), viscosity=(1.0, 2.0), p_ref=1.0)
q = 1.0 if nblk == 1 else 0.02
src = ([1, nc], np.array([[q] * nblk, [-q] * nblk]))
dt = 0.5


def make_sim(law):
    ks = ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=1e-11,
                          max_iterations=300, precond_side="right")
    return ja.Simulator(law, ks, tolerance=1e-9)


part = dd.partition_rcb(g["cell_centroids"], world)
ctx = ja.HIPContext(0)
ctx.comm_init_ipc_only(world, rank)
handles = [None] * world
dist.all_gather_object(handles, ctx.comm_ipc_export())
ok = torch.tensor([1 if ctx.comm_ipc_attach(handles) else 0])
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
assert int(ok[0]) == 1
ctx.comm_ipc_enable(True)
disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, rank, T, g["volumes"], X0, kind=kind, block_n=nblk, sources=src,
                                       block_rows=128, law_params=par, ghost_order=os.environ.get("JH_TEST_GHOST_ORDER", "owner"))
ctx.comm_set_halo_callback(dd.packed_exchange(sub))
if os.environ.get("JH_TEST_PUSH", "1") == "1":  # Krylov-loop exchanges by direct stores into the neighbours' landing buffers
    assert dd.setup_push_halo(disc, sub, rank, world)
okk, its, rep = make_sim(law).solve_ministep(dt)
assert okk
Xl = law.get_state().reshape(-1, nblk)
gathered = [None] * world
dist.all_gather_object(gathered, (sub["cells"][: sub["n_owned"]] - 1, Xl[: sub["n_owned"]], its,
                                  sub["cells"][sub["n_owned"]:] - 1, Xl[sub["n_owned"]:]))
dist.barrier()
ctx.comm_finalize()
if rank == 0:
    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, block_n=nblk, reorder="blocks", block_rows=128)
    law0 = ja.ConservationLaw(disc0, kind, **par)
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    ok0, its0, _ = make_sim(law0).solve_ministep(dt)
    assert ok0
    X_ref = law0.get_state().reshape(nc, nblk)
    X = np.zeros_like(X_ref)
    for own, Xo, its_r, gh, Xg in gathered:
        X[own] = Xo
        assert its_r == its0
        assert np.allclose(Xg, X_ref[gh], rtol=1e-7, atol=1e-9)  # ghosts carry the owner's update
    err = np.abs(X - X_ref).max() / np.abs(X_ref).max()
    assert err <= 1e-7, err
    print("IPC_SOLVER_OK", world, kind, its0, f"{err:.2e}", flush=True)
dist.barrier()
