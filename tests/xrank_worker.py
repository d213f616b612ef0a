"""Worker of tests/test_gpu_xrank.py: several ranks as PROCESSES on the one GPU of the test box, each context restricted to a
disjoint set of compute units (HIPContext.set_cu_mask), so that a kernel in which every wavefront waits for a peer rank cannot keep
that peer off the chip -- the situation of one process per GPU.  The BiCGStab loop then runs as on one rank: five launches per
iteration, the dot products all-reduced inside the consuming kernels (mailbox granules), the push-halo hand-shake inside the
product kernel.  Checked: residual histories bit-identical on all ranks and run to run, equal to the reduction-launch path
(xrank_consumer = 0) to rounding, solution equal to the single-process solve."""
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
from jutul_amd._lib import load as _L, check

kind = os.environ.get("JH_TEST_KIND", "poisson")
dims = tuple(int(v) for v in os.environ.get("JH_TEST_DIMS", "20,16,12").split(","))
block_rows = int(os.environ.get("JH_TEST_BLOCK_ROWS", "128"))
nblk = 2 if kind == "twophase" else 1
g = ja.tet_lattice_mesh(*dims)
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
rtol = 1e-9


def krylov():
    return ja.GenericKrylov("bicgstab", preconditioner=ja.ILUZeroPreconditioner(partition="blocks"), relative_tolerance=rtol,
                            max_iterations=300, precond_side="right")


part = dd.partition_rcb(g["cell_centroids"], world)
ctx = ja.HIPContext(0, comm_timeout_ms=60000)
ncu = torch.cuda.get_device_properties(0).multi_processor_count // world
ctx.set_cu_mask(rank * ncu, ncu)  # compute units of this rank's own
ctx.comm_init_ipc_only(world, rank)
handles = [None] * world
dist.all_gather_object(handles, ctx.comm_ipc_export())
ok = torch.tensor([1 if ctx.comm_ipc_attach(handles) else 0])
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
assert int(ok[0]) == 1
ctx.comm_ipc_enable(True)
assert ctx.comm_xrank_selftest()   # (collective)
ORACLE_PIN = os.environ.get("JH_TEST_ORACLE_PIN") == "1"   # (tests/test_gpu_zz_round6.py: written in a round without a GPU)
if ORACLE_PIN:
    assert ctx.comm_devices_distinct() is False   # one GPU ... but every rank is CU-masked: the promise below is accepted
ctx.comm_set_exclusive(True)
assert ctx.comm_info()["consumer_allreduce"]
disc, law, sub = dd.setup_rank_problem(ctx, g["N"], part, rank, T, g["volumes"], X0, kind=kind, block_n=nblk, sources=src,
                                       block_rows=block_rows, law_params=par, ghost_order="owner")
ctx.comm_set_halo_callback(dd.packed_exchange(sub))
assert dd.setup_push_halo(disc, sub, rank, world)
lsys = ja.LinearizedSystem(disc)
law.synchronize_ghosts()
law.update_equation_and_linearized_system(dt, lsys.jac, lsys.r)
check(_L().jh_unit_diagonalize(lsys.jac.h, lsys.r.h, disc.n_owned))
ks = krylov()


def solve(consumer):
    ctx.set_option("xrank_consumer", consumer)
    dist.barrier()
    out = ja.linear_solve(lsys, ks)
    assert out["ok"], out
    path = ks.last_path()  # the path under test ran: no reduction launches, no finish launches
    assert path["consumer_reduce"] == path["consumer_allreduce"] == bool(consumer), path
    # (2x2 blocks multiply with the CSR tile kernel: consumer-side sums, the halo hand-shake keeps its finish launch)
    assert path["halo_fold"] == (bool(consumer) and nblk == 1) and path["jagged"] == (nblk == 1) and path["push_halo"], path
    return out["residuals"].copy(), lsys.dx.download().copy()


nz_dev, b_dev = lsys.jac.nzval.copy(), lsys.r.download().copy()   # what the solves below see (ghost rows -I)
h1, x1 = solve(1)
h2, x2 = solve(1)
h0, x0 = solve(0)
assert np.array_equal(h1, h2) and np.array_equal(x1, x2), "consumer-side all-reduce is not reproducible run to run"
allh = [None] * world
dist.all_gather_object(allh, (h1.tobytes(), h0.tobytes()))
assert all(a[0] == allh[0][0] for a in allh), "residual histories differ between the ranks (consumer-side path)"
assert all(a[1] == allh[0][1] for a in allh), "residual histories differ between the ranks (reduction-launch path)"
# the two paths form the same sums in different orders (wavefront butterfly vs one-workgroup tree): equal to rounding while the
# residual is well above the noise floor, same iteration count +-1
# (BiCGStab amplifies rounding differences along the iteration: the counts of a 60-iteration two-phase solve at rtol 1e-9 may
# differ by a few, the first residuals may not)
m = min(len(h0), len(h1), 10)
assert abs(len(h0) - len(h1)) <= max(1, len(h0) // 20), (len(h0), len(h1))
assert np.allclose(h1[:m], h0[:m], rtol=1e-7), (h1[:m], h0[:m])
# (both solves stop at ||r|| <= 1e-9 ||r0||: the solutions agree to that times the conditioning -- a 2x2 system mixes pressure and
# saturation scales)
assert np.abs(x1 - x0).max() <= (1e-7 if nblk == 1 else 1e-5) * np.abs(x0).max()
# ORACLE PIN of the consumer-side path (scalar systems): rank 0 gathers every rank's matrix, right-hand side, halo plan and device
# ordering and runs the oracle-driven N-rank emulation (tests/_dd_emulation.py: consistent! before every product, dots over owned
# entries summed over the ranks, one block-Jacobi ILU(0) per rank with the ghost input zeroed).  A converged answer is the same
# for any correct reduction; the residual HISTORY through the granule all-reduce is what is compared: first 10 residuals to 1e-8,
# iteration count +-1.
if nblk == 1 and ORACLE_PIN:
    perm_dev, bp_dev = disc.ordering()
    plans = [None] * world
    dist.all_gather_object(plans, dict(sub={k: sub[k] for k in ("n_local", "n_owned", "N", "send", "recv", "neighbors")}, nz=nz_dev, b=b_dev,
                                       perm=perm_dev, bp=bp_dev))
    if rank == 0:
        from oracle import oracle as o
        from tests import _ilu_checks as ck
        from tests._dd_emulation import emulated_bicgstab, oracle_rank
        ranks_o = [oracle_rank(o, ck, pl["sub"], pl["nz"], pl["b"], pl["perm"], pl["bp"]) for pl in plans]
        hist_o, its_o, _ = emulated_bicgstab(o, ranks_o, "right", rtol, itmax=300)
        m = min(11, len(hist_o), len(h1))
        assert m >= 6 and abs((len(h1) - 1) - its_o) <= 1, (len(h1) - 1, its_o)
        dev = np.abs(h1[:m] / hist_o[:m] - 1.0)
        # (BiCGStab amplifies the rounding difference of the two summation orders along the iteration: 1e-8 on the first six residuals,
        # 1e-7 up to the tenth -- the margin is printed)
        assert dev[:6].max() <= 1e-8 and dev.max() <= 1e-7, (h1[:m], hist_o[:m])
        print("XRANK_ORACLE_OK", world, its_o, len(h1) - 1, f"max rel. deviation of the first {m} residuals {dev.max():.1e}", flush=True)
    dist.barrier()
# the Newton update through the fused step (jh_newton_step) on the consumer-side path
ctx.set_option("xrank_consumer", 1)
sim = ja.Simulator(law, krylov(), tolerance=1e-9)
okk, its, rep = sim.solve_ministep(dt)
assert okk
Xl = law.get_state().reshape(-1, nblk)
gathered = [None] * world
dist.all_gather_object(gathered, (sub["cells"][: sub["n_owned"]] - 1, Xl[: sub["n_owned"]], its, len(h1) - 1))
dist.barrier()
ctx.comm_finalize()
if rank == 0:
    ctx0 = ja.HIPContext(0)
    disc0 = ja.TwoPointPotentialFlowHardCoded(ctx0, g["N"], nc, block_n=nblk, reorder="blocks", block_rows=block_rows)
    law0 = ja.ConservationLaw(disc0, kind, **par)
    law0.set_face_trans(T); law0.set_volumes(g["volumes"]); law0.set_state(X0); law0.set_state0(X0)
    law0.set_sources(src[0], src[1].reshape(-1))
    ok0, its0, _ = ja.Simulator(law0, krylov(), tolerance=1e-9).solve_ministep(dt)
    assert ok0
    X_ref = law0.get_state().reshape(nc, nblk)
    X = np.zeros_like(X_ref)
    for own, Xo, its_r, lin in gathered:
        X[own] = Xo
        assert its_r == its0
    err = np.abs(X - X_ref).max() / np.abs(X_ref).max()
    assert err <= 1e-7, err
    print("XRANK_OK", world, kind, gathered[0][3], f"{err:.2e}", flush=True)
dist.barrier()
