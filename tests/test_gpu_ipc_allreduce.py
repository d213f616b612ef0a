"""Mailbox all-reduce (jh_comm_ipc_*) across PROCESSES: 2 and 4 processes sharing the one GPU of the test box map each other's
mailboxes through hipIpc handles and reduce scalars with the single-wavefront kernel the multi-GPU Krylov loop uses."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_mailbox_allreduce_between_processes(world):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NCCL_DEBUG", None)
    port = 29650 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ipc_allreduce_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"IPC_ALLREDUCE_OK {world}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_mailbox_timeout_names_the_missing_peer():
    """A rank that never arrives in a mailbox all-reduce: the others fail within the time limit with a jh_last_error message
    naming themselves, the missing peer and the epoch (no hang until the job's own time-out)."""
    world = 3
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JH_TEST_TIMEOUT="1", JH_TEST_COMM_TIMEOUT_S="2")
    env.pop("NCCL_DEBUG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", "29657", os.path.join(ROOT, "tests", "ipc_allreduce_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"IPC_TIMEOUT_OK {world}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("world,kind", [(2, "poisson"), (4, "poisson"), (3, "twophase")])
def test_multi_process_newton(world, kind):
    """The distributed Newton step with one PROCESS per rank (all on the test box's single GPU): mailbox all-reduces inside the
    pipelined BiCGStab loop, host-callback ghost exchange; the gathered solution equals the single-process one."""
    # the 3-rank case keeps the reference's ghost order (ascending global id): receive lists are then not contiguous
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JH_TEST_KIND=kind, JH_TEST_GHOST_ORDER="global" if world == 3 else "owner")
    env.pop("NCCL_DEBUG", None)
    port = 29660 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ipc_solver_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"IPC_SOLVER_OK {world} {kind}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
