"""Several ranks with compute units of their own on ONE GPU (CU-masked contexts, tools/micro/cu_mask_probe.hip): the multi-rank
BiCGStab loop with the single-rank launch count -- dot products all-reduced inside the consuming kernels
(ext/JutulPartitionedArraysExt/krylov.jl:51-105), push-halo hand-shake inside the product kernel (consistent! before mul!,
linalg.jl:37-55).  The checks live in tests/xrank_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, kind, port, extra=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", JH_TEST_KIND=kind)
    env.update(extra or {})
    env.pop("NCCL_DEBUG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "xrank_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"XRANK_OK {world} {kind}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r


@pytest.mark.parametrize("world,kind", [(2, "poisson"), (4, "poisson"), (2, "twophase")])
def test_cu_masked_ranks_consumer_side_allreduce(world, kind):
    """2 and 4 CU-masked ranks: residual histories bit-identical on all ranks and run to run, equal to the reduction-launch path
    to rounding, Newton update equal to the single-process one.  (The oracle pin of this path: tests/test_gpu_zz_round6.py.)"""
    run_world(world, kind, 29680 + world + (10 if kind == "twophase" else 0))


def test_cu_masked_ranks_larger_subdomains():
    """Subdomains large enough for several waves of workgroups per kernel and the 256-row default blocks."""
    run_world(2, "poisson", 29689, {"JH_TEST_DIMS": "48,40,36", "JH_TEST_BLOCK_ROWS": "256"})
