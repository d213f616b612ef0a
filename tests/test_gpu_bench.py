"""bench.py contract: one JSON line on stdout with the required keys (also through torch.distributed.run and through the
distributed code path on one rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ONE line on stdout
    return json.loads(lines[0])


def check(d, steps, warmup):
    for k in REQUIRED:
        assert k in d, k
    assert d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(r["kernels"]) == {"assembly", "spmv", "ilu0_apply"}


def test_bench_single_process_with_cpu_baseline():
    d = run([sys.executable, "bench.py", "--cells", "200000", "--steps", "3", "--warmup", "1", "--cpu-cells", "50000"])
    check(d, 3, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "single"


def test_bench_under_torchrun_distributed_path():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", "29541", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--cells", "200000", "--no-cpu"],
            env={"JH_BENCH_FORCE_DIST": "1"})
    check(d, 2, 1)
    assert d["cpu_baseline"] is None


def test_bench_two_processes_sharing_the_gpu():
    """bench.py's N > 1 path (RCB partition, rank-local subdomains, mailbox set-up, barrier + max-over-ranks clock, one JSON line
    from rank 0) with two processes on the box's single GPU; RCCL cannot place two ranks on one device, so the ghost exchange
    uses the host-callback backend (JH_BENCH_HALO=host)."""
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29542", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--cells", "200000", "--no-cpu"],
            env={"JH_BENCH_HALO": "host", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    check(d, 2, 1)
    assert d["n_gpus"] == 2
    assert d["config"]["parallelism"] == "dd2" and d["config"]["scalar_allreduce"] == "mailbox"
    assert 5 <= d["config"]["linear_iterations_per_step"] <= 100
