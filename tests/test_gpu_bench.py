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
    assert set(r["kernels"]) in ({"assembly", "spmv", "ilu0_apply", "ilu0_factor"}, {"assembly", "ilu0_apply+spmv", "ilu0_factor"})
    assert "traffic" in r and "traffic_note" in r
    assert 0 < r["frac_strict"] <= r["frac"] + 1e-9 and all("frac_strict" in k for k in r["kernels"].values())


def test_bench_single_process_with_cpu_baseline():
    d = run([sys.executable, "bench.py", "--cells", "200000", "--steps", "3", "--warmup", "1", "--cpu-cells", "50000"])
    check(d, 3, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "single"
    # same work on both legs: the CPU leg advances the state through the same warm-up + step sequence
    assert abs(c["linear_iterations_per_step"] - d["config"]["linear_iterations_per_step"]) <= 6
    assert c["linear_iterations_first_steps"][0] <= c["linear_iterations_first_steps"][-1]


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


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the two ranks itself (simulate_parray builds all ranks
    itself, ext/JutulPartitionedArraysExt/interface.jl:2-97), rank 0 prints the one line, n_gpus == 2.  On a box with fewer
    devices than ranks the ghost exchange falls back to the host callback (reported in config), on an 8-GPU node the same
    command runs one rank per GPU over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "JH_BENCH_HALO")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--cells", "200000", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    check(d, 2, 1)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["launcher"] == "self" and c["parallelism"] == "dd2"
    assert c["comm_timeouts"] == 0 and c["scalar_allreduce"] in ("mailbox", "rccl") and c["krylov_halo"] in ("push", "rccl", "host-callback")
    import torch
    if torch.cuda.device_count() >= 2:
        assert c["devices_used"] == 2 and c["rccl_ranks"] == 2 and c["halo"] == "rccl"
    else:
        assert c["devices_used"] == 1 and c["halo"].startswith("host-callback")


def test_bench_fails_loudly_when_ranks_do_not_match_gpus():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "50000", "--no-cpu"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "FATAL" in p.stderr and "--gpus 2" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_two_phase_law():
    """configs[3] in the measured bench: --law twophase (2x2 blocks), block byte formulas of SURVEY 8(d)."""
    d = run([sys.executable, "bench.py", "--law", "twophase", "--cells", "200000", "--steps", "3", "--warmup", "1", "--no-cpu"])
    check(d, 3, 1)
    c = d["config"]
    assert c["law"] == "twophase" and c["block_n"] == 2
    k = d["roofline"]["kernels"]
    nc, nf = c["cells"], c["faces"]
    assert k["assembly"]["algorithmic_bytes"] == (24 * 2 + 4 + 8 * 2 + 8 * 4) * nc + (4 + 8 + 8 * 4) * 2 * nf
    b_spmv = (8 * 4 + 4) * (nc + 2 * nf) + (4 + 16 * 2) * nc + 8 * 2 * nc
    assert (k["spmv"]["algorithmic_bytes"] if "spmv" in k else k["ilu0_apply+spmv"]["parts"]["spmv_bytes"]) == b_spmv


def test_bench_rccl_fallback_paths_two_devices():
    """JH_BENCH_NO_MAILBOX=1 JH_BENCH_NO_PUSH=1 with real RCCL between two devices: the result must match the mailbox / push
    run and the 1-rank run (state norm) -- only runs where two devices are visible."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    base = [sys.executable, "bench.py", "--cells", "200000", "--steps", "3", "--warmup", "1", "--no-cpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "JH_BENCH_HALO")}
    one = run(base, env=env)
    fast = run(base + ["--gpus", "2"], env=env)
    slow = run(base + ["--gpus", "2"], env=dict(env, JH_BENCH_NO_MAILBOX="1", JH_BENCH_NO_PUSH="1"))
    assert slow["config"]["scalar_allreduce"] == "rccl" and slow["config"]["krylov_halo"] == "rccl"
    for d in (fast, slow):
        assert d["config"]["rccl_ranks"] == 2 and d["config"]["comm_timeouts"] == 0
        assert abs(d["config"]["state_norm"] - one["config"]["state_norm"]) <= 1e-5 * one["config"]["state_norm"]


def test_bench_seam_path_matches_fused_path():
    """--path seams: the per-Newton entry-point sequence of julia/JutulHIP.jl (through jutul.jl_amd/julia_mirror.py) reaches the
    same state as the fused jh_newton_step path, reports the state downloads it made, and both law kinds run."""
    base = [sys.executable, "bench.py", "--cells", "200000", "--steps", "4", "--warmup", "1", "--no-cpu"]
    fused = run(base)
    seams = run(base + ["--path", "seams"])
    check(seams, 4, 1)
    c = seams["config"]
    assert c["path"] == "seams" and fused["config"]["path"] == "fused" and fused["config"]["seams"] is None
    assert c["seams"]["output_states_downloaded"] == 4 and c["seams"]["value_without_output"] > 0
    assert abs(c["state_norm"] - fused["config"]["state_norm"]) <= 1e-9 * fused["config"]["state_norm"]
    assert c["linear_iterations_first_steps"] == fused["config"]["linear_iterations_first_steps"]
    two = run(base + ["--path", "seams", "--law", "twophase", "--report-every", "2"])
    assert two["config"]["seams"]["output_states_downloaded"] == 2 and two["config"]["block_n"] == 2


def test_bench_selftest_only_two_processes():
    """`bench.py --gpus N --selftest-only`: the pre-flight of a multi-GPU run -- communicator, mailboxes, push halo, one ghost
    exchange of the global cell ids, one all-reduce per path -- prints which path every exchange takes and exits non-zero on a
    fallback or wrong data.  Two processes on the box's one GPU (host-callback state halo requested, like the test above)."""
    env = dict(os.environ, JH_BENCH_HALO="host", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", "bench.py", "--gpus", "2", "--cells", "200000", "--selftest-only"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    st = d["selftest"]
    assert d["ok"] and st["ranks"] == 2 and st["state_halo_ok"] and st["allreduce_ok"]["negotiated"] and st["ghost_rows"] > 0
    assert st["paths"] == {"state_halo": "host-callback", "scalar_allreduce": "mailbox", "krylov_halo": "push"} and not st["problems"]
    # the same with the push halo switched off on request: host-callback Krylov halo is then the requested path, not a fallback
    p = subprocess.run(cmd[:9] + ["29548"] + cmd[10:], cwd=ROOT, env=dict(env, JH_BENCH_NO_PUSH="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert d["ok"] and d["selftest"]["paths"]["krylov_halo"] == "host-callback"


def test_bench_nonlinear_timesteps_line():
    """bench.py --law compressible --timesteps (the nonlinear line of configs[4]): a step is a time step through
    Simulator.solve_timestep; value counts the Newton iterations that solved a linear system, config.nonlinear reports the loop."""
    d = run([sys.executable, "bench.py", "--cells", "200000", "--steps", "4", "--warmup", "1", "--no-cpu", "--law", "compressible",
             "--compressibility", "4", "--newton-tol", "1e-8", "--rtol", "1e-6", "--max-newton", "6", "--timesteps"])
    check(d, 4, 1)
    n = d["config"]["nonlinear"]
    assert n["timesteps"] == 4 and n["ministeps"] >= 4 and n["timestep_cuts"] == n["ministeps"] - 4
    assert n["newton_iterations"] >= 2 * 4  # genuinely nonlinear: more than one Newton update per time step
    assert n["assemblies"] == n["newton_iterations"] + n["ministeps"]  # every ministep ends with an assembly + convergence test
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 4 - n["newton_iterations"]) < 1e-2 * n["newton_iterations"]
    assert "workload" in d["config"] and "full Newton loop" in d["config"]["workload"]
    assert d["roofline"]["frac_strict"] <= d["roofline"]["frac"] + 1e-9
