cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2cfg
timeout 600 python bench.py --cells 1000000 > gpurun_out/r2cfg/bench_1M_1gpu.json 2> gpurun_out/r2cfg/a.err; echo rc=$?
timeout 900 python bench.py --gpus 8 --cells 1000000 > gpurun_out/r2cfg/bench_1M_8ranks_shared_gpu.json 2> gpurun_out/r2cfg/b.err; echo rc=$?
timeout 900 python bench.py --gpus 8 > gpurun_out/r2cfg/bench_10M_8ranks_shared_gpu.json 2> gpurun_out/r2cfg/c.err; echo rc=$?
timeout 900 python bench.py --gpus 8 --law twophase > gpurun_out/r2cfg/bench_twophase_5M_8ranks_shared_gpu.json 2> gpurun_out/r2cfg/d.err; echo rc=$?
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2cfg/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]
        print(f.split("/")[-1], d["value"], d["n_gpus"], c["linear_iterations_per_step"], c["ranks_seen"], c["devices_used"], c["scalar_allreduce"], c["krylov_halo"], c["comm_timeouts"], c["setup_s"], round(c["state_norm"],6))
    except Exception as e: print(f, "ERR", e)
PY
