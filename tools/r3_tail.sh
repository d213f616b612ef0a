R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_ipc_allreduce.py -x -q -m gpu > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3e/pytest.log
for v in 1 0; do
  export JH_TAIL_REDUCE=$v
  python bench.py --no-cpu --steps 40 > gpurun_out/r3e/b10_$v.json 2> gpurun_out/r3e/b10_$v.err; echo rc=$?; tail -2 gpurun_out/r3e/b10_$v.err
  python bench.py --no-cpu --steps 40 --cells 1250000 > gpurun_out/r3e/b1_$v.json 2> gpurun_out/r3e/b1_$v.err; echo rc=$?
  PUSH=1 python tools/overlap_probe.py 2>&1 | tail -1
done
python - <<'PY'
import json
for f in ["b10_1","b10_0","b1_1","b1_0"]:
    try:
        d=json.loads(open(f"gpurun_out/r3e/{f}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(f, d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], {a:k[a]["avg_ms"] for a in k}, d["timing"], d["config"]["state_norm"])
    except Exception as e: print(f, "ERR", e)
PY
