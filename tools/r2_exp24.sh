cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2x
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py tests/test_gpu_fullsize.py -m gpu -x -q -k "limits or twophase or two_phase or bench" > gpurun_out/r2x/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2x/pytest.log
timeout 900 python bench.py --law twophase > gpurun_out/r2x/p.json 2> gpurun_out/r2x/p.err; tail -2 gpurun_out/r2x/p.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2x/p.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("it/s", d["value"], "steps", d["steps"], "its", d["config"]["linear_iterations_per_step"], d["config"]["linear_iterations_first_steps"], {n:(v["avg_ms"], v["frac"]) for n,v in k.items()}, "norm", d["config"]["state_norm"], d["cpu_baseline"])
PY
