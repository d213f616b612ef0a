cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2t
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "assembly or blocks_come or twophase or two_phase" > gpurun_out/r2t/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2t/pytest.log
timeout 600 python bench.py --no-cpu --law twophase --steps 30 --warmup 5 > gpurun_out/r2t/p5.json 2> gpurun_out/r2t/p5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2t/p5.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("it/s", d["value"], "its", d["config"]["linear_iterations_per_step"], {n:(v["avg_ms"], v["frac"]) for n,v in k.items()}, "norm", d["config"]["state_norm"])
PY
