cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2p
JH_SETUP_TIMING=1 python bench.py --no-cpu --steps 60 --warmup 5 > gpurun_out/r2p/b.json 2> gpurun_out/r2p/b.err; grep "setup\]" gpurun_out/r2p/b.err | head -40
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2p/b.json").read().strip().splitlines()[-1])
print("it/s", d["value"], "its", d["config"]["linear_iterations_per_step"], "setup", d["config"]["setup_s"], d["config"].get("setup_phases_s"))
PY
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2p/pytest.log
