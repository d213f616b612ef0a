R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_unstructured.py -x -q -m gpu > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r3d/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export JH_NO_FUSED_PRODUCT=1; fi
  python bench.py --no-cpu --steps 40 > gpurun_out/r3d/b10_$v.json 2> gpurun_out/r3d/b10_$v.err; echo rc=$?; tail -2 gpurun_out/r3d/b10_$v.err
  python bench.py --no-cpu --steps 40 --cells 1250000 > gpurun_out/r3d/b1_$v.json 2> gpurun_out/r3d/b1_$v.err; echo rc=$?
  python bench.py --no-cpu --steps 40 --law twophase > gpurun_out/r3d/b2_$v.json 2> gpurun_out/r3d/b2_$v.err; echo rc=$?
done
python - <<'PY'
import json
for f in ["b10_0","b10_1","b1_0","b1_1","b2_0","b2_1"]:
    try:
        d=json.loads(open(f"gpurun_out/r3d/{f}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(f, d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], {a:k[a]["avg_ms"] for a in k}, [k[a].get("parts") for a in k if "parts" in k[a]], d["timing"], d["config"]["state_norm"])
    except Exception as e: print(f, "ERR", e)
PY
