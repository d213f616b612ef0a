R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3b
python tools/read_probe.py 2>&1 | tail -1; JH_READ_SYNC=1 python tools/read_probe.py 2>&1 | tail -1
python bench.py --no-cpu --steps 40 --path seams > gpurun_out/r3b/seams10.json 2> gpurun_out/r3b/seams10.err; echo rc=$?; tail -2 gpurun_out/r3b/seams10.err
JH_READ_SYNC=1 python bench.py --no-cpu --steps 40 --path seams > gpurun_out/r3b/seams10_sync.json 2> gpurun_out/r3b/seams10_sync.err; echo rc=$?
python bench.py --no-cpu --steps 40 > gpurun_out/r3b/fused10.json 2> gpurun_out/r3b/fused10.err; echo rc=$?
python bench.py --no-cpu --steps 40 --path seams --cells 1250000 > gpurun_out/r3b/seams1.json 2> gpurun_out/r3b/seams1.err; echo rc=$?
python bench.py --no-cpu --steps 40 --cells 1250000 > gpurun_out/r3b/fused1.json 2> gpurun_out/r3b/fused1.err; echo rc=$?
python - <<'PY'
import json
for f in ["seams10","seams10_sync","fused10","seams1","fused1"]:
    try:
        d=json.loads(open(f"gpurun_out/r3b/{f}.json").read().strip().splitlines()[-1])
        se=d["config"].get("seams")
        print(f, d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], se and {k:se[k] for k in ("value_without_output","ms_per_step_without_output","output_ms_per_download")}, d["timing"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench.py tests/test_gpu_julia_sequence.py -x -q -m gpu -k "3M or seam or julia" > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3b/pytest.log
