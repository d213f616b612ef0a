R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_unstructured.py -x -q -m gpu > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r3c/pytest.log
for m in delaunay polyhedral; do
timeout 900 python bench.py --no-cpu --steps 20 --mesh $m --cells 2000000 > gpurun_out/r3c/b_$m.json 2> gpurun_out/r3c/b_$m.err; echo "rc=$?"; tail -3 gpurun_out/r3c/b_$m.err
done
python - <<'PY'
import json
for f in ["b_delaunay","b_polyhedral"]:
    try:
        d=json.loads(open(f"gpurun_out/r3c/{f}.json").read().strip().splitlines()[-1])
        c=d["config"]; k=d["roofline"]["kernels"]
        print(f, c["cells"], c["faces"], d["value"], d["ms_per_step"], c["linear_iterations_per_step"], c["kernels_selected"], c["ilu_kept_fraction"], c["ilu_blocks"], c["ilu_max_levels"], {a:(k[a]["avg_ms"],k[a]["frac"]) for a in k}, c["setup_phases_s"])
    except Exception as e: print(f, "ERR", e)
PY
