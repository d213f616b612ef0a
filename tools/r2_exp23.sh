cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2w
for i in 1 2; do
timeout 600 python bench.py --no-cpu --warmup 5 --steps 60 > gpurun_out/r2w/a$i.json 2> gpurun_out/r2w/a$i.err
python - $i <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r2w/a{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; t=d["timing"]
print("it/s", d["value"], "ms/step", d["ms_per_step"], "sum phases", round(sum(t.values()),3), t, "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"])
PY
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -m gpu -x -q -k "simulator or newton or bench or heat" > gpurun_out/r2w/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r2w/pytest.log
