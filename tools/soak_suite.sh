cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/soak
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/soak/pytest_$i.log 2>&1; echo "run $i rc=$?"; grep -E "passed|failed" gpurun_out/soak/pytest_$i.log | tail -1; done
( time python bench.py > gpurun_out/soak/bench_default.json 2> gpurun_out/soak/bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/soak/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["traffic_note"])
print(json.dumps(d["cpu_baseline"])[:1500])
PY
