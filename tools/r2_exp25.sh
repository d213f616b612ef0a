cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2y
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spmv or bicgstab or multigraph or fullsize" > gpurun_out/r2y/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2y/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 --steps 40 $BENCH_ARGS > gpurun_out/r2y/$tag.json 2> gpurun_out/r2y/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2y/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], k["spmv"]["frac"], "ilu", k["ilu0_apply"]["avg_ms"], "norm", d["config"]["state_norm"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2y/{t}.err").read()[-300:])
PY
}
run d16 JH_X=1



BENCH_ARGS="--cells 1250000" run s_def JH_X=1

