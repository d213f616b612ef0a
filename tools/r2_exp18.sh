cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2r
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 $BENCH_ARGS > gpurun_out/r2r/$tag.json 2> gpurun_out/r2r/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2r/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "setup", d["config"]["setup_s"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2r/{t}.err").read()[-800:])
PY
}
BENCH_ARGS="--steps 40" run s10 JH_X=1
BENCH_ARGS="--law twophase --steps 30" run p5 JH_X=1
BENCH_ARGS="--cells 1250000 --steps 60" run s1 JH_X=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ilu or factor or simulator" > gpurun_out/r2r/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2r/pytest.log
