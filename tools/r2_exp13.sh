cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2o
JH_SETUP_TIMING=1 python bench.py --no-cpu --steps 100 --warmup 5 > gpurun_out/r2o/bisect.json 2> gpurun_out/r2o/bisect.err; grep "setup\]" gpurun_out/r2o/bisect.err | head -12
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --steps 100 --warmup 5 $BENCH_ARGS > gpurun_out/r2o/$tag.json 2> gpurun_out/r2o/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2o/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "blocks", d["config"]["ilu_blocks"], "lev", d["config"]["ilu_max_levels"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"], "setup", d["config"]["setup_s"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2o/{t}.err").read()[-800:])
PY
}
run bisect JH_X=1
run onion JH_BLOCK_ORDER=onion
BENCH_ARGS="--law twophase --steps 30" run P_bisect JH_X=1
BENCH_ARGS="--law twophase --steps 30" run P_onion JH_BLOCK_ORDER=onion
BENCH_ARGS="--cells 1250000" run S_bisect JH_X=1
BENCH_ARGS="--cells 1250000" run S_onion JH_BLOCK_ORDER=onion
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2o/pytest.log
