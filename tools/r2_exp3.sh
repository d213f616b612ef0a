cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2e
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2e/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --steps 30 --warmup 5 $BENCH_ARGS > gpurun_out/r2e/$tag.json 2> gpurun_out/r2e/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2e/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2e/{t}.err").read()[-1500:])
PY
}
run A_old JH_ILU_NO_JAGGED=1
run B_jag JH_X=1
run C_jag_noprog JH_ILU_NO_PROG=1
BENCH_ARGS="--law twophase" run A2_old JH_ILU_NO_JAGGED=1
BENCH_ARGS="--law twophase" run B2_jag JH_X=1
