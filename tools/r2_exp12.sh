cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2n
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --steps 30 --warmup 5 $BENCH_ARGS > gpurun_out/r2n/$tag.json 2> gpurun_out/r2n/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2n/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2n/{t}.err").read()[-800:])
PY
}
run A_plain JH_ILU_NT=0
run B_nt JH_ILU_NT=1
run A2_plain JH_ILU_NT=0
run B2_nt JH_ILU_NT=1
BENCH_ARGS="--law twophase" run P_plain JH_ILU_NT=0
BENCH_ARGS="--law twophase" run P_nt JH_ILU_NT=1
