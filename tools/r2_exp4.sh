cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2f
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --steps 20 --warmup 3 $BENCH_ARGS > gpurun_out/r2f/$tag.json 2> gpurun_out/r2f/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2f/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2f/{t}.err").read()[-1500:])
PY
}
run T128 JH_ILU_FACTOR_THREADS=128
run T256 JH_ILU_FACTOR_THREADS=256
run T512 JH_ILU_FACTOR_THREADS=512
BENCH_ARGS="--law twophase" run P128 JH_ILU_FACTOR_THREADS=128
BENCH_ARGS="--law twophase" run P256 JH_ILU_FACTOR_THREADS=256
BENCH_ARGS="--law twophase" run P512 JH_ILU_FACTOR_THREADS=512
