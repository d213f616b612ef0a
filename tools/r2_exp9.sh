cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2k
run() { tag=$1; shift; env $ENVV timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 "$@" > gpurun_out/r2k/$tag.json 2> gpurun_out/r2k/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2k/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "blocks", d["config"]["ilu_blocks"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2k/{t}.err").read()[-800:])
PY
}
run b416 --block-rows 416
run b448 --block-rows 448
run b480 --block-rows 480
run b512 --block-rows 512
run b544 --block-rows 544
run b576 --block-rows 576
ENVV="JH_ILU_FACTOR_THREADS=384" run t384
ENVV="JH_ILU_FACTOR_THREADS=320" run t320
ENVV="JH_ILU_FACTOR_THREADS=640" run t640
