cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2j
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu --cells 1250000 --steps 100 --warmup 5 "$@" > gpurun_out/r2j/$tag.json 2> gpurun_out/r2j/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2j/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2j/{t}.err").read()[-800:])
PY
}
run b128 --block-rows 128
run b192 --block-rows 192
run b256 --block-rows 256
run b384 --block-rows 384
run b512 --block-rows 512
