cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2l
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu --steps 100 --warmup 5 "$@" > gpurun_out/r2l/$tag.json 2> gpurun_out/r2l/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2l/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "blocks", d["config"]["ilu_blocks"], "lev", d["config"]["ilu_max_levels"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2l/{t}.err").read()[-800:])
PY
}
run b512 --block-rows 512
run b576 --block-rows 576
run b608 --block-rows 608
run b640 --block-rows 640
run b704 --block-rows 704
run b800 --block-rows 800
