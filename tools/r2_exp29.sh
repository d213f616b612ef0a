cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2ad
run() { tag=$1; shift; env "$@" timeout 900 python bench.py --no-cpu --warmup 5 --steps 12 $BENCH_ARGS > gpurun_out/r2ad/$tag.json 2> gpurun_out/r2ad/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2ad/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], d["config"]["linear_iterations_first_steps"], "blocks", d["config"]["ilu_blocks"], "lev", d["config"]["ilu_max_levels"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2ad/{t}.err").read()[-300:])
PY
}
for b in 512 2048 4096 13056 39164; do BENCH_ARGS="--block-rows $b" run b$b JH_X=1; done
