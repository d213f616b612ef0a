cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2ab
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 --steps 40 $BENCH_ARGS > gpurun_out/r2ab/$tag.json 2> gpurun_out/r2ab/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2ab/{t}.json").read().strip().splitlines()[-1])
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], d["config"]["linear_iterations_first_steps"], "lev", d["config"]["ilu_max_levels"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2ab/{t}.err").read()[-300:])
PY
}
for c in 5000000 2500000 20000000; do
for m in 0 40000 300000 100000000; do BENCH_ARGS="--cells $c" run c${c}_m$m JH_PART_TWO_MAX=$m; done; done
for m in 0 40000 300000 100000000; do BENCH_ARGS="--law compressible --cells 5000000" run comp_m$m JH_PART_TWO_MAX=$m; done
