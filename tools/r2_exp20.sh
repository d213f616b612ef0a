cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2s
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 --steps 40 $BENCH_ARGS > gpurun_out/r2s/$tag.json 2> gpurun_out/r2s/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2s/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "lev", d["config"]["ilu_max_levels"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2s/{t}.err").read()[-300:])
PY
}
for m in center rim rev nat; do run s_$m JH_BLOCK_INNER=$m; done
for m in center rim rev; do BENCH_ARGS="--law twophase --steps 25" run p_$m JH_BLOCK_INNER=$m; done
