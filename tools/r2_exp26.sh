cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2aa
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 --steps 60 $BENCH_ARGS > gpurun_out/r2aa/$tag.json 2> gpurun_out/r2aa/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2aa/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "lev", d["config"]["ilu_max_levels"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "setup", d["config"]["setup_s"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2aa/{t}.err").read()[-300:])
PY
}
run t300k JH_SETUP_TIMING=1
grep "blocks:" gpurun_out/r2aa/t300k.err
run t40k JH_SETUP_TIMING=1 JH_PART_TWO_MAX=40000
grep "blocks:" gpurun_out/r2aa/t40k.err
run tall JH_PART_TWO_MAX=100000000
BENCH_ARGS="--law twophase --steps 100" run p_two JH_X=1
BENCH_ARGS="--law twophase --steps 100" run p_one JH_PART_TWO_MAX=0
