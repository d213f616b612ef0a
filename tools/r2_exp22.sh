cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2v
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 --steps 40 $BENCH_ARGS > gpurun_out/r2v/$tag.json 2> gpurun_out/r2v/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2v/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], k["spmv"]["frac"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2v/{t}.err").read()[-300:])
PY
}
for w in 256 128 192 320 384 512 1024; do run wgs$w JH_SPMV_WGS=$w; done
