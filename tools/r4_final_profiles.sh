# the round's evidence on the final tree -> gpurun_out/r04/ (copied into profiles/r04_final_* afterwards)
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; cd $R; O=$R/gpurun_out/r04; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_pytest.log | head -1
# (SKIP_PMC=1: second call, after the PMC summary of the first was copied into profiles/ -- bench.py then quotes roofline.traffic)
if [ -z "$SKIP_PMC" ]; then
bash tools/collect_profiles.sh r04_final > $O/collect.log 2>&1
PMC_PASSES="fetch write sq_time tcc" bash tools/pmc_passes.sh r04_final > $O/pmc.log 2>&1
fi
run() { tag=$1; shift; python bench.py --no-cpu "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[2], d["value"], "it/s", d["ms_per_step"], "ms/step, lin its", d["config"]["linear_iterations_per_step"], {a: (k[a]["avg_ms"], k[a]["frac"]) for a in k}, d["timing"].get("us_per_krylov_iteration"), d["config"].get("nonlinear") and {x: d["config"]["nonlinear"][x] for x in ("newton_iterations_per_timestep", "linear_iterations_per_newton_iteration", "ms_per_timestep")})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
run b1M25 --cells 1250000
run b1M25_cr0 --cells 1250000 --option consumer_reduce=0 --option spmv_waves=4 --option spmv_waves_per_xcd=1024
run b1M --cells 1000000
run twophase5M --law twophase
run delaunay2M --mesh delaunay --cells 2000000
run poly2M --mesh polyhedral --cells 2000000
run cartesian10M --mesh cartesian
run seams10 --path seams
run nonlinear10M --law compressible --compressibility 0.5 --newton-tol 1e-7 --timesteps
JH_BENCH_FORCE_DIST=1 python bench.py --no-cpu > $O/bench_dist_1rank.json 2> $O/bench_dist_1rank.err
bash tools/trace_run.sh r04_1M25 --cells 1250000 > $O/trace_1M25.txt 2>&1
bash tools/trace_run.sh r04_10M > $O/trace_10M.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
