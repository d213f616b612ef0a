# the round's evidence on the final tree -> gpurun_out/r05/ (copied into profiles/r05_final_* afterwards)
# usage: tools/r5_final_profiles.sh [part]   part = tests | setup | pmc | bench | xrank | all (default)
PART=${1:-all}
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
if [ $PART = tests ] || [ $PART = all ]; then
timeout 3000 python -m pytest tests -m gpu -x -q > $O/gpu_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/gpu_pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
fi
if [ $PART = setup ] || [ $PART = all ]; then
# the set-up phases of the default workload as the library prints them (threaded bisection of session 4: never timed on this box)
nproc > $O/setup_phases.txt; grep -m1 "model name" /proc/cpuinfo >> $O/setup_phases.txt; cat /sys/fs/cgroup/cpu.max >> $O/setup_phases.txt 2>/dev/null
python bench.py --no-cpu --steps 10 --warmup 3 --option setup_timing=1 > $O/bench_setup_timing.json 2>> $O/setup_phases.txt; grep -c "jutul_hip setup" $O/setup_phases.txt
JH_SETUP_THREADS=1 python bench.py --no-cpu --steps 10 --warmup 3 --option setup_timing=1 > $O/bench_setup_timing_1thread.json 2> $O/setup_phases_1thread.txt
fi
if [ $PART = pmc ] || [ $PART = all ]; then
bash tools/collect_profiles.sh r05_final > $O/collect.log 2>&1
PMC_PASSES="fetch write sq_time tcc" bash tools/pmc_passes.sh r05_final > $O/pmc_10M.log 2>&1
PMC_PASSES="fetch write sq_time sq_inst tcc" bash tools/pmc_passes.sh r05_final_twophase --law twophase > $O/pmc_twophase.log 2>&1
PMC_PASSES="fetch write sq_time tcc" bash tools/pmc_passes.sh r05_final_1M25 --cells 1253160 > $O/pmc_1M25.log 2>&1
fi
run() { tag=$1; shift; python bench.py --no-cpu "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[2], d["value"], "it/s", d["ms_per_step"], "ms/step, lin its", d["config"]["linear_iterations_per_step"], {a: (k[a]["avg_ms"], k[a]["frac"]) for a in k}, d["timing"].get("us_per_krylov_iteration"), "traffic", d["roofline"]["traffic"], d["config"].get("nonlinear") and {x: d["config"]["nonlinear"][x] for x in ("newton_iterations_per_timestep", "linear_iterations_per_newton_iteration", "ms_per_timestep", "timestep_cuts")})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
if [ $PART = bench ] || [ $PART = all ]; then
run b1M25 --cells 1253160
run b1M --cells 1000000
run twophase5M --law twophase
run delaunay2M --mesh delaunay --cells 2000000
run poly2M --mesh polyhedral --cells 2000000
run cartesian10M --mesh cartesian
run seams10 --path seams
run nonlinear10M --law compressible --compressibility 0.5 --newton-tol 1e-7 --timesteps
run nonlinear10M_hard --law compressible --compressibility 4 --newton-tol 1e-8 --rtol 1e-6 --max-newton 8 --timesteps
run unweighted10M --block-weights none
run unweighted1M25 --cells 1253160 --block-weights none
JH_BENCH_FORCE_DIST=1 python bench.py --no-cpu > $O/bench_dist_1rank.json 2> $O/bench_dist_1rank.err
bash tools/trace_run.sh r05_1M25 --cells 1253160 > $O/trace_1M25.txt 2>&1
bash tools/trace_run.sh r05_10M > $O/trace_10M.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_default_20steps.json 2> $O/bench_default_20steps.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
fi
if [ $PART = xrank ] || [ $PART = all ]; then
tools/xrank_proxy.sh 2 1253160 $O/xrank
tools/xrank_proxy.sh 4 2506320 $O/xrank
tools/xrank_proxy.sh 2 2506320 $O/xrank
JH_BENCH_HALO=host JH_BENCH_CU_MASK=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29585 bench.py --gpus 2 --steps 40 --warmup 5 --cells 2000000 --law twophase --no-cpu > $O/xrank/xrank_twophase_n2.json 2> $O/xrank/xrank_twophase_n2.err; cut -c1-200 $O/xrank/xrank_twophase_n2.json
tools/xrank_trace.sh r05 2 1253160 > $O/xrank_trace.txt 2>&1
fi
