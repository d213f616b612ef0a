# Round 6: verify HEAD end to end on the GPU and refresh the evidence the driver line quotes.
# usage (on the GPU box): bash tools/first_gpu_call.sh [tests|lean|pmc|setup|bench|all]     outputs -> gpurun_out/r06/
# Every output carries the device-code hash of the library it ran (tools/fatbin_hash.py) in its name; stops at the first failing part.
PART=${1:-all}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/r06; mkdir -p $O
H=$(python tools/fatbin_hash.py | cut -d' ' -f1); echo "device code $H  head $(cat .git_head 2>/dev/null)"
fail() { echo "FAILED: $1"; exit 1; }
if [ $PART = tests ] || [ $PART = all ]; then
  # (1) every test that has run on a GPU before (tests/test_gpu_zz_round6.py sorts last and holds the ones that have not)
  timeout 2400 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_zz_round6.py > $O/gpu_pytest_$H.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|error" $O/gpu_pytest_$H.log | tail -2
  [ $rc = 0 ] || { tail -40 $O/gpu_pytest_$H.log; fail "gpu suite"; }
  # (2) the GPU tests written without a GPU: all of them, whatever the first one says
  timeout 1500 python -m pytest tests/test_gpu_zz_round6.py -m gpu -q > $O/gpu_pytest_round6_new_$H.log 2>&1; rc6=$?; echo "round-6 tests rc=$rc6"; tail -25 $O/gpu_pytest_round6_new_$H.log | cut -c1-300
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$H.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 $O/smoke_$H.log
  [ $rc = 0 ] || fail smoke
fi
if [ $PART = lean ] || [ $PART = all ]; then
  JH_TEST_LEAN=1 timeout 900 python -m pytest tests/test_gpu_lean_upload.py -m gpu -x -q > $O/gpu_pytest_lean_cases_$H.log 2>&1; rc=$?; echo "lean cases rc=$rc"; tail -1 $O/gpu_pytest_lean_cases_$H.log
  [ $rc = 0 ] || { tail -40 $O/gpu_pytest_lean_cases_$H.log; fail "lean cases"; }
  JH_OPTIONS=ilu_lean_upload=1 timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_pytest_lean_option_$H.log 2>&1; rc=$?; echo "suite with ilu_lean_upload=1 rc=$rc"; tail -1 $O/gpu_pytest_lean_option_$H.log
  [ $rc = 0 ] || { tail -40 $O/gpu_pytest_lean_option_$H.log; fail "suite under ilu_lean_upload=1"; }
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  bash tools/collect_profiles.sh r06 > $O/collect.log 2>&1; cp gpurun_out/prof_r06/kernel_stats.csv $O/kernel_stats_10M_1gpu_$H.csv
  PMC_PASSES="fetch write sq_time tcc" bash tools/pmc_passes.sh r06 > $O/pmc_10M.log 2>&1
  PMC_PASSES="fetch write sq_time tcc" bash tools/pmc_passes.sh r06_twophase --law twophase > $O/pmc_twophase.log 2>&1
  PMC_PASSES="fetch write sq_time tcc" bash tools/pmc_passes.sh r06_1M25 --cells 1253160 > $O/pmc_1M25.log 2>&1
  cp gpurun_out/pmc_r06/counters.json $O/traffic_10M_1gpu.json
  cp gpurun_out/pmc_r06_twophase/counters.json $O/traffic_twophase_lattice_4983504_1gpu.json
  cp gpurun_out/pmc_r06_1M25/counters.json $O/traffic_poisson_lattice_1253160_1gpu.json
  python - $O/traffic_*.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    d = json.load(open(p)); print(p.split("/")[-1], d["_meta"].get("device_code_hash"), len(d) - 1, "kernels")
PY
fi
setup_run() { tag=$1; thr=$2; shift 2; JH_SETUP_THREADS=$thr python bench.py --no-cpu --steps 10 --warmup 3 --option setup_timing=1 "$@" > $O/setup_$tag.json 2> $O/setup_$tag.txt
  python - $O/setup_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], {k: v for k, v in d.get("timing", {}).items() if "setup" in k or "mesh" in k})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
if [ $PART = setup ] || [ $PART = all ]; then
  ( nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max ) > $O/setup_host.txt 2>/dev/null
  setup_run t16_heap1 16 --option setup_heap=1
  setup_run t16_heap0 16
  setup_run t1_heap1 1 --option setup_heap=1
  setup_run t1_heap0 1
fi
run() { tag=$1; shift; python bench.py --no-cpu "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[2], d["value"], "it/s", d["ms_per_step"], "ms/step, lin its", d["config"]["linear_iterations_per_step"], {a: (k[a]["avg_ms"], k[a]["frac"]) for a in k}, d["timing"].get("us_per_krylov_iteration"), "traffic", d["roofline"]["traffic"], "setup", d["timing"].get("setup_s"))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
if [ $PART = bench ] || [ $PART = all ]; then
  run b1M25 --cells 1253160 --steps 100
  run twophase5M --law twophase --steps 100
  run delaunay2M --mesh delaunay --cells 2000000
  run poly2M --mesh polyhedral --cells 2000000
  run seams10 --path seams
  python bench.py > $O/bench_default_$H.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default_$H.json
fi
