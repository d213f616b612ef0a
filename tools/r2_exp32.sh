cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2ag
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ilu or twophase or two_phase" > gpurun_out/r2ag/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r2ag/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --warmup 5 --steps 40 $BENCH_ARGS > gpurun_out/r2ag/$tag.json 2> gpurun_out/r2ag/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2ag/{t}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "its", d["config"]["linear_iterations_per_step"], "fac", k["ilu0_factor"]["avg_ms"], k["ilu0_factor"]["frac"], "ilu", k["ilu0_apply"]["avg_ms"], "norm", d["config"]["state_norm"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2ag/{t}.err").read()[-400:])
PY
}
run s_diag JH_X=1
BENCH_ARGS="--law twophase --steps 100" run p_diag JH_X=1
