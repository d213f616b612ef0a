cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ipc_allreduce.py -m gpu -x -q > gpurun_out/r2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2m/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --law twophase --steps 30 --warmup 5 > gpurun_out/r2m/$tag.json 2> gpurun_out/r2m/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2m/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], k["spmv"]["frac"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2m/{t}.err").read()[-800:])
PY
}
run A_csr JH_SPMV_NO_JAGGED=1
run B_jag JH_X=1
run A2_csr JH_SPMV_NO_JAGGED=1
run B2_jag JH_X=1
