cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_julia_sequence.py -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2b/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --steps 30 --warmup 5 > gpurun_out/r2b/$tag.json 2> gpurun_out/r2b/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2b/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "levels", d["config"]["ilu_max_levels"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "setup", d["config"]["setup_s"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2b/{t}.err").read()[-1500:])
PY
}
run A_csr_bfs JH_SPMV_NO_JAGGED=1 JH_BLOCK_ORDER=bfs
run B_jag_bfs JH_BLOCK_ORDER=bfs
run C_csr_layers JH_SPMV_NO_JAGGED=1
run D_jag_layers JH_X=1
