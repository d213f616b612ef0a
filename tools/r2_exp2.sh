cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spmv or bicgstab or newton or gmres" > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2d/pytest.log
python tools/spmv_probe.py 2>&1 | grep -v "amdgpu.ids"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu --steps 30 --warmup 5 > gpurun_out/r2d/$tag.json 2> gpurun_out/r2d/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2d/{t}.json").read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "lin its", d["config"]["linear_iterations_per_step"], "spmv", k["spmv"]["avg_ms"], "ilu", k["ilu0_apply"]["avg_ms"], "fac", k["ilu0_factor"]["avg_ms"], "asm", k["assembly"]["avg_ms"], "solve", d["timing"]["linear_solve_ms"])
except Exception as e:
    print(t, "ERR", e); print(open(f"gpurun_out/r2d/{t}.err").read()[-1500:])
PY
}
run A_csr JH_SPMV_NO_JAGGED=1
run B_jag JH_X=1


