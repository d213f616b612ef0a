# block-Jacobi ILU(0) block size vs Newton rate at the full grid and at its 8-GPU share
R=$GRAFT_REPO_ROOT; cd $R
for c in 1250000 10000000; do for b in 128 256 512 1024; do
  python bench.py --cells $c --block-rows $b --no-cpu --steps 10 --warmup 2 > gpurun_out/b_${c}_${b}.json 2> gpurun_out/b_${c}_${b}.err
done; done
python - <<'PY'
import json
for c in [1250000,10000000]:
  for b in [128,256,512,1024]:
    f=f"b_{c}_{b}"
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(f, d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], d["config"]["ilu_max_levels"], {a:k[a]["avg_ms"] for a in k}, d["timing"]["precond_update_ms"])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-1500:])
PY
