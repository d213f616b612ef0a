# strong-scaling proxy: the per-GPU share of the 10M-cell case at 8 GPUs (1.25M cells), with and without the synchronous loop
R=$GRAFT_REPO_ROOT; cd $R
for c in 1250000 10000000; do
  python bench.py --cells $c --no-cpu --steps 12 --warmup 3 > gpurun_out/p_${c}.json 2> gpurun_out/p_${c}.err
  python bench.py --option sync_loop=1 --cells $c --no-cpu --steps 12 --warmup 3 > gpurun_out/p_${c}_sync.json 2> gpurun_out/p_${c}_sync.err
done
python - <<'PY'
import json
for f in ["p_1250000","p_1250000_sync","p_10000000","p_10000000_sync"]:
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(f, d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], {a:k[a]["avg_ms"] for a in k}, d["timing"])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-2000:])
PY
