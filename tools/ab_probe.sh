# A/B of an environment switch: bash tools/ab_probe.sh VAR [cells...]
R=$GRAFT_REPO_ROOT; cd $R
V=$1; shift
for c in "$@"; do
  python bench.py --cells $c --no-cpu --steps 10 --warmup 2 > gpurun_out/ab_${c}_on.json 2> gpurun_out/ab_${c}_on.err
  env $V=1 python bench.py --cells $c --no-cpu --steps 10 --warmup 2 > gpurun_out/ab_${c}_off.json 2> gpurun_out/ab_${c}_off.err
done
python - "$@" <<'PY'
import json,sys
for c in sys.argv[1:]:
  for v in ["on","off"]:
    f=f"ab_{c}_{v}"
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(f, d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], {a:k[a]["avg_ms"] for a in k}, d["timing"]["precond_update_ms"])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-1500:])
PY
