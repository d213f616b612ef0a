R=$GRAFT_REPO_ROOT; cd $R
for w in 128 192 256 320 384 512; do
  python bench.py --option spmv_waves_per_xcd=$((4*w)) --no-cpu --steps 8 --warmup 2 > gpurun_out/sw_$w.json 2>/dev/null
  python - $w <<'PY'
import json,sys
w=sys.argv[1]
d=json.loads(open(f"gpurun_out/sw_{w}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("wgs/xcd",w, d["value"], {a:k[a]["avg_ms"] for a in k})
PY
done
