cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2big
free -g | head -2
for c in 40000000 80000000; do
timeout 1500 python bench.py --no-cpu --cells $c --steps 20 --warmup 5 > gpurun_out/r2big/b$c.json 2> gpurun_out/r2big/b$c.err; echo "rc=$?"
python - $c <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/r2big/b{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; c=d["config"]
    print(c["cells"], "it/s", d["value"], "its", c["linear_iterations_per_step"], "setup", c["setup_s"], c["setup_phases_s"], {n:(v["avg_ms"], v["frac"]) for n,v in k.items()})
except Exception as e: print("ERR", e)
PY
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
