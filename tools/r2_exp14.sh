cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2p
nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/kernel/mm/transparent_hugepage/enabled
JH_SETUP_TIMING=1 python bench.py --no-cpu --steps 100 --warmup 5 > gpurun_out/r2p/b.json 2> gpurun_out/r2p/b.err; grep "setup\]" gpurun_out/r2p/b.err | head -40
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2p/b.json").read().strip().splitlines()[-1])
print("it/s", d["value"], "its", d["config"]["linear_iterations_per_step"], "setup", d["config"]["setup_s"], d["config"].get("setup_phases_s"))
PY
JH_SETUP_THREADS=16 JH_SETUP_TIMING=1 python bench.py --no-cpu --steps 5 --warmup 1 2>&1 | grep "blocks\|ordering"
JH_SETUP_THREADS=32 JH_SETUP_TIMING=1 python bench.py --no-cpu --steps 5 --warmup 1 2>&1 | grep "blocks\|ordering"
