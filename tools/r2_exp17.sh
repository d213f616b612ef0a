cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2q
for n in 8 4; do
timeout 800 python bench.py --gpus $n --steps 10 --warmup 2 --no-cpu > gpurun_out/r2q/g$n.json 2> gpurun_out/r2q/g$n.err; echo "rc=$?"
tail -3 gpurun_out/r2q/g$n.err | cut -c1-300
python - $n <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/r2q/g{n}.json").read().strip().splitlines()[-1]); c=d["config"]
    print(n, "it/s", d["value"], "its", c["linear_iterations_per_step"], "ranks", c["ranks_seen"], "dev", c["devices_used"], c["scalar_allreduce"], c["krylov_halo"], c["halo"], "timeouts", c["comm_timeouts"], "setup", c["setup_s"], c["setup_phases_s"], "norm", c["state_norm"])
except Exception as e: print("ERR", e)
PY
done
