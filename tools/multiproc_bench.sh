# Functional run of bench.py's N-rank path at full size with N processes sharing ONE GPU (host-callback ghost exchange):
# checks partitioning / subdomains / mailboxes / iteration counts at scale; the timings mean nothing.
N=${1:-8}; CELLS=${2:-10000000}
R=$GRAFT_REPO_ROOT; cd $R
free -g | head -2
JH_BENCH_HALO=host HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $N --steps 3 --warmup 1 --cells $CELLS --no-cpu > gpurun_out/mp_$N.json 2> gpurun_out/mp_$N.err
echo rc=$?
python - $N <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/mp_{n}.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, d["config"])
except Exception as e:
    print("ERR", e); print(open(f"gpurun_out/mp_{n}.err").read()[-3000:])
PY
