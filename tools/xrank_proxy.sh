# One-GPU proxy of the multi-rank Krylov loop: N processes on the one GPU, each context masked to 256/N compute units of its own
# (JH_BENCH_CU_MASK=1), mailboxes + push halo; state halo of the Newton step through the host callback (outside the timed solve).
# Prints us per BiCGStab iteration with the consumer-side all-reduce + folded halo hand-shake (xrank) and with the reduction /
# finish launches (JH_BENCH_NO_XRANK=1), and the single-rank figure of the same box.
# usage: tools/xrank_proxy.sh [ranks=2] [cells=1253160] [out=gpurun_out/xrank]
N=${1:-2}; CELLS=${2:-1253160}; OUT=${3:-gpurun_out/xrank}
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; cd $R; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" JH_BENCH_HALO=host JH_BENCH_CU_MASK=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus $N --steps 40 --warmup 5 --cells $CELLS --no-cpu > $OUT/${name}_n${N}_c${CELLS}.json 2> $OUT/${name}_n${N}_c${CELLS}.err
  echo "$name rc=$?"
}
run xrank JH_DUMMY=1
run noxrank JH_BENCH_NO_XRANK=1
timeout 600 python bench.py --steps 40 --warmup 5 --cells $CELLS --no-cpu > $OUT/single_c${CELLS}.json 2> $OUT/single_c${CELLS}.err
python - $OUT $N $CELLS <<'PY'
import json, sys
out, n, cells = sys.argv[1:4]
for name in (f"xrank_n{n}_c{cells}", f"noxrank_n{n}_c{cells}", f"single_c{cells}"):
    try:
        d = json.loads(open(f"{out}/{name}.json").read().strip().splitlines()[-1])
        c = d["config"]
        print(name, {"newton_it_s": d["value"], "us_per_krylov_iteration": d["timing"]["us_per_krylov_iteration"],
                     "lin_its": c["linear_iterations_per_step"], "path": c.get("krylov_path"), "allreduce": c["scalar_allreduce"],
                     "halo": c["krylov_halo"], "cu_masked": c.get("cu_masked_ranks"), "timeouts": c["comm_timeouts"]})
    except Exception as e:
        print(name, "ERR", e); print(open(f"{out}/{name}.err").read()[-2000:])
PY
