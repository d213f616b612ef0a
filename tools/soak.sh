R=$GRAFT_REPO_ROOT; cd $R
python bench.py --no-cpu --steps 300 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single 300 steps', d['value'], d['config']['linear_iterations_per_step'])"
JH_BENCH_HALO=host HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 8 --steps 60 --warmup 2 --cells 10000000 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8 procs 60 steps', d['value'], d['config']['linear_iterations_per_step'], d['config']['krylov_halo'], d['config']['scalar_allreduce'])"
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
