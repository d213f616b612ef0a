R=$GRAFT_REPO_ROOT; cd $R
F='grep -v ^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
{
echo "# tools/overlap_probe.py: rank 0's subdomain (1/8 of the 10M-cell grid) on ONE MI355X, 1-rank RCCL communicator, self-exchange halo"
echo "## single process, no communicator, same cell count (bench.py --cells 1250000)"
python bench.py --cells 1250000 --no-cpu --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('ms/step', d['ms_per_step'], 'its/step', c['linear_iterations_per_step'], 'us/iteration', round(d['ms_per_step']*1e3/c['linear_iterations_per_step'],1), 'block_rows', c['block_rows'])"
echo "## RCCL send/recv (fused pack, direct receive)"
python tools/overlap_probe.py 2>&1 | grep owned
echo "## push halo (PUSH=1)"
PUSH=1 python tools/overlap_probe.py 2>&1 | grep owned
echo "## RCCL send/recv, separate pack kernel (option fused_pack=0), ghosts in global-id order (GHOST_ORDER=global: unpack kernel)"
JH_OPTIONS=fused_pack=0 GHOST_ORDER=global python tools/overlap_probe.py 2>&1 | grep owned
echo "## overlapped exchange on a second stream (option halo_overlap=1)"
JH_OPTIONS=halo_overlap=1 python tools/overlap_probe.py 2>&1 | grep owned
} > gpurun_out/dist_proxy.txt 2>&1
cat gpurun_out/dist_proxy.txt
