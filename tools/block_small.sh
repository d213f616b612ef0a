R=$GRAFT_REPO_ROOT; cd $R
for b in 160 192 224 256 320; do
  python bench.py --cells 1250000 --no-cpu --steps 20 --warmup 3 --block-rows $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; c=d['config']
print($b, d['value'], c['linear_iterations_per_step'], c['ilu_blocks'], c['ilu_max_levels'], {a:k[a]['avg_ms'] for a in k})"
done
