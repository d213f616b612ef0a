# block-size sweep: BiCGStab iterations per step x us per iteration x Newton it/s
# usage: tools/block_sweep.sh <cells> "<block rows list>" [extra bench args]
CELLS=$1; LIST=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; cd $R; O=gpurun_out/block_sweep; mkdir -p $O
for B in $LIST; do
  timeout 900 python bench.py --cells $CELLS --block-rows $B --steps 30 --warmup 3 --no-cpu "$@" > $O/c${CELLS}_b${B}.json 2> $O/c${CELLS}_b${B}.err || { echo "cells $CELLS block_rows $B FAILED: $(tail -1 $O/c${CELLS}_b${B}.err | cut -c1-200)"; continue; }
  python - $O/c${CELLS}_b${B}.json $B <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; k = d["roofline"]["kernels"]
print(f"cells {c['cells']} block_rows {sys.argv[2]:>5s}: {d['value']:8.2f} Newton it/s  its/step {c['linear_iterations_per_step']:6.2f}  us/it {d['timing']['us_per_krylov_iteration']:7.2f}  "
      f"blocks {c['ilu_blocks']} max_rows {c['ilu_max_block_rows']} levels {c['ilu_max_levels']} kept {c['ilu_kept_fraction']}  spmv {k['spmv']['avg_ms']*1e3:6.1f} apply {k['ilu0_apply']['avg_ms']*1e3:6.1f} factor {k['ilu0_factor']['avg_ms']*1e3:6.1f} us  setup {c['setup_s']} s")
PY
done
