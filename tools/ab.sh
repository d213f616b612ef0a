#!/bin/bash
# One parameterised A/B runner (replaces the per-experiment scripts of rounds 1-2): every argument after the output folder is
#   "tag|ENV1=a ENV2=b|bench.py arguments"
# and produces gpurun_out/<dir>/<tag>.json (+ .err) and one summary line.  Example (on the GPU box, through gpurun):
#   bash tools/ab.sh blocks "b384||--block-rows 384" "b512||--block-rows 512" "csr||--option spmv_jagged=0"
#   bash tools/ab.sh small "pend||--cells 1250000" "nopend||--cells 1250000 --option consumer_reduce=0"
# (library switches are context options: bench.py --option key=value, or JH_OPTIONS="key=value,..." in the second field)
# Defaults for every variant: --no-cpu --steps 40 (override inside the third field).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
dir=gpurun_out/$1; shift; mkdir -p "$dir"
for spec in "$@"; do
  IFS='|' read -r tag envs bargs <<< "$spec"
  env $envs timeout 900 python bench.py --no-cpu --steps 40 $bargs > "$dir/$tag.json" 2> "$dir/$tag.err"
  python - "$dir" "$tag" <<'PY'
import json, sys
d_, t = sys.argv[1:3]
try:
    d = json.loads(open(f"{d_}/{t}.json").read().strip().splitlines()[-1])
    c, k = d["config"], d["roofline"]["kernels"]
    print(t, "it/s", d["value"], "ms/step", d["ms_per_step"], "lin its", c["linear_iterations_per_step"], "levels", c["ilu_max_levels"],
          "blocks", c["ilu_blocks"], {a: k[a]["avg_ms"] for a in k}, "solve", d["timing"]["linear_solve_ms"], "us/it", d["timing"].get("us_per_krylov_iteration"), "setup", c["setup_s"],
          c.get("kernels_selected"), flush=True)
except Exception as e:
    print(t, "ERR", e)
    print(open(f"{d_}/{t}.err").read()[-1500:])
PY
done
