# A/B of a context option on the bench: tools/opt_sweep.sh <option> "<values>" [bench args...]  -> factor / apply / spmv / assembly times per value
OPT=$1; VALS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; cd $R; O=gpurun_out/opt_sweep; mkdir -p $O
for V in $VALS; do
  timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu --option $OPT=$V "$@" > $O/${OPT}_$V.json 2> $O/${OPT}_$V.err || { echo "$OPT=$V FAILED: $(tail -1 $O/${OPT}_$V.err | cut -c1-200)"; continue; }
  python - $O/${OPT}_$V.json "$OPT=$V" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; k = d["roofline"]["kernels"]
print(f"{sys.argv[2]:>24s}: {d['value']:8.2f} Newton it/s  its/step {c['linear_iterations_per_step']:6.2f}  us/it {d['timing']['us_per_krylov_iteration']:7.2f}  " +
      "  ".join(f"{n} {v['avg_ms']*1e3:7.1f} us ({v['frac']:.2f})" for n, v in k.items()) + f"  factor kernel: {c['kernels_selected']['ilu0_factor']}")
PY
done
