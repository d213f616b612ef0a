# parameter scan for the nonlinear bench line (--timesteps): compressibility, dt, source scale, max Newton iterations, tolerance
cd ${GRAFT_REPO_ROOT:-.}
CELLS=${CELLS:-1000000}
while read -r cfg; do
  [ -z "$cfg" ] && continue
  set -- $cfg
  python bench.py --cells $CELLS --law compressible --compressibility $1 --dt $2 --source $3 --max-newton $4 --newton-tol $5 --rtol ${6:-1e-3} --source-period ${7:-0} --timesteps --steps ${STEPS:-20} --warmup 2 --no-cpu 2>/tmp/nl_err.txt > /tmp/nl_out.json
  python - "$cfg" <<'PY' || { echo "$cfg FAILED: $(grep -v Warning /tmp/nl_err.txt | tail -2 | cut -c1-300)"; }
import json,sys
d=json.loads(open('/tmp/nl_out.json').read()); n=d['config']['nonlinear']
print(sys.argv[1], '->', d['value'], 'Newton it/s', {k:n[k] for k in ('newton_iterations_per_timestep','linear_iterations_per_newton_iteration','ministeps','timestep_cuts','max_newton_iterations_in_a_ministep')})
PY
done
