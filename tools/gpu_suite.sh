# the full GPU test suite, the smoke test and the default bench line (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; cd $R; mkdir -p gpurun_out/suite
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/suite/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/suite/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/suite/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/suite/smoke.log
( time python bench.py > gpurun_out/suite/bench_default.json 2> gpurun_out/suite/bench_default.err ) 2>&1 | tail -3; cut -c1-400 gpurun_out/suite/bench_default.json
