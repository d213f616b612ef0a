# round-2 first GPU pass: full GPU test suite, then the default bench (10M cells, 100 steps) and the two-phase bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2a/pytest.log
timeout 900 python bench.py > gpurun_out/r2a/bench_10M.json 2> gpurun_out/r2a/bench_10M.err; echo "bench rc=$?"; cat gpurun_out/r2a/bench_10M.json; tail -5 gpurun_out/r2a/bench_10M.err
timeout 900 python bench.py --law twophase --no-cpu --steps 30 > gpurun_out/r2a/bench_2ph.json 2> gpurun_out/r2a/bench_2ph.err; echo "bench2 rc=$?"; cat gpurun_out/r2a/bench_2ph.json; tail -5 gpurun_out/r2a/bench_2ph.err
