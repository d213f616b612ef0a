R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3h
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3h/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3h/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3h/smoke.log
( time python bench.py > gpurun_out/r3h/bench_default.json 2> gpurun_out/r3h/bench_default.err ) 2>&1 | tail -3; cat gpurun_out/r3h/bench_default.json | cut -c1-6000
bash tools/partition_probe.sh
