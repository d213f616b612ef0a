cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2u
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2u/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2u/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2u/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2u/smoke.log
