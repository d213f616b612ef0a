cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2z/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r2z/pytest.log
bash tools/r2_prof_final.sh
