cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2ah
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2ah/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed\|Error\|assert" gpurun_out/r2ah/pytest.log | tail -8
