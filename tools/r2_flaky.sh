cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2fl
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2fl/p$i.log 2>&1; echo "run $i rc=$?"; grep "passed\|failed" gpurun_out/r2fl/p$i.log | tail -1; done
