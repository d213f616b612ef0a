# full GPU test suite + smoke + the unstructured families
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3f
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3f/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3f/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3f/smoke.log
bash tools/ab.sh r3f "delaunay2M||--mesh delaunay --cells 2000000 --steps 20" "poly2M||--mesh polyhedral --cells 2000000 --steps 20" "poly2M_b512||--mesh polyhedral --cells 2000000 --steps 20 --block-rows 512"
