# round 3: the seam path (what JutulHIP.jl drives) beside the fused path; new sequence test
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_julia_sequence.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3a/pytest.log
python bench.py --no-cpu --steps 40 --path seams > gpurun_out/r3a/seams10.json 2> gpurun_out/r3a/seams10.err; echo rc=$?; cat gpurun_out/r3a/seams10.json; tail -3 gpurun_out/r3a/seams10.err
python bench.py --no-cpu --steps 40 > gpurun_out/r3a/fused10.json 2> gpurun_out/r3a/fused10.err; echo rc=$?; cat gpurun_out/r3a/fused10.json
python bench.py --no-cpu --steps 40 --path seams --law twophase > gpurun_out/r3a/seams2ph.json 2> gpurun_out/r3a/seams2ph.err; echo rc=$?; cat gpurun_out/r3a/seams2ph.json
