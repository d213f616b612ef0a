R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unstructured.py -x -q -m gpu > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3g/pytest.log
bash tools/ab.sh r3g "poly2M||--mesh polyhedral --cells 2000000 --steps 20" "poly2M_b512||--mesh polyhedral --cells 2000000 --steps 20 --block-rows 512" \
  "f10_pipe||" "f10_pipe_nopfv|JH_ILU_FACTOR_NO_PFV=1|" "f10_nopipe|JH_ILU_FACTOR_PIPE=0|" "f10_pipe2|JH_ILU_FACTOR_PIPE=2|" \
  "tp_pipe||--law twophase" "tp_nopipe|JH_ILU_FACTOR_PIPE=0|--law twophase" "tp_noprepass|JH_ASM_NO_PREPASS=1|--law twophase"
