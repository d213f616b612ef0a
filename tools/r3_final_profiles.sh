# round-3 evidence on the final tree: kernel stats + bench lines (10M default, two-phase, seams, 1.25M, delaunay, polyhedral) and the
# FETCH / WRITE / SQ passes of the default command; then the full GPU suite once more on the same tree
R=$GRAFT_REPO_ROOT; cd $R
bash tools/collect_profiles.sh r03f 2>&1 | tail -16
PMC_PASSES="fetch write sq_time sq_inst tcc" bash tools/pmc_passes.sh r03f 2>&1 | tail -40
bash tools/collect_profiles.sh r03fp --law twophase 2>&1 | tail -14
bash tools/ab.sh r03f_lines "seams10||--path seams --steps 100" "b1M25||--cells 1250000 --steps 100" "b1M||--cells 1000000 --steps 100" "delaunay2M||--mesh delaunay --cells 2000000 --steps 40" "poly2M||--mesh polyhedral --cells 2000000 --steps 40"
ENVV= bash tools/ab.sh r03f_lines "dist_1rank|JH_BENCH_FORCE_DIST=1|--steps 40"
timeout 1200 python -m pytest tests/test_gpu_ipc_allreduce.py tests/test_gpu_distributed.py tests/test_gpu_bench.py -m gpu -q 2>&1 | tail -3
