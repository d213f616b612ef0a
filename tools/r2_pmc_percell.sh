cd $GRAFT_REPO_ROOT
PMC_PASSES="sq_time sq_inst" bash tools/pmc_passes.sh r02pc --law twophase > gpurun_out/pmc_pc.log 2>&1
grep -n "assemble_tile_kernel" -A34 gpurun_out/pmc_r02pc/counters.txt | head -40
