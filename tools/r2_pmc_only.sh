cd $GRAFT_REPO_ROOT
PMC_PASSES="fetch write sq_time sq_inst tcc tcp" bash tools/pmc_passes.sh r02f > gpurun_out/pmc_f.log 2>&1
PMC_PASSES="fetch write sq_time sq_inst" bash tools/pmc_passes.sh r02fp --law twophase > gpurun_out/pmc_fp.log 2>&1
grep -n "ilu_factor_diag" gpurun_out/pmc_r02f/counters.txt | head -3
