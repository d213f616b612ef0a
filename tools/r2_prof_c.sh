cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r02c > gpurun_out/prof_c.log 2>&1
PMC_PASSES="fetch write sq_time sq_inst tcc tcp" bash tools/pmc_passes.sh r02c > gpurun_out/pmc_c.log 2>&1
tail -5 gpurun_out/prof_c.log | cut -c1-400
