cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r02f > gpurun_out/prof_f.log 2>&1
PMC_PASSES="fetch write sq_time sq_inst tcc tcp" bash tools/pmc_passes.sh r02f > gpurun_out/pmc_f.log 2>&1
bash tools/collect_profiles.sh r02fp --law twophase > gpurun_out/prof_fp.log 2>&1
PMC_PASSES="fetch write sq_time sq_inst" bash tools/pmc_passes.sh r02fp --law twophase > gpurun_out/pmc_fp.log 2>&1
tail -1 gpurun_out/prof_f.log | cut -c1-200; tail -1 gpurun_out/prof_fp.log | cut -c1-200
