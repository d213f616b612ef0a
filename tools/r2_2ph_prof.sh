# two-phase (configs[3], 5M cells, 2x2 blocks) evidence: kernel stats + SQ / traffic counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02_2ph; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x -- python $R/bench.py --no-cpu --law twophase --steps 30 > $O/bench_under_rocprof.json 2> $O/stats.err
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv; rm -rf $O/stats
cd $R && python bench.py --law twophase --no-cpu --steps 30 > $O/bench.json 2> $O/bench.err
PMC_PASSES="fetch write sq_time sq_inst sq_act" bash tools/pmc_passes.sh r02_2ph --law twophase
head -8 $O/kernel_stats.csv | cut -c1-90,250-330; cat $O/bench.json | cut -c1-400
