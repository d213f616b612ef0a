# rocprofv3 evidence for the default bench command (10M cells, 1 GPU): kernel stats of a full default run (100 steps), the bench
# line of that profiled process and of an unprofiled run.  Counters (incl. HBM traffic): tools/pmc_passes.sh.
# Outputs under gpurun_out/prof_$TAG/ ; copy the summaries into profiles/ afterwards.
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x -- python $R/bench.py --no-cpu "$@" > $O/bench_under_rocprof.json 2> $O/stats.err
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv
rm -rf $O/stats
cd $R && python bench.py "$@" > $O/bench.json 2> $O/bench.err
head -14 $O/kernel_stats.csv | cut -c1-100,250-360; cat $O/bench.json
