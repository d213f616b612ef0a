# Collects the rocprofv3 evidence for the default bench command (10M cells, 1 GPU): kernel stats + HBM traffic counters.
# Outputs under gpurun_out/prof_$TAG/ ; copy the summaries into profiles/ afterwards.
TAG=${1:-r01_d}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x -- python $R/bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o x -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o x -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/write.err
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summary.py $O/traffic.json FETCH=$F WRITE=$W > $O/traffic.txt 2>&1
rm -rf $O/stats $O/fetch $O/write
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
head -12 $O/kernel_stats.csv; cat $O/traffic.txt; cat $O/bench.json
