# kernel trace of a short bench run + per-kernel durations and launch gaps (tools/trace_gaps.py)
# usage (GPU box): bash tools/trace_run.sh <tag> <bench.py args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/raw -o x -- python $R/bench.py --no-cpu --steps 20 --warmup 3 "$@" > $O/bench.json 2> $O/err.txt
F=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $F 100 | tee $O/gaps.txt
rm -rf $O/raw
