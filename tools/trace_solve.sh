# kernel timeline of ONE Newton step (assembly .. update) of a short bench run: tools/trace_solve.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/raw -o x -- python $R/bench.py --no-cpu --steps 20 --warmup 3 "$@" > $O/bench.json 2> $O/err.txt
F=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py $F 12 48 | tee $O/timeline.txt
rm -rf $O/raw
