cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PUSH=1 STEPS=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_push -o p -- python $R/tools/overlap_probe.py > $R/gpurun_out/push.log 2>&1
S=$(find $R/gpurun_out/prof_push -name "*kernel_stats.csv" | head -1); head -14 $S | cut -c1-160
