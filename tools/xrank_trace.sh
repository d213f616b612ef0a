# kernel trace of the CU-masked N-rank proxy (tools/xrank_proxy.sh): per-kernel durations and launch gaps of every rank
# usage: tools/xrank_trace.sh <tag> [ranks=2] [cells=1253160] [extra env, e.g. JH_BENCH_NO_XRANK=1]
TAG=$1; N=${2:-2}; CELLS=${3:-1253160}; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; O=$R/gpurun_out/xrank_trace_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" JH_BENCH_HALO=host JH_BENCH_CU_MASK=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/raw -o x_%pid% -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29583 $R/bench.py --gpus $N --steps 20 --warmup 3 --cells $CELLS --no-cpu > $O/bench.json 2> $O/err.txt
echo "rc=$?"
for F in $(find $O/raw -name "*kernel_trace.csv" -size +100k); do
  echo "== $F ($(wc -l < $F) rows)"
  python $R/tools/trace_gaps.py $F 100 | tee -a $O/gaps.txt
  python $R/tools/trace_timeline.py $F 10 60 > $O/timeline_$(basename $F .csv).txt
done
rm -rf $O/raw
