# round 4: consumer-side second reduction stage (PendSum) A/B at the 8-GPU share (1.25M cells) and at 10M cells
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for c in 1250000 10000000; do
  bash tools/ab.sh pend_$c \
    "old||--cells $c --option consumer_reduce=0 --option spmv_waves_per_xcd=1024" \
    "cr0||--cells $c --option consumer_reduce=0" \
    "pend8||--cells $c" \
    "pend16||--cells $c --option spmv_waves=16" \
    "pend4||--cells $c --option spmv_waves=4"
done
