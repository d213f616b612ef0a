# Counter passes for the default bench command (10M cells, 1 GPU): one rocprofv3 --pmc run per counter group (SQ has 8 slots,
# TCC 4; FETCH_SIZE and WRITE_SIZE cannot share a pass), summarised per kernel by tools/pmc_table.py.
# usage (on the GPU box): bash tools/pmc_passes.sh <tag> [extra bench.py args]
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
declare -A G
G[sq_time]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
G[sq_inst]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
G[sq_act]="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_CYCLES SQ_BUSY_CU_CYCLES"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[tcc2]="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
G[ta]="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
# every pass is summarised (and its raw CSV dropped) as soon as it ends: a call that runs out of time keeps what it has
# (the ta group is refused on gfx950 -- "exceeds the capabilities of the hardware" -- and rocprofv3 then hangs: not in the default list)
PASSES=${PMC_PASSES:-"fetch write sq_time sq_inst tcc tcp sq_act tcc2"}
for g in $PASSES; do
  t0=$(date +%s)
  timeout 300 rocprofv3 --pmc ${G[$g]} --output-format csv -d $O/$g -o x -- python $R/bench.py --no-cpu --steps 2 --warmup 1 "$@" > /dev/null 2> $O/$g.err < /dev/null
  F=$(find $O/$g -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python $R/tools/pmc_table.py $O/pass_$g.json $F > /dev/null 2>> $O/$g.err; else echo "pass $g produced no csv"; tail -3 $O/$g.err; fi
  rm -rf $O/$g
  python $R/tools/pmc_table.py --merge $O/counters.json $O/pass_*.json > $O/counters.txt 2>&1
  echo "pass $g: $(( $(date +%s) - t0 )) s"
done
cat $O/counters.txt
