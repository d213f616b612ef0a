# copies the summaries of tools/r2_prof_final.sh (gpurun_out/prof_r02f*, pmc_r02f*) into profiles/r02_final_*
cd "$(dirname "$0")/.."
cp gpurun_out/prof_r02f/kernel_stats.csv profiles/r02_final_kernel_stats_10M_1gpu.csv; cp gpurun_out/prof_r02f/bench.json profiles/r02_final_bench_10M_1gpu.json; cp gpurun_out/prof_r02f/bench_under_rocprof.json profiles/r02_final_bench_under_rocprof_10M_1gpu.json; cp gpurun_out/pmc_r02f/counters.json profiles/r02_final_traffic_10M_1gpu.json; cp gpurun_out/pmc_r02f/counters.txt profiles/r02_final_counters_10M_1gpu.txt
cp gpurun_out/prof_r02fp/kernel_stats.csv profiles/r02_final_twophase_kernel_stats_5M_1gpu.csv; cp gpurun_out/prof_r02fp/bench.json profiles/r02_final_twophase_bench_5M_1gpu.json; cp gpurun_out/pmc_r02fp/counters.json profiles/r02_final_twophase_counters_5M_1gpu.json; cp gpurun_out/pmc_r02fp/counters.txt profiles/r02_final_twophase_counters_5M_1gpu.txt
python - <<'PY'
import json,sys
sys.path.insert(0,'.'); import bench
print("tree", bench.kernel_source_hash(), "profile", json.load(open('profiles/r02_final_traffic_10M_1gpu.json'))['_meta']['kernel_source_hash'])
d=json.loads(open('profiles/r02_final_bench_10M_1gpu.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], d["config"]["setup_s"], {k:(v["avg_ms"],v["frac"]) for k,v in d["roofline"]["kernels"].items()})
PY
