# copies the summaries of tools/r3_final_profiles.sh (gpurun_out/prof_r03f*, pmc_r03f, r03f_lines) into profiles/r03_final_*
cd "$(dirname "$0")/.."
cp gpurun_out/prof_r03f/kernel_stats.csv profiles/r03_final_kernel_stats_10M_1gpu.csv; cp gpurun_out/prof_r03f/bench.json profiles/r03_final_bench_10M_1gpu.json; cp gpurun_out/prof_r03f/bench_under_rocprof.json profiles/r03_final_bench_under_rocprof_10M_1gpu.json; cp gpurun_out/pmc_r03f/counters.json profiles/r03_final_traffic_10M_1gpu.json; cp gpurun_out/pmc_r03f/counters.txt profiles/r03_final_counters_10M_1gpu.txt
cp gpurun_out/prof_r03fp/kernel_stats.csv profiles/r03_final_twophase_kernel_stats_5M_1gpu.csv; cp gpurun_out/prof_r03fp/bench.json profiles/r03_final_twophase_bench_5M_1gpu.json
for t in seams10 b1M25 b1M delaunay2M poly2M dist_1rank; do cp gpurun_out/r03f_lines/$t.json profiles/r03_final_bench_$t.json; done
python - <<'PY'
import json,sys
sys.path.insert(0,'.'); import bench
print("tree", bench.kernel_source_hash(), "profile", json.load(open('profiles/r03_final_traffic_10M_1gpu.json'))['_meta']['kernel_source_hash'])
d=json.loads(open('profiles/r03_final_bench_10M_1gpu.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["linear_iterations_per_step"], d["config"]["setup_s"], {k:(v["avg_ms"],v["frac"]) for k,v in d["roofline"]["kernels"].items()}, d["roofline"]["traffic"], d["roofline"]["traffic_note"])
PY
