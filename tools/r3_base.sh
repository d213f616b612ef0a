# round-3 baseline on the box: 10M bench, 1.25M bench, two-phase bench, proxy
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3base
python bench.py --no-cpu --steps 40 > gpurun_out/r3base/b10.json 2> gpurun_out/r3base/b10.err; cat gpurun_out/r3base/b10.json
python bench.py --no-cpu --steps 40 --cells 1250000 > gpurun_out/r3base/b1.json 2> gpurun_out/r3base/b1.err; cat gpurun_out/r3base/b1.json
python bench.py --no-cpu --steps 40 --law twophase > gpurun_out/r3base/b2.json 2> gpurun_out/r3base/b2.err; cat gpurun_out/r3base/b2.json
PUSH=1 python tools/overlap_probe.py 2>&1 | tail -2
rocm-smi --showcomputepartition 2>&1 | tail -5
amd-smi partition 2>&1 | tail -20
