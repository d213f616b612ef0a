cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 3 4 5 6 7; do JH_ILU_FACTOR_DEBUG=$dbg python tools/factor_probe.py 2>&1 | tail -1; done
for t in 256 1024; do JH_ILU_FACTOR_THREADS=$t python tools/factor_probe.py 2>&1 | tail -1; done
for dbg in 0 1 2 3 4 7; do LAW=twophase JH_ILU_FACTOR_DEBUG=$dbg python tools/factor_probe.py 2>&1 | tail -1; done
for t in 256 1024; do LAW=twophase JH_ILU_FACTOR_THREADS=$t python tools/factor_probe.py 2>&1 | tail -1; done
