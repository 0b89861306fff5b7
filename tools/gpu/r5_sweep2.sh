#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5sw}
mkdir -p $O
cd $R
timeout 300 python3 tools/sweep_heavy.py big T=24,32,40,48,64,off > $O/${T}_sweep_big.log 2>&1; tail -16 $O/${T}_sweep_big.log
timeout 200 python3 tools/sweep_heavy.py T=24,32,48,64 > $O/${T}_sweep_small.log 2>&1; tail -12 $O/${T}_sweep_small.log
timeout 400 python3 tools/prof_ef.py 200000000 10 > $O/${T}_ef2e8.log 2>&1; tail -2 $O/${T}_ef2e8.log
