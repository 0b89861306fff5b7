#!/bin/bash
# round 5, session C: ef_classify with the lean walk -- parity, then the threshold sweep again
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5c}
mkdir -p $O
cd $R
timeout 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
timeout 200 python3 tools/sweep_heavy.py T=32 dbg=0,0x200000 > $O/${T}_sweep_small.log 2>&1; cat $O/${T}_sweep_small.log | tail -20
timeout 300 python3 tools/sweep_heavy.py big T=32 dbg=0,0x200000 > $O/${T}_sweep_big.log 2>&1; cat $O/${T}_sweep_big.log | tail -20
