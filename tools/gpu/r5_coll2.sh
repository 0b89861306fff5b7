#!/bin/bash
# round 5: bench.py's N > 1 path through duet_comm_* (one rank over real RCCL), the plumbing mode, and the rest of the r2/r3 tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5coll2}
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_gpu_r2.py tests/test_gpu_r3.py tests/test_gpu_r5.py -x -q -m gpu --durations=8 -k "not config3_shape" > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -25 $O/${T}_tests.log
