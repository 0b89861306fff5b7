#!/bin/bash
# round 5: the in-library collective -- device-resident rank path, bounded waits -- and the sharded product entries on top of it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5coll}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_r5.py tests/test_gpu_r2.py tests/test_gpu_r3.py -x -q -m gpu --durations=8 -k "not config3_shape" > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -25 $O/${T}_tests.log
