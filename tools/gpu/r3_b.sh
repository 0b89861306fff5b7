#!/bin/bash
# round 3, session B: the whole -m gpu suite (incl. the new sharded / pinned tests)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r3b}
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -40 $O/${T}_tests.log
