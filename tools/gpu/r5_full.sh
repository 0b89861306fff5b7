#!/bin/bash
# round 5: the whole -m gpu suite, then the default bench.py (the line the driver records)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5full}
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=10 > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -16 $O/${T}_tests.log
timeout 1200 python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "rc=$?" >> $O/${T}_bench.err
wc -c $O/${T}_bench.json; tail -2 $O/${T}_bench.err
cp $O/bench_detail_n1.json $O/${T}_bench_detail.json
cat $O/${T}_bench.json
