#!/bin/bash
# round 5, session A: default bench.py (the compact line), then the quick E/F + fused tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5a}
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "rc=$?" >> $O/${T}_bench.err
wc -c $O/${T}_bench.json; tail -2 $O/${T}_bench.err
cp $O/bench_detail_n1.json $O/${T}_bench_detail.json
head -c 6000 $O/${T}_bench.json
