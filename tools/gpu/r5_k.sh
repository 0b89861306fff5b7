#!/bin/bash
# round 5: the symmetric passes' rows stored twice -- parity (cluster, fused, stress) and the fused timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5k}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -4 $O/${T}_tests.log
timeout 300 python3 tools/stress.py 100 300 > $O/${T}_stress.log 2>&1; tail -2 $O/${T}_stress.log
cd /tmp; export TMPDIR=/tmp
for W in small big; do
    A=""; [ $W = big ] && A="big"
    AN=cl_keys; [ $W = big ] && AN='rs_hist<true'
    rm -rf /tmp/prof_${W}
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${W} -- python3 $R/tools/prof_fused.py $A > $O/${T}_fused_${W}.log 2>&1
    python3 $R/tools/timeline.py /tmp/prof_${W} "$AN" > $O/${T}_fused_${W}_timeline.txt 2>&1
    grep "cl_tight\|cl_tier2\|cl_link\|cl_fast\|span" $O/${T}_fused_${W}_timeline.txt
done
