#!/bin/bash
# round 5: the partition scan's divisions by part_max through its reciprocal -- A0 / fused parity (32 debug combinations), stress, timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5h}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_cluster.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
timeout 900 python3 tools/stress.py 150 2000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
tail -3 $O/${T}_stress.log
cd /tmp; export TMPDIR=/tmp
for W in small big; do
    A=""; [ $W = big ] && A="big"
    AN=cl_keys; [ $W = big ] && AN='rs_hist<true'
    rm -rf /tmp/prof_${W}
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${W} -- python3 $R/tools/prof_fused.py $A > $O/${T}_fused_${W}.log 2>&1
    python3 $R/tools/timeline.py /tmp/prof_${W} "$AN" > $O/${T}_fused_${W}_timeline.txt 2>&1
    grep -v "^W2026\|^E2026" $O/${T}_fused_${W}.log | tail -2
    cat $O/${T}_fused_${W}_timeline.txt
done
