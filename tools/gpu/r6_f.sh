#!/bin/bash
# Round 6: stage A0's forks as signal / gate kernels instead of events -- cluster / fused tests, timelines at both sizes, both ways.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6f}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
export TMPDIR=/tmp
cd /tmp
for v in gate event; do
  arg=""; [ $v = event ] && arg="dbg=0x4000000"
  for w in small big; do
    b=""; [ $w = big ] && b="big"
    python3 $R/tools/prof_fused.py $b $arg > $O/${T}_fused_${v}_$w.log 2>&1
    rm -rf /tmp/${T}_tl_${v}_$w
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_${v}_$w -- python3 $R/tools/prof_fused.py $b $arg > /dev/null 2>&1
    first=cl_keys; [ $w = big ] && first='rs_hist<true'
    python3 $R/tools/timeline.py /tmp/${T}_tl_${v}_$w "$first" > $O/${T}_timeline_${v}_$w.txt 2>&1
  done
done
cd $R
timeout 900 python3 tools/stress.py 0 1500 > $O/${T}_stress.log 2>&1
tail -3 $O/${T}_tests.log; tail -2 $O/${T}_stress.log; grep fused $O/${T}_fused_*.log; tail -14 $O/${T}_timeline_gate_small.txt; tail -1 $O/${T}_timeline_event_small.txt $O/${T}_timeline_gate_big.txt $O/${T}_timeline_event_big.txt
