#!/bin/bash
# round 3, session A: stage A0 after the exact-linkage rewrite: parity tests, then the fused pipeline's timelines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r3a}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -15 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
for W in small big; do
  A=""; [ $W = big ] && A="big"
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$W -- python3 $R/tools/prof_fused.py $A > $O/${T}_fused_$W.log 2>&1
  python3 $R/tools/timeline.py /tmp/prof_$W > $O/${T}_fused_${W}_timeline.txt 2>&1
  tail -2 $O/${T}_fused_$W.log
  cat $O/${T}_fused_${W}_timeline.txt
done
