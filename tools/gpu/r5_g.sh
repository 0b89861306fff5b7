#!/bin/bash
# round 5: fork / join events on the producing kernels' own dispatch packets -- parity, then timelines with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5g}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_cluster.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
for W in small big; do
  for D in 0 0x200000; do
    A=""; [ $W = big ] && A="big"
    AN=cl_keys; [ $W = big ] && AN='rs_hist<true'
    rm -rf /tmp/prof_${W}_$D
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${W}_$D -- python3 $R/tools/prof_fused.py $A dbg=$D > $O/${T}_fused_${W}_$D.log 2>&1
    python3 $R/tools/timeline.py /tmp/prof_${W}_$D "$AN" > $O/${T}_fused_${W}_${D}_timeline.txt 2>&1
    grep -v "^W2026\|^E2026" $O/${T}_fused_${W}_$D.log | tail -2
    cat $O/${T}_fused_${W}_${D}_timeline.txt
  done
done
