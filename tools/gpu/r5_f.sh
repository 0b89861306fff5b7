#!/bin/bash
# round 5: E/F inside the fused pipeline -- seed lists of a few long runs merged by rank, device-planned ef_finalize with two tiles per
# workgroup -- parity, then timelines of the fused pipeline at both sizes and E/F alone on stage A0's candidates
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5f}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_r5.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
timeout 600 python3 tools/sweep_heavy.py T=32 > $O/${T}_sweep_small.log 2>&1; tail -6 $O/${T}_sweep_small.log
timeout 900 python3 tools/sweep_heavy.py big T=32 > $O/${T}_sweep_big.log 2>&1; tail -6 $O/${T}_sweep_big.log
cd /tmp; export TMPDIR=/tmp
for W in small big; do
  A=""; [ $W = big ] && A="big"
  rm -rf /tmp/prof_$W
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$W -- python3 $R/tools/prof_fused.py $A > $O/${T}_fused_$W.log 2>&1
  python3 $R/tools/timeline.py /tmp/prof_$W > $O/${T}_fused_${W}_timeline.txt 2>&1
  tail -2 $O/${T}_fused_$W.log
  cat $O/${T}_fused_${W}_timeline.txt
done
