#!/bin/bash
# Round 6: small inputs, the partitions of more than 32 marks on four wavefronts (cl_wide_list / cl_wide_big) -- cluster / fused tests,
# timelines at both sizes against DUET_DBG_CLUSTER_WIDE_OFF, the stress run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6g}
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
export TMPDIR=/tmp
cd /tmp
for v in wide narrow; do
  arg=""; [ $v = narrow ] && arg="dbg=0x8000000"
  for w in small big; do
    b=""; [ $w = big ] && b="big"
    python3 $R/tools/prof_fused.py $b $arg > $O/${T}_fused_${v}_$w.log 2>&1
    [ $w = big ] && [ $v = narrow ] && continue
    rm -rf /tmp/${T}_tl_${v}_$w
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_${v}_$w -- python3 $R/tools/prof_fused.py $b $arg > /dev/null 2>&1
    first=cl_keys; [ $w = big ] && first='rs_hist<true'
    python3 $R/tools/timeline.py /tmp/${T}_tl_${v}_$w "$first" > $O/${T}_timeline_${v}_$w.txt 2>&1
  done
done
for m in 250000 2000000 4000000; do
  python3 $R/tools/prof_fused.py marks=$m > $O/${T}_fused_wide_$m.log 2>&1
  python3 $R/tools/prof_fused.py marks=$m dbg=0x8000000 > $O/${T}_fused_narrow_$m.log 2>&1
done
cd $R
timeout 900 python3 tools/stress.py 0 1600 > $O/${T}_stress.log 2>&1
tail -3 $O/${T}_tests.log; tail -2 $O/${T}_stress.log; grep fused $O/${T}_fused_*.log; tail -16 $O/${T}_timeline_wide_small.txt; tail -1 $O/${T}_timeline_narrow_small.txt $O/${T}_timeline_wide_big.txt
