#!/bin/bash
# Round 6: the fused clustered + phased pipeline with the own-set finalize in its device-planned E/F tail -- the fused / cluster / svim
# tests, a stress run, the config-2 timeline with and without it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6c2}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_r3.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
export TMPDIR=/tmp
cd /tmp
for v in own three; do
  arg=""; [ $v = three ] && arg="dbg=0x800000"
  python3 $R/tools/prof_fused.py $arg > $O/${T}_fused_$v.log 2>&1
  python3 $R/tools/prof_fused.py marks=4000000 $arg >> $O/${T}_fused_$v.log 2>&1
  rm -rf /tmp/${T}_tl_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_$v -- python3 $R/tools/prof_fused.py $arg > /dev/null 2>&1
  python3 $R/tools/timeline.py /tmp/${T}_tl_$v cl_keys > $O/${T}_timeline_$v.txt 2>&1
done
cd $R
tail -3 $O/${T}_tests.log; grep fused $O/${T}_fused_own.log $O/${T}_fused_three.log; tail -6 $O/${T}_timeline_own.txt; tail -5 $O/${T}_timeline_three.txt
