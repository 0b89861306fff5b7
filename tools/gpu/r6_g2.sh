#!/bin/bash
# quick: cluster tests, config-2 timeline (wide), phase logs, fused timings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6g}
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_cluster.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/prof_fused.py > $O/${T}_fused_wide_small.log 2>&1
python3 $R/tools/prof_fused.py dbg=0x8000000 > $O/${T}_fused_narrow_small.log 2>&1
rm -rf /tmp/${T}_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl -- python3 $R/tools/prof_fused.py > /dev/null 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl cl_keys > $O/${T}_timeline_wide_small.txt 2>&1
for m in 250000 2000000 4000000; do
  python3 $R/tools/prof_fused.py marks=$m > $O/${T}_fused_wide_$m.log 2>&1
  python3 $R/tools/prof_fused.py marks=$m dbg=0x8000000 > $O/${T}_fused_narrow_$m.log 2>&1
done
cd $R
timeout 600 python3 tools/stamps_cl.py > $O/${T}_stamps_cl.log 2>&1
tail -3 $O/${T}_tests.log; grep fused $O/${T}_fused_*.log; tail -14 $O/${T}_timeline_wide_small.txt; grep -B2 -A22 "ends last" $O/${T}_stamps_cl.log | cut -c1-160
