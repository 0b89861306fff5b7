#!/bin/bash
# GPU box session E: A0 parity + stress, seed-sort stamps, timelines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2f}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 600 python3 tools/stress.py 200 3000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 300 python3 tools/stamps.py fused > $O/${T}_stamps_fused.log 2>&1
timeout 300 python3 tools/stamps.py cfg2 > $O/${T}_stamps_cfg2.log 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
cd $R
for d in ${T}_tl_fused ${T}_tl_fused_big; do
  python3 tools/timeline.py $O/$d cl_keys > $O/$d.timeline.txt 2>&1
  find $O/$d -name '*.csv' -size +2M -delete
done
tail -2 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; grep "contig block" $O/${T}_stamps_fused.log $O/${T}_stamps_cfg2.log; grep -h "fused ms" $O/${T}_tl_fused.log $O/${T}_tl_fused_big.log; tail -1 $O/${T}_tl_fused.timeline.txt; tail -1 $O/${T}_tl_fused_big.timeline.txt
