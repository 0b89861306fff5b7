#!/bin/bash
# GPU box session B: A0 restructure -- parity tests, stress, timelines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2b}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py tests/test_gpu_parity.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 900 python3 tools/stress.py 1500 3000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
cd $R
for d in ${T}_tl_fused ${T}_tl_fused_big; do
  python3 tools/timeline.py $O/$d cl_keys > $O/$d.timeline.txt 2>&1
  python3 tools/timeline.py $O/$d cl_keys --stats > $O/$d.stats.txt 2>&1
  find $O/$d -name '*.csv' -size +2M -delete
done
tail -3 $O/${T}_tests.log; tail -4 $O/${T}_stress.log; grep -h "fused ms" $O/${T}_tl_fused.log $O/${T}_tl_fused_big.log; tail -1 $O/${T}_tl_fused.timeline.txt; tail -1 $O/${T}_tl_fused_big.timeline.txt
