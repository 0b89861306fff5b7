#!/bin/bash
# GPU box session D: everything -- GPU tests, stress, seed-sort stamps of the fused order, timelines, bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2e}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 600 python3 tools/stress.py 500 2000 > $O/${T}_stress.log 2>&1
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
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/${T}_bench1.json 2> $O/${T}_bench1.err
echo "rc=$?" >> $O/${T}_bench1.err
tail -3 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; grep "contig block" $O/${T}_stamps_fused.log $O/${T}_stamps_cfg2.log; tail -1 $O/${T}_tl_fused.timeline.txt; tail -1 $O/${T}_tl_fused_big.timeline.txt; head -c 300 $O/${T}_bench1.json; echo; tail -2 $O/${T}_bench1.err
