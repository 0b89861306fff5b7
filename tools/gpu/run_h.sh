#!/bin/bash
# GPU box session H: parity of the current build, bench, fused timelines (new schedule and, with dbg=0x800, the round-1 one).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2h}
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 900 python3 tools/stress.py 300 ${STRESS_A0:-2000} > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 300 python3 tools/stamps.py fused > $O/${T}_stamps_fused.log 2>&1
timeout 300 python3 tools/stamps.py cfg2 > $O/${T}_stamps_cfg2.log 2>&1
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/${T}_bench1.json 2> $O/${T}_bench1.err
echo "rc=$?" >> $O/${T}_bench1.err
export TMPDIR=/tmp
cd /tmp
for v in "" "dbg=0x800"; do
  s=${v:+_whole}
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused$s -- python3 $R/tools/prof_fused.py $v > $O/${T}_tl_fused$s.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
cd $R
for d in ${T}_tl_fused ${T}_tl_fused_whole ${T}_tl_fused_big; do
  python3 tools/timeline.py $O/$d cl_keys > $O/$d.timeline.txt 2>&1
  find $O/$d -name '*.csv' -size +2M -delete
done
tail -3 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; grep -h "fused ms" $O/${T}_tl_fused*.log; tail -1 $O/${T}_tl_fused.timeline.txt; tail -1 $O/${T}_tl_fused_whole.timeline.txt; tail -1 $O/${T}_tl_fused_big.timeline.txt; head -c 300 $O/${T}_bench1.json; echo; tail -2 $O/${T}_bench1.err
