#!/bin/bash
# round 4, session A: stage A0 with the record-carrying sort: parity tests, then the fused pipeline's timelines (new sort and,
# dbg=0x10000, the key-only sort of round 3) at 1.0 M and 2e7 marks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4a}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -15 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
for W in small big; do
  for D in 0 0x10000; do
    A=""; [ $W = big ] && A="big"
    rm -rf /tmp/prof_${W}_$D
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${W}_$D -- python3 $R/tools/prof_fused.py $A dbg=$D > $O/${T}_fused_${W}_$D.log 2>&1
    python3 $R/tools/timeline.py /tmp/prof_${W}_$D rs_hist > $O/${T}_fused_${W}_${D}_timeline.txt 2>/dev/null || python3 $R/tools/timeline.py /tmp/prof_${W}_$D cl_keys > $O/${T}_fused_${W}_${D}_timeline.txt 2>&1
    tail -1 $O/${T}_fused_${W}_$D.log
    cat $O/${T}_fused_${W}_${D}_timeline.txt
  done
done
