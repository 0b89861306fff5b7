#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5tl}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for H in 0 1; do
  [ $H = 1 ] && export DUET_HACK_NO_BIG=1
  for W in small big; do
    A=""; [ $W = big ] && A="big"
    AN=cl_keys; [ $W = big ] && AN='rs_hist<true'
    rm -rf /tmp/prof_${W}
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${W} -- python3 $R/tools/prof_fused.py $A > $O/${T}_fused_${W}_$H.log 2>&1
    python3 $R/tools/timeline.py /tmp/prof_${W} "$AN" > $O/${T}_fused_${W}_${H}_timeline.txt 2>&1
    echo "== hack=$H $W"; grep -v "^W2026\|^E2026" $O/${T}_fused_${W}_$H.log | tail -1
    cat $O/${T}_fused_${W}_${H}_timeline.txt
  done
done
