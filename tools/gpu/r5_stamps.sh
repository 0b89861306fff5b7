#!/bin/bash
# round 5: in-kernel phase stamps of ef_classify (diagnostic build) at config 2 and 2e7 marks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for w in cfg2 big; do
  timeout 600 python3 tools/stamps.py $w > $O/${1:-r5s}_stamps_$w.log 2>&1
  grep -A12 "kernel 0" $O/${1:-r5s}_stamps_$w.log | head -14
done
