#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python3 tools/stamps.py cfg2 > $O/${1:-r5sc}_stamps_cfg2.log 2>&1
grep -A12 "kernel 1" $O/${1:-r5sc}_stamps_cfg2.log | grep "contig block"
