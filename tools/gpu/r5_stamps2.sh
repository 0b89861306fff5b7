#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python3 tools/stamps.py fused > $O/r5s2_stamps_fused.log 2>&1
grep -A8 "kernel 1" $O/r5s2_stamps_fused.log | head -12
grep -A6 "kernel 2" $O/r5s2_stamps_fused.log | head -8
