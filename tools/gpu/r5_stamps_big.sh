#!/bin/bash
# round 5: in-kernel phase stamps of the three E/F kernels at 2e7 marks (24 contigs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python3 tools/stamps.py big > $O/${1:-r5sb}_stamps_big.log 2>&1
grep -A40 "kernel 1" $O/${1:-r5sb}_stamps_big.log | head -60
