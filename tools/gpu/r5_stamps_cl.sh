#!/bin/bash
# round 5: the agglomeration chains' phase logs (diagnostic build) at 1.0 M and 2e7 marks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python3 tools/stamps_cl.py > $O/${1:-r5cl}_stamps_cl_small.log 2>&1
cat $O/${1:-r5cl}_stamps_cl_small.log | head -120
timeout 600 python3 tools/stamps_cl.py big > $O/${1:-r5cl}_stamps_cl_big.log 2>&1
grep -A22 "workgroups logged" $O/${1:-r5cl}_stamps_cl_big.log | head -90
