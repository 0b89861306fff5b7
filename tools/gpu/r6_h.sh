#!/bin/bash
# Round 6: phase logs of the four-wavefront units (diagnostic build) at config 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6h}
mkdir -p $O
cd $R
timeout 600 python3 tools/stamps_cl.py > $O/${T}_stamps_cl.log 2>&1
cat $O/${T}_stamps_cl.log | cut -c1-200
