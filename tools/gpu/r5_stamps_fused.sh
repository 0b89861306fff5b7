#!/bin/bash
# round 5: in-kernel phase stamps of ef_seed_sort on the E/F problem stage A0 hands over (type-major candidate order)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python3 tools/stamps.py fused > $O/${1:-r5sf}_stamps_fused.log 2>&1
grep -A14 "kernel 1" $O/${1:-r5sf}_stamps_fused.log | head -20; tail -1 $O/${1:-r5sf}_stamps_fused.log
