#!/bin/bash
# round 5: cl_fast_all's grid (virtual blocks per workgroup) swept through an environment knob (experiment build), fused timelines at 1.0 M marks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5m}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for D in 256 128 64 32; do
    export DUET_GRIDW_DIV=$D
    rm -rf /tmp/prof_g$D
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g$D -- python3 $R/tools/prof_fused.py > $O/${T}_fused_g$D.log 2>&1
    python3 $R/tools/timeline.py /tmp/prof_g$D cl_keys > $O/${T}_fused_g${D}_timeline.txt 2>&1
    echo "gridw = M / $D:"; grep "cl_tight_big\|cl_fast\|span" $O/${T}_fused_g${D}_timeline.txt
done
