#!/bin/bash
# does the fused pipeline run under a counter pass (serialised dispatches)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pp -- python3 $R/tools/prof_fused.py > $O/r6_pmcprobe.log 2>&1
echo "rc=$?" >> $O/r6_pmcprobe.log
tail -5 $O/r6_pmcprobe.log
f=$(find /tmp/pp -name '*counter_collection.csv' | head -1); grep -c "cl_gate\|cl_signal" $f; grep -c "cl_box" $f
