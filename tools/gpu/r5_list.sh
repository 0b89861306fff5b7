#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $O/r5_counters_list.txt 2>&1
wc -l $O/r5_counters_list.txt
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" $O/r5_counters_list.txt | sed 's/Name:[[:space:]]*//' | sort -u | tr '\n' ' ' | head -c 12000
