#!/bin/bash
# round 4: the 1.0 M-mark fused pipeline's timeline with the default (key-only) sort and with the record sort forced
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/${1:-r4z}_prof
mkdir -p $P
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/tools/prof_fused.py > $P/tl_small.log 2>&1
python3 $R/tools/timeline.py /tmp/tl cl_keys > $P/fused_config2_timeline.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl2 -- python3 $R/tools/prof_fused.py dbg=0x40000 > $P/tl_small_rec.log 2>&1
python3 $R/tools/timeline.py /tmp/tl2 'rs_hist<true' > $P/fused_config2_record_sort_timeline.txt 2>&1
cat $P/fused_config2_timeline.txt; tail -1 $P/fused_config2_record_sort_timeline.txt
