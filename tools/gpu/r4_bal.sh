#!/bin/bash
# round 4: what would a perfectly load-balanced walk buy ef_classify?  The same number of marks with every candidate at exactly
# 10 marks (all lanes of the walk loop 10 times) against the bench's U{2..18}: kernel statistics and SQ counters at 2e7 and 2e8.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4bal}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for sz in 20000000 200000000; do
  for d in "" "deg=10"; do
    n=ef_${sz}_${d:-mixed}
    rm -rf /tmp/$n /tmp/${n}_pmc
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$n -- python3 $R/tools/prof_ef.py $sz 10 $d > $O/${T}_$n.log 2>&1
    cp $(find /tmp/$n -name '*kernel_stats.csv' | head -1) $O/${T}_${n}_kernel_stats.csv
    timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/${n}_pmc -- python3 $R/tools/prof_ef.py $sz 5 $d > $O/${T}_${n}_pmc.log 2>&1
    python3 $R/tools/pmc_table.py $(find /tmp/${n}_pmc -name '*counter_collection.csv') > $O/${T}_${n}_pmc.txt 2>&1
    echo "== $n"; tail -1 $O/${T}_$n.log; grep -E "ef_" $O/${T}_${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160; grep -E "kernel|ef_" $O/${T}_${n}_pmc.txt
  done
done
