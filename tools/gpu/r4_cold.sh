#!/bin/bash
# round 4: step E/F with cold caches (1 GiB streamed between steps) against the back-to-back loop, config 2 and 2e7 marks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for sz in 1000000 20000000; do
  for c in "" cold; do
    n=efcold_${sz}_${c:-warm}
    rm -rf /tmp/$n
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$n -- python3 $R/tools/prof_ef.py $sz 20 $c > $O/r4_$n.log 2>&1
    cp $(find /tmp/$n -name '*kernel_stats.csv' | head -1) $O/r4_${n}_kernel_stats.csv
    echo "== $n"; grep -E "ef_" $O/r4_${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
  done
done
