#!/bin/bash
# round 4: step E/F alone on the fused pipeline's own candidates (type-major / re-sorted by position; warm / cold caches)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4efon}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in "" big; do
  for v in "" sorted cold "cold sorted"; do
    name=efon_${w:-small}_$(echo ${v:-warm} | tr ' ' '_')
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$name -- python3 $R/tools/prof_ef_on_fused.py $w $v > $O/$name.log 2>&1
    cp $(find /tmp/$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
    echo "== $name"; tail -1 $O/$name.log; grep -E "ef_classify|ef_seed_sort|ef_finalize" $O/${name}_kernel_stats.csv | cut -d, -f1-4
  done
done
