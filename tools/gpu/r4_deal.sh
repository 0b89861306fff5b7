#!/bin/bash
# round 4: ef_classify dealing its candidates to the lanes by size -- parity, then E/F alone on the bench's problems (uniform
# sizes) and on stage A0's candidates (sizes with a tail), with the dealing and without (dbg=0x80)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4deal}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_r2.py -x -q -m gpu > $O/tests.log 2>&1
echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
cd /tmp; export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$name -- python3 "$@" > $O/$name.log 2>&1
  cp $(find /tmp/$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
  echo "== $name: $(grep -h 'E/F' $O/$name.log | tail -1)"; grep -E "ef_classify|ef_seed_sort|ef_finalize" $O/${name}_kernel_stats.csv | cut -d, -f1-4
}
for d in ${DBGS:-0}; do
  run ef_1e6_dbg$d $R/tools/prof_ef.py 1000000 200 dbg=$d
  run ef_2e7_dbg$d $R/tools/prof_ef.py 20000000 20 dbg=$d
  run ef_2e8_dbg$d $R/tools/prof_ef.py 200000000 10 dbg=$d
  run efon_small_dbg$d $R/tools/prof_ef_on_fused.py dbg=$d
  run efon_big_dbg$d $R/tools/prof_ef_on_fused.py big dbg=$d
done
