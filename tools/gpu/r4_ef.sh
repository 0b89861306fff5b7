#!/bin/bash
# round 4: E/F parity, then per-kernel statistics of the E/F step at 2e7 and 2e8 marks (and config 2 through bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4ef}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -3 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
for sz in 20000000 200000000; do
  rm -rf /tmp/prof_ef_$sz
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ef_$sz -- python3 $R/tools/prof_ef.py $sz 10 > $O/${T}_ef_$sz.log 2>&1
  f=$(find /tmp/prof_ef_$sz -name '*kernel_stats.csv' | head -1)
  cp $f $O/${T}_ef_${sz}_kernel_stats.csv
  grep -E "ef_classify|ef_seed|ef_final" $f | cut -d, -f1-5 | cut -c1-200
  tail -2 $O/${T}_ef_$sz.log
done
