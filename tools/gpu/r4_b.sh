#!/bin/bash
# round 4, session B: record sort -- timelines of the fused pipeline (default; 8-bit digits; no XCD-contiguous tiles) and the
# FETCH_SIZE / WRITE_SIZE passes of the 2e7-mark run
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4b}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_cluster.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -3 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
tl() {  # name, env assignments..., then args of prof_fused
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python3 $R/tools/prof_fused.py "$@" > $O/${T}_$name.log 2>&1
  python3 $R/tools/timeline.py /tmp/prof_$name 'rs_hist<true' > $O/${T}_${name}_timeline.txt 2>&1
  grep "fused ms" $O/${T}_$name.log
  cat $O/${T}_${name}_timeline.txt
}
tl small
tl big big
export DUET_RS_MAXW=8
tl big_w8 big
unset DUET_RS_MAXW
export DUET_RS_NOSWZ=1
tl big_noswz big
unset DUET_RS_NOSWZ
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/prof_fused.py big > $O/${T}_pmc_$c.log 2>&1
done
python3 $R/tools/pmc_table.py $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv') > $O/${T}_fused_2e7_pmc_fetch_write.txt 2>&1
cat $O/${T}_fused_2e7_pmc_fetch_write.txt
