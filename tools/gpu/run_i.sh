#!/bin/bash
# GPU box session I: the rocprofv3 evidence for profiles/ on the final kernels of the round -- kernel stats and FETCH/WRITE counter
# passes per E/F size, fused-pipeline timelines, the fast pass's own statistics (DUET_CL_DEBUG) -- after a parity run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2i}
P=$O/${T}_prof
mkdir -p $P
cd $R
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
DUET_CL_DEBUG=1 timeout 300 python3 tools/prof_fused.py > $O/${T}_cldebug_small.log 2>&1
DUET_CL_DEBUG=1 timeout 300 python3 tools/prof_fused.py big > $O/${T}_cldebug_big.log 2>&1
export TMPDIR=/tmp
cd /tmp
stats() {   # name, command...
  name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$name -- "$@" > $P/$name.log 2>&1
  cp $(find $P/$name -name '*kernel_stats.csv' | head -1) $P/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $P/$name
}
pmc() {     # name, counter, command...
  name=$1; ctr=$2; shift; shift
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $P/${name}_$ctr -- "$@" > $P/${name}_$ctr.log 2>&1
  f=$(find $P/${name}_$ctr -name '*counter_collection.csv' | head -1)
  grep -E "Counter_Name|ef_classify|ef_seed_sort|ef_finalize" $f > $P/${name}_pmc_$ctr.csv 2>/dev/null
  rm -rf $P/${name}_$ctr
}
stats bench_config2 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra
stats ef_2e7 python3 $R/tools/prof_ef.py 20000000 20
stats ef_2e8 python3 $R/tools/prof_ef.py 200000000 20
stats fused_config2 python3 $R/tools/prof_fused.py
stats fused_2e7 python3 $R/tools/prof_fused.py big
for c in FETCH_SIZE WRITE_SIZE; do
  pmc bench_config2 $c python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
  pmc ef_2e7 $c python3 $R/tools/prof_ef.py 20000000 10
  pmc ef_2e8 $c python3 $R/tools/prof_ef.py 200000000 5
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
cd $R
for d in ${T}_tl_fused ${T}_tl_fused_big; do
  python3 tools/timeline.py $O/$d cl_keys > $O/$d.timeline.txt 2>&1
  find $O/$d -name '*.csv' -size +2M -delete
done
timeout 600 python3 tools/e2e_time.py > $O/${T}_e2e.log 2> $O/${T}_e2e.err
tail -2 $O/${T}_tests.log; cat $O/${T}_e2e.log; ls $P | head -40; grep -h "duet_cluster" $O/${T}_cldebug_*.log | sort | uniq | head; tail -1 $O/${T}_tl_fused.timeline.txt; tail -1 $O/${T}_tl_fused_big.timeline.txt
