#!/bin/bash
# Round 5, final collection: the rocprofv3 evidence for profiles/ on the round's final kernels -- kernel stats and FETCH/WRITE
# counter passes per E/F size, kernel stats, timelines, FETCH/WRITE per kernel and SQ counter tables of the fused clustered +
# phased pipeline, stage A0's own statistics (DUET_CL_DEBUG), the end-to-end times (single process and sharded) -- after the
# whole -m gpu suite and a stress run.  Counter passes are separate profiler runs (never --pmc beside a trace); the program
# itself follows `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5z}
P=$O/${T}_prof
mkdir -p $P
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 1500 python3 tools/stress.py 500 3000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
DUET_CL_DEBUG=1 timeout 300 python3 tools/prof_fused.py > $O/${T}_cldebug_small.log 2>&1
DUET_CL_DEBUG=1 timeout 300 python3 tools/prof_fused.py big > $O/${T}_cldebug_big.log 2>&1
timeout 1200 python3 bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
echo "rc=$?" >> $O/${T}_bench_default.err
export TMPDIR=/tmp
cd /tmp
NOW=$(date -u +%Y-%m-%dT%H:%MZ)
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
  cp $f /tmp/${T}_${name}_$ctr.csv
  rm -rf $P/${name}_$ctr
}
stats bench_config2 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra
stats ef_2e7 python3 $R/tools/prof_ef.py 20000000 20
stats ef_2e8 python3 $R/tools/prof_ef.py 200000000 20
stats fused_config2 python3 $R/tools/prof_fused.py
stats ef_on_a0_candidates_2e7 python3 $R/tools/prof_ef_on_fused.py big
stats ef_on_a0_candidates_config2 python3 $R/tools/prof_ef_on_fused.py
stats fused_2e7 python3 $R/tools/prof_fused.py big
for c in FETCH_SIZE WRITE_SIZE; do
  pmc bench_config2 $c python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
  pmc ef_2e7 $c python3 $R/tools/prof_ef.py 20000000 10
  pmc ef_2e8 $c python3 $R/tools/prof_ef.py 200000000 5
  pmc fused_config2 $c python3 $R/tools/prof_fused.py
  pmc fused_2e7 $c python3 $R/tools/prof_fused.py big
done
# the fused pipeline's counter traffic, every kernel (tools/fused_traffic.py; pmc_table for the readable table)
for w in config2 2e7; do
  python3 $R/tools/pmc_table.py /tmp/${T}_fused_${w}_FETCH_SIZE.csv /tmp/${T}_fused_${w}_WRITE_SIZE.csv > $P/fused_${w}_pmc_FETCH_WRITE.txt 2>&1
done
M1=$(grep -o "'marks': [0-9]*" $O/${T}_bench_default.json | head -1 | grep -o "[0-9]*")
( echo "["; python3 $R/tools/fused_traffic.py 1001116 /tmp/${T}_fused_config2_FETCH_SIZE.csv /tmp/${T}_fused_config2_WRITE_SIZE.csv $NOW "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/prof_fused.py"; echo ","; python3 $R/tools/fused_traffic.py 19999876 /tmp/${T}_fused_2e7_FETCH_SIZE.csv /tmp/${T}_fused_2e7_WRITE_SIZE.csv $NOW "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/prof_fused.py big"; echo "]" ) > $P/fused_traffic.json 2> $P/fused_traffic.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_scan -- python3 $R/tools/prof_fused.py big scan > $O/${T}_tl_fused_scan.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_key -- python3 $R/tools/prof_fused.py big dbg=0x10000 > $O/${T}_tl_fused_key.log 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused cl_keys > $P/fused_config2_timeline.txt 2>&1      # (below 1.5 M marks: the key-only sort)
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_big 'rs_hist<true' > $P/fused_2e7_timeline.txt 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_scan 'rs_hist<true' > $P/fused_2e7_scan_order_timeline.txt 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_key cl_keys > $P/fused_2e7_key_only_sort_timeline.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  for w in small big; do
    arg=""; [ $w = big ] && arg="big"
    rm -rf /tmp/${T}_pmc${i}_$w
    timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/${T}_pmc${i}_$w -- python3 $R/tools/prof_fused.py $arg > $O/${T}_pmc${i}_$w.log 2>&1
  done
done
cd $R
python3 tools/pmc_table.py $(find /tmp/${T}_pmc*_small -name '*counter_collection.csv') > $P/fused_config2_pmc_counters.txt 2>&1
python3 tools/pmc_table.py $(find /tmp/${T}_pmc*_big -name '*counter_collection.csv') > $P/fused_2e7_pmc_counters.txt 2>&1
timeout 600 python3 tools/sweep_heavy.py T=off,16,24,32,48 > $P/ef_classify_threshold_sweep_config2.txt 2>&1
timeout 900 python3 tools/sweep_heavy.py big T=off,16,24,32,48 > $P/ef_classify_threshold_sweep_2e7.txt 2>&1
timeout 600 python3 tools/e2e_time.py > $O/${T}_e2e.log 2> $O/${T}_e2e.err
timeout 900 python3 tools/e2e_sharded.py 4000000 8 > $O/${T}_e2e_sharded_4e6.log 2>&1
timeout 1200 python3 tools/e2e_sharded.py 20000000 8 > $O/${T}_e2e_sharded_2e7.log 2>&1
echo $NOW > $P/collected_utc.txt
tail -2 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; cat $O/${T}_e2e.log; cat $O/${T}_e2e_sharded_4e6.log $O/${T}_e2e_sharded_2e7.log; ls $P | head -60; grep -h "duet_cluster" $O/${T}_cldebug_*.log | sort | uniq | head; tail -1 $P/fused_config2_timeline.txt; tail -1 $P/fused_2e7_timeline.txt; tail -1 $P/fused_2e7_scan_order_timeline.txt; tail -1 $P/fused_2e7_key_only_sort_timeline.txt; tail -2 $O/${T}_bench_default.err; head -c 1500 $O/${T}_bench_default.json
