#!/bin/bash
# Round 6, final collection -- ONE gpurun call on ONE box, so that counters, kernel traces and the bench line describe the same build
# on the same device: the whole -m gpu suite, a stress run, bench.py with its defaults, then the rocprofv3
# evidence for profiles/ -- kernel stats and FETCH/WRITE passes per E/F size, kernel stats / timelines / FETCH+WRITE per kernel of the
# fused clustered + phased pipeline --, the E/F launch-structure sweep, the hand-off probe, the end-to-end times at configs[1] and
# configs[2].  Counter passes are separate profiler runs (never --pmc beside a trace); the program itself follows `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6z}
P=$O/${T}_prof
mkdir -p $P
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 1500 python3 tools/stress.py 3000 3000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 1500 python3 bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
echo "rc=$?" >> $O/${T}_bench_default.err
cp $O/bench_detail_n1.json $O/${T}_bench_detail_n1.json
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
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $P/${name}_$ctr -- "$@" > $P/${name}_$ctr.log 2>&1
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
stats fused_2e7 python3 $R/tools/prof_fused.py big
for c in FETCH_SIZE WRITE_SIZE; do
  pmc bench_config2 $c python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
  pmc ef_2e7 $c python3 $R/tools/prof_ef.py 20000000 10
  pmc ef_2e8 $c python3 $R/tools/prof_ef.py 200000000 5
  pmc fused_config2 $c python3 $R/tools/prof_fused.py
  pmc fused_2e7 $c python3 $R/tools/prof_fused.py big
done
for w in config2 2e7; do
  python3 $R/tools/pmc_table.py /tmp/${T}_fused_${w}_FETCH_SIZE.csv /tmp/${T}_fused_${w}_WRITE_SIZE.csv > $P/fused_${w}_pmc_FETCH_WRITE.txt 2>&1
done
( echo "["; python3 $R/tools/fused_traffic.py 1001116 /tmp/${T}_fused_config2_FETCH_SIZE.csv /tmp/${T}_fused_config2_WRITE_SIZE.csv $NOW "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/prof_fused.py"; echo ","; python3 $R/tools/fused_traffic.py 19999876 /tmp/${T}_fused_2e7_FETCH_SIZE.csv /tmp/${T}_fused_2e7_WRITE_SIZE.csv $NOW "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/prof_fused.py big"; echo "]" ) > $P/fused_traffic.json 2> $P/fused_traffic.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused cl_keys > $P/fused_config2_timeline.txt 2>&1      # (below 1.5 M marks: the key-only sort)
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_big 'rs_hist<true' > $P/fused_2e7_timeline.txt 2>&1
echo $NOW > $P/collected_utc.txt
cd $R
timeout 900 python3 tools/own_sweep.py 200 > $P/ef_two_launches_against_three_sweep.txt 2>&1
./tools/probe/handoff_probe 391 7 > $P/handoff_probe.txt 2>&1; ./tools/probe/handoff_probe 1024 7 >> $P/handoff_probe.txt 2>&1
timeout 300 python3 tools/stamps.py cfg2 > $P/stamps_config2_two_launches.txt 2>&1
timeout 300 python3 tools/stamps.py cfg2 3k > $P/stamps_config2_three_launches.txt 2>&1
nproc > $P/e2e_host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $P/e2e_host.txt
DUET_E2E_QUIET=1 timeout 900 python3 tools/e2e_time.py 2e7 4,8,16 > $P/e2e_time_2e7.txt 2>&1
timeout 900 python3 tools/e2e_profile.py 8 2e7 > $P/e2e_profile_2e7_t8.txt 2>&1
DUET_E2E_QUIET=1 timeout 300 python3 tools/e2e_time.py 0 1,2,4,8,16,32 > $P/e2e_time_config2.txt 2>&1
tail -2 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; ls $P | head -60; tail -1 $P/fused_config2_timeline.txt; tail -1 $P/fused_2e7_timeline.txt; grep -v amdgpu $P/ef_two_launches_against_three_sweep.txt; grep -v "duet_ingest\]\|^---\|amdgpu" $P/e2e_time_2e7.txt | tail -6; tail -2 $O/${T}_bench_default.err; head -c 3000 $O/${T}_bench_default.json
