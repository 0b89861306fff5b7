#!/bin/bash
# Round 5, third collection (after ef_seed_sort's last change): the whole -m gpu suite, a stress run, bench.py with its defaults, then
# the kernel statistics and timelines that contain ef_seed_sort -- the fused pipeline at both sizes, E/F on stage A0's candidates, the
# bench's config 2.  The FETCH/WRITE and SQ counter tables of the second collection (04:05Z) stay: the kernels they describe did
# not change, ef_seed_sort moves 83 KB per launch.  The program itself follows `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5zz}
P=$O/${T}_prof
mkdir -p $P
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 1500 python3 tools/stress.py 2000 2000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 1200 python3 bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
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
stats bench_config2 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra
stats ef_2e7 python3 $R/tools/prof_ef.py 20000000 20
stats fused_config2 python3 $R/tools/prof_fused.py
stats ef_on_a0_candidates_2e7 python3 $R/tools/prof_ef_on_fused.py big
stats ef_on_a0_candidates_config2 python3 $R/tools/prof_ef_on_fused.py
stats fused_2e7 python3 $R/tools/prof_fused.py big
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_scan -- python3 $R/tools/prof_fused.py big scan > $O/${T}_tl_fused_scan.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${T}_tl_fused_key -- python3 $R/tools/prof_fused.py big dbg=0x10000 > $O/${T}_tl_fused_key.log 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused cl_keys > $P/fused_config2_timeline.txt 2>&1      # (below 1.5 M marks: the key-only sort)
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_big 'rs_hist<true' > $P/fused_2e7_timeline.txt 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_scan 'rs_hist<true' > $P/fused_2e7_scan_order_timeline.txt 2>&1
python3 $R/tools/timeline.py /tmp/${T}_tl_fused_key cl_keys > $P/fused_2e7_key_only_sort_timeline.txt 2>&1
echo $NOW > $P/collected_utc.txt
cd $R
tail -2 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; ls $P | head -40; tail -1 $P/fused_config2_timeline.txt; tail -1 $P/fused_2e7_timeline.txt; tail -1 $P/fused_2e7_scan_order_timeline.txt; tail -1 $P/fused_2e7_key_only_sort_timeline.txt; grep "ef_" $P/fused_config2_timeline.txt $P/fused_2e7_timeline.txt; tail -2 $O/${T}_bench_default.err; head -c 1500 $O/${T}_bench_default.json
