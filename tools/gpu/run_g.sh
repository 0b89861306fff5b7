#!/bin/bash
# GPU box session G: parity of the current build, bench, then instruction / cycle counters of the fused pipeline's kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2g}
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 600 python3 tools/stress.py 300 1000 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-large > $O/${T}_bench1.json 2> $O/${T}_bench1.err
echo "rc=$?" >> $O/${T}_bench1.err
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --list-avail > $O/${T}_avail.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/${T}_tl_fused_big.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  for w in small big; do
    arg=""; [ $w = big ] && arg="big"
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/${T}_pmc${i}_$w -- python3 $R/tools/prof_fused.py $arg > $O/${T}_pmc${i}_$w.log 2>&1
  done
done
cd $R
for d in ${T}_tl_fused ${T}_tl_fused_big; do
  python3 tools/timeline.py $O/$d cl_keys > $O/$d.timeline.txt 2>&1
  find $O/$d -name '*.csv' -size +2M -delete
done
for w in small big; do
  python3 tools/pmc_table.py $(find $O/${T}_pmc*_$w -name '*counter_collection.csv') > $O/${T}_pmc_$w.txt 2>&1
done
find $O -path "*${T}_pmc*" -name '*.csv' -size +4M -delete
tail -3 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; tail -1 $O/${T}_tl_fused.timeline.txt; tail -1 $O/${T}_tl_fused_big.timeline.txt; head -c 300 $O/${T}_bench1.json; echo; cat $O/${T}_pmc_small.txt
