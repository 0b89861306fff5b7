#!/bin/bash
# round 3: instruction / cycle counters of the fused pipeline's kernels (separate --pmc passes, no trace beside them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r3pmc}
WHICH=${2:-"small big"}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  for w in $WHICH; do
    arg=""; [ $w = big ] && arg="big"
    rm -rf /tmp/${T}_pmc${i}_$w
    timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/${T}_pmc${i}_$w -- python3 $R/tools/prof_fused.py $arg > $O/${T}_pmc${i}_$w.log 2>&1
  done
done
cd $R
for w in $WHICH; do
  python3 tools/pmc_table.py $(find /tmp/${T}_pmc*_$w -name '*counter_collection.csv') > $O/${T}_pmc_$w.txt 2>&1
  grep -E "^kernel|^cl_|^part_|^rx_|^ef_|^scan" $O/${T}_pmc_$w.txt | cut -c1-420
done
