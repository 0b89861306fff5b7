#!/bin/bash
# round 5, on the final kernels: what bounds ef_classify NOW?  The SQ counters of the E/F step at 2e7 marks (bench problem, sizes
# U{2..18}) and on stage A0's candidates (sizes with a tail); every group a profiler run of its own (--pmc alone, no trace beside it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r5pmc2}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_u_$i /tmp/pmc_t_$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_u_$i -- python3 $R/tools/prof_ef.py 20000000 5 > $O/u_$i.log 2>&1
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_t_$i -- python3 $R/tools/prof_ef_on_fused.py big steps=5 > $O/t_$i.log 2>&1
done
cd $R
python3 tools/pmc_table.py $(find /tmp/pmc_u_* -name '*counter_collection.csv') > $O/ef_2e7_uniform_counters.txt 2>&1
python3 tools/pmc_table.py $(find /tmp/pmc_t_* -name '*counter_collection.csv') > $O/ef_2e7_tailed_counters.txt 2>&1
grep -E "^kernel|ef_" $O/ef_2e7_uniform_counters.txt | cut -c1-3000
grep -E "^kernel|ef_" $O/ef_2e7_tailed_counters.txt | cut -c1-3000
