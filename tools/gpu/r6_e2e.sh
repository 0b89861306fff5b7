#!/bin/bash
# Round 6: end to end at BASELINE configs[2]'s size (24 contigs, 2e7 marks as text + BAMs -> phased_sv.vcf), where the time goes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6e2e}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_r2.py tests/test_gpu_r3.py tests/test_gpu_rows.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
DUET_E2E_QUIET=1 timeout 1500 python3 tools/e2e_time.py 2e7 4,8,16,32 > $O/${T}_time_2e7.log 2>&1
timeout 900 python3 tools/e2e_profile.py 8 2e7 > $O/${T}_profile_2e7.log 2>&1
DUET_E2E_QUIET=1 timeout 300 python3 tools/e2e_time.py 0 1,2,4,8,16 > $O/${T}_time_cfg2.log 2>&1
tail -2 $O/${T}_tests.log; grep -v "duet_ingest\]\|^---" $O/${T}_time_2e7.log | tail -6; grep -v "bam \|amdgpu" $O/${T}_profile_2e7.log | tail -22; grep "threads" $O/${T}_time_cfg2.log | grep -v "^---"
