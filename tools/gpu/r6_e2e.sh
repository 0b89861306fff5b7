#!/bin/bash
# Round 6: end to end at BASELINE configs[2]'s size (24 contigs, 2e7 marks as text + BAMs -> phased_sv.vcf), where the time goes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6e2e}
mkdir -p $O
cd $R
export TMPDIR=/tmp
nproc > $O/${T}_host.log; lscpu | grep -E "Model name|^CPU\(s\)" >> $O/${T}_host.log
DUET_E2E_QUIET=1 timeout 1500 python3 tools/e2e_time.py 2e7 4,8,16,32 > $O/${T}_time_2e7.log 2>&1
timeout 900 python3 tools/e2e_profile.py 8 2e7 > $O/${T}_profile_2e7.log 2>&1
timeout 300 python3 tools/e2e_time.py 0 4,8 > $O/${T}_time_cfg2.log 2>&1
cat $O/${T}_host.log; grep -v "duet_ingest\]\|^---" $O/${T}_time_2e7.log | tail -8; tail -25 $O/${T}_profile_2e7.log; grep "threads" $O/${T}_time_cfg2.log | grep -v "^---"
