#!/bin/bash
# round 4: the sharded product path without torch in the ranks -- GPU tests of the sharded entries (plumbing mode, one rank over
# RCCL), then the end-to-end times through 1 / 2 / 4 / 8 ranks (tools/e2e_sharded.py), new path and DUET_COMM=torch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4mg}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests/test_gpu_r2.py tests/test_gpu_r3.py -x -q -m gpu -k "rank or shard or gpus or rccl or multi or svim" > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
timeout 900 python3 tools/e2e_sharded.py 4000000 8 > $O/${T}_e2e_4e6.log 2>&1; cat $O/${T}_e2e_4e6.log | tail -5
timeout 1200 python3 tools/e2e_sharded.py 20000000 8 > $O/${T}_e2e_2e7.log 2>&1; cat $O/${T}_e2e_2e7.log | tail -5
DUET_COMM=torch timeout 900 python3 tools/e2e_sharded.py 4000000 8 > $O/${T}_e2e_4e6_torch.log 2>&1; cat $O/${T}_e2e_4e6_torch.log | tail -5
