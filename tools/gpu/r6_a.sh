#!/bin/bash
# Round 6, first GPU call: the two-launch E/F (ef_finalize_own) -- parity files that touch E/F, a short stress run, the size sweep
# against the three launches, and the bench's N = 1 line without extras.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6a}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_r5.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 600 python3 tools/stress.py 900 0 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 900 python3 tools/own_sweep.py 200 > $O/${T}_own_sweep.log 2>&1
echo "rc=$?" >> $O/${T}_own_sweep.log
timeout 600 python3 bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "rc=$?" >> $O/${T}_bench.err
tail -5 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; cat $O/${T}_own_sweep.log; tail -3 $O/${T}_bench.err; head -c 2500 $O/${T}_bench.json
