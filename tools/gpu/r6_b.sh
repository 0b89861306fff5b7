#!/bin/bash
# Round 6, second GPU call: the two-launch E/F after its first optimisation -- the new parity test, the size sweep, in-kernel stamps of
# both launch structures at config 2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6b}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -m gpu -x -q  > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 900 python3 tools/own_sweep.py 200 > $O/${T}_own_sweep.log 2>&1
echo "rc=$?" >> $O/${T}_own_sweep.log
timeout 300 python3 tools/stamps.py cfg2 > $O/${T}_stamps_two.log 2>&1
timeout 300 python3 tools/stamps.py cfg2 3k > $O/${T}_stamps_three.log 2>&1
timeout 600 python3 tools/stress.py 1200 0 > $O/${T}_stress.log 2>&1; tail -2 $O/${T}_stress.log
tail -3 $O/${T}_tests.log; cat $O/${T}_own_sweep.log; grep -v "contig block" $O/${T}_stamps_two.log | tail -30; grep -A12 "kernel 2" $O/${T}_stamps_three.log
./tools/probe/handoff_probe 391 7 > $O/${T}_handoff.log 2>&1; ./tools/probe/handoff_probe 1024 7 >> $O/${T}_handoff.log 2>&1; cat $O/${T}_handoff.log
