#!/bin/bash
# Round 6: why the own-set finalize is slow on stage A0's candidates (type-major order): host-planned on the same candidates, both
# launch structures, both orders; stamps.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6d2}
cd $R
for a in "" "sorted" "dbg=0x800000" "sorted dbg=0x800000"; do
  echo "== $a" >> $O/${T}_efon.log
  python3 tools/prof_ef_on_fused.py steps=200 $a >> $O/${T}_efon.log 2>&1
done
timeout 300 python3 tools/stamps.py fused > $O/${T}_stamps_fused.log 2>&1
grep -v amdgpu $O/${T}_efon.log; grep -A9 "kernel 2 blocks" $O/${T}_stamps_fused.log | head -12
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q > $O/${T}_tests.log 2>&1; tail -2 $O/${T}_tests.log
timeout 600 python3 tools/own_sweep.py 200 > $O/${T}_own_sweep.log 2>&1; grep -v amdgpu $O/${T}_own_sweep.log
python3 tools/prof_fused.py > $O/${T}_fused.log 2>&1; python3 tools/prof_fused.py dbg=0x800000 >> $O/${T}_fused.log 2>&1; grep fused $O/${T}_fused.log
