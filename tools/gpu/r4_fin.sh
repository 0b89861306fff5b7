#!/bin/bash
# round 4: ef_finalize with 1 / 2 / 4 tiles per workgroup (DUET_FIN_TPB) on the E/F bench problems
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4tpb}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -x -q -m gpu > $O/tests.log 2>&1; tail -1 $O/tests.log
DUET_FIN_TPB=4 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py -x -q -m gpu > $O/tests4.log 2>&1; tail -1 $O/tests4.log
cd /tmp; export TMPDIR=/tmp
for t in 1 2 4; do
  export DUET_FIN_TPB=$t
  for m in 1000000 20000000 200000000; do
    name=ef_${m}_tpb$t
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$name -- python3 $R/tools/prof_ef.py $m 20 > $O/$name.log 2>&1
    cp $(find /tmp/$name -name '*kernel_stats.csv' | head -1) $O/${name}_kernel_stats.csv 2>/dev/null
    echo "== $name: $(grep -h 'E/F' $O/$name.log | tail -1 | cut -c1-110)"; grep -E "ef_finalize" $O/${name}_kernel_stats.csv | cut -d, -f1-4 | cut -c30-
  done
done
