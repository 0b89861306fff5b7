#!/bin/bash
# round 4: the 2e7-mark fused timeline under a list of environment settings (experiment knobs), parity first
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4env}
shift
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -2 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  export $v
  name=big_$(echo $v | tr '=' '_')
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python3 $R/tools/prof_fused.py big > $O/${T}_$name.log 2>&1
  python3 $R/tools/timeline.py /tmp/prof_$name 'rs_hist<true' > $O/${T}_${name}_timeline.txt 2>&1
  echo "== $v $(grep 'fused ms' $O/${T}_$name.log)"
  grep -E "${KERNELS:-part_reduce|span}" $O/${T}_${name}_timeline.txt
  unset ${v%%=*}
done
