#!/bin/bash
# round 4, session C: parity (A0 + fused), then the fused pipeline's timelines at both sizes. Extra args: env assignments to try, e.g. DUET_RS_MAXW=10
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r4c}
shift
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_cluster.py tests/test_gpu_fused.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -5 $O/${T}_tests.log
cd /tmp; export TMPDIR=/tmp
tl() {
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python3 $R/tools/prof_fused.py "$@" > $O/${T}_$name.log 2>&1
  python3 $R/tools/timeline.py /tmp/prof_$name 'rs_hist<true' > $O/${T}_${name}_timeline.txt 2>/dev/null || python3 $R/tools/timeline.py /tmp/prof_$name cl_keys > $O/${T}_${name}_timeline.txt 2>&1
  grep "fused ms" $O/${T}_$name.log
  cat $O/${T}_${name}_timeline.txt
}
tl small     # (below 1.5 M marks the key-only sort: its timeline is anchored at cl_keys by tl())
tl big big
for v in "$@"; do
  export $v
  echo "== $v"
  tl big_$(echo $v | tr '=' '_') big
  unset ${v%%=*}
done
