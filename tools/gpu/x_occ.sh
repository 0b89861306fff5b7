#!/bin/bash
# experiment: does LDS occupancy bound the agglomeration kernels?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
run() {  # name, args..., env in caller
  name=$1; shift
  rm -rf /tmp/x_$name
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/x_$name -- python3 $R/tools/prof_fused.py "$@" > $O/x_$name.log 2>&1
  python3 $R/tools/timeline.py /tmp/x_$name cl_keys > $O/x_$name.timeline.txt 2>&1
  echo "== $name"; grep -E "cl_fast|cl_box|span" $O/x_$name.timeline.txt
}
run base_big big
DUET_X_SIDEGRID=256 run side256_big big
DUET_X_SIDEGRID=128 run side128_big big
DUET_X_PADLDS=20000 run pad20k_big big
DUET_X_SIDEGRID=256 DUET_X_PADLDS=20000 run side256_pad20k_big big
run base_small
DUET_X_PADLDS=20000 run pad20k_small
DUET_X_SIDEGRID=64 run side64_small
