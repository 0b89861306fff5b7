#!/bin/bash
# record sort against key sort below 1.5 M marks, on the round's final build (24-contig genome; each size three times)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r6i}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for m in 900000 1100000 1200000 1300000 1500000; do
  for i in 1 2 3; do
    python3 $R/tools/prof_fused.py marks=$m dbg=0x10000 > $O/${T}_key_${m}_$i.log 2>&1
    python3 $R/tools/prof_fused.py marks=$m dbg=0x40000 > $O/${T}_rec_${m}_$i.log 2>&1
  done
done
grep fused $O/${T}_*.log | sed 's/.*\///' | sort
