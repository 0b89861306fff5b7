#!/bin/bash
# GPU box quick session: E/F parity + stress, seed-sort stamps, short bench, fused timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2h}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_rows.py -m gpu -x -q > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
timeout 600 python3 tools/stress.py 3000 200 > $O/${T}_stress.log 2>&1
echo "rc=$?" >> $O/${T}_stress.log
timeout 300 python3 tools/stamps.py fused > $O/${T}_stamps_fused.log 2>&1
timeout 300 python3 tools/stamps.py cfg2 > $O/${T}_stamps_cfg2.log 2>&1
timeout 600 python3 bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $O/${T}_bench_short.json 2> $O/${T}_bench_short.err
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tl_fused -- python3 $R/tools/prof_fused.py > $O/${T}_tl_fused.log 2>&1
cd $R
python3 tools/timeline.py $O/${T}_tl_fused cl_keys > $O/${T}_tl_fused.timeline.txt 2>&1
find $O/${T}_tl_fused -name '*.csv' -size +2M -delete
tail -2 $O/${T}_tests.log; tail -3 $O/${T}_stress.log; grep "contig block" $O/${T}_stamps_fused.log $O/${T}_stamps_cfg2.log; head -c 400 $O/${T}_bench_short.json; echo; tail -5 $O/${T}_tl_fused.timeline.txt
