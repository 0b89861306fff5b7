#!/bin/bash
# round 5: seed sort's swap pass -- parity, stress, the bench's isolated kernel times and summed-kernel fractions
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r5l}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_r2.py -x -q -m gpu > $O/${T}_tests.log 2>&1
echo "rc=$?" >> $O/${T}_tests.log
tail -4 $O/${T}_tests.log
timeout 300 python3 tools/stress.py 600 100 > $O/${T}_stress.log 2>&1; tail -2 $O/${T}_stress.log
timeout 900 python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open('$O/bench_detail_n1.json'))
print('value', d['value'], 'ms', d['ms_per_step'], d['kernels_us_isolated'])
print(json.dumps(d['summed_kernels_frac_B_EF']))
print(json.dumps(d['extra']['ef_tailed_sizes_2e7'])[:400])
PY
