#!/bin/bash
# round 3, session C: bench.py on one GPU (short), bench.py --gpus 2 in the one-GPU plumbing mode (incl. the fused sharded extra)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r3c}
mkdir -p $O
cd $R
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-large > $O/${T}_bench1.json 2> $O/${T}_bench1.err
echo "rc=$?" >> $O/${T}_bench1.err
export DUET_BENCH_ONE_GPU=1
timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 --genome-marks 2000000 > $O/${T}_bench2.json 2> $O/${T}_bench2.err
echo "rc=$?" >> $O/${T}_bench2.err
tail -3 $O/${T}_bench1.err; tail -3 $O/${T}_bench2.err; python3 - <<PY
import json
d = json.loads(open('$O/${T}_bench2.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'parity_vs_oracle')})
print(json.dumps(d.get('extra', {}).get('fused_clustered_and_phased_sharded'))[:900])
d1 = json.loads(open('$O/${T}_bench1.json').read().strip().splitlines()[-1])
print({k: d1[k] for k in ('value', 'n_gpus', 'ms_per_step', 'parity_vs_oracle')})
print(json.dumps(d1['roofline'])[:1400])
print(d1.get('value_clustered_and_phased'), d1.get('ms_per_step_clustered_and_phased'), d1['extra']['fused_clustered_and_phased_2e7_marks']['ms_per_run'])
print(json.dumps(d1['extra']['three_timed_regions_config2'])[:600])
PY
