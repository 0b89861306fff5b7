#!/bin/bash
# GPU box session A (round 2): canary, GPU tests, bench N=1 / N=2 (one-GPU plumbing), A0 timelines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== canary: child process after GPU init" > $O/r2a_canary.log
timeout 120 python3 -c "
import torch, subprocess
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
print(subprocess.check_output(['echo', 'child-ok']).decode())
print(subprocess.check_output(['python3', '-c', 'print(42)']).decode())
" >> $O/r2a_canary.log 2>&1
echo "rc=$?" >> $O/r2a_canary.log
lscpu | head -20 > $O/r2a_lscpu.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/r2a_tests.log 2>&1
echo "rc=$?" >> $O/r2a_tests.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/r2a_bench1.json 2> $O/r2a_bench1.err
echo "rc=$?" >> $O/r2a_bench1.err
DUET_BENCH_ONE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/r2a_bench2.json 2> $O/r2a_bench2.err
echo "rc=$?" >> $O/r2a_bench2.err
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r2a_tl_fused -- python3 $R/tools/prof_fused.py > $O/r2a_tl_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r2a_tl_fused_big -- python3 $R/tools/prof_fused.py big > $O/r2a_tl_fused_big.log 2>&1
cd $R
for d in r2a_tl_fused r2a_tl_fused_big; do
  python3 tools/timeline.py $O/$d cl_keys > $O/$d.timeline.txt 2>&1
  python3 tools/timeline.py $O/$d cl_keys --stats > $O/$d.stats.txt 2>&1
  find $O/$d -name '*.csv' -size +2M -delete
done
tail -3 $O/r2a_tests.log; head -c 600 $O/r2a_bench1.json; echo; tail -2 $O/r2a_bench1.err; head -c 400 $O/r2a_bench2.json; echo; tail -3 $O/r2a_bench2.err
