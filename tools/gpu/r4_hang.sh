#!/bin/bash
# diagnostic: bench.py --gpus N in the one-GPU plumbing mode under a watchdog that asks every process of the run's process group
# for its Python stacks (faulthandler on SIGABRT) when the run takes longer than $2 seconds
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONFAULTHANDLER=1 DUET_BENCH_ONE_GPU=1
for n in ${1:-2 4}; do
  S=$(date +%s)
  setsid python3 bench.py --gpus $n --steps 20 --warmup 5 > $O/hang_gpus$n.json 2> $O/hang_gpus$n.err &
  pid=$!
  for i in $(seq 1 ${2:-240}); do
    kill -0 $pid 2>/dev/null || break
    sleep 1
  done
  if kill -0 $pid 2>/dev/null; then
    echo "N=$n still running after ${2:-240} s: stacks"
    ps -o pid,ppid,pgid,stat,etime,cmd -g $(ps -o sid= -p $pid | tr -d ' ') | cut -c1-160
    kill -ABRT -- -$pid
    sleep 3
    kill -KILL -- -$pid 2>/dev/null
  fi
  wait $pid; echo "N=$n rc=$? secs=$(( $(date +%s) - S ))"
  tail -c 300 $O/hang_gpus$n.json; echo
  grep -v "Gloo\|amdgpu.ids\|socket.cpp" $O/hang_gpus$n.err | tail -60
done
