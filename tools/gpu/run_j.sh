#!/bin/bash
# GPU box session J: the N > 1 paths on the one GPU there is (all ranks on device 0, gloo): bench.py --gpus 2 started plainly and
# under torch.distributed.run, the product entry (sv_phasing(..., gpus=2)) against the single-GPU output.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${1:-r2j}
mkdir -p $O
cd $R
export DUET_BENCH_ONE_GPU=1 DUET_ONE_GPU=1
timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 --genome-marks 2000000 > $O/${T}_bench_plain.json 2> $O/${T}_bench_plain.err
echo "rc=$?" >> $O/${T}_bench_plain.err
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --genome-marks 2000000 > $O/${T}_bench_torchrun.json 2> $O/${T}_bench_torchrun.err
echo "rc=$?" >> $O/${T}_bench_torchrun.err
timeout 900 python3 - > $O/${T}_product.log 2>&1 <<'PY'
import os, sys, shutil, tempfile, hashlib
sys.path.insert(0, os.getcwd())
from duet_amd import synth
from duet_amd.sv_phasing import sv_phasing
home = tempfile.mkdtemp(prefix='duet_mg_')
try:
    contigs = synth.bench_genome(400000, 3, labels=['1', '2', '3', '4', '5', 'X'])
    synth.write_workdir(home, contigs, dialect='cutesv', seed=1, write_sam=False)
    sv_phasing(home, 50, 2, 4, False)
    one = open(home + '/phased_sv.vcf', 'rb').read()
    os.remove(home + '/phased_sv.vcf')
    # the sharded entry starts its ranks from a parent that holds no GPU (duet_amd/launch.py): a fresh interpreter
    import subprocess
    subprocess.check_call([sys.executable, '-c', 'from duet_amd.sv_phasing import sv_phasing; sv_phasing(%r, 50, 2, 4, False, gpus=2)' % home])
    two = open(home + '/phased_sv.vcf', 'rb').read()
    print('bytes', len(one), len(two), 'identical', one == two, hashlib.sha256(one).hexdigest()[:16])
finally:
    shutil.rmtree(home, ignore_errors=True)
PY
head -c 600 $O/${T}_bench_plain.json; echo; tail -2 $O/${T}_bench_plain.err; head -c 300 $O/${T}_bench_torchrun.json; echo; tail -2 $O/${T}_bench_torchrun.err; tail -3 $O/${T}_product.log
