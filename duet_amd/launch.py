# coding=utf-8
"""One process per GPU, started from a parent that never touches a GPU.

`spawn_ranks(n, argv)` starts n fresh interpreters running `argv`, each with the environment
`torch.distributed` reads (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1,
MASTER_PORT = a free port), waits for all of them and returns the first non-zero exit code (0 when every rank
succeeded).  The children are new processes (fork + exec of the interpreter BEFORE anything in them has
initialised HIP), never a re-exec of a process that holds a GPU.  Used by `bench.py --gpus N` when it is started
plainly (not under torch.distributed.run) and by `sv_phasing(..., gpus=N)` / `duet --gpus N`.

Nothing here imports torch or calls HIP.
"""

import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def under_launcher(env=None):
    """True when this process already is one rank of a launched job (torchrun or spawn_ranks)."""
    env = os.environ if env is None else env
    return 'RANK' in env and 'WORLD_SIZE' in env


def rank_env(rank, world, port, base=None, extra=None):
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    # the host driver only supports dmabuf IPC (RCCL / device-tensor sharing across processes need it)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # what torch.distributed.run does for nproc > 1: without it every rank's CPU tensor ops start one OpenMP thread per
    # core (256 here) and the ranks trample each other -- a 1.25 MB host-side gather took 250 ms instead of 0.7 ms
    env.setdefault('OMP_NUM_THREADS', '1')
    if extra:
        env.update({k: str(v) for k, v in extra.items()})
    return env


def assert_gpu_untouched():
    """The ranks are started from a parent that holds no GPU: a process that has initialised HIP and then forks children
    which initialise it again is what this module exists to avoid.  Raises RuntimeError otherwise."""
    torch = sys.modules.get('torch')
    if torch is not None and torch.cuda.is_initialized():
        raise RuntimeError('spawn_ranks: this process has initialised the GPU (torch.cuda); start the ranks from a fresh interpreter')
    lib = sys.modules.get('duet_amd._lib')
    if lib is not None and getattr(lib, 'CONTEXTS_CREATED', 0) > 0:
        raise RuntimeError('spawn_ranks: this process has created a duet context on a GPU; start the ranks from a fresh interpreter')


def probe_devices(n, timeout=300):
    """A fresh interpreter creates a context on each of the first n devices (library present, n gfx950 devices visible):
    what `duet --gpus N` checks before the external stages run, with the parent staying GPU-free.  -> None, or raises
    RuntimeError with the child's message."""
    code = ('import sys\nfrom duet_amd import _lib\nfor i in range(%d):\n    _lib.Context(i).close()\n' % int(n))
    env = dict(os.environ)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env['PYTHONPATH'] = os.pathsep.join([here] + ([env['PYTHONPATH']] if env.get('PYTHONPATH') else []))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        raise RuntimeError('device probe did not finish within %d s' % timeout)
    if p.returncode != 0:
        tail = p.stderr.decode('utf-8', 'replace').strip().splitlines()[-1:]
        raise RuntimeError('device probe failed for %d GPUs: %s' % (n, tail[0] if tail else 'exit code %d' % p.returncode))


def spawn_ranks(n, argv, extra_env=None, timeout=None, quiet_ranks=True):
    """Run `sys.executable argv...` as n ranks.  Rank 0 inherits stdout; the other ranks' stdout goes to stderr when
    quiet_ranks (a rank-0-prints-one-line contract stays intact even if a library chats on another rank).
    -> exit code (0 = every rank exited 0).  On the first failure the remaining ranks get SIGTERM (their exact
    PIDs, never a pattern)."""
    n = int(n)
    if n < 1:
        raise ValueError('need at least one rank')
    assert_gpu_untouched()
    port = free_port()
    procs = []
    for r in range(n):
        out = None if (r == 0 or not quiet_ranks) else sys.stderr
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(r, n, port, extra=extra_env), stdout=out))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                for o in live:
                    procs[o].send_signal(signal.SIGTERM)
        if live and deadline is not None and time.time() > deadline:
            rc = rc or 124
            for o in live:
                procs[o].kill()
            deadline = None
        if live:
            time.sleep(0.02)
    return rc
