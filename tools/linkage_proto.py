"""Design check (CPU): stage A0 in exact fixed point -- the serial rule of oracle/cluster_oracle.c against the\nround-based evaluation the device uses (all mutual nearest neighbours of a round merged at once, ties by index), partition by\npartition on the test generators' marks.  Prints rounds and work per size class; expects 0 mismatches."""
import sys, ctypes, numpy as np, time
sys.path.insert(0, '/root/repo')
from duet_amd import synth
sys.path.insert(0, '/root/repo/tests')
import subprocess, os
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", "/tmp/liblinkage_proto.so", os.path.join(os.path.dirname(os.path.abspath(__file__)), "linkage_proto.c"), "-lm"])
lib = ctypes.CDLL('/tmp/liblinkage_proto.so')
lib.int_proto.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p]*4 + [ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p]
def run(marks, T=0.9, part_max=100, norm=900.0, gap=1000, name=''):
    st = np.zeros(40, dtype=np.uint64)
    m = {k: np.ascontiguousarray(v) for k, v in marks.items()}
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib.int_proto(len(m['pos']), p(m['contig']), p(m['type']), p(m['pos']), p(m['span']), T, gap, part_max, norm, p(st))
    s = st.reshape(5, 8).astype(np.int64)
    print(name, 'M', len(m['pos']), 'T', T, 'parts', s[:,0].tolist(), 'MISMATCH', s[:,1].tolist(), 'avg rounds', [round(a/max(b,1),2) for a,b in zip(s[:,2], s[:,0])], 'max', s[:,3].tolist(), 'alive/n', [round(a/max(b,1),2) for a,b in zip(s[:,4], s[:,5])])
    return int(s[:,1].sum())
import importlib.util
spec = importlib.util.spec_from_file_location('tgc', '/root/repo/tests/test_gpu_cluster.py')
bad = 0
contigs = [synth.bench_contig('1', 200000, 100000, 1)]
bad += run(synth.raw_marks(contigs, 1), name='config2')
# generators from the GPU test (copied call pattern)
import types
src = open('/root/repo/tests/test_gpu_cluster.py').read()
ns = {}
exec(src.split("@pytest.fixture")[0].replace("pytestmark = pytest.mark.gpu", ""), ns)
exec("def random_marks" + src.split("def random_marks")[1].split("@pytest.mark.parametrize")[0], ns)
for seed in range(8):
    bad += run(ns['sv_like_marks'](seed, 1500), T=[0.9, 0.3, 0.5, 0.7, 1.2, 0.9, 0.15, 0.45][seed], part_max=[100, 100, 128, 100, 60, 100, 100, 100][seed], name='svlike%d' % seed)
for seed in range(6):
    bad += run(ns['random_marks'](seed, 3000 + 700 * seed), T=[0.3, 0.5, 0.9, 1.4, 0.9, 0.05][seed], name='random%d' % seed)
M = 4000
rng = synth.SplitMix(99)
pos = 50000 + rng.between(M, 0, 40) * 2000 + rng.between(M, 0, 5) * 90
span = np.where(rng.chance(M, 1, 3), 300, 200)
marks = dict(contig=np.zeros(M, dtype=np.uint16), type=np.zeros(M, dtype=np.uint8), pos=pos.astype(np.uint32), span=span.astype(np.uint32))
for md in (0.1, 0.2, 90 / 900, 180 / 900, 1 / 3, 1 / 3 + 0.1, 0.0, 1e-300, -1.0, 1e9, float('inf')):
    bad += run(marks, T=md, name='thr')
rng = synth.SplitMix(42); M = 5000
marks = dict(contig=np.zeros(M, dtype=np.uint16), type=np.zeros(M, dtype=np.uint8), pos=(100000 + rng.between(M, 0, 3000)).astype(np.uint32), span=rng.between(M, 100, 140).astype(np.uint32))
bad += run(marks, name='big'); bad += run(marks, T=0.4, part_max=128, name='big'); bad += run(marks, part_max=37, gap=5, name='big')
print('TOTAL MISMATCH', bad)
