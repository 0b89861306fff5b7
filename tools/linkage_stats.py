#!/usr/bin/env python3
"""CPU: what the round-based exact linkage does round by round on the bench's marks (tools/linkage_proto.c's counters) -- how many
clusters would look for a new nearest neighbour if neighbours were kept across rounds, and how many pairs of open rows lie inside one
threshold-graph component (profiles/history/r06_agglomeration_levers_1a_1b_REJECTED.txt)."""
import sys, ctypes, numpy as np, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duet_amd import synth
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", "/tmp/liblinkage_proto.so", os.path.join(os.path.dirname(os.path.abspath(__file__)), "linkage_proto.c"), "-lm"])
lib = ctypes.CDLL('/tmp/liblinkage_proto.so')
lib.int_proto.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p]*4 + [ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p]
nn = (ctypes.c_uint64 * 40).in_dll(lib, 'nn_stats')
cs = (ctypes.c_uint64 * 40).in_dll(lib, 'comp_stats')
def run(marks, name):
    for i in range(40): nn[i] = 0; cs[i] = 0
    st = np.zeros(40, dtype=np.uint64)
    m = {k: np.ascontiguousarray(v) for k, v in marks.items()}
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib.int_proto(len(m['pos']), p(m['contig']), p(m['type']), p(m['pos']), p(m['span']), 0.9, 1000, 100, 900.0, p(st))
    s = st.reshape(5, 8).astype(np.int64)
    a = np.array(list(nn), dtype=np.int64).reshape(5, 8)
    b = np.array(list(cs), dtype=np.int64).reshape(5, 8)
    print('  open partitions per class', b[:,0].tolist(), 'pairs of open rows', b[:,1].tolist(), 'inside one component', b[:,2].tolist(), 'share', [round(x/max(y,1),3) for x,y in zip(b[:,2], b[:,1])])
    print(name, 'marks', len(m['pos']), 'mismatches', s[:, 1].sum())
    for c, lab in enumerate(('<=8', '9..16', '17..32', '33..64', '>64')):
        r, alive, mg, dirty, clean = a[c][:5]
        if r == 0: continue
        full = 12 * alive
        kept = 8 * mg + 90 * dirty
        print('  class %-6s partitions %7d rounds %8d  alive/round %.1f  merges/round %.2f  must-look-again/round %.2f (%.0f %% of alive)  rounds with nobody to look again %.1f %%  | wave instructions: every cluster scans all %d, kept neighbour %d = %.2f x'
              % (lab, s[c][0], r, alive / r, mg / r, dirty / r, 100.0 * dirty / max(alive, 1), 100.0 * clean / r, full, kept, kept / max(full, 1)))
contigs = [synth.bench_contig('1', 200000, 100000, 1)]
run(synth.raw_marks(contigs, 1), 'config2')
g = synth.bench_genome(4000000, 3)
run(synth.raw_marks(g, 1), 'genome 4e6')
