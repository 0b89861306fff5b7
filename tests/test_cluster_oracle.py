# coding=utf-8
"""A0 (span-position clustering): the C statement of this repository's rule, cross-checked against
scipy's average linkage on the agglomeration step.  Parity with the reference is UNPINNED for this stage
(it lives in the external svim binary); see oracle/cluster_oracle.c."""
import numpy as np
import pytest
from scipy.cluster.hierarchy import fcluster, linkage

from duet_amd import synth
from oracle import c_oracle
from tests import helpers as H


def spd(pos, span, i, j, norm=900.0):
    e_i, e_j = pos[i] + span[i], pos[j] + span[j]
    c_i, c_j = pos[i] + span[i] // 2, pos[j] + span[j] // 2
    m = min(abs(pos[i] - pos[j]), abs(e_i - e_j), abs(c_i - c_j))
    smax = max(span[i], span[j])
    return m / norm + (abs(span[i] - span[j]) / smax if smax else 0.0)


def scipy_clusters(pos, span, idx, t):
    n = len(idx)
    if n == 1:
        return [[int(idx[0])]]
    y = [spd(pos, span, idx[a], idx[b]) for a in range(n) for b in range(a + 1, n)]
    lab = fcluster(linkage(np.array(y), method='average'), t, criterion='distance')
    groups = {}
    for k, l in enumerate(lab):
        groups.setdefault(l, []).append(int(idx[k]))
    return sorted(groups.values(), key=lambda g: g[0])


def oracle_groups(res):
    return [[int(x) for x in res['order'][res['cand_off'][i]:res['cand_off'][i + 1]]] for i in range(len(res['cand_off']) - 1)]


@pytest.mark.parametrize('seed', range(6))
def test_against_scipy_average_linkage(seed):
    rng = synth.SplitMix(900 + seed)
    M = 400
    pos = (rng.between(M, 1, 40) * 700 + rng.between(M, -60, 60)).astype(np.int64)     # loose clumps
    span = np.maximum(rng.between(M, 50, 400) + rng.between(M, -20, 20), 1).astype(np.int64)
    contig = rng.below(M, 2)
    mtype = rng.below(M, 2)
    t = [0.3, 0.5, 0.9, 1.4, 0.9, 0.7][seed]
    res = c_oracle.cluster(contig, mtype, pos, span, max_dist=t)
    got = oracle_groups(res)
    assert sorted(x for g in got for x in g) == list(range(M))
    # rebuild the partitions the spec defines, then ask scipy per partition
    centre = pos + span // 2
    order = np.lexsort((np.arange(M), centre, mtype, contig))
    want = []
    start = 0
    for k in range(1, M + 1):
        cut = k == M or contig[order[k]] != contig[order[k - 1]] or mtype[order[k]] != mtype[order[k - 1]] or \
            centre[order[k]] - centre[order[k - 1]] > 1000 or k - start >= 100
        if cut:
            want.extend(scipy_clusters(pos, span, order[start:k], t))
            start = k
    as_sets = lambda gs: sorted(tuple(sorted(g)) for g in gs)
    assert as_sets(got) == as_sets(want)


def test_recovers_generated_candidates():
    """Marks jittered around well-separated candidates cluster back into exactly those candidates."""
    contigs = [synth.bench_contig('21', 600, 300, 5, deg_lo=2, deg_hi=12)]
    for c in contigs:                      # spread candidates out so that neighbours cannot merge
        c.cand_pos = np.arange(len(c.cand_pos), dtype=np.int64) * 5000 + 1000
    marks = synth.raw_marks(contigs, 5, pos_jitter=30, span_jitter_pct=3)
    res = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'], max_dist=0.9)
    truth = marks['truth']
    groups = oracle_groups(res)
    assert len(groups) == len(np.unique(truth))
    for g in groups:
        assert len(set(int(truth[i]) for i in g)) == 1
    # support / mean position / span
    sizes = np.diff(res['cand_off'])
    assert int(sizes.sum()) == len(truth)
    g0 = groups[0]
    assert int(res['cand_pos'][0]) == int(sum(int(marks['pos'][i]) for i in g0) // len(g0))


def test_partition_size_cap_and_determinism():
    M = 350
    pos = np.full(M, 5000, dtype=np.int64) + np.arange(M) % 7
    span = np.full(M, 100, dtype=np.int64)
    z = np.zeros(M, dtype=np.int64)
    a = c_oracle.cluster(z, z, pos, span)
    b = c_oracle.cluster(z, z, pos, span)
    assert np.array_equal(a['order'], b['order']) and np.array_equal(a['cand_off'], b['cand_off'])
    assert len(a['cand_off']) - 1 == 4          # 100 + 100 + 100 + 50 marks, each partition one cluster
    assert list(np.diff(a['cand_off'])) == [100, 100, 100, 50]
