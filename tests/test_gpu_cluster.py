# coding=utf-8
"""-m gpu: stage A0 (span-position clustering) through the C ABI against oracle/cluster_oracle.c, field
for field (order, CSR offsets, candidate contig/type/pos/span)."""
import numpy as np
import pytest

from duet_amd import _lib, synth
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

FIELDS = ('order', 'cand_off', 'cand_contig', 'cand_type', 'cand_pos', 'cand_span')


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def check(ctx, marks, **kw):
    want = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'], **kw)
    # hints on/off (sort key width), fast paths on/off (DUET_DBG_CLUSTER_EXACT = 0x100: everything through the exact linkage)
    # ... the launch structure of large inputs (DUET_DBG_CLUSTER_LARGE = 0x200), the pair sort (DUET_DBG_CLUSTER_PAIRS = 0x400),
    # no bounding-box test (DUET_DBG_CLUSTER_NOBOX = 0x800: every partition through the threshold-graph pair loops);
    # DUET_DBG_CLUSTER_KC2 = 0x1000: partitions with more than two groups left take the second tier;
    # DUET_DBG_CLUSTER_TIERS = 0x2000: the two tiers in their fused (small-input) launches; DUET_DBG_CLUSTER_SMALLCAP = 0x4000 /
    # DUET_DBG_CLUSTER_LSD = 0x8000: where the low bits are sorted locally, groups of more than 3 keys through the
    # one-workgroup-per-group path / plain LSD passes instead
    # DUET_DBG_CLUSTER_KEYSORT = 0x10000: the key-only sort of rounds 1-3 (+ the gather through the permutation) where the
    # default now carries the 16-byte record with the key; DUET_DBG_CLUSTER_NOSYM = 0x20000: the pair tests of one-partition units column
    # by column (every ordered pair) instead of every unordered pair once; DUET_DBG_CLUSTER_RECSORT = 0x40000: the record sort also
    # below 1.25 M marks (the default there is the key-only sort; DUET_DBG_CLUSTER_LARGE implies it)
    for hints, dbg in ((True, 0), (False, 0), (True, 0x100), (True, 0x200), (True, 0x400), (False, 0x600), (True, 0x800),
                       (True, 0xA00), (True, 0x300), (True, 0x1000), (True, 0x1800), (False, 0x1A00), (True, 0x2000), (True, 0x3800), (True, 0x4000), (False, 0x8000),
                       (True, 0x10000), (False, 0x10200), (True, 0x14000), (False, 0x4200), (True, 0x10800),
                       (True, 0x20000), (False, 0x20200), (True, 0x21200),
                       (True, 0x40000), (False, 0x40000), (True, 0x40100), (True, 0x40800), (True, 0x44000), (True, 0x41000), (True, 0x42000), (True, 0x60000),
                       # DUET_DBG_CLUSTER_EVENT_FORKS = 0x4000000 (round 6): the side streams behind events instead of signal / gate kernels
                       (True, 0x4000000), (False, 0x4040000), (True, 0x4002000),
                       # DUET_DBG_CLUSTER_WIDE_OFF = 0x8000000: small inputs keep one wavefront per partition of more than 32 marks (the default
                       # since round 6: four, cl_wide_list / cl_wide_big); DUET_DBG_CLUSTER_WIDE_ALL = 0x10000000: every listed partition on four
                       (True, 0x8000000), (False, 0x8000800), (True, 0x8000100), (True, 0xC000000), (True, 0x10000000), (False, 0x10000800),
                       (True, 0x10000100), (True, 0x14000000), (False, 0x10040000), (True, 0x10040800)):
        ctx.set_debug(dbg)
        try:
            got = ctx.cluster_host(marks['contig'], marks['type'], marks['pos'], marks['span'], hints=hints, **kw)
        finally:
            ctx.set_debug(0)
        for f in FIELDS:
            assert got[f].shape == want[f].shape, (f, got[f].shape, want[f].shape)
            bad = np.nonzero(got[f] != want[f])[0]
            assert bad.size == 0, (f, dbg, bad[:5], got[f][bad[:5]], want[f][bad[:5]])
    return want


def random_marks(seed, M, clumps=40, contigs=3, types=3, spread=60, span_lo=50, span_hi=2000):
    rng = synth.SplitMix(7000 + seed)
    pos = rng.between(M, 0, clumps) * 1500 + rng.between(M, 0, spread) + 1
    span = rng.between(M, span_lo, span_hi)
    span = np.where(rng.chance(M, 1, 2), (span // 100) * 100 + rng.between(M, 0, 8), span)
    return dict(contig=rng.below(M, contigs).astype(np.uint16), type=rng.below(M, types).astype(np.uint8),
                pos=pos.astype(np.uint32), span=span.astype(np.uint32))


def sv_like_marks(seed, n_sv):
    """SV-like input: tight groups of jittered marks, neighbours at distances spread around the thresholds --
    what the threshold-graph fast path (atoms, then linkage over atoms) is made for, including its refusals."""
    rng = synth.SplitMix(31000 + seed)
    near = rng.between(n_sv, 0, 1300)
    gap = np.where(rng.chance(n_sv, 2, 3), near, 2500 + near)
    centre = 10000 + np.cumsum(gap)
    span = rng.between(n_sv, 60, 3000)
    same = rng.chance(n_sv, 1, 2)
    for i in range(1, n_sv):                       # every other SV inherits (roughly) its neighbour's span
        if same[i]:
            span[i] = max(1, int(span[i - 1] * (0.6 + 0.01 * (near[i] % 80))))
    per = 1 + rng.below(n_sv, 25)
    pj = np.array([0, 5, 40, 150])[rng.below(n_sv, 4)]
    sj = np.array([0, 2, 6, 15])[rng.below(n_sv, 4)]
    sv = np.repeat(np.arange(n_sv), per)
    M = len(sv)
    p = centre[sv] + rng.between(M, 0, 301) * pj[sv] // 300
    sp = np.maximum(span[sv] * (100 + rng.between(M, 0, 31) * sj[sv] // 30) // 100, 0)
    perm = rng.permutation(M) if hasattr(rng, 'permutation') else np.argsort(rng.below(M, 1 << 30), kind='stable')
    return dict(contig=((sv // 400) % 2).astype(np.uint16)[perm], type=((sv // 200) % 2).astype(np.uint8)[perm],
                pos=p.astype(np.uint32)[perm], span=sp.astype(np.uint32)[perm])


@pytest.mark.parametrize('seed', range(8))
def test_sv_like_marks(ctx, seed):
    marks = sv_like_marks(seed, 1500)
    check(ctx, marks, max_dist=[0.9, 0.3, 0.5, 0.7, 1.2, 0.9, 0.15, 0.45][seed], part_max=[100, 100, 128, 100, 60, 100, 100, 100][seed])


@pytest.mark.parametrize('seed', range(6))
def test_random_marks(ctx, seed):
    marks = random_marks(seed, 3000 + 700 * seed)
    check(ctx, marks, max_dist=[0.3, 0.5, 0.9, 1.4, 0.9, 0.05][seed])


@pytest.mark.parametrize('seed', range(6))
def test_genome_sized_coordinates(ctx, seed):
    """Positions around 2e8 (28-bit centres: keys of 30+ bits): small inputs then sort the top 16 key bits globally and the low
    bits locally -- sparse data through the rank count in LDS (rx_local), dense clumps through the per-group passes (rx_big)
    -- each case also with groups cut at 3 keys and with plain LSD passes (check()'s debug combinations)."""
    if seed % 2:
        marks = sv_like_marks(40 + seed, 1200 + 300 * seed)
    else:
        marks = random_marks(40 + seed, 6000 + 2500 * seed, clumps=50 + 400 * seed, contigs=1 + seed % 3, types=2)
    marks = dict(marks, pos=(marks['pos'].astype(np.uint64) + 200000000 + 7777 * seed).astype(np.uint32))
    check(ctx, marks, max_dist=[0.9, 0.5, 0.3, 0.9, 1.2, 0.7][seed])


def test_genome_sized_keys_of_35_bits(ctx):
    """20 contigs x 3 types x 28-bit centres: 35 key bits, the width at which LARGE inputs (the launch structure
    DUET_DBG_CLUSTER_LARGE selects in check()) sort the top 24 bits globally and the low 11 locally (rx_local<256>)."""
    for seed in (0, 1):
        marks = random_marks(70 + seed, 9000 + 4000 * seed, clumps=30 + 3000 * seed, contigs=20, types=3, spread=400 + 900 * seed)
        marks = dict(marks, pos=(marks['pos'].astype(np.uint64) + 240000000).astype(np.uint32))
        check(ctx, marks, max_dist=[0.9, 0.4][seed])


@pytest.mark.parametrize('g', [255, 256, 257, 1023, 1024, 1025, 3000])
def test_local_sort_group_sizes_at_its_limits(ctx, g):
    """Groups (keys that agree in the top 16 key bits: one type within 16 k centres here) of exactly g keys around the local
    sort's limits -- 256 keys for the rank count in a tile's window, 1024 for the one-workgroup LDS path, LSD passes beyond --
    between ordinary small groups, on both sides of a tile boundary."""
    rng = synth.SplitMix(4242 + g)
    parts = []
    base = 200000000
    for k, size in enumerate([40, g, 17, g, 300, 5, g]):
        win = base + k * 16384 * 3                   # every group in a window of its own (16384 centres = the low 14 bits)
        pos = win + rng.between(size, 0, 12000)
        parts.append((pos, rng.between(size, 30, 4000)))
    pos = np.concatenate([p for p, _ in parts]).astype(np.uint32)
    span = np.concatenate([s_ for _, s_ in parts]).astype(np.uint32)
    M = len(pos)
    perm = np.argsort(rng.below(M, 1 << 30), kind='stable')
    marks = dict(contig=np.zeros(M, dtype=np.uint16), type=np.zeros(M, dtype=np.uint8), pos=pos[perm], span=span[perm])
    check(ctx, marks, part_gap=1000)


def test_genome_sized_coordinates_edge_sizes(ctx):
    """... around the local sort's tile (2048 positions) and halo (256): groups that start in one tile and end in the next."""
    for M in (1, 2, 3, 255, 2047, 2048, 2049, 2303, 2304, 2305, 4097, 6144):
        marks = random_marks(500 + M, M, clumps=2 + M % 5, contigs=1, types=1 + M % 2, spread=900)
        marks = dict(marks, pos=(marks['pos'].astype(np.uint64) + 230000000).astype(np.uint32))
        check(ctx, marks)


def test_edge_sizes(ctx):
    for M in (1, 2, 3, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097):
        check(ctx, random_marks(M, M, clumps=3))


def test_big_partitions_hit_the_cap(ctx):
    """Hundreds of marks within one gap-free stretch: partitions are cut every part_max marks and the large
    (49..128 marks) kernel variant runs."""
    rng = synth.SplitMix(42)
    M = 5000
    pos = 100000 + rng.between(M, 0, 3000)
    marks = dict(contig=np.zeros(M, dtype=np.uint16), type=np.zeros(M, dtype=np.uint8), pos=pos.astype(np.uint32),
                 span=rng.between(M, 100, 140).astype(np.uint32))
    want = check(ctx, marks)
    check(ctx, marks, part_max=128, max_dist=0.4)
    check(ctx, marks, part_max=37, part_gap=5)
    assert len(want['cand_off']) - 1 < M


def test_threshold_exactly_on_a_pair_distance(ctx):
    """max_dist equal to pair distances that occur (d = k/900 for equal spans, d = 1/3 + ...): the fast path's
    guard band must hand these partitions to the exact agglomeration."""
    M = 4000
    rng = synth.SplitMix(99)
    pos = 50000 + rng.between(M, 0, 40) * 2000 + rng.between(M, 0, 5) * 90
    span = np.where(rng.chance(M, 1, 3), 300, 200)
    marks = dict(contig=np.zeros(M, dtype=np.uint16), type=np.zeros(M, dtype=np.uint8), pos=pos.astype(np.uint32),
                 span=span.astype(np.uint32))
    for md in (0.1, 0.2, 90 / 900, 180 / 900, 1 / 3, 1 / 3 + 0.1, 0.0, 1e-300, -1.0, 1e9, float('inf')):
        check(ctx, marks, max_dist=md)


def test_positions_near_the_32_bit_edge(ctx):
    M = 900
    rng = synth.SplitMix(5)
    pos = 0xFFFFFFFF - rng.between(M, 0, 3000)
    span = rng.between(M, 0, 6000)
    marks = dict(contig=np.zeros(M, dtype=np.uint16), type=np.zeros(M, dtype=np.uint8), pos=pos.astype(np.uint32),
                 span=span.astype(np.uint32))
    check(ctx, marks)
    check(ctx, marks, normalizer=1e-4)
    check(ctx, marks, normalizer=1e12, max_dist=1e-7)


def test_zero_spans_and_identical_marks(ctx):
    M = 600
    marks = dict(contig=np.zeros(M, dtype=np.uint16), type=(np.arange(M) % 2).astype(np.uint8),
                 pos=np.full(M, 777, dtype=np.uint32), span=np.zeros(M, dtype=np.uint32))
    check(ctx, marks)


def test_config2_marks_recover_candidates(ctx):
    """The raw marks behind BASELINE config 2 (1.0M marks): full size, GPU == oracle, and clustering groups
    the marks of a candidate together whenever candidates are far apart."""
    contigs = H.case_contigs('config2', 1)
    marks = synth.raw_marks(contigs, 1)
    assert 990000 < len(marks['pos']) < 1010000
    want = check(ctx, marks)
    assert 60000 < len(want['cand_off']) - 1 < 140000


def test_genome_marks(ctx):
    marks = synth.raw_marks(synth.bench_genome(400000, 3), 3)
    check(ctx, marks)


def test_between_one_and_four_million_marks(ctx):
    """1.5 M marks (367 radix tiles): the sort's middle regime -- digit totals by atomics + one rx_offsets launch per pass
    (up to 256 tiles every scatter block sums the tiles before it itself; beyond 1024 the generic scan takes over)."""
    marks = synth.raw_marks([synth.bench_contig('1', 300000, 150000, 11)], 11)
    assert 1400000 < len(marks['pos']) < 1600000
    want = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'])
    for dbg in (0x10000, 0):             # DUET_DBG_CLUSTER_KEYSORT: the key-only sort this test was written for; 0: the record sort (the default from 1.25 M marks on)
        ctx.set_debug(dbg)
        try:
            got = ctx.cluster_host(marks['contig'], marks['type'], marks['pos'], marks['span'])
        finally:
            ctx.set_debug(0)
        for f in FIELDS:
            assert np.array_equal(got[f], want[f]), (f, dbg)


def test_between_two_and_four_million_marks(ctx):
    """2.3 M marks: still a small input (gate forks, one launch for the classes up to 64 marks), but beyond the sizes at which the partitions of
    more than 64 marks take the wide units and the side stream's join sits inside cl_pc_sums (2 M marks): cl_tight_big + cl_link_one behind a
    gate, a gate kernel of its own in front of cl_pc_sums."""
    marks = synth.raw_marks([synth.bench_contig('1', 460000, 230000, 13)], 13)
    assert 2200000 < len(marks['pos']) < 2500000
    want = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'])
    got = ctx.cluster_host(marks['contig'], marks['type'], marks['pos'], marks['span'])
    for f in FIELDS:
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize('M', [3, 200, 5000, 70000])
def test_wide_keys_few_marks(ctx, M):
    """Positions next to 2^32 with contig ids next to 65535 and many types: 33 + 16 + 8 = 57 key bits.  With few marks the
    index still fits the key's spare bits, and the local ordering of the low bits (32-bit words) must not be offered more
    than 31 of them: such keys take more global passes (round-3 advisor finding); contig + type + index do not fit one record
    word either, so this is the key-only sort."""
    rng = synth.SplitMix(9100 + M)
    pos = (0xFFFFFFFF - 3000000 + rng.between(M, 0, 40) * 70000 + rng.between(M, 0, 900)).astype(np.uint32)
    span = rng.between(M, 30, 3000).astype(np.uint32)
    contig = (65535 - rng.below(M, 3) * 20000).astype(np.uint16)
    typ = (255 - rng.below(M, 2) * 200).astype(np.uint8)
    check(ctx, dict(contig=contig, type=typ, pos=pos, span=span))


def test_record_sort_digit_plans(ctx):
    """The record sort's pass plan over sizes and key widths: one, two and three global digits, digits of fewer than eight
    bits, no local stage at all (tiny keys), and a contig / type / index word filled to its 32 bits."""
    cases = [
        (300, dict(clumps=3, contigs=1, types=1, spread=50), 0),                # 13-bit keys: a single digit, lo = 5
        (40000, dict(clumps=300, contigs=2, types=2, spread=700), 0),           # 21-bit keys
        (150000, dict(clumps=4000, contigs=24, types=2, spread=900), 200000000),# 34-bit keys, 13 top bits: two digits
        (9000, dict(clumps=50, contigs=4000, types=3, spread=300), 0),          # 12 contig + 2 type + 14 index bits
        (33000, dict(clumps=60, contigs=60000, types=2, spread=300), 0),        # 16 + 1 + 15 = 32 bits of the record word
    ]
    for k, (M, kw, off) in enumerate(cases):
        marks = random_marks(880 + k, M, **kw)
        if off:
            marks = dict(marks, pos=(marks['pos'].astype(np.uint64) + off).astype(np.uint32))
        check(ctx, marks, max_dist=[0.9, 0.5, 0.9, 0.3, 0.9][k])
