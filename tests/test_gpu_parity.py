# coding=utf-8
"""-m gpu: the HIP path through the C ABI against the oracles and the reference goldens.
Bit-exact: pred (u8) and ps (u32) per candidate, and the bytes of phased_sv.vcf."""
import os
import shutil

import numpy as np
import pytest

from duet_amd import _lib, engine, synth
from duet_amd.sv_phasing import sv_phasing
from oracle import c_oracle
from tests import helpers as H
from tests import soa_fuzz
from tests.test_c_oracle import materialise_bams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def check_against_c_oracle(ctx, soa, svlen_thres=50, suppread_thres=2):
    rc, want_pred, want_ps = c_oracle.ef(soa, svlen_thres, suppread_thres)
    assert rc == 0
    pred, ps = ctx.run_host(soa, svlen_thres, suppread_thres)
    bad = np.nonzero((pred != want_pred) | (ps != want_ps))[0]
    assert bad.size == 0, 'first mismatches at %s: got %s/%s want %s/%s' % (
        bad[:5], pred[bad[:5]], ps[bad[:5]], want_pred[bad[:5]], want_ps[bad[:5]])
    return pred, ps


def run_product(home, svlen_thres, suppread_thres, python_path=None):
    """sv_phasing through the native host path (default) or the Python host path; both end in the HIP kernels.
    With python_path=None both are run and must agree."""
    outs = []
    for force_py in ((False, True) if python_path is None else (python_path,)):
        old = os.environ.get('DUET_NATIVE_INGEST')
        os.environ['DUET_NATIVE_INGEST'] = '0' if force_py else '1'
        try:
            sv_phasing(home, svlen_thres, suppread_thres, 4, False)
        finally:
            if old is None:
                del os.environ['DUET_NATIVE_INGEST']
            else:
                os.environ['DUET_NATIVE_INGEST'] = old
        with open(os.path.join(home, 'phased_sv.vcf')) as f:
            outs.append(f.read())
    assert all(o == outs[0] for o in outs)
    return outs[0]


@pytest.mark.parametrize('name,src,params', H.full_cases(), ids=[c[0] for c in H.full_cases()])
def test_golden_cases_bytes(name, src, params, tmp_path):
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    os.remove(os.path.join(home, 'phased_sv.vcf'))
    materialise_bams(home)
    with open(os.path.join(src, 'phased_sv.vcf')) as f:
        want = f.read()
    assert run_product(home, params['svlen_thres'], params['suppread_thres']) == want


def test_seeded_cases_sha(tmp_path):
    for p in H.seeded_plan():
        if p['kind'] == 'config2':
            continue
        home = str(tmp_path / ('%s_%d_%s' % (p['kind'], p['seed'], p['dialect'])))
        H.build_case(home, p['kind'], p['seed'], p['dialect'], write_sam=False)
        got = run_product(home, p['svlen_thres'], p['suppread_thres'], python_path=bool(p['seed'] % 2))
        assert H.sha256_bytes(got.encode()) == p['output_sha256'], p
        shutil.rmtree(home)


def test_config2_text_sha(tmp_path):
    """BASELINE config 2 (1 contig, ~1M marks / 200k reads / 100k candidates) through the whole
    product path, byte-identical to the reference's phased_sv.vcf (sha256 captured in make_golden)."""
    plan = [p for p in H.seeded_plan() if p['kind'] == 'config2']
    assert plan, 'config2 golden missing'
    p = plan[0]
    home = str(tmp_path / 'config2')
    H.build_case(home, 'config2', p['seed'], p['dialect'], write_sam=False)
    got = run_product(home, p['svlen_thres'], p['suppread_thres'])
    assert sum(1 for l in got.splitlines() if not l.startswith('#')) == p['rows']
    assert H.sha256_bytes(got.encode()) == p['output_sha256']


def test_config2_soa_vs_oracle(ctx):
    soa = engine.soa_from_synth(H.case_contigs('config2', 1))
    assert 990000 < soa.n_marks < 1010000 and soa.n_cands == 100000
    pred, _ = check_against_c_oracle(ctx, soa)
    assert 40000 < int((pred != 0).sum()) < 60000


def test_config3_full_size_soa_vs_oracle(ctx):
    """BASELINE configs[2] at full size on one GPU: 24 contigs, 2e7 marks, 2e6 candidates, every (pred, ps) against the C oracle."""
    soa = engine.soa_from_synth(synth.bench_genome(20000000, 3))
    assert soa.n_contigs == 24 and 19900000 < soa.n_marks < 20100000 and 1990000 < soa.n_cands < 2010000
    pred, _ = check_against_c_oracle(ctx, soa)
    assert 800000 < int((pred != 0).sum()) < 1200000


@pytest.mark.parametrize('seed', range(12))
def test_fuzz_soa(ctx, seed):
    soa = soa_fuzz.random_soa(seed, n_contigs=1 + seed % 5, sorted_pos=bool(seed % 3))
    check_against_c_oracle(ctx, soa, 50, 2)
    check_against_c_oracle(ctx, soa, 0, 0)
    check_against_c_oracle(ctx, soa, 52, 4)


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_finalize_with_two_and_four_tiles_per_workgroup(ctx, seed):
    """ef_finalize takes 2 / 4 tiles of 256 candidates per workgroup from 1 M / 8 M candidates on; the debug bits 0x20 / 0x80 force
    that on small problems: contigs that start and end inside a workgroup's candidates, contigs without seeds, several phase sets."""
    soa = soa_fuzz.random_soa(5000 + seed, n_contigs=6, cands_per_contig=(100, 1500), reads_per_contig=(100, 700), n_ps=(1, 12))
    for dbg in (0x20, 0x80):
        ctx.set_debug(dbg)
        try:
            for thres in ((50, 2), (0, 0)):
                check_against_c_oracle(ctx, soa, *thres)
        finally:
            ctx.set_debug(0)


@pytest.mark.parametrize('n_ps', [3, 40, 130, 300, 500, 700, 1900, 2300, 6000])
def test_unsorted_candidates_seed_set_paths(ctx, n_ps):
    """Candidates NOT in position order with n_ps phase sets per contig: the seed list arrives unsorted.  Few distinct seeds go
    through ef_seed_sort's hash set (up to 2048 of them), more through the run merges / the bitonic network; each case also with
    the hash set switched off (debug bit 0x40) -- the seed sets (and everything decided from them) are the same."""
    soa = soa_fuzz.random_soa(900 + n_ps, n_contigs=2, cands_per_contig=(9000, 14000), reads_per_contig=(4000, 6000),
                              n_ps=(n_ps, n_ps), ps_spread=3000000, deg=(1, 6), empty_contig_rate=0, no_seed_contig_rate=0,
                              sorted_pos=False)
    check_against_c_oracle(ctx, soa)
    seeds = [ctx.seed_ps(k).copy() for k in range(2)]
    ctx.set_debug(0x40)
    try:
        check_against_c_oracle(ctx, soa)
        for k in range(2):
            assert np.array_equal(ctx.seed_ps(k), seeds[k]) and np.all(seeds[k][1:] > seeds[k][:-1])
    finally:
        ctx.set_debug(0)


def seed_list_soa(seqs):
    """One contig per sequence; candidate i of a contig is a class-1 candidate whose two marks are one read of phase set seq[i]:
    ef_seed_sort's list for the contig is seq with consecutive repeats dropped."""
    tags, mark_read, cand_off, ctg_off, read_off = [], [], [0], [0], [0]
    cols = dict(pos=[], svlen=[], svread=[], refread=[], gt=[])
    for seq in seqs:
        base = len(tags)
        for i, ps in enumerate(seq):
            tags.append((1 + i % 2, 10 + i % 50, ps))
            mark_read += [i, i]                                # (contig-local read ids)
            cand_off.append(len(mark_read))
            cols['pos'].append(1000 + 7 * i); cols['svlen'].append(100); cols['svread'].append(5); cols['refread'].append(5); cols['gt'].append(1)
        ctg_off.append(len(cols['pos'])); read_off.append(len(tags))
    hap = np.array([t[0] for t in tags]); pc = np.array([t[1] for t in tags]); ps = np.array([t[2] for t in tags])
    # mark_read: global read index = contig's read offset + local id
    mr = np.array(mark_read, dtype=np.uint64)
    co = np.array(cand_off)
    for k in range(len(seqs)):
        lo, hi = co[ctg_off[k]], co[ctg_off[k + 1]]
        mr[lo:hi] += read_off[k]
    return engine.EfSoA(cand_ctg_off=ctg_off, read_off=read_off, read_tag=engine.pack_tags(hap, pc, ps), cand_pos=cols['pos'],
                        cand_svlen=cols['svlen'], cand_svread=cols['svread'], cand_refread=cols['refread'], cand_gt_ok=cols['gt'],
                        cand_off=cand_off, mark_read=mr)


def test_seed_lists_with_local_disorder(ctx):
    """ef_seed_sort exchanges the two entries of every descent at once when each looks like one transposition, and lets
    unique_copy check the order: lists where that is all it takes (A B A B at phase-set boundaries), lists where two exchanges two
    places apart disturb each other (1 5 2 6 3: the check fails, the ladder runs after all -- few descents: the run merge; many:
    the hash set), descents in a row, a descending list, and tiles of 256 candidates cut anywhere through the patterns; each also
    with the hash set and the exchanges switched off (debug bit 0x40).  The seed arrays are np.sort(list(set(...))) of the
    sequence."""
    rng = np.random.default_rng(5)
    blocks = lambda pat, reps, step: [x + step * j for j in range(reps) for x in pat]
    seqs = [
        blocks([10, 50], 300, 100),                                        # ascending outright
        blocks([10, 50, 10, 50, 10, 90], 150, 100),                        # A B A B A C: one exchange each
        blocks([10, 50, 20, 60, 30], 100, 100),                            # exchanges that disturb each other, many of them
        [10, 50, 20, 60, 30],                                              # ... and few
        blocks([30, 20, 10], 120, 100),                                    # descents in a row
        list(range(5000, 0, -7)),                                          # descending
        [7],                                                               # one seed
        blocks([10, 50, 10], 2, 100) + list(range(1000, 4000, 3)),         # a few descents in front of a long ascending run
        (np.sort(rng.integers(1, 1 << 30, size=3000)) + rng.integers(-2, 3, size=3000)).tolist(),     # jitter on an ascending list
    ]
    soa = seed_list_soa(seqs)
    for dbg in (0, 0x40):
        ctx.set_debug(dbg)
        try:
            check_against_c_oracle(ctx, soa)
            for k, seq in enumerate(seqs):
                assert np.array_equal(ctx.seed_ps(k), np.unique(np.array(seq, dtype=np.uint32))), (dbg, k)
        finally:
            ctx.set_debug(0)


def test_multi_ps_heavy(ctx):
    """Most candidates see several phase sets (more than the 4 groups a summary holds, and more
    multi-PS candidates per workgroup than summary slots): exercises both ef_finalize paths."""
    soa = soa_fuzz.random_soa(77, n_contigs=2, cands_per_contig=(2000, 3000), reads_per_contig=(60, 90),
                              n_ps=(9, 14), deg=(4, 30), empty_contig_rate=0, no_seed_contig_rate=0)
    check_against_c_oracle(ctx, soa)
    soa = soa_fuzz.random_soa(78, n_contigs=3, cands_per_contig=(1000, 2000), reads_per_contig=(40, 200),
                              n_ps=(2, 5), deg=(2, 12), empty_contig_rate=0, no_seed_contig_rate=0)
    check_against_c_oracle(ctx, soa)


def test_two_ps_everywhere_summary_pool(ctx):
    """Nearly every candidate sees exactly two phase sets: every tile needs more group-summary slots than its own 64, so the
    shared pool is used -- and, with few candidates (a pool of 256 slots), exhausted, which leaves the rest to ef_finalize's
    walk over the marks.  Run twice on one context: the pool's counter has to start from zero again."""
    small = soa_fuzz.random_soa(79, n_contigs=2, cands_per_contig=(1500, 2500), reads_per_contig=(30, 60),
                                n_ps=(2, 2), deg=(4, 16), empty_contig_rate=0, no_seed_contig_rate=0)
    big = soa_fuzz.random_soa(80, n_contigs=3, cands_per_contig=(20000, 30000), reads_per_contig=(200, 400),
                              n_ps=(2, 2), deg=(3, 12), empty_contig_rate=0, no_seed_contig_rate=0)
    for soa in (small, big, small):
        check_against_c_oracle(ctx, soa)
        check_against_c_oracle(ctx, soa, 0, 0)


HEAVY_ALL, HEAVY_OFF, WALK_R4, FP_DECIDE = 0x80000, 0x100000, 0x200000, 0x400000     # include/duet_ef.h: DUET_DBG_EF_HEAVY_ALL / _OFF / _WALK_R4 / _FP_DECIDE


@pytest.mark.parametrize('case', ['fuzz', 'multi_ps', 'two_ps_pool', 'tail', 'cross_chunk', 'config2'])
def test_wave_cooperative_walk_and_lane_walk_agree(ctx, case):
    """ef_classify has two walks over a candidate's marks: one lane, mark after mark -- and the whole wavefront, 64 marks per
    step (first voter by ballot and find-first-set, counts by population count, PC sums by a wave reduction), taken for
    candidates with more than 32 marks.  Each walk forced on every candidate (debug bits), and the default mix: same bytes
    as the oracle -- first-seen group order, the third-group flag, chunk crossings (a candidate whose marks span several LDS
    passes re-enters the cooperative walk with its state) and absent marks included."""
    if case == 'fuzz':
        soas = [soa_fuzz.random_soa(4000 + i, n_contigs=1 + i % 4, sorted_pos=bool(i % 2), deg=(1, 70)) for i in range(6)]
    elif case == 'multi_ps':
        soas = [soa_fuzz.random_soa(77, n_contigs=2, cands_per_contig=(2000, 3000), reads_per_contig=(60, 90), n_ps=(9, 14),
                                    deg=(4, 130), empty_contig_rate=0, no_seed_contig_rate=0),
                soa_fuzz.random_soa(78, n_contigs=3, cands_per_contig=(1000, 2000), reads_per_contig=(40, 200), n_ps=(2, 5),
                                    deg=(2, 12), empty_contig_rate=0, no_seed_contig_rate=0)]
    elif case == 'two_ps_pool':
        soas = [soa_fuzz.random_soa(79, n_contigs=2, cands_per_contig=(1500, 2500), reads_per_contig=(30, 60), n_ps=(2, 2),
                                    deg=(4, 66), empty_contig_rate=0, no_seed_contig_rate=0)]
    elif case == 'tail':
        # most candidates small, one in fifty with 33..200 marks: the mix the default threshold splits
        soas = []
        for i, big in enumerate((33, 64, 65, 129, 200)):
            soas.append(soa_fuzz.random_soa(4100 + i, n_contigs=2, cands_per_contig=(600, 900), reads_per_contig=(300, 500),
                                            deg=(1, 14), big_deg=big, empty_contig_rate=0, no_seed_contig_rate=0, n_ps=(1, 4)))
    elif case == 'cross_chunk':
        soas = [soa_fuzz.random_soa(101, n_contigs=2, cands_per_contig=(300, 600), reads_per_contig=(500, 900), big_deg=9000,
                                    empty_contig_rate=0, no_seed_contig_rate=0, n_ps=(1, 3)),
                soa_fuzz.random_soa(102, n_contigs=1, cands_per_contig=(300, 400), reads_per_contig=(500, 900), big_deg=3100,
                                    deg=(20, 90), empty_contig_rate=0, no_seed_contig_rate=0)]
    else:
        soas = [engine.soa_from_synth(H.case_contigs('config2', 1))]
    for soa in soas:
        for dbg in (HEAVY_ALL, HEAVY_OFF, 0, WALK_R4, WALK_R4 | HEAVY_OFF, FP_DECIDE, FP_DECIDE | WALK_R4 | HEAVY_OFF):
            ctx.set_debug(dbg)
            try:
                check_against_c_oracle(ctx, soa)
                if case != 'config2':
                    check_against_c_oracle(ctx, soa, 0, 0)
            finally:
                ctx.set_debug(0)


def decision_boundary_soa():
    """One contig, one phase set (every candidate class 1 or 0), candidates built ON the thresholds of predict_hp
    (sv_phasing_fn.py:142-183) and one step to either side: sv_ratio = svread / (svread + refread) at 0.24, 0.3, 0.45, 0.75, 0.9, 1
    (small and 32-bit-sized counts), totsc_ratio = max / min of the two PC sums at 9.72, |hap2_avgsc - hap1_avgsc| at 2400 with
    quotients that are not exact (thirds, sevenths), hapread_ratio at 0.75, one-haplotype and two-haplotype votes, PC 0."""
    PS = 5017
    sv_pairs = []
    for pn, qd in ((6, 25), (3, 10), (9, 20), (3, 4), (9, 10), (1, 1)):
        for k in (1, 7, 1000, 40000000, 170000000):
            s0, tot = pn * k, qd * k
            if tot >= 2 ** 32:
                continue
            for ds in (-1, 0, 1):
                s = s0 + ds
                if 2 <= s <= tot:
                    sv_pairs.append((s, tot - s))
    sv_pairs += [(4, 0), (3, 0), (5, 1), (2 ** 32 - 1, 0), (2 ** 31, 2 ** 31 - 1), (11, 11), (20, 20)]
    votes = []                                     # (pcs of hap-1 voters, pcs of hap-2 voters, untagged marks)
    def split(t, h):
        base = [t // h] * h
        for i in range(t - (t // h) * h):
            base[i] += 1
        assert max(base) <= 8100 and sum(base) == t
        return base
    # totsc_ratio around 9.72 (both sums positive)
    for k in (1, 3, 17):
        for d in (-1, 0, 1):
            votes.append((split(972 * k + d, max(1, k)), split(100 * k, 1), 0))
            votes.append((split(100 * k, 1), split(972 * k + d, max(1, k)), 1))
    # |a2 - a1| around 2400 with inexact quotients: t2 / 7 - t1 / 3 = 2400 <=> 3 t2 - 7 t1 = 50400
    for t1 in (3000, 3001, 2999):
        for d in (-1, 0, 1):
            t2 = (50400 + 7 * t1) // 3 + d
            votes.append((split(t1, 3), split(t2, 7), 0))
            votes.append((split(t2, 7), split(t1, 3), 0))
            votes.append((split(t1, 3), split(t2, 7), 5))          # hapread_ratio <= 0.75: the gate looks at the difference
    # one haplotype only: the difference is one quotient, around 2400 (t = 2400 h, +- 1), with untagged marks so that hp <= 0.75
    for h in (1, 3, 7):
        for d in (-1, 0, 1):
            votes.append((split(2400 * h + d, h), [], h))
            votes.append(([], split(2400 * h + d, h), h + 1))
            votes.append((split(2400 * h + d, h), [], 0))
    # PC 0 votes (sums 0 with votes present), no votes at all, hapread_ratio exactly 0.75
    votes += [([0, 0], [], 0), ([0], [0], 0), ([], [], 4), ([300, 300, 300], [], 1), ([300], [100], 2), ([8100], [8100, 8100], 0)]
    tags, mark_read, cand_off = [], [], [0]
    cols = dict(pos=[], svlen=[], svread=[], refread=[], gt=[])
    pos = 1000
    for (p1, p2, n_un) in votes:
        for (sv, rf) in sv_pairs:
            for pc in p1:
                mark_read.append(len(tags)); tags.append((1, pc, PS))
            for pc in p2:
                mark_read.append(len(tags)); tags.append((2, pc, PS))
            mark_read += [engine.MARK_ABSENT] * n_un
            cand_off.append(len(mark_read))
            cols['pos'].append(pos); pos += 3
            cols['svlen'].append(100); cols['svread'].append(sv); cols['refread'].append(rf); cols['gt'].append(1)
    hap = np.array([t[0] for t in tags]); pc = np.array([t[1] for t in tags]); ps = np.array([t[2] for t in tags])
    C = len(cols['pos'])
    return engine.EfSoA(cand_ctg_off=[0, C], read_tag=engine.pack_tags(hap, pc, ps), cand_pos=cols['pos'], cand_svlen=cols['svlen'],
                        cand_svread=np.array(cols['svread'], dtype=np.uint64), cand_refread=np.array(cols['refread'], dtype=np.uint64),
                        cand_gt_ok=cols['gt'], cand_off=cand_off, mark_read=np.array(mark_read, dtype=np.uint64))


def test_decision_on_the_thresholds_integer_and_binary64(ctx):
    """ef_classify decides classes 0 and 1 in integers wherever integers provably give what the binary64 expressions of
    sv_phasing_fn.py:112-183 give (and falls back to those expressions for an exact tie of two rounded quotients or sums beyond
    2^31): candidates ON every threshold and one step to either side -- sv_ratio at 0.24 / 0.3 / 0.45 / 0.75 / 0.9 / 1 with counts
    from 6 to 4e9, totsc_ratio at 9.72, the average-score difference at 2400 with inexact quotients, one-haplotype votes -- against
    the C oracle (binary64, pinned to the reference's known answers), with the integer form, and with DUET_DBG_EF_FP_DECIDE."""
    soa = decision_boundary_soa()
    assert soa.n_cands > 5000
    rc, want_pred, _ = c_oracle.ef(soa, 50, 2)
    assert rc == 0 and len(set(want_pred.tolist())) == 4           # every outcome occurs
    for dbg in (0, FP_DECIDE, HEAVY_ALL, WALK_R4):
        ctx.set_debug(dbg)
        try:
            check_against_c_oracle(ctx, soa, 50, 2)
        finally:
            ctx.set_debug(0)


def test_long_candidates_cross_lds_chunks(ctx):
    """Candidates with more marks than one LDS pass holds (4096) and than a whole workgroup's pass."""
    soa = soa_fuzz.random_soa(101, n_contigs=2, cands_per_contig=(300, 600), reads_per_contig=(500, 900),
                              big_deg=9000, empty_contig_rate=0, no_seed_contig_rate=0)
    assert int(np.diff(soa.cand_off.astype(np.int64)).max()) == 9000
    check_against_c_oracle(ctx, soa)


def test_many_seed_sets_global_sort(ctx):
    """> 16384 distinct seed PS in one contig: the seed sort leaves LDS for the in-HBM network."""
    soa = soa_fuzz.random_soa(202, n_contigs=1, cands_per_contig=(150000, 150000), reads_per_contig=(100000, 100000),
                              n_ps=(60000, 60000), ps_spread=4000000, deg=(1, 2), empty_contig_rate=0,
                              no_seed_contig_rate=0, sorted_pos=False)
    check_against_c_oracle(ctx, soa)
    assert ctx.seed_ps(0).size > 16384
    s = ctx.seed_ps(0)
    assert np.all(s[1:] > s[:-1])


def test_many_contigs(ctx):
    """-a style contig universe: thousands of small contigs, many empty or seedless."""
    soa = soa_fuzz.random_soa(303, n_contigs=3000, cands_per_contig=(0, 12), reads_per_contig=(4, 30), n_ps=(1, 3),
                              empty_contig_rate=3, no_seed_contig_rate=4)
    check_against_c_oracle(ctx, soa)


def test_plan_reuse_and_relayout(ctx):
    """Back-to-back runs on one context: same layout (nothing is memset between runs), then a different one."""
    a = soa_fuzz.random_soa(7, n_contigs=4)
    b = soa_fuzz.random_soa(8, n_contigs=2)
    for soa in (a, a, b, a, b, b):
        check_against_c_oracle(ctx, soa)


def test_division_by_zero_is_reported(ctx):
    soa = soa_fuzz.random_soa(9, n_contigs=2, allow_divzero=True, empty_contig_rate=0, no_seed_contig_rate=0)
    rc, _, _ = c_oracle.ef(soa, 50, 0)
    assert rc == -5
    with pytest.raises(ZeroDivisionError):
        ctx.run_host(soa, 50, 0)
    # with -r 2 such candidates never reach the decision (svread = 0 < 2), as upstream
    check_against_c_oracle(ctx, soa, 50, 2)


def test_device_pointer_entry_unaligned_marks(ctx):
    """duet_ef_run_device on torch-owned buffers; mark_read deliberately 4-byte (not 16-byte) aligned
    so the scalar staging variant of ef_classify runs too."""
    import torch
    from duet_amd.devmem import DeviceProblem
    soa = soa_fuzz.random_soa(11, n_contigs=3, cands_per_contig=(300, 500))
    rc, want_pred, want_ps = c_oracle.ef(soa, 50, 2)
    for mis, dbg in ((4, 0), (0, 0), (4, HEAVY_ALL), (0, HEAVY_ALL)):
        dp = DeviceProblem(soa, 50, 2, misalign_marks=mis)
        assert (dp.problem.mark_read % 16 != 0) == bool(mis)
        torch.cuda.synchronize()
        ctx.set_debug(dbg)
        try:
            stream = dp.run(ctx)
            ctx.check(stream)
        finally:
            ctx.set_debug(0)
        pred, ps = dp.results()
        assert np.array_equal(pred, want_pred) and np.array_equal(ps, want_ps)


def test_idempotent_and_linear_in_contigs(ctx):
    """Size-independent properties at the full config-2 size: running twice gives the same bytes;
    a problem made of two copies of a contig gives each copy the single-contig answer."""
    one = H.case_contigs('config2', 1)
    soa1 = engine.soa_from_synth(one)
    p1, s1 = ctx.run_host(soa1, 50, 2)
    p1b, s1b = ctx.run_host(soa1, 50, 2)
    assert np.array_equal(p1, p1b) and np.array_equal(s1, s1b)
    soa2 = engine.soa_from_synth(one + one)
    p2, s2 = ctx.run_host(soa2, 50, 2)
    C = soa1.n_cands
    assert np.array_equal(p2[:C], p1) and np.array_equal(p2[C:], p1)
    assert np.array_equal(s2[:C], s1) and np.array_equal(s2[C:], s1)


OWN_OFF, OWN_ALL, OWN_SMALLTAB = 0x800000, 0x1000000, 0x2000000     # include/duet_ef.h: DUET_DBG_EF_OWN_OFF / _ALL / _SMALLTAB


def test_two_launches_three_launches_and_the_array_free_walk_agree(ctx):
    """Round 6: up to 1024 tiles and 64 contigs step E/F is TWO launches -- every finalize tile builds its contig's seed set itself
    (ef_finalize_own: hash set + ordered blocks in LDS) -- beyond that the three of rounds 1-5.  Every case through: the default,
    the three launches forced, the two launches forced (also on the large case), and the two launches with a seed set of 8 entries
    (contigs with more distinct seeds answer "is this PS a seed" / "which seed is nearest" by walking the contig's seed entries);
    the seed arrays the ABI hands out (made on demand after a two-launch run) are the same in all of them."""
    cases = [soa_fuzz.random_soa(6100 + s, n_contigs=1 + s % 5, sorted_pos=bool(s % 3)) for s in range(6)]
    cases.append(soa_fuzz.random_soa(6200, n_contigs=40, cands_per_contig=(0, 400), reads_per_contig=(5, 300), n_ps=(1, 9)))     # tiles over many contigs
    cases.append(soa_fuzz.random_soa(6201, n_contigs=64, cands_per_contig=(0, 40), reads_per_contig=(5, 30)))                     # K = 64: the last two-launch K
    cases.append(soa_fuzz.random_soa(6202, n_contigs=65, cands_per_contig=(0, 40), reads_per_contig=(5, 30)))                     # K = 65: three launches
    for n_ps in (3, 500, 2300, 6000):                                      # 2300 / 6000 distinct seeds: beyond the set in LDS by themselves
        cases.append(soa_fuzz.random_soa(6300 + n_ps, n_contigs=2, cands_per_contig=(9000, 14000), reads_per_contig=(4000, 6000),
                                         n_ps=(n_ps, n_ps), ps_spread=3000000, deg=(1, 6), empty_contig_rate=0, no_seed_contig_rate=0,
                                         sorted_pos=False))
    cases.append(soa_fuzz.random_soa(6400, n_contigs=2, cands_per_contig=(2000, 3000), reads_per_contig=(60, 90), n_ps=(9, 14), deg=(4, 30),
                                     empty_contig_rate=0, no_seed_contig_rate=0))                  # >= 3 voter groups: the marks-based vote
    cases.append(soa_fuzz.random_soa(6401, n_contigs=2, cands_per_contig=(1500, 2500), reads_per_contig=(30, 60), n_ps=(2, 2), deg=(4, 16),
                                     empty_contig_rate=0, no_seed_contig_rate=0))                  # the summary pool, exhausted
    cases.append(soa_fuzz.random_soa(6402, n_contigs=3, allow_divzero=True, empty_contig_rate=0, no_seed_contig_rate=0))
    cases.append(engine.soa_from_synth(H.case_contigs('config2', 1)))
    cases.append(engine.soa_from_synth(synth.bench_genome(4000000, 5)))   # 24 contigs, 4e5 candidates = 1563 tiles: three launches by default
    for i, soa in enumerate(cases):
        supp = 0 if i == len(cases) - 3 else 2
        seeds = None
        for dbg in (0, OWN_OFF, OWN_ALL, OWN_ALL | OWN_SMALLTAB):
            ctx.set_debug(dbg)
            try:
                rc, want_pred, want_ps = c_oracle.ef(soa, 50, supp)
                if rc != 0:
                    with pytest.raises(ZeroDivisionError):
                        ctx.run_host(soa, 50, supp)
                    continue
                check_against_c_oracle(ctx, soa, 50, supp)
                if soa.n_contigs <= 8:
                    got = [ctx.seed_ps(k).copy() for k in range(soa.n_contigs)]
                    if seeds is None:
                        seeds = got
                    assert all(np.array_equal(a, b) for a, b in zip(got, seeds)), (i, dbg)
                if supp == 2 and soa.n_cands < 200000:
                    check_against_c_oracle(ctx, soa, 0, 0)
            finally:
                ctx.set_debug(0)
