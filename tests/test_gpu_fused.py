# coding=utf-8
"""-m gpu: the fused SVIM-mode pipeline (raw marks -> A0 clustering -> E/F phasing on the device) against the
composition of the two C oracles with the adapter rules of include/duet_ef.h restated in numpy."""
import numpy as np
import pytest

from duet_amd import _lib, engine, synth
from duet_amd.devmem import DeviceSvim
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def oracle_fused(marks, read_tag, depth, depth_off, depth_bin, svlen_thres, suppread_thres, n_contigs, max_dist=0.9):
    cl = c_oracle.cluster(marks['contig'], marks['type'], marks['pos'], marks['span'], max_dist=max_dist)
    N = len(cl['cand_pos'])
    off = cl['cand_off'].astype(np.int64)
    support = np.diff(off)
    k = cl['cand_contig'].astype(np.int64)
    nb = np.diff(depth_off.astype(np.int64))[k]
    bins = np.minimum(cl['cand_pos'].astype(np.int64) // depth_bin, np.maximum(nb - 1, 0))
    d = np.where(nb > 0, depth[np.minimum(depth_off[k].astype(np.int64) + bins, len(depth) - 1)], 0).astype(np.int64)
    soa = engine.EfSoA(cand_ctg_off=np.searchsorted(k, np.arange(n_contigs + 1)), read_tag=read_tag,
                       cand_pos=cl['cand_pos'], cand_svlen=cl['cand_span'], cand_svread=support,
                       cand_refread=np.maximum(d - support, 0), cand_gt_ok=np.ones(N, dtype=np.uint8),
                       cand_off=off, mark_read=marks['read'][cl['order']])
    rc, pred, ps = c_oracle.ef(soa, svlen_thres, suppread_thres)
    assert rc == 0
    return cl, pred, ps


@pytest.mark.parametrize('kind,seed', [('chr21', 21), ('genome_small', 3), ('config2', 1)])
def test_fused_pipeline_matches_oracle_composition(kind, seed):
    fused_case(H.case_contigs(kind, seed), seed, (True, False, True, False))


def test_fused_pipeline_config3_size():
    """2e7 raw marks over 24 contigs (BASELINE configs[2]'s size): the launch structure of large inputs, the generic
    tile-offset scan of the sort, scans with a spine launch -- at their real size."""
    fused_case(synth.bench_genome(20000000, 3), 3, (False, True))


def test_fused_pipeline_with_contigs_that_have_no_marks():
    """Contigs without a mark in front of, between and behind the others (and without depth bins): step E/F's contig
    offsets are written by cl_emit where the candidates' contig changes -- the empty ones get theirs from the candidate
    that opens the next non-empty contig, the trailing ones from the last candidate."""
    fused_case(H.case_contigs('genome_small', 7), 7, (False, True, False, True), contig_ids=lambda K: [2 + i + 2 * (i // 3) for i in range(K)], extra=3)


def fused_case(contigs, seed, waits, contig_ids=None, extra=0):
    soa = engine.soa_from_synth(contigs)
    marks = synth.raw_marks(contigs, seed, reads_of=soa)
    depth, depth_off = synth.depth_bins(contigs, 1000, seed)
    n_contigs = len(contigs)
    if contig_ids is not None:
        # the same marks on other contig numbers: ids[i] for contig i, `extra` empty contigs behind the last
        ids = np.asarray(contig_ids(n_contigs), dtype=np.int64)
        n_new = int(ids.max()) + 1 + extra
        marks = dict(marks)
        marks['contig'] = ids[marks['contig'].astype(np.int64)].astype(marks['contig'].dtype)
        nb = np.zeros(n_new, dtype=np.int64)
        nb[ids] = np.diff(depth_off.astype(np.int64))
        depth_off = np.concatenate([[0], np.cumsum(nb)]).astype(np.uint32)       # (ids ascend: the bins keep their order)
        n_contigs = n_new
    want_cl, want_pred, want_ps = oracle_fused(marks, soa.read_tag, depth, depth_off, 1000, 50, 2, n_contigs)
    ctx = _lib.Context(0)
    ds = DeviceSvim(marks, soa.read_tag, depth, depth_off, 1000, 50, 2)
    # with / without the host round trip; reruns reuse every workspace.  Small inputs sort 8-byte keys by default: every second
    # run takes the record sort as well (DUET_DBG_CLUSTER_RECSORT = 0x40000; the default from 1.25 M marks on).  The device-planned
    # ef_finalize takes two tiles per workgroup from a bound of 8 M candidates on: the last run of a small case forces that
    # (DUET_DBG_EF_FIN_TPB2 = 0x20), and the wave-cooperative walk of ef_classify on every candidate (0x80000)
    for it, wait in enumerate(waits):
        # (... and, every other time from the second on, the own-set finalize in the device-planned E/F tail: DUET_DBG_EF_OWN_ALL, round 6)
        ctx.set_debug((0x40000 if it % 2 else 0) | (0x20 | 0x80000 if it == len(waits) - 1 and len(waits) > 2 else 0) | (0x1000000 if it % 2 == 1 else 0))
        ds.run_fused(ctx, wait=wait)
        got = ds.fetch()
        assert ds.n_found == len(want_cl['cand_pos'])
        for f in ('order', 'cand_off', 'cand_contig', 'cand_type', 'cand_pos', 'cand_span'):
            assert np.array_equal(got[f], want_cl[f]), f
        assert np.array_equal(got['pred'], want_pred)
        assert np.array_equal(got['ps'], want_ps)
    assert int((want_pred != 0).sum()) > 0
    ctx.close()


@pytest.mark.parametrize('kind,seed', [('chr21', 3), ('genome_small', 5)])
def test_svim_mode_from_bams_matches_the_cpu_pipeline(kind, seed):
    """BAMs alone -> native signature extraction -> A0 -> E/F on the device, against oracle/svim_oracle.py's CPU
    pipeline (Python extraction, C cluster oracle, adapter, C E/F oracle)."""
    import shutil
    import tempfile
    from duet_amd import svim_mode
    from oracle import svim_oracle
    home = tempfile.mkdtemp(prefix='duet_svim_')
    try:
        synth.write_svim_workdir(home, H.case_contigs(kind, seed), seed)
        got = svim_mode.phase_from_bams(home, 50, 2, 2)
        want = svim_oracle.phase_workdir(home, got['chroms'], 50, 2)
        for f in ('cand_contig', 'cand_type', 'cand_pos', 'cand_span', 'support', 'pred', 'ps'):
            assert np.array_equal(got[f], want[f]), f
        assert int((got['pred'] != 0).sum()) > 0
        text = svim_mode.rows_text(home, got)
        assert text.count('\n') == int((got['pred'] != 0).sum())
        if kind == 'genome_small':                       # round 4: every SVTYPE the sign rule of sv_phasing_fn.py:225 distinguishes
            for t in ('<INS>', '<DEL>', '<DUP>', '<INV>'):
                assert text.count('SVTYPE=' + t) > 50, t
            assert 'SVLEN=-' in [l for l in text.split('\n') if '<INV>' in l][0] and 'SVLEN=-' not in [l for l in text.split('\n') if '<DUP>' in l][0]
    finally:
        shutil.rmtree(home, ignore_errors=True)


def test_cli_svim_gpu_caller_writes_phased_sv_vcf(tmp_path, monkeypatch):
    """`duet BAM REF OUT -b svim-gpu -c 0.9`: the stage table skips the external SV caller and ends in
    svim_mode.sv_phasing_from_bams; `-c` reaches the clustering (sv_calling.py:13-15, utils.py:27-28).  Parity unpinned
    for the clustering half (svim itself is external): checked against this repository's own CPU pipeline."""
    import sys
    from duet_amd import cli, svim_mode
    from oracle import svim_oracle
    home = str(tmp_path / 'out')
    synth.write_svim_workdir(home, H.case_contigs('genome_small', 5), 5)
    monkeypatch.setattr(sys, 'argv', ['duet', 'in.bam', 'ref.fa', home, '-b', 'svim-gpu', '-c', '0.5', '-s', '50', '-r', '2', '-t', '2'])
    from duet_amd.utils import parse_args
    a = parse_args(None)
    todo = cli.pipeline(a)
    assert [fn.__name__ for fn, _ in todo] == ['snp_calling', 'snp_phasing', 'sv_phasing_from_bams']
    fn, args = todo[-1]
    assert args[5] == 0.5
    fn(*args)
    text = open(home + '/phased_sv.vcf').read()
    head, rows = text[:text.index('#CHROM')], [l for l in text.split('\n') if l and not l.startswith('#')]
    assert head.startswith('##fileformat=VCFv4.2\n##source=Duet\n') and '##contig=<ID=chr1,length=249250621>' in head
    chroms = svim_mode.init_chrom_list(False, home)
    want = svim_oracle.phase_workdir(home, chroms, 50, 2, max_dist=0.5, min_sv_size=50)
    assert len(rows) == int((want['pred'] != 0).sum()) > 0
    assert text.endswith(svim_mode.rows_text(home, dict(want, chroms=chroms)))
    # and -c changes the clustering
    other = svim_oracle.phase_workdir(home, chroms, 50, 2, max_dist=0.05, min_sv_size=50)
    assert len(other['pred']) != len(want['pred'])
