# coding=utf-8
"""SVIM-mode signature extraction: the native BAM pass (libduet_ingest.so) against the Python statement of the rule
(oracle/svim_oracle.py) on synthetic haplotagged BAMs whose SV evidence sits in the CIGARs.  CPU only."""
import shutil
import tempfile

import numpy as np
import pytest

from duet_amd import synth
from duet_amd.native import NativeIngest
from duet_amd.read_file import init_chrom_list
from oracle import svim_oracle
from tests import helpers as H


@pytest.mark.parametrize('kind,seed', [('chr21', 3), ('fuzz', 2), ('genome_small', 5)])
def test_native_extraction_matches_the_rule(kind, seed):
    home = tempfile.mkdtemp(prefix='duet_svim_')
    try:
        contigs = H.case_contigs(kind, seed)
        synth.write_svim_workdir(home, contigs, seed)
        chroms = init_chrom_list(False, home)
        want = svim_oracle.extract_workdir(home, chroms)
        ing, got = NativeIngest.extract(home + '/snp_phasing/', chroms, thread=2)
        assert ing is not None, got
        for f in ('contig', 'type', 'pos', 'span'):
            assert np.array_equal(got[f], want[f]), f
        assert len(got['pos']) > 0
        if kind == 'genome_small':                       # every SV type the sign rule distinguishes shows up (round 4: DUP / INV)
            assert np.bincount(got['type'], minlength=4).min() > 100
        # the marks' reads resolve to the same tags
        tags = [None if r == 0xFFFFFFFF else int(got['read_tag'][r]) for r in got['read']]
        assert tags == [None if t is None else svim_oracle.pack_tag(t) for t in want['tag']]
        for k in range(len(chroms)):
            d = got['depth'][got['depth_off'][k]:got['depth_off'][k + 1]]
            assert d.tolist() == want['depth'][k], k
        ing.close()
    finally:
        shutil.rmtree(home, ignore_errors=True)


def test_filters(tmp_path):
    """Unmapped / secondary / low-MAPQ alignments and short indels leave no marks."""
    from duet_amd import bamio
    lines = ['a\t0\tchr1\t1000\t60\t100M50I100M\t*\t0\t0\t*\t*\tNM:i:1\tHP:i:1\tPC:i:10\tPS:i:7',
             'b\t256\tchr1\t1000\t60\t100M50I100M\t*\t0\t0\t*\t*\tNM:i:1',
             'c\t4\tchr1\t1000\t60\t100M50I100M\t*\t0\t0\t*\t*\tNM:i:1',
             'd\t0\tchr1\t1000\t19\t100M50D100M\t*\t0\t0\t*\t*\tNM:i:1',
             'e\t2048\tchr1\t5000\t20\t10S100M39I5M40D7N3=2X41I9H\t*\t0\t0\t*\t*\tNM:i:1']
    d = tmp_path / 'snp_phasing'
    d.mkdir()
    stem = str(d / 'chr1.bam')
    with open(stem + '.sam', 'w') as f:
        f.write(''.join(l + '\n' for l in lines))
    bamio.write_bam_from_sam_lines(stem, [('chr1', 249250621)], lines)
    chroms = init_chrom_list(False, str(tmp_path))
    want = svim_oracle.extract_workdir(str(tmp_path), chroms)
    ing, got = NativeIngest.extract(str(d) + '/', chroms)
    assert ing is not None, got
    assert got['pos'].tolist() == want['pos'].tolist() == [1100, 5105, 5157]
    assert got['type'].tolist() == [1, 0, 1] and got['span'].tolist() == [50, 40, 41]
    assert got['read'].tolist() == [0, 0xFFFFFFFF, 0xFFFFFFFF]
    ing.close()


def test_header_of_the_svim_gpu_mode_comes_from_the_bam_reference_lists(tmp_path):
    from duet_amd import bamio, svim_mode
    d = tmp_path / 'snp_phasing'
    d.mkdir()
    line = ['a\t0\t%s\t1000\t60\t100M\t*\t0\t0\t*\t*\tNM:i:1']
    bamio.write_bam_from_sam_lines(str(d / 'chr2.bam'), [('chr1', 249250621), ('chr2', 243199373)], [line[0] % 'chr2'])
    bamio.write_bam_from_sam_lines(str(d / 'X.bam'), [('X', 155270560)], [line[0] % 'X'])
    assert bamio.read_refs(str(d / 'chr2.bam')) == [('chr1', 249250621), ('chr2', 243199373)]
    head = svim_mode.header_text(str(tmp_path), init_chrom_list(False, str(tmp_path)))
    assert head.count('##contig=') == 2
    assert head.index('##contig=<ID=chr2,length=243199373>') < head.index('##contig=<ID=X,length=155270560>')
    assert head.endswith('FORMAT\tVALUE\n')


def test_split_read_duplications_and_inversions(tmp_path):
    """Hand-made split reads: a read that goes back on the reference (tandem duplication), reads that turn round at a
    breakpoint (inversion; right ends / left ends meeting), and pairs that are neither (overlap on the read, too short, too long)."""
    from duet_amd import bamio
    L = ['dupf\t0\tchr1\t10501\t60\t500M400S\t*\t0\t0\t*\t*\tNM:i:1',          # [10500, 11000) then back to 10200: DUP [10200, 11000)
         'dupf\t2048\tchr1\t10201\t60\t500H400M\t*\t0\t0\t*\t*\tNM:i:1',
         'dupr\t16\tchr1\t20001\t60\t400S500M\t*\t0\t0\t*\t*\tNM:i:1',          # reverse: a = [20000, 20500) (qs 0), b = [20300, 20700)
         'dupr\t2064\tchr1\t20301\t60\t400M500H\t*\t0\t0\t*\t*\tNM:i:1',
         'invr\t0\tchr1\t30001\t60\t500M400S\t*\t0\t0\t*\t*\tNM:i:1',           # forward up to 30500, then reverse with right end 33500
         'invr\t2064\tchr1\t33101\t60\t400M500H\t*\t0\t0\t*\t*\tNM:i:1',
         'invl\t16\tchr1\t43001\t60\t400S500M\t*\t0\t0\t*\t*\tNM:i:1',          # reverse starting at 43000, then forward from 40000
         'invl\t2048\tchr1\t40001\t60\t500H400M\t*\t0\t0\t*\t*\tNM:i:1',
         'shrt\t0\tchr1\t50001\t60\t500M400S\t*\t0\t0\t*\t*\tNM:i:1',           # goes back by 30 bases only: nothing
         'shrt\t2048\tchr1\t50471\t60\t500H400M\t*\t0\t0\t*\t*\tNM:i:1',
         'ovlp\t0\tchr1\t60001\t60\t500M400S\t*\t0\t0\t*\t*\tNM:i:1',           # the segments overlap on the READ by 100: nothing
         'ovlp\t2064\tchr1\t63001\t60\t500M400H\t*\t0\t0\t*\t*\tNM:i:1']
    d = tmp_path / 'snp_phasing'
    d.mkdir()
    stem = str(d / 'chr1.bam')
    with open(stem + '.sam', 'w') as f:
        f.write(''.join(l + '\n' for l in L))
    bamio.write_bam_from_sam_lines(stem, [('chr1', 249250621)], L)
    chroms = init_chrom_list(False, str(tmp_path))
    want = svim_oracle.extract_workdir(str(tmp_path), chroms)
    ing, got = NativeIngest.extract(str(d) + '/', chroms)
    assert ing is not None, got
    for f in ('type', 'pos', 'span'):
        assert got[f].tolist() == want[f].tolist(), f
    assert got['type'].tolist() == [3, 3, 2, 2]
    assert got['pos'].tolist() == [10201, 20001, 30501, 40001] and got['span'].tolist() == [800, 700, 3000, 3000]
    ing.close()
