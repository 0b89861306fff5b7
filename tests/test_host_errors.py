# coding=utf-8
"""Error behaviour of the host ingest mirrors upstream's (exception TYPE at the same input), and the native path
declines exactly those inputs so that the Python path gets to raise."""
import os

import pytest

from duet_amd import bamio, native
from duet_amd import sv_phasing_fn as F
from duet_amd.read_file import init_chrom_list

CHROMS = init_chrom_list(False, '')
GOOD = 'chr1\t100\tid\tN\t<DEL>\t.\tPASS\tPRECISE;SVTYPE=DEL;SVLEN=-80;END=180;RE=5;RNAMES=a,b;STRAND=+-\tGT:DR:DV:PL:GQ\t0/1:3:5:1,2,3:9'
SAM = ['a\t0\tchr1\t90\t60\t*\t*\t0\t0\t*\t*\tNM:i:1\tHP:i:1\tPC:i:100\tPS:i:50']


def workdir(tmp_path, lines):
    home = str(tmp_path)
    os.makedirs(home + '/sv_calling')
    os.makedirs(home + '/snp_phasing')
    with open(home + '/sv_calling/variants.vcf', 'w') as f:
        f.write('\n'.join(lines) + '\n')
    bamio.write_bam_from_sam_lines(home + '/snp_phasing/chr1.bam', [('chr1', 1000000)], SAM)
    return home


CASES = [
    ('svlen_key_inside_other_key', [GOOD.replace('SVLEN=-80', 'XSVLEN=5;SVLEN=-80')], ValueError),   # int('=5') upstream
    ('blank_line', [GOOD, '', GOOD], IndexError),                                                      # s[0] on []
    ('no_svtype', [GOOD.replace('SVTYPE=DEL;', '')], IndexError),                                      # [][0]
    ('support_not_a_number', [GOOD.replace('RE=5', 'RE=five')], ValueError),
    ('later_record_without_names', [GOOD, GOOD.replace('RNAMES=a,b;', '')], IndexError),
    ('pos_not_a_number', [GOOD.replace('\t100\t', '\t1e2\t')], ValueError),
    # an 11th whitespace token (second sample column, or a space inside INFO): upstream appends its derived columns
    # after the LAST token (read_file.py:37), generate_callinfo then iterates an int (sv_phasing_fn.py:47)
    ('eleven_columns', [GOOD + '\t0/1:1:2:3,4,5:6'], TypeError),
    ('space_inside_info', [GOOD.replace('PRECISE;', 'PRECISE; ')], IndexError),       # INFO is 'PRECISE;': no SVTYPE -> [][0] at parse
]


@pytest.mark.parametrize('name,lines,exc', CASES, ids=[c[0] for c in CASES])
def test_python_path_raises_like_upstream_and_native_declines(name, lines, exc, tmp_path):
    home = workdir(tmp_path, lines)
    ing = native.NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', CHROMS, 2)
    assert ing is not None and ing.handle is None, 'the native path must hand this input to the Python path'
    with pytest.raises(exc):
        F.generate_callinfo(home + '/sv_calling/variants.vcf', F.read_hap_bam(home + '/snp_phasing/', 2, False), False)


def test_bad_tag_values_are_rejected(tmp_path):
    from duet_amd import engine
    with pytest.raises(ValueError):
        engine.pack_tags([1], [-1], [5])
    with pytest.raises(ValueError):
        engine.pack_tags([1], [1], [0xFFFFFFFF])


def test_values_that_do_not_fit_the_abi_are_rejected_not_wrapped():
    """cand_off is built as int64 on the host; more than 2**32 - 1 marks must raise, not wrap, when it is narrowed."""
    import numpy as np
    from duet_amd import engine
    ok = dict(cand_ctg_off=[0, 1], read_tag=np.zeros(1, np.uint64), cand_pos=[5], cand_svlen=[60], cand_svread=[3],
              cand_refread=[1], cand_gt_ok=[1], cand_off=np.array([0, 1], dtype=np.int64), mark_read=[0])
    engine.EfSoA(**ok)
    for name, bad in (('cand_off', np.array([0, 1 << 32], dtype=np.int64)), ('cand_pos', np.array([1 << 33], dtype=np.int64)),
                      ('cand_svread', np.array([-1], dtype=np.int64)), ('mark_read', np.array([1 << 32], dtype=np.int64))):
        with pytest.raises(ValueError):
            engine.EfSoA(**dict(ok, **{name: bad}))
