# coding=utf-8
"""libduet_ingest.so (native VCF/BAM ingest + row emission) against the Python host path and the goldens.
The C oracle stands in for the GPU here (tests only)."""
import os
import shutil

import numpy as np
import pytest

from duet_amd import engine, native
from duet_amd import sv_phasing_fn as F
from duet_amd.read_file import init_chrom_list
from oracle import c_oracle
from tests import helpers as H
from tests.test_c_oracle import materialise_bams

CHROMS = init_chrom_list(False, '')


def native_text(home, svlen_thres, suppread_thres):
    ing = native.NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', CHROMS, 4)
    assert ing is not None and ing.handle, getattr(ing, 'why', 'library missing')
    rc, pred, ps = c_oracle.ef(ing.soa, svlen_thres, suppread_thres)
    assert rc == 0
    text = ing.emit(pred, ps, False).decode('ascii')
    return text, ing


def test_library_exports():
    import ctypes
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in native.EXPORTS:
        assert hasattr(lib, name)


@pytest.mark.parametrize('name,src,params', H.full_cases(), ids=[c[0] for c in H.full_cases()])
def test_full_cases_bytes_and_arrays(name, src, params, tmp_path):
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    with open(os.path.join(src, 'phased_sv.vcf')) as f:
        want = f.read()
    got, ing = native_text(home, params['svlen_thres'], params['suppread_thres'])
    assert got == want
    # field-for-field identical to the Python ingest (same read numbering: order of first appearance)
    tab, soa = F.generate_callinfo(home + '/sv_calling/variants.vcf', F.read_hap_bam(home + '/snp_phasing/', 4, False), False)
    for field, _ in engine.EfSoA.FIELDS:
        assert np.array_equal(getattr(ing.soa, field), getattr(soa, field)), field
    ing.close()


def test_seeded_cases(tmp_path):
    n = 0
    for p in H.seeded_plan():
        if p['kind'] == 'config2' or (p['kind'] == 'fuzz' and p['seed'] % 3):
            continue
        home = str(tmp_path / ('%s_%d_%s' % (p['kind'], p['seed'], p['dialect'])))
        H.build_case(home, p['kind'], p['seed'], p['dialect'], write_sam=False)
        got, ing = native_text(home, p['svlen_thres'], p['suppread_thres'])
        ing.close()
        assert H.sha256_bytes(got.encode()) == p['output_sha256'], p
        shutil.rmtree(home)
        n += 1
    assert n >= 60


def _tiny_case(tmp_path, vcf_lines, sam_lines):
    from duet_amd import bamio
    home = str(tmp_path / 'w')
    os.makedirs(home + '/sv_calling')
    os.makedirs(home + '/snp_phasing')
    with open(home + '/sv_calling/variants.vcf', 'w') as f:
        f.write('\n'.join(vcf_lines) + '\n')
    bamio.write_bam_from_sam_lines(home + '/snp_phasing/chr1.bam', [('chr1', 1000000)], sam_lines)
    return home


REC = 'chr1\t100\tid\tN\t<DEL>\t.\tPASS\tPRECISE;SVTYPE=DEL;SVLEN=-80;END=180;RE=5;RNAMES=a,b;STRAND=+-\tGT:DR:DV:PL:GQ\t0/1:3:5:1,2,3:9'
SAM = ['a\t0\tchr1\t90\t60\t*\t*\t0\t0\t*\t*\tNM:i:1\tHP:i:1\tPC:i:100\tPS:i:50']


def test_declines_what_it_cannot_vouch_for(tmp_path):
    """Blank lines, non-ASCII bytes, malformed numbers -> the native path says UNSUPPORTED (Python takes over)."""
    for i, bad in enumerate(([REC, '', REC], [REC.replace('id', 'ié')], [REC.replace('RE=5', 'RE=five')],
                             [REC.replace('\t100\t', '\t1e2\t')], [REC.replace('GT:DR:DV:PL:GQ\t0/1:3:5:1,2,3:9', 'GT\t0/1')])):
        home = _tiny_case(tmp_path / str(i), bad, SAM)
        ing = native.NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', CHROMS, 1)
        assert ing is not None and ing.handle is None and ing.why, bad


def test_two_call_parse_reports_through_finish(tmp_path):
    """duet_ingest_parse_vcf_begin / _finish (what NativeIngest.load runs beside the BAM loop): _finish without _begin is
    invalid, a failed first half is reported by _finish with its error text, and a second _finish is invalid again."""
    import ctypes
    lib = native.load()
    names = (ctypes.c_char_p * 1)(b'1')
    h = lib.duet_ingest_create(1, names)
    try:
        assert lib.duet_ingest_parse_vcf_finish(h) != native.OK
        missing = str(tmp_path / 'no_such.vcf').encode()
        assert lib.duet_ingest_parse_vcf_begin(h, missing, 2) != native.OK
        assert lib.duet_ingest_parse_vcf_finish(h) != native.OK
        assert b'cannot read' in lib.duet_ingest_error(h)
        assert lib.duet_ingest_parse_vcf_finish(h) != native.OK
        vcf = tmp_path / 'empty.vcf'
        vcf.write_text('##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n')
        assert lib.duet_ingest_parse_vcf_begin(h, str(vcf).encode(), 2) == native.OK
        assert lib.duet_ingest_parse_vcf_finish(h) == native.OK
    finally:
        lib.duet_ingest_destroy(h)


def test_many_records_parallel_bam_path(tmp_path):
    """A BAM of more than 4096 alignments read with several threads takes the batch path (records examined in parallel,
    applied in file order): same arrays as the one-thread loop and as the Python ingest, and a declining record still declines."""
    from duet_amd import bamio, synth
    home = str(tmp_path / 'w')
    c = synth.bench_contig('1', 6000, 400, 5, length=3000000)
    synth.write_workdir(home, [c], dialect='cutesv', seed=1, write_sam=True)
    vcf, sams = home + '/sv_calling/variants.vcf', home + '/snp_phasing/'
    one = native.NativeIngest.load(vcf, sams, CHROMS, 1)
    four = native.NativeIngest.load(vcf, sams, CHROMS, 4)
    assert one.handle is not None and four.handle is not None
    assert one.soa.n_reads > 4096
    tab, soa = F.generate_callinfo(vcf, F.read_hap_bam(sams, 4, False), False)
    for field, _ in engine.EfSoA.FIELDS:
        assert np.array_equal(getattr(one.soa, field), getattr(four.soa, field)), field
        assert np.array_equal(getattr(four.soa, field), getattr(soa, field)), field
    one.close()
    four.close()
    # the same alignments with a read whose PC is negative somewhere in the middle: declined by both paths
    lines = synth.sam_lines(c)
    k = next(i for i, l in enumerate(lines) if i > 5000 and 'PC:i:' in l)
    import re
    lines[k] = re.sub(r'PC:i:\d+', 'PC:i:-7', lines[k])
    bamio.write_bam_from_sam_lines(sams + 'chr1.bam', [('chr1', 3000000)], lines)
    for t in (1, 4):
        ing = native.NativeIngest.load(vcf, sams, CHROMS, t)
        assert ing is not None and ing.handle is None and 'PC/PS out of range' in ing.why, (t, ing.why)


def test_whole_contigs_dealt_to_the_workers(tmp_path):
    """Round 6, duet_ingest_add_bams: several contigs' BAMs in one call, whole contigs dealt to the workers (largest file first; each
    contig alone takes the batched record path of the test above) -- the same arrays as contig after contig with one thread and as the
    Python ingest (sv_phasing_fn.py:15-29 builds one dict per contig: nothing crosses contigs); a contig whose BAM declines makes the
    call decline with THAT contig's reason, whatever the others did meanwhile."""
    import re
    from duet_amd import bamio, synth
    home = str(tmp_path / 'w')
    contigs = [synth.bench_contig(l, 5000 + 900 * i, 300 + 40 * i, 11 + i, length=3000000) for i, l in enumerate(('1', '2', '3', '7', 'X'))]
    synth.write_workdir(home, contigs, dialect='cutesv', seed=2, write_sam=True)
    vcf, sams = home + '/sv_calling/variants.vcf', home + '/snp_phasing/'
    one = native.NativeIngest.load(vcf, sams, CHROMS, 1)
    many = native.NativeIngest.load(vcf, sams, CHROMS, 8)
    few = native.NativeIngest.load(vcf, sams, CHROMS, 3)           # (fewer workers than contigs)
    assert one.handle is not None and many.handle is not None and few.handle is not None
    tab, soa = F.generate_callinfo(vcf, F.read_hap_bam(sams, 4, False), False)
    for field, _ in engine.EfSoA.FIELDS:
        assert np.array_equal(getattr(one.soa, field), getattr(many.soa, field)), field
        assert np.array_equal(getattr(one.soa, field), getattr(few.soa, field)), field
        assert np.array_equal(getattr(many.soa, field), getattr(soa, field)), field
    assert one.log_lines(CHROMS) == many.log_lines(CHROMS)
    for ing in (one, many, few):
        ing.close()
    lines = synth.sam_lines(contigs[3])
    k = next(i for i, l in enumerate(lines) if i > 3000 and 'PC:i:' in l)
    lines[k] = re.sub(r'PC:i:\d+', 'PC:i:-7', lines[k])
    bamio.write_bam_from_sam_lines(sams + 'chr7.bam', [('chr7', 3000000)], lines)
    for t in (1, 3, 8):
        ing = native.NativeIngest.load(vcf, sams, CHROMS, t)
        assert ing is not None and ing.handle is None and 'PC/PS out of range' in ing.why, (t, ing.why)


def test_python_int_forms_accepted(tmp_path):
    home = _tiny_case(tmp_path, [REC.replace('\t100\t', '\t+1_00\t').replace('RE=5', 'RE=0_5')], SAM)
    ing = native.NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', CHROMS, 1)
    assert ing.handle and int(ing.soa.cand_pos[0]) == 100 and int(ing.soa.cand_svread[0]) == 5
    tab, soa = F.generate_callinfo(home + '/sv_calling/variants.vcf', F.read_hap_bam(home + '/snp_phasing/', 1, False), False)
    assert int(soa.cand_pos[0]) == 100 and int(soa.cand_svread[0]) == 5
    ing.close()


def rows_from_pool(ing, rows, pred, ps):
    """The data rows rebuilt in Python from what duet_ingest_get_rows hands to the device (text pool, CHROM ranks,
    sign flags): the same ordering rule and layout as duet_rows_run_device, so the arrays are checked without a GPU."""
    soa = ing.soa
    pool = rows['pool'].tobytes()
    off = rows['str_off']
    keep = np.nonzero(pred)[0]
    ctg = np.searchsorted(soa.cand_ctg_off, keep, side='right') - 1
    cls = []
    for c in keep:
        m = soa.mark_read[soa.cand_off[c]:soa.cand_off[c + 1]]
        m = m[m != 0xFFFFFFFF]
        cls.append(min(len(set((soa.read_tag[m] & 0xFFFFFFFF).tolist())), 2))
    order = sorted(range(len(keep)), key=lambda i: (int(rows['chrom_rank'][keep[i]]), int(soa.cand_pos[keep[i]]), int(ctg[i]),
                                                    cls[i], int(keep[i])))
    hp = {1: '1|0', 2: '0|1', 3: '1|1'}
    out = []
    for n, i in enumerate(order):
        c = int(keep[i])
        t = [pool[off[4 * c + f]:off[4 * c + f + 1]].decode() for f in range(4)]
        mag = int(soa.cand_svlen[c])
        signed = mag if (rows['plus'][c] or mag == 0) else -mag
        out.append('%s\t%d\tDuet.%d\t%s\t%s\t.\tPASS\tSVLEN=%d;SVTYPE=<%s>\tHP:PS\t%s:%d\n' % (
            t[0], int(soa.cand_pos[c]), n + 1, t[1], t[2], signed, t[3], hp[int(pred[c])], int(ps[c])))
    return ''.join(out)


@pytest.mark.parametrize('name,src,params', H.full_cases()[:4], ids=[c[0] for c in H.full_cases()[:4]])
def test_rows_pool_and_header(name, src, params, tmp_path):
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    with open(os.path.join(src, 'phased_sv.vcf')) as f:
        want = f.read()
    ing = native.NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', init_chrom_list(False, home), 2)
    assert ing is not None and ing.handle is not None
    rc, pred, ps = c_oracle.ef(ing.soa, params['svlen_thres'], params['suppread_thres'])
    assert rc == 0
    rows = ing.rows()
    assert rows is not None and rows['n_chrom_texts'] >= 1
    assert ing.header(False).decode() + rows_from_pool(ing, rows, pred, ps) == want
    ing.close()


@pytest.mark.parametrize('name,src,params', H.full_cases()[:6], ids=[c[0] for c in H.full_cases()[:6]])
def test_sharded_entry_points(name, src, params, tmp_path):
    """What the N-GPU path builds on (duet_amd/multi.py): the pre-count agrees with the parsed call set, an ingest that owns
    a subset of the contigs holds exactly their candidates, and the per-rank text blocks, numbered from the counts of all
    ranks and laid end to end in the order of their CHROM texts, are the single-process file body."""
    home = str(tmp_path / name)
    shutil.copytree(src, home)
    materialise_bams(home)
    vcf, sam = home + '/sv_calling/variants.vcf', home + '/snp_phasing/'
    whole = native.NativeIngest.load(vcf, sam, CHROMS, 2)
    assert whole is not None and whole.handle
    K = whole.soa.n_contigs
    per_contig = np.diff(np.asarray(whole.soa.cand_ctg_off, dtype=np.int64))
    n_rec, n_bytes = native.NativeIngest.precount(vcf, CHROMS)
    assert np.array_equal(np.asarray(n_rec[:K], dtype=np.int64), per_contig)
    assert all((b > 0) == (r > 0) for r, b in zip(n_rec[:K], n_bytes[:K]))
    rc, pred, ps = c_oracle.ef(whole.soa, params['svlen_thres'], params['suppread_thres'])
    assert rc == 0
    want = b''.join(l for l in whole.emit(pred, ps, False).splitlines(keepends=True) if not l.startswith(b'#'))     # the rows
    # two "ranks": the contigs dealt out alternately among those that have records
    have = [k for k in range(K) if per_contig[k]]
    owned = [have[0::2], have[1::2]]
    parts, kept = [], []
    for mine in owned:
        ing = native.NativeIngest.load(vcf, sam, CHROMS, 2, owned=mine)
        assert ing is not None and ing.handle
        mine_cnt = np.diff(np.asarray(ing.soa.cand_ctg_off, dtype=np.int64))
        assert all(mine_cnt[k] == (per_contig[k] if k in mine else 0) for k in range(K))
        rc, p1, s1 = c_oracle.ef(ing.soa, params['svlen_thres'], params['suppread_thres'])
        assert rc == 0
        parts.append((ing, p1, s1))
        kept.append(ing.count_kept(p1))
    total = kept[0] + kept[1]
    assert np.array_equal(total, whole.count_kept(pred))
    # rows are numbered in the order of the CHROM texts (byte order), a slot's rows consecutively
    from duet_amd import multi
    order = multi.text_order(CHROMS)
    base = np.ones(2 * K, dtype=np.int64)
    run = 1                                              # (the ID column counts from 1: write_file.py:9-14)
    for slot in order:
        base[slot] = run
        run += int(total[slot])
    blocks = {}
    for ing, p1, s1 in parts:
        text, off, ln = ing.emit_blocks(p1, s1, base)
        for slot in range(2 * K):
            if ln[slot]:
                assert slot not in blocks
                blocks[slot] = text[int(off[slot]):int(off[slot]) + int(ln[slot])]
        ing.close()
    body = b''.join(blocks.get(slot, b'') for slot in order)
    assert body == want
    whole.close()
