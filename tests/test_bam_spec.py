# coding=utf-8
"""BAM files assembled BY HAND from the SAM/BAM specification (SAMv1 section 4: BGZF members, the header block, the
alignment record's field table, the typed auxiliary fields, and section 4.2.2's CG:B:I long-CIGAR convention), with the
text `samtools view` prints for them written out beside the bytes.  None of the bytes below come from the repository's
own writer (duet_amd/bamio.py write_bam_from_sam_lines, which every other test uses), so this pins both readers --
bamio.iter_bam_records/tail_tokens (Python host path) and libduet_ingest.so (native host path) -- to the format itself:
  * every aux value type, incl. all seven B-array subtypes and a float printed with %g,
  * tags that do NOT end in HP, PC, PS order (the reference only looks at the last three tokens, sv_phasing_fn.py:27-29),
  * a read with more CIGAR operations than n_cigar_op can hold: placeholder CIGAR + CG:B:I, which htslib folds back
    (and drops from the line) on reading,
  * records that straddle BGZF members, an empty member and a stored (uncompressed) member in mid-stream,
  * a header with several @SQ lines.
CPU only."""
import os
import struct
import zlib

import numpy as np

from duet_amd import bamio
from duet_amd import sv_phasing_fn as F
from duet_amd.native import NativeIngest
from duet_amd.read_file import init_chrom_list
from oracle import svim_oracle

REFS = [('chr1', 249250621), ('chr2', 243199373), ('chrUn_gl000220', 161802)]
SEQ_CODE = '=ACMGRSVTWYHKDBN'                      # SAMv1 4.2: 4-bit base codes
CIG_CODE = 'MIDNSHP=X'                            # SAMv1 4.2: op codes 0..8


def aux(tag, typ, payload):
    return tag.encode() + typ.encode() + payload


def b_array(tag, sub, fmt, values):
    return aux(tag, 'B', sub.encode() + struct.pack('<I', len(values)) + b''.join(struct.pack('<' + fmt, v) for v in values))


def record(qname, flag, ref_id, pos1, mapq, cigar_ops, seq, qual, aux_bytes, next_ref=-1, next_pos1=0, tlen=0, bin_=0):
    """One alignment record, field by field as the table in SAMv1 section 4.2 lists them."""
    name = qname.encode() + b'\0'
    cig = b''.join(struct.pack('<I', (n << 4) | CIG_CODE.index(op)) for n, op in cigar_ops)
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= SEQ_CODE.index(ch) << (4 if i % 2 == 0 else 0)
    ql = bytes(ord(c) - 33 for c in qual) if qual != '*' else b'\xff' * l_seq
    body = struct.pack('<iiBBHHHIiii', ref_id, pos1 - 1, len(name), mapq, bin_, len(cigar_ops), flag, l_seq, next_ref,
                       next_pos1 - 1, tlen) + name + cig + bytes(packed) + ql + aux_bytes
    return struct.pack('<I', len(body)) + body


def bgzf_member(data, level=6):
    """One BGZF block: a gzip member (RFC 1952) with FEXTRA = the 'BC' subfield holding the total block size - 1."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    total = 12 + 6 + len(body) + 8
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255]) + struct.pack('<H', 6) + b'BC' + struct.pack('<HH', 2, total - 1) +
            body + struct.pack('<II', zlib.crc32(data) & 0xFFFFFFFF, len(data)))


EOF_MEMBER = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')     # SAMv1 4.1.2


def long_cigar_ops():
    """66,001 operations: (10M 45I 10M 52D) x 16,500, then 10M -- more than the 65,535 n_cigar_op can express."""
    ops = []
    for _ in range(16500):
        ops += [(10, 'M'), (45, 'I'), (10, 'M'), (52, 'D')]
    ops.append((10, 'M'))
    return ops


def build():
    """-> (bytes of the BAM, expected `samtools view` lines, expected tag dict, long-CIGAR ops)"""
    text = ('@HD\tVN:1.6\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in REFS) +
            '@PG\tID:whatshap\tPN:whatshap\tVN:1.0\n').encode()
    head = b'BAM\1' + struct.pack('<I', len(text)) + text + struct.pack('<I', len(REFS))
    for name, ln in REFS:
        head += struct.pack('<I', len(name) + 1) + name.encode() + b'\0' + struct.pack('<I', ln)

    hp = lambda v: aux('HP', 'C', struct.pack('<B', v))
    recs, lines = [], []
    # 1. the usual layout: ..., HP, PC, PS last (whatshap haplotag appends them in this order)
    recs.append(record('r1', 0, 0, 1000, 60, [(8, 'M')], 'ACGTACGT', 'IIIIHHHH',
                       aux('NM', 'C', b'\x02') + hp(1) + aux('PC', 'S', struct.pack('<H', 300)) + aux('PS', 'I', struct.pack('<I', 100000))))
    lines.append('r1\t0\tchr1\t1000\t60\t8M\t*\t0\t0\tACGTACGT\tIIIIHHHH\tNM:i:2\tHP:i:1\tPC:i:300\tPS:i:100000')
    # 2. the three tags present but in another order: token [-2] is HP -> the reference does not see a tagged read
    recs.append(record('r2', 16, 0, 2000, 60, [(4, 'M')], 'ACGT', '*',
                       aux('PS', 'I', struct.pack('<I', 100000)) + hp(2) + aux('PC', 'C', b'\x09')))
    lines.append('r2\t16\tchr1\t2000\t60\t4M\t*\t0\t0\tACGT\t*\tPS:i:100000\tHP:i:2\tPC:i:9')
    # 3. a tag AFTER PS: token [-2] is PS
    recs.append(record('r3', 0, 0, 3000, 37, [(4, 'M')], 'NNNN', '!!!!',
                       hp(1) + aux('PC', 'C', b'\x07') + aux('PS', 'S', struct.pack('<H', 777)) + aux('XZ', 'Z', b'tail\0')))
    lines.append('r3\t0\tchr1\t3000\t37\t4M\t*\t0\t0\tNNNN\t!!!!\tHP:i:1\tPC:i:7\tPS:i:777\tXZ:Z:tail')
    # 4. every aux value type in front of HP, PC, PS (PC as a signed 32-bit, PS as an unsigned one above 2^31)
    every = (aux('XA', 'A', b'Q') + aux('Xc', 'c', struct.pack('<b', -5)) + aux('XC', 'C', struct.pack('<B', 250)) +
             aux('Xs', 's', struct.pack('<h', -30000)) + aux('XS', 'S', struct.pack('<H', 60000)) +
             aux('Xi', 'i', struct.pack('<i', -2000000000)) + aux('XI', 'I', struct.pack('<I', 4000000000)) +
             aux('Xf', 'f', struct.pack('<f', 0.25)) + aux('XY', 'Z', b'a:b;c=d\0') + aux('XH', 'H', b'1AE301\0') +
             b_array('Ba', 'c', 'b', [-1, 2]) + b_array('Bb', 'C', 'B', [0, 255]) + b_array('Bc', 's', 'h', [-300, 300]) +
             b_array('Bd', 'S', 'H', [65535]) + b_array('Be', 'i', 'i', [-70000, 70000]) + b_array('Bf', 'I', 'I', [4000000000]) +
             b_array('Bg', 'f', 'f', [1.5, -2.0, 0.25]) + b_array('Bh', 'C', 'B', []))
    recs.append(record('r4', 2048, 1, 500, 20, [(2, 'S'), (3, 'M'), (1, 'I'), (2, 'M'), (4, 'D'), (1, '='), (1, 'X'), (3, 'H')],
                       'ACGTRYKMN', 'ABCDEFGHI',
                       every + hp(2) + aux('PC', 'i', struct.pack('<i', 8100)) + aux('PS', 'I', struct.pack('<I', 4000000000)),
                       next_ref=0, next_pos1=1000, tlen=-17))
    lines.append('r4\t2048\tchr2\t500\t20\t2S3M1I2M4D1=1X3H\tchr1\t1000\t-17\tACGTRYKMN\tABCDEFGHI\tXA:A:Q\tXc:i:-5\tXC:i:250\tXs:i:-30000\t'
                 'XS:i:60000\tXi:i:-2000000000\tXI:i:4000000000\tXf:f:0.25\tXY:Z:a:b;c=d\tXH:H:1AE301\tBa:B:c,-1,2\tBb:B:C,0,255\t'
                 'Bc:B:s,-300,300\tBd:B:S,65535\tBe:B:i,-70000,70000\tBf:B:I,4000000000\tBg:B:f,1.5,-2,0.25\tBh:B:C\t'
                 'HP:i:2\tPC:i:8100\tPS:i:4000000000')
    # 5. long CIGAR, CG in front of the haplotype tags.  Stored: <l_seq>S<ref_len>N.  ref_len = 16500 * (10+10+52) + 10
    ops = long_cigar_ops()
    l_seq = sum(n for n, op in ops if op in 'MIS=X')
    ref_len = sum(n for n, op in ops if op in 'MDN=X')
    seq = 'A' * 16                                         # (a short stand-in read: l_seq is what the placeholder must match)
    cg = b_array('CG', 'I', 'I', [(n << 4) | CIG_CODE.index(op) for n, op in ops])
    real = ''.join('%d%s' % o for o in ops)
    recs.append(record('r5', 0, 0, 50000, 60, [(16, 'S'), (ref_len, 'N')], seq, '*',
                       aux('NM', 'C', b'\x00') + cg + hp(1) + aux('PC', 'C', b'\x2a') + aux('PS', 'S', struct.pack('<H', 50001))))
    lines.append('r5\t0\tchr1\t50000\t60\t%s\t*\t0\t0\t%s\t*\tNM:i:0\tHP:i:1\tPC:i:42\tPS:i:50001' % (real, seq))
    # 6. the same with CG as the LAST aux field: folded away, the line still ends in HP, PC, PS
    recs.append(record('r6', 0, 0, 60000, 60, [(16, 'S'), (ref_len, 'N')], seq, '*',
                       hp(2) + aux('PC', 'C', b'\x2b') + aux('PS', 'S', struct.pack('<H', 50001)) + cg))
    lines.append('r6\t0\tchr1\t60000\t60\t%s\t*\t0\t0\t%s\t*\tHP:i:2\tPC:i:43\tPS:i:50001' % (real, seq))
    # 7. a CG tag that must NOT be folded: the first stored operation does not soft-clip the whole read
    recs.append(record('r7', 0, 0, 70000, 60, [(15, 'S'), (100, 'N')], seq, '*',
                       b_array('CG', 'I', 'I', [(100 << 4) | 0, (60 << 4) | 1]) + hp(1) + aux('PC', 'C', b'\x2c') +
                       aux('PS', 'S', struct.pack('<H', 50001))))
    lines.append('r7\t0\tchr1\t70000\t60\t15S100N\t*\t0\t0\t%s\t*\tCG:B:I,1600,961\tHP:i:1\tPC:i:44\tPS:i:50001' % seq)
    # 8. unmapped, no CIGAR, no sequence, fewer than three aux fields
    recs.append(record('r8', 4, -1, 0, 0, [], '', '*', aux('RG', 'Z', b'g\0')))
    lines.append('r8\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\tRG:Z:g')
    # 9. r1 again (a supplementary alignment, --tag-supplementary): the later line wins (sv_phasing_fn.py:29)
    recs.append(record('r1', 2048, 0, 90000, 60, [(8, 'M')], 'ACGTACGT', '*',
                       hp(2) + aux('PC', 'S', struct.pack('<H', 8101)) + aux('PS', 'I', struct.pack('<I', 90001))))
    lines.append('r1\t2048\tchr1\t90000\t60\t8M\t*\t0\t0\tACGTACGT\t*\tHP:i:2\tPC:i:8101\tPS:i:90001')

    payload = head + b''.join(recs)
    # BGZF members cut at arbitrary byte offsets (inside the header, inside records, inside the CG arrays), one empty
    # member and one stored (level 0) member in mid-stream, then the end-of-file marker
    cuts = [0, 7, 151, 152, 400, 401, 1000, 70000, 70001, 200000, 333333, len(payload) - 5, len(payload)]
    cuts = sorted(set(c for c in cuts if c <= len(payload)))
    blob = b''
    for i in range(len(cuts) - 1):
        piece = payload[cuts[i]:cuts[i + 1]]
        for j in range(0, len(piece), 60000):
            blob += bgzf_member(piece[j:j + 60000], level=0 if i == 6 else 6)
        if i == 3:
            blob += bgzf_member(b'')
    blob += EOF_MEMBER
    tags = {'r1': (2, 8101, 90001), 'r4': (2, 8100, 4000000000), 'r5': (1, 42, 50001), 'r6': (2, 43, 50001), 'r7': (1, 44, 50001)}
    return blob, lines, tags, ops


def test_python_reader_prints_the_spec_text(tmp_path):
    blob, lines, tags, _ = build()
    path = str(tmp_path / 'chr1.bam')
    with open(path, 'wb') as f:
        f.write(blob)
    got = []
    for name, mandatory, aux_fields in bamio.iter_bam_records(path):
        got.append('\t'.join(mandatory() + aux_fields))
        assert bamio.tail_tokens(mandatory, aux_fields) == got[-1].split()[-3:]
    assert got == lines
    # the reference's rule on those lines (sv_phasing_fn.py:26-29)
    want = {}
    for ln in lines:
        s = ln.split()
        if 'PC:i:' in s[-2]:
            want[s[0]] = (int(s[-3][5:]), int(s[-2][5:]), int(s[-1][5:]))
    assert want == tags


def _workdir(tmp_path, blob, lines):
    home = str(tmp_path)
    os.makedirs(home + '/sv_calling')
    os.makedirs(home + '/snp_phasing')
    with open(home + '/snp_phasing/chr1.bam', 'wb') as f:
        f.write(blob)
    with open(home + '/snp_phasing/chr1.bam.sam', 'w') as f:
        f.write(''.join(l + '\n' for l in lines))
    names = 'r1,r2,r3,r4,r5,r6,r7,r8,zz'
    with open(home + '/sv_calling/variants.vcf', 'w') as f:
        f.write('##contig=<ID=chr1,length=249250621>\n')
        f.write('chr1\t1000\tid\tN\t<DEL>\t.\tPASS\tPRECISE;SVTYPE=DEL;SVLEN=-80;END=1080;RE=9;RNAMES=%s;STRAND=+-\t'
                'GT:DR:DV:PL:GQ\t0/1:3:9:1,2,3:9\n' % names)
    return home


def test_both_host_paths_build_the_same_tag_table(tmp_path):
    blob, lines, tags, _ = build()
    home = _workdir(tmp_path, blob, lines)
    chroms = init_chrom_list(False, home)
    tab, soa = F.generate_callinfo(home + '/sv_calling/variants.vcf', F.read_hap_bam(home + '/snp_phasing/', 2, False), False)
    ing = NativeIngest.load(home + '/sv_calling/variants.vcf', home + '/snp_phasing/', chroms, 2)
    assert ing is not None and ing.handle, getattr(ing, 'why', None)
    from duet_amd import engine
    for field, _ in engine.EfSoA.FIELDS:
        assert np.array_equal(getattr(ing.soa, field), getattr(soa, field)), field
    # mark i of the one candidate is read r<i+1>; the tagged ones carry exactly the spec-derived triples
    got = {}
    for name, m in zip('r1,r2,r3,r4,r5,r6,r7,r8,zz'.split(','), soa.mark_read):
        if m != engine.MARK_ABSENT:
            t = int(soa.read_tag[m])
            got[name] = (t >> 62, (t >> 32) & 0x3FFFFFFF, t & 0xFFFFFFFF)
    assert got == tags
    assert ing.lib.duet_ingest_bam_has_alignments(ing.handle, 0) == 1
    ing.close()


def test_signature_extraction_reads_the_real_cigar_from_cg(tmp_path):
    """SVIM mode walks the CIGAR for insertions / deletions >= 40 bases: for r5 / r6 that is the CG array (33,000 marks
    each), not the 2-operation placeholder (which has none)."""
    blob, lines, tags, ops = build()
    home = _workdir(tmp_path, blob, lines)
    chroms = init_chrom_list(False, home)
    # spec-derived expectation for one long read, by walking the operation list here
    want = []
    for start in (50000, 60000):
        ref = start - 1
        for n, op in ops:
            if op in 'ID' and n >= 40:
                want.append((1 if op == 'I' else 0, ref + 1, n))
            if op in 'MDN=X':
                ref += n
    ing, got = NativeIngest.extract(home + '/snp_phasing/', chroms, thread=2)
    assert ing is not None, got
    assert len(got['pos']) == len(want) == 66000
    assert list(zip(got['type'].tolist(), got['pos'].tolist(), got['span'].tolist())) == want
    rule = svim_oracle.extract_workdir(home, chroms)
    for f in ('contig', 'type', 'pos', 'span'):
        assert np.array_equal(got[f], rule[f]), f
    d = got['depth'][got['depth_off'][0]:got['depth_off'][1]]
    assert d.tolist() == rule['depth'][0]
    ing.close()
