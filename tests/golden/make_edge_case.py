#!/usr/bin/env python3
# coding=utf-8
"""Hand-written edge-case work dir (three contigs = three caller layouts, since the layout is decided per contig)
run through the UNMODIFIED reference (dev container only) -> tests/golden/cases/edge_handwritten.
Covers: both CHROM spellings in one file, unsorted POS, SVLEN=>n / missing / '.', '+'/'_' integer forms, duplicate and
empty read names, './.' vs '.', equal (chrom, pos) sort ties, DUP:TANDEM sign, '.' read counts in all three layouts,
SUPPORT= vs RE=, later SAM line wins, reads with a single aux field, bare-named BAM, unlisted contig, ##contig
lines in both spellings / with whitespace."""
import json
import os
import shutil
import stat
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def cs(ch, pos, typ, svlen, re, names, gt, dr, dv, extra=''):
    return '\t'.join([ch, str(pos), 'c%d' % pos, 'N', '<%s>' % typ, '.', 'PASS',
                      'PRECISE;SVTYPE=%s;%sEND=%d;RE=%s;RNAMES=%s;STRAND=+-%s' % (typ, svlen, pos + 10, re, names, extra),
                      'GT:DR:DV:PL:GQ', '%s:%s:%s:1,2,3:9' % (gt, dr, dv)])


def sn(ch, pos, typ, svlen, supp, names, gt, gq, dr, dv):
    return '\t'.join([ch, str(pos), 's%d' % pos, 'N', '<%s>' % typ, '60', 'PASS',
                      'PRECISE;SVTYPE=%s;SVLEN=%d;END=%d;SUPPORT=%d;RNAMES=%s;COVERAGE=1,2,3;AF=0.5' % (
                          typ, svlen, pos + 5, supp, names), 'GT:GQ:DR:DV', '%s:%s:%s:%s' % (gt, gq, dr, dv)])


def sv(ch, pos, typ, svlen, supp, names, sample):
    return '\t'.join([ch, str(pos), 'v%d' % pos, 'N', '<%s>' % typ, '7', 'PASS',
                      'SVTYPE=%s;END=%d;SVLEN=%d;SUPPORT=%d;STD_SPAN=1.5;STD_POS=2.25;READS=%s' % (
                          typ, pos + 9, svlen, supp, names), 'GT:DP:AD', sample])


def sam(name, ch, hp=None, pc=None, ps=None, extra_tags='NM:i:1'):
    core = '%s\t0\t%s\t10\t60\t*\t*\t0\t0\t*\t*\t%s' % (name, ch, extra_tags)
    if hp is not None:
        core += '\tHP:i:%d\tPC:i:%d\tPS:i:%d' % (hp, pc, ps)
    return core


def build(home):
    os.makedirs(home + '/sv_calling')
    os.makedirs(home + '/snp_phasing')
    hdr = ['##fileformat=VCFv4.2', '##contig=<ID=chr1,length=1000>', '##contig=<ID=1,length=1000>',
           '##contig=<ID=2,length=900>', '##contig=<ID=chr2,length=900,assembly="hg 19">', '##contig=<ID=chrX,length=800>',
           '##contig=<ID=chrUn_1,length=5>', '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS']
    recs = [
        cs('chr1', 100, 'INS', 'SVLEN=88;', 5, 'a1,a2,a3,a4,a5', '0/1', 7, 5),
        cs('1', 150, 'DEL', 'SVLEN=-120;', 4, 'a1,a1,a2,zz', '1/1', 0, 4),
        cs('chr1', 90, 'DEL', 'SVLEN=>300;', 6, 'a3,a4,a5,a6,b1,b2', '0/1', 2, 6),
        cs('chr1', 200, 'DUP', '', 3, 'b1,b2,b3', '0/1', 1, 3),
        cs('chr1', 210, 'DUP:TANDEM', 'SVLEN=75;', 3, 'b1,b2,b3', '0/1', '.', 3),
        cs('chr1', 220, 'INV', 'SVLEN=.;', 9, 'b1', '0/1', 3, 9),
        cs('chr1', 230, 'INS', 'SVLEN=60;', 2, '', '0/1', 0, 2),
        cs('chr1', 240, 'INS', 'SVLEN=60;', 4, 'c1,c2,c3,c4', './.', 0, 4),
        cs('chr1', 250, 'INS', 'SVLEN=60;', 4, 'c1,c2,c3,c4', '.', 0, 4),
        cs('chr1', 250, 'DEL', 'SVLEN=-60;', 4, 'c1,c2,c3,c4', '0/1', 0, 4),
        cs('chr1', 260, 'INS', 'SVLEN=+70;', '0_4', 'a1,b1,c1,d1', '0/1', 1, 4),
        cs('chr1', 270, 'DEL', 'SVLEN=-80;XSVLEN=5;', 20, ','.join(['a1'] * 20), '0/1', 20, 20, ';SUPPORT_X=1'),
        cs('chr1', 250, 'INS', 'SVLEN=61;', 5, 'zz1,zz2,zz3,zz4', '0/1', 0, 5),
        sn('chr2', 300, 'INS', 90, 5, 'e1,e2,e3,e4,e5', '0/1', 60, 7, 5),
        sn('2', 310, 'DEL', -95, 4, 'e1,e2,f1,f2', '1/1', '.', 0, 4),
        sn('chr2', 320, 'DEL', -99, 6, 'f1,f2,f3,g1,g2,g3', '0/1', 3, 2, 6),
        sn('chr2', 330, 'INS', 55, 4, 'g1,g2,g3,q', '0/1', 0, 0, 4),
        sv('chrX', 400, 'DEL', -100, 5, 'h1,h2,h3,h4,h5', '0/1:12:7,5'),
        sv('chrX', 410, 'DUP:TANDEM', 100, 2, 'h1,h2', './.:.:.,.'),
        sv('chrX', 420, 'INS', 100, 4, 'h1,i1,i2,i3', '1/1:4:.,4'),
        sv('chrX', 430, 'INS', 100, 9, 'i1,i2,i3,i4,i5,i6,i7,i8,i9', '0/1:10:1,9'),
        '\t'.join(['chrUn_1', '1', 'u', 'N', '<DEL>', '.', 'PASS', 'SVTYPE=DEL;SVLEN=-80;RE=5;RNAMES=a1', 'GT:DR:DV:PL:GQ',
                   '0/1:3:5:1,2,3:9']),
    ]
    with open(home + '/sv_calling/variants.vcf', 'w') as f:
        f.write('\n'.join(hdr + recs) + '\n')
    c1 = [sam('a1', 'chr1', 1, 300, 50), sam('a2', 'chr1', 1, 200, 50), sam('a3', 'chr1', 2, 100, 50), sam('a4', 'chr1'),
          sam('a5', 'chr1', 1, 9000, 50), sam('a6', 'chr1', 2, 8100, 50), sam('b1', 'chr1', 1, 1370, 700),
          sam('b2', 'chr1', 1, 1369, 700), sam('b3', 'chr1', 2, 0, 50), sam('c1', 'chr1', 1, 8101, 700),
          sam('c2', 'chr1', 2, 8101, 700), sam('c3', 'chr1', 1, 10, 700), sam('c4', 'chr1', 2, 10, 700),
          sam('d1', 'chr1', 2, 500, 900), sam('a2', 'chr1', 2, 250, 50, 'NM:i:2\tSA:Z:x'),
          sam('noaux', 'chr1', None, None, None, 'NM:i:0')]
    c2 = [sam('e%d' % i, 'chr2', 1 + i % 2, 100 * i, 300) for i in range(1, 6)] + \
         [sam('f%d' % i, 'chr2', 2, 2400 + i, 600) for i in range(1, 4)] + [sam('g%d' % i, 'chr2', 1, 972, 300) for i in range(1, 4)]
    cx = [sam('h%d' % i, 'chrX', 1, 50 * i, 400) for i in range(1, 6)] + [sam('i%d' % i, 'chrX', 2, 40 * i, 400) for i in range(1, 10)]
    for stem, lines in (('chr1', c1), ('2', c2), ('chrX', cx)):
        open(home + '/snp_phasing/%s.bam' % stem, 'w').close()
        with open(home + '/snp_phasing/%s.bam.sam' % stem, 'w') as f:
            f.write('\n'.join(lines) + '\n')


def main():
    if not os.path.isdir('/root/reference/src'):
        sys.exit('reference not present')
    sys.path.insert(0, '/root/reference/src')
    tmp = tempfile.mkdtemp(prefix='duet_edge_')
    shim = os.path.join(tmp, 'shim')
    os.makedirs(shim)
    with open(os.path.join(shim, 'samtools'), 'w') as f:
        f.write('#!/bin/sh\nfor a; do last="$a"; done\ncat "$last.sam"\n')
    os.chmod(os.path.join(shim, 'samtools'), 0o755)
    os.environ['PATH'] = shim + os.pathsep + os.environ['PATH']
    home = os.path.join(HERE, 'cases', 'edge_handwritten')
    if os.path.isdir(home):
        shutil.rmtree(home)
    build(home)
    from duet.sv_phasing import sv_phasing
    sv_phasing(home, 50, 2, 4, False)
    for n in os.listdir(home + '/snp_phasing'):
        if n.endswith('.bam'):
            os.remove(os.path.join(home, 'snp_phasing', n))
    with open(os.path.join(home, 'params.json'), 'w') as f:
        json.dump(dict(seed=0, dialect='mixed', svlen_thres=50, suppread_thres=2), f)
    print(open(os.path.join(home, 'phased_sv.vcf')).read())
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
