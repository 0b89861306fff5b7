# coding=utf-8
"""phased_sv.vcf writer (mirror of src/duet/write_file.py: print_sv :6-17, print_sv_header :19-45).

The output contract is byte-exact: fixed header block, the caller VCF's ##contig lines (first
whitespace token of each; listed contigs in LIST order incl. call-less ones, or every contig line in
file order with -a), column line ending in VALUE, then one row per phased call numbered Duet.1..N
with INFO `SVLEN=<signed>;SVTYPE=<X>` (literal angle brackets) and FORMAT HP:PS.
"""

import logging

from duet_amd.read_file import init_chrom_list, read_file

_HEAD = ''.join(line + '\n' for line in (
    '##fileformat=VCFv4.2',
    '##source=Duet',
    '##ALT=<ID=INS,Description="Insertion of novel sequence relative to the reference">',
    '##ALT=<ID=DEL,Description="Deletion relative to the reference">',
    '##FILTER=<ID=PASS,Description="SV calls passed phasing criterion">',
    '##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Estimated length of the variant">',
    '##FORMAT=<ID=HP,Number=1,Type=String,Description="Haplotype of the SV call">',
    '##FORMAT=<ID=PS,Number=1,Type=String,Description="Phase set which the SV call belongs to">',
))
_COLS = '\t'.join(('#CHROM', 'POS', 'ID', 'REF', 'ALT', 'QUAL', 'FILTER', 'INFO', 'FORMAT', 'VALUE')) + '\n'


def header_text(tokens, chrom_list, include_all_ctgs):
    picked = []
    if include_all_ctgs:
        picked = [t[0] for t in tokens if '##contig=<ID=' in t[0]]
    else:
        for c in chrom_list[:24]:
            with_chr, bare = '##contig=<ID=chr%s,' % c, '##contig=<ID=%s,' % c
            picked.extend(t[0] for t in tokens if with_chr in t[0] or bare in t[0])
    return _HEAD + ''.join(p + '\n' for p in picked) + _COLS


def print_sv_header(vcf_path, output_path, include_all_ctgs, tokens=None):
    if tokens is None:
        tokens = read_file(vcf_path)
    chrom_list = init_chrom_list(include_all_ctgs, vcf_path[:len(vcf_path) - 24])
    with open(output_path, 'w') as out:
        out.write(header_text(tokens, chrom_list, include_all_ctgs))


def rows_text(phased_callset):
    return ''.join('%s\t%s\tDuet.%d\t%s\t%s\t.\tPASS\tSVLEN=%s;SVTYPE=<%s>\tHP:PS\t%s:%s\n' % (
        c['chrom'], c['pos'], n, c['ref'], c['alt'], c['svlen'], c['svtype'], c['hp'], c['ps'])
        for n, c in enumerate(phased_callset, 1))


def print_sv(phased_callset, output_path):
    logging.info('write phased callset into .vcf file')
    with open(output_path, 'a') as out:
        out.write(rows_text(phased_callset))
