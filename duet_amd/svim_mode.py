# coding=utf-8
"""SVIM mode without the external caller: the haplotagged BAMs of <home>/snp_phasing alone ->
raw SV marks (CIGAR insertions / deletions, native BAM pass) -> stage A0 -> adapter -> step E/F, the last three in one
device pipeline (duet_svim_phase_device).  Upstream runs `svim alignment` here (src/duet/sv_calling.py:13-15) and reads
its VCF back; the extraction and clustering rules used instead are this repository's own (oracle/svim_oracle.py,
oracle/cluster_oracle.c; parity unpinned), step E/F is the pinned one.
"""

import logging
import os
import time

import numpy as np

from duet_amd import bamio, engine
from duet_amd.native import NativeIngest
from duet_amd.read_file import init_chrom_list


def phase_from_bams(home, svlen_thres=50, suppread_thres=2, thread=4, include_all_ctgs=False, max_dist=0.9,
                    min_sv_size=40, min_mapq=20, depth_bin=1000, ctx=None):
    """-> dict(chroms, cand_contig u16[N], cand_type u8[N] (1 INS / 0 DEL), cand_pos, cand_span, support, pred, ps)"""
    from duet_amd.devmem import DeviceSvim
    chroms = init_chrom_list(include_all_ctgs, home)
    ing, got = NativeIngest.extract(home + '/snp_phasing/', chroms, thread, min_sv_size, min_mapq, depth_bin)
    if ing is None:
        raise RuntimeError('signature extraction declined the input: %s' % got)
    ing.close()
    N0 = dict(chroms=chroms, cand_contig=np.zeros(0, np.uint16), cand_type=np.zeros(0, np.uint8),
              cand_pos=np.zeros(0, np.uint32), cand_span=np.zeros(0, np.uint32), support=np.zeros(0, np.int64),
              pred=np.zeros(0, np.uint8), ps=np.zeros(0, np.uint32))
    N0['n_marks'] = 0
    if len(got['pos']) == 0:
        return N0
    if ctx is None:
        ctx = engine.default_context()
    ds = DeviceSvim(got, got['read_tag'], got['depth'], got['depth_off'], depth_bin, svlen_thres, suppread_thres,
                    max_dist=max_dist, device='cuda:%d' % ctx.device_id)
    ds.run_fused(ctx)
    ctx.check(ds.torch.cuda.current_stream(ds.device).cuda_stream)
    out = ds.fetch()
    return dict(chroms=chroms, cand_contig=out['cand_contig'], cand_type=out['cand_type'], cand_pos=out['cand_pos'],
                cand_span=out['cand_span'], support=np.diff(out['cand_off'].astype(np.int64)), pred=out['pred'],
                ps=out['ps'], n_marks=len(got['pos']))


def spelled_contigs(home, chroms):
    """CHROM text per contig: the spelling of its BAM (chr<c>.bam else <c>.bam, as sv_phasing_fn.py:19-24 looks them up)."""
    names = []
    for c in chroms:
        names.append('chr' + c if os.path.exists(os.path.join(home, 'snp_phasing', 'chr' + c + '.bam')) else c)
    return names


def rows_text(home, res):
    """Rows in phased_sv.vcf's layout (write_file.py:6-17) for the phased candidates of `res`: symbolic REF/ALT,
    sorted like sv_phasing_fn.py:229 (CHROM as text, POS), ids renumbered."""
    names = spelled_contigs(home, res['chroms'])
    hp = {1: '1|0', 2: '0|1', 3: '1|1'}
    keep = np.nonzero(res['pred'])[0]
    order = sorted(keep, key=lambda i: (names[int(res['cand_contig'][i])], int(res['cand_pos'][i])))
    out = []
    for n, i in enumerate(order):
        t = 'INS' if int(res['cand_type'][i]) == 1 else 'DEL'
        ln = int(res['cand_span'][i])
        out.append('%s\t%d\tDuet.%d\tN\t<%s>\t.\tPASS\tSVLEN=%d;SVTYPE=<%s>\tHP:PS\t%s:%d\n' % (
            names[int(res['cand_contig'][i])], int(res['cand_pos'][i]), n + 1, t, ln if t == 'INS' else -ln, t,
            hp[int(res['pred'][i])], int(res['ps'][i])))
    return ''.join(out)


def header_text(home, chroms):
    """phased_sv.vcf header lines (write_file.py:19-45).  Upstream copies the ##contig lines of the caller VCF; this mode
    has no caller VCF, so every listed contig that has a BAM contributes one line from the BAM's own reference list."""
    from duet_amd.write_file import _COLS, _HEAD
    lines = []
    for c, name in zip(chroms, spelled_contigs(home, chroms)):
        bam = os.path.join(home, 'snp_phasing', name + '.bam')
        if not os.path.exists(bam):
            continue
        for ref, length in bamio.read_refs(bam):
            if ref == name:
                lines.append('##contig=<ID=%s,length=%d>\n' % (ref, length))
                break
    return _HEAD + ''.join(lines) + _COLS


def sv_phasing_from_bams(home, svlen_thres, suppread_thres, thread, include_all_ctgs, cluster_max_distance=0.9, device=0):
    """`duet ... -b svim-gpu -c <max distance>`: SV calling (signatures + clustering, what `-b svim` delegates to the
    external `svim alignment ... --cluster_max_distance c`, sv_calling.py:13-15) AND SV phasing on the GPU, from the
    haplotagged BAMs of <home>/snp_phasing -> <home>/phased_sv.vcf.  The clustering half is this repository's own rule
    (parity unpinned); step E/F is the pinned one."""
    bar = '*' * 25
    logging.info('%s SV CALLING + PHASING (GPU, svim-gpu) STARTED %s' % (bar, bar))
    t0 = time.time()
    logging.info('create output .vcf file')
    chroms = init_chrom_list(include_all_ctgs, home)
    out_vcf = home + '/phased_sv.vcf'
    with open(out_vcf, 'w') as out:
        out.write(header_text(home, chroms))
    logging.info('extract SNP and SV signatures from the haplotagged alignments')
    res = phase_from_bams(home, svlen_thres, suppread_thres, thread, include_all_ctgs, max_dist=cluster_max_distance,
                          min_sv_size=max(int(svlen_thres), 1), ctx=engine.default_context(int(device)))
    logging.info('  %d SV marks clustered into %d candidates, %d phased (clustering rule: parity unpinned)' % (
        res['n_marks'], len(res['pred']), int(np.count_nonzero(res['pred']))))
    logging.info('write phased callset into .vcf file')
    with open(out_vcf, 'a') as out:
        out.write(rows_text(home, res))
    logging.info('%s SV CALLING + PHASING COMPLETED IN %ss %s' % (bar, round(time.time() - t0, 3), bar))
