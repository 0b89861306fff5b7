# coding=utf-8
"""Stage driver of step E/F (mirror of src/duet/sv_phasing.py:8-20): same paths, same log lines.

Two host paths feed the same HIP kernels:
  * native  -- libduet_ingest.so tokenises the VCF, reads the BAM tags and formats the rows (C++);
  * python  -- duet_amd/read_file.py + sv_phasing_fn.py + write_file.py, which mirror upstream's Python
               including its exceptions.
The native path declines anything it does not vouch for (non-ASCII bytes, malformed numbers, missing
fields ...) and the Python path then takes over, so the observable behaviour is upstream's either way.
Set DUET_NATIVE_INGEST=0 to force the Python path; DUET_DEVICE_ROWS=0 keeps the native path but formats the rows on
the host instead of on the device.

Two additive keyword arguments (upstream's five positionals are unchanged): `device` = HIP device index of a
single-GPU run, `gpus` = N > 1 shards the contigs over N GPUs of the node, one process per GPU
(duet_amd/multi.py: longest-processing-time-first on mark counts, the three kernels per rank, ONE all-gather of the
(pred, ps) records over RCCL, rank 0 writes phased_sv.vcf).
"""

import logging
import os
import time

from duet_amd import engine
from duet_amd.read_file import init_chrom_list
from duet_amd.sv_phasing_fn import generate_phased_callset
from duet_amd.write_file import print_sv, print_sv_header

_BAR = '*' * 25


def load_native(home, thread, include_all_ctgs, caller_vcf, log=True):
    """Native ingest of <home> -> (NativeIngest, chrom_list) or (None, chrom_list) when the Python path has to take
    over (library missing, input declined, or switched off)."""
    chrom_list = init_chrom_list(include_all_ctgs, home)
    if os.environ.get('DUET_NATIVE_INGEST') == '0' or os.environ.get('DUET_USE_SAMTOOLS') == '1':
        return None, chrom_list
    from duet_amd.native import NativeIngest
    ing = NativeIngest.load(caller_vcf, home + '/snp_phasing/', chrom_list, thread)
    if ing is None:
        return None, chrom_list
    if ing.handle is None:
        if log:
            logging.info('native ingest declined (%s); using the Python path' % ing.why)
        return None, chrom_list
    return ing, chrom_list


def log_ingest(ing, chrom_list):
    """Upstream's log lines for the two ingest steps (sv_phasing_fn.py:12,30-33,37,41-45)."""
    snp, sv = ing.log_lines(chrom_list)
    logging.info('extract SNP signatures')
    for line in snp:
        logging.info(line)
    logging.info('extract SV signatures')
    for line in sv:
        logging.info(line)


def write_header(ing, include_all_ctgs, out_vcf):
    """Upstream creates the output file with its header BEFORE any candidate is evaluated (sv_phasing.py:16), so a
    later exception (e.g. ZeroDivisionError, sv_phasing_fn.py:123) leaves a header-only file; so does this."""
    with open(out_vcf, 'wb') as out:
        out.write(ing.header(include_all_ctgs))


def _native(home, svlen_thres, suppread_thres, thread, include_all_ctgs, caller_vcf, out_vcf, ctx):
    ing, chrom_list = load_native(home, thread, include_all_ctgs, caller_vcf)
    if ing is None:
        return False
    try:
        write_header(ing, include_all_ctgs, out_vcf)
        log_ingest(ing, chrom_list)
        logging.info('integrate read weight information')
        logging.info('calculate read weight statistics')
        logging.info('predict SV haplotypes in the callset')
        rows = None if os.environ.get('DUET_DEVICE_ROWS') == '0' else ing.rows()
        if rows is not None and ing.soa.n_cands:
            # (pred, ps) stay on the device; it also orders and formats the rows (duet_ef_rows_run_host)
            body = ctx.ef_rows_host(ing.soa, rows, svlen_thres, suppread_thres)[0]
        else:
            pred, ps = engine.run_ef(ing.soa, svlen_thres, suppread_thres, ctx=ctx)
            body = ing.emit_rows(pred, ps)
        logging.info('write phased callset into .vcf file')
    except BaseException:
        ing.close()
        raise
    ing.close_in_background()                       # (the rows are out: freeing the ingest's memory needs nobody's attention)
    with open(out_vcf, 'ab') as out:
        out.write(body)
    return True


def sv_phasing(home, svlen_thres, suppread_thres, thread, include_all_ctgs, device=0, gpus=1):
    logging.info('%s SV PHASING STARTED %s' % (_BAR, _BAR))
    t0 = time.time()
    caller_vcf = home + '/sv_calling/variants.vcf'
    out_vcf = home + '/phased_sv.vcf'
    logging.info('create output .vcf file')
    done = False
    # (DUET_FORCE_RANKS=1: the one-process-per-GPU path even with one GPU -- rank 0 of 1 over RCCL; tests use it to take the
    # collective through the real backend on a one-GPU box)
    if int(gpus) > 1 or os.environ.get('DUET_FORCE_RANKS') == '1':
        from duet_amd import multi
        done = multi.sv_phasing_sharded(home, svlen_thres, suppread_thres, thread, include_all_ctgs, int(gpus))
    if not done:
        ctx = engine.default_context(int(device))
        if not _native(home, svlen_thres, suppread_thres, thread, include_all_ctgs, caller_vcf, out_vcf, ctx):
            print_sv_header(caller_vcf, out_vcf, include_all_ctgs)
            rows = generate_phased_callset(caller_vcf, home + '/snp_phasing/', svlen_thres, suppread_thres, thread,
                                           include_all_ctgs, ctx=ctx)
            print_sv(rows, out_vcf)
    logging.info('%s SV PHASING COMPLETED IN %ss %s' % (_BAR, round(time.time() - t0, 3), _BAR))
