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
"""

import logging
import os
import time

from duet_amd import engine
from duet_amd.read_file import init_chrom_list
from duet_amd.sv_phasing_fn import generate_phased_callset
from duet_amd.write_file import print_sv, print_sv_header

_BAR = '*' * 25


def _native(home, svlen_thres, suppread_thres, thread, include_all_ctgs, caller_vcf, out_vcf):
    if os.environ.get('DUET_NATIVE_INGEST') == '0' or os.environ.get('DUET_USE_SAMTOOLS') == '1':
        return False
    from duet_amd.native import NativeIngest
    chrom_list = init_chrom_list(include_all_ctgs, home)
    logging.info('extract SNP signatures')
    logging.info('extract SV signatures')
    ing = NativeIngest.load(caller_vcf, home + '/snp_phasing/', chrom_list, thread)
    if ing is None:
        return False
    if ing.handle is None:
        logging.info('native ingest declined (%s); using the Python path' % ing.why)
        return False
    try:
        logging.info('integrate read weight information')
        logging.info('calculate read weight statistics')
        logging.info('predict SV haplotypes in the callset')
        rows = None if os.environ.get('DUET_DEVICE_ROWS') == '0' else ing.rows()
        if rows is not None and ing.soa.n_cands:
            # (pred, ps) stay on the device; it also orders and formats the rows (duet_ef_rows_run_host)
            body = engine.default_context().ef_rows_host(ing.soa, rows, svlen_thres, suppread_thres)[0]
            logging.info('write phased callset into .vcf file')
            text = ing.header(include_all_ctgs) + body
        else:
            pred, ps = engine.run_ef(ing.soa, svlen_thres, suppread_thres)
            logging.info('write phased callset into .vcf file')
            text = ing.emit(pred, ps, include_all_ctgs)
    finally:
        ing.close()
    with open(out_vcf, 'wb') as out:
        out.write(text)
    return True


def sv_phasing(home, svlen_thres, suppread_thres, thread, include_all_ctgs):
    logging.info('%s SV PHASING STARTED %s' % (_BAR, _BAR))
    t0 = time.time()
    caller_vcf = home + '/sv_calling/variants.vcf'
    out_vcf = home + '/phased_sv.vcf'
    logging.info('create output .vcf file')
    if not _native(home, svlen_thres, suppread_thres, thread, include_all_ctgs, caller_vcf, out_vcf):
        print_sv_header(caller_vcf, out_vcf, include_all_ctgs)
        rows = generate_phased_callset(caller_vcf, home + '/snp_phasing/', svlen_thres, suppread_thres, thread,
                                       include_all_ctgs)
        print_sv(rows, out_vcf)
    logging.info('%s SV PHASING COMPLETED IN %ss %s' % (_BAR, round(time.time() - t0, 3), _BAR))
