# coding=utf-8
"""The `duet` command: BAM REFERENCE OUTPUT [-t -m -c -s -r -a -b] as upstream (src/duet/duet:14-28), two additive
flags (--device, --gpus).  The pipeline is a table of stages; B, C, D shell out to external tools (duet_amd/stages.py),
the last one -- SV phasing, steps E/F -- runs on the MI355X."""

import logging
import os
import time

from duet_amd import engine, stages
from duet_amd.sv_phasing import sv_phasing
from duet_amd.utils import check_envs, parse_args, set_logging

_BAR = '*' * 25


def pipeline(a):
    """(callable, arguments) per stage, in upstream's fixed order: SNP call, SV call, SNP phase, SV phase.
    `-b svim-gpu` (additive): no external SV caller -- signatures, `--cluster_max_distance` clustering and phasing all run on
    the GPU from the haplotagged BAMs (duet_amd/svim_mode.py)."""
    if a.sv_caller == 'svim-gpu':
        from duet_amd.svim_mode import sv_phasing_from_bams
        return (
            (stages.snp_calling, (a.OUTPUT, a.REFERENCE, a.BAM, a.min_allele_frequency, a.thread, a.include_all_ctgs)),
            (stages.snp_phasing, (a.OUTPUT, a.REFERENCE, a.BAM, a.thread)),
            (sv_phasing_from_bams, (a.OUTPUT, a.sv_min_size, a.min_support_read, a.thread, a.include_all_ctgs,
                                    a.cluster_max_distance, a.device, a.gpus)),
        )
    return (
        (stages.snp_calling, (a.OUTPUT, a.REFERENCE, a.BAM, a.min_allele_frequency, a.thread, a.include_all_ctgs)),
        (stages.sv_calling, (a.OUTPUT, a.REFERENCE, a.BAM, a.cluster_max_distance, a.sv_min_size, a.thread, a.sv_caller,
                             a.min_support_read)),
        (stages.snp_phasing, (a.OUTPUT, a.REFERENCE, a.BAM, a.thread)),
        (sv_phasing, (a.OUTPUT, a.sv_min_size, a.min_support_read, a.thread, a.include_all_ctgs, a.device, a.gpus)),
    )


def main(argv):
    a = parse_args(argv)
    check_envs(a.REFERENCE, a.BAM)
    os.makedirs(a.OUTPUT, exist_ok=True)
    set_logging(a.OUTPUT)
    began = time.time()
    logging.info(_BAR + ' DUET STARTED ' + _BAR)
    todo = pipeline(a)
    if a.gpus > 1:
        # N devices and the library, checked in a child BEFORE the external stages run (hours of Clair3 / WhatsHap should
        # not end in "no GPU"); this process stays GPU-free -- it starts the ranks later (duet_amd/launch.py)
        from duet_amd import launch
        launch.probe_devices(1 if os.environ.get('DUET_ONE_GPU') == '1' else a.gpus)
    for fn, args in todo[:-1]:
        fn(*args)
    if a.gpus <= 1 and os.environ.get('DUET_FORCE_RANKS') != '1':
        engine.default_context(a.device)      # fail before the last stage if there is no MI355X / no library
    elif a.gpus <= 1:
        # (DUET_FORCE_RANKS=1: the last stage starts a rank process even for one GPU -- this process must stay GPU-free,
        # so the device is probed in a child)
        from duet_amd import launch
        launch.probe_devices(1)
    fn, args = todo[-1]
    fn(*args)
    logging.info('%s DUET FINISHED IN %ss %s' % (_BAR, round(time.time() - began, 3), _BAR))
    logging.info('OUTPUT .VCF FILE AT ' + a.OUTPUT + '/phased_sv.vcf')
