# coding=utf-8
"""Flags, logging and input checks of the `duet` command (mirror of src/duet/utils.py:8-50).

Flag names, defaults, help texts and the positional order are upstream's (utils.py:19-44), so an
existing command line works unchanged.  Two additive options select the device (--device) or shard the contigs
over several GPUs (--gpus).
"""

import argparse
import logging
import os
import sys

_OPTIONS = (
    # short, long, type, default, help
    ('-t', '--thread', int, 4, 'number of threads to use [%(default)s] (SV phasing: while the caller VCF is first read beside the '
                               'BAM loop both use this many workers -- up to twice the number for a few milliseconds; '
                               'DUET_INGEST_STRICT_THREADS=1 splits the budget instead and keeps it a hard bound)'),
    ('-m', '--min_allele_frequency', float, 0.25,
     'minimum allele frequency required to call a candidate SNP [%(default)s]'),
    ('-c', '--cluster_max_distance', float, 0.9,
     'maximum span-position distance between SV marks in a cluster to call a SV candidates, '
     'when the base SV caller is SVIM [%(default)s]'),
    ('-s', '--sv_min_size', int, 50, 'minimum SV size to be reported [%(default)s]'),
    ('-r', '--min_support_read', int, 2,
     'minimum number of reads that support a SV to be reported [%(default)s]'),
)
_POSITIONALS = (
    ('BAM', 'sorted alignment file in .bam format (along with .bai file in the same directory)'),
    ('REFERENCE', 'indexed reference genome in .fasta format (along with .fai file in the same directory)'),
    ('OUTPUT', 'working and output directory (existing files in the directory will be overwritten)'),
)


def build_parser():
    ap = argparse.ArgumentParser(
        description='SNP-Assisted Structural Variant Calling and Phasing Using Oxford Nanopore Sequencing')
    for short, long_, typ, default, text in _OPTIONS:
        ap.add_argument(short, long_, type=typ, default=default, help=text)
    ap.add_argument('-a', '--include_all_ctgs', action='store_true',
                    help='call variants on all contigs, otherwise call chr{1..22,X,Y} [%(default)s]')
    ap.add_argument('-b', '--sv_caller', type=str, default='cutesv',
                    help='choose the base SV caller from cuteSV ("cutesv"), Sniffles (sniffles), or SVIM ("svim") '
                         '[%(default)s]; "svim-gpu" clusters the SV marks with --cluster_max_distance on the GPU instead of '
                         'running svim')
    ap.add_argument('--device', type=int, default=0, help='HIP device index for SV phasing [%(default)s]')
    ap.add_argument('--gpus', type=int, default=1,
                    help='number of GPUs for SV phasing: contigs are sharded over them, one process per GPU [%(default)s]')
    for name, text in _POSITIONALS:
        ap.add_argument(name, type=str, help=text)
    return ap


def parse_args(argv):
    # upstream ignores `argv` and reads sys.argv (utils.py:43); so does this
    return build_parser().parse_args()


def set_logging(home):
    fmt = logging.Formatter('%(asctime)s [%(levelname)s] %(message)s', datefmt='%H:%M:%S')
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    for handler in (logging.FileHandler(home + '/run_duet.log', mode='w'), logging.StreamHandler()):
        handler.setFormatter(fmt)
        root.addHandler(handler)


def add_stream_logging(home):
    """Logging of a rank process started by duet_amd/multi.py: the same line format on stderr, appended to
    <home>/run_duet.log when the parent logs there (DUET_RANK_LOG=1)."""
    fmt = logging.Formatter('%(asctime)s [%(levelname)s] %(message)s', datefmt='%H:%M:%S')
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    handlers = [logging.StreamHandler()]
    if os.environ.get('DUET_RANK_LOG') == '1':
        handlers.append(logging.FileHandler(home + '/run_duet.log', mode='a'))
    for handler in handlers:
        handler.setFormatter(fmt)
        root.addHandler(handler)


def check_envs(ref_path, aln_path):
    if not os.path.exists(aln_path + '.bai'):
        sys.exit("[ERROR] Alignment index .bai file not found, please run 'samtools index " + aln_path + "' first")
    if not os.path.exists(ref_path + '.fai'):
        sys.exit("[ERROR] Reference index .fai file not found, please run 'samtools faidx " + ref_path + "' first")
