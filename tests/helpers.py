# coding=utf-8
"""Shared helpers for the test-suite: case construction mirroring tests/golden/make_golden.py."""
import hashlib
import json
import os

from duet_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def sha256_bytes(b):
    return hashlib.sha256(b).hexdigest()


def inputs_digest(home):
    h = hashlib.sha256()
    with open(os.path.join(home, 'sv_calling', 'variants.vcf'), 'rb') as f:
        h.update(f.read())
    d = os.path.join(home, 'snp_phasing')
    for n in sorted(os.listdir(d)):
        if n.endswith('.sam'):
            h.update(n.encode())
            with open(os.path.join(d, n), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def case_contigs(kind, seed):
    if kind == 'fuzz':
        return synth.fuzz_case(seed, n_contigs=2 + seed % 3)
    if kind == 'chr21':
        return [synth.bench_contig('21', 1500, 1500, seed, deg_lo=2, deg_hi=14)]
    if kind == 'config2':
        return [synth.bench_contig('1', 200000, 100000, seed)]
    if kind == 'genome_small':
        return synth.bench_genome(200000, seed)
    raise ValueError(kind)


def build_case(home, kind, seed, dialect, write_bam=True, write_sam=True):
    contigs = case_contigs(kind, seed)
    synth.write_workdir(home, contigs, dialect=dialect, seed=int(seed), write_bam=write_bam, write_sam=write_sam)
    return contigs


def seeded_plan():
    with open(os.path.join(GOLDEN, 'seeded.json')) as f:
        return json.load(f)


def full_cases():
    d = os.path.join(GOLDEN, 'cases')
    out = []
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name, 'params.json')) as f:
            out.append((name, os.path.join(d, name), json.load(f)))
    return out
