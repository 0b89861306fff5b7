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
    if kind == 'config2_8d':            # SURVEY 8d's config-2 generator to the letter (20 % absent names, exponential PC)
        return [synth.bench_contig('1', 200000, 100000, seed, literal_8d=True)]
    if kind == 'genome_small':
        return synth.bench_genome(200000, seed)
    raise ValueError(kind)


def build_case(home, kind, seed, dialect, write_bam=True, write_sam=True):
    if kind == 'fuzz_a':                # -a mode (tests/golden/make_golden_r2.py: build_all_ctgs_case)
        contigs, listing, header_contigs = synth.fuzz_case_all_ctgs(seed)
        synth.write_workdir(home, contigs, dialect=dialect, seed=int(seed), write_bam=write_bam, write_sam=write_sam,
                            listing=listing, header_contigs=header_contigs, extra_contig_records='chrM' not in listing)
        return contigs
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


def seeded_r2_plan():
    """Round-2 pins (tests/golden/make_golden_r2.py): -a fuzz cases, genome_small in the SVIM / Sniffles dialects,
    config 3, and the ZeroDivisionError case."""
    with open(os.path.join(GOLDEN, 'seeded_r2.json')) as f:
        return json.load(f)


def all_ctgs_cases():
    d = os.path.join(GOLDEN, 'cases_a')
    out = []
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name, 'params.json')) as f:
            out.append((name, os.path.join(d, name), json.load(f)))
    return out


def listing_of(home):
    """The contig universe of a -a work dir (what the `tabix --list-chroms` shim prints)."""
    with open(os.path.join(home, 'snp_calling', 'pileup.vcf.gz.chroms')) as f:
        return f.read().split('\n')[:-1]


def install_tabix_shim(tmp_dir, monkeypatch):
    """-a mode asks `tabix --list-chroms <home>/snp_calling/pileup.vcf.gz` (read_file.py:13-15); tabix is not in the
    image, so a PATH shim prints the listing laid beside that path."""
    shim = os.path.join(str(tmp_dir), 'shim')
    os.makedirs(shim, exist_ok=True)
    p = os.path.join(shim, 'tabix')
    with open(p, 'w') as f:
        f.write('#!/bin/sh\nfor a; do last="$a"; done\ncat "$last.chroms"\n')
    os.chmod(p, 0o755)
    monkeypatch.setenv('PATH', shim + os.pathsep + os.environ['PATH'])
