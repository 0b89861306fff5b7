# coding=utf-8
"""-m gpu: rows of phased_sv.vcf formatted on the device (duet_rows_run_device) against the host formatter of the
native ingest and against the Python oracle's full text -- long REF/ALT texts, mixed CHROM spellings inside a
contig, many contigs, every caller dialect."""
import os
import shutil
import tempfile

import numpy as np
import pytest

from duet_amd import engine, synth
from duet_amd.devmem import DeviceProblem, device_rows
from duet_amd.native import NativeIngest
from duet_amd.read_file import init_chrom_list
from oracle import ef_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def scramble_vcf(path, seed):
    """Rewrite some records: long REF / ALT texts, the other CHROM spelling (both are accepted, read_file.py:30)."""
    rng = synth.SplitMix(777 + seed)
    out = []
    with open(path) as f:
        lines = f.read().split('\n')
    for ln in lines:
        if not ln or ln[0] == '#':
            out.append(ln)
            continue
        t = ln.split('\t')
        r = rng.one(12)
        if r == 0:
            t[4] = ''.join('ACGT'[b] for b in rng.below(1 + rng.one(3000), 4))
        elif r == 1:
            t[3] = ''.join('ACGT'[b] for b in rng.below(1 + rng.one(700), 4))
        elif r == 2 and t[0] != 'chrM':
            t[0] = t[0][3:] if t[0].startswith('chr') else 'chr' + t[0]
        out.append('\t'.join(t))
    with open(path, 'w') as f:
        f.write('\n'.join(out))


@pytest.mark.parametrize('kind,seed,dialect', [('fuzz', 1, 'cutesv'), ('fuzz', 2, 'sniffles'), ('fuzz', 3, 'svim'),
                                               ('genome_small', 4, 'cutesv'), ('chr21', 5, 'svim')])
def test_device_rows_equal_host_rows_and_oracle(kind, seed, dialect):
    home = tempfile.mkdtemp(prefix='duet_rows_')
    try:
        H.build_case(home, kind, seed, dialect, write_bam=True, write_sam=True)
        vcf = os.path.join(home, 'sv_calling', 'variants.vcf')
        scramble_vcf(vcf, seed)
        want_text = ef_oracle.sv_phasing_text(home, 50, 2)
        ing = NativeIngest.load(vcf, home + '/snp_phasing/', init_chrom_list(False, home), 2)
        assert ing is not None and ing.handle is not None, getattr(ing, 'why', None)
        rows = ing.rows()
        assert rows is not None
        ctx = engine.default_context()
        dp = DeviceProblem(ing.soa, 50, 2)
        stream = dp.run(ctx)
        ctx.check(stream)
        pred, ps = dp.results()
        host = ing.emit(pred, ps, False)
        header = ing.header(False)
        body, n_rows = device_rows(ctx, dp, rows, stream=stream)
        assert n_rows == int((pred != 0).sum())
        assert header + body == host
        body2, n_rows2 = ctx.ef_rows_host(ing.soa, rows, 50, 2)           # the host-array entry: same bytes
        assert (body2, n_rows2) == (body, n_rows)
        assert (header + body).decode() == want_text
        assert n_rows > 0
        ing.close()
    finally:
        shutil.rmtree(home, ignore_errors=True)


def test_sv_phasing_uses_the_device_rows(monkeypatch):
    """The stage driver's default path formats the rows on the device."""
    from duet_amd import _lib
    from duet_amd.sv_phasing import sv_phasing
    calls = []
    real = _lib.Context.ef_rows_host
    monkeypatch.setattr(_lib.Context, 'ef_rows_host', lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    home = tempfile.mkdtemp(prefix='duet_rows_')
    try:
        H.build_case(home, 'chr21', 9, 'cutesv')
        sv_phasing(home, 50, 2, 2, False)
        assert calls
        with open(os.path.join(home, 'phased_sv.vcf')) as f:
            assert f.read() == ef_oracle.sv_phasing_text(home, 50, 2)
    finally:
        shutil.rmtree(home, ignore_errors=True)
