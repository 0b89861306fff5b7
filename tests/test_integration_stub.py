# coding=utf-8
"""integration/ef_gpu.py (the stub INTEGRATION.md tells a maintainer to add to upstream) on upstream-shaped
`callstat` lists, with the C oracle standing in for the device call."""
import importlib.util
import os
import shutil

import numpy as np

from duet_amd import engine
from duet_amd import sv_phasing_fn as F
from duet_amd.read_file import init_chrom_list
from oracle import c_oracle
from tests import helpers as H
from tests.test_c_oracle import materialise_bams

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_stub(real_run=False):
    spec = importlib.util.spec_from_file_location('ef_gpu_stub', os.path.join(REPO, 'integration', 'ef_gpu.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    if real_run:
        return m                             # the stub exactly as a maintainer would add it: its own _run, the real library

    class View(object):                      # the stub's ctypes problem -> the arrays the oracle wants
        pass

    def run(problem, pred, ps):
        import ctypes
        v = View()
        v.n_contigs, v.n_cands = problem.n_contigs, problem.n_cands

        def arr(ptr, n, ct, dt):
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(n,)).astype(dt) if n else np.zeros(0, dt)
        C, M, R, K = problem.n_cands, problem.n_marks, problem.n_reads, problem.n_contigs
        v.cand_ctg_off = arr(problem.cand_ctg_off, K + 1, ctypes.c_uint32, np.uint32)
        v.read_tag = arr(problem.read_tag, R, ctypes.c_uint64, np.uint64)
        for name in ('cand_pos', 'cand_svlen', 'cand_svread', 'cand_refread'):
            setattr(v, name, arr(getattr(problem, name), C, ctypes.c_uint32, np.uint32))
        v.cand_gt_ok = arr(problem.cand_gt_ok, C, ctypes.c_uint8, np.uint8)
        v.cand_off = arr(problem.cand_off, C + 1, ctypes.c_uint32, np.uint32)
        v.mark_read = arr(problem.mark_read, M, ctypes.c_uint32, np.uint32)
        rc, p, s = c_oracle.ef(v, problem.svlen_thres, problem.suppread_thres)
        if rc == -5:
            raise ZeroDivisionError('division by zero')
        pred[:] = p
        ps[:] = s
    m._run = run
    return m


def upstream_callstat(tab, soa):
    """What upstream's generate_callinfo would hand over, rebuilt from the product's own tables."""
    out = []
    for i in range(len(tab)):
        marks = []
        for name, m in zip(tab.names[i], soa.mark_read[soa.cand_off[i]:soa.cand_off[i + 1]]):
            if m == engine.MARK_ABSENT:
                marks.append([name])
            else:
                t = int(soa.read_tag[m])
                marks.append([name, t >> 62, t & 0xFFFFFFFF, (t >> 32) & 0x3FFFFFFF])
        out.append(dict(chrom=tab.chrom[i], pos=int(tab.pos[i]), ref=tab.ref[i], alt=tab.alt[i],
                        svlen=int(tab.svlen_abs[i]), svtype=tab.svtype[i], svreadinfo=marks, svread=int(tab.svread[i]),
                        callgt=tab.gt[i], refread=int(tab.refread[i])))
    return out


def stub_rows_match_goldens(stub, tmp_path, cases):
    from duet_amd import write_file as W, read_file as RF
    for name, src, params in cases:
        home = str(tmp_path / name)
        shutil.copytree(src, home)
        materialise_bams(home)
        vcf = home + '/sv_calling/variants.vcf'
        tab, soa = F.generate_callinfo(vcf, F.read_hap_bam(home + '/snp_phasing/', 4, False), False)
        rows = stub.phase_on_gpu(upstream_callstat(tab, soa), init_chrom_list(False, home), params['svlen_thres'],
                                 params['suppread_thres'])
        rows.sort(key=lambda r: (r['chrom'], r['pos']))              # upstream's line 229 stays in place
        head = W.header_text(RF.read_file(vcf), init_chrom_list(False, home), False)
        with open(os.path.join(src, 'phased_sv.vcf')) as f:
            assert head + W.rows_text(rows) == f.read(), name


def test_stub_reproduces_golden_rows(tmp_path):
    stub_rows_match_goldens(load_stub(), tmp_path, H.full_cases()[:6])
