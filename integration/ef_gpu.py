# coding=utf-8
"""The ctypes stub a maintainer would drop into upstream as src/duet/ef_gpu.py.

It consumes exactly what upstream's generate_callinfo returns (sv_phasing_fn.py:49-68: a list of dicts with
chrom, pos, svlen, svtype, svread, refread, callgt, ref, alt and svreadinfo = [name] or [name, hap, ps, pc] per
mark) and replaces lines 189-228 of generate_phased_callset.  tests/test_integration_stub.py runs it on such
lists -- with the C oracle in place of `_run` where there is no GPU, and with the real `_run` (this file unchanged,
libduet_ef.so on an MI355X) in tests/test_gpu_r2.py -- and compares with the reference's golden rows.
"""
import ctypes, os, numpy as np

class _Problem(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ('n_contigs', 'n_cands', 'n_marks', 'n_reads')] + \
               [(n, ctypes.c_void_p) for n in ('cand_ctg_off', 'read_tag', 'cand_pos', 'cand_svlen', 'cand_svread',
                                               'cand_refread', 'cand_gt_ok', 'cand_off', 'mark_read')] + \
               [('svlen_thres', ctypes.c_uint32), ('suppread_thres', ctypes.c_uint32)]

_lib = _ctx = None


def _run(problem, pred, ps):
    """duet_ef_run_host on device 0 (library and context created on first use)."""
    global _lib, _ctx
    if _lib is None:
        _lib = ctypes.CDLL(os.environ.get('DUET_EF_LIB', 'libduet_ef.so'))   # on the loader path, or named explicitly
        _lib.duet_ctx_create.restype = ctypes.c_void_p
        _lib.duet_last_error.restype = ctypes.c_char_p
        _lib.duet_last_error.argtypes = [ctypes.c_void_p]
        _ctx = _lib.duet_ctx_create(0)
        if not _ctx:
            raise RuntimeError(_lib.duet_last_error(None).decode())
    rc = _lib.duet_ef_run_host(ctypes.c_void_p(_ctx), ctypes.byref(problem), pred.ctypes.data_as(ctypes.c_void_p),
                               ps.ctypes.data_as(ctypes.c_void_p), None)
    if rc == -5:
        raise ZeroDivisionError('division by zero')                 # what :123 raises
    if rc:
        raise RuntimeError(_lib.duet_last_error(ctypes.c_void_p(_ctx)).decode())


def phase_on_gpu(callstat, chrom_list, svlen_thres, suppread_thres):
    """callstat: the list of dicts generate_callinfo returns (sv_phasing_fn.py:49-68)."""
    # contig of each call, by CHROM text (sv_phasing_fn.py:198, :208)
    idx = {}
    for k, c in enumerate(chrom_list):
        idx['chr' + c] = idx[c] = k
    ctg = np.array([idx[c['chrom']] for c in callstat])            # already contig-major (:50-51)
    ctg_off = np.searchsorted(ctg, np.arange(len(chrom_list) + 1)).astype(np.uint32)
    # tag table: one word per distinct (hap, ps, pc) triple is enough -- the join (:47-48) already happened
    tags, tag_id, mark, off = [], {}, [], [0]
    for c in callstat:
        for r in c['svreadinfo']:                                   # [name] or [name, hap, ps, pc]
            if len(r) == 1:
                mark.append(0xFFFFFFFF)
            else:
                key = (r[1] if r[1] in (1, 2) else 3, min(r[3], (1 << 30) - 2), r[2])
                if key not in tag_id:
                    tag_id[key] = len(tags)
                    tags.append((key[0] << 62) | (key[1] << 32) | key[2])
                mark.append(tag_id[key])
        off.append(len(mark))
    u32 = lambda xs: np.ascontiguousarray(xs, dtype=np.uint32)
    a = dict(cand_ctg_off=ctg_off, read_tag=np.array(tags, dtype=np.uint64),
             cand_pos=u32([c['pos'] for c in callstat]), cand_svlen=u32([c['svlen'] for c in callstat]),
             cand_svread=u32([c['svread'] for c in callstat]), cand_refread=u32([c['refread'] for c in callstat]),
             cand_gt_ok=np.array([c['callgt'] != './.' for c in callstat], dtype=np.uint8),
             cand_off=u32(off), mark_read=u32(mark))
    p = _Problem(len(chrom_list), len(callstat), len(mark), len(tags),
                 *[a[n].ctypes.data for n in ('cand_ctg_off', 'read_tag', 'cand_pos', 'cand_svlen', 'cand_svread',
                                              'cand_refread', 'cand_gt_ok', 'cand_off', 'mark_read')],
                 max(svlen_thres, 0), max(suppread_thres, 0))
    pred = np.zeros(len(callstat), dtype=np.uint8)
    ps = np.zeros(len(callstat), dtype=np.uint32)
    _run(p, pred, ps)
    # emission order of :206-228: contig, then PS-class 0,1,2, then file order (only matters for sort ties)
    def n_ps(c):
        return min(len(set(s[2] for s in c['svreadinfo'] if len(s) > 1)), 2)
    order = sorted(np.nonzero(pred)[0], key=lambda i: (ctg[i], n_ps(callstat[i]), i))
    hp = {1: '1|0', 2: '0|1', 3: '1|1'}
    return [dict(ps=int(ps[i]), hp=hp[int(pred[i])], chrom=callstat[i]['chrom'], pos=callstat[i]['pos'],
                 svlen=callstat[i]['svlen'] if callstat[i]['svtype'] in ['INS', 'DUP'] else -callstat[i]['svlen'],
                 svtype=callstat[i]['svtype'], ref=callstat[i]['ref'], alt=callstat[i]['alt']) for i in order]
