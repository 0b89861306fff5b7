# coding=utf-8
"""TEST INFRASTRUCTURE ONLY -- ctypes loader of oracle/_build/libef_oracle.so (the scalar C
restatement in ef_oracle.c). Same rules as ef_oracle.py: never imported by the product path."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_build', 'libef_oracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = ctypes.CDLL(SO)
        _lib.duet_oracle_ef.restype = ctypes.c_int
        _lib.duet_oracle_ef.argtypes = [ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_void_p] * 9 + \
            [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data if a.size else 0)


def ef(soa, svlen_thres, suppread_thres):
    """soa: any object with the EfSoA attributes. -> (rc, pred u8[C], ps u32[C])"""
    lib = load()
    C = soa.n_cands
    pred = np.zeros(max(C, 1), dtype=np.uint8)
    ps = np.zeros(max(C, 1), dtype=np.uint32)
    clamp = lambda v: 0 if v < 0 else min(int(v), 0xFFFFFFFF)
    rc = lib.duet_oracle_ef(soa.n_contigs, C, _p(soa.read_tag), _p(soa.cand_ctg_off), _p(soa.cand_pos),
                            _p(soa.cand_svlen), _p(soa.cand_svread), _p(soa.cand_refread), _p(soa.cand_gt_ok),
                            _p(soa.cand_off), _p(soa.mark_read), clamp(svlen_thres), clamp(suppread_thres),
                            _p(pred), _p(ps))
    return rc, pred[:C], ps[:C]


def cluster(contig, mtype, pos, span, max_dist=0.9, part_gap=1000, part_max=100, normalizer=900.0):
    """A0 oracle (cluster_oracle.c). Arrays: contig u16[M], mtype u8[M], pos u32[M], span u32[M].
    -> dict(order u32[M], cand_off u32[N+1], cand_contig u16[N], cand_type u8[N], cand_pos u32[N], cand_span u32[N])"""
    lib = load()
    fn = lib.duet_oracle_cluster
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 4 + [ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32,
                                                                 ctypes.c_double] + [ctypes.c_void_p] * 7
    contig = np.ascontiguousarray(contig, dtype=np.uint16)
    mtype = np.ascontiguousarray(mtype, dtype=np.uint8)
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    span = np.ascontiguousarray(span, dtype=np.uint32)
    M = len(pos)
    order = np.zeros(max(M, 1), dtype=np.uint32)
    off = np.zeros(M + 1, dtype=np.uint32)
    n = ctypes.c_uint32(0)
    cc = np.zeros(max(M, 1), dtype=np.uint16)
    ct = np.zeros(max(M, 1), dtype=np.uint8)
    cp = np.zeros(max(M, 1), dtype=np.uint32)
    cs = np.zeros(max(M, 1), dtype=np.uint32)
    rc = fn(M, _p(contig), _p(mtype), _p(pos), _p(span), float(max_dist), int(part_gap), int(part_max),
            float(normalizer), _p(order), _p(off), ctypes.byref(n), _p(cc), _p(ct), _p(cp), _p(cs))
    if rc:
        raise RuntimeError('duet_oracle_cluster failed: %d' % rc)
    N = n.value
    return dict(order=order[:M], cand_off=off[:N + 1], cand_contig=cc[:N], cand_type=ct[:N], cand_pos=cp[:N],
                cand_span=cs[:N])
