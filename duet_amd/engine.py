# coding=utf-8
"""Structure-of-arrays form of one step-E/F problem and the call into the HIP library.

The layout is the one include/duet_ef.h documents: candidates in callset order (contig-major in
chrom-list order, file order inside a contig -- generate_callinfo, sv_phasing_fn.py:50-67), marks in
CSR form, each mark an index into the concatenated per-contig read-tag table or MARK_ABSENT.
"""

import numpy as np

from duet_amd import _lib

MARK_ABSENT = _lib.MARK_ABSENT
PS_LIMIT = 0xFFFFFFFE          # 0xFFFFFFFF is the device's "empty" marker


class EfSoA(object):
    """Host arrays of one E/F problem (all C-contiguous, exact dtypes of the C ABI)."""

    FIELDS = (('cand_ctg_off', np.uint32), ('read_tag', np.uint64), ('cand_pos', np.uint32),
              ('cand_svlen', np.uint32), ('cand_svread', np.uint32), ('cand_refread', np.uint32),
              ('cand_gt_ok', np.uint8), ('cand_off', np.uint32), ('mark_read', np.uint32))

    def __init__(self, **kw):
        for name, dt in self.FIELDS:
            src = np.asarray(kw[name])
            if src.dtype != dt and src.size and src.dtype.kind in 'iu':
                # the C ABI counts marks, candidates and reads in 32 bits (include/duet_ef.h): a wider host array must not
                # wrap silently when it is narrowed
                lim = int(np.iinfo(dt).max)
                if int(src.min()) < 0 or int(src.max()) > lim:
                    raise ValueError('%s does not fit the %d-bit arrays of the C ABI' % (name, 8 * np.dtype(dt).itemsize))
            a = np.ascontiguousarray(src, dtype=dt)
            setattr(self, name, a)
        self.read_off = np.ascontiguousarray(kw.get('read_off', [0, len(self.read_tag)]), dtype=np.uint32)
        self.validate()

    @property
    def n_contigs(self):
        return len(self.cand_ctg_off) - 1

    @property
    def n_cands(self):
        return len(self.cand_pos)

    @property
    def n_marks(self):
        return len(self.mark_read)

    @property
    def n_reads(self):
        return len(self.read_tag)

    def algorithmic_bytes(self):
        """B_EF = 12*M + 27*C + 8*R  (SURVEY.md section 8d)."""
        return 12 * self.n_marks + 27 * self.n_cands + 8 * self.n_reads

    def validate(self):
        C = self.n_cands
        if len(self.cand_ctg_off) < 1 or self.cand_ctg_off[0] != 0 or self.cand_ctg_off[-1] != C:
            raise ValueError('cand_ctg_off must run from 0 to the number of candidates')
        for name in ('cand_svlen', 'cand_svread', 'cand_refread', 'cand_gt_ok'):
            if len(getattr(self, name)) != C:
                raise ValueError(name + ' has the wrong length')
        if len(self.cand_off) != C + 1 or (C and self.cand_off[0] != 0) or \
                (C and int(self.cand_off[-1]) != self.n_marks):
            raise ValueError('cand_off must be a CSR offset array over mark_read')
        if C and np.any(self.cand_off[1:] <= self.cand_off[:-1]):
            raise ValueError('every candidate needs at least one mark (an empty RNAMES list still '
                             'yields one empty name upstream)')
        if self.n_marks:
            live = self.mark_read[self.mark_read != MARK_ABSENT]
            if live.size and int(live.max()) >= self.n_reads:
                raise ValueError('mark_read index beyond the tag table')


def pack_tags(hap, pc, ps):
    """HP/PC/PS integers -> device tag words. hap other than 1/2 becomes 3 ("other": upstream never
    counts it as a haplotype vote in the one-PS branch and raises KeyError in the multi-PS branch);
    pc saturates at 2**30-2 (only `pc <= 8100` and sums of such values are ever used)."""
    hap = np.asarray(hap, dtype=np.int64)
    pc = np.asarray(pc, dtype=np.int64)
    ps = np.asarray(ps, dtype=np.int64)
    if pc.size and int(pc.min()) < 0:
        raise ValueError('negative PC tag')
    if ps.size and (int(ps.min()) < 0 or int(ps.max()) > PS_LIMIT):
        raise ValueError('PS tag outside [0, 2**32-2]')
    code = np.where((hap == 1) | (hap == 2), hap, 3).astype(np.uint64)
    pcc = np.minimum(pc, (1 << 30) - 2).astype(np.uint64)
    return (code << np.uint64(62)) | (pcc << np.uint64(32)) | ps.astype(np.uint64)


def soa_from_synth(contigs):
    """Direct SoA from duet_amd.synth contigs (no text round trip), contigs in the given order."""
    parts = [c.soa_parts() for c in contigs]
    read_off = np.zeros(len(parts) + 1, dtype=np.int64)
    ctg_off = np.zeros(len(parts) + 1, dtype=np.int64)
    mark_base = 0
    mr, co = [], [np.zeros(1, dtype=np.int64)]
    for i, p in enumerate(parts):
        read_off[i + 1] = read_off[i] + len(p['read_tag'])
        ctg_off[i + 1] = ctg_off[i] + len(p['cand_pos'])
        m = p['mark_read'].copy()
        m = np.where(m >= 0, m + read_off[i], MARK_ABSENT)
        mr.append(m)
        co.append(p['cand_off'][1:] + mark_base)
        mark_base += int(p['cand_off'][-1])
    cat = lambda k, dt: np.concatenate([p[k] for p in parts]).astype(dt) if parts else np.zeros(0, dtype=dt)
    return EfSoA(cand_ctg_off=ctg_off, read_off=read_off, read_tag=cat('read_tag', np.uint64),
                 cand_pos=cat('cand_pos', np.uint32), cand_svlen=cat('cand_svlen_abs', np.uint32),
                 cand_svread=cat('cand_svread', np.uint32), cand_refread=cat('cand_refread', np.uint32),
                 cand_gt_ok=cat('cand_gt_ok', np.uint8), cand_off=np.concatenate(co),
                 mark_read=np.concatenate(mr) if mr else np.zeros(0, dtype=np.uint32))


_default_ctx = {}


def default_context(device_id=0):
    """Process-wide context per device (created on first use; raises when no MI355X / no library)."""
    ctx = _default_ctx.get(device_id)
    if ctx is None:
        ctx = _lib.Context(device_id)
        _default_ctx[device_id] = ctx
    return ctx


def run_ef(soa, svlen_thres, suppread_thres, ctx=None, want_stats=False):
    """Filter + PS-class + seed sets + vote + decision for every candidate of `soa` on the GPU.
    -> pred u8[C] (0 filtered / 1 '1|0' / 2 '0|1' / 3 '1|1'), ps u32[C]."""
    if ctx is None:
        ctx = default_context()
    return ctx.run_host(soa, svlen_thres, suppread_thres, want_stats=want_stats)
