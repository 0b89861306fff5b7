# coding=utf-8
"""ctypes binding of include/duet_ingest.h (libduet_ingest.so): native VCF/BAM ingest into the SoA problem and
native row emission.  `NativeIngest.load()` returns None when the input is something the native code does not
vouch for (it says so instead of guessing); callers then use the Python host path, which mirrors upstream's
exceptions."""

import ctypes
import os

import numpy as np

from duet_amd import engine

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libduet_ingest.so')

OK, UNSUPPORTED = 0, 1

EXPORTS = ('duet_ingest_create', 'duet_ingest_destroy', 'duet_ingest_error', 'duet_ingest_add_bam', 'duet_ingest_add_bams',
           'duet_ingest_parse_vcf', 'duet_ingest_parse_vcf_begin', 'duet_ingest_parse_vcf_finish', 'duet_ingest_get_arrays',
           'duet_ingest_emit', 'duet_ingest_free', 'duet_ingest_header',
           'duet_ingest_get_rows', 'duet_ingest_set_extraction', 'duet_ingest_get_marks', 'duet_ingest_bam_has_alignments',
           'duet_ingest_vcf_precount', 'duet_ingest_set_owned', 'duet_ingest_count_kept', 'duet_ingest_emit_blocks',
           'duet_ingest_cand_slots')


class IngestArrays(ctypes.Structure):
    _fields_ = [('n_contigs', ctypes.c_uint32), ('n_cands', ctypes.c_uint32), ('n_marks', ctypes.c_uint32),
                ('n_reads', ctypes.c_uint32)] + \
               [(n, ctypes.c_void_p) for n in ('cand_ctg_off', 'read_off', 'read_tag', 'cand_pos', 'cand_svlen',
                                               'cand_svread', 'cand_refread', 'cand_gt_ok', 'cand_off', 'mark_read')]


class IngestRows(ctypes.Structure):
    _fields_ = [('n_cands', ctypes.c_uint32), ('n_chrom_texts', ctypes.c_uint32), ('max_pos', ctypes.c_uint32),
                ('pool', ctypes.c_void_p), ('pool_bytes', ctypes.c_uint64), ('str_off', ctypes.c_void_p),
                ('cand_chrom_rank', ctypes.c_void_p), ('cand_plus', ctypes.c_void_p)]


class IngestMarks(ctypes.Structure):
    _fields_ = [('n_marks', ctypes.c_uint32), ('n_contigs', ctypes.c_uint32), ('n_reads', ctypes.c_uint32),
                ('depth_bin', ctypes.c_uint32)] + \
               [(n, ctypes.c_void_p) for n in ('mark_contig', 'mark_type', 'mark_pos', 'mark_span', 'mark_read', 'read_tag',
                                               'read_off', 'depth', 'depth_off')]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            return None
        lib = ctypes.CDLL(LIB_PATH)
        lib.duet_ingest_create.restype = ctypes.c_void_p
        lib.duet_ingest_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
        lib.duet_ingest_destroy.argtypes = [ctypes.c_void_p]
        lib.duet_ingest_destroy.restype = None
        lib.duet_ingest_error.restype = ctypes.c_char_p
        lib.duet_ingest_error.argtypes = [ctypes.c_void_p]
        lib.duet_ingest_add_bam.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        lib.duet_ingest_add_bams.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
        lib.duet_ingest_parse_vcf.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        lib.duet_ingest_parse_vcf_begin.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        lib.duet_ingest_parse_vcf_finish.argtypes = [ctypes.c_void_p]
        lib.duet_ingest_get_arrays.argtypes = [ctypes.c_void_p, ctypes.POINTER(IngestArrays)]
        lib.duet_ingest_emit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        lib.duet_ingest_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.POINTER(ctypes.c_uint64)]
        lib.duet_ingest_get_rows.argtypes = [ctypes.c_void_p, ctypes.POINTER(IngestRows)]
        lib.duet_ingest_set_extraction.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        lib.duet_ingest_get_marks.argtypes = [ctypes.c_void_p, ctypes.POINTER(IngestMarks)]
        lib.duet_ingest_free.argtypes = [ctypes.c_void_p]
        lib.duet_ingest_free.restype = None
        lib.duet_ingest_bam_has_alignments.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.duet_ingest_vcf_precount.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.duet_ingest_set_owned.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.duet_ingest_count_kept.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.duet_ingest_cand_slots.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.duet_ingest_emit_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64),
                                                ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def _view(ptr, n, dtype):
    if not n or not ptr:
        return np.zeros(0, dtype=dtype)
    ct = {np.uint8: ctypes.c_uint8, np.uint16: ctypes.c_uint16, np.uint32: ctypes.c_uint32, np.uint64: ctypes.c_uint64}[dtype]
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(n,))


class NativeIngest(object):
    """One ingest: tag dicts per contig + caller VCF -> EfSoA (arrays stay owned by the native object)."""

    def __init__(self, handle, lib, soa, why=None, bam_contigs=()):
        self.handle, self.lib, self.soa, self.why = handle, lib, soa, why
        self.bam_contigs = tuple(bam_contigs)          # contig indices for which a BAM file was found (:19-24)

    def log_lines(self, chrom_list):
        """The per-contig lines upstream logs while it reads the BAMs (sv_phasing_fn.py:30-33: only contigs that have a
        BAM file) and while it joins the VCF (:41-45: every contig) -> (snp_lines, sv_lines)."""
        yes, no = '  signatures extracted from ', '  no signature from '
        snp = [(yes if self.lib.duet_ingest_bam_has_alignments(self.handle, k) == 1 else no) + chrom_list[k]
               for k in self.bam_contigs]
        off = self.soa.cand_ctg_off
        sv = [(yes if off[k + 1] > off[k] else no) + c for k, c in enumerate(chrom_list)]
        return snp, sv

    @staticmethod
    def precount(vcf_path, chrom_list):
        """Records and line bytes of every listed contig in the caller VCF (one memchr pass, nothing tokenised).
        -> (n_records int64[K], n_bytes int64[K]) or None when the library is missing / the file cannot be read."""
        lib = load()
        if lib is None:
            return None
        names = (ctypes.c_char_p * len(chrom_list))(*[c.encode('utf-8') for c in chrom_list])
        h = lib.duet_ingest_create(len(chrom_list), names)
        if not h:
            return None
        try:
            rec = np.zeros(max(len(chrom_list), 1), dtype=np.uint64)
            byt = np.zeros(max(len(chrom_list), 1), dtype=np.uint64)
            if lib.duet_ingest_vcf_precount(h, vcf_path.encode(), rec.ctypes.data, byt.ctypes.data) != OK:
                return None
            return rec[:len(chrom_list)].astype(np.int64), byt[:len(chrom_list)].astype(np.int64)
        finally:
            lib.duet_ingest_destroy(h)

    @classmethod
    def load(cls, vcf_path, sam_home, chrom_list, thread=4, owned=None, plan=None):
        """-> NativeIngest, or None when the native library is missing / declines the input (reason logged by
        the caller through .why of the returned tuple).  owned (sharded runs): the contig indices this rank reads --
        the other listed contigs' BAMs are not opened and their records are dropped at their first token (their rank
        vouches for them); header lines and the contig list stay complete.
        plan (sharded runs, instead of owned): a callable (records per contig, line bytes per contig) -> owned, called with the
        pre-count taken on THIS handle -- the file the pre-count read stays loaded for the parse that follows (a rank that
        pre-counted through NativeIngest.precount read the caller VCF twice)."""
        lib = load()
        if lib is None:
            return None
        names = (ctypes.c_char_p * len(chrom_list))(*[c.encode('utf-8') for c in chrom_list])
        h = lib.duet_ingest_create(len(chrom_list), names)
        if not h:
            return None

        def decline():
            why = lib.duet_ingest_error(h).decode('utf-8', 'replace')
            lib.duet_ingest_destroy(h)
            return cls(None, lib, None, why)

        if plan is not None:
            rec = np.zeros(max(len(chrom_list), 1), dtype=np.uint64)
            byt = np.zeros(max(len(chrom_list), 1), dtype=np.uint64)
            if lib.duet_ingest_vcf_precount(h, vcf_path.encode(), rec.ctypes.data, byt.ctypes.data) != OK:
                return decline()
            owned = plan(rec[:len(chrom_list)].astype(np.int64), byt[:len(chrom_list)].astype(np.int64))
        mine = None
        if owned is not None:
            mine = np.zeros(max(len(chrom_list), 1), dtype=np.uint8)
            mine[list(owned)] = 1
            if lib.duet_ingest_set_owned(h, mine.ctypes.data) != OK:
                return decline()
        # the first half of the VCF parse (read, tokenise, pick the listed contigs' records) needs nothing from the BAMs: it
        # runs on a thread of its own beside the BAM loop (ctypes releases the GIL); the join of the mark names follows.
        # Both halves get `thread` workers: the first half is over after a few milliseconds, and splitting the budget between
        # them cost 10 ms of a 37 ms run at -t 4 (while both run, up to 2 x thread workers exist; DUET_INGEST_STRICT_THREADS=1
        # splits the budget instead and keeps `thread` a strict upper bound).
        import threading
        t_vcf = t_bam = max(1, int(thread))
        if os.environ.get('DUET_INGEST_STRICT_THREADS') == '1':
            t_vcf = max(1, int(thread) // 2)
            t_bam = max(1, int(thread) - t_vcf)
        first_half = threading.Thread(target=lib.duet_ingest_parse_vcf_begin, args=(h, vcf_path.encode(), t_vcf))
        first_half.start()
        with_bam, paths = [], []
        bam_ok = True
        try:
            for k, c in enumerate(chrom_list):
                if mine is not None and not mine[k]:
                    continue
                for cand in (sam_home + 'chr' + c + '.bam', sam_home + c + '.bam'):
                    if os.path.exists(cand):
                        with_bam.append(k)
                        paths.append(cand.encode())
                        break
            if with_bam:
                # every contig has a tag dict of its own (sv_phasing_fn.py:15-29): the contigs go to the workers whole
                # (duet_ingest_add_bams), not one after the other
                ks = (ctypes.c_int * len(with_bam))(*with_bam)
                ps = (ctypes.c_char_p * len(paths))(*paths)
                bam_ok = lib.duet_ingest_add_bams(h, len(with_bam), ks, ps, t_bam) == OK
        finally:
            first_half.join()                       # (also on an exception: nobody else owns the handle it works on)
        if not bam_ok:
            return decline()
        if lib.duet_ingest_parse_vcf_finish(h) != OK:
            return decline()
        a = IngestArrays()
        if lib.duet_ingest_get_arrays(h, ctypes.byref(a)) != OK:
            return decline()
        K, C, M, R = a.n_contigs, a.n_cands, a.n_marks, a.n_reads
        soa = engine.EfSoA(cand_ctg_off=_view(a.cand_ctg_off, K + 1, np.uint32), read_off=_view(a.read_off, K + 1, np.uint32),
                           read_tag=_view(a.read_tag, R, np.uint64), cand_pos=_view(a.cand_pos, C, np.uint32),
                           cand_svlen=_view(a.cand_svlen, C, np.uint32), cand_svread=_view(a.cand_svread, C, np.uint32),
                           cand_refread=_view(a.cand_refread, C, np.uint32), cand_gt_ok=_view(a.cand_gt_ok, C, np.uint8),
                           cand_off=_view(a.cand_off, C + 1, np.uint32), mark_read=_view(a.mark_read, M, np.uint32))
        return cls(h, lib, soa, bam_contigs=with_bam)

    @classmethod
    def extract(cls, sam_home, chrom_list, thread=4, min_sv_size=40, min_mapq=20, depth_bin=1000, only=None):
        """SVIM mode: the haplotagged BAMs alone -> raw SV marks (CIGAR insertions / deletions), tag tables, binned depth.
        -> (NativeIngest, dict(contig, type, pos, span, read, read_tag, read_off, depth, depth_off, depth_bin)) with
        numpy COPIES of the arrays, or (None, reason)."""
        lib = load()
        if lib is None:
            return None, 'libduet_ingest.so is missing'
        names = (ctypes.c_char_p * len(chrom_list))(*[c.encode('utf-8') for c in chrom_list])
        h = lib.duet_ingest_create(len(chrom_list), names)
        if not h:
            return None, 'duet_ingest_create failed'

        def decline():
            why = lib.duet_ingest_error(h).decode('utf-8', 'replace')
            lib.duet_ingest_destroy(h)
            return None, why

        if lib.duet_ingest_set_extraction(h, 1, int(min_sv_size), int(min_mapq), int(depth_bin)) != OK:
            return decline()
        for k, c in enumerate(chrom_list):
            if only is not None and k not in only:
                continue                            # (sharded runs: another rank's contig)
            for cand in (sam_home + 'chr' + c + '.bam', sam_home + c + '.bam'):
                if os.path.exists(cand):
                    if lib.duet_ingest_add_bam(h, k, cand.encode(), int(thread)) != OK:
                        return decline()
                    break
        m = IngestMarks()
        if lib.duet_ingest_get_marks(h, ctypes.byref(m)) != OK:
            return decline()
        M, K, R = m.n_marks, m.n_contigs, m.n_reads
        depth_off = _view(m.depth_off, K + 1, np.uint32).copy()
        out = dict(contig=_view(m.mark_contig, M, np.uint16).copy(), type=_view(m.mark_type, M, np.uint8).copy(),
                   pos=_view(m.mark_pos, M, np.uint32).copy(), span=_view(m.mark_span, M, np.uint32).copy(),
                   read=_view(m.mark_read, M, np.uint32).copy(), read_tag=_view(m.read_tag, R, np.uint64).copy(),
                   read_off=_view(m.read_off, K + 1, np.uint32).copy(),
                   depth=_view(m.depth, int(depth_off[-1]) if K else 0, np.uint32).copy(), depth_off=depth_off,
                   depth_bin=int(m.depth_bin))
        return cls(h, lib, None), out

    def emit(self, pred, ps, include_all_ctgs):
        """Full text of phased_sv.vcf (header + rows) as bytes."""
        pred = np.ascontiguousarray(pred, dtype=np.uint8)
        ps = np.ascontiguousarray(ps, dtype=np.uint32)
        text = ctypes.c_void_p()
        n = ctypes.c_uint64()
        mode = -1 if include_all_ctgs == -1 else (1 if include_all_ctgs else 0)
        rc = self.lib.duet_ingest_emit(self.handle, pred.ctypes.data, ps.ctypes.data, mode,
                                       ctypes.byref(text), ctypes.byref(n))
        if rc != OK:
            raise RuntimeError('duet_ingest_emit: ' + self.lib.duet_ingest_error(self.handle).decode('utf-8', 'replace'))
        try:
            return ctypes.string_at(text.value, n.value)
        finally:
            self.lib.duet_ingest_free(text)

    def count_kept(self, pred):
        """Rows (pred != 0) per CHROM-text slot, slot = 2 * contig + (0: chr<name>, 1: <name>) -> int64[2K]"""
        K = self.soa.n_contigs
        pred = np.ascontiguousarray(pred, dtype=np.uint8)
        kept = np.zeros(max(2 * K, 1), dtype=np.uint64)
        if self.lib.duet_ingest_count_kept(self.handle, pred.ctypes.data, kept.ctypes.data) != OK:
            raise RuntimeError('duet_ingest_count_kept failed')
        return kept[:2 * K].astype(np.int64)

    def cand_slots(self):
        """Every candidate's CHROM-text slot (2 * contig + spelling) -> uint32[C]"""
        slot = np.zeros(max(self.soa.n_cands, 1), dtype=np.uint32)
        if self.lib.duet_ingest_cand_slots(self.handle, slot.ctypes.data) != OK:
            raise RuntimeError('duet_ingest_cand_slots failed')
        return slot[:self.soa.n_cands]

    def emit_blocks(self, pred, ps, id_base):
        """This ingest's rows with the rows of slot s numbered id_base[s], id_base[s] + 1, ...
        -> (bytes, slot_off int64[2K], slot_len int64[2K])"""
        K = self.soa.n_contigs
        pred = np.ascontiguousarray(pred, dtype=np.uint8)
        ps = np.ascontiguousarray(ps, dtype=np.uint32)
        base = np.ascontiguousarray(id_base, dtype=np.uint64)
        off = np.zeros(max(2 * K, 1), dtype=np.uint64)
        ln = np.zeros(max(2 * K, 1), dtype=np.uint64)
        text = ctypes.c_void_p()
        n = ctypes.c_uint64()
        rc = self.lib.duet_ingest_emit_blocks(self.handle, pred.ctypes.data, ps.ctypes.data, base.ctypes.data,
                                              ctypes.byref(text), ctypes.byref(n), off.ctypes.data, ln.ctypes.data)
        if rc != OK:
            raise RuntimeError('duet_ingest_emit_blocks: ' + self.lib.duet_ingest_error(self.handle).decode('utf-8', 'replace'))
        try:
            return ctypes.string_at(text.value, n.value), off[:2 * K].astype(np.int64), ln[:2 * K].astype(np.int64)
        finally:
            self.lib.duet_ingest_free(text)

    def emit_rows(self, pred, ps):
        """The data rows alone (no header) as bytes."""
        return self.emit(pred, ps, -1)

    def header(self, include_all_ctgs):
        """The header lines of phased_sv.vcf as bytes."""
        text = ctypes.c_void_p()
        n = ctypes.c_uint64()
        rc = self.lib.duet_ingest_header(self.handle, 1 if include_all_ctgs else 0, ctypes.byref(text), ctypes.byref(n))
        if rc != OK:
            raise RuntimeError('duet_ingest_header: ' + self.lib.duet_ingest_error(self.handle).decode('utf-8', 'replace'))
        try:
            return ctypes.string_at(text.value, n.value)
        finally:
            self.lib.duet_ingest_free(text)

    def rows(self):
        """What the device-side row emission needs (views owned by the native object), or None when it declines:
        dict(pool u8[], str_off u32[4C+1], chrom_rank u16[C], plus u8[C], n_chrom_texts, max_pos)."""
        r = IngestRows()
        if self.lib.duet_ingest_get_rows(self.handle, ctypes.byref(r)) != OK:
            return None
        C = r.n_cands
        return dict(pool=_view(r.pool, r.pool_bytes, np.uint8), str_off=_view(r.str_off, 4 * C + 1, np.uint32),
                    chrom_rank=_view(r.cand_chrom_rank, C, np.uint16), plus=_view(r.cand_plus, C, np.uint8),
                    n_chrom_texts=r.n_chrom_texts, max_pos=r.max_pos, pool_bytes=r.pool_bytes)

    def close(self):
        if self.handle:
            self.soa = None
            self.lib.duet_ingest_destroy(self.handle)
            self.handle = None

    def close_in_background(self):
        """Hand the native object to a thread that frees it (round 6: at configs[2]'s size the name tables, the tag words and the
        565 MB VCF buffer take 75 ms to give back -- nothing the caller waits for: the rows are already out).  The thread is not a
        daemon: the interpreter's exit waits for it."""
        if self.handle:
            import threading
            h, self.handle, self.soa = self.handle, None, None
            threading.Thread(target=self.lib.duet_ingest_destroy, args=(h,)).start()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
