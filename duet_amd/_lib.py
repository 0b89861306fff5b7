# coding=utf-8
"""ctypes binding of include/duet_ef.h (libduet_ef.so, hand-written HIP for gfx950).

There is deliberately NO CPU fallback: if the shared library is missing or no MI355X is visible,
every entry point raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
(or `make -C duet_amd/csrc`).
"""

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libduet_ef.so')

DUET_OK = 0
DUET_ERR_INVALID = -1
DUET_ERR_DIV_ZERO = -5
DUET_ERR_TIMEOUT = -6
MARK_ABSENT = 0xFFFFFFFF
N_KERNELS = 3
KERNEL_NAMES = ('ef_classify', 'ef_seed_sort', 'ef_finalize')

# every symbol include/duet_ef.h declares (checked by tests/test_abi.py)
EXPORTS = ('duet_abi_version', 'duet_ctx_create', 'duet_ctx_destroy', 'duet_last_error',
           'duet_ctx_set_profiling', 'duet_ctx_set_debug', 'duet_ef_run_device', 'duet_ef_check', 'duet_ef_run_host',
           'duet_ef_profile_collect', 'duet_ef_get_seed_ps', 'duet_cluster_run_device', 'duet_cluster_run_host', 'duet_svim_phase_device', 'duet_svim_phase_host', 'duet_rows_run_device',
           'duet_ef_rows_run_host', 'duet_eval_run_host', 'duet_comm_unique_id', 'duet_comm_create', 'duet_comm_allgather_device',
           'duet_comm_allgather_host', 'duet_comm_destroy', 'duet_comm_set_timeout', 'duet_comm_block_bytes',
           'duet_comm_ef_allgather', 'duet_comm_rccl_version', 'duet_comm_info', 'duet_comm_selftest')


class EfProblem(ctypes.Structure):
    _fields_ = [('n_contigs', ctypes.c_uint32), ('n_cands', ctypes.c_uint32), ('n_marks', ctypes.c_uint32),
                ('n_reads', ctypes.c_uint32),
                ('cand_ctg_off', ctypes.c_void_p), ('read_tag', ctypes.c_void_p), ('cand_pos', ctypes.c_void_p),
                ('cand_svlen', ctypes.c_void_p), ('cand_svread', ctypes.c_void_p), ('cand_refread', ctypes.c_void_p),
                ('cand_gt_ok', ctypes.c_void_p), ('cand_off', ctypes.c_void_p), ('mark_read', ctypes.c_void_p),
                ('svlen_thres', ctypes.c_uint32), ('suppread_thres', ctypes.c_uint32)]


class EfStats(ctypes.Structure):
    _fields_ = [('algorithmic_bytes', ctypes.c_uint64), ('n_seed_ps', ctypes.c_uint32),
                ('n_profiled_runs', ctypes.c_uint32), ('kernel_ms', ctypes.c_float * N_KERNELS),
                ('total_ms', ctypes.c_float)]


class ClusterProblem(ctypes.Structure):
    _fields_ = [('n_marks', ctypes.c_uint32), ('part_gap', ctypes.c_uint32), ('part_max', ctypes.c_uint32),
                ('n_contigs_hint', ctypes.c_uint32), ('n_types_hint', ctypes.c_uint32), ('max_pos_hint', ctypes.c_uint32),
                ('max_span_hint', ctypes.c_uint32), ('reserved', ctypes.c_uint32),
                ('max_dist', ctypes.c_double), ('normalizer', ctypes.c_double),
                ('mark_contig', ctypes.c_void_p), ('mark_type', ctypes.c_void_p), ('mark_pos', ctypes.c_void_p),
                ('mark_span', ctypes.c_void_p)]


class ClusterResult(ctypes.Structure):
    _fields_ = [('order', ctypes.c_void_p), ('cand_off', ctypes.c_void_p), ('cand_contig', ctypes.c_void_p),
                ('cand_type', ctypes.c_void_p), ('cand_pos', ctypes.c_void_p), ('cand_span', ctypes.c_void_p),
                ('n_cands', ctypes.c_void_p)]


class SvimProblem(ctypes.Structure):
    _fields_ = [('marks', ClusterProblem), ('mark_read', ctypes.c_void_p), ('read_tag', ctypes.c_void_p),
                ('n_reads', ctypes.c_uint32), ('n_contigs', ctypes.c_uint32), ('depth', ctypes.c_void_p),
                ('depth_off', ctypes.c_void_p), ('depth_bin', ctypes.c_uint32), ('svlen_thres', ctypes.c_uint32),
                ('suppread_thres', ctypes.c_uint32), ('reserved', ctypes.c_uint32)]


class RowsProblem(ctypes.Structure):
    _fields_ = [('n_contigs', ctypes.c_uint32), ('n_cands', ctypes.c_uint32), ('cand_ctg_off', ctypes.c_void_p),
                ('pred', ctypes.c_void_p), ('ps', ctypes.c_void_p), ('cand_pos', ctypes.c_void_p),
                ('cand_svlen', ctypes.c_void_p), ('cand_plus', ctypes.c_void_p), ('cand_chrom_rank', ctypes.c_void_p),
                ('n_chrom_texts', ctypes.c_uint32), ('max_pos', ctypes.c_uint32), ('pool', ctypes.c_void_p),
                ('pool_bytes', ctypes.c_uint64), ('str_off', ctypes.c_void_p), ('cand_off', ctypes.c_void_p),
                ('mark_read', ctypes.c_void_p), ('read_tag', ctypes.c_void_p)]


class EvalProblem(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ('n_base', 'n_calls', 'n_groups', 'n_keys', 'n_base_uid', 'n_call_uid', 'refdist',
                                               'reserved')] + [('ratio', ctypes.c_double)] + \
               [(n, ctypes.c_void_p) for n in ('base_off', 'base_pos', 'base_len', 'base_uid', 'base_hp', 'call_key', 'call_pos',
                                               'call_len', 'call_uid', 'call_group', 'call_hp')]


class EvalCounts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ('call_tp', 'base_tp', 'call_gt', 'base_gt', 'call_hp', 'base_hp')]


class DuetLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load libduet_ef.so (once). Raises DuetLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DuetLibraryError('%s is missing: the HIP extension has not been built '
                               '(run __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64 and loads
    # them by unversioned name, so if this library pulled in /opt/rocm's copy first, torch would load a
    # second runtime and find no GPU.  Importing torch first makes the loader bind our DT_NEEDED
    # libamdhip64.so.7 to the copy torch already mapped (same SONAME).
    # (DUET_NO_TORCH=1: a process that will never import torch -- the rank processes of `duet --gpus N`, duet_amd/multi.py --
    # skips this: the library then binds /opt/rocm's runtime, as RCCL does)
    if os.environ.get('DUET_NO_TORCH') != '1':
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.duet_abi_version.restype = ctypes.c_int
    lib.duet_ctx_create.restype = ctypes.c_void_p
    lib.duet_ctx_create.argtypes = [ctypes.c_int]
    lib.duet_ctx_destroy.restype = None
    lib.duet_ctx_destroy.argtypes = [ctypes.c_void_p]
    lib.duet_last_error.restype = ctypes.c_char_p
    lib.duet_last_error.argtypes = [ctypes.c_void_p]
    lib.duet_ctx_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.duet_ctx_set_debug.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.duet_ef_run_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(EfProblem), ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]
    lib.duet_ef_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.duet_ef_run_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(EfProblem), ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.POINTER(EfStats)]
    lib.duet_ef_profile_collect.argtypes = [ctypes.c_void_p, ctypes.POINTER(EfStats)]
    lib.duet_ef_get_seed_ps.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    lib.duet_cluster_run_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(ClusterProblem),
                                            ctypes.POINTER(ClusterResult), ctypes.c_void_p]
    lib.duet_cluster_run_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(ClusterProblem), ctypes.POINTER(ClusterResult)]
    lib.duet_rows_run_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(RowsProblem), ctypes.c_void_p, ctypes.c_uint64,
                                         ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p]
    lib.duet_ef_rows_run_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(EfProblem), ctypes.POINTER(RowsProblem), ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    lib.duet_eval_run_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(EvalProblem), ctypes.POINTER(EvalCounts)]
    lib.duet_svim_phase_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(SvimProblem), ctypes.POINTER(ClusterResult),
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p]
    lib.duet_svim_phase_host.argtypes = [ctypes.c_void_p, ctypes.POINTER(SvimProblem), ctypes.POINTER(ClusterResult),
                                         ctypes.c_void_p, ctypes.c_void_p]
    lib.duet_comm_unique_id.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.duet_comm_create.restype = ctypes.c_void_p
    lib.duet_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.duet_comm_allgather_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    lib.duet_comm_allgather_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    lib.duet_comm_rccl_version.argtypes = [ctypes.c_void_p]
    lib.duet_comm_set_timeout.argtypes = [ctypes.c_void_p, ctypes.c_double]
    lib.duet_comm_block_bytes.restype = ctypes.c_uint64
    lib.duet_comm_block_bytes.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.duet_comm_ef_allgather.argtypes = [ctypes.c_void_p, ctypes.POINTER(EfProblem), ctypes.c_void_p, ctypes.c_uint32,
                                           ctypes.c_uint32, ctypes.c_void_p]
    lib.duet_comm_info.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 5
    lib.duet_comm_selftest.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.duet_comm_destroy.restype = None
    lib.duet_comm_destroy.argtypes = [ctypes.c_void_p]
    _lib = lib
    return lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None and a.size else ctypes.c_void_p(0)


CONTEXTS_CREATED = 0            # > 0: this process has initialised HIP (duet_amd/launch.py refuses to start ranks from it)


class Context(object):
    """One duet_ctx on one HIP device."""

    def __init__(self, device_id=0):
        global CONTEXTS_CREATED
        self.lib = load()
        self.handle = self.lib.duet_ctx_create(int(device_id))
        if not self.handle:
            raise DuetLibraryError('duet_ctx_create(%d) failed: %s' % (
                device_id, self.lib.duet_last_error(None).decode('utf-8', 'replace')))
        self.device_id = int(device_id)
        CONTEXTS_CREATED += 1

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.duet_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self.lib.duet_last_error(self.handle).decode('utf-8', 'replace')

    def _raise(self, rc):
        msg = self.lib.duet_last_error(self.handle).decode('utf-8', 'replace')
        if rc == DUET_ERR_DIV_ZERO:
            raise ZeroDivisionError('division by zero')        # what upstream raises (sv_phasing_fn.py:123)
        raise DuetLibraryError('duet_ef call failed (%d): %s' % (rc, msg))

    def set_profiling(self, mode):
        """0 off, 1 (or True) events around ef_classify only, 2 around every kernel."""
        rc = self.lib.duet_ctx_set_profiling(self.handle, int(mode))
        if rc:
            self._raise(rc)

    def set_debug(self, flags):
        rc = self.lib.duet_ctx_set_debug(self.handle, int(flags))
        if rc:
            self._raise(rc)

    # -- host arrays ---------------------------------------------------------------------------
    def run_host(self, soa, svlen_thres, suppread_thres, want_stats=False):
        """soa: engine.EfSoA with numpy arrays. Returns (pred u8[C], ps u32[C][, stats])."""
        prob, keep = problem_from_arrays(soa, svlen_thres, suppread_thres)
        C = soa.n_cands
        pred = np.zeros(C, dtype=np.uint8)
        ps = np.zeros(C, dtype=np.uint32)
        stats = EfStats()
        # (statistics only when asked for: the seed count behind a two-launch run costs an ef_seed_sort launch of its own)
        rc = self.lib.duet_ef_run_host(self.handle, ctypes.byref(prob), _ptr(pred), _ptr(ps), ctypes.byref(stats) if want_stats else None)
        del keep
        if rc:
            self._raise(rc)
        return (pred, ps, stats) if want_stats else (pred, ps)

    # -- device pointers (torch tensors' data_ptr()) -----------------------------------------------
    def run_device(self, prob, out_pred_ptr, out_ps_ptr, stream=0):
        rc = self.lib.duet_ef_run_device(self.handle, ctypes.byref(prob), ctypes.c_void_p(out_pred_ptr),
                                         ctypes.c_void_p(out_ps_ptr), ctypes.c_void_p(stream))
        if rc:
            self._raise(rc)

    def check(self, stream=0):
        rc = self.lib.duet_ef_check(self.handle, ctypes.c_void_p(stream))
        if rc:
            self._raise(rc)

    def profile_collect(self):
        st = EfStats()
        rc = self.lib.duet_ef_profile_collect(self.handle, ctypes.byref(st))
        if rc:
            self._raise(rc)
        return st

    # -- stage A0: span-position clustering ---------------------------------------------------------
    def ef_rows_host(self, soa, rows, svlen_thres, suppread_thres):
        """duet_ef_rows_run_host: host arrays in (an EfSoA and NativeIngest.rows()), the text of the data rows out.
        -> (bytes, number of rows)"""
        p, keep = problem_from_arrays(soa, svlen_thres, suppread_thres)
        pool = np.ascontiguousarray(rows['pool'], dtype=np.uint8)
        str_off = np.ascontiguousarray(rows['str_off'], dtype=np.uint32)
        rank = np.ascontiguousarray(rows['chrom_rank'], dtype=np.uint16)
        plus = np.ascontiguousarray(rows['plus'], dtype=np.uint8)
        r = RowsProblem()
        r.n_contigs, r.n_cands = soa.n_contigs, soa.n_cands
        r.n_chrom_texts, r.max_pos = int(rows['n_chrom_texts']), int(rows['max_pos'])
        r.pool, r.pool_bytes, r.str_off = pool.ctypes.data, int(rows['pool_bytes']), str_off.ctypes.data
        r.cand_chrom_rank, r.cand_plus = rank.ctypes.data, plus.ctypes.data
        cap = int(rows['pool_bytes']) + 96 * soa.n_cands + 64
        out = np.empty(cap, dtype=np.uint8)
        n = ctypes.c_uint64(0)
        n_rows = ctypes.c_uint32(0)
        rc = self.lib.duet_ef_rows_run_host(self.handle, ctypes.byref(p), ctypes.byref(r), out.ctypes.data, ctypes.c_uint64(cap),
                                            ctypes.byref(n), ctypes.byref(n_rows))
        del keep
        if rc:
            self._raise(rc)
        return out[:n.value].tobytes(), n_rows.value

    def rows_device(self, prob, out_ptr, cap, stream):
        """duet_rows_run_device: -> (bytes written, rows)."""
        n = ctypes.c_uint64(0)
        rows = ctypes.c_uint32(0)
        rc = self.lib.duet_rows_run_device(self.handle, ctypes.byref(prob), ctypes.c_void_p(out_ptr), ctypes.c_uint64(cap),
                                           ctypes.byref(n), ctypes.byref(rows), ctypes.c_void_p(stream))
        if rc:
            self._raise(rc)
        return n.value, rows.value

    def cluster_host(self, contig, mtype, pos, span, max_dist=0.9, part_gap=1000, part_max=100, normalizer=900.0,
                     hints=True):
        """Cluster SV marks (host numpy arrays) into candidates on the GPU.
        -> dict(order, cand_off, cand_contig, cand_type, cand_pos, cand_span) trimmed to the candidate count."""
        contig = np.ascontiguousarray(contig, dtype=np.uint16)
        mtype = np.ascontiguousarray(mtype, dtype=np.uint8)
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        span = np.ascontiguousarray(span, dtype=np.uint32)
        M = len(pos)
        prob = ClusterProblem()
        prob.n_marks, prob.part_gap, prob.part_max = M, int(part_gap), int(part_max)
        if hints and M:
            prob.n_contigs_hint = int(contig.max()) + 1
            prob.n_types_hint = int(mtype.max()) + 1
            prob.max_pos_hint = int(pos.max())
            prob.max_span_hint = max(int(span.max()), 1)
        prob.max_dist, prob.normalizer = float(max_dist), float(normalizer)
        prob.mark_contig, prob.mark_type, prob.mark_pos, prob.mark_span = [
            a.ctypes.data if a.size else None for a in (contig, mtype, pos, span)]
        out = dict(order=np.zeros(max(M, 1), dtype=np.uint32), cand_off=np.zeros(M + 1, dtype=np.uint32),
                   cand_contig=np.zeros(max(M, 1), dtype=np.uint16), cand_type=np.zeros(max(M, 1), dtype=np.uint8),
                   cand_pos=np.zeros(max(M, 1), dtype=np.uint32), cand_span=np.zeros(max(M, 1), dtype=np.uint32))
        n = ctypes.c_uint32(0)
        res = ClusterResult()
        for k in ('order', 'cand_off', 'cand_contig', 'cand_type', 'cand_pos', 'cand_span'):
            setattr(res, k, out[k].ctypes.data)
        res.n_cands = ctypes.addressof(n)
        rc = self.lib.duet_cluster_run_host(self.handle, ctypes.byref(prob), ctypes.byref(res))
        if rc:
            self._raise(rc)
        N = n.value
        return dict(order=out['order'][:M], cand_off=out['cand_off'][:N + 1], cand_contig=out['cand_contig'][:N],
                    cand_type=out['cand_type'][:N], cand_pos=out['cand_pos'][:N], cand_span=out['cand_span'][:N])

    def svim_host(self, marks, read_tag, depth, depth_off, depth_bin, svlen_thres, suppread_thres, max_dist=0.9, part_gap=1000,
                  part_max=100, normalizer=900.0):
        """The fused SVIM-mode pipeline on host arrays (duet_svim_phase_host): raw marks dict(contig, type, pos, span, read) ->
        dict(cand_off, cand_contig, cand_type, cand_pos, cand_span, pred, ps) trimmed to the candidate count."""
        arr = {k: np.ascontiguousarray(marks[k], dtype=dt) for k, dt in (('contig', np.uint16), ('type', np.uint8), ('pos', np.uint32),
                                                                        ('span', np.uint32), ('read', np.uint32))}
        read_tag = np.ascontiguousarray(read_tag, dtype=np.uint64)
        depth = np.ascontiguousarray(depth, dtype=np.uint32)
        depth_off = np.ascontiguousarray(depth_off, dtype=np.uint32)
        M = len(arr['pos'])
        p = SvimProblem()
        p.marks.n_marks, p.marks.part_gap, p.marks.part_max = M, int(part_gap), int(part_max)
        p.marks.max_dist, p.marks.normalizer = float(max_dist), float(normalizer)
        if M:
            p.marks.n_contigs_hint = int(arr['contig'].max()) + 1
            p.marks.n_types_hint = int(arr['type'].max()) + 1
            p.marks.max_pos_hint = int(arr['pos'].max())
            p.marks.max_span_hint = max(int(arr['span'].max()), 1)
        p.marks.mark_contig, p.marks.mark_type = arr['contig'].ctypes.data, arr['type'].ctypes.data
        p.marks.mark_pos, p.marks.mark_span = arr['pos'].ctypes.data, arr['span'].ctypes.data
        p.mark_read = arr['read'].ctypes.data
        p.read_tag = read_tag.ctypes.data if read_tag.size else None
        p.n_reads, p.n_contigs = len(read_tag), len(depth_off) - 1
        p.depth = depth.ctypes.data if depth.size else None
        p.depth_off = depth_off.ctypes.data
        p.depth_bin, p.svlen_thres, p.suppread_thres = int(depth_bin), int(svlen_thres), int(suppread_thres)
        out = dict(cand_off=np.zeros(M + 1, dtype=np.uint32), cand_contig=np.zeros(max(M, 1), dtype=np.uint16),
                   cand_type=np.zeros(max(M, 1), dtype=np.uint8), cand_pos=np.zeros(max(M, 1), dtype=np.uint32),
                   cand_span=np.zeros(max(M, 1), dtype=np.uint32))
        pred, ps = np.zeros(max(M, 1), dtype=np.uint8), np.zeros(max(M, 1), dtype=np.uint32)
        n = ctypes.c_uint32(0)
        res = ClusterResult()
        for k in out:
            setattr(res, k, out[k].ctypes.data)
        res.order = None
        res.n_cands = ctypes.addressof(n)
        rc = self.lib.duet_svim_phase_host(self.handle, ctypes.byref(p), ctypes.byref(res), _ptr(pred), _ptr(ps))
        if rc:
            self._raise(rc)
        N = n.value
        return dict(cand_off=out['cand_off'][:N + 1], cand_contig=out['cand_contig'][:N], cand_type=out['cand_type'][:N],
                    cand_pos=out['cand_pos'][:N], cand_span=out['cand_span'][:N], pred=pred[:N], ps=ps[:N])

    def eval_counts(self, arrays, refdist, ratio):
        """duet_eval_run_host: `arrays` = dict of the flat host arrays (duet_amd/evaluation.py: flatten) -> EvalCounts."""
        p = EvalProblem()
        keep = {}
        for name, dt in (('base_off', np.uint32), ('base_pos', np.uint32), ('base_len', np.uint32), ('base_uid', np.uint32),
                         ('base_hp', np.uint8), ('call_key', np.uint32), ('call_pos', np.uint32), ('call_len', np.uint32),
                         ('call_uid', np.uint32), ('call_group', np.uint32), ('call_hp', np.uint8)):
            keep[name] = np.ascontiguousarray(arrays[name], dtype=dt)
            setattr(p, name, keep[name].ctypes.data if keep[name].size else None)
        p.n_base, p.n_calls = len(keep['base_pos']), len(keep['call_pos'])
        p.n_keys, p.n_groups = len(keep['base_off']) - 1, int(arrays['n_groups'])
        p.n_base_uid, p.n_call_uid = int(arrays['n_base_uid']), int(arrays['n_call_uid'])
        p.refdist, p.ratio = clamp_u32(refdist), float(ratio)
        out = EvalCounts()
        rc = self.lib.duet_eval_run_host(self.handle, ctypes.byref(p), ctypes.byref(out))
        del keep
        if rc:
            self._raise(rc)
        return out

    def seed_ps(self, contig, cap=1 << 20):
        out = np.zeros(cap, dtype=np.uint32)
        n = self.lib.duet_ef_get_seed_ps(self.handle, int(contig), _ptr(out), cap)
        if n < 0:
            self._raise(n)
        return out[:min(n, cap)].copy()


def clamp_u32(v):
    v = int(v)
    return 0 if v < 0 else (0xFFFFFFFF if v > 0xFFFFFFFF else v)


def problem_from_arrays(soa, svlen_thres, suppread_thres):
    """EfProblem over host numpy arrays; returns (problem, keepalive)."""
    p = EfProblem()
    p.n_contigs, p.n_cands, p.n_marks, p.n_reads = soa.n_contigs, soa.n_cands, soa.n_marks, soa.n_reads
    keep = [soa.cand_ctg_off, soa.read_tag, soa.cand_pos, soa.cand_svlen, soa.cand_svread, soa.cand_refread,
            soa.cand_gt_ok, soa.cand_off, soa.mark_read]
    (p.cand_ctg_off, p.read_tag, p.cand_pos, p.cand_svlen, p.cand_svread, p.cand_refread, p.cand_gt_ok,
     p.cand_off, p.mark_read) = [a.ctypes.data if a.size else None for a in keep]
    p.svlen_thres = clamp_u32(svlen_thres)
    p.suppread_thres = clamp_u32(suppread_thres)
    return p, keep


def problem_from_device(soa_host, dev_ptrs, svlen_thres, suppread_thres):
    """EfProblem whose arrays are device pointers (dict name -> int) and cand_ctg_off the host array."""
    p = EfProblem()
    p.n_contigs, p.n_cands, p.n_marks, p.n_reads = (soa_host.n_contigs, soa_host.n_cands, soa_host.n_marks,
                                                    soa_host.n_reads)
    p.cand_ctg_off = soa_host.cand_ctg_off.ctypes.data
    for name in ('read_tag', 'cand_pos', 'cand_svlen', 'cand_svread', 'cand_refread', 'cand_gt_ok', 'cand_off',
                 'mark_read'):
        setattr(p, name, dev_ptrs[name] or None)
    p.svlen_thres = clamp_u32(svlen_thres)
    p.suppread_thres = clamp_u32(suppread_thres)
    return p
