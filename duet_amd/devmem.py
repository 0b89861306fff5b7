# coding=utf-8
"""Device residency for E/F problems: PyTorch is used only as the allocator / stream provider;
the kernels see raw pointers through include/duet_ef.h."""

import numpy as np

from duet_amd import _lib

DEVICE_FIELDS = ('read_tag', 'cand_pos', 'cand_svlen', 'cand_svread', 'cand_refread', 'cand_gt_ok',
                 'cand_off', 'mark_read')


class DeviceProblem(object):
    """An EfSoA uploaded once to HBM, plus output buffers, ready for repeated duet_ef_run_device."""

    def __init__(self, soa, svlen_thres, suppread_thres, device='cuda:0', misalign_marks=0, n_cands_max=None,
                 n_out=1, trailer=0):
        import torch
        self.torch = torch
        self.soa = soa
        self.device = torch.device(device)
        self.buffers = {}
        ptrs = {}
        for name in DEVICE_FIELDS:
            a = getattr(soa, name)
            pad = misalign_marks if name == 'mark_read' else 0
            raw = torch.zeros(a.nbytes + pad + 64, dtype=torch.uint8, device=self.device)
            if a.nbytes:
                raw[pad:pad + a.nbytes] = torch.from_numpy(np.frombuffer(a.tobytes(), dtype=np.uint8).copy()).to(self.device)
            self.buffers[name] = raw
            ptrs[name] = raw.data_ptr() + pad
        # results live in ONE block -- ps u32[n_max] then pred u8[n_max] -- so that a multi-GPU job can
        # hand the whole block to a single all-gather (duet_amd/dist.py)
        from duet_amd.dist import record_bytes
        self.n_max = max(int(n_cands_max or soa.n_cands), soa.n_cands, 1)
        # n_out > 1: rotating result blocks, so that the all-gather of one job can run beside the kernels of the next
        # (the blocks are slices of ONE allocation, so that several jobs' results can go into one collective)
        rb = record_bytes(self.n_max) + int(trailer)       # trailer: caller-defined bytes behind the records (status word)
        self.out_storage = torch.zeros(rb * max(1, n_out), dtype=torch.uint8, device=self.device)
        self.out_blocks = [self.out_storage[i * rb:(i + 1) * rb] for i in range(max(1, n_out))]
        self.out_block = self.out_blocks[0]
        self.problem = _lib.problem_from_device(soa, ptrs, svlen_thres, suppread_thres)

    def run(self, ctx, stream=None, slot=0):
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device).cuda_stream
        blk = self.out_blocks[slot]
        ctx.run_device(self.problem, blk.data_ptr() + 4 * self.n_max, blk.data_ptr(), stream)
        return stream

    def load_results(self, pred, ps, slot=0):
        """Put host (pred, ps) -- e.g. the merged records of a multi-GPU run -- into result block `slot`, so that the
        device-side row emission can read them."""
        torch = self.torch
        blk, C = self.out_blocks[slot], self.soa.n_cands
        if C:
            blk[:4 * C] = torch.from_numpy(np.ascontiguousarray(ps, dtype=np.uint32).view(np.uint8)).to(self.device)
            blk[4 * self.n_max:4 * self.n_max + C] = torch.from_numpy(np.ascontiguousarray(pred, dtype=np.uint8)).to(self.device)

    def results(self, slot=0):
        """-> (pred u8[C], ps u32[C]) on the host (synchronises)."""
        from duet_amd.dist import unpack_block
        return unpack_block(self.out_blocks[slot].cpu().numpy(), self.n_max, self.soa.n_cands)


def device_rows(ctx, dp, rows, slot=0, stream=None):
    """Rows of phased_sv.vcf for the results in dp.out_blocks[slot], formatted on the device
    (duet_rows_run_device). `rows` = NativeIngest.rows().  -> (bytes of the rows, number of rows)."""
    torch = dp.torch
    if stream is None:
        stream = torch.cuda.current_stream(dp.device).cuda_stream
    soa = dp.soa

    def up(a):
        a = np.ascontiguousarray(a)
        t = torch.zeros(a.nbytes + 64, dtype=torch.uint8, device=dp.device)
        if a.nbytes:
            t[:a.nbytes] = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dp.device)
        return t

    keep = [up(rows['pool']), up(rows['str_off']), up(rows['chrom_rank']), up(rows['plus'])]
    ctg_off = np.ascontiguousarray(soa.cand_ctg_off, dtype=np.uint32)
    blk = dp.out_blocks[slot]
    p = _lib.RowsProblem()
    p.n_contigs, p.n_cands = soa.n_contigs, soa.n_cands
    p.cand_ctg_off = ctg_off.ctypes.data
    p.pred, p.ps = blk.data_ptr() + 4 * dp.n_max, blk.data_ptr()
    ef = dp.problem
    p.cand_pos, p.cand_svlen = ef.cand_pos, ef.cand_svlen
    p.cand_plus, p.cand_chrom_rank = keep[3].data_ptr(), keep[2].data_ptr()
    p.n_chrom_texts, p.max_pos = int(rows['n_chrom_texts']), int(rows['max_pos'])
    p.pool, p.pool_bytes, p.str_off = keep[0].data_ptr(), int(rows['pool_bytes']), keep[1].data_ptr()
    p.cand_off, p.mark_read, p.read_tag = ef.cand_off, ef.mark_read, ef.read_tag
    cap = int(rows['pool_bytes']) + 96 * soa.n_cands + 64
    out = torch.empty(cap, dtype=torch.uint8, device=dp.device)
    n, n_rows = ctx.rows_device(p, out.data_ptr(), cap, stream)
    return out[:n].cpu().numpy().tobytes(), n_rows


class DeviceCluster(object):
    """Raw SV marks uploaded once + result buffers, for repeated duet_cluster_run_device (stage A0)."""

    def __init__(self, marks, max_dist=0.9, part_gap=1000, part_max=100, normalizer=900.0, device='cuda:0'):
        import ctypes
        import torch
        self.torch = torch
        self.device = torch.device(device)
        M = len(marks['pos'])
        self.M = M
        self.keep = {}

        def up(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            t = torch.zeros(a.nbytes + 64, dtype=torch.uint8, device=self.device)
            if a.nbytes:
                t[:a.nbytes] = torch.from_numpy(np.frombuffer(a.tobytes(), dtype=np.uint8).copy()).to(self.device)
            return t

        p = _lib.ClusterProblem()
        p.n_marks, p.part_gap, p.part_max = M, int(part_gap), int(part_max)
        p.max_dist, p.normalizer = float(max_dist), float(normalizer)
        if M:
            p.n_contigs_hint = int(np.max(marks['contig'])) + 1
            p.n_types_hint = int(np.max(marks['type'])) + 1
            p.max_pos_hint = int(np.max(marks['pos']))
            p.max_span_hint = max(int(np.max(marks['span'])), 1)
        for field, key, dt in (('mark_contig', 'contig', np.uint16), ('mark_type', 'type', np.uint8),
                               ('mark_pos', 'pos', np.uint32), ('mark_span', 'span', np.uint32)):
            self.keep[field] = up(marks[key], dt)
            setattr(p, field, self.keep[field].data_ptr())
        self.problem = p
        r = _lib.ClusterResult()
        sizes = dict(order=4 * M, cand_off=4 * (M + 1), cand_contig=2 * M, cand_type=M, cand_pos=4 * M, cand_span=4 * M,
                     n_cands=4)
        for k, nbytes in sizes.items():
            self.keep['out_' + k] = torch.zeros(nbytes + 64, dtype=torch.uint8, device=self.device)
            setattr(r, k, self.keep['out_' + k].data_ptr())
        self.result = r
        self._ct = ctypes

    def run(self, ctx, stream=None):
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device).cuda_stream
        ct = self._ct
        rc = ctx.lib.duet_cluster_run_device(ctx.handle, ct.byref(self.problem), ct.byref(self.result),
                                             ct.c_void_p(stream))
        if rc:
            ctx._raise(rc)

    def n_cands(self):
        return int(self.keep['out_n_cands'][:4].cpu().numpy().view(np.uint32)[0])


class DeviceSvim(DeviceCluster):
    """Fused SVIM-mode pipeline on resident inputs: raw marks (+ their read indices), read tags, binned depth."""

    def __init__(self, marks, read_tag, depth, depth_off, depth_bin=1000, svlen_thres=50, suppread_thres=2,
                 max_dist=0.9, device='cuda:0'):
        DeviceCluster.__init__(self, marks, max_dist=max_dist, device=device)
        torch = self.torch

        def up(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            t = torch.zeros(a.nbytes + 64, dtype=torch.uint8, device=self.device)
            if a.nbytes:
                t[:a.nbytes] = torch.from_numpy(np.frombuffer(a.tobytes(), dtype=np.uint8).copy()).to(self.device)
            return t

        self.keep['sv_read'] = up(marks['read'], np.uint32)
        self.keep['sv_tag'] = up(read_tag, np.uint64)
        self.keep['sv_depth'] = up(depth, np.uint32)
        self.depth_off = np.ascontiguousarray(depth_off, dtype=np.uint32)
        p = _lib.SvimProblem()
        p.marks = self.problem
        p.mark_read = self.keep['sv_read'].data_ptr()
        p.read_tag = self.keep['sv_tag'].data_ptr()
        p.n_reads = len(read_tag)
        p.n_contigs = len(self.depth_off) - 1
        p.depth = self.keep['sv_depth'].data_ptr()
        p.depth_off = self.depth_off.ctypes.data
        p.depth_bin, p.svlen_thres, p.suppread_thres = int(depth_bin), int(svlen_thres), int(suppread_thres)
        self.sv_problem = p
        self.out_pred = torch.zeros(self.M + 64, dtype=torch.uint8, device=self.device)
        self.out_ps = torch.zeros(self.M + 16, dtype=torch.int32, device=self.device)
        self.n_found = 0

    def run_fused(self, ctx, stream=None, wait=True):
        """wait=True: the call learns the candidate count (one host round trip inside).  wait=False: fully
        asynchronous; fetch() reads the count from the device."""
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device).cuda_stream
        ct = self._ct
        n = ct.c_uint32(0)
        rc = ctx.lib.duet_svim_phase_device(ctx.handle, ct.byref(self.sv_problem), ct.byref(self.result),
                                            ct.c_void_p(self.out_pred.data_ptr()), ct.c_void_p(self.out_ps.data_ptr()),
                                            ct.byref(n) if wait else None, ct.c_void_p(stream))
        if rc:
            ctx._raise(rc)
        self.n_found = n.value if wait else None
        return stream

    def fetch(self):
        """-> dict of the cluster arrays + pred/ps, trimmed to the candidate count (synchronises)."""
        if self.n_found is None:
            self.n_found = self.n_cands()
        N, M = self.n_found, self.M
        g = lambda key, dt, n: self.keep['out_' + key][:n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
        return dict(order=g('order', np.uint32, M), cand_off=g('cand_off', np.uint32, N + 1),
                    cand_contig=g('cand_contig', np.uint16, N), cand_type=g('cand_type', np.uint8, N),
                    cand_pos=g('cand_pos', np.uint32, N), cand_span=g('cand_span', np.uint32, N),
                    pred=self.out_pred[:N].cpu().numpy(), ps=self.out_ps[:N].cpu().numpy().view(np.uint32))
