# coding=utf-8
"""Device residency for E/F problems: PyTorch is used only as the allocator / stream provider;
the kernels see raw pointers through include/duet_ef.h."""

import numpy as np

from duet_amd import _lib

DEVICE_FIELDS = ('read_tag', 'cand_pos', 'cand_svlen', 'cand_svread', 'cand_refread', 'cand_gt_ok',
                 'cand_off', 'mark_read')


class DeviceProblem(object):
    """An EfSoA uploaded once to HBM, plus output buffers, ready for repeated duet_ef_run_device."""

    def __init__(self, soa, svlen_thres, suppread_thres, device='cuda:0', misalign_marks=0, n_cands_max=None):
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
        self.out_block = torch.zeros(record_bytes(self.n_max), dtype=torch.uint8, device=self.device)
        self.out_ps_ptr = self.out_block.data_ptr()
        self.out_pred_ptr = self.out_block.data_ptr() + 4 * self.n_max
        self.problem = _lib.problem_from_device(soa, ptrs, svlen_thres, suppread_thres)

    def run(self, ctx, stream=None):
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device).cuda_stream
        ctx.run_device(self.problem, self.out_pred_ptr, self.out_ps_ptr, stream)
        return stream

    def results(self):
        """-> (pred u8[C], ps u32[C]) on the host (synchronises)."""
        from duet_amd.dist import unpack_block
        return unpack_block(self.out_block.cpu().numpy(), self.n_max, self.soa.n_cands)
