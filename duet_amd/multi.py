# coding=utf-8
"""Step E/F over the N GPUs of one node: contigs shard, one all-gather reassembles the call set.

Every quantity of step E/F is per contig (tag dicts sv_phasing_fn.py:15-18, join :47, seed sets :195-203,
decisions :206-212), so nothing crosses contigs before the final sort (:229).  `sv_phasing_sharded` -- reached
through `sv_phasing(..., gpus=N)` / `duet --gpus N` -- starts one process per GPU from the (GPU-untouched) calling
process (duet_amd/launch.py).  Every rank
  1. reads the inputs with the native ingest (the same deterministic arrays on every rank; no broadcast needed),
  2. assigns contigs to ranks longest-processing-time-first on mark counts (duet_amd/dist.py) and keeps its own,
  3. runs ef_classify -> ef_seed_sort -> ef_finalize on its shard (libduet_ef.so, no data-path exchange),
  4. contributes its block `ps u32[n_max] | pred u8[n_max] | status` to ONE all_gather_into_tensor (RCCL over xGMI
     under backend "nccl"),
and rank 0 puts the records back into callset order, orders and formats the rows (on its device,
duet_rows_run_device) and appends them to the phased_sv.vcf whose header it wrote first.

DUET_ONE_GPU=1 is a plumbing mode for a box with a single GPU: every rank uses device 0 and the collective goes
through gloo.  Exit codes of a rank: 0 ok, 3 the native ingest declined the input (the caller falls back to the
single-process Python path, which raises what upstream raises), 5 division by zero (sv_phasing_fn.py:123).
"""

import logging
import os
import sys

import numpy as np

from duet_amd import dist as D
from duet_amd import launch

RC_DECLINED = 3
RC_DIV_ZERO = 5
TRAILER = 16                      # bytes after the records: status u32 + padding


def hip_compute(device_id):
    """-> compute(sub_soa, svlen_thres, suppread_thres, n_max) -> (torch.uint8 block on the device, status)"""
    from duet_amd import _lib
    from duet_amd.devmem import DeviceProblem
    ctx = _lib.Context(device_id)

    def compute(sub, svlen_thres, suppread_thres, n_max):
        import torch
        dp = DeviceProblem(sub, svlen_thres, suppread_thres, device='cuda:%d' % device_id, n_cands_max=n_max, trailer=TRAILER)
        status = 0
        if sub.n_cands:
            stream = dp.run(ctx)
            try:
                ctx.check(stream)
            except ZeroDivisionError:
                status = RC_DIV_ZERO
        else:
            torch.cuda.synchronize()
        return dp.out_block, status

    compute.ctx = ctx
    return compute


def block_from_arrays(pred, ps, n_max):
    """Host (pred, ps) -> the record block as a CPU torch tensor (used by the CPU tests' stand-in compute)."""
    import torch
    block = np.zeros(D.record_bytes(n_max) + TRAILER, dtype=np.uint8)
    block[:4 * len(ps)] = np.ascontiguousarray(ps, dtype=np.uint32).view(np.uint8)
    block[4 * n_max:4 * n_max + len(pred)] = pred
    return torch.from_numpy(block)


def rank_body(home, svlen_thres, suppread_thres, thread, include_all_ctgs, rank, world, compute, backend,
              device_rows_ctx=None, device_id=0):
    """What one rank does once the process group exists.  -> exit code."""
    import torch
    from duet_amd import sv_phasing as S
    caller_vcf = home + '/sv_calling/variants.vcf'
    out_vcf = home + '/phased_sv.vcf'
    ing, chrom_list = S.load_native(home, max(1, int(thread) // world), include_all_ctgs, caller_vcf, log=(rank == 0))
    if ing is None:
        return RC_DECLINED
    try:
        soa = ing.soa
        if rank == 0:
            S.write_header(ing, include_all_ctgs, out_vcf)
            S.log_ingest(ing, chrom_list)
            logging.info('integrate read weight information')
            logging.info('calculate read weight statistics')
            logging.info('predict SV haplotypes in the callset')
        owned = D.lpt_assign(D.contig_mark_counts(soa), world)
        sizes = D.shard_sizes(soa, owned)
        n_max = max(max(sizes), 1)
        sub = D.shard_soa(soa, owned[rank])
        block, status = compute(sub, svlen_thres, suppread_thres, n_max)
        rb = D.record_bytes(n_max)
        block[rb:rb + 4] = torch.from_numpy(np.array([status], dtype=np.uint32).view(np.uint8)).to(block.device)
        if backend == 'gloo' and block.device.type != 'cpu':
            block = block.cpu()
        gathered = D.allgather_records(block, world)                    # the ONE collective of the path
        if rank != 0:
            return 0
        g = gathered.cpu().numpy()
        for r in range(world):
            if int(g[r, rb:rb + 4].view(np.uint32)[0]) == RC_DIV_ZERO:
                return RC_DIV_ZERO
        pred, ps = D.merge_results(soa, owned, [D.unpack_block(g[r], n_max, sizes[r]) for r in range(world)])
        rows = None
        if device_rows_ctx is not None and os.environ.get('DUET_DEVICE_ROWS') != '0' and soa.n_cands:
            rows = ing.rows()
        if rows is not None:
            from duet_amd.devmem import DeviceProblem, device_rows
            full = DeviceProblem(soa, svlen_thres, suppread_thres, device='cuda:%d' % device_id)
            full.load_results(pred, ps)
            body = device_rows(device_rows_ctx, full, rows)[0]
        else:
            body = ing.emit_rows(pred, ps)
        logging.info('write phased callset into .vcf file')
        with open(out_vcf, 'ab') as out:
            out.write(body)
        return 0
    finally:
        ing.close()


def rank_main(argv):
    """Entry of one rank process: python -m duet_amd.multi HOME SVLEN SUPP THREAD ALL_CTGS"""
    home, svlen_thres, suppread_thres, thread, all_ctgs = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4] == '1'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    one_gpu = os.environ.get('DUET_ONE_GPU') == '1'
    device_id = 0 if one_gpu else local_rank
    import torch
    import torch.distributed as td
    if rank == 0:
        from duet_amd.utils import add_stream_logging
        add_stream_logging(home)
    torch.cuda.set_device(device_id)
    backend = 'gloo' if one_gpu else 'nccl'
    if one_gpu:
        td.init_process_group('gloo', rank=rank, world_size=world)
    else:
        td.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device_id))
    try:
        compute = hip_compute(device_id)               # raises when libduet_ef.so / the GPU is missing: no fallback
        return rank_body(home, svlen_thres, suppread_thres, thread, all_ctgs, rank, world, compute, backend,
                         device_rows_ctx=compute.ctx, device_id=device_id)
    finally:
        td.destroy_process_group()


def sv_phasing_sharded(home, svlen_thres, suppread_thres, thread, include_all_ctgs, gpus, extra_env=None):
    """Parent side: start `gpus` ranks and wait.  -> True when phased_sv.vcf has been written, False when the native
    ingest declined the input (the caller then runs the single-process Python path).  Raises ZeroDivisionError /
    RuntimeError like the single-GPU path would."""
    argv = ['-m', 'duet_amd.multi', home, str(int(svlen_thres)), str(int(suppread_thres)), str(int(thread)),
            '1' if include_all_ctgs else '0']
    env = {'PYTHONPATH': os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                         ([os.environ['PYTHONPATH']] if os.environ.get('PYTHONPATH') else []))}
    if any(isinstance(h, logging.FileHandler) for h in logging.getLogger().handlers):
        for h in logging.getLogger().handlers:
            h.flush()
        env['DUET_RANK_LOG'] = '1'
    if extra_env:
        env.update(extra_env)
    rc = launch.spawn_ranks(gpus, argv, extra_env=env)
    for h in logging.getLogger().handlers:              # rank 0 appended to the log file: go on behind its lines
        if isinstance(h, logging.FileHandler) and h.stream is not None:
            h.acquire()
            try:
                h.stream.seek(0, os.SEEK_END)
            finally:
                h.release()
    if rc == 0:
        return True
    if rc == RC_DECLINED:
        logging.info('native ingest declined the input; using the single-process Python path')
        return False
    if rc == RC_DIV_ZERO:
        raise ZeroDivisionError('division by zero')         # what upstream raises (sv_phasing_fn.py:123)
    raise RuntimeError('multi-GPU SV phasing failed: a rank exited with code %d' % rc)


if __name__ == '__main__':
    sys.exit(rank_main(sys.argv[1:]))
