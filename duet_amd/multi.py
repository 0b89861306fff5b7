"""Step E/F over the N GPUs of one node: contigs shard, one all-gather reassembles the call set.

Every quantity of step E/F is per contig (tag dicts sv_phasing_fn.py:15-18, join :47, seed sets :195-203,
decisions :206-212), so nothing crosses contigs before the final sort (:229).  `sv_phasing_sharded` -- reached
through `sv_phasing(..., gpus=N)` / `duet --gpus N` -- starts one process per GPU from the (GPU-untouched) calling
process (duet_amd/launch.py).  Every rank
  1. counts the caller VCF's records and line bytes per listed contig (one memchr pass, the same numbers on every rank)
     and assigns contigs to ranks longest-processing-time-first on the line bytes (duet_amd/dist.py): the assignment and
     every rank's candidate count are known before anything is parsed,
  2. reads ITS contigs only -- their BAMs, their records (native ingest with an ownership mask) --,
  3. runs ef_classify -> ef_seed_sort -> ef_finalize on its shard (libduet_ef.so, no data-path exchange),
  4. contributes its block `ps u32[n_max] | pred u8[n_max] | status | rows kept per CHROM text` to ONE all-gather (RCCL over
     xGMI inside libduet_ef.so: steps 3 and 4 are one library call, duet_comm_ef_allgather -- the kernels write into the block
     on the device, a small kernel appends the trailer, ncclAllGather runs on the kernels' stream),
  5. numbers and formats the rows of its own contigs (it alone holds their REF / ALT texts): a CHROM text belongs to one
     contig, the file is the text blocks in byte order (sv_phasing_fn.py:229 sorts CHROM as text first), and the gathered
     counts tell every rank where its blocks' numbering starts; the blocks go to a part file beside the output,
and the parent appends the blocks, in the order of their texts, to the phased_sv.vcf whose header rank 0 wrote first.

DUET_ONE_GPU=1 is a plumbing mode for a box with a single GPU: every rank uses device 0 and the collective goes
through gloo.  Exit codes of a rank: 0 ok, 3 the native ingest declined the input (the caller falls back to the
single-process Python path, which raises what upstream raises), 5 division by zero (sv_phasing_fn.py:123).
DUET_RDZV_TIMEOUT (seconds, default 300) bounds the rendezvous (the TCP star AND ncclCommInitRank inside the library) and the
collective (the stream behind ncclAllGather is polled against a deadline): a rank that runs into it leaves with code 7 through
os._exit, and the launcher ends the others.  DUET_RANK_TIMEOUT (default 3600) bounds the whole run of the ranks: on expiry the
children are killed and the call fails -- it never hangs.
"""

import json
import logging
import os
import sys

import numpy as np

from duet_amd import dist as D
from duet_amd import launch

RC_DECLINED = 3
RC_DIV_ZERO = 5
RC_TIMEOUT = 7                    # a collective step ran into DUET_RDZV_TIMEOUT
STATUS_BYTES = 16                 # status u32 + padding, in front of the per-text counts


def trailer_bytes(n_contigs):
    """Bytes behind the records of a block: status word (+ padding), then rows kept per CHROM-text slot (u64[2K])."""
    return STATUS_BYTES + 16 * int(n_contigs)


TRAILER = STATUS_BYTES            # (kept for callers that only carry the status word: bench.py)


def hip_compute(device_id):
    """-> compute(sub_soa, svlen_thres, suppread_thres, n_max) -> (torch.uint8 block on the device, status)"""
    from duet_amd import _lib
    from duet_amd.devmem import DeviceProblem
    ctx = _lib.Context(device_id)

    def compute(sub, svlen_thres, suppread_thres, n_max):
        import torch
        dp = DeviceProblem(sub, svlen_thres, suppread_thres, device='cuda:%d' % device_id, n_cands_max=n_max,
                           trailer=trailer_bytes(sub.n_contigs))
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


def block_np(pred, ps, n_max, n_contigs=0):
    """Host (pred, ps) -> the record block `ps u32[n_max] | pred u8[n_max] | trailer` as a numpy uint8 array."""
    block = np.zeros(D.record_bytes(n_max) + trailer_bytes(n_contigs), dtype=np.uint8)
    block[:4 * len(ps)] = np.ascontiguousarray(ps, dtype=np.uint32).view(np.uint8)
    block[4 * n_max:4 * n_max + len(pred)] = pred
    return block


def block_from_arrays(pred, ps, n_max, n_contigs=0):
    """... as a CPU torch tensor (used by the CPU tests' stand-in compute)."""
    import torch
    return torch.from_numpy(block_np(pred, ps, n_max, n_contigs))


def hip_compute_np(device_id):
    """The rank compute without torch: host arrays through duet_ef_run_host (upload, the three kernels, download).
    -> compute(sub_soa, svlen_thres, suppread_thres, n_max) -> (numpy uint8 block, status)"""
    from duet_amd import _lib
    ctx = _lib.Context(device_id)

    def compute(sub, svlen_thres, suppread_thres, n_max):
        status = 0
        pred, ps = np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint32)
        if sub.n_cands:
            try:
                pred, ps = ctx.run_host(sub, svlen_thres, suppread_thres)
            except ZeroDivisionError:
                status = RC_DIV_ZERO
                pred, ps = np.zeros(sub.n_cands, dtype=np.uint8), np.zeros(sub.n_cands, dtype=np.uint32)
        return block_np(pred, ps, n_max, sub.n_contigs), status

    compute.ctx = ctx
    return compute


def part_path(home, rank):
    return os.path.join(home, 'phased_sv.vcf.part%d' % rank)


def text_order(chrom_list):
    """CHROM-text slots (2 * contig + spelling) in the byte order of their texts: the order of their blocks in the file."""
    texts = []
    for k, c in enumerate(chrom_list):
        texts.append((('chr' + c).encode('utf-8'), 2 * k))
        texts.append((c.encode('utf-8'), 2 * k + 1))
    return [slot for _, slot in sorted(texts)]


def plan_shards(caller_vcf, chrom_list, world):
    """-> (owned[rank] = ascending contig indices, records per contig) or None (no native library / unreadable VCF)."""
    from duet_amd.native import NativeIngest
    pre = NativeIngest.precount(caller_vcf, chrom_list)
    if pre is None:
        return None
    n_rec, n_bytes = pre
    return D.lpt_assign(n_bytes, world), n_rec


def rank_body(home, svlen_thres, suppread_thres, thread, include_all_ctgs, rank, world, compute, backend,
              device_rows_ctx=None, device_id=0, gather=None):
    """What one rank does once the ranks can talk to each other.  -> exit code.
    gather: an object with allgather(numpy uint8 block) -> numpy [world, bytes] (duet_amd/comm.py: the in-library RCCL
    collective, or the TCP star of the one-GPU plumbing mode); None: torch.distributed's default process group (`backend`
    "nccl" / "gloo": what the CPU tests and DUET_COMM=torch use)."""
    from duet_amd import sv_phasing as S
    from duet_amd.native import NativeIngest
    from duet_amd.read_file import init_chrom_list
    caller_vcf = home + '/sv_calling/variants.vcf'
    out_vcf = home + '/phased_sv.vcf'
    if os.path.exists(part_path(home, rank)):
        os.remove(part_path(home, rank))
    chrom_list = init_chrom_list(include_all_ctgs, home)
    if os.environ.get('DUET_NATIVE_INGEST') == '0' or os.environ.get('DUET_USE_SAMTOOLS') == '1':
        return RC_DECLINED
    # the pre-count (records and line bytes per listed contig: one memchr pass, the same numbers on every rank) on the handle that
    # then parses: contigs -> ranks longest-processing-time-first on the line bytes, every rank's candidate count known up front
    shard = {}

    def plan(n_rec, n_bytes):
        shard['owned'] = D.lpt_assign(n_bytes, world)
        shard['n_rec'] = n_rec
        return shard['owned'][rank]

    ing = NativeIngest.load(caller_vcf, home + '/snp_phasing/', chrom_list, max(1, int(thread) // world), plan=plan)
    if ing is None or 'owned' not in shard:
        return RC_DECLINED
    owned, n_rec = shard['owned'], shard['n_rec']
    sizes = [int(sum(int(n_rec[k]) for k in o)) for o in owned]
    n_max = max(max(sizes), 1)
    if ing is None or ing.handle is None:
        if rank == 0 and ing is not None:
            logging.info('native ingest declined (%s); using the Python path' % ing.why)
        return RC_DECLINED
    try:
        soa = ing.soa
        K = soa.n_contigs
        if soa.n_cands != sizes[rank]:
            raise RuntimeError('rank %d parsed %d candidates, the pre-count said %d' % (rank, soa.n_cands, sizes[rank]))
        if rank == 0:
            S.write_header(ing, include_all_ctgs, out_vcf)
            logging.info('extract SNP signatures')
            logging.info('extract SV signatures')
            logging.info('  (%d ranks, contigs per rank: %s)' % (world, ' '.join(str(len(o)) for o in owned)))
            logging.info('integrate read weight information')
            logging.info('calculate read weight statistics')
            logging.info('predict SV haplotypes in the callset')
        rb = D.record_bytes(n_max)
        tb = trailer_bytes(K)
        if gather is not None:
            # the ONE collective of the path, with the rank's whole data path in front of it: the in-library RCCL gather runs the
            # three kernels straight into the rank's record block on the device, appends the trailer (status word, rows kept per
            # CHROM-text slot) with a small kernel and gathers on the kernels' stream -- nothing visits the host before the
            # collective (duet_comm_ef_allgather); the TCP star of the plumbing mode / the CPU tests assembles the same block from
            # `compute`'s output
            g = gather.ef_allgather(soa, svlen_thres, suppread_thres, ing.cand_slots(), 2 * K, n_max, compute)
            if g.shape != (world, rb + tb):
                raise RuntimeError('gathered blocks of shape %s, expected (%d, %d)' % (g.shape, world, rb + tb))
            pred, ps = D.unpack_block(g[rank], n_max, soa.n_cands)
        else:
            block, status = compute(soa, svlen_thres, suppread_thres, n_max)
            is_np = isinstance(block, np.ndarray)
            if (block.size if is_np else block.numel()) != rb + tb:
                raise RuntimeError('record block of %d bytes, expected %d' % (block.size if is_np else block.numel(), rb + tb))
            mine = block if is_np else block.cpu().numpy()
            pred, ps = D.unpack_block(mine, n_max, soa.n_cands)
            kept = ing.count_kept(pred) if status == 0 else np.zeros(2 * K, dtype=np.int64)
            trailer = np.zeros(tb, dtype=np.uint8)
            trailer[:4] = np.array([status], dtype=np.uint32).view(np.uint8)
            trailer[STATUS_BYTES:] = kept.astype(np.uint64).view(np.uint8)
            import torch
            if is_np:
                block = torch.from_numpy(block)
            block[rb:] = torch.from_numpy(trailer).to(block.device)
            if backend == 'gloo' and block.device.type != 'cpu':
                block = block.cpu()
            gathered = D.allgather_records(block, world)                    # the ONE collective of the path
            g = gathered.cpu().numpy()
        total = np.zeros(2 * K, dtype=np.int64)
        for r in range(world):
            st = int(g[r, rb:rb + 4].view(np.uint32)[0])
            if st & 0x80000000:
                # (include/duet_ef.h, DUET_COMM_STATUS_RANK_FAILED: that rank's own part failed; it contributed a block so that
                # nobody waits for the time limit, and says why itself)
                raise RuntimeError('rank %d failed before the collective (duet_status %d)' % (r, -(st & 0xFFFF)))
            if st == RC_DIV_ZERO:
                return RC_DIV_ZERO
            total += g[r, rb + STATUS_BYTES:rb + tb].view(np.uint64).astype(np.int64)
        # where each CHROM text's rows start counting: after the rows of every text that sorts before it
        id_base = np.ones(2 * K, dtype=np.int64)
        run = 1
        for slot in text_order(chrom_list):
            id_base[slot] = run
            run += int(total[slot])
        body, off, ln = ing.emit_blocks(pred, ps, id_base)
        with open(part_path(home, rank) + '.tmp', 'wb') as out:
            out.write(body)
        with open(part_path(home, rank) + '.json', 'w') as out:
            json.dump(dict(off=[int(x) for x in off], len=[int(x) for x in ln]), out)
        os.replace(part_path(home, rank) + '.tmp', part_path(home, rank))
        return 0
    finally:
        ing.close()


def assemble(home, include_all_ctgs, world):
    """Append the ranks' blocks to phased_sv.vcf (which holds the header) in the byte order of their CHROM texts, and remove
    the part files.  Run by the parent once every rank has exited with 0."""
    from duet_amd.read_file import init_chrom_list
    chrom_list = init_chrom_list(include_all_ctgs, home)
    parts = []
    for r in range(world):
        with open(part_path(home, r) + '.json') as f:
            parts.append((json.load(f), part_path(home, r)))
    logging.info('write phased callset into .vcf file')
    with open(home + '/phased_sv.vcf', 'ab') as out:
        files = [open(path, 'rb') for _, path in parts]
        try:
            for slot in text_order(chrom_list):
                for (meta, _), f in zip(parts, files):
                    if meta['len'][slot]:
                        f.seek(meta['off'][slot])
                        out.write(f.read(meta['len'][slot]))
        finally:
            for f in files:
                f.close()
    for _, path in parts:
        os.remove(path)
        os.remove(path + '.json')


def rank_main(argv):
    """Entry of one rank process: python -m duet_amd.multi HOME SVLEN SUPP THREAD ALL_CTGS"""
    home, svlen_thres, suppread_thres, thread, all_ctgs = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4] == '1'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    one_gpu = os.environ.get('DUET_ONE_GPU') == '1'
    device_id = 0 if one_gpu else local_rank
    if os.environ.get('DUET_COMM', '') != 'torch':
        # No torch in this process: numpy, the native libraries, and for the collective either RCCL inside libduet_ef.so
        # (duet_comm_*: ncclAllGather over xGMI) or, in the one-GPU plumbing mode, the TCP star that also carries RCCL's id.
        os.environ['DUET_NO_TORCH'] = '1'
        from duet_amd import comm
        if rank == 0:
            from duet_amd.utils import add_stream_logging
            add_stream_logging(home)
        limit = float(os.environ.get('DUET_RDZV_TIMEOUT', '300'))
        compute = hip_compute_np(device_id)            # raises when libduet_ef.so / the GPU is missing: no fallback
        star = comm.TcpStar(rank, world, timeout=limit)
        gather = None
        try:
            gather = comm.HostGather(star) if one_gpu else comm.RcclGather(compute.ctx, star)
            return rank_body(home, svlen_thres, suppread_thres, thread, all_ctgs, rank, world, compute, gather.name,
                             device_rows_ctx=compute.ctx, device_id=device_id, gather=gather)
        except comm.CommTimeout as e:
            # a step of the RCCL path ran into DUET_RDZV_TIMEOUT (a rank is missing or stuck).  A helper thread may still sit
            # inside RCCL and the stream may never drain: no clean-up, no interpreter shutdown -- say why and leave NOW with a
            # non-zero code (the launcher then ends the other ranks).  Never a re-exec.
            sys.stderr.write('rank %d: %s\n' % (rank, e))
            sys.stderr.flush()
            logging.shutdown()
            os._exit(RC_TIMEOUT)
        finally:
            if gather is not None:
                gather.close()
            star.close()
    import torch
    import torch.distributed as td
    if rank == 0:
        from duet_amd.utils import add_stream_logging
        add_stream_logging(home)
    torch.cuda.set_device(device_id)
    backend = 'gloo' if one_gpu else 'nccl'
    import datetime
    limit = datetime.timedelta(seconds=float(os.environ.get('DUET_RDZV_TIMEOUT', '300')))
    if one_gpu:
        td.init_process_group('gloo', rank=rank, world_size=world, timeout=limit)
    else:
        td.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device_id), timeout=limit)
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
    rc = launch.spawn_ranks(gpus, argv, extra_env=env, timeout=float(os.environ.get('DUET_RANK_TIMEOUT', '3600')))
    for h in logging.getLogger().handlers:              # rank 0 appended to the log file: go on behind its lines
        if isinstance(h, logging.FileHandler) and h.stream is not None:
            h.acquire()
            try:
                h.stream.seek(0, os.SEEK_END)
            finally:
                h.release()
    if rc == 0:
        assemble(home, include_all_ctgs, int(gpus))
        return True
    for r in range(int(gpus)):                          # whatever a failed run left behind
        for path in (part_path(home, r), part_path(home, r) + '.json', part_path(home, r) + '.tmp'):
            if os.path.exists(path):
                os.remove(path)
    if rc == 124:
        raise RuntimeError('multi-GPU SV phasing: the ranks did not finish within DUET_RANK_TIMEOUT; they were killed')
    if rc == RC_DECLINED:
        logging.info('native ingest declined the input; using the single-process Python path')
        return False
    if rc == RC_DIV_ZERO:
        raise ZeroDivisionError('division by zero')         # what upstream raises (sv_phasing_fn.py:123)
    if rc == RC_TIMEOUT:
        raise RuntimeError('multi-GPU SV phasing: a rank gave up on the collective after DUET_RDZV_TIMEOUT (a rank is missing or stuck)')
    raise RuntimeError('multi-GPU SV phasing failed: a rank exited with code %d' % rc)


if __name__ == '__main__':
    sys.exit(rank_main(sys.argv[1:]))
