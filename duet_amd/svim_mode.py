# coding=utf-8
"""SVIM mode without the external caller: the haplotagged BAMs of <home>/snp_phasing alone ->
raw SV marks (CIGAR insertions / deletions, native BAM pass) -> stage A0 -> adapter -> step E/F, the last three in one
device pipeline (duet_svim_phase_device).  Upstream runs `svim alignment` here (src/duet/sv_calling.py:13-15) and reads
its VCF back; the extraction and clustering rules used instead are this repository's own (oracle/svim_oracle.py,
oracle/cluster_oracle.c; parity unpinned), step E/F is the pinned one.
"""

import logging
import os
import time

import numpy as np

from duet_amd import bamio, engine
from duet_amd.native import NativeIngest
from duet_amd.read_file import init_chrom_list


def device_compute(ctx):
    """-> compute(extracted arrays, svlen_thres, suppread_thres, max_dist, depth_bin) -> dict of result arrays, on ctx's GPU:
    stage A0 -> adapter -> step E/F in one device pipeline (duet_svim_phase_device)."""
    def compute(got, svlen_thres, suppread_thres, max_dist, depth_bin):
        from duet_amd.devmem import DeviceSvim
        ds = DeviceSvim(got, got['read_tag'], got['depth'], got['depth_off'], depth_bin, svlen_thres, suppread_thres,
                        max_dist=max_dist, device='cuda:%d' % ctx.device_id)
        ds.run_fused(ctx)
        ctx.check(ds.torch.cuda.current_stream(ds.device).cuda_stream)
        out = ds.fetch()
        return dict(cand_contig=out['cand_contig'], cand_type=out['cand_type'], cand_pos=out['cand_pos'],
                    cand_span=out['cand_span'], support=np.diff(out['cand_off'].astype(np.int64)), pred=out['pred'], ps=out['ps'])
    return compute


def host_compute(ctx):
    """The same through host arrays (duet_svim_phase_host): what a rank of `-b svim-gpu --gpus N` uses -- no torch in the process."""
    def compute(got, svlen_thres, suppread_thres, max_dist, depth_bin):
        out = ctx.svim_host(got, got['read_tag'], got['depth'], got['depth_off'], depth_bin, svlen_thres, suppread_thres, max_dist=max_dist)
        return dict(cand_contig=out['cand_contig'], cand_type=out['cand_type'], cand_pos=out['cand_pos'], cand_span=out['cand_span'],
                    support=np.diff(out['cand_off'].astype(np.int64)), pred=out['pred'], ps=out['ps'])
    return compute


def phase_from_bams(home, svlen_thres=50, suppread_thres=2, thread=4, include_all_ctgs=False, max_dist=0.9,
                    min_sv_size=40, min_mapq=20, depth_bin=1000, ctx=None, only=None, compute=None):
    """-> dict(chroms, cand_contig u16[N], cand_type u8[N] (1 INS / 0 DEL), cand_pos, cand_span, support, pred, ps).
    only: the contig indices to read (a rank of a sharded run; the contig numbering stays the whole list's);
    compute: what turns the extracted arrays into results (default: the GPU pipeline on ctx)."""
    chroms = init_chrom_list(include_all_ctgs, home)
    ing, got = NativeIngest.extract(home + '/snp_phasing/', chroms, thread, min_sv_size, min_mapq, depth_bin, only=only)
    if ing is None:
        raise RuntimeError('signature extraction declined the input: %s' % got)
    ing.close()
    N0 = dict(chroms=chroms, cand_contig=np.zeros(0, np.uint16), cand_type=np.zeros(0, np.uint8),
              cand_pos=np.zeros(0, np.uint32), cand_span=np.zeros(0, np.uint32), support=np.zeros(0, np.int64),
              pred=np.zeros(0, np.uint8), ps=np.zeros(0, np.uint32))
    N0['n_marks'] = 0
    if len(got['pos']) == 0:
        return N0
    if compute is None:
        compute = device_compute(ctx if ctx is not None else engine.default_context())
    out = compute(got, svlen_thres, suppread_thres, max_dist, depth_bin)
    return dict(out, chroms=chroms, n_marks=len(got['pos']))


SV_TYPE_NAMES = ('DEL', 'INS', 'INV', 'DUP')          # the extraction's type codes (oracle/svim_oracle.py)


def spelled_contigs(home, chroms):
    """CHROM text per contig: the spelling of its BAM (chr<c>.bam else <c>.bam, as sv_phasing_fn.py:19-24 looks them up)."""
    names = []
    for c in chroms:
        names.append('chr' + c if os.path.exists(os.path.join(home, 'snp_phasing', 'chr' + c + '.bam')) else c)
    return names


def rows_text(home, res):
    """Rows in phased_sv.vcf's layout (write_file.py:6-17) for the phased candidates of `res`: symbolic REF/ALT,
    sorted like sv_phasing_fn.py:229 (CHROM as text, POS), ids renumbered."""
    names = spelled_contigs(home, res['chroms'])
    hp = {1: '1|0', 2: '0|1', 3: '1|1'}
    keep = np.nonzero(res['pred'])[0]
    order = sorted(keep, key=lambda i: (names[int(res['cand_contig'][i])], int(res['cand_pos'][i])))
    out = []
    for n, i in enumerate(order):
        t = SV_TYPE_NAMES[int(res['cand_type'][i]) & 3]
        ln = int(res['cand_span'][i])
        # (the sign rule of sv_phasing_fn.py:225: positive for INS and DUP, negative for everything else)
        out.append('%s\t%d\tDuet.%d\tN\t<%s>\t.\tPASS\tSVLEN=%d;SVTYPE=<%s>\tHP:PS\t%s:%d\n' % (
            names[int(res['cand_contig'][i])], int(res['cand_pos'][i]), n + 1, t, ln if t in ('INS', 'DUP') else -ln, t,
            hp[int(res['pred'][i])], int(res['ps'][i])))
    return ''.join(out)


def header_text(home, chroms):
    """phased_sv.vcf header lines (write_file.py:19-45).  Upstream copies the ##contig lines of the caller VCF; this mode
    has no caller VCF, so every listed contig that has a BAM contributes one line from the BAM's own reference list."""
    from duet_amd.write_file import _COLS, _HEAD
    lines = []
    for c, name in zip(chroms, spelled_contigs(home, chroms)):
        bam = os.path.join(home, 'snp_phasing', name + '.bam')
        if not os.path.exists(bam):
            continue
        for ref, length in bamio.read_refs(bam):
            if ref == name:
                lines.append('##contig=<ID=%s,length=%d>\n' % (ref, length))
                break
    return _HEAD + ''.join(lines) + _COLS


# ------------------------------------------------------------------------------------------------------------------
# the same over the N GPUs of one node (BASELINE configs[3]: `--sv_caller svim --cluster_max_distance 0.9`, 8 GPUs)
# ------------------------------------------------------------------------------------------------------------------
# Partitions of stage A0 never cross a contig and step E/F is per contig, so RAW MARKS shard by contig exactly like
# candidates do: a rank reads whole contigs' BAMs (marks, tag tables, depth bins), runs the device pipeline on them,
# and the candidates are reassembled by one all-gather of fixed-size records -- sized by a 16-byte exchange of the
# counts, since how many candidates a rank finds is a result, not an input.  Rank 0 writes the rows.

REC_WORDS = 5                     # u32 per candidate: contig | type << 16 | pred << 24, pos, span, support, ps


def bam_weights(home, chroms):
    """Bytes of each listed contig's BAM (0 without one): what the contig -> rank assignment balances.  Every rank
    computes the same numbers from the directory alone."""
    out = []
    for c, name in zip(chroms, spelled_contigs(home, chroms)):
        path = os.path.join(home, 'snp_phasing', name + '.bam')
        out.append(os.path.getsize(path) if os.path.exists(path) else 0)
    return out


def pack_records(res):
    n = len(res['pred'])
    rec = np.zeros((n, REC_WORDS), dtype=np.uint32)
    if n:
        rec[:, 0] = res['cand_contig'].astype(np.uint32) | (res['cand_type'].astype(np.uint32) << 16) | \
            (res['pred'].astype(np.uint32) << 24)
        rec[:, 1], rec[:, 2] = res['cand_pos'], res['cand_span']
        rec[:, 3], rec[:, 4] = res['support'], res['ps']
    return rec


def unpack_records(rec):
    rec = rec.reshape(-1, REC_WORDS)
    return dict(cand_contig=(rec[:, 0] & 0xFFFF).astype(np.uint16), cand_type=((rec[:, 0] >> 16) & 0xFF).astype(np.uint8),
                pred=(rec[:, 0] >> 24).astype(np.uint8), cand_pos=rec[:, 1].copy(), cand_span=rec[:, 2].copy(),
                support=rec[:, 3].astype(np.int64), ps=rec[:, 4].copy())


def rank_body(home, svlen_thres, suppread_thres, thread, include_all_ctgs, max_dist, rank, world, compute, to_device=None,
              star=None, gather=None):
    """One rank of the sharded SVIM mode.  compute as in phase_from_bams.  Rank 0 appends the rows to the file that
    already holds the header.  -> exit code (5: division by zero on some rank).
    star / gather (duet_amd/comm.py; round 4): how many candidates a rank finds is a result, not an input, so the ranks first
    tell each other their counts -- 16 bytes each, a control message over the TCP star that also carried RCCL's id -- and then
    ONE all-gather (gather.allgather: RCCL inside libduet_ef.so, or the star in the one-GPU plumbing mode) moves the fixed-size
    candidate records.  Without them: torch.distributed's default group for both (the CPU tests over gloo, DUET_COMM=torch);
    to_device: where its tensors live."""
    from duet_amd import dist as D
    chroms = init_chrom_list(include_all_ctgs, home)
    owned = D.lpt_assign(bam_weights(home, chroms), world)
    status, res = 0, None
    try:
        res = phase_from_bams(home, svlen_thres, suppread_thres, max(1, int(thread) // world), include_all_ctgs,
                              max_dist=max_dist, min_sv_size=max(int(svlen_thres), 1), only=set(owned[rank]), compute=compute)
    except ZeroDivisionError:
        status = 5
    rec = pack_records(res) if res is not None else np.zeros((0, REC_WORDS), dtype=np.uint32)
    if gather is not None:
        mine = np.array([len(rec), status, res['n_marks'] if res is not None else 0, 0], dtype=np.int32)
        counts = np.frombuffer(b''.join(star.allgather(mine.tobytes())), dtype=np.int32).reshape(world, 4)
        if int(counts[:, 1].max()) != 0:
            return 5
        n_max = max(int(counts[:, 0].max()), 1)
        slot = np.zeros(n_max * REC_WORDS, dtype=np.uint32)
        slot[:rec.size] = rec.reshape(-1)
        g = gather.allgather(slot.view(np.uint8))                 # the ONE data-path collective: fixed-size candidate records
        if rank != 0:
            return 0
        g = np.ascontiguousarray(g).view(np.uint32).reshape(world, n_max, REC_WORDS)
        return _merge_and_write(home, chroms, g, counts, world)
    import torch
    import torch.distributed as td
    dev = to_device if to_device is not None else torch.device('cpu')
    mine = torch.tensor([len(rec), status, res['n_marks'] if res is not None else 0, 0], dtype=torch.int32, device=dev)
    counts = torch.empty(4 * world, dtype=torch.int32, device=dev)
    td.all_gather_into_tensor(counts, mine)                       # 16 bytes per rank: how large the records' slots must be
    counts = counts.cpu().numpy().reshape(world, 4)
    if int(counts[:, 1].max()) != 0:
        return 5
    n_max = max(int(counts[:, 0].max()), 1)
    slot = torch.zeros(n_max * REC_WORDS, dtype=torch.int32, device=dev)
    if len(rec):
        slot[:rec.size] = torch.from_numpy(rec.reshape(-1).view(np.int32)).to(dev)
    gathered = torch.empty(world * n_max * REC_WORDS, dtype=torch.int32, device=dev)
    td.all_gather_into_tensor(gathered, slot)                     # the ONE data-path collective: fixed-size candidate records
    if rank != 0:
        return 0
    g = gathered.cpu().numpy().view(np.uint32).reshape(world, n_max, REC_WORDS)
    return _merge_and_write(home, chroms, g, counts, world)


def _merge_and_write(home, chroms, g, counts, world):
    """rank 0: the gathered records [world, n_max, REC_WORDS] -> rows appended to phased_sv.vcf"""
    parts = [unpack_records(g[r, :int(counts[r, 0])]) for r in range(world)]
    merged = {k: np.concatenate([p_[k] for p_ in parts]) for k in parts[0]}
    # contigs are owned whole and a rank's candidates come contig-major: a stable sort by contig is the single-GPU order
    order = np.argsort(merged['cand_contig'], kind='stable')
    merged = {k: v[order] for k, v in merged.items()}
    merged['chroms'] = chroms
    logging.info('  %d SV marks clustered into %d candidates on %d GPUs, %d phased (clustering rule: parity unpinned)' % (
        int(counts[:, 2].sum()), len(merged['pred']), world, int(np.count_nonzero(merged['pred']))))
    logging.info('write phased callset into .vcf file')
    with open(home + '/phased_sv.vcf', 'a') as out:
        out.write(rows_text(home, merged))
    return 0


def rank_main(argv):
    """Entry of one rank process: python -m duet_amd.svim_mode HOME SVLEN SUPP THREAD ALL_CTGS MAX_DIST"""
    import datetime
    home, svlen_thres, suppread_thres, thread = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    all_ctgs, max_dist = argv[4] == '1', float(argv[5])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    one_gpu = os.environ.get('DUET_ONE_GPU') == '1'
    device_id = 0 if one_gpu else int(os.environ.get('LOCAL_RANK', rank))
    if os.environ.get('DUET_COMM', '') != 'torch':
        # (no torch in this process: see duet_amd/multi.py rank_main)
        os.environ['DUET_NO_TORCH'] = '1'
        from duet_amd import _lib, comm
        if rank == 0:
            from duet_amd.utils import add_stream_logging
            add_stream_logging(home)
        ctx = _lib.Context(device_id)                    # raises when libduet_ef.so / the GPU is missing: no fallback
        star = comm.TcpStar(rank, world, timeout=float(os.environ.get('DUET_RDZV_TIMEOUT', '300')))
        gather = None
        try:
            gather = comm.HostGather(star) if one_gpu else comm.RcclGather(ctx, star)
            return rank_body(home, svlen_thres, suppread_thres, thread, all_ctgs, max_dist, rank, world, host_compute(ctx),
                             star=star, gather=gather)
        finally:
            if gather is not None:
                gather.close()
            star.close()
    import torch
    import torch.distributed as td
    from duet_amd import _lib
    if rank == 0:
        from duet_amd.utils import add_stream_logging
        add_stream_logging(home)
    torch.cuda.set_device(device_id)
    limit = datetime.timedelta(seconds=float(os.environ.get('DUET_RDZV_TIMEOUT', '300')))
    if one_gpu:
        td.init_process_group('gloo', rank=rank, world_size=world, timeout=limit)
    else:
        td.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device_id), timeout=limit)
    try:
        ctx = _lib.Context(device_id)                    # raises when libduet_ef.so / the GPU is missing: no fallback
        return rank_body(home, svlen_thres, suppread_thres, thread, all_ctgs, max_dist, rank, world, device_compute(ctx),
                         to_device=None if one_gpu else torch.device('cuda', device_id))
    finally:
        td.destroy_process_group()


def sv_phasing_from_bams(home, svlen_thres, suppread_thres, thread, include_all_ctgs, cluster_max_distance=0.9, device=0,
                         gpus=1):
    """`duet ... -b svim-gpu -c <max distance>`: SV calling (signatures + clustering, what `-b svim` delegates to the
    external `svim alignment ... --cluster_max_distance c`, sv_calling.py:13-15) AND SV phasing on the GPU, from the
    haplotagged BAMs of <home>/snp_phasing -> <home>/phased_sv.vcf.  The clustering half is this repository's own rule
    (parity unpinned); step E/F is the pinned one."""
    bar = '*' * 25
    logging.info('%s SV CALLING + PHASING (GPU, svim-gpu) STARTED %s' % (bar, bar))
    t0 = time.time()
    logging.info('create output .vcf file')
    chroms = init_chrom_list(include_all_ctgs, home)
    out_vcf = home + '/phased_sv.vcf'
    with open(out_vcf, 'w') as out:
        out.write(header_text(home, chroms))
    logging.info('extract SNP and SV signatures from the haplotagged alignments')
    if int(gpus) > 1 or os.environ.get('DUET_FORCE_RANKS') == '1':       # (see sv_phasing.py)
        from duet_amd import launch
        argv = ['-m', 'duet_amd.svim_mode', home, str(int(svlen_thres)), str(int(suppread_thres)), str(int(thread)),
                '1' if include_all_ctgs else '0', repr(float(cluster_max_distance))]
        env = {'PYTHONPATH': os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                             ([os.environ['PYTHONPATH']] if os.environ.get('PYTHONPATH') else []))}
        for h in logging.getLogger().handlers:
            h.flush()
        if any(isinstance(h, logging.FileHandler) for h in logging.getLogger().handlers):
            env['DUET_RANK_LOG'] = '1'
        rc = launch.spawn_ranks(int(gpus), argv, extra_env=env, timeout=float(os.environ.get('DUET_RANK_TIMEOUT', '3600')))
        if rc == 5:
            raise ZeroDivisionError('division by zero')
        if rc == 124:
            raise RuntimeError('svim-gpu on %d GPUs: the ranks did not finish within DUET_RANK_TIMEOUT; they were killed' % int(gpus))
        if rc:
            raise RuntimeError('svim-gpu on %d GPUs failed: a rank exited with code %d' % (int(gpus), rc))
        logging.info('%s SV CALLING + PHASING COMPLETED IN %ss %s' % (bar, round(time.time() - t0, 3), bar))
        return
    res = phase_from_bams(home, svlen_thres, suppread_thres, thread, include_all_ctgs, max_dist=cluster_max_distance,
                          min_sv_size=max(int(svlen_thres), 1), ctx=engine.default_context(int(device)))
    logging.info('  %d SV marks clustered into %d candidates, %d phased (clustering rule: parity unpinned)' % (
        res['n_marks'], len(res['pred']), int(np.count_nonzero(res['pred']))))
    logging.info('write phased callset into .vcf file')
    with open(out_vcf, 'a') as out:
        out.write(rows_text(home, res))
    logging.info('%s SV CALLING + PHASING COMPLETED IN %ss %s' % (bar, round(time.time() - t0, 3), bar))


if __name__ == '__main__':
    import sys
    sys.exit(rank_main(sys.argv[1:]))
