# coding=utf-8
"""Contig sharding of one E/F problem over the GPUs of a node.

Every quantity of step E/F is per contig -- tag dicts (sv_phasing_fn.py:15-18), the join (:47), the
seed sets (:195-203), the decisions (:206-212) -- so contigs are independent units until the final
sort (:229).  Ranks therefore own whole contigs (longest-processing-time-first on mark count), run
the three kernels locally with no data-path exchange, and reassemble the per-candidate (pred, ps)
records with ONE all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
One process per GPU; the process group is the caller's (torch.distributed).
"""

import numpy as np

from duet_amd import engine


def lpt_assign(weights, n_ranks):
    """Longest-processing-time-first: heaviest contig to the lightest rank. -> list (per rank) of
    ascending contig indices. Deterministic (ties: lower contig index first, lower rank first)."""
    order = sorted(range(len(weights)), key=lambda k: (-int(weights[k]), k))
    load = [0] * n_ranks
    owned = [[] for _ in range(n_ranks)]
    for k in order:
        r = min(range(n_ranks), key=lambda i: (load[i], i))
        owned[r].append(k)
        load[r] += int(weights[k])
    return [sorted(o) for o in owned]


def contig_mark_counts(soa):
    off = soa.cand_off.astype(np.int64)
    return off[soa.cand_ctg_off[1:].astype(np.int64)] - off[soa.cand_ctg_off[:-1].astype(np.int64)]


def shard_soa(soa, contig_ids):
    """Sub-problem holding only `contig_ids` (ascending), with reads and marks re-based."""
    ctg_off, read_off, cand_off = [0], [0], [np.zeros(1, dtype=np.int64)]
    tags, marks = [], []
    cols = {n: [] for n in ('cand_pos', 'cand_svlen', 'cand_svread', 'cand_refread', 'cand_gt_ok')}
    mbase = 0
    coff = soa.cand_off.astype(np.int64)
    for k in contig_ids:
        c0, c1 = int(soa.cand_ctg_off[k]), int(soa.cand_ctg_off[k + 1])
        r0, r1 = int(soa.read_off[k]), int(soa.read_off[k + 1])
        m0, m1 = int(coff[c0]), int(coff[c1])
        tags.append(soa.read_tag[r0:r1])
        m = soa.mark_read[m0:m1].astype(np.int64)
        marks.append(np.where(m == engine.MARK_ABSENT, engine.MARK_ABSENT, m - r0 + read_off[-1]))
        for n in cols:
            cols[n].append(getattr(soa, n)[c0:c1])
        cand_off.append(coff[c0 + 1:c1 + 1] - m0 + mbase)
        mbase += m1 - m0
        ctg_off.append(ctg_off[-1] + (c1 - c0))
        read_off.append(read_off[-1] + (r1 - r0))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)
    return engine.EfSoA(cand_ctg_off=ctg_off, read_off=read_off, read_tag=cat(tags, np.uint64),
                        cand_pos=cat(cols['cand_pos'], np.uint32), cand_svlen=cat(cols['cand_svlen'], np.uint32),
                        cand_svread=cat(cols['cand_svread'], np.uint32),
                        cand_refread=cat(cols['cand_refread'], np.uint32),
                        cand_gt_ok=cat(cols['cand_gt_ok'], np.uint8), cand_off=np.concatenate(cand_off),
                        mark_read=cat(marks, np.uint32))


def shard_sizes(soa, owned):
    """Candidates per rank for an assignment."""
    cnt = np.diff(soa.cand_ctg_off.astype(np.int64))
    return [int(sum(cnt[k] for k in o)) for o in owned]


def record_bytes(n_cands_max):
    """Bytes of one rank's all-gather slot: ps u32[n] followed by pred u8[n], padded to 16."""
    return (5 * int(n_cands_max) + 15) // 16 * 16


def allgather_records(local_block, world_size, group=None):
    """ONE collective: every rank contributes its fixed-size uint8 block (see record_bytes).
    local_block: torch.uint8 tensor [record_bytes] on the rank's device (or CPU for gloo).
    -> torch.uint8 tensor [world_size, record_bytes]"""
    import torch
    import torch.distributed as dist
    out = torch.empty(world_size * local_block.numel(), dtype=torch.uint8, device=local_block.device)
    dist.all_gather_into_tensor(out, local_block, group=group)
    return out.view(world_size, local_block.numel())


class GroupedGather(object):
    """The all-gather of many jobs' record blocks, `group` jobs per collective.

    `storage` is one uint8 tensor of n_groups * group blocks of `rb` bytes (a job's results go into the block
    next_slot() returns); when a group is full its blocks go out in ONE asynchronous all_gather_into_tensor, which
    overlaps the jobs of the next group(s); a group's buffer is reused only after its collective has completed.
    With world == 1 nothing is communicated and the slots simply rotate -- unless `always` is set (a one-rank process
    group over the real backend: the collective machinery runs with nobody to talk to)."""

    def __init__(self, storage, rb, world, group, dist_mod=None, always=False):
        self.storage, self.rb, self.world, self.dist = storage, int(rb), int(world), dist_mod
        self.comm = int(world) > 1 or (bool(always) and dist_mod is not None)
        self.G = int(group) if self.comm else 1
        self.n_groups = max(1, storage.numel() // (self.rb * self.G))
        self.pending = [None] * self.n_groups
        self.gathered = [None] * self.n_groups
        if self.comm:
            import torch                                     # receive buffers up front: no allocation inside a timed region
            for g in range(self.n_groups):
                self.gathered[g] = torch.empty(self.world * self.G * self.rb, dtype=torch.uint8, device=storage.device)
        self.filled, self.group, self.last, self.slot = 0, 0, None, 0

    def next_slot(self):
        """Slot (block index in `storage`) for the next job; waits for the group's previous collective first."""
        g = self.group
        if self.filled == 0 and self.pending[g] is not None:
            self.pending[g].wait()
            self.pending[g] = None
        self.slot = g * self.G + self.filled
        return self.slot

    def job_enqueued(self):
        """The job writing into the slot of the last next_slot() has been enqueued on the current stream."""
        self.filled += 1
        if self.filled == self.G:
            self.flush()

    def flush(self):
        g, k = self.group, self.filled
        if self.comm and k:
            import torch
            if self.gathered[g] is None:
                self.gathered[g] = torch.empty(self.world * self.G * self.rb, dtype=torch.uint8, device=self.storage.device)
            src = self.storage[g * self.G * self.rb:(g * self.G + k) * self.rb]
            self.pending[g] = self.dist.all_gather_into_tensor(self.gathered[g][:self.world * k * self.rb], src, async_op=True)
            self.last = (g, k)
        self.filled = 0
        self.group = (g + 1) % self.n_groups

    def drain(self):
        self.flush()
        for g in range(self.n_groups):
            if self.pending[g] is not None:
                self.pending[g].wait()
                self.pending[g] = None

    def last_job_blocks(self):
        """After drain(): uint8 [world, rb], every rank's block of the most recent job (None when world == 1)."""
        if not self.comm or self.last is None:
            return None
        g, k = self.last
        return self.gathered[g][:self.world * k * self.rb].view(self.world, k, self.rb)[:, k - 1, :]


def unpack_block(block_u8, n_cands_max, n_cands):
    """numpy uint8 block -> (pred u8[n_cands], ps u32[n_cands])"""
    ps = block_u8[:4 * n_cands_max].view(np.uint32)[:n_cands]
    pred = block_u8[4 * n_cands_max:4 * n_cands_max + n_cands]
    return pred.copy(), ps.copy()


def merge_results(soa, owned, per_rank):
    """per_rank[r] = (pred, ps) of rank r's shard (its contigs ascending) -> arrays in the original
    candidate order of `soa`."""
    pred = np.zeros(soa.n_cands, dtype=np.uint8)
    ps = np.zeros(soa.n_cands, dtype=np.uint32)
    for r, contigs in enumerate(owned):
        at = 0
        for k in contigs:
            c0, c1 = int(soa.cand_ctg_off[k]), int(soa.cand_ctg_off[k + 1])
            pred[c0:c1] = per_rank[r][0][at:at + c1 - c0]
            ps[c0:c1] = per_rank[r][1][at:at + c1 - c0]
            at += c1 - c0
    return pred, ps
