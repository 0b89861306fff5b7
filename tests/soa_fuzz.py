# coding=utf-8
"""Random E/F problems built directly in SoA form (no text), for HIP-vs-oracle parity tests.
Distributions are biased to the decision thresholds; knobs force the rare device paths."""
import numpy as np

from duet_amd import engine, synth

ABSENT = engine.MARK_ABSENT


def random_soa(seed, n_contigs=3, cands_per_contig=(50, 400), reads_per_contig=(10, 300), n_ps=(1, 6),
               deg=(1, 20), big_deg=0, empty_contig_rate=8, no_seed_contig_rate=8, ps_spread=40000,
               sorted_pos=True, absent_rate=5, allow_divzero=False):
    rng = synth.SplitMix(0x50A00000 + seed)
    pc_edge = np.array(synth._PC_EDGE, dtype=np.int64)
    ratio_edge = np.array(synth._RATIO_EDGE, dtype=np.int64)
    tags, ctg_off, read_off = [], [0], [0]
    cols = dict(pos=[], svlen=[], svread=[], refread=[], gt=[])
    offs, marks = [np.zeros(1, dtype=np.int64)], []
    mbase = 0
    for k in range(n_contigs):
        C = 0 if (empty_contig_rate and rng.one(empty_contig_rate) == 0) else \
            cands_per_contig[0] + rng.one(cands_per_contig[1] - cands_per_contig[0] + 1)
        R = reads_per_contig[0] + rng.one(reads_per_contig[1] - reads_per_contig[0] + 1)
        P = n_ps[0] + rng.one(n_ps[1] - n_ps[0] + 1)
        ps_vals = rng.between(P, 1, ps_spread)
        hap = rng.between(R, 1, 2)
        pc = np.where(rng.chance(R, 1, 2), pc_edge[rng.below(R, len(pc_edge))], rng.between(R, 0, 10000))
        if no_seed_contig_rate and rng.one(no_seed_contig_rate) == 0:
            pc = pc + 8101                      # nobody votes -> no seed -> contig dropped
        grp = (np.arange(R) * P // max(R, 1)) % P
        grp = np.where(rng.chance(R, 1, 8), rng.below(R, P), grp)
        tags.append(engine.pack_tags(hap, pc, ps_vals[grp]))
        if C:
            d = rng.between(C, deg[0], deg[1])
            if big_deg:
                d[rng.below(max(1, C // 50), C)] = big_deg
            off = np.zeros(C + 1, dtype=np.int64)
            np.cumsum(d, out=off[1:])
            M = int(off[-1])
            centre = rng.below(C, R)
            spread = 1 + rng.one(max(2, R // P))
            mr = (np.repeat(centre, d) + rng.below(M, spread)) % R + read_off[-1]
            mr = np.where(rng.chance(M, 1, absent_rate), ABSENT, mr) if absent_rate else mr
            p = rng.between(C, 1, ps_spread + 5000)
            cols['pos'].append(np.sort(p) if sorted_pos else p)
            cols['svlen'].append(np.where(rng.chance(C, 1, 10), rng.between(C, 45, 55), rng.between(C, 50, 5000)))
            edge = rng.chance(C, 1, 2)
            pr = ratio_edge[rng.below(C, len(ratio_edge))]
            sv = np.where(edge, pr[:, 0], rng.between(C, 1, 30))
            rf = np.where(edge, pr[:, 1], rng.between(C, 0, 30))
            if allow_divzero:
                z = rng.chance(C, 1, 30)
                sv, rf = np.where(z, 0, sv), np.where(z, 0, rf)
            cols['svread'].append(sv)
            cols['refread'].append(rf)
            cols['gt'].append((rng.below(C, 6) != 0).astype(np.uint8))
            offs.append(off[1:] + mbase)
            marks.append(mr)
            mbase += M
        ctg_off.append(ctg_off[-1] + C)
        read_off.append(read_off[-1] + R)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)
    return engine.EfSoA(cand_ctg_off=ctg_off, read_off=read_off, read_tag=cat(tags, np.uint64),
                        cand_pos=cat(cols['pos'], np.uint32), cand_svlen=cat(cols['svlen'], np.uint32),
                        cand_svread=cat(cols['svread'], np.uint32), cand_refread=cat(cols['refread'], np.uint32),
                        cand_gt_ok=cat(cols['gt'], np.uint8), cand_off=np.concatenate(offs),
                        mark_read=cat(marks, np.uint32))
