# coding=utf-8
"""ef_classify decides classes 0 and 1 in INTEGERS where integers provably agree with the binary64 expressions of predict_hp
(sv_phasing_fn.py:112-183) and says `need_fp` where they might not (duet_amd/csrc/duet_ef.hip: decide01_int; the kernel then evaluates
the binary64 form for that candidate).  This is the claim itself, checked on the CPU: the integer rules restated in Python's exact
integers against the binary64 expressions in Python floats (IEEE binary64, the arithmetic the reference runs in) --
on a dense grid around every threshold, on random numbers up to the 32-bit range, and on the 20,000 boundary-biased known answers
captured from the reference (tests/golden/kat_random.npz)."""
import os
import random

import numpy as np

from oracle import ef_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def decide_fp(cls, h1, h2, t1, t2, allhap, deg, s, r):
    """classes 0 / 1 of predict_hp on the vote's numbers, binary64 as upstream (oracle/ef_oracle.py: decide)."""
    hapread_ratio = allhap / deg if deg else float('nan')
    a1 = t1 / h1 if h1 > 0 else 0
    a2 = t2 / h2 if h2 > 0 else 0
    sv_ratio = s / (s + r)
    lo, hi = min(t1, t2), max(t1, t2)
    totsc_ratio = hi / lo if lo > 0 else 0
    onehap = hi if lo == 0 else 0
    diff = abs(a2 - a1)
    if cls == 0:
        return 3 if (sv_ratio == 1 and s >= 4) else 0
    gate = hapread_ratio > 0.75 or (hapread_ratio <= 0.75 and diff <= 2400)
    pred = 0
    if onehap != 0:
        if sv_ratio <= 0.24:
            pred = 0
        elif sv_ratio <= 0.9:
            if gate:
                pred = 1 if a1 > 0 else 2
        elif gate:
            pred = 3
    else:
        if sv_ratio <= 0.3:
            pred = 0
        elif sv_ratio <= 0.45:
            pred = 0 if r > 10 else (1 if t1 > t2 else 2)
        elif sv_ratio <= 0.75:
            pred = 3 if totsc_ratio <= 9.72 else (1 if t1 > t2 else 2)
        else:
            pred = 3
    return pred


def decide_int(cls, h1, h2, t1, t2, allhap, deg, s, r):
    """decide01_int of duet_ef.hip, line by line -> (pred, need_fp).  32-bit arithmetic there: every intermediate is checked to stay
    below 2^31 here (the ranges need_fp guards)."""
    sr = s + r
    need_fp = (sr >> 26) != 0 or ((t1 | t2) >> 21) != 0 or ((h1 | h2) >> 8) != 0 or (deg >> 28) != 0
    if cls == 0:
        return (3 if (r == 0 and s >= 4) else 0), False
    if need_fp:
        return 0, True
    for x in (s * 25, sr * 9, s * 20, 4 * allhap, 3 * deg, t2 * h1, t1 * h2, 2400 * h1 * h2, 25 * max(t1, t2), 243 * min(t1, t2)):
        assert x < 2 ** 31
    le024, le09, le03, le045, le075 = s * 25 <= sr * 6, s * 10 <= sr * 9, s * 10 <= sr * 3, s * 20 <= sr * 9, s * 4 <= sr * 3
    hp_le = deg != 0 and 4 * allhap <= 3 * deg
    hp_gt = deg != 0 and not hp_le
    lo, hi = min(t1, t2), max(t1, t2)
    diff_le = True
    if h1 != 0 and h2 != 0:
        A, B = abs(t2 * h1 - t1 * h2), 2400 * h1 * h2
        need_fp = A == B and t1 != 0 and t2 != 0
        diff_le = A <= B
    elif (h1 | h2) != 0:
        diff_le = (t1 if h1 != 0 else t2) <= 2400 * (h1 | h2)
    gate = (hp_le and diff_le) or hp_gt
    pred = 0
    if lo == 0 and hi != 0:
        if le024:
            pred = 0
        elif le09:
            if gate:
                pred = 1 if (h1 != 0 and t1 != 0) else 2
        elif gate:
            pred = 3
    else:
        ratio_le = lo == 0 or 25 * hi <= 243 * lo
        if le03:
            pred = 0
        elif le045:
            pred = 0 if r > 10 else (1 if t1 > t2 else 2)
        elif le075:
            pred = 3 if ratio_le else (1 if t1 > t2 else 2)
        else:
            pred = 3
    return pred, need_fp


def agree(cls, h1, h2, t1, t2, allhap, deg, s, r):
    pi, nf = decide_int(cls, h1, h2, t1, t2, allhap, deg, s, r)
    if nf:
        return None
    assert pi == decide_fp(cls, h1, h2, t1, t2, allhap, deg, s, r), (cls, h1, h2, t1, t2, allhap, deg, s, r)
    return True


def test_sv_ratio_thresholds_every_small_pair_and_32_bit_counts():
    n = 0
    for s in range(1, 140):
        for r in range(0, 140):
            for (h1, h2, t1, t2) in ((2, 0, 600, 0), (1, 1, 300, 100), (0, 0, 0, 0), (1, 1, 973, 100)):
                for cls in (0, 1):
                    n += agree(cls, h1, h2, t1, t2, h1 + h2, max(h1 + h2, 1), s, r) is True
    rng = random.Random(7)
    for pn, qd in ((6, 25), (9, 10), (3, 10), (9, 20), (3, 4), (1, 1)):
        for _ in range(20000):
            k = rng.randrange(1, ((2 ** 26 - 1) if rng.random() < 0.8 else (2 ** 32 - 1)) // qd)
            s0, tot = pn * k, qd * k
            for ds in (-1, 0, 1):
                s = s0 + ds
                if 1 <= s <= tot:
                    n += agree(1, 1, 1, 300, 100, 2, 2, s, tot - s) is True
                    n += agree(1, 2, 0, 600, 0, 2, 2, s, tot - s) is True
    assert n > 400000


def test_average_score_difference_and_totsc_ratio_thresholds():
    rng = random.Random(11)
    n = fb = 0
    # |t2 / h2 - t1 / h1| around 2400 for every small h1, h2 (inexact quotients), hapread_ratio <= 0.75 so that the gate looks at it
    for h1 in range(1, 24):
        for h2 in range(1, 24):
            for t1 in (0, 1, h1 * 17, h1 * 2400, h1 * 5000 + 3):
                base = 2400 * h1 * h2 + t1 * h2                          # t2 h1 = base  <=>  the difference is exactly 2400
                for d in range(-3, 4):
                    t2 = (base + h1 - 1) // h1 + d
                    if t2 < 0 or t2 > 8100 * h2 or t1 > 8100 * h1:
                        continue
                    for (s, r) in ((3, 3), (9, 1), (10, 1)):
                        for (tt1, tt2, hh1, hh2) in ((t1, t2, h1, h2), (t2, t1, h2, h1)):
                            ok = agree(1, hh1, hh2, tt1, tt2, hh1 + hh2, 2 * (hh1 + hh2), s, r)
                            n += ok is True
                            fb += ok is None
    # max / min around 9.72
    for lo in list(range(1, 3000)) + [rng.randrange(1, 2 ** 21 // 10) for _ in range(30000)]:
        for d in (-1, 0, 1):
            hi = (243 * lo) // 25 + d
            if hi < lo:
                continue
            for (s, r) in ((6, 4), (3, 1), (5, 5)):
                ok1 = agree(1, 3, 2, hi, lo, 5, 5, s, r)
                ok2 = agree(1, 2, 3, lo, hi, 5, 5, s, r)
                n += (ok1 is True) + (ok2 is True)
                fb += (ok1 is None) + (ok2 is None)
    assert n > 300000 and fb < n // 20                                # (the fall-back is for exact ties and big sums: rare)


def test_random_votes():
    rng = random.Random(13)
    n = 0
    for _ in range(300000):
        h1, h2 = rng.choice((0, 0, 1, 2, 3, 9, 40, 255, 70000)), rng.choice((0, 0, 1, 2, 5, 17, 255, 65535))
        t1 = rng.randrange(0, 8100 * h1 + 1) if h1 else 0
        t2 = rng.randrange(0, 8100 * h2 + 1) if h2 else 0
        if rng.random() < 0.2:
            t1 = 0
        extra = rng.randrange(0, 5)
        deg = h1 + h2 + extra + rng.randrange(0, 4)
        s, r = rng.choice(((rng.randrange(1, 60), rng.randrange(0, 60)), (rng.randrange(1, 2 ** 32), rng.randrange(0, 2 ** 32))))
        n += agree(rng.choice((0, 1, 1, 1)), h1, h2, t1, t2, h1 + h2, max(deg, 1), s, r) is True
    assert n > 150000


def test_reference_known_answers_classes_0_and_1():
    """the 20,000 vectors captured from the unmodified reference (make_golden.py): the integer form on the vote's numbers gives the
    reference's pred for every class-0 / class-1 vector (or asks for binary64)"""
    with np.load(os.path.join(HERE, 'golden', 'kat_random.npz')) as zf:
        z = {k: zf[k] for k in zf.files}
    off = z['off']
    sets = [set(int(x) for x in z['oneps_val'][z['oneps_off'][i]:z['oneps_off'][i + 1]]) for i in range(len(z['oneps_off']) - 1)]
    n = fb = 0
    for i in range(len(z['cls'])):
        cls = int(z['cls'][i])
        if cls == 2:
            continue
        cd = O.Candidate()
        cd.pos, cd.svread, cd.refread = int(z['pos'][i]), int(z['svread'][i]), int(z['refread'][i])
        a, b = off[i], off[i + 1]
        cd.marks = [(int(h), int(p), int(c)) if t else None
                    for t, h, p, c in zip(z['m_tagged'][a:b], z['m_hap'][a:b], z['m_ps'][a:b], z['m_pc'][a:b])]
        if cd.svread + cd.refread == 0:
            continue
        hap1, hap2, hap0, allhap, t1, t2, ps = O.vote(cd, cls, sets[int(z['oneps_set'][i])])
        pi, nf = decide_int(cls, hap1, hap2, t1, t2, allhap, len(cd.marks), cd.svread, cd.refread)
        fb += nf
        if not nf:
            assert pi == int(z['pred'][i]), i
            n += 1
    assert n > 10000 and fb < 500
