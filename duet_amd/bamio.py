# coding=utf-8
"""Minimal BGZF/BAM writer and reader (stdlib zlib + struct only).

Why it exists: the reference obtains read haplotype tags by running `samtools view` on the
per-contig haplotagged BAMs and looking at the LAST THREE whitespace-separated fields of every text
line (sv_phasing_fn.py:25-29).  Neither samtools nor pysam is available in the build image or on
the GPU box, so the host side carries its own reader that reproduces exactly those three tokens,
and a writer so that the "synthetic BAM" workloads are real BAM files.

Layout follows the SAM/BAM specification (SAMv1 section 4): BGZF = concatenated gzip members with
a 'BC' extra subfield holding the block size; BAM = magic, header text, reference list, then
alignment records with typed auxiliary fields.
"""

import struct
import zlib

_BGZF_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
_SEQ_CODE = '=ACMGRSVTWYHKDBN'


# ---------------------------------------------------------------------------------------------
# BGZF
# ---------------------------------------------------------------------------------------------

def _bgzf_block(data, level=6):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = len(body) + 25          # total block length - 1
    if bsize > 65535:
        raise ValueError('BGZF block too large')
    head = struct.pack('<BBBBIBBHBBHH', 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    tail = struct.pack('<II', zlib.crc32(data) & 0xFFFFFFFF, len(data))
    return head + body + tail


def bgzf_compress(payload, level=6, chunk=0xFF00):
    out = []
    for i in range(0, len(payload), chunk):
        out.append(_bgzf_block(payload[i:i + chunk], level))
    out.append(_BGZF_EOF)
    return b''.join(out)


def bgzf_decompress(blob):
    """Concatenated BGZF members -> bytes. Any gzip member is accepted (BSIZE is not required)."""
    out = []
    pos = 0
    n = len(blob)
    while pos < n:
        if blob[pos:pos + 2] != b'\x1f\x8b':
            raise ValueError('not a gzip/BGZF stream at offset %d' % pos)
        d = zlib.decompressobj(31)
        out.append(d.decompress(blob[pos:]))
        out.append(d.flush())
        used = n - pos - len(d.unused_data)
        if used <= 0:
            raise ValueError('truncated BGZF stream')
        pos += used
    return b''.join(out)


# ---------------------------------------------------------------------------------------------
# writer (from SAM text lines)
# ---------------------------------------------------------------------------------------------

def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _encode_aux(field):
    tag, typ, val = field.split(':', 2)
    t = tag.encode('ascii')
    if typ == 'i':
        v = int(val)
        if -128 <= v < 0:
            return t + b'c' + struct.pack('<b', v)
        if 0 <= v < 256:
            return t + b'C' + struct.pack('<B', v)
        if -32768 <= v < 0:
            return t + b's' + struct.pack('<h', v)
        if 0 <= v < 65536:
            return t + b'S' + struct.pack('<H', v)
        if v < 0:
            return t + b'i' + struct.pack('<i', v)
        return t + b'I' + struct.pack('<I', v)
    if typ == 'A':
        return t + b'A' + val.encode('ascii')[:1]
    if typ == 'f':
        return t + b'f' + struct.pack('<f', float(val))
    if typ == 'Z':
        return t + b'Z' + val.encode('ascii') + b'\0'
    if typ == 'H':
        return t + b'H' + val.encode('ascii') + b'\0'
    if typ == 'B':
        sub, rest = val[0], val[2:].split(',') if len(val) > 2 else []
        fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}[sub]
        conv = float if sub == 'f' else int
        return t + b'B' + sub.encode('ascii') + struct.pack('<I', len(rest)) + \
            b''.join(struct.pack('<' + fmt, conv(x)) for x in rest)
    raise ValueError('unsupported aux type ' + typ)


def encode_record(line, ref_index):
    f = line.rstrip('\n').split('\t')
    qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
    name = qname.encode('ascii') + b'\0'
    ref_id = ref_index.get(rname, -1)
    next_id = ref_id if rnext == '=' else ref_index.get(rnext, -1)
    p0 = int(pos) - 1
    cig = b''
    n_cig = 0
    ref_len = 0
    if cigar != '*':
        num = ''
        for ch in cigar:
            if ch.isdigit():
                num += ch
            else:
                op = 'MIDNSHP=X'.index(ch)
                cig += struct.pack('<I', (int(num) << 4) | op)
                if op in (0, 2, 3, 7, 8):
                    ref_len += int(num)
                n_cig += 1
                num = ''
    l_seq = 0 if seq == '*' else len(seq)
    sq = bytearray((l_seq + 1) // 2)
    for i in range(l_seq):
        code = _SEQ_CODE.index(seq[i].upper()) if seq[i].upper() in _SEQ_CODE else 15
        sq[i >> 1] |= code << (4 if (i & 1) == 0 else 0)
    if l_seq and qual != '*':
        ql = bytes(ord(ch) - 33 for ch in qual)
    else:
        ql = b'\xff' * l_seq
    aux = b''.join(_encode_aux(x) for x in f[11:])
    end = p0 + (ref_len if ref_len else 1)
    core = struct.pack('<iiBBHHHIiii', ref_id, p0, len(name), int(mapq), _reg2bin(max(p0, 0), max(end, 1)),
                       n_cig, int(flag), l_seq, next_id, int(pnext) - 1, int(tlen))
    body = core + name + cig + bytes(sq) + ql + aux
    return struct.pack('<I', len(body)) + body


def write_bam_from_sam_lines(path, refs, lines, level=1):
    """refs: list of (name, length). lines: SAM alignment lines (no header)."""
    text = '@HD\tVN:1.6\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in refs)
    tb = text.encode('ascii')
    parts = [b'BAM\1', struct.pack('<I', len(tb)), tb, struct.pack('<I', len(refs))]
    for name, ln in refs:
        nb = name.encode('ascii') + b'\0'
        parts.append(struct.pack('<I', len(nb)) + nb + struct.pack('<I', int(ln)))
    ref_index = {r[0]: i for i, r in enumerate(refs)}
    for ln in lines:
        parts.append(encode_record(ln, ref_index))
    with open(path, 'wb') as f:
        f.write(bgzf_compress(b''.join(parts), level))


# ---------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------

_INT_FMT = {ord('c'): ('<b', 1), ord('C'): ('<B', 1), ord('s'): ('<h', 2), ord('S'): ('<H', 2),
            ord('i'): ('<i', 4), ord('I'): ('<I', 4)}


def _fmt_float(x):
    # samtools prints %g
    return '%g' % x


def _real_cigar_from_cg(buf, aux_at, end, n_cig, cig_at, l_seq, ref_id, pos):
    """SAMv1 section 4.2.2: a CIGAR of more than 65535 operations does not fit the 16-bit n_cigar_op field; BAM then
    stores the placeholder `<l_seq>S<ref_len>N` and the real operations in a CG:B:I tag.  htslib (and therefore
    `samtools view`, which the reference parses) moves them back on reading and DROPS the CG tag -- under exactly
    these conditions: the record is placed (ref_id >= 0, pos >= 0), its first stored operation is a soft clip of the
    whole read, and a CG tag of type B with subtype I or i holds at least n_cigar_op values.
    -> (tuple of real cigar words, (start, end) of the CG field inside the aux block) or (None, None)."""
    if n_cig == 0 or ref_id < 0 or pos < 0:
        return None, None
    first = struct.unpack_from('<I', buf, cig_at)[0]
    if (first & 15) != 4 or (first >> 4) != l_seq:
        return None, None
    p = aux_at
    while p + 3 <= end:
        tag, typ = buf[p:p + 2], buf[p + 2]
        q = p + 3
        if typ in _INT_FMT:
            q += _INT_FMT[typ][1]
        elif typ == 65:
            q += 1
        elif typ == 102:
            q += 4
        elif typ in (90, 72):
            q = buf.index(b'\0', q) + 1
        elif typ == 66:
            sub = buf[q]
            cnt = struct.unpack_from('<I', buf, q + 1)[0]
            w = 4 if sub == 102 else _INT_FMT[sub][1]
            if tag == b'CG':
                if sub in (ord('I'), ord('i')) and n_cig <= cnt < (1 << 29):
                    return struct.unpack_from('<%dI' % cnt, buf, q + 5), (p, q + 5 + w * cnt)
                return None, None
            q += 5 + w * cnt
        else:
            raise ValueError('bad aux type byte %d' % typ)
        if tag == b'CG':
            return None, None                   # a CG tag of another type is left alone
        p = q
    return None, None


def _aux_fields(buf, p, end, skip=None):
    """Decode all aux fields of one record into SAM text fields (`skip` = byte range of a field to leave out)."""
    out = []
    while p < end:
        if skip is not None and p == skip[0]:
            p = skip[1]
            continue
        tag = buf[p:p + 2].decode('ascii')
        typ = buf[p + 2]
        p += 3
        if typ in _INT_FMT:
            fmt, w = _INT_FMT[typ]
            out.append('%s:i:%d' % (tag, struct.unpack_from(fmt, buf, p)[0]))
            p += w
        elif typ == 65:      # 'A'
            out.append('%s:A:%s' % (tag, chr(buf[p])))
            p += 1
        elif typ == 102:     # 'f'
            out.append('%s:f:%s' % (tag, _fmt_float(struct.unpack_from('<f', buf, p)[0])))
            p += 4
        elif typ in (90, 72):  # 'Z', 'H'
            q = buf.index(b'\0', p)
            out.append('%s:%s:%s' % (tag, chr(typ), buf[p:q].decode('ascii', 'replace')))
            p = q + 1
        elif typ == 66:      # 'B'
            sub = buf[p]
            cnt = struct.unpack_from('<I', buf, p + 1)[0]
            p += 5
            if sub == 102:
                vals = struct.unpack_from('<%df' % cnt, buf, p)
                p += 4 * cnt
                txt = ','.join(_fmt_float(v) for v in vals)
            else:
                fmt, w = _INT_FMT[sub]
                vals = struct.unpack_from('<%d%s' % (cnt, fmt[1]), buf, p)
                p += w * cnt
                txt = ','.join(str(v) for v in vals)
            out.append('%s:B:%s%s' % (tag, chr(sub), (',' + txt) if cnt else ''))
        else:
            raise ValueError('bad aux type byte %d' % typ)
    return out


def read_refs(path):
    """The reference list of a BAM's header, [(name, length)], decompressing only as many BGZF members as it spans."""
    with open(path, 'rb') as f:
        blob = f.read(1 << 22)
    buf, pos = b'', 0

    def need(n):
        nonlocal buf, pos
        while len(buf) < n and pos < len(blob):
            d = zlib.decompressobj(31)
            buf += d.decompress(blob[pos:]) + d.flush()
            pos = len(blob) - len(d.unused_data)
        return len(buf) >= n

    if not need(12) or buf[:4] != b'BAM\1':
        return []
    p = 8 + struct.unpack_from('<I', buf, 4)[0]
    if not need(p + 4):
        return []
    n_ref = struct.unpack_from('<I', buf, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        if not need(p + 4):
            break
        l_name = struct.unpack_from('<I', buf, p)[0]
        if not need(p + 8 + l_name):
            break
        refs.append((buf[p + 4:p + 4 + l_name - 1].decode('ascii'), struct.unpack_from('<I', buf, p + 4 + l_name)[0]))
        p += 8 + l_name
    return refs


def iter_bam_records(path):
    """Yield (qname, sam_fields list[str]) per alignment, fields as `samtools view` prints them."""
    with open(path, 'rb') as f:
        blob = f.read()
    if not blob:
        return
    buf = bgzf_decompress(blob)
    if buf[:4] != b'BAM\1':
        raise ValueError(path + ': not a BAM file')
    l_text = struct.unpack_from('<I', buf, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from('<I', buf, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from('<I', buf, p)[0]
        refs.append(buf[p + 4:p + 4 + l_name - 1].decode('ascii'))
        p += 4 + l_name + 4
    n = len(buf)
    while p + 4 <= n:
        bs = struct.unpack_from('<I', buf, p)[0]
        rec_end = p + 4 + bs
        (ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, next_id, next_pos, tlen) = \
            struct.unpack_from('<iiBBHHHIiii', buf, p + 4)
        q = p + 36
        qname = buf[q:q + l_name - 1].decode('ascii')
        q += l_name
        cig_at = q
        q += 4 * n_cig
        seq_at = q
        q += (l_seq + 1) // 2
        qual_at = q
        q += l_seq
        real_cigar, cg_field = _real_cigar_from_cg(buf, q, rec_end, n_cig, cig_at, l_seq, ref_id, pos)
        aux = _aux_fields(buf, q, rec_end, cg_field)

        def mandatory(n_cig=n_cig, cig_at=cig_at, real_cigar=real_cigar, seq_at=seq_at, qual_at=qual_at, l_seq=l_seq,
                      ref_id=ref_id, next_id=next_id, qname=qname, flag=flag, pos=pos, mapq=mapq, next_pos=next_pos, tlen=tlen):
            if real_cigar is not None:
                cigar = ''.join('%d%s' % (o >> 4, 'MIDNSHP=X'[o & 15]) for o in real_cigar)
            elif n_cig:
                ops = struct.unpack_from('<%dI' % n_cig, buf, cig_at)
                cigar = ''.join('%d%s' % (o >> 4, 'MIDNSHP=X'[o & 15]) for o in ops)
            else:
                cigar = '*'
            if l_seq:
                sq = buf[seq_at:seq_at + (l_seq + 1) // 2]
                seq = ''.join(_SEQ_CODE[(sq[i >> 1] >> (4 if (i & 1) == 0 else 0)) & 15] for i in range(l_seq))
                ql = buf[qual_at:qual_at + l_seq]
                qual = '*' if ql[0] == 255 else ''.join(chr(b + 33) for b in ql)
            else:
                seq, qual = '*', '*'
            rname = refs[ref_id] if 0 <= ref_id < len(refs) else '*'
            if next_id < 0:
                rnext = '*'
            elif next_id == ref_id:
                rnext = '='
            else:
                rnext = refs[next_id]
            return [qname, str(flag), rname, str(pos + 1), str(mapq), cigar, rnext, str(next_pos + 1),
                    str(tlen), seq, qual]

        yield qname, mandatory, aux
        p = rec_end


def tail_tokens(mandatory, aux):
    """The last three whitespace-separated tokens of the text line `samtools view` would print.

    Fast path: >= 3 aux fields, none of the last three containing whitespace."""
    if len(aux) >= 3:
        t = aux[-3:]
        if not any((' ' in x) or ('\t' in x) for x in t):
            return t
    toks = '\t'.join(mandatory() + aux).split()
    return toks[-3:]
