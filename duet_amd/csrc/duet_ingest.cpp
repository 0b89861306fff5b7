// duet_ingest.cpp -- native host-side ingest and row emission for Duet's step E/F (see include/duet_ingest.h).
// Plain C++17 + zlib; no GPU code.  Every rule below restates the reference's Python for well-formed
// ASCII input and bails out with DUET_INGEST_UNSUPPORTED otherwise, so that the Python host path (which
// mirrors upstream's exceptions) takes over.

#include <zlib.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "duet_ingest.h"

namespace {

constexpr uint32_t kAbsent = 0xFFFFFFFFu;
constexpr uint32_t kPcSat = (1u << 30) - 2;

struct Span {
    const char *p;
    size_t n;
};

inline bool is_py_space(unsigned char c)
{   // str.split() / str.strip() whitespace, ASCII part
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f);
}

// Python int(text) for plain ASCII decimals: [+-]digits, single underscores between digits.
bool py_int(const char *s, size_t n, long long &out)
{
    size_t i = 0;
    bool neg = false;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
    if (i >= n) return false;
    unsigned long long v = 0;
    bool prev_digit = false;
    int digits = 0;
    for (; i < n; ++i) {
        const char c = s[i];
        if (c >= '0' && c <= '9') {
            if (++digits > 18) return false;
            v = v * 10 + (unsigned)(c - '0');
            prev_digit = true;
        } else if (c == '_' && prev_digit && i + 1 < n && s[i + 1] >= '0' && s[i + 1] <= '9') {
            prev_digit = false;
        } else {
            return false;
        }
    }
    out = neg ? -(long long)v : (long long)v;
    return true;
}

const char *find_sub(const char *hay, size_t hn, const char *needle, size_t nn)
{
    if (nn > hn) return nullptr;
    return (const char *)memmem(hay, hn, needle, nn);
}

inline bool contains(Span s, const char *needle) { return find_sub(s.p, s.n, needle, strlen(needle)) != nullptr; }

// ---------------------------------------------------------------------------------------------
// read-name -> index table (open addressing over an arena of names)
// ---------------------------------------------------------------------------------------------
// Layout chosen for the join of ~10^6 mark names against ~10^5..10^7 read names, which is bound by cache
// misses: a probe touches one slot (hash tag + arena offset) and, on a tag match, one arena entry
// {index, length, bytes}; lookups are issued in batches with software prefetch (see NameBatch).
struct NameTable {
    std::vector<char> arena;          // entries: u32 index, u32 length, bytes
    std::vector<uint64_t> slots;      // (hash32 << 32) | (arena offset + 1), 0 = empty
    uint32_t count = 0, mask = 0;

    static constexpr uint64_t kFnvBasis = 1469598103934665603ull, kFnvPrime = 1099511628211ull;
    static uint64_t hash(const char *s, size_t n)
    {
        uint64_t h = kFnvBasis;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= kFnvPrime; }
        return h ^ (h >> 29);
    }
    void grow()
    {
        const uint32_t cap = slots.empty() ? 1024u : (uint32_t)slots.size() * 2u;
        std::vector<uint64_t> ns(cap, 0);
        mask = cap - 1;
        size_t o = 0;
        while (o < arena.size()) {
            uint32_t len;
            memcpy(&len, arena.data() + o + 4, 4);
            const uint64_t h = hash(arena.data() + o + 8, len);
            uint32_t s = (uint32_t)h & mask;
            while (ns[s]) s = (s + 1) & mask;
            ns[s] = ((h >> 32) << 32) | (uint64_t)(o + 1);
            o += 8 + len;
        }
        slots.swap(ns);
    }
    // probe with a precomputed hash; -1 when absent
    int find_hashed(const char *s, size_t n, uint64_t hh) const
    {
        if (slots.empty()) return -1;
        const uint32_t tag = (uint32_t)(hh >> 32);
        uint32_t h = (uint32_t)hh & mask;
        for (;;) {
            const uint64_t v = slots[h];
            if (!v) return -1;
            if ((uint32_t)(v >> 32) == tag) {
                const char *e = arena.data() + ((uint32_t)v - 1);
                uint32_t idx, len;
                memcpy(&idx, e, 4);
                memcpy(&len, e + 4, 4);
                if (len == n && memcmp(e + 8, s, n) == 0) return (int)idx;
            }
            h = (h + 1) & mask;
        }
    }
    int find(const char *s, size_t n) const { return find_hashed(s, n, hash(s, n)); }
    void prefetch_slot(uint64_t hh) const
    {
        if (!slots.empty()) __builtin_prefetch(&slots[(uint32_t)hh & mask]);
    }
    void prefetch_entry(uint64_t hh) const
    {   // first slot whose tag matches (if any): bring its arena entry in
        if (slots.empty()) return;
        const uint32_t tag = (uint32_t)(hh >> 32);
        uint32_t h = (uint32_t)hh & mask;
        for (int step = 0; step < 4; ++step) {
            const uint64_t v = slots[h];
            if (!v) return;
            if ((uint32_t)(v >> 32) == tag) { __builtin_prefetch(arena.data() + ((uint32_t)v - 1)); return; }
            h = (h + 1) & mask;
        }
    }
    // room for n more names without rehashing on the way
    void reserve(size_t n)
    {
        while (((size_t)count + n + 1) * 2 > slots.size()) grow();
    }
    uint32_t find_or_add(const char *s, size_t n, bool &added) { return find_or_add_hashed(s, n, hash(s, n), added); }
    uint32_t find_or_add_hashed(const char *s, size_t n, uint64_t hh, bool &added)
    {
        if (((size_t)count + 1) * 2 > slots.size()) grow();
        const uint32_t tag = (uint32_t)(hh >> 32);
        uint32_t h = (uint32_t)hh & mask;
        for (;;) {
            const uint64_t v = slots[h];
            if (!v) break;
            if ((uint32_t)(v >> 32) == tag) {
                const char *e = arena.data() + ((uint32_t)v - 1);
                uint32_t idx, len;
                memcpy(&idx, e, 4);
                memcpy(&len, e + 4, 4);
                if (len == n && memcmp(e + 8, s, n) == 0) { added = false; return idx; }
            }
            h = (h + 1) & mask;
        }
        const uint32_t idx = count++;
        const size_t o = arena.size();
        const uint32_t len = (uint32_t)n;
        arena.resize(o + 8 + n);
        memcpy(arena.data() + o, &idx, 4);
        memcpy(arena.data() + o + 4, &len, 4);
        memcpy(arena.data() + o + 8, s, n);
        slots[h] = ((uint64_t)tag << 32) | (uint64_t)(o + 1);
        added = true;
        return idx;
    }
};

// A whole file in memory that nobody has touched before the readers do: malloc, not a vector (whose resize would write every page
// from ONE thread first -- 30 ms per 100 MB), filled by `threads` readers with pread, each first-touching its own stretch.
struct RawBuf {
    char *p = nullptr;
    size_t n = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    const char *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { free(p); p = nullptr; n = 0; }
};

bool read_file_parallel(const char *path, RawBuf &buf, int threads)
{
    buf.clear();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 0) { close(fd); return false; }
    const size_t sz = (size_t)sb.st_size;
    if (sz == 0) { close(fd); return true; }
    buf.p = (char *)malloc(sz);
    if (!buf.p) { close(fd); return false; }
    buf.n = sz;
    int T = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    if (sz < (8u << 20)) T = 1;
    std::vector<char> ok((size_t)T, 1);
    auto work = [&](int t) {
        size_t at = sz * (size_t)t / (size_t)T;
        const size_t hi = sz * (size_t)(t + 1) / (size_t)T;
        while (at < hi) {
            const ssize_t got = pread(fd, buf.p + at, hi - at, (off_t)at);
            if (got <= 0) { ok[(size_t)t] = 0; return; }
            at += (size_t)got;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    close(fd);
    for (char c : ok)
        if (!c) { buf.clear(); return false; }
    return true;
}

bool read_file(const char *path, std::vector<char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    const size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return got == buf.size();
}

}  // namespace

struct duet_ingest {
    std::vector<std::string> contigs;
    std::unordered_map<std::string, int> owner;
    bool alias = false;
    std::vector<NameTable> tables;
    std::vector<std::vector<uint64_t>> tags;
    std::vector<uint8_t> bam_has_aln;             // per contig: its BAM printed at least one alignment line (:30-33)
    std::string err;
    // VCF
    RawBuf vcf;
    std::string vcf_path;                          // what `vcf` holds (duet_ingest_vcf_precount reads it first, parse reuses it)
    std::vector<uint8_t> skip;                     // per contig: its records are another rank's (duet_ingest_set_owned); empty = none
    std::vector<Span> contig_lines;               // first tokens containing '##contig=<ID='
    std::vector<uint32_t> cand_ctg_off, read_off, cand_pos, cand_svlen, cand_svread, cand_refread, cand_off, mark_read;
    std::vector<uint8_t> cand_gt_ok, cand_plus;   // cand_plus: svtype is exactly INS or DUP (sign rule, :225)
    std::vector<uint64_t> read_tag;
    std::vector<Span> c_chrom, c_ref, c_alt, c_type;
    bool parsed = false;
    // device-side row emission: the texts of every candidate in one pool (built on first request)
    std::vector<char> pool;
    struct FreeDeleter { void operator()(char *p) const { free(p); } };
    std::unique_ptr<char, FreeDeleter> pool_raw;     // the texts (malloc: first touched by the workers that fill it)
    std::vector<uint32_t> str_off;
    std::vector<uint16_t> chrom_rank;
    uint32_t n_chrom_texts = 0, max_pos = 0;
    bool rows_ready = false;
    int threads = 4;                               // what parse_vcf was given
    void *stage = nullptr;                         // between duet_ingest_parse_vcf_begin and _finish: the first half's result
    ~duet_ingest();
    // SVIM-mode signature extraction (optional, set before add_bam): CIGAR indels -> raw SV marks, binned depth
    bool extract = false;
    uint32_t min_sv_size = 40, min_mapq = 20, depth_bin = 1000;
    std::vector<uint16_t> m_contig;
    std::vector<uint8_t> m_type;
    std::vector<uint32_t> m_pos, m_span, m_local, m_read;      // m_local: index in the contig's tag table or kAbsent
    std::vector<std::vector<uint32_t>> depth;                   // per contig
    std::vector<uint32_t> depth_flat, depth_off, tag_off;
    std::vector<uint64_t> tag_flat;
};

namespace {

int unsupported(duet_ingest *g, const std::string &why)
{
    g->err = why;
    return DUET_INGEST_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// BGZF / BAM
// ---------------------------------------------------------------------------------------------
struct Block {
    size_t in_off, in_len, out_off, out_len;
};

int inflate_bgzf(duet_ingest *g, const std::vector<char> &file, std::vector<unsigned char> &out, int threads)
{
    std::vector<Block> blocks;
    size_t p = 0, total = 0;
    const unsigned char *d = (const unsigned char *)file.data();
    while (p < file.size()) {
        if (p + 18 > file.size() || d[p] != 31 || d[p + 1] != 139 || d[p + 2] != 8 || !(d[p + 3] & 4))
            return unsupported(g, "not a BGZF stream");
        const unsigned xlen = d[p + 10] | (d[p + 11] << 8);
        size_t q = p + 12, xend = p + 12 + xlen;
        int bsize = -1;
        while (q + 4 <= xend && xend <= file.size()) {
            const unsigned slen = d[q + 2] | (d[q + 3] << 8);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2) bsize = d[q + 4] | (d[q + 5] << 8);
            q += 4 + slen;
        }
        if (bsize < 0) return unsupported(g, "gzip member without BGZF block size");
        const size_t blen = (size_t)bsize + 1;
        if (p + blen > file.size() || blen < xlen + 20) return unsupported(g, "truncated BGZF block");
        const unsigned char *tail = d + p + blen - 4;
        const size_t isize = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((size_t)tail[3] << 24);
        blocks.push_back(Block{p + 12 + xlen, blen - xlen - 20, total, isize});
        total += isize;
        p += blen;
    }
    out.resize(total);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        z_stream zs;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size() || bad.load()) return;
            const Block &b = blocks[i];
            if (b.out_len == 0) continue;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            zs.next_in = (Bytef *)(d + b.in_off);
            zs.avail_in = (uInt)b.in_len;
            zs.next_out = out.data() + b.out_off;
            zs.avail_out = (uInt)b.out_len;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = 1; return; }
        }
    };
    int nt = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    if (blocks.size() < 8) nt = 1;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load()) return unsupported(g, "BGZF inflate failed");
    return DUET_INGEST_OK;
}

struct Aux {
    char tag[2];
    unsigned char type;
    size_t val_off;          // offset of the value bytes
    size_t next;
};

// length of one aux field's value; 0 on error
bool aux_step(const unsigned char *b, size_t p, size_t end, Aux &a)
{
    if (p + 3 > end) return false;
    a.tag[0] = (char)b[p];
    a.tag[1] = (char)b[p + 1];
    a.type = b[p + 2];
    a.val_off = p + 3;
    size_t q = p + 3;
    switch (a.type) {
        case 'c': case 'C': case 'A': q += 1; break;
        case 's': case 'S': q += 2; break;
        case 'i': case 'I': case 'f': q += 4; break;
        case 'Z': case 'H':
            while (q < end && b[q]) ++q;
            if (q >= end) return false;
            ++q;
            break;
        case 'B': {
            if (q + 5 > end) return false;
            const unsigned char sub = b[q];
            const size_t cnt = b[q + 1] | (b[q + 2] << 8) | (b[q + 3] << 16) | ((size_t)b[q + 4] << 24);
            const size_t w = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
            q += 5 + w * cnt;
            break;
        }
        default: return false;
    }
    if (q > end) return false;
    a.next = q;
    return true;
}

inline bool aux_is_int(unsigned char t) { return t == 'c' || t == 'C' || t == 's' || t == 'S' || t == 'i' || t == 'I'; }

long long aux_int(const unsigned char *b, const Aux &a)
{
    const unsigned char *v = b + a.val_off;
    switch (a.type) {
        case 'c': return (int8_t)v[0];
        case 'C': return v[0];
        case 's': return (int16_t)(v[0] | (v[1] << 8));
        case 'S': return (uint16_t)(v[0] | (v[1] << 8));
        case 'i': return (int32_t)(v[0] | (v[1] << 8) | (v[2] << 16) | ((uint32_t)v[3] << 24));
        default: return (uint32_t)(v[0] | (v[1] << 8) | (v[2] << 16) | ((uint32_t)v[3] << 24));
    }
}

bool aux_has_space(const unsigned char *b, const Aux &a)
{
    if (a.type == 'A') return is_py_space(b[a.val_off]);
    if (a.type == 'Z' || a.type == 'H') {
        for (size_t q = a.val_off; b[q]; ++q)
            if (is_py_space(b[q])) return true;
    }
    return false;
}

}  // namespace

namespace {

// What one alignment record contributes, as a pure function of its bytes (so records can be examined in parallel and
// applied in file order afterwards): a reason to decline, or the record's tag word when the last three whitespace tokens of
// its text line are HP / PC / PS (sv_phasing_fn.py:25-29) -- plus where its (real) CIGAR is.
struct BamRec {
    const char *why = nullptr;        // the native path declines the BAM
    const char *name = nullptr;
    uint32_t name_len = 0;
    bool tagged = false;
    uint64_t word = 0, hash = 0;      // hash of the name, for tagged records
    size_t cig_at = 0;
    uint32_t n_ops = 0;
};

void parse_bam_record(const unsigned char *b, size_t p, size_t end, BamRec &o)
{
    auto u32 = [&](size_t x) { return (uint32_t)(b[x] | (b[x + 1] << 8) | (b[x + 2] << 16) | ((uint32_t)b[x + 3] << 24)); };
    auto decline = [&](const char *why) { o.why = why; };
    const unsigned l_name = b[p + 12];
    const unsigned n_cig = b[p + 16] | (b[p + 17] << 8);
    const size_t l_seq = u32(p + 20);
    size_t q = p + 36;
    const char *name = (const char *)b + q;
    const size_t name_len = l_name ? l_name - 1 : 0;
    o.name = name;
    o.name_len = (uint32_t)name_len;
    q += l_name + 4 * (size_t)n_cig + (l_seq + 1) / 2 + l_seq;
    if (q > end) return decline("corrupt BAM record");
    // SAMv1 section 4.2.2: a CIGAR of more than 65535 operations is stored as the placeholder <l_seq>S<ref_len>N with
    // the real operations in a CG:B:I tag; htslib -- hence the `samtools view` text the reference parses -- moves
    // them back on reading and drops the tag, when the record is placed, its first stored operation soft-clips the
    // whole read and the (first) CG tag is a B array of I / i with at least n_cigar_op values.
    size_t cg_lo = 0, cg_hi = 0, cig_at = p + 36 + l_name;
    uint32_t n_ops = n_cig;
    if (n_cig && (int32_t)u32(p + 4) >= 0 && (int32_t)u32(p + 8) >= 0) {
        const uint32_t first = u32(cig_at);
        if ((first & 15u) == 4u && (first >> 4) == l_seq) {
            size_t a = q;
            while (a < end) {
                Aux x;
                if (!aux_step(b, a, end, x)) return decline("corrupt aux field");
                if (x.tag[0] == 'C' && x.tag[1] == 'G') {
                    if (x.type == 'B' && (b[x.val_off] == 'I' || b[x.val_off] == 'i')) {
                        const uint32_t cnt = u32(x.val_off + 1);
                        if (cnt >= n_cig && cnt < (1u << 29)) { cg_lo = a; cg_hi = x.next; cig_at = x.val_off + 5; n_ops = cnt; }
                    }
                    break;
                }
                a = x.next;
            }
        }
    }
    o.cig_at = cig_at;
    o.n_ops = n_ops;
    // the last three aux fields are the last three whitespace tokens of the text line -- provided there
    // are at least three and none of them contains whitespace
    Aux last[3];
    int n_aux = 0;
    while (q < end) {
        if (cg_hi && q == cg_lo) { q = cg_hi; continue; }       // the CG tag is not printed (see above)
        Aux a;
        if (!aux_step(b, q, end, a)) return decline("corrupt aux field");
        last[0] = last[1]; last[1] = last[2]; last[2] = a;
        ++n_aux;
        q = a.next;
    }
    if (n_aux < 3) {
        // The last three tokens then reach into the mandatory columns (..., SEQ, QUAL, aux...): tok[-2] is
        // aux[0] (2 aux fields), QUAL (1) or SEQ (0).  Such a line is skipped unless tok[-2] contains
        // 'PC:i:', in which case upstream goes on to int() a column that is not a tag -- left to Python.
        bool maybe = false;
        if (n_aux == 2) {
            const Aux &a0 = last[1];
            maybe = (aux_is_int(a0.type) && a0.tag[0] == 'P' && a0.tag[1] == 'C') || a0.type == 'Z' || a0.type == 'H' ||
                    aux_has_space(b, last[1]) || aux_has_space(b, last[2]);
        } else if (n_aux == 1) {
            static const unsigned char pat[5] = {'P' - 33, 'C' - 33, ':' - 33, 'i' - 33, ':' - 33};
            const unsigned char *ql = b + (p + 36 + l_name + 4 * (size_t)n_cig + (l_seq + 1) / 2);
            maybe = l_seq >= 5 && ql[0] != 255 && memmem(ql, l_seq, pat, 5) != nullptr;
            maybe = maybe || aux_has_space(b, last[2]);
        }
        if (maybe) return decline("alignment with fewer than three aux fields and a PC-like column");
        return;
    }
    // 'PC:i:' in tok[-2]
    const Aux &t2 = last[1];
    bool hit = aux_is_int(t2.type) && t2.tag[0] == 'P' && t2.tag[1] == 'C';
    if (!hit && (t2.type == 'Z' || t2.type == 'H')) {
        const char *v = (const char *)b + t2.val_off;
        if (strstr(v, "PC:i:")) return decline("string aux value containing 'PC:i:'");
    }
    if (hit) {
        if (!aux_is_int(last[0].type) || !aux_is_int(last[2].type))
            return decline("HP/PS neighbours of PC are not integers");
        const long long hap = aux_int(b, last[0]), pc = aux_int(b, t2), ps = aux_int(b, last[2]);
        if (pc < 0 || ps < 0 || ps > 0xFFFFFFFELL) return decline("PC/PS out of range");
        for (size_t i = 0; i < name_len; ++i)
            if ((unsigned char)name[i] >= 0x80) return decline("non-ASCII read name");
        const uint64_t code = (hap == 1 || hap == 2) ? (uint64_t)hap : 3ull;
        const uint64_t pcc = pc > (long long)kPcSat ? kPcSat : (uint64_t)pc;
        const uint64_t word = (code << 62) | (pcc << 32) | (uint64_t)ps;
        o.tagged = true;
        o.word = word;
        o.hash = NameTable::hash(name, name_len);
    }
}

}  // namespace

// header lines of phased_sv.vcf (write_file.py:19-45) appended to `out`
static int build_header(duet_ingest *g, int include_all_ctgs, std::string &out)
{
    out +=
        "##fileformat=VCFv4.2\n"
        "##source=Duet\n"
        "##ALT=<ID=INS,Description=\"Insertion of novel sequence relative to the reference\">\n"
        "##ALT=<ID=DEL,Description=\"Deletion relative to the reference\">\n"
        "##FILTER=<ID=PASS,Description=\"SV calls passed phasing criterion\">\n"
        "##INFO=<ID=SVLEN,Number=1,Type=Integer,Description=\"Estimated length of the variant\">\n"
        "##FORMAT=<ID=HP,Number=1,Type=String,Description=\"Haplotype of the SV call\">\n"
        "##FORMAT=<ID=PS,Number=1,Type=String,Description=\"Phase set which the SV call belongs to\">\n";
    if (include_all_ctgs) {
        for (const Span &l : g->contig_lines) { out.append(l.p, l.n); out += '\n'; }
    } else {
        const size_t lim = std::min<size_t>(24, g->contigs.size());
        if (g->contigs.size() < 24) return unsupported(g, "default mode walks 24 contigs");   // upstream: IndexError
        for (size_t k = 0; k < lim; ++k) {
            const std::string a = "##contig=<ID=chr" + g->contigs[k] + ",", b = "##contig=<ID=" + g->contigs[k] + ",";
            for (const Span &l : g->contig_lines)
                if (find_sub(l.p, l.n, a.data(), a.size()) || find_sub(l.p, l.n, b.data(), b.size())) {
                    out.append(l.p, l.n);
                    out += '\n';
                }
        }
    }
    out += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tVALUE\n";
    return DUET_INGEST_OK;
}

extern "C" {

duet_ingest *duet_ingest_create(int n_contigs, const char *const *contig_names)
{
    if (n_contigs < 0 || (n_contigs > 0 && !contig_names)) return nullptr;
    duet_ingest *g = new duet_ingest();
    g->contigs.reserve(n_contigs);
    for (int k = 0; k < n_contigs; ++k) {
        g->contigs.emplace_back(contig_names[k] ? contig_names[k] : "");
        const std::string &c = g->contigs.back();
        for (const std::string &nm : {std::string("chr") + c, c}) {
            auto it = g->owner.find(nm);
            if (it == g->owner.end()) g->owner.emplace(nm, k);
            else if (it->second != k) g->alias = true;
        }
    }
    g->tables.resize(n_contigs);
    g->tags.resize(n_contigs);
    g->bam_has_aln.assign(n_contigs, 0);
    return g;
}

void duet_ingest_destroy(duet_ingest *g) { delete g; }
const char *duet_ingest_error(const duet_ingest *g) { return g ? g->err.c_str() : "null"; }
void duet_ingest_free(void *p) { free(p); }

int duet_ingest_add_bam(duet_ingest *g, int contig, const char *path, int threads)
{
    if (!g || contig < 0 || contig >= (int)g->contigs.size() || !path) return DUET_INGEST_INVALID;
    const bool timing = getenv("DUET_INGEST_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[duet_ingest] bam %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<char> file;
    if (!read_file(path, file)) { g->err = std::string("cannot read ") + path; return DUET_INGEST_IO; }
    if (file.empty()) return DUET_INGEST_OK;                       // an empty file prints nothing (no alignments)
    lap("read");
    std::vector<unsigned char> buf;
    int rc = inflate_bgzf(g, file, buf, threads);
    if (rc) return rc;
    lap("inflate");
    const unsigned char *b = buf.data();
    const size_t n = buf.size();
    auto u32 = [&](size_t p) { return (uint32_t)(b[p] | (b[p + 1] << 8) | (b[p + 2] << 16) | ((uint32_t)b[p + 3] << 24)); };
    if (n < 12 || memcmp(b, "BAM\1", 4) != 0) return unsupported(g, "not a BAM file");
    size_t p = 8 + (size_t)u32(4);
    if (p + 4 > n) return unsupported(g, "truncated BAM header");
    const uint32_t n_ref = u32(p);
    p += 4;
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > n) return unsupported(g, "truncated BAM header");
        p += 4 + (size_t)u32(p) + 4;
    }
    NameTable &tab = g->tables[contig];
    std::vector<uint64_t> &tags = g->tags[contig];
    struct Pend { const char *name; uint32_t len; uint8_t type; uint32_t pos, span; };
    std::vector<Pend> pend;
    // split reads (SVIM mode): the kept alignments of one read name are its segments; in read orientation a segment covers
    // [qs, qe) (qs = leading clip, or the trailing clip of a reverse alignment) and [rs, re) on the reference
    struct Seg { const char *name; uint32_t len; uint64_t qs, qe, rs, re; uint32_t line; bool rev; };
    std::vector<Seg> segs;
    uint32_t line_no = 0;
    std::vector<int64_t> dep;
    // record boundaries first (a hop per record), then every record's contribution -- a pure function of its bytes, examined
    // in parallel when there are many of them -- applied in file order: the first declining record decides, later lines win
    std::vector<size_t> at;
    while (p + 4 <= n) {
        const size_t bs = u32(p), end = p + 4 + bs;
        if (end > n || bs < 32) return unsupported(g, "truncated BAM record");
        at.push_back(p);
        p = end;
    }
    const size_t NR = at.size();
    if (NR) g->bam_has_aln[contig] = 1;
    auto rec_end = [&](size_t i) { return at[i] + 4 + (size_t)u32(at[i]); };
    std::vector<BamRec> recs;
    // (also with ONE worker -- a contig of its own in duet_ingest_add_bams: the records examined first, the name table then filled with
    // the slot of the record eight ahead already asked for; the record-by-record path missed the cache once per read name)
    const bool batch = !g->extract && NR >= 4096;
    if (batch) {
        recs.resize(NR);
        const int T = threads > 32 ? 32 : (threads < 1 ? 1 : threads);
        auto work = [&](int t) {
            for (size_t i = NR * t / T, hi = NR * (t + 1) / T; i < hi; ++i) parse_bam_record(b, at[i], rec_end(i), recs[i]);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
        size_t tagged = 0;
        for (size_t i = 0; i < NR; ++i) {
            if (recs[i].why) return unsupported(g, recs[i].why);
            tagged += recs[i].tagged;
        }
        tab.reserve(tagged);
    }
    for (size_t ri = 0; ri < NR; ++ri) {
        const size_t p = at[ri];
        BamRec one;
        if (!batch) {
            parse_bam_record(b, p, rec_end(ri), one);
            if (one.why) return unsupported(g, one.why);
        }
        const BamRec &rec = batch ? recs[ri] : one;
        const char *name = rec.name;
        const size_t name_len = rec.name_len;
        const size_t cig_at = rec.cig_at;
        const uint32_t n_ops = rec.n_ops;
        if (g->extract) {
            // SVIM-mode signatures: insertions / deletions of at least min_sv_size inside the alignment's CIGAR
            // (primary and supplementary alignments with MAPQ >= min_mapq), and the alignment's span for the depth
            const unsigned flag = b[p + 18] | (b[p + 19] << 8), mapq = b[p + 13];
            const int32_t pos0 = (int32_t)u32(p + 8);
            if (!(flag & 0x104) && mapq >= g->min_mapq && pos0 >= 0 && n_ops) {
                uint64_t ref = (uint64_t)pos0;
                const size_t cg = cig_at;
                {
                    uint64_t lead = 0, trail = 0, aligned = 0;
                    uint32_t i = 0;
                    for (; i < n_ops; ++i) {
                        const uint32_t v = u32(cg + 4 * (size_t)i), op = v & 15u;
                        if (op != 4 && op != 5) break;
                        lead += v >> 4;
                    }
                    for (uint32_t j = n_ops; j > 0; --j) {
                        const uint32_t v = u32(cg + 4 * (size_t)(j - 1)), op = v & 15u;
                        if (op != 4 && op != 5) break;
                        trail += v >> 4;
                    }
                    uint64_t rlen = 0;
                    for (uint32_t k = 0; k < n_ops; ++k) {
                        const uint32_t v = u32(cg + 4 * (size_t)k), op = v & 15u;
                        if (op == 0 || op == 1 || op == 7 || op == 8) aligned += v >> 4;
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += v >> 4;
                    }
                    if (aligned) {
                        const bool rev = (flag & 0x10) != 0;
                        const uint64_t qs = rev ? trail : lead;
                        segs.push_back({name, (uint32_t)name_len, qs, qs + aligned, (uint64_t)pos0, (uint64_t)pos0 + rlen, line_no, rev});
                    }
                }
                for (uint32_t i = 0; i < n_ops; ++i) {
                    const uint32_t v = u32(cg + 4 * (size_t)i), op = v & 15u, len = v >> 4;
                    if (op == 1 || op == 2) {
                        if (len >= g->min_sv_size && ref < 0xFFFFFFFEull) {
                            pend.push_back({name, (uint32_t)name_len, (uint8_t)(op == 1 ? 1 : 0), (uint32_t)ref + 1u, len});
                        }
                        if (op == 2) ref += len;
                    } else if (op == 0 || op == 3 || op == 7 || op == 8) {
                        ref += len;
                    }
                }
                // depth[b] counts the alignments that cover the middle of bin b
                const uint64_t w = g->depth_bin, half = w / 2, s0 = (uint64_t)pos0, e0 = ref;
                if (e0 > s0) {
                    const uint64_t lo = s0 <= half ? 0 : (s0 - half + w - 1) / w;            // first b with b*w + half >= s0
                    const uint64_t hi = e0 <= half ? 0 : (e0 - half + w - 1) / w;            // first b with b*w + half >= e0
                    if (hi > lo) {
                        if (dep.size() < hi + 1) dep.resize(hi + 1, 0);
                        dep[lo] += 1;
                        dep[hi] -= 1;
                    }
                }
            }
        }
        ++line_no;
        if (rec.tagged) {
            if (batch && ri + 8 < NR && recs[ri + 8].tagged) tab.prefetch_slot(recs[ri + 8].hash);
            bool added;
            const uint32_t idx = tab.find_or_add_hashed(name, name_len, rec.hash, added);
            if (added) tags.push_back(rec.word); else tags[idx] = rec.word;      // later lines win (:29)
        }
    }
    lap("records");
    if (g->extract && !segs.empty()) {
        // split-read marks (oracle/svim_oracle.py: SVIM_inter.py's insertion / deletion cases and, round 4, its tandem-duplication
        // and inversion cases -- type codes DEL 0 / INS 1 / INV 2 / DUP 3): reads in order of first
        // appearance, their segments sorted by (qs, qe, line); consecutive segments on one strand whose gaps on the read
        // and on the reference differ by at least min_sv_size
        std::unordered_map<std::string, uint32_t> group_of;
        std::vector<std::vector<uint32_t>> groups;
        for (uint32_t i = 0; i < segs.size(); ++i) {
            auto it = group_of.emplace(std::string(segs[i].name, segs[i].len), (uint32_t)groups.size());
            if (it.second) groups.emplace_back();
            groups[it.first->second].push_back(i);
        }
        const int64_t tol = 5, max_del = 100000, min_sv = (int64_t)g->min_sv_size;
        for (auto &gr : groups) {
            if (gr.size() < 2) continue;
            std::sort(gr.begin(), gr.end(), [&](uint32_t x, uint32_t y) {
                const Seg &a = segs[x], &c = segs[y];
                if (a.qs != c.qs) return a.qs < c.qs;
                if (a.qe != c.qe) return a.qe < c.qe;
                return a.line < c.line;
            });
            for (size_t i = 0; i + 1 < gr.size(); ++i) {
                const Seg &a = segs[gr[i]], &c = segs[gr[i + 1]];
                const int64_t dread = (int64_t)c.qs - (int64_t)a.qe;
                if (a.rev != c.rev) {
                    // opposite strands: the segments meet at their right ends (forward, then reverse) or at their left ends: INV
                    if (dread < -tol) continue;
                    const int64_t p1 = a.rev ? (int64_t)a.rs : (int64_t)a.re, p2 = a.rev ? (int64_t)c.rs : (int64_t)c.re;
                    const int64_t lo = p1 < p2 ? p1 : p2, sp = p1 < p2 ? p2 - p1 : p1 - p2;
                    if (sp >= min_sv && sp <= max_del && lo < 0xFFFFFFFEll) pend.push_back({a.name, a.len, 2, (uint32_t)lo + 1u, (uint32_t)sp});
                    continue;
                }
                const int64_t dref = a.rev ? (int64_t)a.rs - (int64_t)c.re : (int64_t)c.rs - (int64_t)a.re;
                if (dread < -tol) continue;
                if (dref < -tol) {
                    // the read goes back on the reference: a tandem duplication of the stretch both segments cover: DUP
                    const int64_t s0 = a.rev ? (int64_t)a.rs : (int64_t)c.rs, e0 = a.rev ? (int64_t)c.re : (int64_t)a.re;
                    if (e0 - s0 >= min_sv && e0 - s0 <= max_del && s0 < 0xFFFFFFFEll) pend.push_back({a.name, a.len, 3, (uint32_t)s0 + 1u, (uint32_t)(e0 - s0)});
                    continue;
                }
                const int64_t dev = dread - dref;
                const uint64_t anchor = a.rev ? c.re : a.re;
                if (anchor >= 0xFFFFFFFEull) continue;
                if (dev >= min_sv && dev <= 0xFFFFFFFFll)
                    pend.push_back({a.name, a.len, 1, (uint32_t)anchor + 1u, (uint32_t)dev});
                else if (-dev >= min_sv && -dev <= max_del)
                    pend.push_back({a.name, a.len, 0, (uint32_t)anchor + 1u, (uint32_t)(-dev)});
            }
        }
    }
    if (g->extract) {
        for (const Pend &m : pend) {
            const int id = tab.find_hashed(m.name, m.len, NameTable::hash(m.name, m.len));
            g->m_contig.push_back((uint16_t)contig);
            g->m_type.push_back(m.type);
            g->m_pos.push_back(m.pos);
            g->m_span.push_back(m.span);
            g->m_local.push_back(id < 0 ? kAbsent : (uint32_t)id);
        }
        if (g->depth.size() < g->contigs.size()) g->depth.resize(g->contigs.size());
        std::vector<uint32_t> &d = g->depth[contig];
        d.assign(dep.empty() ? 0 : dep.size() - 1, 0);
        int64_t run = 0;
        for (size_t i = 0; i + 1 < dep.size(); ++i) {
            run += dep[i];
            d[i] = (uint32_t)run;
        }
    }
    return DUET_INGEST_OK;
}

// Several contigs' BAMs at once (round 6).  A contig's tag table, tag words and "has alignments" flag are its own, so whole
// contigs go to the workers -- the largest files first, a shared counter hands them out -- instead of one contig after the other with
// the workers meeting at every stage of each (inflate, records, then ONE thread filling that contig's name table in file order: with
// 24 contigs that serial part was what the BAM side cost).  With fewer contigs than workers every contig keeps `threads / n`
// workers of its own.  SVIM-mode extraction appends to shared arrays in contig order: it takes the contigs one by one as before.
// Returns the status of the first contig (in the caller's order) that failed, with that contig's message.
int duet_ingest_add_bams(duet_ingest *g, int n, const int *contigs, const char *const *paths, int threads)
{
    if (!g || n < 0 || (n && (!contigs || !paths))) return DUET_INGEST_INVALID;
    if (threads < 1) threads = 1;
    if (n <= 1 || threads == 1 || g->extract) {
        for (int i = 0; i < n; ++i) {
            const int rc = duet_ingest_add_bam(g, contigs[i], paths[i], threads);
            if (rc) return rc;
        }
        return DUET_INGEST_OK;
    }
    for (int i = 0; i < n; ++i)
        if (contigs[i] < 0 || contigs[i] >= (int)g->contigs.size() || !paths[i]) return DUET_INGEST_INVALID;
    std::vector<int> order((size_t)n);
    std::vector<long long> bytes((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        order[(size_t)i] = i;
        struct stat sb;
        if (stat(paths[i], &sb) == 0) bytes[(size_t)i] = (long long)sb.st_size;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bytes[(size_t)a] > bytes[(size_t)b]; });
    const int outer = n < threads ? n : threads, inner = threads / outer > 1 ? threads / outer : 1;
    std::vector<int> rcs((size_t)n, DUET_INGEST_OK);
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int j = next.fetch_add(1);
            if (j >= n) return;
            const int i = order[(size_t)j];
            rcs[(size_t)i] = duet_ingest_add_bam(g, contigs[i], paths[i], inner);
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < outer; ++t) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[(size_t)i]) {
            // (failing contigs may have written the message side by side: the first one's again, alone)
            g->tables[(size_t)contigs[i]] = NameTable();
            g->tags[(size_t)contigs[i]].clear();
            return duet_ingest_add_bam(g, contigs[i], paths[i], threads);
        }
    return DUET_INGEST_OK;
}

namespace {

struct Rec { Span tok[10]; };

// One pass over an INFO column: the first ';'-separated item CONTAINING each key (read_file.py:34,38,40,48 use
// `in`).  Every key ends with '=', so an item contains a key iff some '=' in it is preceded by the key's letters.
struct InfoHits { Span svlen{nullptr, 0}, svtype{nullptr, 0}, supp{nullptr, 0}, names{nullptr, 0}; };

inline bool ends_with(const char *item, size_t eq, const char *key, size_t kn)
{   // bytes item[eq-kn+1 .. eq] == key (key includes its trailing '=')
    return eq + 1 >= kn && memcmp(item + eq + 1 - kn, key, kn) == 0;
}

void scan_info(Span info, InfoHits &h)
{
    size_t i = 0;
    while (i <= info.n) {
        // the item's end and the '=' signs inside it, by memchr: most of an INFO column is the list of read names, which a
        // byte-by-byte loop walked at a byte per step (a third of the ingest's second phase)
        const char *semi = i < info.n ? (const char *)memchr(info.p + i, ';', info.n - i) : nullptr;
        const size_t j = semi ? (size_t)(semi - info.p) : info.n;
        bool len_ = false, type_ = false, supp_ = false, names_ = false;
        const char *it = info.p + i;
        for (const char *eq = i < j ? (const char *)memchr(it, '=', j - i) : nullptr; eq;
             eq = eq + 1 < info.p + j ? (const char *)memchr(eq + 1, '=', (size_t)(info.p + j - (eq + 1))) : nullptr) {
            const size_t e = (size_t)(eq - it);
            if (ends_with(it, e, "SVLEN=", 6)) len_ = true;
            if (ends_with(it, e, "SVTYPE=", 7)) type_ = true;
            if (ends_with(it, e, "SUPPORT=", 8) || ends_with(it, e, "SR=", 3) || ends_with(it, e, "RE=", 3)) supp_ = true;
            if (ends_with(it, e, "RNAMES=", 7) || ends_with(it, e, "READS=", 6)) names_ = true;
        }
        const Span item{info.p + i, j - i};
        if (len_ && !h.svlen.p) h.svlen = item;
        if (type_ && !h.svtype.p) h.svtype = item;
        if (supp_ && !h.supp.p) h.supp = item;
        if (names_ && !h.names.p) h.names = item;
        i = j + 1;
    }
}

struct Layout { size_t supp_cut, rn_cut; int fmt_kind; };

// what the first half of the VCF parse hands to the second (duet_ingest::stage between the two calls)
struct PartA { std::vector<std::vector<Rec>> per; std::vector<Span> contig_lines; std::string why; };
struct VcfStage {
    int rc = DUET_INGEST_OK;
    std::string err;
    int T = 1;
    std::vector<PartA> pa;
};

// Caller VCF, first half: read, index the lines, tokenise, pick the records of the listed contigs (parallel over lines).
// Touches only the VCF side of *g (vcf, contig_lines, threads) and reports through st.err -- it may run beside
// duet_ingest_add_bam calls on the same object.
int vcf_begin(duet_ingest *g, const char *path, int threads, VcfStage &st)
{
    auto decline = [&](const std::string &why) { st.err = why; return DUET_INGEST_UNSUPPORTED; };
    const bool timing = getenv("DUET_INGEST_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[duet_ingest] %-14s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    g->threads = threads > 0 ? threads : 1;
    if (g->alias) return decline("contig list names a contig twice");
    if (g->vcf_path != path || g->vcf.empty()) {
        if (!read_file_parallel(path, g->vcf, threads)) { st.err = std::string("cannot read ") + path; return DUET_INGEST_IO; }
        g->vcf_path = path;
    }
    const char *d = g->vcf.data();
    const size_t n = g->vcf.size();
    lap("read");
    const int K = (int)g->contigs.size();
    int T = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    if (n < (1u << 20)) T = 1;

    // ---- lines (universal newlines: \n, \r, \r\n) ------------------------------------------------
    // Every worker scans its stretch of the file once: NUL bytes, carriage returns, and the places of the '\n' (round 6: the read,
    // these checks and the line index were ONE thread's 140 ms per 4e6 marks, the largest serial piece of the ingest).  A file
    // with a '\r' anywhere takes the serial scan below (universal newlines cut lines in three ways).
    std::vector<size_t> line_lo, line_hi;
    {
        std::vector<std::vector<size_t>> nl((size_t)T);
        std::vector<char> has_nul((size_t)T, 0), has_cr_t((size_t)T, 0);
        auto scan = [&](int t) {
            const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
            if (lo >= hi) return;
            if (memchr(d + lo, 0, hi - lo)) { has_nul[(size_t)t] = 1; return; }
            if (memchr(d + lo, '\r', hi - lo)) { has_cr_t[(size_t)t] = 1; return; }
            std::vector<size_t> &v = nl[(size_t)t];
            v.reserve((hi - lo) / 256 + 16);
            size_t p = lo;
            while (p < hi) {
                const char *q = (const char *)memchr(d + p, '\n', hi - p);
                if (!q) break;
                v.push_back((size_t)(q - d));
                p = (size_t)(q - d) + 1;
            }
        };
        {
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back(scan, t);
            scan(0);
            for (auto &th : pool) th.join();
        }
        for (char c : has_nul)
            if (c) return decline("NUL byte in the VCF");
        bool has_cr = false;
        for (char c : has_cr_t) has_cr = has_cr || c;
        if (!has_cr) {
            size_t total = 0;
            std::vector<size_t> base((size_t)T + 1, 0);
            for (int t = 0; t < T; ++t) { base[(size_t)t] = total; total += nl[(size_t)t].size(); }
            // (the last line may lack its newline: the last newline over all stretches says)
            size_t last_nl = (size_t)-1;
            for (int t = T - 1; t >= 0 && last_nl == (size_t)-1; --t)
                if (!nl[(size_t)t].empty()) last_nl = nl[(size_t)t].back();
            const bool open_tail = n && (last_nl == (size_t)-1 || last_nl + 1 < n);
            const size_t L0 = total + (open_tail ? 1 : 0);
            line_lo.resize(L0);
            line_hi.resize(L0);
            auto fill = [&](int t) {
                const std::vector<size_t> &v = nl[(size_t)t];
                size_t at = base[(size_t)t];
                // the line that ends at v[i] starts behind the newline before it (the previous stretch's last one for i = 0)
                size_t prev = (size_t)-1;
                for (int u = t - 1; u >= 0 && prev == (size_t)-1; --u)
                    if (!nl[(size_t)u].empty()) prev = nl[(size_t)u].back();
                for (size_t i = 0; i < v.size(); ++i, ++at) {
                    line_lo[at] = prev + 1;            // ((size_t)-1 + 1 == 0: the file's first line)
                    line_hi[at] = v[i];
                    prev = v[i];
                }
            };
            {
                std::vector<std::thread> pool;
                for (int t = 1; t < T; ++t) pool.emplace_back(fill, t);
                fill(0);
                for (auto &th : pool) th.join();
            }
            if (open_tail) { line_lo[total] = last_nl + 1; line_hi[total] = n; }
        } else {
            size_t p = 0;
            while (p < n) {
                size_t e = p;
                while (e < n && d[e] != '\n' && d[e] != '\r') ++e;
                line_lo.push_back(p);
                line_hi.push_back(e);
                p = e + 1;
                if (e < n && d[e] == '\r' && p < n && d[p] == '\n') ++p;
            }
        }
    }
    const size_t L = line_lo.size();
    lap("lines");

    // ---- phase A (parallel over lines): tokenise, pick the records of listed contigs ------------------
    st.T = T;
    st.pa.assign(T, PartA());
    std::vector<PartA> &pa = st.pa;
    auto work_a = [&](int t) {
        PartA &o = pa[t];
        o.per.resize(K);
        const size_t lo = L * t / T, hi = L * (t + 1) / T;
        if (lo < hi) {      // non-ASCII bytes anywhere in this thread's stretch of the file
            unsigned char acc = 0;
            for (size_t i = line_lo[lo]; i < line_hi[hi - 1]; ++i) acc |= (unsigned char)d[i];
            if (acc & 0x80) { o.why = "non-ASCII byte in the VCF"; return; }
        }
        std::string key;
        for (size_t li = lo; li < hi; ++li) {
            const size_t p = line_lo[li], e = line_hi[li];
            Rec r;
            int nt = 0;
            size_t i = p;
            // the first token decides what the line is; only a listed contig's record needs the other nine
            while (i < e && is_py_space((unsigned char)d[i])) ++i;
            if (i >= e) {    // a blank line raises IndexError upstream (read_file.py:30)
                o.why = "blank line in the VCF";
                return;
            }
            {
                size_t j = i;
                while (j < e && !is_py_space((unsigned char)d[j])) ++j;
                r.tok[0] = Span{d + i, j - i};
                nt = 1;
                i = j;
            }
            if (contains(r.tok[0], "##contig=<ID=")) o.contig_lines.push_back(r.tok[0]);
            key.assign(r.tok[0].p, r.tok[0].n);
            auto it = g->owner.find(key);
            if (it != g->owner.end()) {
                if (!g->skip.empty() && g->skip[it->second]) continue;     // another rank's contig: that rank vouches for the record
                while (i < e) {
                    while (i < e && is_py_space((unsigned char)d[i])) ++i;
                    if (i >= e) break;
                    size_t j = i;
                    while (j < e && !is_py_space((unsigned char)d[j])) ++j;
                    if (nt < 10) r.tok[nt] = Span{d + i, j - i};
                    ++nt;
                    i = j;
                }
                if (nt < 10) { o.why = "record with fewer than 10 columns"; return; }
                // an 11th token shifts upstream's appended columns (read_file.py:37): TypeError in generate_callinfo
                if (nt > 10) { o.why = "record with more than 10 columns"; return; }
                o.per[it->second].push_back(r);
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(work_a, t);
        work_a(0);
        for (auto &th : pool) th.join();
    }
    lap("phase A");
    for (int t = 0; t < T; ++t)
        if (!pa[t].why.empty()) return decline(pa[t].why);
    g->contig_lines.clear();
    for (int t = 0; t < T; ++t) g->contig_lines.insert(g->contig_lines.end(), pa[t].contig_lines.begin(), pa[t].contig_lines.end());
    return DUET_INGEST_OK;
}

// ... second half: callset order, numbers, and the join of the mark names against the tag dicts (parallel over candidates)
int vcf_finish(duet_ingest *g, VcfStage &st)
{
    const bool timing = getenv("DUET_INGEST_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[duet_ingest] %-14s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int K = (int)g->contigs.size(), T = st.T;
    std::vector<PartA> &pa = st.pa;
    // callset order: contig-major, file order inside a contig
    std::vector<const Rec *> recs;
    g->cand_ctg_off.assign(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        for (int t = 0; t < T; ++t)
            for (const Rec &r : pa[t].per[k]) recs.push_back(&r);
        g->cand_ctg_off[k + 1] = (uint32_t)recs.size();
    }
    const size_t C = recs.size();

    g->read_off.assign(K + 1, 0);
    g->read_tag.clear();
    for (int k = 0; k < K; ++k) {
        g->read_off[k + 1] = g->read_off[k] + (uint32_t)g->tags[k].size();
        g->read_tag.insert(g->read_tag.end(), g->tags[k].begin(), g->tags[k].end());
    }

    lap("merge");
    // ---- layout switches from each contig's FIRST record (read_file.py:41,49,57,63,65) ---------------
    std::vector<Layout> layout(K, Layout{0, 0, 0});
    for (int k = 0; k < K; ++k) {
        if (g->cand_ctg_off[k] == g->cand_ctg_off[k + 1]) continue;
        const Rec &r0 = *recs[g->cand_ctg_off[k]];
        InfoHits h;
        scan_info(r0.tok[7], h);
        if (!h.supp.p) return unsupported(g, "first record without a support count");
        if (!h.names.p) return unsupported(g, "first record without read names");
        layout[k].supp_cut = contains(h.supp, "SUPPORT=") ? 8 : 3;
        layout[k].rn_cut = contains(h.names, "RNAMES=") ? 7 : 6;
        const Span s = r0.tok[9];
        int nsub = 1;
        size_t last = 0;
        for (size_t i = 0; i < s.n; ++i)
            if (s.p[i] == ':') { ++nsub; last = i + 1; }
        if (nsub < 3) return unsupported(g, "sample column with fewer than three fields");
        layout[k].fmt_kind = nsub > 4 ? 0 : (memchr(s.p + last, ',', s.n - last) ? 2 : 1);
    }

    // ---- phase B (parallel over candidates): numbers + join of the mark names ------------------------
    g->cand_pos.assign(C, 0); g->cand_svlen.assign(C, 0); g->cand_svread.assign(C, 0); g->cand_refread.assign(C, 0);
    g->cand_gt_ok.assign(C, 0); g->cand_plus.assign(C, 0);
    g->c_chrom.assign(C, Span{nullptr, 0}); g->c_ref.assign(C, Span{nullptr, 0}); g->c_alt.assign(C, Span{nullptr, 0});
    g->c_type.assign(C, Span{nullptr, 0});
    std::vector<uint32_t> deg(C, 0);
    struct PartB { std::vector<uint32_t> marks; std::string why; };
    int TB = C < 4096 ? 1 : T;
    std::vector<PartB> pb(TB);
    auto work_b = [&](int t) {
        PartB &o = pb[t];
        const size_t lo = C * t / TB, hi = C * (t + 1) / TB;
        // (collected in a vector of the thread's own and handed over at the end: the pb[] headers of all threads share a
        // cache line, and a push_back per mark through them made four threads as slow as one)
        std::vector<uint32_t> marks;
        marks.reserve((hi - lo) * 12);
        struct HandOver { std::vector<uint32_t> &from, &to; ~HandOver() { to.swap(from); } } hand_over{marks, o.marks};
        int k = 0;
        for (size_t c = lo; c < hi; ++c) {
            while (c >= g->cand_ctg_off[k + 1]) ++k;
            const Layout &ly = layout[k];
            const NameTable &tab = g->tables[k];
            const uint32_t rbase = g->read_off[k];
            const Rec &r = *recs[c];
            long long v;
            if (!py_int(r.tok[1].p, r.tok[1].n, v) || v < 0 || v > 0xFFFFFFFFLL) { o.why = "POS"; return; }
            g->cand_pos[c] = (uint32_t)v;
            InfoHits h;
            scan_info(r.tok[7], h);
            // SVLEN: missing or exactly 'SVLEN=.' -> 0; '>' anywhere in the item -> skip 7 characters (Q20)
            long long svlen = 0;
            if (h.svlen.p && !(h.svlen.n == 7 && memcmp(h.svlen.p, "SVLEN=.", 7) == 0)) {
                const size_t cut = memchr(h.svlen.p, '>', h.svlen.n) ? 7 : 6;
                if (h.svlen.n < cut || !py_int(h.svlen.p + cut, h.svlen.n - cut, svlen)) { o.why = "SVLEN"; return; }
            }
            if (svlen < 0) svlen = -svlen;
            if (svlen > 0xFFFFFFFFLL) { o.why = "SVLEN range"; return; }
            g->cand_svlen[c] = (uint32_t)svlen;
            if (!h.svtype.p || h.svtype.n < 7) { o.why = "SVTYPE"; return; }
            const Span ty{h.svtype.p + 7, h.svtype.n - 7};
            g->c_type[c] = ty;
            g->cand_plus[c] = (ty.n == 3 && (memcmp(ty.p, "INS", 3) == 0 || memcmp(ty.p, "DUP", 3) == 0)) ? 1 : 0;
            if (!h.supp.p || h.supp.n < ly.supp_cut || !py_int(h.supp.p + ly.supp_cut, h.supp.n - ly.supp_cut, v) ||
                v < 0 || v > 0xFFFFFFFFLL) { o.why = "support count"; return; }
            g->cand_svread[c] = (uint32_t)v;
            if (!h.names.p || h.names.n < ly.rn_cut) { o.why = "read names"; return; }
            {   // ','.split keeps empty names.  Lookups go in batches of 32: hash + prefetch the slots, prefetch the
                // matching arena entries, then compare -- the join is cache-miss bound, not compute bound.
                const char *s = h.names.p + ly.rn_cut;
                const size_t sn = h.names.n - ly.rn_cut;
                size_t i = 0;
                uint32_t cnt = 0;
                bool more = true;
                while (more) {
                    Span nm[32];
                    uint64_t hh[32];
                    int nb = 0;
                    while (nb < 32) {
                        size_t j = i;
                        uint64_t fnv = NameTable::kFnvBasis;           // (NameTable::hash, folded into the scan for the comma)
                        for (; j < sn && s[j] != ','; ++j) { fnv ^= (unsigned char)s[j]; fnv *= NameTable::kFnvPrime; }
                        nm[nb] = Span{s + i, j - i};
                        hh[nb] = fnv ^ (fnv >> 29);
                        tab.prefetch_slot(hh[nb]);
                        ++nb;
                        if (j >= sn) { more = false; break; }
                        i = j + 1;
                    }
                    for (int q = 0; q < nb; ++q) tab.prefetch_entry(hh[q]);
                    for (int q = 0; q < nb; ++q) {
                        const int idx = tab.find_hashed(nm[q].p, nm[q].n, hh[q]);
                        marks.push_back(idx < 0 ? kAbsent : rbase + (uint32_t)idx);
                    }
                    cnt += (uint32_t)nb;
                }
                deg[c] = cnt;
            }
            // sample column: up to the first three ':' fields and the last one
            const Span s = r.tok[9];
            Span sub[3] = {{nullptr, 0}, {nullptr, 0}, {nullptr, 0}}, lastf{nullptr, 0};
            {
                size_t i = 0;
                int f = 0;
                for (;;) {
                    size_t j = i;
                    while (j < s.n && s.p[j] != ':') ++j;
                    if (f < 3) sub[f] = Span{s.p + i, j - i};
                    lastf = Span{s.p + i, j - i};
                    ++f;
                    if (j >= s.n) break;
                    i = j + 1;
                }
                g->cand_gt_ok[c] = (sub[0].n == 3 && memcmp(sub[0].p, "./.", 3) == 0) ? 0 : 1;
                auto opt_int = [&](Span x, long long &out) {
                    if (x.n == 1 && x.p[0] == '.') { out = 0; return true; }
                    return py_int(x.p, x.n, out);
                };
                long long ref = 0, other = 0;
                if (ly.fmt_kind == 2) {
                    const char *cm = (const char *)memchr(lastf.p, ',', lastf.n);
                    // upstream: k = find(','); with k == -1 it slices [:-1] / [0:] -- left to the Python path
                    if (!cm) { o.why = "AD without a comma"; return; }
                    if (!opt_int(Span{lastf.p, (size_t)(cm - lastf.p)}, ref) ||
                        !opt_int(Span{cm + 1, lastf.n - (size_t)(cm - lastf.p) - 1}, other)) { o.why = "AD"; return; }
                } else {
                    if (f < 3 || !opt_int(sub[1], ref) || !opt_int(sub[2], other)) { o.why = "sample counts"; return; }
                }
                if (ref < 0 || ref > 0xFFFFFFFFLL) { o.why = "reference-read count range"; return; }
                g->cand_refread[c] = (uint32_t)ref;
            }
            g->c_chrom[c] = r.tok[0];
            g->c_ref[c] = r.tok[3];
            g->c_alt[c] = r.tok[4];
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < TB; ++t) pool.emplace_back(work_b, t);
        work_b(0);
        for (auto &th : pool) th.join();
    }
    lap("phase B");
    for (int t = 0; t < TB; ++t)
        if (!pb[t].why.empty()) return unsupported(g, pb[t].why);
    g->cand_off.assign(C + 1, 0);
    uint64_t total = 0;
    for (size_t c = 0; c < C; ++c) {
        total += deg[c];
        if (total > 0xFFFFFFF0ull) return unsupported(g, "too many marks");
        g->cand_off[c + 1] = (uint32_t)total;
    }
    g->mark_read.resize(total);
    {
        size_t at = 0;
        for (int t = 0; t < TB; ++t) {
            if (!pb[t].marks.empty()) memcpy(g->mark_read.data() + at, pb[t].marks.data(), pb[t].marks.size() * 4);
            at += pb[t].marks.size();
        }
    }
    lap("finish");
    g->parsed = true;
    return DUET_INGEST_OK;
}

}  // namespace

duet_ingest::~duet_ingest() { delete (VcfStage *)stage; }

int duet_ingest_parse_vcf(duet_ingest *g, const char *path, int threads)
{
    if (!g || !path) return DUET_INGEST_INVALID;
    VcfStage st;
    const int rc = vcf_begin(g, path, threads, st);
    if (rc) { g->err = st.err; return rc; }
    return vcf_finish(g, st);
}

int duet_ingest_parse_vcf_begin(duet_ingest *g, const char *path, int threads)
{
    if (!g || !path) return DUET_INGEST_INVALID;
    delete (VcfStage *)g->stage;
    VcfStage *st = new VcfStage;
    g->stage = st;
    st->rc = vcf_begin(g, path, threads, *st);
    return st->rc;
}

int duet_ingest_parse_vcf_finish(duet_ingest *g)
{
    if (!g || !g->stage) return DUET_INGEST_INVALID;
    VcfStage *st = (VcfStage *)g->stage;
    g->stage = nullptr;
    int rc = st->rc;
    if (rc) g->err = st->err;
    else rc = vcf_finish(g, *st);
    delete st;
    return rc;
}

int duet_ingest_get_arrays(const duet_ingest *g, duet_ingest_arrays *o)
{
    if (!g || !o || !g->parsed) return DUET_INGEST_INVALID;
    o->n_contigs = (uint32_t)g->contigs.size();
    o->n_cands = (uint32_t)g->cand_pos.size();
    o->n_marks = (uint32_t)g->mark_read.size();
    o->n_reads = (uint32_t)g->read_tag.size();
    o->cand_ctg_off = g->cand_ctg_off.data();
    o->read_off = g->read_off.data();
    o->read_tag = g->read_tag.data();
    o->cand_pos = g->cand_pos.data();
    o->cand_svlen = g->cand_svlen.data();
    o->cand_svread = g->cand_svread.data();
    o->cand_refread = g->cand_refread.data();
    o->cand_gt_ok = g->cand_gt_ok.data();
    o->cand_off = g->cand_off.data();
    o->mark_read = g->mark_read.data();
    return DUET_INGEST_OK;
}

// slot of a candidate's CHROM text: 2 * contig + (0: spelled chr<name>, 1: spelled <name>)
static inline uint32_t text_slot(const duet_ingest *g, uint32_t contig, const Span &chrom)
{
    const std::string &nm = g->contigs[contig];
    return 2u * contig + ((chrom.n == nm.size() && memcmp(chrom.p, nm.data(), nm.size()) == 0) ? 1u : 0u);
}

// rows of every candidate with pred != 0, sorted as upstream sorts them (:206-229).  id_base == null: numbered 1, 2, ... ;
// else the rows of CHROM-text slot s are numbered id_base[s], id_base[s] + 1, ... (a rank of a sharded run numbers its
// blocks as the whole call set would) and slot_off / slot_len receive each slot's byte range in `out`.
static int emit_rows(duet_ingest *g, const uint8_t *pred, const uint32_t *ps, const uint64_t *id_base, std::string &out,
                     uint64_t *slot_off, uint64_t *slot_len)
{
    const size_t C = g->cand_pos.size();
    // emission order (:206-228): contig, PS-class 0/1/2, file order; candidates are already contig-major
    std::vector<uint32_t> idx;
    std::vector<uint8_t> cls;
    for (size_t c = 0; c < C; ++c) {
        if (!pred[c]) continue;
        int n_ps = 0;
        uint32_t first = 0;
        for (uint32_t m = g->cand_off[c]; m < g->cand_off[c + 1]; ++m) {
            const uint32_t r = g->mark_read[m];
            if (r == kAbsent) continue;
            const uint32_t p = (uint32_t)g->read_tag[r];
            if (n_ps == 0) { n_ps = 1; first = p; }
            else if (p != first) { n_ps = 2; break; }
        }
        idx.push_back((uint32_t)c);
        cls.push_back((uint8_t)n_ps);
    }
    std::vector<uint32_t> ord(idx.size());
    for (uint32_t i = 0; i < ord.size(); ++i) ord[i] = i;
    const int K = (int)g->contigs.size();
    std::vector<uint32_t> ctg_of(idx.size());
    {
        int k = 0;
        for (size_t i = 0; i < idx.size(); ++i) {
            while (k < K && idx[i] >= g->cand_ctg_off[k + 1]) ++k;
            ctg_of[i] = (uint32_t)k;
        }
    }
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
        if (ctg_of[a] != ctg_of[b]) return ctg_of[a] < ctg_of[b];
        return cls[a] < cls[b];
    });
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {      // (:229) chrom as text, pos as int
        const Span &x = g->c_chrom[idx[a]], &y = g->c_chrom[idx[b]];
        const int c = memcmp(x.p, y.p, std::min(x.n, y.n));
        if (c) return c < 0;
        if (x.n != y.n) return x.n < y.n;
        return g->cand_pos[idx[a]] < g->cand_pos[idx[b]];
    });
    static const char *const kHp[4] = {"", "1|0", "0|1", "1|1"};
    char num[64];
    uint64_t row = 0;
    uint32_t cur_slot = 0xFFFFFFFFu;
    if (slot_off) for (int i = 0; i < 2 * K; ++i) { slot_off[i] = 0; slot_len[i] = 0; }
    for (uint32_t oi : ord) {
        const uint32_t c = idx[oi];
        if (id_base) {
            const uint32_t sl = text_slot(g, ctg_of[oi], g->c_chrom[c]);
            if (sl != cur_slot) {
                if (cur_slot != 0xFFFFFFFFu && slot_off) slot_len[cur_slot] = out.size() - slot_off[cur_slot];
                cur_slot = sl;
                row = id_base[sl] - 1;
                if (slot_off) slot_off[sl] = out.size();
            }
        }
        ++row;
        out.append(g->c_chrom[c].p, g->c_chrom[c].n);
        int w = snprintf(num, sizeof(num), "\t%u\tDuet.%llu\t", g->cand_pos[c], (unsigned long long)row);
        out.append(num, w);
        out.append(g->c_ref[c].p, g->c_ref[c].n);
        out += '\t';
        out.append(g->c_alt[c].p, g->c_alt[c].n);
        const uint32_t mag = g->cand_svlen[c];
        if (g->cand_plus[c] || mag == 0) w = snprintf(num, sizeof(num), "\t.\tPASS\tSVLEN=%u;SVTYPE=<", mag);
        else w = snprintf(num, sizeof(num), "\t.\tPASS\tSVLEN=-%u;SVTYPE=<", mag);
        out.append(num, w);
        out.append(g->c_type[c].p, g->c_type[c].n);
        w = snprintf(num, sizeof(num), ">\tHP:PS\t%s:%u\n", kHp[pred[c] & 3], ps[c]);
        out.append(num, w);
    }
    if (id_base && cur_slot != 0xFFFFFFFFu && slot_off) slot_len[cur_slot] = out.size() - slot_off[cur_slot];
    return DUET_INGEST_OK;
}

static int hand_over(const std::string &out, char **text, uint64_t *len)
{
    char *buf = (char *)malloc(out.size() + 1);
    if (!buf) return DUET_INGEST_INVALID;
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *text = buf;
    *len = out.size();
    return DUET_INGEST_OK;
}

int duet_ingest_emit(duet_ingest *g, const uint8_t *pred, const uint32_t *ps, int include_all_ctgs, char **text,
                     uint64_t *len)
{
    if (!g || !g->parsed || !text || !len) return DUET_INGEST_INVALID;
    const size_t C = g->cand_pos.size();
    if (C && (!pred || !ps)) return DUET_INGEST_INVALID;
    std::string out;
    out.reserve(4096 + C * 48);
    if (include_all_ctgs >= 0) {
        const int hrc = build_header(g, include_all_ctgs, out);
        if (hrc) return hrc;
    }
    const int rc = emit_rows(g, pred, ps, nullptr, out, nullptr, nullptr);
    if (rc) return rc;
    return hand_over(out, text, len);
}

int duet_ingest_count_kept(duet_ingest *g, const uint8_t *pred, uint64_t *kept)
{
    if (!g || !g->parsed || !kept) return DUET_INGEST_INVALID;
    const size_t C = g->cand_pos.size();
    if (C && !pred) return DUET_INGEST_INVALID;
    const int K = (int)g->contigs.size();
    for (int i = 0; i < 2 * K; ++i) kept[i] = 0;
    int k = 0;
    for (size_t c = 0; c < C; ++c) {
        while (k < K && c >= g->cand_ctg_off[k + 1]) ++k;
        if (pred[c]) ++kept[text_slot(g, (uint32_t)k, g->c_chrom[c])];
    }
    return DUET_INGEST_OK;
}

int duet_ingest_cand_slots(duet_ingest *g, uint32_t *slot)
{
    if (!g || !g->parsed) return DUET_INGEST_INVALID;
    const size_t C = g->cand_pos.size();
    if (C && !slot) return DUET_INGEST_INVALID;
    const int K = (int)g->contigs.size();
    int k = 0;
    for (size_t c = 0; c < C; ++c) {
        while (k < K && c >= g->cand_ctg_off[k + 1]) ++k;
        slot[c] = (uint32_t)text_slot(g, (uint32_t)k, g->c_chrom[c]);
    }
    return DUET_INGEST_OK;
}

int duet_ingest_emit_blocks(duet_ingest *g, const uint8_t *pred, const uint32_t *ps, const uint64_t *id_base, char **text,
                            uint64_t *len, uint64_t *slot_off, uint64_t *slot_len)
{
    if (!g || !g->parsed || !text || !len || !id_base || !slot_off || !slot_len) return DUET_INGEST_INVALID;
    const size_t C = g->cand_pos.size();
    if (C && (!pred || !ps)) return DUET_INGEST_INVALID;
    std::string out;
    out.reserve(64 + C * 48);
    const int rc = emit_rows(g, pred, ps, id_base, out, slot_off, slot_len);
    if (rc) return rc;
    return hand_over(out, text, len);
}

int duet_ingest_set_owned(duet_ingest *g, const uint8_t *owned)
{
    if (!g || g->parsed) return DUET_INGEST_INVALID;
    g->skip.clear();
    if (owned) {
        g->skip.resize(g->contigs.size());
        for (size_t k = 0; k < g->contigs.size(); ++k) g->skip[k] = owned[k] ? 0 : 1;
    }
    return DUET_INGEST_OK;
}

int duet_ingest_vcf_precount(duet_ingest *g, const char *path, uint64_t *n_records, uint64_t *n_bytes)
{
    if (!g || !path || !n_records || !n_bytes) return DUET_INGEST_INVALID;
    if (g->vcf_path != path || g->vcf.empty()) {
        if (!read_file_parallel(path, g->vcf, g->threads)) { g->err = std::string("cannot read ") + path; return DUET_INGEST_IO; }
        g->vcf_path = path;
    }
    const char *d = g->vcf.data();
    const size_t n = g->vcf.size();
    const size_t K = g->contigs.size();
    for (size_t k = 0; k < K; ++k) { n_records[k] = 0; n_bytes[k] = 0; }
    const bool has_cr = n && memchr(d, '\r', n);
    std::string key;
    size_t p = 0;
    while (p < n) {
        size_t e;
        if (!has_cr) {
            const char *q = (const char *)memchr(d + p, '\n', n - p);
            e = q ? (size_t)(q - d) : n;
        } else {
            e = p;
            while (e < n && d[e] != '\n' && d[e] != '\r') ++e;
        }
        size_t i = p;
        while (i < e && is_py_space((unsigned char)d[i])) ++i;
        size_t j = i;
        while (j < e && !is_py_space((unsigned char)d[j])) ++j;
        if (j > i && d[i] != '#') {
            key.assign(d + i, j - i);
            auto it = g->owner.find(key);
            if (it != g->owner.end()) { ++n_records[it->second]; n_bytes[it->second] += e - p; }
        }
        p = e + 1;
        if (has_cr && e < n && d[e] == '\r' && p < n && d[p] == '\n') ++p;
    }
    return DUET_INGEST_OK;
}

int duet_ingest_header(duet_ingest *g, int include_all_ctgs, char **text, uint64_t *len)
{
    if (!g || !g->parsed || !text || !len) return DUET_INGEST_INVALID;
    std::string out;
    const int rc = build_header(g, include_all_ctgs, out);
    if (rc) return rc;
    char *buf = (char *)malloc(out.size() + 1);
    if (!buf) return DUET_INGEST_INVALID;
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *text = buf;
    *len = out.size();
    return DUET_INGEST_OK;
}

int duet_ingest_get_rows(duet_ingest *g, duet_ingest_rows *o)
{
    if (!g || !g->parsed || !o) return DUET_INGEST_INVALID;
    const size_t C = g->cand_pos.size();
    if (!g->rows_ready) {
        const bool timing = getenv("DUET_INGEST_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[duet_ingest] rows %-9s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        // (round 6: the three passes over the candidates' spans -- lengths, offsets, copies -- by `threads` workers each; at 2e6
        // candidates they were 120 ms of a 640 ms run on one thread, the pool's zero-fill included)
        const int T = (int)std::min<size_t>((size_t)std::max(1, g->threads), std::max<size_t>(1, C / 4096));
        auto run = [&](auto work) {
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto &th : pool) th.join();
        };
        std::vector<size_t> part((size_t)T + 1, 0);
        std::vector<uint32_t> part_mp((size_t)T, 0);
        run([&](int t) {
            const size_t lo = C * (size_t)t / T, hi = C * (size_t)(t + 1) / T;
            size_t sum = 0;
            uint32_t mp = 0;
            for (size_t c = lo; c < hi; ++c) {
                sum += g->c_chrom[c].n + g->c_ref[c].n + g->c_alt[c].n + g->c_type[c].n;
                mp = std::max(mp, g->cand_pos[c]);
            }
            part[(size_t)t + 1] = sum;
            part_mp[(size_t)t] = mp;
        });
        for (int t = 0; t < T; ++t) part[(size_t)t + 1] += part[(size_t)t];
        lap("lengths");
        const size_t total = part[(size_t)T];
        if (total >= 0xFFFFFFF0ull) return unsupported(g, "candidate texts exceed 4 GiB");
        g->pool.clear();
        g->pool.shrink_to_fit();
        g->pool_raw.reset((char *)malloc(total ? total : 1));          // (not a vector: its resize would write every page from one thread first)
        if (!g->pool_raw) return unsupported(g, "out of memory for the candidate texts");
        g->str_off.resize(4 * C + 1);
        uint32_t mp = 0;
        for (int t = 0; t < T; ++t) mp = std::max(mp, part_mp[(size_t)t]);
        size_t at = total;
        run([&](int t) {
            const size_t lo = C * (size_t)t / T, hi = C * (size_t)(t + 1) / T;
            size_t a = part[(size_t)t];
            char *pool = g->pool_raw.get();
            for (size_t c = lo; c < hi; ++c) {
                // (the texts lie where the caller VCF's lines lie: four cache misses per candidate unless they are asked for ahead)
                if (c + 16 < hi) {
                    __builtin_prefetch(g->c_chrom[c + 16].p);
                    __builtin_prefetch(g->c_ref[c + 16].p);
                    __builtin_prefetch(g->c_alt[c + 16].p);
                    __builtin_prefetch(g->c_type[c + 16].p);
                }
                const Span *f[4] = {&g->c_chrom[c], &g->c_ref[c], &g->c_alt[c], &g->c_type[c]};
                for (int i = 0; i < 4; ++i) {
                    g->str_off[4 * c + i] = (uint32_t)a;
                    if (f[i]->n) memcpy(pool + a, f[i]->p, f[i]->n);
                    a += f[i]->n;
                }
            }
        });
        lap("copies");
        g->str_off[4 * C] = (uint32_t)at;
        g->max_pos = mp;
        // rank of each CHROM text among the distinct ones, in byte order (what Python's string compare does at :229).
        // Neighbouring candidates nearly always share the text: compare with the previous one before hashing.
        auto same = [](const Span &x, const Span &y) { return x.n == y.n && memcmp(x.p, y.p, x.n) == 0; };
        std::vector<std::string> texts;
        {
            // (by the workers: every candidate's CHROM text is a cache miss in the caller VCF's buffer -- one thread walking two
            // million of them twice was 100 ms of a 520 ms run)
            std::vector<std::vector<std::string>> local((size_t)T);
            run([&](int t) {
                const size_t lo = C * (size_t)t / T, hi = C * (size_t)(t + 1) / T;
                std::unordered_map<std::string, int> seen;
                for (size_t c = lo; c < hi; ++c) {
                    if (c + 16 < hi) __builtin_prefetch(g->c_chrom[c + 16].p);
                    if (c > lo && same(g->c_chrom[c], g->c_chrom[c - 1])) continue;
                    std::string x(g->c_chrom[c].p, g->c_chrom[c].n);
                    if (seen.emplace(x, 1).second) local[(size_t)t].push_back(std::move(x));
                }
            });
            std::unordered_map<std::string, int> seen;
            for (auto &v : local)
                for (auto &x : v)
                    if (seen.emplace(x, 1).second) texts.push_back(std::move(x));
        }
        std::sort(texts.begin(), texts.end(), [](const std::string &x, const std::string &y) {
            const int c = memcmp(x.data(), y.data(), std::min(x.size(), y.size()));
            return c ? c < 0 : x.size() < y.size();
        });
        if (texts.size() > 65536) return unsupported(g, "more than 65536 distinct CHROM texts");
        std::unordered_map<std::string, uint16_t> rank;
        for (size_t i = 0; i < texts.size(); ++i) rank.emplace(texts[i], (uint16_t)i);
        g->chrom_rank.resize(C);
        run([&](int t) {
            const size_t lo = C * (size_t)t / T, hi = C * (size_t)(t + 1) / T;
            for (size_t c = lo; c < hi; ++c) {
                if (c + 16 < hi) __builtin_prefetch(g->c_chrom[c + 16].p);
                g->chrom_rank[c] = (c > lo && same(g->c_chrom[c], g->c_chrom[c - 1])) ? g->chrom_rank[c - 1]
                                                                                      : rank.find(std::string(g->c_chrom[c].p, g->c_chrom[c].n))->second;
            }
        });
        g->n_chrom_texts = (uint32_t)texts.size();
        g->rows_ready = true;
        lap("ranks");
    }
    o->n_cands = (uint32_t)C;
    o->pool = g->pool_raw ? g->pool_raw.get() : g->pool.data();
    o->pool_bytes = g->str_off.empty() ? 0 : g->str_off[4 * C];
    o->str_off = g->str_off.data();
    o->cand_chrom_rank = g->chrom_rank.data();
    o->n_chrom_texts = g->n_chrom_texts;
    o->max_pos = g->max_pos;
    o->cand_plus = g->cand_plus.data();
    return DUET_INGEST_OK;
}

int duet_ingest_bam_has_alignments(const duet_ingest *g, int contig)
{
    if (!g || contig < 0 || contig >= (int)g->contigs.size()) return DUET_INGEST_INVALID;
    return g->bam_has_aln[contig];
}

int duet_ingest_set_extraction(duet_ingest *g, int enable, uint32_t min_sv_size, uint32_t min_mapq, uint32_t depth_bin)
{
    if (!g || depth_bin == 0) return DUET_INGEST_INVALID;
    if (g->contigs.size() > 65535) return unsupported(g, "more than 65535 contigs");
    g->extract = enable != 0;
    g->min_sv_size = min_sv_size;
    g->min_mapq = min_mapq;
    g->depth_bin = depth_bin;
    return DUET_INGEST_OK;
}

int duet_ingest_get_marks(duet_ingest *g, duet_ingest_marks *o)
{
    if (!g || !o || !g->extract) return DUET_INGEST_INVALID;
    const size_t K = g->contigs.size(), M = g->m_pos.size();
    g->tag_off.assign(K + 1, 0);
    g->tag_flat.clear();
    for (size_t k = 0; k < K; ++k) {
        g->tag_off[k + 1] = g->tag_off[k] + (uint32_t)g->tags[k].size();
        g->tag_flat.insert(g->tag_flat.end(), g->tags[k].begin(), g->tags[k].end());
    }
    g->m_read.resize(M);
    for (size_t i = 0; i < M; ++i)
        g->m_read[i] = g->m_local[i] == kAbsent ? kAbsent : g->tag_off[g->m_contig[i]] + g->m_local[i];
    g->depth.resize(K);
    g->depth_off.assign(K + 1, 0);
    g->depth_flat.clear();
    for (size_t k = 0; k < K; ++k) {
        g->depth_off[k + 1] = g->depth_off[k] + (uint32_t)g->depth[k].size();
        g->depth_flat.insert(g->depth_flat.end(), g->depth[k].begin(), g->depth[k].end());
    }
    o->n_marks = (uint32_t)M;
    o->n_contigs = (uint32_t)K;
    o->n_reads = (uint32_t)g->tag_flat.size();
    o->depth_bin = g->depth_bin;
    o->mark_contig = g->m_contig.data();
    o->mark_type = g->m_type.data();
    o->mark_pos = g->m_pos.data();
    o->mark_span = g->m_span.data();
    o->mark_read = g->m_read.data();
    o->read_tag = g->tag_flat.data();
    o->read_off = g->tag_off.data();
    o->depth = g->depth_flat.data();
    o->depth_off = g->depth_off.data();
    return DUET_INGEST_OK;
}

}  // extern "C"
