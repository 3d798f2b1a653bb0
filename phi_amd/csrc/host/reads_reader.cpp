// reads_reader.cpp -- FASTA/FASTQ (plain or gzip) -> concatenated bases + offsets, the arguments
// of phi_add_reads.  Own implementation of kseq as the reference uses it (src/ILP_index.cpp:313-328,
// src/kseq.h:192-233), down to its behaviour on malformed input: the next header is the next '>' or
// '@' character wherever it stands, sequence lines are taken as they are and end at a line starting
// with '>', '@' or '+', whole quality lines are read until they cover the sequence, and a quality
// string of another length (or none) ends the reading of the file, as the reference's loop does.  Also: output naming
// (src/misc.cpp:58-87) and the FASTA writer (src/ILP_index.cpp:1590-1598).
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <zlib.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/phi_host.h"
#include "gz_source.h"

struct phi_reads {
    std::vector<char> bases;
    std::vector<int64_t> off{0};
    std::vector<char> names;                          // NUL-terminated names back to back
    std::vector<int64_t> name_off;
};

static int fail(char *err, int cap, int code, const char *fmt, ...)
{
    if (err && cap > 0) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, (size_t)cap, fmt, ap);
        va_end(ap);
    }
    return code;
}

namespace {
// A byte source with kstream's two primitives (kseq.h:100-150): one character, or the rest of the line.
struct ByteSrc {
    FILE *fp = nullptr;                               // plain file ...
    GzSource *gz = nullptr;                           // ... or gzip: inflated off this thread (gz_source.h), on many threads for BGZF
    // ... or text that is already in memory, followed by blocks a callback hands over (the rest of a stream whose
    // beginning the device has taken: phi_reads_stream_open_blocks)
    const char *mem = nullptr;
    size_t mem_n = 0, mem_at = 0;
    phi_text_block_fn next_block = nullptr;
    void *next_user = nullptr;
    bool from_blocks = false;
    bool failed = false;                              // the source broke (corrupt gzip stream, callback error): not a clean end
    std::vector<char> buf;
    size_t begin = 0, end = 0;
    bool is_eof = false;                              // the source has no more bytes (nothing to do with kseq's flag of that name)
    // kseq reads its stream in blocks of 65 536 bytes (kseq.h:242) and flags the end when a block comes back short
    // (kseq.h:81,113): at the moment every byte is consumed its flag is already up -- unless the stream's length is a
    // multiple of the block size, when only a further read, of 0 bytes, raises it.  What the last bytes of a file mean (a
    // bare header character, a lone CR) depends on that flag, so it is kept here as kseq has it: `total` counts the bytes
    // of the stream from its very first one, zero_read says that kseq would have made its read of 0 bytes.
    static const uint64_t KSEQ_BLOCK = 65536;
    uint64_t total = 0;
    bool zero_read = false;
    bool exhausted() { return begin >= end && !refill(); }
    bool kseq_eof() { return exhausted() && (zero_read || total % KSEQ_BLOCK != 0); }

    static int inflate_threads()
    {
        const char *e = getenv("PHI_HOST_THREADS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }
    bool open(const char *path)
    {
        FILE *f = fopen(path, "rb");
        if (!f) return false;
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, f);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            fclose(f);
            gz = new GzSource();
            if (!gz->open(path, inflate_threads())) { delete gz; gz = nullptr; return false; }
        } else {
            buf.resize((size_t)4 << 20);
            rewind(f);
            fp = f;
        }
        return true;
    }
    void close()
    {
        if (fp) fclose(fp);
        if (gz) { gz->close(); delete gz; }
        fp = nullptr; gz = nullptr;
    }
    void open_blocks(const char *prefix, size_t n_prefix, phi_text_block_fn fn, void *user, uint64_t stream_offset)
    {
        mem = prefix; mem_n = n_prefix; mem_at = 0; next_block = fn; next_user = user; from_blocks = true;
        total = stream_offset;
    }
    bool refill()                                     // false: nothing more to read
    {
        if (is_eof) return false;
        begin = 0; end = 0;
        if (from_blocks) {
            while (end == 0) {
                if (mem_at < mem_n) {
                    const size_t n = std::min(mem_n - mem_at, (size_t)4 << 20);
                    buf.assign(mem + mem_at, mem + mem_at + n);
                    mem_at += n;
                } else {
                    const char *p = nullptr;
                    const int64_t n = next_block ? next_block(next_user, &p) : 0;
                    if (n <= 0) { is_eof = true; failed = n < 0; return false; }
                    buf.assign(p, p + n);
                }
                end = buf.size();
            }
            total += end;
            return true;
        }
        if (gz) {
            while (end == 0) {
                if (!gz->next(buf)) { is_eof = true; failed = !gz->ok(); return false; }   // a corrupt stream is an error, not the end of the file
                end = buf.size();
            }
            total += end;
            return true;
        }
        const long n = (long)fread(buf.data(), 1, buf.size(), fp);
        end = n > 0 ? (size_t)n : 0;
        if (end < buf.size()) is_eof = true;
        total += end;
        return end > 0;
    }
    int getc()
    {
        if (begin >= end && !refill()) { zero_read = true; return -1; }   // (kseq.h:78-83)
        return (unsigned char)buf[begin++];
    }
    // the bytes up to the next '\n' (consumed, not stored) appended to out; -1 when the source was already
    // exhausted.  As ks_getuntil2(KS_SEP_LINE): a trailing '\r' is dropped when the string is longer than 1.
    long line(std::vector<char> &out)
    {
        if (kseq_eof()) return -1;
        for (;;) {
            if (begin >= end && !refill()) { zero_read = true; break; }
            const char *p = buf.data() + begin;
            const char *nl = (const char *)memchr(p, '\n', end - begin);
            const size_t n = nl ? (size_t)(nl - p) : end - begin;
            out.insert(out.end(), p, p + n);
            begin += n + (nl ? 1 : 0);
            if (nl) break;
        }
        if (out.size() > 1 && out.back() == '\r') out.pop_back();
        return (long)out.size();
    }
    // the rest of the line, dropped: false when the source ended before a '\n'
    bool skip_line()
    {
        for (;;) {
            if (begin >= end && !refill()) { zero_read = true; return false; }
            const char *p = buf.data() + begin;
            const char *nl = (const char *)memchr(p, '\n', end - begin);
            if (nl) { begin += (size_t)(nl - p) + 1; return true; }
            begin = end;
        }
    }
    // the same for a string whose bytes are not kept (the quality): its length and the number of carriage returns it
    // ends with stand for it -- kseq strips ONE trailing CR from the ACCUMULATED string after every line it appends
    // (kseq.h:146), so "..~\r\r" followed by an empty line loses both
    long line_len(size_t &len, size_t &trail_cr)
    {
        if (kseq_eof()) return -1;
        for (;;) {
            if (begin >= end && !refill()) { zero_read = true; break; }
            const char *p = buf.data() + begin;
            const char *nl = (const char *)memchr(p, '\n', end - begin);
            const size_t n = nl ? (size_t)(nl - p) : end - begin;
            if (n) {
                size_t t = 0;
                while (t < n && p[n - 1 - t] == '\r') t++;
                trail_cr = t == n ? trail_cr + n : t;
                len += n;
            }
            begin += n + (nl ? 1 : 0);
            if (nl) break;
        }
        if (len > 1 && trail_cr > 0) { len--; trail_cr--; }
        return (long)len;
    }
    // the bytes up to the next white space (ks_getuntil with KS_SEP_SPACE); *dret = the delimiter
    long word(std::string &out, int *dret)
    {
        out.clear();
        *dret = 0;
        if (kseq_eof()) return -1;
        for (;;) {
            if (begin >= end && !refill()) { zero_read = true; break; }
            size_t i = begin;
            while (i < end && !isspace((unsigned char)buf[i])) i++;
            out.append(buf.data() + begin, i - begin);
            const bool hit = i < end;
            if (hit) *dret = (unsigned char)buf[i];
            begin = i + 1;
            if (hit) break;
        }
        return (long)out.size();
    }
};

// kseq_read (kseq.h:192-233) as the reference instantiates it (ILP_index.cpp:8): >= 0 length of the
// sequence, -1 end of file, -2 quality string missing or of another length (the caller stops reading:
// ILP_index.cpp:322).  Sequence lines are taken as they are (no filtering), empty lines skipped.
struct KseqState {
    bool want_names = true;                           // the streaming reader has no use for names: the header line is skipped whole
    int last_char = 0;
    std::string name, comment_sink;
    std::vector<char> seq, sink;
};
long kseq_next(ByteSrc &ks, KseqState &st)
{
    int c;
    if (st.last_char == 0) {                          // jump to the next header character, wherever it is
        while ((c = ks.getc()) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        st.last_char = c;
    }
    st.seq.clear();
    if (st.want_names) {
        if (ks.word(st.name, &c) < 0) return -1;
        if (c != '\n') { st.sink.clear(); ks.line(st.sink); }     // the comment
    } else {
        // name and comment together are the header line (a name that ends with the file ends the record list the
        // same way: ks.word returns the name, the sequence loop below finds nothing)
        if (ks.kseq_eof()) return -1;
        (void)ks.skip_line();
    }
    while ((c = ks.getc()) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;                      // empty line
        st.seq.push_back((char)c);
        ks.line(st.seq);                              // the rest of the line
    }
    if (c == '>' || c == '@') st.last_char = c;
    if (c != '+') return (long)st.seq.size();         // FASTA
    if (!ks.skip_line()) return -2;                   // the rest of the '+' line
    size_t ql = 0, q_cr = 0;
    while (ks.line_len(ql, q_cr) >= 0 && ql < st.seq.size()) {}
    st.last_char = 0;
    if (ql != st.seq.size()) return -2;
    return (long)st.seq.size();
}
}  // namespace

extern "C" {

int phi_reads_read(const char *path, phi_reads **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    ByteSrc ks;
    if (!ks.open(path)) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path);
    phi_reads *r = new phi_reads();
    KseqState st;
    while (kseq_next(ks, st) >= 0) {                  // (:322: any negative value ends the loop)
        r->name_off.push_back((int64_t)r->names.size());
        r->names.insert(r->names.end(), st.name.begin(), st.name.end());
        r->names.push_back('\0');
        r->bases.insert(r->bases.end(), st.seq.begin(), st.seq.end());
        r->off.push_back((int64_t)r->bases.size());
    }
    const bool broke = ks.failed;
    ks.close();
    if (broke) { delete r; return fail(err, err_cap, PHI_HOST_ERR_IO, "gzip stream corrupt in the reads file %s", path); }
    *out = r;
    return PHI_HOST_OK;
}

// ---- streaming reader: the same records, a chunk at a time into buffers the caller owns
//      (pinned, for the device copy) while the chunk before is on the GPU.
struct phi_reads_stream {
    ByteSrc ks;
    KseqState st;
    bool pending = false, done = false;               // st.seq holds a record that did not fit the chunk before
    int64_t total_reads = 0, total_bases = 0;
};

int phi_reads_stream_open(const char *path, phi_reads_stream **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    phi_reads_stream *s = new phi_reads_stream();
    s->st.want_names = false;
    if (!s->ks.open(path)) { delete s; return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path); }
    *out = s;
    return PHI_HOST_OK;
}

int phi_reads_stream_open_blocks(const char *prefix, int64_t n_prefix, phi_text_block_fn next, void *user, int64_t stream_offset,
                                 phi_reads_stream **out, char *err, int err_cap)
{
    if (!out || n_prefix < 0 || (n_prefix > 0 && !prefix) || stream_offset < 0) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "bad arguments");
    phi_reads_stream *s = new phi_reads_stream();
    s->st.want_names = false;
    s->ks.open_blocks(prefix, (size_t)n_prefix, next, user, (uint64_t)stream_offset);
    *out = s;
    return PHI_HOST_OK;
}

int64_t phi_reads_stream_next(phi_reads_stream *s, char *bases, int64_t bases_cap, int64_t *off, int64_t reads_cap,
                              char *err, int err_cap)
{
    if (!s || !bases || !off || bases_cap <= 0 || reads_cap <= 0) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "bad arguments");
    int64_t n = 0, nb = 0;
    off[0] = 0;
    while (n < reads_cap && !s->done) {
        if (!s->pending) {
            if (kseq_next(s->ks, s->st) < 0) {
                s->done = true;
                if (s->ks.failed) return fail(err, err_cap, PHI_HOST_ERR_IO, "the reads text broke off: gzip stream corrupt, or the source failed");
                break;
            }
            s->pending = true;
        }
        const int64_t len = (int64_t)s->st.seq.size();
        if (len > bases_cap)
            return fail(err, err_cap, PHI_HOST_ERR_INVALID, "a read of more than %lld bases does not fit a chunk", (long long)bases_cap);
        if (nb + len > bases_cap) break;              // goes into the next chunk
        if (len) memcpy(bases + nb, s->st.seq.data(), (size_t)len);
        nb += len;
        off[++n] = nb;
        s->pending = false;
        s->total_reads++;
        s->total_bases += len;
    }
    return n;
}

int64_t phi_reads_stream_reads(const phi_reads_stream *s) { return s ? s->total_reads : 0; }
int64_t phi_reads_stream_bases(const phi_reads_stream *s) { return s ? s->total_bases : 0; }

void phi_reads_stream_close(phi_reads_stream *s)
{
    if (!s) return;
    s->ks.close();
    delete s;
}

// ---- the (inflated) text of a reads file as it is, for the device-side record splitter (phi_add_reads_text): no parsing
//      here at all.  Plain files are read with several preads at once straight into the caller's (pinned) buffer.
struct phi_text_stream {
    int fd = -1;
    int64_t size = 0, pos = 0;
    GzSource *gz = nullptr;
    std::vector<char> blk;
    size_t blk_at = 0;
    bool eof = false;
};

int phi_text_stream_open(const char *path, phi_text_stream **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path);
    unsigned char magic[2] = {0, 0};
    const ssize_t got = pread(fd, magic, 2, 0);
    phi_text_stream *s = new phi_text_stream();
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        ::close(fd);
        s->gz = new GzSource();
        if (!s->gz->open(path, ByteSrc::inflate_threads())) { delete s->gz; delete s; return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path); }
    } else {
        struct stat st;
        s->fd = fd;
        s->size = (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) ? (int64_t)st.st_size : -1;   // -1: a pipe, read sequentially
    }
    *out = s;
    return PHI_HOST_OK;
}

int64_t phi_text_stream_read(phi_text_stream *s, char *buf, int64_t cap, char *err, int err_cap)
{
    if (!s || !buf || cap <= 0) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "bad arguments");
    if (s->eof) return 0;
    int64_t n = 0;
    if (s->gz) {
        while (n < cap) {
            if (s->blk_at == s->blk.size()) {
                s->blk_at = 0;
                if (!s->gz->next(s->blk)) {
                    s->blk.clear();
                    s->eof = true;
                    if (!s->gz->ok()) return fail(err, err_cap, PHI_HOST_ERR_IO, "gzip stream corrupt in the reads file");
                    break;
                }
            }
            const size_t take = std::min<size_t>((size_t)(cap - n), s->blk.size() - s->blk_at);
            memcpy(buf + n, s->blk.data() + s->blk_at, take);
            s->blk_at += take; n += (int64_t)take;
        }
        return n;
    }
    if (s->size < 0) {                                  // not a regular file
        while (n < cap) {
            const ssize_t r = ::read(s->fd, buf + n, (size_t)(cap - n));
            if (r < 0) return fail(err, err_cap, PHI_HOST_ERR_IO, "read error on the reads file");
            if (r == 0) { s->eof = true; break; }
            n += r;
        }
        return n;
    }
    const int64_t want = std::min<int64_t>(cap, s->size - s->pos);
    if (want <= 0) { s->eof = true; return 0; }
    // a page-cache read is a memcpy in the kernel, ~3 GB/s on one thread: several at once
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(ByteSrc::inflate_threads(), 8), want / ((int64_t)4 << 20)));
    std::atomic<int> bad{0};
    auto part = [&](int t) {
        int64_t lo = want * t / nt, hi = want * (t + 1) / nt;
        while (lo < hi) {
            const ssize_t r = pread(s->fd, buf + lo, (size_t)(hi - lo), (off_t)(s->pos + lo));
            if (r <= 0) { bad.store(1); return; }          // (a file that shrank under us)
            lo += r;
        }
    };
    if (nt == 1) part(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(part, t);
        part(0);
        for (auto &t : th) t.join();
    }
    if (bad.load()) return fail(err, err_cap, PHI_HOST_ERR_IO, "read error on the reads file");
    s->pos += want;
    if (s->pos >= s->size) s->eof = true;
    return want;
}

void phi_text_stream_close(phi_text_stream *s)
{
    if (!s) return;
    if (s->fd >= 0) ::close(s->fd);
    if (s->gz) { s->gz->close(); delete s->gz; }
    delete s;
}

void phi_reads_free(phi_reads *r) { delete r; }
int64_t phi_reads_count(const phi_reads *r) { return (int64_t)r->name_off.size(); }
const char *phi_reads_bases(const phi_reads *r) { return r->bases.data(); }
const int64_t *phi_reads_off(const phi_reads *r) { return r->off.data(); }
const char *phi_reads_name(const phi_reads *r, int64_t i)
{
    return (i >= 0 && i < (int64_t)r->name_off.size()) ? r->names.data() + r->name_off[(size_t)i] : "";
}

int phi_hap_name(const char *gfa_path, const char *reads_path, char *out, int cap)
{
    if (!gfa_path || !reads_path || !out) return -1;
    auto base = [](const std::string &p) {
        const size_t i = p.find_last_of("/\\");
        return i == std::string::npos ? p : p.substr(i + 1);
    };
    std::string name = base(gfa_path);
    size_t dot = name.find_last_of('.');
    if (dot != std::string::npos) name.resize(dot);
    name += "_";
    name += base(reads_path);
    dot = name.find_last_of('.');
    if (dot != std::string::npos) name.resize(dot);
    if ((int)name.size() + 1 > cap) return -1;
    memcpy(out, name.c_str(), name.size() + 1);
    return (int)name.size();
}

int phi_write_fasta(const char *path, const char *name, const char *seq, int64_t len)
{
    if (!path || !name || (len > 0 && !seq)) return PHI_HOST_ERR_INVALID;
    FILE *fp = fopen(path, "w");
    if (!fp) return PHI_HOST_ERR_IO;
    fprintf(fp, ">%s LN:%lld\n", name, (long long)len);
    // 80-column lines, laid out in blocks of 64 K lines (5 MB) and written whole: a chromosome is 180 MB of them.  (Round 4 tried
    // all threads laying the lines out into a shared mapping of the file, and into blocks written with pwrite at their offsets:
    // on tmpfs both were SLOWER than this loop, 0.19 / 0.17 against 0.15 s for 176 MB -- what costs is the file system giving
    // the file its pages, one at a time under the file's lock.)
    const int64_t LINES = 1 << 16;
    std::vector<char> blk((size_t)(LINES * 81));
    bool ok = true;
    for (int64_t i = 0; i < len && ok; i += LINES * 80) {
        char *o = blk.data();
        for (int64_t j = i; j < len && j < i + LINES * 80; j += 80) {
            const int64_t n = len - j < 80 ? len - j : 80;
            memcpy(o, seq + j, (size_t)n);
            o[n] = '\n';
            o += n + 1;
        }
        ok = fwrite(blk.data(), 1, (size_t)(o - blk.data()), fp) == (size_t)(o - blk.data());
    }
    return (fclose(fp) == 0 && ok) ? PHI_HOST_OK : PHI_HOST_ERR_IO;
}

}  // extern "C"
