// reads_reader.cpp -- FASTA/FASTQ (plain or gzip) -> concatenated bases + offsets, the arguments
// of phi_add_reads.  Own implementation of the record rules of kseq as the reference uses it
// (src/ILP_index.cpp:313-328, src/kseq.h:192-233): a record starts at a line beginning with '>'
// or '@', its name is the first word, sequence lines run until a line starting with '>', '@' or
// '+'; after '+' as many quality characters as bases are skipped.  Also: output naming
// (src/misc.cpp:58-87) and the FASTA writer (src/ILP_index.cpp:1590-1598).
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../../include/phi_host.h"

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
// the whole (possibly gzip-compressed) file in memory
bool slurp(const char *path, std::vector<char> &buf)
{
    if (FILE *fp = fopen(path, "rb")) {
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, fp);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) && fseek(fp, 0, SEEK_END) == 0) {
            const long sz = ftell(fp);
            if (sz >= 0) {
                rewind(fp);
                buf.resize((size_t)sz);
                const size_t n = sz ? fread(buf.data(), 1, (size_t)sz, fp) : 0;
                fclose(fp);
                buf.resize(n);
                return true;
            }
        }
        fclose(fp);
    } else {
        return false;
    }
    gzFile fp = gzopen(path, "r");
    if (!fp) return false;
    gzbuffer(fp, 1 << 20);
    size_t len = 0;
    buf.resize((size_t)1 << 22);
    for (;;) {
        if (buf.size() - len < ((size_t)1 << 20)) buf.resize(buf.size() * 2);
        const int n = gzread(fp, buf.data() + len, (unsigned)std::min<size_t>(buf.size() - len, (size_t)1 << 30));
        if (n <= 0) break;
        len += (size_t)n;
    }
    gzclose(fp);
    buf.resize(len);
    return true;
}

// one line [p, e) of the buffer (without the newline / a trailing CR); returns the start of the next
inline const char *next_line(const char *p, const char *end, const char *&e)
{
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    e = nl ? nl : end;
    const char *nx = nl ? nl + 1 : end;
    if (e > p && e[-1] == '\r') e--;
    return nx;
}
}  // namespace

extern "C" {

int phi_reads_read(const char *path, phi_reads **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    std::vector<char> buf;
    if (!slurp(path, buf)) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path);
    phi_reads *r = new phi_reads();
    r->bases.resize(buf.size());                      // the bases are a subset of the file's bytes
    char *bases = r->bases.data();
    size_t nb = 0;
    const char *p = buf.data(), *const end = buf.data() + buf.size();
    const char *e = p;
    bool have = p < end;
    const char *line = p;
    if (have) p = next_line(p, end, e);
    auto advance = [&]() { have = p < end; line = p; if (have) p = next_line(p, end, e); };
    while (have) {
        if (e == line || (line[0] != '>' && line[0] != '@')) { advance(); continue; }
        const char *b = line + 1;
        while (b < e && !isspace((unsigned char)*b)) b++;
        r->name_off.push_back((int64_t)r->names.size());
        r->names.insert(r->names.end(), line + 1, b);
        r->names.push_back('\0');
        const size_t start = nb;
        advance();
        while (have && (e == line || (line[0] != '>' && line[0] != '@' && line[0] != '+'))) {
            const size_t n = (size_t)(e - line);
            memcpy(bases + nb, line, n);
            unsigned bad = 0;
            for (size_t i = 0; i < n; i++) bad |= (unsigned)((unsigned char)line[i] - 33) > 93u;   // not isgraph
            if (bad) {
                size_t k = nb;
                for (size_t i = 0; i < n; i++) if (isgraph((unsigned char)line[i])) bases[k++] = line[i];
                nb = k;
            } else nb += n;
            advance();
        }
        const size_t len = nb - start;
        r->off.push_back((int64_t)nb);
        if (have && line[0] == '+') {                 // FASTQ: skip the quality block
            size_t q = 0;
            advance();
            while (have && q < len) { q += (size_t)(e - line); advance(); }
        }
    }
    r->bases.resize(nb);
    *out = r;
    return PHI_HOST_OK;
}

// ---- streaming reader: the same record rules as phi_reads_read, one line at a time,
//      so that a reads file of any size goes through fixed buffers the caller owns (pinned, for the
//      device copy) while the chunk before is on the GPU.
struct phi_reads_stream {
    FILE *fp = nullptr;                               // plain file ...
    gzFile gz = nullptr;                              // ... or gzip
    std::vector<char> ibuf;                           // file bytes not yet parsed: [pos, fill)
    size_t pos = 0, fill = 0;
    bool eof = false;
    int state = 0;                                    // 0 between records, 1 in sequence lines, 2 in quality lines
    size_t rec_len = 0, qual = 0;                     // bases of the open record; quality characters skipped so far
    std::vector<char> carry;                          // bases of the open record parsed during an earlier call
    int64_t total_reads = 0, total_bases = 0;
};

namespace {
// more file bytes behind the unparsed tail; false at end of file
bool stream_fill(phi_reads_stream *s)
{
    if (s->eof) return false;
    if (s->pos > 0) {
        memmove(s->ibuf.data(), s->ibuf.data() + s->pos, s->fill - s->pos);
        s->fill -= s->pos;
        s->pos = 0;
    }
    if (s->fill == s->ibuf.size()) s->ibuf.resize(s->ibuf.size() * 2);        // a line longer than the buffer
    const size_t room = s->ibuf.size() - s->fill;
    long n;
    if (s->gz) n = gzread(s->gz, s->ibuf.data() + s->fill, (unsigned)std::min<size_t>(room, (size_t)1 << 30));
    else n = (long)fread(s->ibuf.data() + s->fill, 1, room, s->fp);
    if (n <= 0) { s->eof = true; return false; }
    s->fill += (size_t)n;
    return true;
}

// the next complete line [b, e) (newline and a trailing CR stripped); false when the file is exhausted
bool stream_line(phi_reads_stream *s, const char *&b, const char *&e)
{
    for (;;) {
        const char *base = s->ibuf.data();
        const char *nl = s->fill > s->pos ? (const char *)memchr(base + s->pos, '\n', s->fill - s->pos) : nullptr;
        if (nl) {
            b = base + s->pos; e = nl;
            s->pos = (size_t)(nl - base) + 1;
            if (e > b && e[-1] == '\r') e--;
            return true;
        }
        if (stream_fill(s)) continue;
        if (s->fill > s->pos) {                       // last line without a newline
            b = s->ibuf.data() + s->pos; e = s->ibuf.data() + s->fill;
            s->pos = s->fill;
            if (e > b && e[-1] == '\r') e--;
            return true;
        }
        return false;
    }
}
}  // namespace

int phi_reads_stream_open(const char *path, phi_reads_stream **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path);
    unsigned char magic[2] = {0, 0};
    const size_t got = fread(magic, 1, 2, fp);
    phi_reads_stream *s = new phi_reads_stream();
    s->ibuf.resize((size_t)8 << 20);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        fclose(fp);
        s->gz = gzopen(path, "r");
        if (!s->gz) { delete s; return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path); }
        gzbuffer(s->gz, 1 << 20);
    } else {
        rewind(fp);
        s->fp = fp;
    }
    *out = s;
    return PHI_HOST_OK;
}

int64_t phi_reads_stream_next(phi_reads_stream *s, char *bases, int64_t bases_cap, int64_t *off, int64_t reads_cap,
                              char *err, int err_cap)
{
    if (!s || !bases || !off || bases_cap <= 0 || reads_cap <= 0) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "bad arguments");
    size_t nb = 0, rec_start = 0;
    int64_t n = 0;
    off[0] = 0;
    if (!s->carry.empty()) {                          // the record that did not fit the chunk before
        if ((int64_t)s->carry.size() > bases_cap)
            return fail(err, err_cap, PHI_HOST_ERR_INVALID, "a read of more than %lld bases does not fit a chunk", (long long)bases_cap);
        memcpy(bases, s->carry.data(), s->carry.size());
        nb = s->carry.size();
        s->carry.clear();
    }
    auto finish_record = [&]() {
        off[++n] = (int64_t)nb;
        s->total_reads++;
        s->total_bases += (int64_t)s->rec_len;
        rec_start = nb;
    };
    const char *b, *e;
    for (;;) {
        if (n == reads_cap) return n;                 // only between records (state 0 or 2)
        if (!stream_line(s, b, e)) {
            if (s->state == 1) { s->state = 0; finish_record(); }
            return n;
        }
        if (s->state == 2) {                          // quality block: as many characters as bases (kseq.h:221-230)
            s->qual += (size_t)(e - b);
            if (s->qual >= s->rec_len) s->state = 0;
            continue;
        }
        if (s->state == 1) {
            if (e == b || (b[0] != '>' && b[0] != '@' && b[0] != '+')) {          // a sequence line
                const size_t ln = (size_t)(e - b);
                if (nb + ln > (size_t)bases_cap) {
                    // chunk full inside a record: hand back the finished records, keep this one's bases
                    if (n == 0)
                        return fail(err, err_cap, PHI_HOST_ERR_INVALID, "a read of more than %lld bases does not fit a chunk", (long long)bases_cap);
                    s->carry.assign(bases + rec_start, bases + nb);
                    s->pos = (size_t)(b - s->ibuf.data());       // this line is parsed again by the next call
                    return n;
                }
                char *dst = bases + nb;
                memcpy(dst, b, ln);
                unsigned bad = 0;
                for (size_t i = 0; i < ln; i++) bad |= (unsigned)((unsigned char)b[i] - 33) > 93u;       // not isgraph
                size_t kept = ln;
                if (bad) {
                    kept = 0;
                    for (size_t i = 0; i < ln; i++) if (isgraph((unsigned char)b[i])) dst[kept++] = b[i];
                }
                nb += kept;
                s->rec_len += kept;
                continue;
            }
            s->state = 0;
            finish_record();
            if (b[0] == '+') {
                s->qual = 0;
                s->state = s->rec_len > 0 ? 2 : 0;
                continue;
            }
            // a header line ends the record and opens the next: fall through
        }
        if (e > b && (b[0] == '>' || b[0] == '@')) {
            if (n == reads_cap) { s->pos = (size_t)(b - s->ibuf.data()); return n; }
            s->state = 1;
            s->rec_len = 0;
            rec_start = nb;
        }
    }
}

int64_t phi_reads_stream_reads(const phi_reads_stream *s) { return s ? s->total_reads : 0; }
int64_t phi_reads_stream_bases(const phi_reads_stream *s) { return s ? s->total_bases : 0; }

void phi_reads_stream_close(phi_reads_stream *s)
{
    if (!s) return;
    if (s->fp) fclose(s->fp);
    if (s->gz) gzclose(s->gz);
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
    for (int64_t i = 0; i < len; i += 80) {
        const int64_t n = len - i < 80 ? len - i : 80;
        fwrite(seq + i, 1, (size_t)n, fp);
        fputc('\n', fp);
    }
    return fclose(fp) == 0 ? PHI_HOST_OK : PHI_HOST_ERR_IO;
}

}  // extern "C"
