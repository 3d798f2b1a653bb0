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
