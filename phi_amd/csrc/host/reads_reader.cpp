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
#include <string>
#include <vector>
#include "../../../include/phi_host.h"

struct phi_reads {
    std::string bases;
    std::vector<int64_t> off{0};
    std::vector<std::string> names;
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
class Lines {
public:
    explicit Lines(const char *path) { fp_ = gzopen(path, "r"); if (fp_) gzbuffer(fp_, 1 << 20); }
    ~Lines() { if (fp_) gzclose(fp_); }
    bool ok() const { return fp_ != nullptr; }
    bool next(std::string &line)
    {
        line.clear();
        bool got = false;
        for (;;) {
            if (pos_ == len_) {
                len_ = gzread(fp_, buf_, sizeof buf_);
                pos_ = 0;
                if (len_ <= 0) { len_ = 0; break; }
            }
            got = true;
            const char *nl = (const char *)memchr(buf_ + pos_, '\n', (size_t)(len_ - pos_));
            if (nl) {
                line.append(buf_ + pos_, (size_t)(nl - (buf_ + pos_)));
                pos_ = (int)(nl - buf_) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(buf_ + pos_, (size_t)(len_ - pos_));
            pos_ = len_;
        }
        if (!got) return false;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
private:
    gzFile fp_ = nullptr;
    char buf_[1 << 16];
    int pos_ = 0, len_ = 0;
};
}  // namespace

extern "C" {

int phi_reads_read(const char *path, phi_reads **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    Lines in(path);
    if (!in.ok()) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to open the reads file %s", path);
    phi_reads *r = new phi_reads();
    std::string line;
    bool have = in.next(line);
    while (have) {
        if (line.empty() || (line[0] != '>' && line[0] != '@')) { have = in.next(line); continue; }
        size_t a = 1, b = 1;
        while (b < line.size() && !isspace((unsigned char)line[b])) b++;
        r->names.emplace_back(line, a, b - a);
        const size_t start = r->bases.size();
        have = in.next(line);
        while (have && (line.empty() || (line[0] != '>' && line[0] != '@' && line[0] != '+'))) {
            for (char ch : line) if (isgraph((unsigned char)ch)) r->bases.push_back(ch);
            have = in.next(line);
        }
        const size_t len = r->bases.size() - start;
        r->off.push_back((int64_t)r->bases.size());
        if (have && line[0] == '+') {                 // FASTQ: skip the quality block
            size_t q = 0;
            have = in.next(line);
            while (have && q < len) { q += line.size(); have = in.next(line); }
        }
    }
    *out = r;
    return PHI_HOST_OK;
}

void phi_reads_free(phi_reads *r) { delete r; }
int64_t phi_reads_count(const phi_reads *r) { return (int64_t)r->names.size(); }
const char *phi_reads_bases(const phi_reads *r) { return r->bases.data(); }
const int64_t *phi_reads_off(const phi_reads *r) { return r->off.data(); }
const char *phi_reads_name(const phi_reads *r, int64_t i)
{
    return (i >= 0 && i < (int64_t)r->names.size()) ? r->names[(size_t)i].c_str() : "";
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
