// host_selftest.cpp -- a driver of the host-side readers for the sanitizer builds (`make sanitize`: AddressSanitizer +
// UndefinedBehaviorSanitizer, and ThreadSanitizer; CPU only -- the GPU pool offers no sanitizer).  Test infrastructure.
//
//   host_selftest gfa  FILE...     phi_gfa_read of every file; prints a checksum of the arrays per file (or "error CODE")
//   host_selftest reads FILE...    every file through the three reads readers -- phi_reads_read, phi_reads_stream_* in small
//                                  chunks, phi_text_stream_* + phi_reads_stream_open_blocks -- which must agree; checksum per file
// The same checksums come out of the plain library (tests/test_cpu_sanitizers.py compares them).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../../include/phi_host.h"

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

static int do_gfa(const char *path)
{
    phi_graph *g = nullptr;
    char err[512] = "";
    const int rc = phi_gfa_read(path, &g, err, sizeof err);
    if (rc) { printf("%s error %d\n", path, rc); return 0; }
    uint64_t h = 0xcbf29ce484222325ull;
    const int32_t nv = phi_graph_n_vtx(g), nw = phi_graph_n_walks(g);
    h = fnv(h, &nv, 4); h = fnv(h, &nw, 4);
    h = fnv(h, phi_graph_seq_off(g), (size_t)(nv + 1) * 8);
    h = fnv(h, phi_graph_seq_concat(g), (size_t)phi_graph_seq_off(g)[nv]);
    h = fnv(h, phi_graph_adj_off(g), (size_t)(nv + 1) * 8);
    h = fnv(h, phi_graph_adj(g), (size_t)phi_graph_n_edges(g) * 4);
    h = fnv(h, phi_graph_walk_off(g), (size_t)(nw + 1) * 8);
    h = fnv(h, phi_graph_walk_vtx(g), (size_t)phi_graph_walk_off(g)[nw] * 4);
    h = fnv(h, phi_graph_topo_rank(g), (size_t)nv * 4);
    for (int32_t v = 0; v < nv; v++) { const char *s = phi_graph_seg_name(g, v); h = fnv(h, s, strlen(s) + 1); }
    for (int32_t w = 0; w < nw; w++) { const char *s = phi_graph_hap_name(g, w); h = fnv(h, s, strlen(s) + 1); }
    printf("%s %016llx\n", path, (unsigned long long)h);
    phi_graph_free(g);
    return 0;
}

struct Blocks { phi_text_stream *ts; std::vector<char> buf; char err[256]; };
static int64_t next_block(void *user, const char **block)
{
    Blocks *b = (Blocks *)user;
    const int64_t n = phi_text_stream_read(b->ts, b->buf.data(), (int64_t)b->buf.size(), b->err, sizeof b->err);
    *block = b->buf.data();
    return n;
}

static int do_reads(const char *path)
{
    char err[512] = "";
    phi_reads *r = nullptr;
    int rc = phi_reads_read(path, &r, err, sizeof err);
    if (rc) {
        // the other readers must fail on it too
        phi_reads_stream *s = nullptr;
        std::vector<char> b(1 << 12); std::vector<int64_t> o(65);
        int64_t n = phi_reads_stream_open(path, &s, err, sizeof err) == 0 ? 1 : -1;
        while (n > 0) n = phi_reads_stream_next(s, b.data(), (int64_t)b.size(), o.data(), 64, err, sizeof err);
        phi_reads_stream_close(s);
        printf("%s error %d %s\n", path, rc, n < 0 ? "(stream too)" : "(STREAM ACCEPTED IT)");
        return n < 0 ? 0 : 1;
    }
    const int64_t nr = phi_reads_count(r);
    const int64_t *off = phi_reads_off(r);
    std::string all(phi_reads_bases(r), (size_t)off[nr]);
    std::vector<int64_t> lens;
    for (int64_t i = 0; i < nr; i++) lens.push_back(off[i + 1] - off[i]);
    uint64_t h = fnv(fnv(0xcbf29ce484222325ull, all.data(), all.size()), lens.data(), lens.size() * 8);
    phi_reads_free(r);
    // the streaming reader in chunks of 4 kbases / 64 reads (a longer read: chunks as long as it)
    for (int variant = 0; variant < 2; variant++) {
        phi_reads_stream *s = nullptr;
        Blocks bl{nullptr, std::vector<char>(777), ""};
        if (variant == 0) rc = phi_reads_stream_open(path, &s, err, sizeof err);
        else {
            rc = phi_text_stream_open(path, &bl.ts, err, sizeof err);
            if (!rc) rc = phi_reads_stream_open_blocks(nullptr, 0, next_block, &bl, 0, &s, err, sizeof err);
        }
        if (rc) { printf("%s: reader %d cannot open: %s\n", path, variant, err); return 1; }
        size_t cap = 4096;
        for (int64_t l : lens) if ((size_t)l > cap) cap = (size_t)l;
        std::vector<char> b(cap); std::vector<int64_t> o(65);
        std::string got; std::vector<int64_t> glens;
        for (;;) {
            const int64_t n = phi_reads_stream_next(s, b.data(), (int64_t)b.size(), o.data(), 64, err, sizeof err);
            if (n < 0) { printf("%s: reader %d failed: %s\n", path, variant, err); return 1; }
            if (n == 0) break;
            got.append(b.data(), (size_t)o[n]);
            for (int64_t i = 0; i < n; i++) glens.push_back(o[i + 1] - o[i]);
        }
        phi_reads_stream_close(s);
        if (bl.ts) phi_text_stream_close(bl.ts);
        if (got != all || glens != lens) { printf("%s: reader %d DISAGREES with phi_reads_read\n", path, variant); return 1; }
    }
    printf("%s %016llx\n", path, (unsigned long long)h);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_selftest gfa|reads FILE...\n"); return 2; }
    int bad = 0;
    for (int i = 2; i < argc; i++) bad |= strcmp(argv[1], "gfa") == 0 ? do_gfa(argv[i]) : do_reads(argv[i]);
    return bad;
}
