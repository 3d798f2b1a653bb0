// gfa_reader.cpp -- GFA (S/L/W lines, plain or gzip) -> the flat forward-strand graph arrays of
// phi_set_graph.  Own implementation of the behaviour of the reference's vendored gfatools subset
// plus ILP_index::read_gfa (paths relative to /root/reference):
//   segment ids = first-seen order over S- and L-lines            src/gfa-base.cpp:75-96
//   S-line: name, sequence ('*' = none)                           src/gfa-io.cpp:214-277
//   L-line: v, strand, w, strand (overlap ignored: PHI graphs use 0M)  src/gfa-io.cpp:279-365
//   W-line: sample, hap index, ..., walk; names resolved against the segments seen so far
//                                                                  src/gfa-io.cpp:367-432
//   walk flip by majority strand                                   src/gfa-io.cpp:64-115
//   segments without sequence are dropped with their arcs          src/gfa-base.cpp:201-213, 306-326
//   every arc gets its complement                                  src/gfa-base.cpp:269-304
//   forward-strand adjacency with target orientation dropped       src/ILP_index.cpp:53-84
//   walks -> paths, haps names sample.hap; reverse vertex = error  src/ILP_index.cpp:96-113
//   Kahn topological order with a FIFO queue                       src/ILP_index.cpp:115-154
// Differences, by design: duplicate L-lines are merged; adjacency lists are sorted by target id
// (the reference's order depends on an unstable radix sort); a cyclic graph is an error.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "gz_source.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <queue>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/phi_host.h"

struct phi_graph {
    std::vector<std::string> seg_names, hap_names;
    std::string seq_concat;
    std::vector<int64_t> seq_off, adj_off, walk_off;
    std::vector<int32_t> adj, walk_vtx, topo_rank;
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

// the whole (possibly gzip-compressed) file in memory; lines and fields are slices of it
static bool slurp(const char *path, std::vector<char> &buf)
{
    // plain files: one read of the whole file; gzip (magic 1f 8b): inflate through zlib
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
    // gzip: inflated by gz_source.h -- on many threads when the file is block gzip (BGZF), else by one thread
    GzSource gz;
    const char *e = getenv("PHI_HOST_THREADS");
    int nt = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (!gz.open(path, nt)) return false;
    size_t len = 0;
    buf.clear();
    std::vector<char> blk;
    while (gz.next(blk)) {
        if (buf.capacity() < len + blk.size()) buf.reserve(std::max(buf.capacity() * 2, len + blk.size()));
        buf.insert(buf.end(), blk.begin(), blk.end());
        len += blk.size();
    }
    gz.close();
    return true;
}

struct Slice { const char *p; size_t n; };

// PHI_TIMING=1: stage timings on stderr
struct StageTimer {
    bool on = getenv("PHI_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *stage)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[phi timing] gfa_read: %-28s %8.3f ms\n", stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// segment name -> id: open addressing over slices of the file buffer
class NameTable {
public:
    NameTable() { grow(1 << 16); }
    static uint64_t hash(const char *p, size_t n)
    {
        uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9E3779B97F4A7C15ull);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
        uint64_t w = 0;
        if (i < n) { memcpy(&w, p + i, n - i); h = (h ^ w) * 0x100000001b3ull; }
        h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
        return h;
    }
    int32_t find(const char *p, size_t n) const
    {
        for (size_t i = hash(p, n) & mask_;; i = (i + 1) & mask_) {
            const int32_t id = slot_[i];
            if (id < 0) return -1;
            const Slice &k = keys_[id];
            if (k.n == n && memcmp(k.p, p, n) == 0) return id;
        }
    }
    int32_t add(const char *p, size_t n)            // id of an existing or new name
    {
        const int32_t f = find(p, n);
        if (f >= 0) return f;
        if ((keys_.size() + 1) * 2 > slot_.size()) grow(slot_.size() * 2);
        const int32_t id = (int32_t)keys_.size();
        keys_.push_back(Slice{p, n});
        insert(id);
        return id;
    }
    const std::vector<Slice> &keys() const { return keys_; }
private:
    void insert(int32_t id)
    {
        size_t i = hash(keys_[id].p, keys_[id].n) & mask_;
        while (slot_[i] >= 0) i = (i + 1) & mask_;
        slot_[i] = id;
    }
    void grow(size_t cap)
    {
        slot_.assign(cap, -1);
        mask_ = cap - 1;
        for (int32_t id = 0; id < (int32_t)keys_.size(); id++) insert(id);
    }
    std::vector<int32_t> slot_;
    std::vector<Slice> keys_;
    size_t mask_ = 0;
};

// fields of a line: up to cap tab-separated slices
static int split_tabs(const char *p, const char *e, Slice *f, int cap)
{
    int n = 0;
    while (n < cap) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
        if (!t || n == cap - 1) { f[n++] = Slice{p, (size_t)(e - p)}; break; }
        f[n++] = Slice{p, (size_t)(t - p)};
        p = t + 1;
    }
    return n;
}

template <class F> static void parallel_for(int n, F fn)
{
    int nt = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("PHI_HOST_THREADS")) nt = atoi(e);
    nt = std::max(1, std::min(std::min(nt, 16), n));
    if (nt == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    std::atomic<int> next{0};
    auto work = [&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

extern "C" {

int phi_gfa_read(const char *path, phi_graph **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    StageTimer tm;
    std::vector<char> buf;
    if (!slurp(path, buf)) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to load the GFA file %s", path);
    tm.lap("read / inflate");

    NameTable table;
    std::vector<Slice> seqs;                                         // per segment; n = 0: no sequence
    std::vector<char> has_seq;
    std::vector<std::pair<uint32_t, uint32_t>> arcs;                 // oriented vertices v = seg<<1 | strand
    struct Walk { std::string sample; int hap; std::vector<uint32_t> v; Slice text; int32_t n_known; bool any_rev = false; };
    std::vector<Walk> walks;
    auto add_seg = [&](const Slice &f) {
        const int32_t id = table.add(f.p, f.n);
        if ((size_t)id == seqs.size()) { seqs.push_back(Slice{nullptr, 0}); has_seq.push_back(0); }
        return id;
    };

    const char *p = buf.data(), *const end = buf.data() + buf.size();
    Slice f[8];
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (le > p && le[-1] == '\r') le--;
        if (le - p >= 3 && p[1] == '\t' && (p[0] == 'S' || p[0] == 'L' || p[0] == 'W')) {
            const char t = p[0];
            const int nf = split_tabs(p, le, f, t == 'W' ? 8 : 6);
            if (t == 'S' && nf >= 3) {
                const int32_t id = add_seg(f[1]);
                if (f[2].n > 0 && f[2].p[0] != '*') { seqs[id] = f[2]; has_seq[id] = 1; }
                else { seqs[id] = Slice{nullptr, 0}; has_seq[id] = 0; }
            } else if (t == 'L' && nf >= 5) {
                if (f[2].n == 1 && f[4].n == 1) {
                    const char ov = f[2].p[0], ow = f[4].p[0];
                    if ((ov == '+' || ov == '-') && (ow == '+' || ow == '-')) {
                        const uint32_t v = (uint32_t)add_seg(f[1]) << 1 | (ov != '+');
                        const uint32_t w = (uint32_t)add_seg(f[3]) << 1 | (ow != '+');
                        arcs.emplace_back(v, w);
                    }
                }
            } else if (t == 'W' && nf >= 7) {
                Walk wk;
                wk.sample.assign(f[1].p, f[1].n);
                wk.hap = atoi(std::string(f[2].p, f[2].n).c_str());
                wk.text = f[6];
                wk.n_known = (int32_t)seqs.size();           // names are resolved against the segments seen so far
                walks.push_back(std::move(wk));
            }
        }
        p = next;
    }
    tm.lap("S / L lines");
    // the walks' vertex lists: one host thread per W-line
    parallel_for((int)walks.size(), [&](int wi) {
        Walk &wk = walks[wi];
        const char *s = wk.text.p;
        const size_t n = wk.text.n;
        wk.v.reserve(n / 4 + 4);
        size_t i = 0;
        while (i < n) {
            if (s[i] == '>' || s[i] == '<') {
                size_t j = i + 1;
                while (j < n && s[j] != '>' && s[j] != '<' && s[j] != '\t') j++;
                const int32_t id = table.find(s + i + 1, j - i - 1);
                if (id >= 0 && id < wk.n_known) {
                    wk.v.push_back((uint32_t)id << 1 | (s[i] == '<'));
                    wk.any_rev |= s[i] == '<';
                }
                i = j;
            } else if (s[i] == '\t') break;                // optional tags after the walk
            else i++;
        }
    });
    const std::vector<Slice> &names = table.keys();
    tm.lap("W lines (threads)");

    const int32_t n_seg = (int32_t)names.size();
    // gfa_walk_flip: the first walk to touch a segment fixes its strand; a walk that disagrees
    // with the majority of its vertices is reverse-complemented
    bool any_rev = false;
    for (const Walk &w : walks) any_rev |= w.any_rev;
    if (any_rev) {                                     // all-forward walks agree with every first touch: nothing to flip
        std::vector<int8_t> strand(n_seg, 0);
        for (const Walk &w : walks)
            for (uint32_t v : w.v)
                if (strand[v >> 1] == 0) strand[v >> 1] = (v & 1) ? -1 : 1;
        parallel_for((int)walks.size(), [&](int wi) {
            Walk &w = walks[wi];
            int64_t agree = 0;
            for (uint32_t v : w.v) agree += (((v & 1) ? -1 : 1) == strand[v >> 1]);
            if (agree >= (int64_t)w.v.size() - agree) return;
            std::reverse(w.v.begin(), w.v.end());
            for (uint32_t &v : w.v) v ^= 1;
        });
    }
    // arcs: drop those touching a sequence-less segment, add complements, merge duplicates
    {
        std::vector<std::pair<uint32_t, uint32_t>> all;
        all.reserve(arcs.size() * 2);
        for (auto &a : arcs) {
            if (!has_seq[a.first >> 1] || seqs[a.first >> 1].n == 0) continue;
            if (!has_seq[a.second >> 1] || seqs[a.second >> 1].n == 0) continue;
            all.push_back(a);
            all.emplace_back(a.second ^ 1, a.first ^ 1);
        }
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        arcs.swap(all);
    }

    tm.lap("walk flips, arcs");
    phi_graph *g = new phi_graph();
    g->seg_names.reserve(n_seg);
    for (int32_t i = 0; i < n_seg; i++) g->seg_names.emplace_back(names[i].p, names[i].n);
    g->seq_off.assign(n_seg + 1, 0);
    for (int32_t i = 0; i < n_seg; i++) g->seq_off[i + 1] = g->seq_off[i] + (int64_t)seqs[i].n;
    g->seq_concat.resize((size_t)g->seq_off[n_seg]);
    for (int32_t i = 0; i < n_seg; i++) if (seqs[i].n) memcpy(&g->seq_concat[(size_t)g->seq_off[i]], seqs[i].p, seqs[i].n);
    g->adj_off.assign(n_seg + 1, 0);
    for (auto &a : arcs) if (!(a.first & 1)) g->adj_off[(a.first >> 1) + 1]++;
    for (int32_t i = 0; i < n_seg; i++) g->adj_off[i + 1] += g->adj_off[i];
    g->adj.resize((size_t)g->adj_off[n_seg]);
    {
        std::vector<int64_t> cur(g->adj_off.begin(), g->adj_off.end() - 1);
        for (auto &a : arcs) if (!(a.first & 1)) g->adj[cur[a.first >> 1]++] = (int32_t)(a.second >> 1);
        // two oriented targets can collapse onto one segment once orientation is dropped
        std::vector<int64_t> off2(n_seg + 1, 0);
        std::vector<int32_t> adj2;
        for (int32_t u = 0; u < n_seg; u++) {
            auto b = g->adj.begin() + g->adj_off[u], e = g->adj.begin() + g->adj_off[u + 1];
            std::sort(b, e);
            e = std::unique(b, e);
            adj2.insert(adj2.end(), b, e);
            off2[u + 1] = (int64_t)adj2.size();
        }
        g->adj.swap(adj2);
        g->adj_off.swap(off2);
    }
    g->walk_off.assign(walks.size() + 1, 0);
    for (size_t w = 0; w < walks.size(); w++) {
        g->walk_off[w + 1] = g->walk_off[w] + (int64_t)walks[w].v.size();
        g->hap_names.push_back(walks[w].sample + "." + std::to_string(walks[w].hap));
    }
    g->walk_vtx.resize((size_t)g->walk_off[walks.size()]);
    {
        std::atomic<int64_t> bad{-1};                  // (walk << 32 | vertex) of the first reverse-strand vertex
        parallel_for((int)walks.size(), [&](int wi) {
            int32_t *dst = g->walk_vtx.data() + g->walk_off[wi];
            const std::vector<uint32_t> &v = walks[wi].v;
            for (size_t i = 0; i < v.size(); i++) {
                if (v[i] & 1) {
                    int64_t expect = -1;
                    bad.compare_exchange_strong(expect, (int64_t)wi << 32 | v[i]);
                    return;
                }
                dst[i] = (int32_t)(v[i] >> 1);
            }
        });
        int64_t first = -1;                            // report the lowest walk, as the sequential loop did
        if (bad.load() >= 0)
            for (size_t w = 0; w < walks.size() && first < 0; w++)
                for (uint32_t v : walks[w].v) if (v & 1) { first = (int64_t)w << 32 | v; break; }
        if (first >= 0) {
            const int code = fail(err, err_cap, PHI_HOST_ERR_WALK, "Error: Walk %d has reverse strand vertices %u", (int)(first >> 32), (uint32_t)first);
            delete g;
            return code;
        }
    }
    tm.lap("arrays");
    // Kahn's algorithm, FIFO
    {
        std::vector<int32_t> indeg(n_seg, 0);
        for (int32_t v : g->adj) indeg[v]++;
        std::queue<int32_t> q;
        for (int32_t i = 0; i < n_seg; i++) if (indeg[i] == 0) q.push(i);
        g->topo_rank.assign(n_seg, 0);
        int32_t n_done = 0;
        while (!q.empty()) {
            const int32_t u = q.front();
            q.pop();
            g->topo_rank[u] = n_done++;
            for (int64_t x = g->adj_off[u]; x < g->adj_off[u + 1]; x++)
                if (--indeg[g->adj[x]] == 0) q.push(g->adj[x]);
        }
        if (n_done != n_seg) {
            const int code = fail(err, err_cap, PHI_HOST_ERR_CYCLE, "graph is not acyclic: %d of %d vertices sorted", n_done, n_seg);
            delete g;
            return code;
        }
    }
    tm.lap("topological order");
    *out = g;
    return PHI_HOST_OK;
}

void phi_graph_free(phi_graph *g) { delete g; }
int32_t phi_graph_n_vtx(const phi_graph *g) { return (int32_t)g->seg_names.size(); }
int32_t phi_graph_n_walks(const phi_graph *g) { return (int32_t)g->hap_names.size(); }
int64_t phi_graph_n_edges(const phi_graph *g) { return (int64_t)g->adj.size(); }
const char *phi_graph_seq_concat(const phi_graph *g) { return g->seq_concat.data(); }
const int64_t *phi_graph_seq_off(const phi_graph *g) { return g->seq_off.data(); }
const int64_t *phi_graph_adj_off(const phi_graph *g) { return g->adj_off.data(); }
const int32_t *phi_graph_adj(const phi_graph *g) { return g->adj.data(); }
const int64_t *phi_graph_walk_off(const phi_graph *g) { return g->walk_off.data(); }
const int32_t *phi_graph_walk_vtx(const phi_graph *g) { return g->walk_vtx.data(); }
const int32_t *phi_graph_topo_rank(const phi_graph *g) { return g->topo_rank.data(); }
const char *phi_graph_hap_name(const phi_graph *g, int32_t w)
{
    return (w >= 0 && w < (int32_t)g->hap_names.size()) ? g->hap_names[w].c_str() : "";
}
const char *phi_graph_seg_name(const phi_graph *g, int32_t v)
{
    return (v >= 0 && v < (int32_t)g->seg_names.size()) ? g->seg_names[v].c_str() : "";
}

}  // extern "C"
