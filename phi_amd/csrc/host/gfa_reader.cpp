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
#include <algorithm>
#include <queue>
#include <string>
#include <unordered_map>
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

// whole lines from a (possibly gzip-compressed) file
class LineReader {
public:
    explicit LineReader(const char *path) { fp_ = gzopen(path, "r"); if (fp_) gzbuffer(fp_, 1 << 20); }
    ~LineReader() { if (fp_) gzclose(fp_); }
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
        if (!got) return false;                       // end of file
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;                                  // last line without a newline
    }
private:
    gzFile fp_ = nullptr;
    char buf_[1 << 16];
    int pos_ = 0, len_ = 0;
};

static void split_tabs(const std::string &s, std::vector<std::pair<const char *, size_t>> &f)
{
    f.clear();
    size_t a = 0;
    for (;;) {
        size_t b = s.find('\t', a);
        if (b == std::string::npos) { f.emplace_back(s.data() + a, s.size() - a); break; }
        f.emplace_back(s.data() + a, b - a);
        a = b + 1;
    }
}

extern "C" {

int phi_gfa_read(const char *path, phi_graph **out, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    LineReader in(path);
    if (!in.ok()) return fail(err, err_cap, PHI_HOST_ERR_IO, "failed to load the GFA file %s", path);

    std::unordered_map<std::string, int32_t> name2id;
    std::vector<std::string> names, seqs;
    std::vector<char> has_seq;
    std::vector<std::pair<uint32_t, uint32_t>> arcs;                 // oriented vertices v = seg<<1 | strand
    struct Walk { std::string sample; int hap; std::vector<uint32_t> v; };
    std::vector<Walk> walks;
    auto add_seg = [&](const char *p, size_t n) {
        std::string key(p, n);
        auto it = name2id.find(key);
        if (it != name2id.end()) return it->second;
        const int32_t id = (int32_t)names.size();
        name2id.emplace(key, id);
        names.push_back(std::move(key));
        seqs.emplace_back();
        has_seq.push_back(0);
        return id;
    };

    std::string line;
    std::vector<std::pair<const char *, size_t>> f;
    while (in.next(line)) {
        if (line.size() < 3 || line[1] != '\t') continue;
        const char t = line[0];
        if (t != 'S' && t != 'L' && t != 'W') continue;
        split_tabs(line, f);
        if (t == 'S' && f.size() >= 3) {
            const int32_t id = add_seg(f[1].first, f[1].second);
            if (f[2].second > 0 && f[2].first[0] != '*') { seqs[id].assign(f[2].first, f[2].second); has_seq[id] = 1; }
            else { seqs[id].clear(); has_seq[id] = 0; }
        } else if (t == 'L' && f.size() >= 5) {
            if (f[2].second != 1 || f[4].second != 1) continue;
            const char ov = f[2].first[0], ow = f[4].first[0];
            if ((ov != '+' && ov != '-') || (ow != '+' && ow != '-')) continue;
            const uint32_t v = (uint32_t)add_seg(f[1].first, f[1].second) << 1 | (ov != '+');
            const uint32_t w = (uint32_t)add_seg(f[3].first, f[3].second) << 1 | (ow != '+');
            arcs.emplace_back(v, w);
        } else if (t == 'W' && f.size() >= 7) {
            Walk wk;
            wk.sample.assign(f[1].first, f[1].second);
            wk.hap = atoi(std::string(f[2].first, f[2].second).c_str());
            const char *s = f[6].first;
            const size_t n = f[6].second;
            size_t i = 0;
            while (i < n) {
                if (s[i] == '>' || s[i] == '<') {
                    size_t j = i + 1;
                    while (j < n && s[j] != '>' && s[j] != '<') j++;
                    auto it = name2id.find(std::string(s + i + 1, j - i - 1));
                    if (it != name2id.end()) wk.v.push_back((uint32_t)it->second << 1 | (s[i] == '<'));
                    i = j;
                } else i++;
            }
            walks.push_back(std::move(wk));
        }
    }

    const int32_t n_seg = (int32_t)names.size();
    // gfa_walk_flip: the first walk to touch a segment fixes its strand; a walk that disagrees
    // with the majority of its vertices is reverse-complemented
    {
        std::vector<int8_t> strand(n_seg, 0);
        for (const Walk &w : walks)
            for (uint32_t v : w.v)
                if (strand[v >> 1] == 0) strand[v >> 1] = (v & 1) ? -1 : 1;
        for (Walk &w : walks) {
            int64_t agree = 0;
            for (uint32_t v : w.v) agree += (((v & 1) ? -1 : 1) == strand[v >> 1]);
            if (agree >= (int64_t)w.v.size() - agree) continue;
            std::reverse(w.v.begin(), w.v.end());
            for (uint32_t &v : w.v) v ^= 1;
        }
    }
    // arcs: drop those touching a sequence-less segment, add complements, merge duplicates
    {
        std::vector<std::pair<uint32_t, uint32_t>> all;
        all.reserve(arcs.size() * 2);
        for (auto &a : arcs) {
            if (!has_seq[a.first >> 1] || seqs[a.first >> 1].empty()) continue;
            if (!has_seq[a.second >> 1] || seqs[a.second >> 1].empty()) continue;
            all.push_back(a);
            all.emplace_back(a.second ^ 1, a.first ^ 1);
        }
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        arcs.swap(all);
    }

    phi_graph *g = new phi_graph();
    g->seg_names = names;
    g->seq_off.assign(n_seg + 1, 0);
    for (int32_t i = 0; i < n_seg; i++) g->seq_off[i + 1] = g->seq_off[i] + (int64_t)seqs[i].size();
    g->seq_concat.reserve((size_t)g->seq_off[n_seg]);
    for (int32_t i = 0; i < n_seg; i++) g->seq_concat += seqs[i];
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
        for (uint32_t v : walks[w].v) {
            if (v & 1) {
                const int code = fail(err, err_cap, PHI_HOST_ERR_WALK, "Error: Walk %d has reverse strand vertices %u", (int)w, v);
                delete g;
                return code;
            }
            g->walk_vtx.push_back((int32_t)(v >> 1));
        }
        g->walk_off[w + 1] = (int64_t)g->walk_vtx.size();
        g->hap_names.push_back(walks[w].sample + "." + std::to_string(walks[w].hap));
    }
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
