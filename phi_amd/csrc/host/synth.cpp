// synth.cpp -- native generator of the synthetic stand-ins for the benchmark configurations
// (libphi_synth.so; bench / test infrastructure, not part of the drop-in path).
//
// The reference's 49-haplotype MHC graph and its chr6-scale extrapolation need network downloads and
// external tools (data/preprocess.py:34-55, data/chop_graph.sh), so BASELINE.json's configurations run on
// the generator model of SURVEY.md section 8(d) -- the same model as phi_amd/synth.py (uniform backbone,
// bi-allelic sites every ~250 bp: 80 % SNP, 15 % indel of 1-50 bp, 5 % SV of 50-5000 bp some of them copies
// of other backbone segments, walks following founder haplotypes per ~20 kb block with private mutations,
// nodes chopped to <= 30 bp as chop_graph.sh:62 does, reads from a mosaic of walks with substitution errors
// on both strands).  phi_amd/synth.py draws from numpy's PCG64 and builds the arrays in Python, which takes
// minutes at chromosome scale (170 Mbp x 200 walks = 1.2 G walk entries, 34 M reads); this one draws every
// value from a counter-based generator -- SplitMix64 of (seed, stream, index) -- so that any part can be made
// by any thread, in any order, in chunks, and is the same on every machine.
#include <stdint.h>
#include <stdlib.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// value number i of stream s of the generator seeded with `seed`
inline uint64_t draw(uint64_t seed, uint64_t stream, uint64_t i) { return mix64(mix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + i * 0x9E3779B97F4A7C15ull); }
inline double unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

template <class F> void par_for(int64_t n, int threads, F fn)
{
    if (threads < 1) threads = 1;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(1 << 16, (n + threads * 8 - 1) / (threads * 8)));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t lo = next.fetch_add(chunk);
            if (lo >= n) break;
            fn(lo, std::min(n, lo + chunk));
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

const char ACGT[4] = {'A', 'C', 'G', 'T'};
inline int code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

}  // namespace

struct phi_syn {
    int threads = 1;
    uint64_t seed = 0;
    int32_t n_vtx = 0, n_walks = 0;
    int64_t n_sites = 0;
    std::vector<char> seq;                 // node sequences, node order
    std::vector<int64_t> seq_off, adj_off, walk_off;
    std::vector<int32_t> adj, topo_rank;
    int32_t *walk_vtx = nullptr;           // malloc: 4.8 GB at chromosome scale, filled by the threads that touch it first
    // units: piece i = nodes [piece_first[i], piece_first[i] + piece_n[i]); allele a of site i likewise
    std::vector<int32_t> piece_first, piece_n, al_first[2], al_n[2];
    std::vector<int64_t> site_block;       // block of every site (founder switches happen at block seams)
    std::vector<uint8_t> founder;          // [n_founders][n_sites]
    std::vector<int8_t> walk_founder;      // [n_walks][n_blocks]
    int64_t n_blocks = 0;
    int32_t n_founders = 0;
    // sample
    std::vector<char> hap;
    ~phi_syn() { free(walk_vtx); }
    inline int allele(int32_t h, int64_t site) const
    {
        const int f = walk_founder[(size_t)h * (size_t)n_blocks + (size_t)site_block[(size_t)site]];
        const int priv = unit(draw(seed, 70 + (uint64_t)h, (uint64_t)site)) < 0.01;
        return founder[(size_t)f * (size_t)n_sites + (size_t)site] ^ priv;
    }
};

extern "C" {

// 0 on success; *out owns everything
int phi_syn_graph(int64_t backbone_len, int32_t n_walks, uint64_t seed, int32_t site_spacing, int32_t chop, int32_t block_len,
                  int32_t n_founders, int32_t max_sv, int32_t threads, phi_syn **out)
{
    if (!out || backbone_len < 4 * (int64_t)(max_sv + chop) || n_walks < 1 || site_spacing < 4 || chop < 1 || n_founders < 1 || n_founders > 127) return -1;
    phi_syn *g = new (std::nothrow) phi_syn();
    if (!g) return -2;
    g->threads = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
    if (g->threads > 64) g->threads = 64;
    g->seed = seed; g->n_walks = n_walks; g->n_founders = n_founders;
    // ---- backbone
    std::vector<char> backbone((size_t)backbone_len);
    par_for(backbone_len, g->threads, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) backbone[(size_t)i] = ACGT[draw(seed, 1, (uint64_t)i) & 3];
    });
    // ---- sites on a jittered grid (ordered, distinct), kept when the previous site's reference allele ended before
    struct Site { int64_t pos; int32_t ref_len; int32_t alt_len; int64_t alt_src; int8_t alt_kind; };   // alt_kind 0 none, 1 SNP base, 2 random, 3 backbone copy
    std::vector<Site> sites;
    {
        const int64_t lo_pos = chop + 1, hi_pos = backbone_len - max_sv - chop - 1;
        const int64_t n_grid = std::max<int64_t>(1, backbone_len / site_spacing);
        int64_t cur = 0;
        for (int64_t j = 0; j < n_grid; j++) {
            const int64_t p = j * site_spacing + (int64_t)(draw(seed, 2, (uint64_t)j) % (uint64_t)site_spacing);
            if (p < lo_pos || p >= hi_pos || p < cur + 1) continue;
            const double u = unit(draw(seed, 3, (uint64_t)j));
            Site s{p, 0, 0, 0, 0};
            if (u < 0.80) { s.ref_len = 1; s.alt_len = 1; s.alt_kind = 1; }
            else if (u < 0.95) {
                const int32_t L = 1 + (int32_t)(draw(seed, 4, (uint64_t)j) % 50);
                if (draw(seed, 5, (uint64_t)j) & 1) { s.ref_len = L; }                      // deletion
                else { s.alt_len = L; s.alt_kind = 2; }                                    // insertion
            } else {
                const int32_t L = 50 + (int32_t)(draw(seed, 4, (uint64_t)j) % (uint64_t)(max_sv - 49));
                const double r = unit(draw(seed, 5, (uint64_t)j));
                if (r < 0.4) s.ref_len = L;
                else if (r < 0.7) { s.alt_len = L; s.alt_kind = 2; }
                else { s.alt_len = L; s.alt_kind = 3; s.alt_src = (int64_t)(draw(seed, 6, (uint64_t)j) % (uint64_t)(backbone_len - L)); }
            }
            if (p + s.ref_len >= backbone_len - chop) break;
            sites.push_back(s);
            cur = p + s.ref_len;
        }
    }
    const int64_t ns = (int64_t)sites.size();
    g->n_sites = ns;
    // ---- units -> nodes (consecutive ids: piece 0, ref 0, alt 0, piece 1, ...), sequences, offsets
    auto n_nodes = [&](int64_t len) { return (int32_t)((len + chop - 1) / chop); };
    g->piece_first.resize(ns + 1); g->piece_n.resize(ns + 1);
    for (int a = 0; a < 2; a++) { g->al_first[a].resize(ns); g->al_n[a].resize(ns); }
    std::vector<int64_t> piece_lo(ns + 1), piece_hi(ns + 1), unit_seq(3 * ns + 2);   // unit u = 3i (piece), 3i+1 (ref), 3i+2 (alt): sequence offset
    int64_t node = 0, so = 0, cur = 0;
    for (int64_t i = 0; i <= ns; i++) {
        piece_lo[i] = cur; piece_hi[i] = i < ns ? sites[(size_t)i].pos : backbone_len;
        const int64_t plen = piece_hi[i] - piece_lo[i];
        g->piece_first[i] = (int32_t)node; g->piece_n[i] = n_nodes(plen);
        unit_seq[3 * i] = so; so += plen; node += g->piece_n[i];
        if (i < ns) {
            const Site &s = sites[(size_t)i];
            g->al_first[0][i] = (int32_t)node; g->al_n[0][i] = n_nodes(s.ref_len); unit_seq[3 * i + 1] = so; so += s.ref_len; node += g->al_n[0][i];
            g->al_first[1][i] = (int32_t)node; g->al_n[1][i] = n_nodes(s.alt_len); unit_seq[3 * i + 2] = so; so += s.alt_len; node += g->al_n[1][i];
            cur = s.pos + s.ref_len;
        }
        if (node >= ((int64_t)1 << 31) - 4) { delete g; return -3; }
    }
    unit_seq[3 * ns + 1] = so;
    g->n_vtx = (int32_t)node;
    g->seq.resize((size_t)so);
    g->seq_off.resize((size_t)node + 1);
    par_for(ns + 1, g->threads, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            auto emit = [&](int64_t first, int32_t n, int64_t off, int64_t len) {
                for (int32_t j = 0; j < n; j++) g->seq_off[(size_t)(first + j)] = off + (int64_t)j * chop;
                (void)len;
            };
            const int64_t plen = piece_hi[i] - piece_lo[i];
            memcpy(&g->seq[(size_t)unit_seq[3 * i]], &backbone[(size_t)piece_lo[i]], (size_t)plen);
            emit(g->piece_first[i], g->piece_n[i], unit_seq[3 * i], plen);
            if (i == ns) continue;
            const Site &s = sites[(size_t)i];
            if (s.ref_len) memcpy(&g->seq[(size_t)unit_seq[3 * i + 1]], &backbone[(size_t)s.pos], (size_t)s.ref_len);
            emit(g->al_first[0][i], g->al_n[0][i], unit_seq[3 * i + 1], s.ref_len);
            char *alt = &g->seq[(size_t)unit_seq[3 * i + 2]];
            if (s.alt_kind == 1) alt[0] = ACGT[(code_of(backbone[(size_t)s.pos]) + 1 + (int)(draw(seed, 7, (uint64_t)i) % 3)) & 3];
            else if (s.alt_kind == 2) for (int32_t j = 0; j < s.alt_len; j++) alt[j] = ACGT[draw(seed, 8, (uint64_t)i * 8192 + (uint64_t)j) & 3];
            else if (s.alt_kind == 3) memcpy(alt, &backbone[(size_t)s.alt_src], (size_t)s.alt_len);
            emit(g->al_first[1][i], g->al_n[1][i], unit_seq[3 * i + 2], s.alt_len);
        }
    });
    g->seq_off[(size_t)node] = so;
    // ---- edges (forward, targets ascending): inside a unit to the next node; a piece's last node to the first node
    //      of each non-empty allele and, when an allele is empty, to the next piece; an allele's last node to the next piece
    g->adj_off.assign((size_t)node + 1, 0);
    {
        std::vector<uint8_t> deg((size_t)node, 1);             // inner nodes: one edge
        for (int64_t i = 0; i <= ns; i++) {
            const int64_t last = (int64_t)g->piece_first[i] + g->piece_n[i] - 1;
            if (i == ns) { deg[(size_t)last] = 0; break; }
            const int e0 = g->al_n[0][i] == 0, e1 = g->al_n[1][i] == 0;
            deg[(size_t)last] = (uint8_t)((!e0) + (!e1) + ((e0 || e1) ? 1 : 0));
        }
        for (int64_t v = 0; v < node; v++) g->adj_off[(size_t)v + 1] = g->adj_off[(size_t)v] + deg[(size_t)v];
        g->adj.resize((size_t)g->adj_off[(size_t)node]);
        par_for(node, g->threads, [&](int64_t lo, int64_t hi) {
            for (int64_t v = lo; v < hi; v++) if (deg[(size_t)v] == 1) g->adj[(size_t)g->adj_off[(size_t)v]] = (int32_t)(v + 1);
        });
        for (int64_t i = 0; i < ns; i++) {
            const int64_t last = (int64_t)g->piece_first[i] + g->piece_n[i] - 1;
            int64_t x = g->adj_off[(size_t)last];
            const int32_t next_piece = g->piece_first[i + 1];
            for (int a = 0; a < 2; a++)
                if (g->al_n[a][i]) {
                    g->adj[(size_t)x++] = g->al_first[a][i];
                    g->adj[(size_t)g->adj_off[(size_t)g->al_first[a][i] + g->al_n[a][i] - 1]] = next_piece;
                }
            if (!g->al_n[0][i] || !g->al_n[1][i]) g->adj[(size_t)x++] = next_piece;
        }
    }
    g->topo_rank.resize((size_t)node);
    for (int64_t v = 0; v < node; v++) g->topo_rank[(size_t)v] = (int32_t)v;       // ids were issued left to right
    // ---- walks: founder haplotypes per block, hops at block seams, private mutations
    g->site_block.resize((size_t)ns);
    for (int64_t i = 0; i < ns; i++) g->site_block[(size_t)i] = sites[(size_t)i].pos / block_len;
    g->n_blocks = backbone_len / block_len + 1;
    g->founder.resize((size_t)n_founders * (size_t)ns);
    par_for(ns, g->threads, [&](int64_t lo, int64_t hi) {
        for (int f = 0; f < n_founders; f++)
            for (int64_t i = lo; i < hi; i++) g->founder[(size_t)f * (size_t)ns + (size_t)i] = unit(draw(seed, 20 + (uint64_t)f, (uint64_t)i)) < 0.35;
    });
    g->walk_founder.resize((size_t)n_walks * (size_t)g->n_blocks);
    for (int32_t h = 0; h < n_walks; h++) {
        int f = (int)(draw(seed, 40, (uint64_t)h) % (uint64_t)n_founders);
        for (int64_t b = 0; b < g->n_blocks; b++) {
            if (unit(draw(seed, 41, (uint64_t)h * (uint64_t)g->n_blocks + (uint64_t)b)) < 0.3) f = (int)(draw(seed, 42, (uint64_t)h * (uint64_t)g->n_blocks + (uint64_t)b) % (uint64_t)n_founders);
            g->walk_founder[(size_t)h * (size_t)g->n_blocks + (size_t)b] = (int8_t)f;
        }
    }
    g->walk_off.assign((size_t)n_walks + 1, 0);
    {
        std::vector<int64_t> cnt((size_t)n_walks, 0);
        int64_t pieces = 0;
        for (int64_t i = 0; i <= ns; i++) pieces += g->piece_n[i];
        par_for(n_walks, g->threads, [&](int64_t lo, int64_t hi) {
            for (int64_t h = lo; h < hi; h++) {
                int64_t c = pieces;
                for (int64_t i = 0; i < ns; i++) c += g->al_n[g->allele((int32_t)h, i)][i];
                cnt[(size_t)h] = c;
            }
        });
        for (int32_t h = 0; h < n_walks; h++) g->walk_off[(size_t)h + 1] = g->walk_off[(size_t)h] + cnt[(size_t)h];
    }
    const int64_t n_entries = g->walk_off[(size_t)n_walks];
    if (n_entries > ((int64_t)1 << 32) - 64) { delete g; return -3; }     // (PHI_MAX_ENTRIES of the library)
    g->walk_vtx = (int32_t *)malloc((size_t)std::max<int64_t>(n_entries, 1) * 4);
    if (!g->walk_vtx) { delete g; return -2; }
    {
        std::atomic<int32_t> next{0};
        auto work = [&]() {
            for (;;) {
                const int32_t h = next.fetch_add(1);
                if (h >= n_walks) break;
                int32_t *w = g->walk_vtx + g->walk_off[(size_t)h];
                for (int64_t i = 0; i <= ns; i++) {
                    for (int32_t j = 0, f = g->piece_first[i]; j < g->piece_n[i]; j++) *w++ = f + j;
                    if (i == ns) break;
                    const int a = g->allele(h, i);
                    for (int32_t j = 0, f = g->al_first[a][i]; j < g->al_n[a][i]; j++) *w++ = f + j;
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < g->threads; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    *out = g;
    return 0;
}

void phi_syn_free(phi_syn *g) { delete g; }
int32_t phi_syn_n_vtx(const phi_syn *g) { return g->n_vtx; }
int32_t phi_syn_n_walks(const phi_syn *g) { return g->n_walks; }
int64_t phi_syn_n_sites(const phi_syn *g) { return g->n_sites; }
const char *phi_syn_seq(const phi_syn *g) { return g->seq.data(); }
const int64_t *phi_syn_seq_off(const phi_syn *g) { return g->seq_off.data(); }
const int64_t *phi_syn_adj_off(const phi_syn *g) { return g->adj_off.data(); }
const int32_t *phi_syn_adj(const phi_syn *g) { return g->adj.data(); }
const int64_t *phi_syn_walk_off(const phi_syn *g) { return g->walk_off.data(); }
const int32_t *phi_syn_walk_vtx(const phi_syn *g) { return g->walk_vtx; }
const int32_t *phi_syn_topo_rank(const phi_syn *g) { return g->topo_rank.data(); }

// The sample reads are drawn from: a mosaic of n_mosaic distinct walks, part i taken from walk i between the
// fractions cuts[i-1] and cuts[i] of that walk's own length.  Returns the mosaic's length.
int64_t phi_syn_sample(phi_syn *g, uint64_t seed, int32_t n_mosaic, int32_t *walks_out, double *cuts_out)
{
    if (!g || n_mosaic < 1 || n_mosaic > g->n_walks || n_mosaic > 64) return -1;
    std::vector<int32_t> ws;
    for (uint64_t t = 0; (int32_t)ws.size() < n_mosaic; t++) {
        const int32_t h = (int32_t)(draw(seed, 1, t) % (uint64_t)g->n_walks);
        if (std::find(ws.begin(), ws.end(), h) == ws.end()) ws.push_back(h);
    }
    std::vector<double> cuts;
    for (int32_t i = 0; i + 1 < n_mosaic; i++) cuts.push_back(unit(draw(seed, 2, (uint64_t)i)));
    std::sort(cuts.begin(), cuts.end());
    g->hap.clear();
    for (int32_t i = 0; i < n_mosaic; i++) {
        const int32_t h = ws[(size_t)i];
        const int32_t *w = g->walk_vtx + g->walk_off[(size_t)h];
        const int64_t ne = g->walk_off[(size_t)h + 1] - g->walk_off[(size_t)h];
        int64_t L = 0;
        for (int64_t e = 0; e < ne; e++) L += g->seq_off[(size_t)w[e] + 1] - g->seq_off[(size_t)w[e]];
        const int64_t a = i == 0 ? 0 : (int64_t)(cuts[(size_t)i - 1] * (double)L), b = i == n_mosaic - 1 ? L : (int64_t)(cuts[(size_t)i] * (double)L);
        int64_t o = 0;
        for (int64_t e = 0; e < ne && o < b; e++) {
            const int64_t s0 = g->seq_off[(size_t)w[e]], len = g->seq_off[(size_t)w[e] + 1] - s0;
            const int64_t lo = std::max<int64_t>(a, o), hi = std::min<int64_t>(b, o + len);
            if (hi > lo) g->hap.insert(g->hap.end(), g->seq.begin() + (s0 + lo - o), g->seq.begin() + (s0 + hi - o));
            o += len;
        }
        if (walks_out) walks_out[i] = h;
        if (cuts_out && i + 1 < n_mosaic) cuts_out[i] = cuts[(size_t)i];
    }
    return (int64_t)g->hap.size();
}

// Reads r_lo .. r_hi-1 of the read set `seed` of the current sample, read_len bases each, written back to back
// into out ((r_hi - r_lo) * read_len bytes): a uniform start on the mosaic, substitution errors at rate sub_err,
// every other read (by its own draw) reverse-complemented.  Any range, any order, any thread count: the same bytes.
int phi_syn_reads(const phi_syn *g, uint64_t seed, int64_t r_lo, int64_t r_hi, int32_t read_len, double sub_err, char *out, int32_t threads)
{
    if (!g || !out || r_lo < 0 || r_hi < r_lo || read_len < 1) return -1;
    const int64_t L = (int64_t)g->hap.size();
    if (L < read_len) return -1;
    const uint32_t thr = (uint32_t)(sub_err * 65536.0);
    par_for(r_hi - r_lo, threads > 0 ? threads : g->threads, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            const uint64_t r = (uint64_t)(r_lo + i);
            const int64_t start = (int64_t)(draw(seed, 10, r) % (uint64_t)(L - read_len + 1));
            const bool rc = draw(seed, 11, r) & 1;
            char *o = out + i * (int64_t)read_len;
            const char *src = g->hap.data() + start;
            for (int32_t j = 0; j < read_len; j += 4) {
                const uint64_t e = draw(seed, 12, r * 4096 + (uint64_t)(j >> 2));     // 4 x (16-bit error draw) per value
                for (int32_t q = 0; q < 4 && j + q < read_len; q++) {
                    char b = src[j + q];
                    const uint32_t x = (uint32_t)(e >> (16 * q)) & 0xFFFFu;
                    if (x < thr) b = ACGT[(code_of(b) + 1 + (int)(x % 3)) & 3];
                    if (rc) o[read_len - 1 - (j + q)] = ACGT[3 - code_of(b)];
                    else o[j + q] = b;
                }
            }
        }
    });
    return 0;
}


// ---- the same configuration as FILES (the command line's inputs): GFA 1.1 (S / L / W lines, names = 1-based vertex ids,
//      as phi_amd/synth.py write_gfa) and FASTA / 4-line FASTQ reads.  Formatted by all threads into buffers, written with
//      pwrite at the offsets their sizes give: 12 GB of W-lines and 10 GB of reads take seconds on a RAM disk.
}  // extern "C"
namespace {
inline char *put_u64(char *p, uint64_t v)
{
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
}
struct Part { std::vector<char> buf; };
// parts[i] made by make(i, buf) on all threads, a wave of `wave` parts at a time, appended to fd in order
template <class F> int write_parts(int fd, int64_t &at, int64_t n_parts, int threads, int64_t wave, F make)
{
    std::vector<Part> parts((size_t)wave);
    for (int64_t base = 0; base < n_parts; base += wave) {
        const int64_t m = std::min(wave, n_parts - base);
        par_for(m, threads, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; i++) { parts[(size_t)i].buf.clear(); make(base + i, parts[(size_t)i].buf); } });
        std::vector<int64_t> off((size_t)m + 1, at);
        for (int64_t i = 0; i < m; i++) off[(size_t)i + 1] = off[(size_t)i] + (int64_t)parts[(size_t)i].buf.size();
        if (ftruncate(fd, off[(size_t)m]) != 0) return -1;
        std::atomic<int> bad{0};
        par_for(m, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                const char *p = parts[(size_t)i].buf.data();
                int64_t left = (int64_t)parts[(size_t)i].buf.size(), o = off[(size_t)i];
                while (left > 0) {
                    const ssize_t w = pwrite(fd, p, (size_t)std::min<int64_t>(left, (int64_t)1 << 30), (off_t)o);
                    if (w <= 0) { bad.store(1); return; }
                    p += w; o += w; left -= w;
                }
            }
        });
        if (bad.load()) return -1;
        at = off[(size_t)m];
    }
    return 0;
}
}  // namespace
extern "C" {

// returns the bytes written, negative on error
int64_t phi_syn_write_gfa(const phi_syn *g, const char *path, int32_t threads)
{
    if (!g || !path) return -1;
    if (threads < 1) threads = g->threads;
    const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) return -1;
    int64_t at = 0;
    const char hdr[] = "H\tVN:Z:1.1\n";
    if (pwrite(fd, hdr, sizeof hdr - 1, 0) != (ssize_t)(sizeof hdr - 1)) { close(fd); return -1; }
    at = sizeof hdr - 1;
    const int64_t VCH = 1 << 16, n_vch = (g->n_vtx + VCH - 1) / VCH;
    // S-lines, then L-lines, by chunks of vertices
    int rc = write_parts(fd, at, n_vch, threads, (int64_t)threads * 4, [&](int64_t c, std::vector<char> &b) {
        const int64_t lo = c * VCH, hi = std::min<int64_t>(g->n_vtx, lo + VCH);
        b.resize((size_t)((hi - lo) * 16 + (g->seq_off[(size_t)hi] - g->seq_off[(size_t)lo])));
        char *p = b.data();
        for (int64_t v = lo; v < hi; v++) {
            *p++ = 'S'; *p++ = '\t'; p = put_u64(p, (uint64_t)v + 1); *p++ = '\t';
            const int64_t n = g->seq_off[(size_t)v + 1] - g->seq_off[(size_t)v];
            memcpy(p, g->seq.data() + g->seq_off[(size_t)v], (size_t)n); p += n;
            *p++ = '\n';
        }
        b.resize((size_t)(p - b.data()));
    });
    if (!rc) rc = write_parts(fd, at, n_vch, threads, (int64_t)threads * 4, [&](int64_t c, std::vector<char> &b) {
        const int64_t lo = c * VCH, hi = std::min<int64_t>(g->n_vtx, lo + VCH);
        b.resize((size_t)((g->adj_off[(size_t)hi] - g->adj_off[(size_t)lo]) * 36 + 16));
        char *p = b.data();
        for (int64_t u = lo; u < hi; u++)
            for (int64_t x = g->adj_off[(size_t)u]; x < g->adj_off[(size_t)u + 1]; x++) {
                *p++ = 'L'; *p++ = '\t'; p = put_u64(p, (uint64_t)u + 1); memcpy(p, "\t+\t", 3); p += 3;
                p = put_u64(p, (uint64_t)g->adj[(size_t)x] + 1); memcpy(p, "\t+\t0M\n", 6); p += 6;
            }
        b.resize((size_t)(p - b.data()));
    });
    // W-lines: a walk is one line; pieces of 4 M entries, the first carrying the line's head, the last its line feed
    struct WP { int32_t h; int64_t lo, hi; };
    std::vector<WP> wps;
    const int64_t WCHE = (int64_t)4 << 20;
    for (int32_t h = 0; h < g->n_walks; h++) {
        const int64_t e0 = g->walk_off[(size_t)h], e1 = g->walk_off[(size_t)h + 1];
        for (int64_t lo = e0; lo < e1 || lo == e0; lo += WCHE) { wps.push_back(WP{h, lo, std::min(e1, lo + WCHE)}); if (e1 == e0) break; }
    }
    std::vector<int64_t> walk_len((size_t)g->n_walks, 0);
    par_for(g->n_walks, threads, [&](int64_t lo, int64_t hi) {
        for (int64_t h = lo; h < hi; h++) {
            int64_t L = 0;
            for (int64_t e = g->walk_off[(size_t)h]; e < g->walk_off[(size_t)h + 1]; e++) L += g->seq_off[(size_t)g->walk_vtx[e] + 1] - g->seq_off[(size_t)g->walk_vtx[e]];
            walk_len[(size_t)h] = L;
        }
    });
    if (!rc) rc = write_parts(fd, at, (int64_t)wps.size(), threads, (int64_t)threads * 2, [&](int64_t i, std::vector<char> &b) {
        const WP &w = wps[(size_t)i];
        b.resize((size_t)((w.hi - w.lo) * 11 + 96));
        char *p = b.data();
        if (w.lo == g->walk_off[(size_t)w.h]) {
            p += sprintf(p, "W\tsyn%03d\t%d\tchr\t0\t%lld\t", (int)w.h, (int)(w.h % 2), (long long)walk_len[(size_t)w.h]);
        }
        for (int64_t e = w.lo; e < w.hi; e++) { *p++ = '>'; p = put_u64(p, (uint64_t)g->walk_vtx[e] + 1); }
        if (w.hi == g->walk_off[(size_t)w.h + 1]) *p++ = '\n';
        b.resize((size_t)(p - b.data()));
    });
    close(fd);
    return rc ? -1 : at;
}

// reads r_lo .. r_hi-1 of read set `seed` as a FASTA (fastq = 0) or 4-line FASTQ file; returns the bytes written
int64_t phi_syn_write_reads(const phi_syn *g, uint64_t seed, int64_t r_lo, int64_t r_hi, int32_t read_len, double sub_err, const char *path,
                            int32_t fastq, int32_t threads)
{
    if (!g || !path || r_hi < r_lo) return -1;
    if (threads < 1) threads = g->threads;
    const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) return -1;
    int64_t at = 0;
    const int64_t RCH = 1 << 16, n_ch = (r_hi - r_lo + RCH - 1) / RCH;
    std::vector<char> qual((size_t)read_len, 'I');
    const int rc = write_parts(fd, at, n_ch, threads, (int64_t)threads * 4, [&](int64_t c, std::vector<char> &b) {
        const int64_t lo = r_lo + c * RCH, hi = std::min(r_hi, lo + RCH);
        std::vector<char> bases((size_t)((hi - lo) * read_len));
        phi_syn_reads(g, seed, lo, hi, read_len, sub_err, bases.data(), 1);
        b.resize((size_t)((hi - lo) * ((fastq ? 2 : 1) * (int64_t)read_len + 32)));
        char *p = b.data();
        for (int64_t r = lo; r < hi; r++) {
            *p++ = fastq ? '@' : '>'; *p++ = 'r'; p = put_u64(p, (uint64_t)r); *p++ = '\n';
            memcpy(p, bases.data() + (r - lo) * read_len, (size_t)read_len); p += read_len; *p++ = '\n';
            if (fastq) { *p++ = '+'; *p++ = '\n'; memcpy(p, qual.data(), (size_t)read_len); p += read_len; *p++ = '\n'; }
        }
        b.resize((size_t)(p - b.data()));
    });
    close(fd);
    return rc ? -1 : at;
}

}  // extern "C"
