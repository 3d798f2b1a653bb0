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
//
// How the work is laid out (a chromosome-scale GFA is >= 10 GB of text, 99 % of it W-lines):
//   * a plain file is MAPPED, never copied: lines and fields are slices of the mapping, the pages of a walk
//     are given back (MADV_DONTNEED) as soon as its vertices are resolved, so the anonymous memory of the
//     reader is the output arrays only.  A gzip file is inflated first (gz_source.h: on many threads when it
//     is block gzip) and then parsed the same way;
//   * lines are found and split into fields by all host threads over slices of the text cut at line ends;
//   * segment ids are first-seen order over S- and L-lines: ONE thread enters the names of the S-lines -- into a
//     table that maps names of the form <prefix><decimal number> (the names of every chopped pangenome graph:
//     "17", "s17") by direct indexing and any other name by open addressing --, then all threads resolve the
//     L-lines against it, checking that each name's S-line stands before the line (else: one thread, in order);
//   * the walks are resolved by all threads over PIECES of the W-lines (a 170-Mbp walk is one line of 60 MB):
//     a counting pass fixes where every piece writes, the second pass looks the names up and writes the
//     vertices straight into the final array.
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "gz_source.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/phi_host.h"

struct GfaState;                                       // the text, its slices and the name table, while the walks are still text
struct phi_graph {
    GfaState *state = nullptr;
    bool name_index_ok = false;                        // every name is <prefix><number> and every W-line stands behind all S-lines
    std::vector<int32_t> num2id;                       // the name table's direct index (number -> segment id, -1: none)
    std::string prefix;
    std::vector<char> name_arena;                      // segment names, NUL-terminated, back to back
    std::vector<int64_t> name_off;
    std::vector<std::string> hap_names;
    char *seq_concat = nullptr;                        // malloc'd (not zero-filled: every byte is written)
    int32_t *walk_vtx = nullptr;
    std::vector<int64_t> seq_off, adj_off, walk_off;
    std::vector<int32_t> adj, topo_rank;
    int32_t n_seg = 0;
    GfaState *retired = nullptr;                       // the reader's state after the walks are resolved: only its mapping of the file is left
    std::thread reaper;                                // frees the line tables once the walks are resolved
    void let_state_go();
    ~phi_graph();
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

static int host_threads()
{
    int nt = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("PHI_HOST_THREADS")) nt = atoi(e);
    return std::max(1, std::min(nt, 16));
}

template <class F> static void parallel_for(int64_t n, F fn)
{
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n));
    if (nt == 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<int64_t> next{0};
    auto work = [&]() { for (int64_t i; (i = next.fetch_add(1)) < n;) fn(i); };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

// The text of the file: a read-only mapping (plain file) or the inflated bytes (gzip).
struct Text {
    const char *p = nullptr;
    size_t n = 0;
    void *map = nullptr;
    size_t map_n = 0;
    std::vector<char> own;
    ~Text() { if (map) munmap(map, map_n); }
    // 0 ok, -1 cannot open / read, -2 gzip stream corrupt
    int load(const char *path)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return -1;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return -1; }
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            ::close(fd);
            GzSource gz;
            if (!gz.open(path, host_threads())) return -1;
            std::vector<char> blk;
            while (gz.next(blk)) {
                if (own.capacity() < own.size() + blk.size()) own.reserve(std::max(own.capacity() * 2, own.size() + blk.size()));
                own.insert(own.end(), blk.begin(), blk.end());
            }
            const bool ok = gz.ok();
            gz.close();
            if (!ok) return -2;
            p = own.data(); n = own.size();
            return 0;
        }
        if (!S_ISREG(st.st_mode) || st.st_size == 0) {
            // a pipe / character device / empty file: read what there is
            char tmp[1 << 16];
            for (ssize_t r; (r = read(fd, tmp, sizeof tmp)) > 0;) own.insert(own.end(), tmp, tmp + r);
            ::close(fd);
            p = own.data(); n = own.size();
            return 0;
        }
        map_n = (size_t)st.st_size;
        map = mmap(nullptr, map_n, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (map == MAP_FAILED) { map = nullptr; return -1; }
        (void)madvise(map, map_n, MADV_WILLNEED);
        p = (const char *)map; n = map_n;
        return 0;
    }
    // the pages wholly inside [lo, hi) are not needed again
    void done_with(const char *lo, const char *hi) const
    {
        if (!map) return;
        const uintptr_t pg = 4096, a = ((uintptr_t)lo + pg - 1) & ~(pg - 1), b = (uintptr_t)hi & ~(pg - 1);
        if (b > a) (void)madvise((void *)a, b - a, MADV_DONTNEED);
    }
};

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

// segment name -> id.  Names <prefix><canonical decimal number below a bound that grows with the table> are indexed
// directly (the prefix is that of the first name entered); every other name goes through open addressing over slices
// of the text, with a 32-bit tag of the hash beside the id so that a probe touches the text only to confirm.
class NameTable {
public:
    NameTable() { grow(1 << 12); }
    int32_t size() const { return (int32_t)keys_.size(); }
    // every name is a plain decimal number in the direct index: the W-line parser then reads the digits and looks the
    // number up in one pass over the text
    bool all_direct() const { return n_hashed_ == 0 && prefix_n_ == 0 && !keys_.empty(); }
    int32_t by_number(uint64_t num) const { return num < direct_.size() ? direct_[(size_t)num] : -1; }
    const std::vector<Slice> &keys() const { return keys_; }
    bool no_hashed_names() const { return n_hashed_ == 0; }
    std::string prefix() const { return std::string(prefix_, prefix_n_); }
    const std::vector<int32_t> &direct() const { return direct_; }
    // (the reader's all-at-once entry of names <prefix><number>: the prefix as add() takes it from the first name, then the
    //  finished key list and direct index)
    void set_prefix_of(const char *p, size_t n)
    {
        size_t l = 0;
        while (l < n && !(p[l] >= '0' && p[l] <= '9')) l++;
        prefix_n_ = 0;
        if (l <= sizeof prefix_) { memcpy(prefix_, p, l); prefix_n_ = l; }
    }
    void adopt(std::vector<Slice> &&keys, std::vector<int32_t> &&direct) { keys_ = std::move(keys); direct_ = std::move(direct); n_hashed_ = 0; }

    int32_t find(const char *p, size_t n) const
    {
        const int64_t num = number(p, n);
        if (num >= 0 && (size_t)num < direct_.size() && direct_[(size_t)num] >= 0) return direct_[(size_t)num];
        if (n_hashed_ == 0) return -1;
        const uint64_t h = hash(p, n);
        const uint32_t tag = (uint32_t)(h >> 32) | 1u;
        for (size_t i = h & mask_;; i = (i + 1) & mask_) {
            const Slot s = slot_[i];
            if (s.tag == 0) return -1;
            if (s.tag == tag) {
                const Slice &k = keys_[(size_t)s.id];
                if (k.n == n && memcmp(k.p, p, n) == 0) return s.id;
            }
        }
    }
    int32_t add(const char *p, size_t n)            // id of an existing or new name
    {
        const int32_t f = find(p, n);
        if (f >= 0) return f;
        const int32_t id = (int32_t)keys_.size();
        if (id == 0) {                               // the prefix of the direct index: the leading non-digits of the first name
            size_t l = 0;
            while (l < n && !(p[l] >= '0' && p[l] <= '9')) l++;
            if (l <= sizeof prefix_) { memcpy(prefix_, p, l); prefix_n_ = l; }
        }
        keys_.push_back(Slice{p, n});
        const int64_t num = number(p, n);
        const int64_t bound = std::max<int64_t>(1 << 20, 16 * (int64_t)keys_.size());
        if (num >= 0 && num < bound) {
            if ((size_t)num >= direct_.size()) direct_.resize((size_t)std::min<int64_t>(bound, std::max<int64_t>(num + 1, 2 * (int64_t)direct_.size())), -1);
            direct_[(size_t)num] = id;
            return id;
        }
        if ((n_hashed_ + 1) * 2 > slot_.size()) grow(slot_.size() * 2);
        insert(id);
        n_hashed_++;
        return id;
    }
private:
    struct Slot { int32_t id; uint32_t tag; };       // tag 0 = empty
public:
    // the number behind the prefix when the name is <prefix><decimal without leading zeros, at most 9 digits>, else -1
    int64_t number(const char *p, size_t n) const
    {
        if (n <= prefix_n_ || n - prefix_n_ > 9) return -1;
        if (prefix_n_ && memcmp(p, prefix_, prefix_n_) != 0) return -1;
        const char *d = p + prefix_n_;
        const size_t nd = n - prefix_n_;
        if (d[0] == '0' && nd > 1) return -1;
        int64_t v = 0;
        for (size_t i = 0; i < nd; i++) {
            const unsigned c = (unsigned)(d[i] - '0');
            if (c > 9) return -1;
            v = v * 10 + c;
        }
        return v;
    }
private:
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
    void insert(int32_t id)
    {
        const uint64_t h = hash(keys_[(size_t)id].p, keys_[(size_t)id].n);
        size_t i = h & mask_;
        while (slot_[i].tag != 0) i = (i + 1) & mask_;
        slot_[i] = Slot{id, (uint32_t)(h >> 32) | 1u};
    }
    void grow(size_t cap)
    {
        std::vector<Slot> old;
        old.swap(slot_);
        slot_.assign(cap, Slot{-1, 0});
        mask_ = cap - 1;
        for (const Slot &s : old) if (s.tag) insert(s.id);
    }
    std::vector<Slot> slot_;
    std::vector<Slice> keys_;
    std::vector<int32_t> direct_;
    size_t mask_ = 0, n_hashed_ = 0;
    char prefix_[16];
    size_t prefix_n_ = 0;
};

// fields of a line: up to cap tab-separated slices
static int split_tabs(const char *p, const char *e, Slice *f, int cap)
{
    int n = 0;
    while (n < cap) {
        // (the last field asked for is the rest of the line, not searched: for a W-line that is the walk, megabytes long)
        const char *t = n == cap - 1 ? nullptr : (const char *)memchr(p, '\t', (size_t)(e - p));
        if (!t) { f[n++] = Slice{p, (size_t)(e - p)}; break; }
        f[n++] = Slice{p, (size_t)(t - p)};
        p = t + 1;
    }
    return n;
}

namespace {
struct Rec {                                          // an S-line (a = name, b = sequence or null) or an L-line (a, b = names)
    const char *a, *b;
    uint32_t an, bn;
    char type, ov, ow;
};
struct WRec {
    Slice sample, text;
    int hap;
    size_t before;                                    // index of the next S/L record of its slice: what the line can name
    int32_t n_known = 0;
};
struct SliceOut { std::vector<Rec> recs; std::vector<WRec> walks; size_t n_S = 0, n_L = 0; };

void scan_slice(const char *p, const char *end, SliceOut &o)
{
    Slice f[8];
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *next = nl ? nl + 1 : end;
        if (le > p && le[-1] == '\r') le--;
        if (le - p >= 3 && p[1] == '\t' && (p[0] == 'S' || p[0] == 'L' || p[0] == 'W')) {
            const char t = p[0];
            const int nf = split_tabs(p, le, f, t == 'W' ? 7 : 6);
            if (t == 'S' && nf >= 3) {
                const bool has = f[2].n > 0 && f[2].p[0] != '*';
                o.recs.push_back(Rec{f[1].p, has ? f[2].p : nullptr, (uint32_t)f[1].n, has ? (uint32_t)f[2].n : 0u, 'S', 0, 0});
                o.n_S++;
            } else if (t == 'L' && nf >= 5) {
                if (f[2].n == 1 && f[4].n == 1) {
                    const char ov = f[2].p[0], ow = f[4].p[0];
                    if ((ov == '+' || ov == '-') && (ow == '+' || ow == '-')) {
                        o.recs.push_back(Rec{f[1].p, f[3].p, (uint32_t)f[1].n, (uint32_t)f[3].n, 'L', ov, ow});
                        o.n_L++;
                    }
                }
            } else if (t == 'W' && nf >= 7) {
                WRec w;
                w.sample = f[1];
                w.hap = atoi(std::string(f[2].p, f[2].n).c_str());
                w.text = f[6];                                        // (optional tags may follow: cut off below, where the text is read anyway)
                w.before = o.recs.size();
                o.walks.push_back(w);
            }
        }
        p = next;
    }
}

inline bool is_step(char c) { return c == '>' || c == '<'; }

struct Piece { int32_t walk; const char *lo, *hi; int64_t n = 0, out = 0, dropped = 0; bool any_rev = false, has_tab = false; };
}  // namespace

struct GfaState {
    Text text;
    std::vector<SliceOut> so;                          // (the W-line records live in here)
    NameTable table;
    std::vector<WRec *> walks;                         // in file order
    int32_t n_seg = 0;
};
// The reader's tables are freed on a thread of their own once the walks are resolved; the MAPPING of the file stays until the
// graph itself goes: unmapping the 11 GB of a chromosome-scale GFA takes 0.2-0.5 s (measured at config 5, as the "walks" stage
// of the command line) and holds the address-space lock against every allocation and page fault of the other threads while it
// does.  The pages are the file's own, clean and shared: they cost nothing to keep.
void phi_graph::let_state_go()
{
    GfaState *st = state;
    state = nullptr;
    if (!st) return;
    if (reaper.joinable()) reaper.join();
    delete retired;
    retired = st;
    reaper = std::thread([st]() {
        std::vector<SliceOut>().swap(st->so);
        std::vector<WRec *>().swap(st->walks);
        st->table = NameTable();
        if (!st->text.map) { std::vector<char>().swap(st->text.own); st->text.p = nullptr; st->text.n = 0; }
    });
}
phi_graph::~phi_graph() { if (reaper.joinable()) reaper.join(); free(seq_concat); free(walk_vtx); delete state; delete retired; }

static int resolve_walks(GfaState &st, phi_graph *g, StageTimer &tm, char *err, int err_cap);

static int gfa_read_impl(const char *path, phi_graph **out, bool defer, phi_walk_text_fn on_text, void *user, char *err, int err_cap)
{
    if (!path || !out) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "null argument");
    *out = nullptr;
    StageTimer tm;
    GfaState *stp = new GfaState();
    struct StateGuard { GfaState *&p; ~StateGuard() { delete p; } } state_guard{stp};      // (handed to the graph when the walks stay text)
    Text &text = stp->text;
    if (const int lr = text.load(path))
        return fail(err, err_cap, PHI_HOST_ERR_IO, lr == -2 ? "gzip stream corrupt in the GFA file %s" : "failed to load the GFA file %s", path);
    tm.lap(text.map ? "map" : "read / inflate");

    // ---- lines and fields, on all threads over slices of the text cut at line ends
    const char *const t0 = text.p, *const tend = text.p + text.n;
    // Cuts: a pangenome GFA is a few hundred MB of short S- and L-lines (millions of records) followed by W-lines of tens of MB
    // each.  Equal slices by bytes would leave all the records to the two or three threads whose slices hold the short lines, so
    // the text is cut FINE (a few MB) wherever a line ends within 64 KB of the nominal cut -- everywhere among short lines,
    // almost nowhere inside the W-lines -- and every 32nd cut is COARSE: moved on to the next line end however far that is,
    // so that the long lines are still shared out.  All cuts are looked for at once, by all threads.
    const size_t fine_bytes = getenv("PHI_GFA_SLICE") ? std::max<size_t>(64, (size_t)atoll(getenv("PHI_GFA_SLICE"))) : ((size_t)4 << 20);
    const size_t n_nominal = std::max<size_t>(1, std::min<size_t>((size_t)1 << 16, text.n / fine_bytes + 1));
    const size_t coarse_every = std::max<size_t>(1, n_nominal / ((size_t)host_threads() * 4));
    std::vector<const char *> cand(n_nominal, nullptr);
    parallel_for((int64_t)n_nominal - 1, [&](int64_t j) {
        const size_t i = (size_t)j + 1;
        const char *q = t0 + text.n / n_nominal * i;
        if (q[-1] == '\n') { cand[i] = q; return; }
        const size_t room = (size_t)(tend - q), window = i % coarse_every == 0 ? room : std::min<size_t>(room, (size_t)64 << 10);
        const char *nl = (const char *)memchr(q, '\n', window);
        if (nl && nl + 1 < tend) cand[i] = nl + 1;
    });
    std::vector<const char *> cut{t0};
    for (size_t i = 1; i < n_nominal; i++) if (cand[i] && cand[i] > cut.back()) cut.push_back(cand[i]);
    const int n_slices = (int)cut.size();
    cut.push_back(tend);
    std::vector<SliceOut> &so = stp->so;
    so.resize((size_t)n_slices);
    const bool populate = text.map && getenv("PHI_GFA_POPULATE") && atoi(getenv("PHI_GFA_POPULATE")) != 0;   // (experiment: page tables filled slice by slice by the thread about to read it)
    parallel_for(n_slices, [&](int64_t i) {
        if (populate) {
            const uintptr_t a = (uintptr_t)cut[(size_t)i] & ~(uintptr_t)4095, b = ((uintptr_t)cut[(size_t)i + 1] + 4095) & ~(uintptr_t)4095;
            (void)madvise((void *)a, b - a, 22 /* MADV_POPULATE_READ */);
        }
        scan_slice(cut[(size_t)i], cut[(size_t)i + 1], so[(size_t)i]);
    });
    tm.lap("lines + fields (threads)");
    // the walk fields are known now, long before their names can be resolved: a caller that resolves them on the device
    // starts sending the text there (a thread of its own, joined before this call returns)
    std::vector<phi_host_walk_text> wtexts;
    std::thread text_thread;
    struct TextJoin { std::thread &t; ~TextJoin() { if (t.joinable()) t.join(); } } text_join{text_thread};
    if (defer && on_text) {
        for (SliceOut &s_ : so) for (WRec &w_ : s_.walks) wtexts.push_back(phi_host_walk_text{w_.text.p, (int64_t)w_.text.n});
        text_thread = std::thread([&]() { on_text(user, wtexts.data(), (int32_t)wtexts.size()); });
    }

    // ---- segment ids in first-seen order over S- and L-lines.  In every graph a tool writes, an L-line names segments whose
    //      S-lines stand before it, so the ids are the order of the S-lines: one thread enters the S-line names (and fixes what
    //      each W-line may name), then all threads resolve the L-lines against the finished table, checking for each name that
    //      its S-line does come first.  A file where that does not hold (an L-line that introduces a segment) is done again
    //      by one thread, record by record.
    NameTable &table = stp->table;
    std::vector<Slice> seqs;                                         // per segment; n = 0: no sequence
    std::vector<std::pair<uint32_t, uint32_t>> arcs;                 // oriented vertices v = seg<<1 | strand
    std::vector<WRec *> &walks = stp->walks;
    size_t n_rec = 0;
    std::vector<size_t> rec_base((size_t)n_slices + 1, 0);
    for (int i = 0; i < n_slices; i++) { n_rec += so[(size_t)i].recs.size(); rec_base[(size_t)i + 1] = n_rec; }
    auto add_seg = [&](const char *p, uint32_t n) {
        const int32_t id = table.add(p, n);
        if ((size_t)id == seqs.size()) seqs.push_back(Slice{nullptr, 0});
        return id;
    };
    auto one_thread = [&]() {
        table = NameTable();
        seqs.clear(); arcs.clear(); walks.clear();
        seqs.reserve(n_rec); arcs.reserve(n_rec);
        for (SliceOut &s : so) {
            size_t wi = 0;
            for (size_t i = 0; i <= s.recs.size(); i++) {
                while (wi < s.walks.size() && s.walks[wi].before == i) {
                    s.walks[wi].n_known = table.size();              // names are resolved against the segments seen so far
                    walks.push_back(&s.walks[wi++]);
                }
                if (i == s.recs.size()) break;
                const Rec &r = s.recs[i];
                if (r.type == 'S') {
                    const int32_t id = add_seg(r.a, r.an);
                    seqs[(size_t)id] = Slice{r.b, r.bn};              // a later S-line of the same name replaces the sequence
                } else {
                    const uint32_t v = (uint32_t)add_seg(r.a, r.an) << 1 | (r.ov != '+');
                    const uint32_t w = (uint32_t)add_seg(r.b, r.bn) << 1 | (r.ow != '+');
                    arcs.emplace_back(v, w);
                }
            }
        }
    };
    {
        std::vector<size_t> first_rec;                               // per segment: the record that introduced it
        // S-lines and W-lines.  Every tool names its segments <prefix><number>, each once: then the ids are simply the order of
        // the S-lines and all threads enter them at once (a compare-and-swap on the number's slot finds a name given twice).
        // Anything else -- a name of another form, a name twice, a number far beyond the count -- goes in file order on one thread.
        std::vector<size_t> s_base((size_t)n_slices + 1, 0), l_base((size_t)n_slices + 1, 0);
        for (int i = 0; i < n_slices; i++) { s_base[(size_t)i + 1] = s_base[(size_t)i] + so[(size_t)i].n_S; l_base[(size_t)i + 1] = l_base[(size_t)i] + so[(size_t)i].n_L; }
        const size_t n_S = s_base[(size_t)n_slices], n_L = l_base[(size_t)n_slices];
        auto bulk = [&]() -> bool {
            if (n_S == 0 || n_S >= (size_t)INT32_MAX || getenv("PHI_GFA_NAMES_SERIAL")) return false;
            const Rec *first = nullptr;
            for (int i = 0; i < n_slices && !first; i++) for (const Rec &r : so[(size_t)i].recs) if (r.type == 'S') { first = &r; break; }
            table.set_prefix_of(first->a, first->an);
            std::vector<int64_t> mx((size_t)n_slices, -1);
            std::atomic<int> bad{0};
            parallel_for(n_slices, [&](int64_t si) {
                size_t k = s_base[(size_t)si];
                int64_t m = -1;
                for (const Rec &r : so[(size_t)si].recs) {
                    if (r.type != 'S') continue;
                    const int64_t num = table.number(r.a, r.an);
                    if (num < 0 || num >= std::max<int64_t>(1 << 20, 16 * (int64_t)(k + 1))) { bad.store(1); return; }
                    m = std::max(m, num);
                    k++;
                }
                mx[(size_t)si] = m;
            });
            if (bad.load()) return false;
            std::vector<int32_t> direct((size_t)(*std::max_element(mx.begin(), mx.end()) + 1), -1);
            std::vector<Slice> keys(n_S);
            seqs.assign(n_S, Slice{nullptr, 0});
            first_rec.assign(n_S, 0);
            parallel_for(n_slices, [&](int64_t si) {
                SliceOut &s_ = so[(size_t)si];
                size_t k = s_base[(size_t)si], wi = 0;
                for (size_t i = 0; i <= s_.recs.size(); i++) {
                    while (wi < s_.walks.size() && s_.walks[wi].before == i) s_.walks[wi++].n_known = (int32_t)k;
                    if (i == s_.recs.size()) break;
                    const Rec &r = s_.recs[i];
                    if (r.type != 'S') continue;
                    int32_t none = -1;
                    if (!__atomic_compare_exchange_n(&direct[(size_t)table.number(r.a, r.an)], &none, (int32_t)k, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { bad.store(1); return; }
                    keys[k] = Slice{r.a, r.an};
                    seqs[k] = Slice{r.b, r.bn};
                    first_rec[k] = rec_base[(size_t)si] + i;
                    k++;
                }
            });
            if (bad.load()) return false;
            table.adopt(std::move(keys), std::move(direct));
            for (SliceOut &s_ : so) for (WRec &w_ : s_.walks) walks.push_back(&w_);
            return true;
        };
        if (!bulk()) {
            table = NameTable();
            seqs.clear(); walks.clear(); first_rec.clear();
            seqs.reserve(n_rec);
            for (int si = 0; si < n_slices; si++) {
                SliceOut &s = so[(size_t)si];
                size_t wi = 0;
                for (size_t i = 0; i <= s.recs.size(); i++) {
                    while (wi < s.walks.size() && s.walks[wi].before == i) {
                        s.walks[wi].n_known = table.size();
                        walks.push_back(&s.walks[wi++]);
                    }
                    if (i == s.recs.size()) break;
                    const Rec &r = s.recs[i];
                    if (r.type != 'S') continue;
                    const int32_t id = add_seg(r.a, r.an);
                    if ((size_t)id == first_rec.size()) first_rec.push_back(rec_base[(size_t)si] + i);
                    seqs[(size_t)id] = Slice{r.b, r.bn};
                }
            }
        }
        tm.lap("  S-line names");
        // L-lines, all threads: both names known, and known BEFORE the line; every line writes its arc where it stands
        arcs.resize(n_L);
        std::atomic<int> hazard{0};
        parallel_for(n_slices, [&](int64_t si) {
            const SliceOut &s = so[(size_t)si];
            size_t k = l_base[(size_t)si];
            for (size_t i = 0; i < s.recs.size(); i++) {
                const Rec &r = s.recs[i];
                if (r.type != 'L') continue;
                const int32_t a = table.find(r.a, r.an), b2 = table.find(r.b, r.bn);
                const size_t at = rec_base[(size_t)si] + i;
                if (a < 0 || b2 < 0 || first_rec[(size_t)a] > at || first_rec[(size_t)b2] > at) { hazard.store(1); return; }
                arcs[k++] = std::make_pair((uint32_t)a << 1 | (r.ov != '+'), (uint32_t)b2 << 1 | (r.ow != '+'));
            }
        });
        tm.lap("  L-line names (threads)");
        if (hazard.load()) one_thread();
    }
    tm.lap("  arcs together");
    // (the records -- 19 M of them, 0.76 GB, at chromosome scale -- are freed on a thread of their own while this one waits for the side
    //  threads below: 50 ms here; left to the thread that lets the state go they cost phi_set_graph, which runs by then, as much)
    std::thread free_recs([&so]() { for (SliceOut &s : so) std::vector<Rec>().swap(s.recs); });
    struct RecsJoin { std::thread &t; ~RecsJoin() { if (t.joinable()) t.join(); } } recs_join{free_recs};
    const int32_t n_seg = table.size();
    const int64_t n_walks = (int64_t)walks.size();
    stp->n_seg = n_seg;
    tm.lap("segment ids, links");

    // ---- nothing of the walks is needed for the sequences, the names, the adjacency and the topological order: one thread
    //      makes them while the others resolve the W-lines (99 % of a pangenome GFA's bytes)
    phi_graph *g = new phi_graph();
    int side_rc = 0;                                               // 1: out of memory, 2: cycle, 3: a link onto a reverse strand
    int32_t side_sorted = 0;
    int copy_rc = 0;
    bool side_joined = false;
    // (side thread 2: the sequences and the names, copied together -- by all threads when the walks stay text and nobody else
    //  needs them)
    std::thread side_copy([&]() {
        StageTimer tc;
        g->n_seg = n_seg;
        const std::vector<Slice> &names = table.keys();
        g->name_off.assign((size_t)n_seg + 1, 0);
        for (int32_t i = 0; i < n_seg; i++) g->name_off[(size_t)i + 1] = g->name_off[(size_t)i] + (int64_t)names[(size_t)i].n + 1;
        g->name_arena.resize((size_t)g->name_off[(size_t)n_seg]);
        g->seq_off.assign((size_t)n_seg + 1, 0);
        for (int32_t i = 0; i < n_seg; i++) g->seq_off[(size_t)i + 1] = g->seq_off[(size_t)i] + (int64_t)seqs[(size_t)i].n;
        g->seq_concat = (char *)malloc(std::max<size_t>(1, (size_t)g->seq_off[(size_t)n_seg]));
        if (!g->seq_concat) { copy_rc = 1; return; }
        auto copy = [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                if (seqs[(size_t)i].n) memcpy(g->seq_concat + g->seq_off[(size_t)i], seqs[(size_t)i].p, seqs[(size_t)i].n);
                char *d = g->name_arena.data() + g->name_off[(size_t)i];
                memcpy(d, names[(size_t)i].p, names[(size_t)i].n);
                d[names[(size_t)i].n] = '\0';
            }
        };
        const int64_t CH = 1 << 16, n_ch = ((int64_t)n_seg + CH - 1) / CH;
        if (defer) parallel_for(n_ch, [&](int64_t c) { copy(c * CH, std::min<int64_t>(n_seg, (c + 1) * CH)); });
        else copy(0, n_seg);                                         // (one thread: the others are on the W-lines)
        for (int64_t w = 0; w < n_walks; w++)
            g->hap_names.push_back(std::string(walks[(size_t)w]->sample.p, walks[(size_t)w]->sample.n) + "." + std::to_string(walks[(size_t)w]->hap));
        tc.lap("  [copy thread] sequences, names");
    });
    struct CopyJoiner { std::thread &t; ~CopyJoiner() { if (t.joinable()) t.join(); } } copy_joiner{side_copy};
    std::thread side([&]() {
        StageTimer ts;
        // arcs: those touching a segment without sequence are dropped; an arc v -> w and its complement w' -> v' give the
        // forward-strand adjacency (target orientation dropped): u -> w when v = u+, and w -> v when w is a reverse strand
        {
            // A link onto the REVERSE strand of its target (L a + b -, L a - b -) is refused: the forward arc it implies is the
            // COMPLEMENT b+ -> a(-/+), which the reference appends behind its sorted arc array and whose arc index shows it only
            // when the appended arcs happen to break the array's sort order (gfa-base.cpp:269-303 re-sorts on a vertex-count
            // test that never fires) -- the reference's own adjacency for such a file depends on where the line stands.  Rounds
            // 1-3 kept the arc silently; a graph whose meaning the reference itself does not fix is an error here.
            // (L a - b +: neither the arc nor its complement leaves a forward strand: nothing to add, as in the reference.)
            // (all threads, in pieces of the arc list: the targets of a vertex arrive in any order and are sorted below)
            const int64_t n_arcs = (int64_t)arcs.size(), ACH = 1 << 16, n_ach = (n_arcs + ACH - 1) / ACH;
            std::atomic<int64_t> first_rev{INT64_MAX};
            std::vector<int64_t> cnt((size_t)n_seg + 1, 0);
            auto each = [&](int64_t lo, int64_t hi, auto fn) {
                for (int64_t i = lo; i < hi; i++) {
                    const uint32_t v = arcs[(size_t)i].first, w = arcs[(size_t)i].second;
                    if (seqs[v >> 1].n == 0 || seqs[w >> 1].n == 0) continue;
                    if (!(v & 1)) fn(v >> 1, w >> 1);
                    if (w & 1) fn(w >> 1, v >> 1);
                }
            };
            parallel_for(n_ach, [&](int64_t c) {
                const int64_t lo = c * ACH, hi = std::min(n_arcs, lo + ACH);
                for (int64_t i = lo; i < hi; i++) {
                    const auto &a = arcs[(size_t)i];
                    if ((a.second & 1) && seqs[a.first >> 1].n && seqs[a.second >> 1].n) {
                        int64_t cur = first_rev.load();
                        while (i < cur && !first_rev.compare_exchange_weak(cur, i)) {}
                        return;
                    }
                }
                each(lo, hi, [&](uint32_t u, uint32_t) { __atomic_fetch_add(&cnt[(size_t)u + 1], 1, __ATOMIC_RELAXED); });
            });
            if (first_rev.load() != INT64_MAX) { side_rc = 3; side_sorted = (int32_t)(arcs[(size_t)first_rev.load()].second >> 1); return; }
            for (int32_t i = 0; i < n_seg; i++) cnt[(size_t)i + 1] += cnt[(size_t)i];
            std::vector<int32_t> tgt((size_t)cnt[(size_t)n_seg]);
            {
                std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
                parallel_for(n_ach, [&](int64_t c) {
                    each(c * ACH, std::min(n_arcs, (c + 1) * ACH), [&](uint32_t u, uint32_t x) { tgt[(size_t)__atomic_fetch_add(&cur[u], 1, __ATOMIC_RELAXED)] = (int32_t)x; });
                });
            }
            // duplicate links, and oriented targets that collapse onto one segment, are merged
            const int64_t VCH = 1 << 16, n_vch = ((int64_t)n_seg + VCH - 1) / VCH;
            std::vector<int32_t> deg((size_t)n_seg, 0);
            parallel_for(n_vch, [&](int64_t c) {
                for (int64_t u = c * VCH, ue = std::min<int64_t>(n_seg, u + VCH); u < ue; u++) {
                    int32_t *b = tgt.data() + cnt[(size_t)u], *e = tgt.data() + cnt[(size_t)u + 1];
                    if (e - b > 1) { std::sort(b, e); e = std::unique(b, e); }
                    deg[(size_t)u] = (int32_t)(e - b);
                }
            });
            g->adj_off.assign((size_t)n_seg + 1, 0);
            for (int32_t u = 0; u < n_seg; u++) g->adj_off[(size_t)u + 1] = g->adj_off[(size_t)u] + deg[(size_t)u];
            g->adj.resize((size_t)g->adj_off[(size_t)n_seg]);
            parallel_for(n_vch, [&](int64_t c) {
                for (int64_t u = c * VCH, ue = std::min<int64_t>(n_seg, u + VCH); u < ue; u++)
                    if (deg[(size_t)u]) memcpy(g->adj.data() + g->adj_off[(size_t)u], tgt.data() + cnt[(size_t)u], (size_t)deg[(size_t)u] * 4);
            });
        }
        ts.lap("  [side thread] adjacency");
        // Kahn's algorithm, FIFO
        {
            std::vector<int32_t> indeg((size_t)n_seg, 0), q((size_t)n_seg);
            for (int32_t v : g->adj) indeg[(size_t)v]++;
            int32_t head = 0, tail = 0;
            for (int32_t i = 0; i < n_seg; i++) if (indeg[(size_t)i] == 0) q[(size_t)tail++] = i;
            g->topo_rank.assign((size_t)n_seg, 0);
            while (head < tail) {
                const int32_t u = q[(size_t)head];
                g->topo_rank[(size_t)u] = head++;
                for (int64_t x = g->adj_off[(size_t)u]; x < g->adj_off[(size_t)u + 1]; x++)
                    if (--indeg[(size_t)g->adj[(size_t)x]] == 0) q[(size_t)tail++] = g->adj[(size_t)x];
            }
            if (head != n_seg) { side_rc = 2; side_sorted = head; return; }
        }
        ts.lap("  [side thread] topological order");
    });
    struct Joiner { std::thread &t; bool &done; ~Joiner() { if (!done && t.joinable()) t.join(); } } joiner{side, side_joined};

    // ---- the walks' vertices: resolved here by all threads (resolve_walks below) -- or left as text for the caller, who
    //      resolves them on the device (phi_gfa_read_deferred) and comes back for the host path only if the text is not of
    //      the kind the device takes
    g->prefix = table.prefix();
    g->name_index_ok = table.no_hashed_names();
    for (int64_t wi = 0; wi < n_walks && g->name_index_ok; wi++) g->name_index_ok = walks[(size_t)wi]->n_known == n_seg;
    if (defer) {
        if (g->name_index_ok) g->num2id = table.direct();
    } else if (const int wrc = resolve_walks(*stp, g, tm, err, err_cap)) {
        side.join();
        side_copy.join();
        delete g;
        return wrc;
    }

    // ---- what the side thread made meanwhile
    side.join();
    side_joined = true;
    side_copy.join();
    if (copy_rc) side_rc = 1;
    if (side_rc == 1) { delete g; return fail(err, err_cap, PHI_HOST_ERR_INVALID, "out of memory"); }
    if (side_rc == 3) {
        const std::string nm(g->name_arena.data() + g->name_off[(size_t)side_sorted]);
        delete g;
        return fail(err, err_cap, PHI_HOST_ERR_UNSUPPORTED, "a link onto the reverse strand of segment %s: the reference's arc index shows the forward arc such a link implies "
                    "only sometimes (gfa-base.cpp:269-303); write the link from the forward strand", nm.c_str());
    }
    if (side_rc == 2) {
        const int code = fail(err, err_cap, PHI_HOST_ERR_CYCLE, "graph is not acyclic: %d of %d vertices sorted", side_sorted, n_seg);
        delete g;
        return code;
    }
    tm.lap("wait for arrays + topological order (side thread)");
    if (free_recs.joinable()) free_recs.join();                // (before the state changes hands)
    if (defer) {
        if (text_thread.joinable()) text_thread.join();
        tm.lap("wait for the walk text's consumer");
        g->state = stp;
        stp = nullptr;
    } else {
        g->state = stp;
        stp = nullptr;
        g->let_state_go();
    }
    *out = g;
    return PHI_HOST_OK;
}

// The walks' vertices from the text of the W-lines, on all host threads: the reference's rules (names resolved against the
// segments seen so far, unknown names left out, walks flipped by majority strand, a reverse vertex left over is an error).
static int resolve_walks(GfaState &st, phi_graph *g, StageTimer &tm, char *err, int err_cap)
{
    Text &text = st.text;
    NameTable &table = st.table;
    std::vector<WRec *> &walks = st.walks;
    const int32_t n_seg = st.n_seg;
    const int64_t n_walks = (int64_t)walks.size();
    // ---- the walks' vertices: pieces of the W-lines, cut at steps, on all threads
    std::vector<Piece> pieces;
    const size_t PIECE = getenv("PHI_GFA_PIECE") ? std::max<size_t>(16, (size_t)atoll(getenv("PHI_GFA_PIECE"))) : ((size_t)1 << 20);
    // (twice only when a W-line carries tags behind its walk: the pass that counts the steps reads every byte of the
    //  walks anyway and looks for a tab while it does -- a search of its own was one more pass over 10 GB of text at
    //  chromosome scale; a walk with a tab is cut off there and the pieces are made again)
    for (int round = 0; round < 2; round++) {
        pieces.clear();
        for (int64_t wi = 0; wi < n_walks; wi++) {
            const char *s = walks[(size_t)wi]->text.p, *const e = s + walks[(size_t)wi]->text.n;
            const char *lo = s;
            while (lo < e) {
                const char *hi = (size_t)(e - lo) > PIECE + PIECE / 2 ? lo + PIECE : e;
                while (hi < e && !is_step(*hi)) hi++;                // a name never straddles two pieces
                pieces.push_back(Piece{(int32_t)wi, lo, hi});
                lo = hi;
            }
            if (s == e) pieces.push_back(Piece{(int32_t)wi, s, e});
        }
        parallel_for((int64_t)pieces.size(), [&](int64_t i) {
            Piece &pc = pieces[(size_t)i];
            int64_t n = 0;
            int tab = 0;
            for (const char *q = pc.lo; q < pc.hi; q++) { n += is_step(*q); tab |= *q == '\t'; }
            pc.n = n;
            pc.has_tab = tab != 0;
        });
        bool any_tab = false;
        for (const Piece &pc : pieces)
            if (pc.has_tab) {
                any_tab = true;
                WRec &w = *walks[(size_t)pc.walk];
                if (const char *tab = (const char *)memchr(w.text.p, '\t', w.text.n)) w.text.n = (size_t)(tab - w.text.p);
            }
        if (!any_tab) break;
    }
    g->walk_off.assign((size_t)n_walks + 1, 0);
    {
        int64_t o = 0;
        size_t pi = 0;
        for (int64_t wi = 0; wi < n_walks; wi++) {
            g->walk_off[(size_t)wi] = o;
            for (; pi < pieces.size() && pieces[pi].walk == wi; pi++) { pieces[pi].out = o; o += pieces[pi].n; }
        }
        g->walk_off[(size_t)n_walks] = o;
    }
    const int64_t n_entries = g->walk_off[(size_t)n_walks];
    uint32_t *wv = (uint32_t *)malloc(std::max<size_t>(1, (size_t)n_entries) * 4);
    if (!wv) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "out of memory for %lld walk entries", (long long)n_entries);
    g->walk_vtx = (int32_t *)wv;
    const uint32_t DROP = 0xFFFFFFFFu;
    parallel_for((int64_t)pieces.size(), [&](int64_t i) {
        Piece &pc = pieces[(size_t)i];
        const int32_t n_known = walks[(size_t)pc.walk]->n_known;
        const char *const e = walks[(size_t)pc.walk]->text.p + walks[(size_t)pc.walk]->text.n;
        uint32_t *dst = wv + pc.out;
        const char *q = pc.lo;
        while (q < pc.hi && !is_step(*q)) q++;                       // (bytes before the first step of a walk)
        const bool numeric = table.all_direct();
        while (q < pc.hi) {
            const char *nm = q + 1, *r = nm;
            int32_t id;
            uint64_t num = 0;
            if (numeric) while (r < e && (unsigned)(*r - '0') <= 9u && r - nm < 10) num = num * 10 + (unsigned)(*r++ - '0');
            if (numeric && r > nm && (r == e || is_step(*r)) && (nm[0] != '0' || r - nm == 1)) id = table.by_number(num);
            else {
                while (r < e && !is_step(*r)) r++;
                id = table.find(nm, (size_t)(r - nm));
            }
            if (id >= 0 && id < n_known) { *dst++ = (uint32_t)id << 1 | (*q == '<'); pc.any_rev |= *q == '<'; }
            else { *dst++ = DROP; pc.dropped++; }
            q = r;
        }
        text.done_with(pc.lo, pc.hi);
    });
    bool any_rev = false;
    int64_t dropped = 0;
    for (const Piece &pc : pieces) { any_rev |= pc.any_rev; dropped += pc.dropped; }
    if (dropped) {                                                   // steps naming no known segment are left out
        int64_t o = 0;
        for (int64_t wi = 0; wi < n_walks; wi++) {
            const int64_t lo = g->walk_off[(size_t)wi], hi = g->walk_off[(size_t)wi + 1];
            g->walk_off[(size_t)wi] = o;
            for (int64_t x = lo; x < hi; x++) if (wv[x] != DROP) wv[o++] = wv[x];
        }
        g->walk_off[(size_t)n_walks] = o;
    }
    tm.lap("W lines (threads)");

    // gfa_walk_flip: the first walk to touch a segment fixes its strand; a walk that disagrees
    // with the majority of its vertices is reverse-complemented
    if (any_rev) {                                     // all-forward walks agree with every first touch: nothing to flip
        std::vector<int8_t> strand((size_t)n_seg, 0);
        for (int64_t x = 0, n = g->walk_off[(size_t)n_walks]; x < n; x++)
            if (strand[wv[x] >> 1] == 0) strand[wv[x] >> 1] = (wv[x] & 1) ? -1 : 1;
        parallel_for(n_walks, [&](int64_t wi) {
            uint32_t *b = wv + g->walk_off[(size_t)wi], *e = wv + g->walk_off[(size_t)wi + 1];
            int64_t agree = 0;
            for (uint32_t *v = b; v < e; v++) agree += (((*v & 1) ? -1 : 1) == strand[*v >> 1]);
            if (agree >= (e - b) - agree) return;
            std::reverse(b, e);
            for (uint32_t *v = b; v < e; v++) *v ^= 1;
        });
    }
    // oriented vertices -> segment ids; a reverse-strand vertex left in a walk is an error (ILP_index.cpp:104-107)
    {
        const int64_t n = g->walk_off[(size_t)n_walks], CH = (int64_t)1 << 20, n_ch = (n + CH - 1) / CH;
        std::atomic<int64_t> first_bad{INT64_MAX};
        parallel_for(n_ch, [&](int64_t c) {
            const int64_t lo = c * CH, hi = std::min(n, lo + CH);
            for (int64_t x = lo; x < hi; x++) {
                if (wv[x] & 1) {
                    int64_t cur = first_bad.load();
                    while (x < cur && !first_bad.compare_exchange_weak(cur, x)) {}
                    return;
                }
                wv[x] >>= 1;
            }
        });
        if (first_bad.load() != INT64_MAX) {
            const int64_t x = first_bad.load();
            const int64_t w = (int64_t)(std::upper_bound(g->walk_off.begin(), g->walk_off.end(), x) - g->walk_off.begin()) - 1;
            // (entries before x in other chunks may already be converted; x itself is not)
            return fail(err, err_cap, PHI_HOST_ERR_WALK, "Error: Walk %d has reverse strand vertices %u", (int)w, wv[x]);
        }
    }
    tm.lap("walk flips, vertex ids");

    return PHI_HOST_OK;
}

extern "C" {

int phi_gfa_read(const char *path, phi_graph **out, char *err, int err_cap) { return gfa_read_impl(path, out, false, nullptr, nullptr, err, err_cap); }

/* The same, but the walks stay TEXT: on_text (optional) is called on a thread of its own as soon as the walk fields are known
 * -- long before the names are --, and the graph comes back without walk_off / walk_vtx.  The caller resolves the walks on the
 * device (include/phi_amd.h phi_walk_text_*) with phi_graph_name_index and reports their offsets with phi_graph_set_walk_off,
 * or has them resolved here after all: phi_graph_resolve_walks. */
int phi_gfa_read_deferred(const char *path, phi_graph **out, phi_walk_text_fn on_text, void *user, char *err, int err_cap)
{
    return gfa_read_impl(path, out, true, on_text, user, err, err_cap);
}
int phi_graph_walks_deferred(const phi_graph *g) { return g && g->state != nullptr; }
int phi_graph_walk_texts(const phi_graph *g, phi_host_walk_text *out, int32_t cap)
{
    if (!g || !g->state) return PHI_HOST_ERR_INVALID;
    const int32_t n = (int32_t)g->state->walks.size();
    if (out) for (int32_t i = 0; i < n && i < cap; i++) out[i] = phi_host_walk_text{g->state->walks[(size_t)i]->text.p, (int64_t)g->state->walks[(size_t)i]->text.n};
    return n;
}
int phi_graph_name_index(const phi_graph *g, const char **prefix, int32_t *prefix_n, const int32_t **num2id, int64_t *n_num)
{
    if (!g || !g->state || !g->name_index_ok) return PHI_HOST_ERR_UNSUPPORTED;
    if (prefix) *prefix = g->prefix.data();
    if (prefix_n) *prefix_n = (int32_t)g->prefix.size();
    if (num2id) *num2id = g->num2id.data();
    if (n_num) *n_num = (int64_t)g->num2id.size();
    return PHI_HOST_OK;
}
int phi_graph_resolve_walks(phi_graph *g, char *err, int err_cap)
{
    if (!g || !g->state) return fail(err, err_cap, PHI_HOST_ERR_INVALID, "the walks of this graph are resolved already");
    StageTimer tm;
    const int rc = resolve_walks(*g->state, g, tm, err, err_cap);
    g->let_state_go();
    return rc;
}
int phi_graph_set_walk_off(phi_graph *g, const int64_t *walk_off)
{
    if (!g || !g->state || !walk_off) return PHI_HOST_ERR_INVALID;
    const size_t n = g->state->walks.size();
    g->walk_off.assign(walk_off, walk_off + n + 1);
    g->let_state_go();
    return PHI_HOST_OK;
}

void phi_graph_free(phi_graph *g) { delete g; }
int32_t phi_graph_n_vtx(const phi_graph *g) { return g->n_seg; }
int32_t phi_graph_n_walks(const phi_graph *g) { return (int32_t)g->hap_names.size(); }
int64_t phi_graph_n_edges(const phi_graph *g) { return (int64_t)g->adj.size(); }
const char *phi_graph_seq_concat(const phi_graph *g) { return g->seq_concat; }
const int64_t *phi_graph_seq_off(const phi_graph *g) { return g->seq_off.data(); }
const int64_t *phi_graph_adj_off(const phi_graph *g) { return g->adj_off.data(); }
const int32_t *phi_graph_adj(const phi_graph *g) { return g->adj.data(); }
const int64_t *phi_graph_walk_off(const phi_graph *g) { return g->walk_off.data(); }
const int32_t *phi_graph_walk_vtx(const phi_graph *g) { return g->walk_vtx; }
const int32_t *phi_graph_topo_rank(const phi_graph *g) { return g->topo_rank.data(); }
const char *phi_graph_hap_name(const phi_graph *g, int32_t w)
{
    return (w >= 0 && w < (int32_t)g->hap_names.size()) ? g->hap_names[(size_t)w].c_str() : "";
}
const char *phi_graph_seg_name(const phi_graph *g, int32_t v)
{
    return (v >= 0 && v < g->n_seg) ? g->name_arena.data() + g->name_off[(size_t)v] : "";
}

}  // extern "C"
