// phi_ctx.h -- the context behind the opaque phi_ctx handle of include/phi_amd.h.
#pragma once
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/phi_amd.h"
#include "phi_kernels.h"

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PhiAnchorHost {       // one dp anchor on the host (certificate / branch-and-bound)
    uint32_t slot;           // minimiser identity: dense id (rank of first occurrence among the walk minimisers)
    uint32_t e0, e1;         // first / last walk entry (phi_ent_t of phi_kernels.h: unsigned, below PHI_MAX_ENTRIES)
};

// host array that is NOT value-initialised on allocation: the threads that fill it touch its pages
// first (a std::vector would zero 38 MB on one thread before the threaded pass starts)
template <class T> struct PhiRawBuf {
    T *p = nullptr;
    size_t n = 0;
    PhiRawBuf() = default;
    PhiRawBuf(const PhiRawBuf &) = delete;
    PhiRawBuf &operator=(const PhiRawBuf &) = delete;
    ~PhiRawBuf() { free(p); }
    bool resize(size_t m)
    {
        free(p);
        p = m ? static_cast<T *>(malloc(m * sizeof(T))) : nullptr;
        n = p ? m : 0;
        return m == 0 || p != nullptr;
    }
    size_t size() const { return n; }
    T *data() const { return p; }
    T &operator[](size_t i) const { return p[i]; }
};

// view of host anchors: the pinned download buffer, or an owned vector
struct PhiAnchorSpan {
    PhiAnchorHost *p = nullptr;
    int64_t n = 0;
    size_t size() const { return (size_t)n; }
    PhiAnchorHost &operator[](int64_t i) const { return p[i]; }
    PhiAnchorHost *begin() const { return p; }
    PhiAnchorHost *end() const { return p + n; }
};

struct PhiComm;               // phi_comm.hip: the RCCL communicator of a multi-GPU job
struct PhiIpc;                // phi_ipc.hip: a group of PROCESSES of one node that map each other's hit vectors
#define PHI_HIT_RING 4        // hit buffers of a context in such a group (two otherwise)
struct PhiPeers;              // phi_comm.hip: the contexts of one process that exchange through peer-mapped memory

struct phi_ctx {
    int device = 0;
    PhiComm *comm = nullptr;
    PhiPeers *peers = nullptr;
    int peer_rank = 0;
    PhiIpc *ipc = nullptr;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t aux_stream = nullptr;    // the host thread's copies inside phi_set_graph, while the GPU thread works on `stream`
    std::string last_error;
    std::mutex err_mu;                                // guards last_error

    // params (main.cpp:118-131)
    int32_t k = 31, w = 25, recombination = 100;
    float threshold = 1.0f;
    uint32_t flags = PHI_FLAG_QCLP | PHI_FLAG_MIXED;
    int64_t solve_budget = 4096;                      // DP runs the exact search may use (phi_set_solve_budget); <= 0 = no limit

    // ---- graph, host side (decode, validation)
    bool have_graph = false;
    int32_t n_vtx = 0, n_walks = 0;
    std::vector<char> h_seq;
    std::vector<int64_t> h_seq_off, h_adj_off, h_walk_off, h_walk_base;   // h_walk_base: flat base offset of each walk
    std::vector<int32_t> h_adj, h_topo_rank, h_topo;
    PhiRawBuf<int32_t> h_walk_vtx;                    // host copy of the walk entries
    int64_t n_entries = 0, walk_bases = 0;

    // ---- the walks resolved on the device from the text of the W-lines (walk_text.hip): the text while it is being uploaded and
    //      resolved; afterwards d_walk_vtx holds the entries and phi_set_graph(walk_vtx = NULL) takes them from there
    struct PhiWalkText {
        bool ready = false;
        int32_t n_walks = 0;
        std::vector<int64_t> tile0, t_len;            // first 4-KB tile and bytes of every walk's text in d_text
        std::vector<int32_t> ends;                    // first and last vertex of every walk (what phi_set_graph's host pass looks at)
        DevBuf d_text;
    } wtext;
    bool walks_on_device = false;
    int64_t walks_on_device_n = 0;

    // ---- graph, device side
    DevBuf d_seq, d_seq_off, d_walk_vtx, d_walk_off, d_topo, d_in_off, d_in_src;
    DevBuf d_e_out, d_st_rec, d_st_mask, d_in_packed;  // DP step stream (dp.hip)
    int dp_nw = 1;                                    // waves of the DP workgroup
    DevBuf d_wwords, d_wstarts, d_wbad, d_wascii;     // class space (contexts.hip): packed bases, start bitmap, non-ACGT mask, flat ASCII (only if needed)
    // graph-side de-duplication: walk entries of equal context form a class that is sketched once
    DevBuf d_vlen;                                    // bases of every vertex
    DevBuf d_ent_cls;                                 // class of every walk entry
    DevBuf d_cls_rep, d_cls_left, d_cls_mult, d_cls_base, d_cls_rec_off;   // per class: representative entry, has a left base, entries, first base in class space, first record
    int64_t n_cls = 0, cls_bases = 0;
    double index_gpu_ms = 0.0;                        // GPU time of the walk sketch (classes, class-space sketch, table): phi_index_stats
    // class records: (hash, class, position relative to the vertex, first / last entry of the representative under the k-mer, table slot)
    DevBuf d_rec_hash, d_rec_pos, d_rec_cls, d_rec_rel, d_rec_slot, d_rec_e0, d_rec_e1;
    int64_t n_rec = 0;
    DevBuf d_u_keys, d_u_rep, d_u_uid, d_u_replist;   // walk-minimiser table: keys, first record, dense id; dense id -> first record
    DevBuf d_u_kv;                                    // the same table as (key, id) pairs, for the read probes
    void *h_stage[2] = {nullptr, nullptr};            // pinned staging buffers of large uploads from pageable memory (phi_abi.hip upload_staged)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    DevBuf d_wpre;                                    // per DP run: prefix sums of the anchor weights (dp_events.hip)
    DevBuf d_rowdiag;                                 // class-lane blocks: the rows' own columns (dp_events.hip)
    DevBuf d_in_s, d_last_walk;                       // solve on the device copy of the anchors: a flag / the last walk per minimiser
    int64_t n_unique = 0;                             // distinct walk minimisers
    uint64_t u_cap = 0;
    DevBuf d_hit;                                     // uint8 per distinct walk minimiser
    // In a group of processes (phi_ipc.hip) the hit vector exists FOUR times and phi_reset_reads rotates d_hit <- alt.hit <-
    // hit_extra[0] <- hit_extra[1] <- d_hit: the peers read the vector of read set g while this rank scores read set g + 1,
    // and it is zeroed two read sets later.  hit_idx = which buffer of the ring (in the order the peers mapped them) d_hit is.
    DevBuf hit_extra[PHI_HIT_RING - 2];
    int hit_n = 2, hit_idx = 0;

    // ---- reads
    // The reference's read spectrum Sp_R (ILP_index.cpp:622-635) = the hit flags (read hashes that are walk minimisers, d_hit)
    // + the NOVEL read hashes.  The read kernels only LOG the novel ones, coalesced: d_novlog holds 1 << nov_shift entries per
    // 512-window chunk of every batch since the last flush (d_novcnt of them valid), d_ovlist what those entries could not
    // hold and what the byte-wise routine found.  sp_flush (phi_abi.hip) enters what has been logged into the set d_sp_keys
    // when something asks for |Sp_R| or for the list (phi_reads_stats, phi_solve, phi_spectrum_export / _import): once per
    // read set, with every lane of the launch an insert, instead of a dependent atomic behind each probe of the scoring wave.
    DevBuf d_sp_keys;                                 // the set (open addressing), valid for generation sp_set_gen
    uint64_t sp_cap = 0;
    DevBuf d_sp_cnt;                                  // [PHI_STRIPES][8] u64: keys in the set
    int64_t sp_set_gen = -1;                          // the generation of reads whose hashes the set holds (another one: emptied before use)
    DevBuf d_novlog, d_novcnt;
    int32_t nov_shift = 6;
    int64_t log_chunks = 0, log_done = 0;             // chunks in the log; of those, already in the set
    int64_t logged_done = 0;                          // the n_logged stripes' sum at the last flush
    DevBuf d_ovlist;
    int64_t ov_cap = 0, ov_done = 0, ov_bound = 0;    // overflow list: capacity; entries already in the set; what the batches so far may have sent there at most
    bool async_batches = false;                       // batches went in without a wait behind them (phi_add_reads_device): their overflow shows at the next check
    int64_t last_log_chunks = 0;                      // log chunks of the last batch (a replay rewinds the log by them)
    int64_t reads_bases = 0, reads_count = 0;
    int64_t spectrum_override = -1;
    DevBuf d_rbases, d_roff, d_roff_made, d_export, d_peer_send;
    // device scalars: [0] err(u32 in low half) [1] n_bad ... [8..10] three rotating overflow counters (generation g: g % 3)
    DevBuf d_scalars;
    DevBuf d_stripes;                                 // [2][PHI_STRIPES][8] u64: novel hashes logged (with duplicates), emitted records
    int64_t sp_gen = 0;
    // Double buffers: what a generation of reads accumulates in place (d_hit, d_stripes) exists twice.  phi_reset_reads swaps
    // the two on the host; what the ended generation filled is zeroed by the waves of the next read launch (sketch.hip
    // clean_finish), ready for the generation after: no reset launch.  The log needs none of it: a generation starts it over.
    struct PhiReadBufs {
        DevBuf hit, stripes;
        bool needs_clean = false;                     // filled by the ended generation, not zeroed yet
    } alt;
    bool next_flag_zeroed = false;                    // a launch of this generation has zeroed the overflow counter of the next one

    uint32_t *h_err = nullptr;                        // pinned copy of the device error word, fetched behind a batch's last kernel
    // ---- reads as raw text (phi_add_reads_text, reads_text.hip): the device finds the records
    struct PhiTextStream {
        bool active = false, irregular = false, started = false, detached = false;
        bool carry_stale = false;                     // the carry is on the device only (pieces from parked text): h_carry is made when asked for
        int mode = 0;                                 // 0 FASTA, 1 FASTQ with four lines per record
        int64_t fed = 0, taken = 0;                   // stream bytes handed over / taken as whole records
        uint32_t carry_cap = 0, chunk_cap = 0, line_cap = 0;
        DevBuf text[2], bases[2], roff[2], tile_cnt, ls, pre, blk, sum;   // two slots: chunk i + 1 is copied while chunk i is sketched
        int slot = 0;                                 // slot of the last chunk
        uint32_t carry_len = 0, carry_at = 0;         // the carry: text[slot][carry_at, carry_at + carry_len)
        std::vector<char> h_carry;                    // the same bytes on the host (what phi_reads_text_end hands back)
        PhiTextSummary *h_sum = nullptr;              // pinned
        hipEvent_t ev_copy = nullptr;
        int last_slot = -1;                           // the chunk whose sketch may still be running: replayed if the spectrum set overflowed
        int64_t last_reads = 0, last_bases = 0;
        bool last_uniform = false;
        int dbg_slot = 0;                             // what the last piece took (phi_reads_text_last_batch: the parity tests)
        int64_t dbg_reads = 0, dbg_bases = 0;
        uint32_t why = 0, first_bad = 0;
    } text;

    // ---- scratch for sketch passes and compaction
    DevBuf d_blk_cnt, d_blk_off, d_flags, d_flags2, d_list, d_list2, d_list3, d_walk_last;
    DevBuf d_sel_off, d_sel_tri;                       // the filter's selected class records, packed per class (phi_solve: the anchors are expanded from these)
    DevBuf d_adj_off, d_adj, d_topo_rank, d_cnt_edge, d_walk_err;   // the walk-entry pass on the GPU (phi_walk_edges_kernel)
    DevBuf d_sa_cnt, d_sa_cur, d_sa_off, d_sa_idx;    // minimiser -> anchors CSR, built on the GPU

    // ---- solve state
    DevBuf d_m_rec, d_m_group, d_g_keys, d_g_rep, d_g_cnt, d_slot_maxcnt, d_slot_multi;
    DevBuf d_a_e1, d_g_off, d_g_span, d_a_weight;
    DevBuf d_dmax, d_bstart, d_top, d_ent, d_word;
    // event-driven DP (dp_events.hip)
    bool dp_dense_ready = false;                      // the every-vertex stream of dp.hip is on the device (fallback of the 4-wave event kernel)
    bool dp_events = false;
    int32_t n_k = 0;                                  // compact steps
    int64_t n_ev = 0;                                 // events
    std::vector<int32_t> h_cstep, h_kstep;            // topological step -> compact step (-1) and back
    // blocks of compact steps solved in parallel (dp_events.hip DP_ROW / DP_PATH, <= 64 walks): per graph, the steps a cut
    // may not sit before (a recombination edge would cross it / it would split a pair of allele steps); per solve, the cuts
    std::vector<int32_t> h_k_cut_ok;                  // [n_k + 1]: 1 = structurally a cut may sit before step k
    bool dp_blocks = false;
    int32_t n_blk = 0, blk_ring = 1024, blk_max_len = 0;
    std::vector<int32_t> h_blk_lo;
    DevBuf d_blk_lo, d_blk_ev, d_blk_S, d_row_out, d_rowend, d_blk_keys, d_blk_carry, d_cov, d_cov2, d_stepdiff;
    // more than 64 walks: the blocks' rows run on class lanes (dp_events.hip), regrouped per DP run
    bool dp_cls = false;
    int32_t blk_cls_target = 16;        // class-lane blocks: steps per block aimed at (halved when a block holds more than 64 classes)
    bool blk_no_small = false;          // this solve met a block task with more than 8 live runs on a lane: no 256-step blocks
    int32_t blk_ls = 64;                // row length of the per-(block, walk) tables
    DevBuf d_lane_walk, d_walk_lane, d_coff, d_blk_ncls, d_rownew, d_blk_bad;
    DevBuf d_seg_lo, d_seg_row, d_seg_S;  // the chain over the blocks cut into segments (dp_events.hip)
    DevBuf d_k_rec, d_k_in, d_cvtx, d_ev_e, d_ev_off, d_ev, d_off_end, d_off_start, d_scan_blk, d_scan_blk64, d_scan_blkoff;
    // The kept anchors as triples (minimiser id, first entry, last entry) in HBM, and whether the host has its copy:
    // a large model whose anchors all span an edge is solved on the device copy (solve_dev.hip); h_kept / h_dp are then
    // empty until something needs them (branch and bound, phi_kept_anchors: phi_host_anchors fetches them).
    DevBuf d_anchors, d_cov_all, d_cov_w, d_slots, d_slots2, d_segs, d_ctr, d_vmax;
    int64_t n_kept = 0, n_dp = 0;
    bool anchors_host = false;
    PhiAnchorSpan h_kept, h_dp;                       // kept anchors (in h_pin); dp anchors (span >= 1 edge: h_kept itself or h_dp_own)
    std::vector<PhiAnchorHost> h_dp_own;
    void *h_pin = nullptr;                            // pinned host buffer the kept anchors are downloaded into
    size_t h_pin_cap = 0;
    std::future<void> pin_future;                     // its allocation, started by phi_set_graph on a thread of its own
    std::future<int> dp_alloc_future;                 // the DP's entry-sized buffers of a chromosome-scale graph, allocated beside phi_set_graph's host pass
    std::vector<uint64_t> h_kept_hash;
    std::vector<int32_t> h_path_vtx, h_path_hap;
    std::vector<int64_t> h_n_minimizers, h_n_anchors;
    phi_result result{};
    bool solved = false;

    // ---- profiling of the sketch kernel
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    int prof_period = 1;                              // every prof_period-th sketch launch is bracketed
    uint64_t prof_seq = 0;
    size_t prof_used = 0;
    int64_t prof_bases = 0;
    double prof_ms_done = 0.0;
    int64_t prof_n_done = 0;
};

// PHI_TIMING=1: stage timings of phi_set_graph / phi_solve on stderr
struct PhiStageTimer {
    bool on;
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit PhiStageTimer(const char *w) : on(getenv("PHI_TIMING") != nullptr), what(w), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *stage)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[phi timing] %s: %-28s %8.3f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Blocking copies / fills ON THE CONTEXT'S STREAM.  hipMemcpy / hipMemset would go through the null stream, whose hardware
// queue the runtime creates at first use (~20 ms, and as much again when the process ends) for nothing: every other
// piece of work of a context runs on its own streams.
static inline hipError_t phi_copy_sync(phi_ctx *c, void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}
static inline hipError_t phi_memset_sync(phi_ctx *c, void *dst, int v, size_t n)
{
    const hipError_t e = hipMemsetAsync(dst, v, n, c->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->stream);
}

// walk of a walk entry
static inline int32_t phi_entry_walk(const phi_ctx *c, int64_t e)
{
    return (int32_t)(std::upper_bound(c->h_walk_off.begin(), c->h_walk_off.end(), e) - c->h_walk_off.begin()) - 1;
}

// ---- host threads for the O(walk entries) preparation of phi_set_graph
static inline int phi_host_threads()
{
    const char *e = getenv("PHI_HOST_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    return n > 16 ? 16 : n;
}

// fn(lo, hi, worker) over [0, n) in chunks handed out dynamically; worker < phi_host_threads()
template <class F> static void phi_parallel_chunks(int64_t n, int64_t chunk, F fn)
{
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    int nt = phi_host_threads();
    if (nt > n_chunks) nt = (int)n_chunks;
    if (nt <= 1) {
        for (int64_t i = 0; i < n_chunks; i++) fn(i * chunk, std::min(n, (i + 1) * chunk), 0);
        return;
    }
    std::atomic<int64_t> next{0};
    auto work = [&](int worker) {
        for (;;) {
            const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_chunks) break;
            fn(i * chunk, std::min(n, (i + 1) * chunk), worker);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
}

// first error raised by any worker
struct PhiHostError {
    std::atomic<int> flag{0};
    std::mutex m;
    int code = 0;
    std::string msg;
    bool failed() const { return flag.load(std::memory_order_relaxed) != 0; }
    void set(int code_, const char *fmt, ...)
    {
        std::lock_guard<std::mutex> g(m);
        if (flag.load()) return;
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        code = code_; msg = buf;
        flag.store(1);
    }
};

// phi_solve.cpp: host orchestration of the exact solve on top of the DP kernel
int phi_solve_impl(phi_ctx *c);
// one batch of reads the device can address; replay: the same batch again after the spectrum set was regrown
int phi_add_reads_device_impl(phi_ctx *c, const void *d_bases, const void *d_read_off, int64_t n_reads, int64_t n_bases, bool replay);
// helpers shared between phi_abi.hip and phi_solve.hip
int phi_fail(phi_ctx *c, int code, const char *fmt, ...);
int phi_dev_ensure(phi_ctx *c, DevBuf &b, size_t bytes);
void phi_pool_flush(int device);                    // the pool of large device buffers let go (phi_abi.hip) back to the driver
int phi_hip_check(phi_ctx *c, hipError_t e, const char *what);
int phi_sync_check(phi_ctx *c);
// pinned host buffer of at least `bytes` (contents are not kept)
int phi_pin_ensure(phi_ctx *c, size_t bytes);
int phi_host_anchors(phi_ctx *c);                      // the host copy of the kept anchors, fetched if it is not there
// phi_ipc.hip
int phi_ipc_wait_pending(phi_ctx *c);                  // the context's stream waits for the last gather (every observer of the hit vector: phi_flush_reset)
int phi_ipc_before_generation(phi_ctx *c, int64_t gen); // phi_reset_reads, two resets in a row: before a hit buffer is zeroed by a launch of its own
void phi_ipc_launch_args(phi_ctx *c, PhiSketchArgs &A);  // every read launch: the flags its waves publish / look at
// sums of the striped counters (waits for the stream): novel hashes logged (with duplicates), emitted records
int phi_read_counts(phi_ctx *c, uint64_t *n_logged, uint64_t *n_emitted);
int phi_spectrum_count(phi_ctx *c, uint64_t *n_distinct);
int phi_scan_counts_wide(phi_ctx *c, const int32_t *cnt, int64_t n, int64_t *off);
// flags[n] (0/1) -> ascending list of flagged indices (int32) in out
int phi_compact(phi_ctx *c, const uint8_t *flags, int64_t n, DevBuf &out, int64_t *n_out);
