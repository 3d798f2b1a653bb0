// phi_kernels.h -- launch interface between the C-ABI layer (phi_abi.hip) and the kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#ifndef PHI_TPB
#define PHI_TPB 256            // lanes per workgroup of the sketch kernel (4 waves)

#endif
#define PHI_WCH 512            // window positions per wave (each wave sketches its own chunk)
#define PHI_MAX_W 256
#define PHI_MAX_K 64              // k-mers of up to 32 bases are 2-bit values in 64 bits (the fast kernels); 33 .. 64 take the
#define PHI_MAX_K_PACKED 32       // exact byte-wise routine for every window (slow, exact: the reference is string based, any k)
#define PHI_MAX_PROBE 4096     // linear-probe bound of the open-addressed tables
// log of the read hashes that are not walk minimisers: 1 << shift entries per chunk of 512 windows, 1.5x what random
// sequence emits at this w (2 / (w + 1) per window), a power of two in [16, 512]; what a chunk has beyond goes to the
// generation's overflow list
static inline int phi_nov_shift(int w)
{
    const int want = (3 * 2 * PHI_WCH) / (2 * (w + 1));
    int sh = 4;
    while ((1 << sh) < want && sh < 9) sh++;
    return sh;
}
#define PHI_STRIPES 256          // counters are striped over 256 cache lines: one hot address
                                 // serialises at ~12 ns per atomic (MI355X_MICROARCH.md "fanin")
#define PHI_RCAP 32            // DP run-length states 0..31 (an anchor spans <= k-1 <= 31 edges)

// A walk entry is an index into the concatenated walks (walk_off / walk_vtx of phi_set_graph).  Entries are held in
// 32 bits, UNSIGNED, wherever they are stored (class representatives, record and anchor entries, events, path
// stretches): every load widens to 64 bits before it indexes anything.  Per-entry arrays are indexed with 64 bits.
typedef uint32_t phi_ent_t;
#define PHI_MAX_ENTRIES (((int64_t)1 << 32) - 64)   // (a little below 2^32: loops like `for (e = e0; e <= e1; e++)` and e + 1 never wrap)

enum { PHI_MODE_COUNT = 0, PHI_MODE_WRITE = 1, PHI_MODE_PROBE = 2 };

// words (u64) of a rank's flag block in a group of processes (phi_ipc.hip): 0 .. 3 the flags the peers read; the error word;
// local: last exchange whose read set is scored, last exchange gathered, workgroups of the gathers that have ended
#define PHI_MB_ERR 8
#define PHI_MB_SCORED 16
#define PHI_MB_GATHERED 17
#define PHI_MB_BLOCKS 18

// bits of the device error word
#define PHI_KERR_SENTINEL 1u    // a minimiser hashed to the empty-slot sentinel
#define PHI_KERR_TABLE_FULL 2u  // probe bound exceeded
#define PHI_KERR_WALK_EDGE 4u   // consecutive walk vertices not joined by a graph edge
#define PHI_KERR_CSR_ID 32u      // a minimiser id outside the table (internal error)
#define PHI_KERR_DP_QUEUE 16u    // event DP: more live young runs on a lane than its queue holds
#define PHI_KERR_DP_CLASSES 64u  // block DP on class lanes: a block holds more than 64 classes of walks

struct PhiSketchArgs {
    const uint64_t *words;                 // packed bases (+2 padding words)
    const unsigned long long *starts;      // sequence-start bitmap
    int64_t n_bases;
    int32_t k, w;
    // bases outside ACGTacgt: one bit per base (null = none), the flat ASCII they live in, and
    // allslow = take the exact byte-wise path for every window (ordered write of such sequences)
    const unsigned long long *badbits;
    const uint8_t *ascii;
    int32_t allslow;
    // PHI_MODE_COUNT / PHI_MODE_WRITE
    int32_t *block_cnt;
    const int64_t *block_off;
    uint64_t *out_hash;
    int64_t *out_pos;
    // PHI_MODE_PROBE
    unsigned long long *n_logged;          // [PHI_STRIPES][8] striped counter of novel hashes logged (with duplicates: an upper bound of the set's growth)
    unsigned long long *n_emitted;         // [PHI_STRIPES][8] striped counter of emitted records
    const uint64_t *u_kv; uint64_t u_mask; // walk-minimiser table as (key, dense id) pairs: one 16-byte load per probe
    uint8_t *hit;                          // per distinct walk minimiser (dense id)
    uint32_t *err;
    // PHI_MODE_PROBE: log of the NOVEL read hashes (emitted, not walk minimisers): 1 << nov_shift entries per chunk,
    // chunk log_base + i of the log, written in order by the wave that owns the chunk (coalesced); nov_cnt = how many.
    // What a chunk (pooled kernel: a wave) has beyond its entries, and what the byte-wise routine finds, goes to
    // ov_list[atomicAdd(ov_count)]; a full list raises PHI_KERR_TABLE_FULL (the host grows it and replays the batch).
    // The read-spectrum set is made from these when |Sp_R| is asked for (phi_abi.hip sp_flush, table.hip).
    uint64_t *nov_log; uint16_t *nov_cnt; int64_t log_base; int32_t nov_shift;
    uint64_t *ov_list; unsigned long long *ov_count; int64_t ov_cap;
    // PHI_MODE_PROBE reads ASCII: the 2-bit pack, the bases outside ACGTacgt and the read-start bitmap of a chunk
    // are made by the wave that sketches it (no preparation launch): `ascii` + read offsets
    const int64_t *read_off; int64_t n_reads;
    uint32_t reads_per_base_q32;           // n_reads / n_bases in 0.32 fixed point (first guess of the read-start search: no division per wave)
    int32_t uniform_len;                   // > 0: every read has this length (>= 32) and there are NO offsets: read r starts at r * uniform_len
    double inv_len;                        // 1.0 / uniform_len
    uint32_t inv_len_q32;                  // floor(2^32 / uniform_len)
    int32_t wave_stride;                   // reads, k <= 32: wave g takes the chunks g, g + wave_stride, ... (set by the launcher)
    // ... and, in the first launch after a reset, every wave also empties its share of the buffers the PREVIOUS
    // generation of reads filled (the other half of the context's double buffers), for the generation after this one
    int32_t q_clean;
    unsigned long long *ov_zero;           // the overflow counter of the generation after this one (three rotate)
    // a context in a group of processes (phi_ipc.hip): this rank's flag block.  The launch's first wave publishes "the
    // exchanges issued before this launch have their read sets scored" (ipc_scored: the stream is in order, so every
    // earlier scoring launch has ended) -- the gather kernels on the group's stream wait for that flag, no launch or event
    // on this stream in between --; and before a wave zeroes its share of a retired hit vector it makes sure this rank's
    // gather ipc_need has ended (by then every peer has read that vector; true long before, except when a peer lags)
    unsigned long long *ipc_mb; unsigned long long ipc_scored, ipc_need;
    uint64_t *q_hit_words; int64_t q_n_hit_words; uint64_t *q_stripes; int64_t q_n_stripe_words;
};

// sketch.hip
void phi_launch_pack_ascii(hipStream_t st, const uint8_t *bases, int64_t n, uint64_t *words, int64_t n_words,
                           uint32_t *badbits, unsigned long long *n_bad);
void phi_launch_mark_starts(hipStream_t st, const int64_t *seq_off, int64_t n_seq, unsigned long long *starts);

void phi_launch_sketch_bytes(hipStream_t st, int mode, const PhiSketchArgs &A, const unsigned long long *batch_bad);
void phi_launch_reset_reads(hipStream_t st, uint64_t *hit_words, int64_t n_hit_words, uint64_t *stripes, int64_t n_stripe_words);
int64_t phi_sketch_num_blocks(int64_t n_bases);
void phi_launch_sketch(hipStream_t st, int mode, const PhiSketchArgs &A, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void phi_launch_scan_counts(hipStream_t st, const int32_t *cnt, int64_t n, int64_t *off);

// table.hip
// Insert every walk-minimiser hash; u_rep[slot] = smallest record index holding that hash
// (deterministic on every rank); rec_slot[i] = slot of record i.
void phi_launch_table_build(hipStream_t st, const uint64_t *rec_hash, int64_t n_rec, uint64_t *u_keys,
                            uint32_t *u_rep, uint64_t u_mask, uint32_t *rec_slot, uint32_t *err);
void phi_launch_rep_flags(hipStream_t st, const uint32_t *rec_slot, int64_t n_rec, const uint32_t *u_rep,
                          uint8_t *flags);
void phi_launch_table_compact(hipStream_t st, const int32_t *rep_list, int64_t n_unique, const uint64_t *rec_hash, int64_t n_rec,
                              uint64_t *keys, uint32_t *uid, uint64_t mask, uint32_t *rec_slot, uint32_t *err);
void phi_launch_share_hist(hipStream_t st, const uint64_t *keys, int64_t cap, const int32_t *n_walks_of, unsigned long long *hist);
void phi_launch_slot_uid(hipStream_t st, const int32_t *rep_list, int64_t n_unique, const uint32_t *rec_slot,
                         uint32_t *u_uid);
// (key, id) pairs of the walk-minimiser table for the read probes: kv[2s] = keys[s], kv[2s+1] = uid[s]
void phi_launch_table_pairs(hipStream_t st, const uint64_t *keys, const uint32_t *uid, int64_t cap, uint64_t *kv);
void phi_launch_fill_u64(hipStream_t st, uint64_t *p, int64_t n, uint64_t v);
void phi_launch_fill_u32(hipStream_t st, uint32_t *p, int64_t n, uint32_t v);
void phi_launch_iota_i64(hipStream_t st, int64_t *p, int64_t n, int64_t step);   // p[i] = i * step
// insert a list of hashes into the spectrum set (multi-GPU spectrum merge)
void phi_launch_spectrum_insert(hipStream_t st, const uint64_t *hashes, int64_t n, uint64_t *sp_keys,
                                uint64_t sp_mask, unsigned long long *sp_count, const uint64_t *u_keys, uint64_t u_mask,
                                const uint32_t *u_uid, uint8_t *hit, uint32_t *err);
// the logged novel hashes of chunks [c_lo, c_hi) (1 << shift entries each, cnt[c] of them valid) into the spectrum set
void phi_launch_spectrum_flush(hipStream_t st, const uint64_t *nov_log, const uint16_t *nov_cnt, int64_t c_lo, int64_t c_hi, int32_t shift,
                               uint64_t *sp_keys, uint64_t sp_mask, unsigned long long *sp_count, uint32_t *err);
// *n_out += number of non-zero bytes of flags[0..n)
void phi_launch_count_flags(hipStream_t st, const uint8_t *flags, int64_t n, unsigned long long *n_out);
// compact the occupied slots of the spectrum set into a list; *n_out receives the count
void phi_launch_spectrum_export(hipStream_t st, const uint64_t *sp_keys, int64_t cap, uint64_t *out,
                                unsigned long long *n_out);

// contexts.hip: graph-side de-duplication (entries of equal context form a class, sketched once)
struct PhiClassArgs {
    const int32_t *walk_vtx; const int64_t *walk_off; int32_t n_walks; int64_t n_entries;
    const int32_t *vlen; const uint8_t *seq; const int64_t *seq_off;
    int32_t tail_need;                     // w + k - 2: bases of a window beyond its first
    uint64_t seed;
    uint64_t *t_keys; uint32_t *t_rep; uint32_t *t_mult; uint64_t t_mask;   // class table: fingerprint, smallest entry, entries
    uint32_t *ent_slot;                    // per entry: its table slot
    uint32_t *err;
};
struct PhiExpandArgs {
    const int32_t *ent_cls; int64_t e_lo, e_hi;
    const int32_t *cls_rec_off; const phi_ent_t *cls_rep;
    const uint8_t *sel; const int32_t *sel_cnt;          // optional: selected records, and their number per class
    int32_t *block_cnt; const int64_t *block_off;
    // kind 0: minimisers of one walk
    const uint64_t *rec_hash; const int32_t *rec_rel; const int64_t *ent_base; uint64_t *out_hash; int64_t *out_pos;
    // kind 1: anchors (dense minimiser id, first entry, last entry)
    const uint32_t *rec_slot; const uint32_t *u_uid; const phi_ent_t *rec_e0; const phi_ent_t *rec_e1; uint32_t *out_tri;
};
void phi_launch_vlen(hipStream_t st, const int64_t *seq_off, int64_t n_vtx, int32_t *vlen);
void phi_launch_walk_bases(hipStream_t st, const int32_t *walk_vtx, const int32_t *vlen, const int64_t *walk_off, int32_t n_walks,
                           int64_t n_entries, unsigned long long *out);
void phi_launch_walk_rec_counts(hipStream_t st, const int32_t *ent_cls, const int32_t *cls_rec_off, const int64_t *walk_off,
                                int32_t n_walks, int64_t n_entries, unsigned long long *out);
void phi_launch_class_insert(hipStream_t st, const PhiClassArgs &A);
void phi_launch_class_verify(hipStream_t st, const PhiClassArgs &A, uint8_t *is_rep);
void phi_launch_class_ids(hipStream_t st, const phi_ent_t *cls_rep, int64_t n_cls, const uint32_t *ent_slot, int64_t n_entries,
                          const uint32_t *t_mult, uint32_t *t_cid, int32_t *cls_mult, int32_t *ent_cls);
void phi_launch_class_len(hipStream_t st, const PhiClassArgs &A, const phi_ent_t *cls_rep, int64_t n_cls, int32_t *cls_len, uint8_t *cls_left);
void phi_launch_pack_classes(hipStream_t st, const uint8_t *seq, const int64_t *seq_off, const int32_t *walk_vtx, const int32_t *vlen,
                             const phi_ent_t *cls_rep, const uint8_t *cls_left, const int64_t *cls_base, int64_t n_cls, uint64_t *words,
                             int64_t n_words, uint32_t *badbits, uint8_t *ascii, unsigned long long *n_bad);
void phi_launch_class_rec(hipStream_t st, const int64_t *raw_pos, int64_t n_raw, const int64_t *cls_base, int64_t n_cls,
                          const phi_ent_t *cls_rep, const uint8_t *cls_left, const int32_t *walk_vtx, const int32_t *vlen, int32_t k,
                          uint8_t *keep, int32_t *r_cls, int32_t *r_rel, phi_ent_t *r_e0, phi_ent_t *r_e1);
void phi_launch_class_rec_gather(hipStream_t st, const int32_t *idx, int64_t n, const uint64_t *raw_hash, const int32_t *r_cls,
                                 const int32_t *r_rel, const phi_ent_t *r_e0, const phi_ent_t *r_e1, uint64_t *o_hash, int32_t *o_cls,
                                 int32_t *o_rel, phi_ent_t *o_e0, phi_ent_t *o_e1);
void phi_launch_class_rec_off(hipStream_t st, const int32_t *rec_cls, int64_t n_rec, int64_t n_cls, int32_t *off);
int64_t phi_expand_num_blocks(int64_t n_entries);
void phi_launch_expand_count(hipStream_t st, const PhiExpandArgs &A);
void phi_launch_expand_write(hipStream_t st, const PhiExpandArgs &A, int kind);
void phi_launch_class_sel_count(hipStream_t st, const uint8_t *sel, const int32_t *cls_rec_off, int64_t n_cls, int32_t *sel_cnt);
void phi_launch_class_sel_tri(hipStream_t st, const uint8_t *sel, const int32_t *cls_rec_off, int64_t n_cls, const int32_t *sel_off, const phi_ent_t *cls_rep,
                              const uint32_t *rec_slot, const uint32_t *u_uid, const phi_ent_t *rec_e0, const phi_ent_t *rec_e1, int32_t *sel_tri);
// (a_e1 != NULL: the DP's per-anchor arrays, the anchors per walk and the checks of phi_launch_anchor_prep are made on the way)
void phi_launch_expand_tri(hipStream_t st, const int32_t *ent_cls, int64_t e_lo, int64_t e_hi, const int32_t *sel_off, const int32_t *sel_tri,
                           const int64_t *block_off, uint32_t *out_tri, phi_ent_t *a_e1, uint8_t *a_span, const int64_t *walk_off, int32_t n_walks,
                           unsigned long long *walk_cnt, unsigned long long *prep, int64_t n_anchors);
void phi_launch_mark_list(hipStream_t st, const int32_t *list, const int32_t *through, int64_t n, uint8_t *sel);
void phi_launch_share_count_cls(hipStream_t st, const int32_t *ent_cls, int64_t e_lo, int64_t e_hi, const int32_t *cls_rec_off,
                                const uint32_t *rec_slot, int32_t walk, int32_t *last_walk, int32_t *n_walks_of);
void phi_launch_entry_len_range(hipStream_t st, const int32_t *walk_vtx, const int32_t *vlen, int64_t e_lo, int64_t n, int32_t *lens);

// anchors.hip
#define PHI_KERR_FP_COLLISION 8u   // two different vertex lists share a fingerprint: reseed
// walk entries -> out-edge index per entry, walks per edge, walks per vertex (see anchors.hip)
void phi_launch_walk_edges(hipStream_t st, const int32_t *walk_vtx, const int64_t *walk_off, int32_t n_walks, int64_t n_entries, int32_t n_vtx,
                           const int64_t *adj_off, const int32_t *adj, const int64_t *seq_off, const int32_t *topo_rank,
                           uint8_t *e_out, int32_t *cnt_edge, unsigned long long *st_mask, int32_t nw64, int32_t *err);
// CSR minimiser id -> anchor indices of a triple list (id, e0, e1): cnt / cur zeroed by the caller, off from a scan of cnt
void phi_launch_csr_count(hipStream_t st, const uint32_t *triples, int64_t n, int64_t n_ids, int32_t *cnt, uint32_t *err);
void phi_launch_csr_scatter(hipStream_t st, const uint32_t *triples, int64_t n, int64_t n_ids, const int32_t *off, int32_t *cur,
                            int32_t *idx);
void phi_launch_csr_sort(hipStream_t st, const int32_t *off, int64_t n_ids, int32_t *idx);
int64_t phi_compact_num_blocks(int64_t n);
void phi_launch_flag_count(hipStream_t st, const uint8_t *flags, int64_t n, int32_t *block_cnt);
void phi_launch_flag_write(hipStream_t st, const uint8_t *flags, int64_t n, const int64_t *block_off, int32_t *out);
void phi_launch_match_flags(hipStream_t st, const uint32_t *rec_slot, int64_t n_rec, const uint32_t *u_uid,
                            const uint8_t *hit, uint8_t *flags);

struct PhiFilterArgs {
    const uint32_t *rec_slot; const phi_ent_t *rec_e0; const phi_ent_t *rec_e1;   // per class record (entries of the class representative)
    const int32_t *walk_vtx;
    const int32_t *rec_cls; const int32_t *cls_mult;   // class of a record, walk entries in a class (its multiplicity)
    const int32_t *m_rec;                          // matched records (ascending)
    uint64_t *g_keys; int32_t *g_rep; uint32_t *g_cnt; uint64_t g_mask; uint64_t seed;   // group table
    int32_t *m_group;                              // group slot of each matched anchor
    const uint32_t *u_uid;                         // minimiser-table slot -> dense minimiser id
    uint32_t *slot_maxcnt;                         // per dense minimiser id: largest group
    uint8_t *slot_multi;                           // per id: has an anchor spanning >= 2 vertices
    float limit;                                   // threshold * num_walks (ILP_index.cpp:698)
    unsigned long long *counters;                  // [0] filtered  [1] in model
    uint32_t *err;
};
void phi_launch_group_insert(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched);
void phi_launch_group_count(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched);
void phi_launch_group_max(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched);
void phi_launch_slot_count(hipStream_t st, const PhiFilterArgs &A, int64_t u_cap);
void phi_launch_kept_flags(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched, uint8_t *kept, uint8_t *dp);
void phi_launch_gather_i32(hipStream_t st, const int32_t *src, const uint32_t *idx, int64_t n, int32_t *out);
void phi_launch_gather_u64(hipStream_t st, const uint64_t *src, const int32_t *idx, int64_t n, uint64_t *out);
void phi_launch_entry_csr(hipStream_t st, const phi_ent_t *a_e1, int64_t n_a, int64_t n_entries, int64_t *g_off);

// dp.hip
#define PHI_DP_CHUNK 128        // steps staged through LDS at a time
#define PHI_DP_RING 2048        // steps whose leaving states are kept in LDS
#define PHI_DP_NEED_ENTRY 1     // step flag: a recombination can enter this vertex
#define PHI_DP_NEED_TOPS 2      // step flag: a recombination can leave this vertex
#define PHI_DP_MAX_WALKS 1022       // one walk per lane of sixteen waves; a walk id + 1 has 10 bits in the packed tops

struct PhiDpArgs {
    int32_t n_vtx, n_walks;
    // step stream (static per graph, built by phi_set_graph)
    const int32_t *st_rec;               // [n_vtx][8]: flags | n_in<<8, overflow start, 3 inline in-edges, vertex, -, -
    const unsigned long long *st_mask;   // [n_vtx][NW]: walks on the vertex of each step
    const int32_t *in_packed;            // in-edges beyond the third of a step: back<<8 | out-edge index
    const int64_t *walk_off;
    // per run
    const uint64_t *word;                // per walk entry: out-edge index + spans of weight-1 anchors ending there
    const int64_t *g_off; const uint8_t *g_span; const uint8_t *a_weight;   // CSR, only for overflowing entries
    int32_t cost;                        // 2 * (R / 2)
    // outputs
    int32_t *dmax;                       // per entry (only where a path can end or leave): best score
    int32_t *bstart;                     // per entry: walk index where the run attaining it began (ties: the longest run)
    int32_t *tops;                       // [n_vtx][5] by step: top1 value/walk/out-edge, top2 value/walk
    int32_t *ent_src, *ent_h;            // per step: source step and walk of the best recombination entry
};
void phi_launch_dp(hipStream_t st, const PhiDpArgs &A);

// event-driven DP (dp_events.hip): up to PHI_DP_EVENT_MAX_WALKS walks
#define PHI_DP_EVENT_MAX_WALKS 256
#define PHI_DP_EVENT_SAFE_WALKS 128  // beyond: per-lane queues of 16 runs; a deeper one makes the caller fall back to dp.hip
#define PHI_DP_LANE_ONLY 4      // compact-step flag: a walk starts or ends on the vertex (no ENTRY / TOPS work)
#define PHI_DP_PAIR 8           // compact-step flag: this step and the next have no TOPS and no walk in common
                                // (two alleles of one site): the consumer may take them in one iteration
struct PhiDpEventArgs {
    int32_t n_k, n_walks;                // compact steps (vertices with ENTRY / TOPS / a walk start or end)
    int64_t n_ev;                        // events = walk entries on those vertices
    // static per graph
    const int32_t *k_rec;                // [n_k][8]: flags | n_in<<8, overflow start, 3 inline in-edges (compact steps back<<8 | out-edge), vertex
    const int32_t *k_in_packed;
    const int64_t *walk_off;
    const phi_ent_t *ev_e;               // [n_ev] walk entry of each event, ascending
    const int64_t *ev_off;               // [n_walks + 1] first event of each walk
    // per run
    void *ev;                            // [n_ev] 48-byte event records (phi_dp_event_fill_kernel)
    const int64_t *g_off; const uint8_t *g_span; const uint8_t *a_weight;   // CSR of the dp anchors by last entry
    int32_t cost;
    // outputs
    int32_t *dmax, *bstart;              // per entry, written at query events
    int32_t *tops;                       // [n_k] packed 16-byte tops
    int32_t *ent_src, *ent_h;            // per compact step
    uint32_t *err;                       // PHI_KERR_DP_QUEUE
    int32_t q_limit;                     // 0 = the kernel's queue depth; tests lower it to provoke the fallback
    // blocks of steps solved in parallel (<= 64 walks, dp_events.hip DP_ROW / DP_PATH): block b = steps [blk_lo[b], blk_lo[b+1])
    int32_t n_blk, blk_ring;             // blk_ring: 256, 1024 or 2048 >= the longest block
    int32_t blk_max_len;                 // the longest block in steps (0: not known)
    int32_t lane_stride;                 // row length of the per-(block, walk) tables: 64 (<= 64 walks) or 256
    const int32_t *lane_walk;            // DP_ROW on class lanes (> 64 walks): [n_blk][64] the walk that plays class lane l (-1: none)
    int32_t *rownew_out;                 // DP_ROW on class lanes: [n_blk * 65] best key of a run begun inside the block on the unit lane
    int32_t *rowdiag_out;                // DP_ROW on class lanes: [n_blk][64] row l's own column l (row_out holds NEGK there)
    const int32_t *blk_lo;               // [n_blk + 1]
    const int32_t *blk_ev;               // [n_blk][lane_stride]: first event of each walk inside the block
    const int32_t *blk_S;                // DP_PATH in: [n_blk][lane_stride] key of each walk entering the block (NEGK: none)
    int32_t *row_out; int32_t *rowend_out;   // DP_ROW out: [n_blk * (n_walks + 1)][64] keys at the block's end; best value of a path ending inside
    int32_t *blk_keys_out; int32_t *blk_carry;   // DP_PATH out: [n_blk][64] keys at the block's end; start of the run that carries them (-1: before the block)
};
#define PHI_DP_BLOCK_MAX 2048        // steps of a block at most (= the larger ring of tops)
#define PHI_DP_NEGK (-(1 << 30))      // "no run" in key space
void phi_launch_dp_events(hipStream_t st, const PhiDpEventArgs &A);
void phi_launch_dp_block_rows(hipStream_t st, const PhiDpEventArgs &A);
void phi_launch_dp_block_paths(hipStream_t st, const PhiDpEventArgs &A);
// the solve's bookkeeping on the device copy of the anchors (solve_dev.hip)
void phi_launch_anchor_prep(hipStream_t st, const uint32_t *tri, int64_t n, const int64_t *walk_off, int32_t n_walks, phi_ent_t *a_e1, uint8_t *a_span,
                            unsigned long long *walk_cnt, unsigned long long *out);
void phi_launch_vertex_most(hipStream_t st, const int64_t *g_off, int64_t n_entries, const int32_t *walk_vtx, int32_t *vmax);
void phi_launch_sum_i32(hipStream_t st, const int32_t *v, int64_t n, unsigned long long *out);
void phi_launch_repeat_walk(hipStream_t st, const uint32_t *tri, int64_t lo, int64_t hi, int32_t walk, int32_t *last, uint8_t *flags);
void phi_launch_weights(hipStream_t st, const uint32_t *tri, int64_t n, const uint8_t *in_s, uint8_t *wgt);
void phi_launch_path_cover(hipStream_t st, bool clear, const phi_ent_t *segs, int32_t n_seg, const int64_t *g_off, const uint32_t *tri, const uint8_t *wgt,
                           int32_t *cov_all, int32_t *cov_w, unsigned long long *ctr, uint32_t *twice, int64_t twice_cap);
void phi_launch_uncovered_slots(hipStream_t st, const uint32_t *slots, int64_t n, const int32_t *cov_all, unsigned long long *ctr, uint32_t *out);
// more than 64 walks: the blocks' rows on class lanes (dp_events.hip)
struct PhiBlkClassArgs {
    int32_t n_blk, n_walks, lane_stride;
    const int32_t *blk_lo, *blk_ev;      // [n_blk + 1], [n_blk][lane_stride]
    const int64_t *ev_off, *walk_off;
    const void *ev;                      // this run's event records
    int32_t *lane_walk;                  // out [n_blk][64]: the walk that plays class lane l (-1: none)
    int32_t *walk_lane;                  // out [n_blk][lane_stride]: the class lane of every walk
    int32_t *coff;                       // out [n_blk][lane_stride]: the walk's offset from the walk that plays its class lane
    int32_t *blk_ncls;                   // out [n_blk]
    uint32_t *err;                       // PHI_KERR_DP_CLASSES
};
void phi_launch_blk_classes(hipStream_t st, const PhiBlkClassArgs &G);
void phi_launch_blk_chain(hipStream_t st, const PhiBlkClassArgs &G, const int32_t *rows, const int32_t *rownew, const int32_t *rowdiag, int32_t *blk_S);
void phi_launch_blk_chain_segments(hipStream_t st, const PhiBlkClassArgs &G, const int32_t *rows, const int32_t *rownew, const int32_t *rowdiag, int32_t *blk_S,
                                   int32_t n_seg, const int32_t *d_seg_lo, int32_t *seg_row, int32_t *seg_S);
void phi_launch_blk_check(hipStream_t st, const int32_t *keys, const int32_t *S, int32_t n_blk, int32_t LS, int32_t n_walks, int32_t *bad);
void phi_launch_carry_resolve(hipStream_t st, const int32_t *carry, int32_t LS, int32_t b_from, int32_t h, int32_t *out);
void phi_launch_dp_block_paths_wide(hipStream_t st, const PhiDpEventArgs &A);
void phi_launch_cut_cov(hipStream_t st, const phi_ent_t *a_e1, const uint8_t *a_span, int64_t n_a, int32_t *diff);
void phi_launch_cut_clean(hipStream_t st, const int32_t *cov_excl, int64_t n_entries, int32_t *clean);
void phi_launch_cut_clean_direct(hipStream_t st, const int64_t *g_off, const uint8_t *g_span, int64_t n_entries, int32_t *clean);
void phi_launch_cut_events(hipStream_t st, const phi_ent_t *ev_e, int64_t n_ev, const int64_t *ev_off, const int64_t *walk_off, int32_t n_walks,
                           const int32_t *walk_vtx, const int32_t *cvtx, const int32_t *ncl_excl, int32_t *stepdiff);
void phi_launch_blk_ev(hipStream_t st, const int32_t *blk_lo, int32_t n_blk, const phi_ent_t *ev_e, const int64_t *ev_off, int32_t n_walks,
                       const int32_t *walk_vtx, const int32_t *cvtx, int32_t *blk_ev);
void phi_launch_event_flags(hipStream_t st, const int32_t *walk_vtx, int64_t n_entries, const int32_t *cvtx, uint8_t *flags);
void phi_launch_event_off(hipStream_t st, const phi_ent_t *ev_e, int64_t n_ev, const int64_t *walk_off, int32_t n_walks,
                          int64_t *ev_off);
int64_t phi_scan_i32_num_blocks(int64_t n);
void phi_launch_scan_i32(hipStream_t st, const int32_t *cnt, int64_t n, int32_t *off, int32_t *blk, int64_t *blk_off);
// same with 64-bit sums: off[0..n] int64, blk int64 scratch
void phi_launch_scan_i64(hipStream_t st, const int32_t *cnt, int64_t n, int64_t *off, int64_t *blk, int64_t *blk_off);
void phi_launch_scan_sums_i64(hipStream_t st, const int64_t *v, int64_t n, int64_t *off);
void phi_launch_dp_event_fill(hipStream_t st, const PhiDpEventArgs &A, const uint8_t *e_out, const int32_t *walk_vtx,
                              const int32_t *cvtx, const phi_ent_t *a_e1, const int32_t *wpre, int64_t n_entries);
void phi_launch_scan_u8(hipStream_t st, const uint8_t *cnt, int64_t n, int32_t *off, int32_t *blk, int64_t *blk_off);
int phi_dp_num_waves(int n_walks);
void phi_launch_dp_words(hipStream_t st, const uint8_t *e_out, const int64_t *g_off, const uint8_t *g_span,
                         const uint8_t *a_weight, int64_t n_entries, uint64_t *word);

// reads_text.hip: the records of a FASTA / FASTQ text found on the device
#define PHI_TEXT_IRREGULAR_CR 1u       // a carriage return in the chunk
#define PHI_TEXT_IRREGULAR_LINES 2u    // more lines than the line tables hold
#define PHI_TEXT_IRREGULAR_LAYOUT 4u   // a line that does not fit the regular layout (first_bad = its index)
struct PhiTextSummary {
    uint32_t n_nl;                       // line feeds = whole lines in the buffer
    uint32_t n_rec;                      // whole records taken
    uint32_t err;                        // PHI_TEXT_IRREGULAR_*
    uint32_t first_bad;
    uint64_t n_bases;                    // their sequence bytes
    uint32_t cons_end;                   // buffer offset where the carry (the bytes not taken) begins
    uint32_t n_cons_lines;
    uint32_t not_uniform;                // some record's sequence is of another length than the first one's
    uint32_t pad_;
};
struct PhiTextArgs {
    const uint8_t *buf;                  // device buffer [carry | chunk]
    uint32_t start, end;                 // the text is buf[start, end)
    int32_t mode;                        // 0 FASTA, 1 FASTQ with four lines per record
    uint32_t line_cap;                   // lines the tables below hold
    uint32_t *tile_cnt;                  // [tiles + 1]
    uint32_t *ls;                        // [line_cap + 2] line starts
    uint64_t *pre;                       // [line_cap + 2] per line: records begun << 32 | sequence bytes, then their exclusive prefix sums
    uint64_t *blk;                       // [phi_text_scan_blocks(line_cap)] scratch of the scan
    int64_t *read_off;                   // out [records + 1]
    uint8_t *bases;                      // out: the sequence bytes, back to back
    PhiTextSummary *sum;
};
uint32_t phi_text_num_tiles(uint32_t start, uint32_t end);
uint32_t phi_text_scan_blocks(uint32_t line_cap);
void phi_launch_reads_text(hipStream_t st, const PhiTextArgs &A);

// code-object warm-up, one per translation unit (phi_ctx_create)
void phi_warm_sketch(hipStream_t st);
void phi_warm_table(hipStream_t st);
void phi_warm_anchors(hipStream_t st);
void phi_warm_contexts(hipStream_t st);
void phi_warm_dp(hipStream_t st);
void phi_warm_dp_events(hipStream_t st);
void phi_warm_solve_dev(hipStream_t st);
void phi_warm_reads_text(hipStream_t st);
void phi_warm_walk_text(hipStream_t st);
