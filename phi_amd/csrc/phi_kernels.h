// phi_kernels.h -- launch interface between the C-ABI layer (phi_abi.hip) and the kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define PHI_TPB 256            // lanes per workgroup of the sketch kernel (4 waves)
#define PHI_CH 2048            // window positions per workgroup
#define PHI_MAX_W 256
#define PHI_MAX_K 32
#define PHI_MAX_PROBE 4096     // linear-probe bound of the open-addressed tables
#define PHI_RCAP 32            // DP run-length states 0..31 (an anchor spans <= k-1 <= 31 edges)

enum { PHI_MODE_COUNT = 0, PHI_MODE_WRITE = 1, PHI_MODE_PROBE = 2 };

// bits of the device error word
#define PHI_KERR_SENTINEL 1u    // a minimiser hashed to the empty-slot sentinel
#define PHI_KERR_TABLE_FULL 2u  // probe bound exceeded
#define PHI_KERR_WALK_EDGE 4u   // consecutive walk vertices not joined by a graph edge

struct PhiSketchArgs {
    const uint64_t *words;                 // packed bases (+2 padding words)
    const unsigned long long *starts;      // sequence-start bitmap
    int64_t n_bases;
    int32_t k, w;
    // PHI_MODE_COUNT / PHI_MODE_WRITE
    int32_t *block_cnt;
    const int64_t *block_off;
    uint64_t *out_hash;
    int64_t *out_pos;
    // PHI_MODE_PROBE
    uint64_t *sp_keys; uint64_t sp_mask;   // read spectrum set
    unsigned long long *sp_count;
    unsigned long long *n_emitted;
    const uint64_t *u_keys; const uint32_t *u_uid; uint64_t u_mask;   // walk-minimiser table: slot -> dense id
    uint8_t *hit;                          // per distinct walk minimiser (dense id)
    uint32_t *err;
};

// sketch.hip
void phi_launch_pack_ascii(hipStream_t st, const uint8_t *bases, int64_t n, uint64_t *words, int64_t n_words,
                           unsigned long long *n_bad);
void phi_launch_mark_starts(hipStream_t st, const int64_t *seq_off, int64_t n_seq, unsigned long long *starts);
void phi_launch_pack_walks(hipStream_t st, const uint8_t *seq_concat, const int64_t *seq_off,
                           const int32_t *walk_vtx, const int64_t *ebase, int64_t n_entries, uint64_t *words,
                           int64_t n_words, unsigned long long *n_bad);
int64_t phi_sketch_num_blocks(int64_t n_bases);
void phi_launch_sketch(hipStream_t st, int mode, const PhiSketchArgs &A);
void phi_launch_scan_counts(hipStream_t st, const int32_t *cnt, int64_t n, int64_t *off);

// table.hip
// Insert every walk-minimiser hash; u_rep[slot] = smallest record index holding that hash
// (deterministic on every rank); rec_slot[i] = slot of record i.
void phi_launch_table_build(hipStream_t st, const uint64_t *rec_hash, int64_t n_rec, uint64_t *u_keys,
                            uint32_t *u_rep, uint64_t u_mask, uint32_t *rec_slot, uint32_t *err);
void phi_launch_rep_flags(hipStream_t st, const uint32_t *rec_slot, int64_t n_rec, const uint32_t *u_rep,
                          uint8_t *flags);
void phi_launch_slot_uid(hipStream_t st, const int32_t *rep_list, int64_t n_unique, const uint32_t *rec_slot,
                         uint32_t *u_uid);
void phi_launch_fill_u64(hipStream_t st, uint64_t *p, int64_t n, uint64_t v);
void phi_launch_fill_u32(hipStream_t st, uint32_t *p, int64_t n, uint32_t v);
// insert a list of hashes into the spectrum set (multi-GPU spectrum merge)
void phi_launch_spectrum_insert(hipStream_t st, const uint64_t *hashes, int64_t n, uint64_t *sp_keys,
                                uint64_t sp_mask, unsigned long long *sp_count, uint32_t *err);
// compact the occupied slots of the spectrum set into a list; *n_out receives the count
void phi_launch_spectrum_export(hipStream_t st, const uint64_t *sp_keys, int64_t cap, uint64_t *out,
                                unsigned long long *n_out);

// anchors.hip
#define PHI_KERR_FP_COLLISION 8u   // two different vertex lists share a fingerprint: reseed
void phi_launch_locate(hipStream_t st, const int64_t *rec_pos, int64_t n_rec, const int64_t *ebase,
                       int64_t n_entries, int32_t k, int32_t *rec_e0, int32_t *rec_e1);
void phi_launch_lower_bound(hipStream_t st, const int64_t *a, int64_t n, const int64_t *keys, int64_t m,
                            int64_t *out);
int64_t phi_compact_num_blocks(int64_t n);
void phi_launch_flag_count(hipStream_t st, const uint8_t *flags, int64_t n, int32_t *block_cnt);
void phi_launch_flag_write(hipStream_t st, const uint8_t *flags, int64_t n, const int64_t *block_off, int32_t *out);
void phi_launch_match_flags(hipStream_t st, const uint32_t *rec_slot, int64_t n_rec, const uint32_t *u_uid,
                            const uint8_t *hit, uint8_t *flags);

struct PhiFilterArgs {
    const uint32_t *rec_slot; const int32_t *rec_e0; const int32_t *rec_e1;   // per walk record
    const int32_t *walk_vtx;
    const int32_t *m_rec;                          // matched records (ascending)
    uint64_t *g_keys; int32_t *g_rep; uint32_t *g_cnt; uint64_t g_mask; uint64_t seed;   // group table
    int32_t *m_group;                              // group slot of each matched anchor
    uint32_t *slot_maxcnt;                         // per minimiser-table slot: largest group
    uint8_t *slot_multi;                           // per slot: has an anchor spanning >= 2 vertices
    float limit;                                   // threshold * num_walks (ILP_index.cpp:698)
    unsigned long long *counters;                  // [0] filtered  [1] in model
    uint32_t *err;
};
void phi_launch_group_insert(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched);
void phi_launch_group_count(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched);
void phi_launch_group_max(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched);
void phi_launch_slot_count(hipStream_t st, const PhiFilterArgs &A, int64_t u_cap);
void phi_launch_kept_flags(hipStream_t st, const PhiFilterArgs &A, int64_t n_matched, uint8_t *kept, uint8_t *dp);
void phi_launch_gather_i32(hipStream_t st, const int32_t *src, const int32_t *idx, int64_t n, int32_t *out);
void phi_launch_gather_u64(hipStream_t st, const uint64_t *src, const int32_t *idx, int64_t n, uint64_t *out);
void phi_launch_entry_csr(hipStream_t st, const int32_t *a_e1, int64_t n_a, int64_t n_entries, int64_t *g_off);

// dp.hip
struct PhiDpArgs {
    int32_t n_vtx, n_walks;
    const int32_t *topo;                 // vertices in topological order
    const int64_t *in_off; const int32_t *in_src;   // reverse adjacency (by vertex)
    const int64_t *walk_off; const int32_t *walk_vtx;
    // dp anchors sorted by last entry: CSR over walk entries
    const int64_t *g_off; const uint8_t *g_span;
    const uint8_t *a_weight;             // weight (0/1) of each dp anchor in this run
    int32_t cost;                        // 2 * (R / 2)
    // outputs
    int32_t *dmax;                       // per entry: best score of a path ending there
    uint8_t *qbest;                      // per entry: run length attaining it (ties: longest)
    int32_t *lent;                       // per entry: walk index where the capped run began
    int32_t *top1v, *top1h, *top1n, *top2v, *top2h;   // per vertex: best leaving states by next vertex
    int32_t *ent_v, *ent_u, *ent_h;      // per vertex: recombination entry value and its source
};
void phi_launch_dp(hipStream_t st, const PhiDpArgs &A);
