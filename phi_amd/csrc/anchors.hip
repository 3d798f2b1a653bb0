// anchors.hip -- anchor extraction and the shared-anchor filter on the GPU.
//
// Replaces, from the reference:
//   the base->vertex map and unique-vertex list of index_kmers   src/ILP_index.cpp:374-381, 416-438
//   compute_anchors                                              src/ILP_index.cpp:495-526, 643-655
//   the shared-anchor filter                                     src/ILP_index.cpp:670-743
//
// A walk follows graph edges of a DAG, so the unique vertices under a k-mer, sorted by
// topological rank (:432-435), are consecutive walk entries: an anchor is (walk, first entry,
// last entry) and its "v1_v2_..._" key (:680-683) is the vertex sequence walk_vtx[e0..e1].
// Groups of equal keys are counted in an open-addressed table keyed by a seeded 64-bit
// fingerprint of (minimiser, vertex list); every hit is verified against the group's
// representative anchor, so a fingerprint collision is detected (PHI_KERR_FP_COLLISION, the
// caller reseeds) instead of merging two groups.
#include <hip/hip_runtime.h>
#include "phi_dev.h"
#include "phi_kernels.h"

static inline unsigned grid_for(int64_t n, int tpb)
{
    int64_t nb = (n + tpb - 1) / tpb;
    if (nb > 256 * 16) nb = 256 * 16;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------- walk entries -> out-edges
// For every walk entry: the index of the graph edge the walk leaves its vertex by (255 = the walk ends
// here), the number of walks crossing every edge, and (st_mask != null) the walks present on every
// vertex -- what ILP_index::read_gfa keeps in `paths` / `haps` (ILP_index.cpp:85-113).  err[0] = first
// failure: 1 empty segment, 2 step without a graph edge, 3 more than 254 out-edges, 4 vertex out of range; err[1..3] = walk, u, v.
__global__ void __launch_bounds__(256) phi_walk_edges_kernel(const int32_t *__restrict__ walk_vtx, const int64_t *__restrict__ walk_off,
                                                             int32_t n_walks, int64_t n_entries, int32_t n_vtx, const int64_t *__restrict__ adj_off,
                                                             const int32_t *__restrict__ adj, const int64_t *__restrict__ seq_off,
                                                             const int32_t *__restrict__ topo_rank, uint8_t *__restrict__ e_out,
                                                             int32_t *__restrict__ cnt_edge, unsigned long long *__restrict__ st_mask,
                                                             int32_t nw64, int32_t *__restrict__ err)
{
    GRID_STRIDE(e, n_entries) {
        int32_t lo = 0, hi = n_walks;                  // walk of entry e: walk_off[lo] <= e < walk_off[lo + 1]
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if (walk_off[mid] <= e) lo = mid; else hi = mid;
        }
        const int32_t h = lo;
        const int32_t u = walk_vtx[e];
        auto fail = [&](int32_t code, int32_t v) {
            if (atomicCAS(&err[0], 0, code) == 0) { err[1] = h; err[2] = u; err[3] = v; }
        };
        if ((uint32_t)u >= (uint32_t)n_vtx) { fail(4, -1); e_out[e] = 255; continue; }        // (the range check of the walk entries: nothing indexed with u before this)
        if (seq_off[u + 1] == seq_off[u]) { fail(1, -1); e_out[e] = 255; continue; }
        if (st_mask) atomicOr(&st_mask[(size_t)topo_rank[u] * nw64 + (h >> 6)], 1ull << (h & 63));
        if (e + 1 < walk_off[h + 1]) {
            const int32_t v = walk_vtx[e + 1];
            int64_t x = adj_off[u];
            const int64_t xe = adj_off[u + 1];
            while (x < xe && adj[x] != v) x++;
            if (x == xe) { fail(2, v); e_out[e] = 255; continue; }
            if (x - adj_off[u] >= 255) { fail(3, v); e_out[e] = 255; continue; }
            e_out[e] = (uint8_t)(x - adj_off[u]);
            atomicAdd(&cnt_edge[x], 1);
        } else {
            e_out[e] = 255;
        }
    }
}

void phi_launch_walk_edges(hipStream_t st, const int32_t *walk_vtx, const int64_t *walk_off, int32_t n_walks, int64_t n_entries, int32_t n_vtx,
                           const int64_t *adj_off, const int32_t *adj, const int64_t *seq_off, const int32_t *topo_rank,
                           uint8_t *e_out, int32_t *cnt_edge, unsigned long long *st_mask, int32_t nw64, int32_t *err)
{
    if (n_entries > 0)
        hipLaunchKernelGGL(phi_walk_edges_kernel, dim3(grid_for(n_entries, 256)), dim3(256), 0, st, walk_vtx, walk_off, n_walks,
                           n_entries, n_vtx, adj_off, adj, seq_off, topo_rank, e_out, cnt_edge, st_mask, nw64, err);
}

// ------------------------------------------------------------------------- minimiser -> anchors (CSR)
// The certificate on the host walks the anchors of each minimiser; grouping 10^6-10^7 anchors by
// minimiser id is a scatter the GPU does in microseconds: count per id, scan (phi_launch_scan_i32),
// scatter through atomic cursors, then put every short list in ascending anchor order (deterministic).
__global__ void __launch_bounds__(256) phi_csr_count_kernel(const uint32_t *__restrict__ triples, int64_t n, int64_t n_ids,
                                                            int32_t *__restrict__ cnt, uint32_t *__restrict__ err)
{
    GRID_STRIDE(i, n) {
        const uint32_t id = (uint32_t)triples[3 * i];
        if ((int64_t)id >= n_ids) { atomicOr(err, PHI_KERR_CSR_ID); continue; }
        atomicAdd(&cnt[id], 1);
    }
}
__global__ void __launch_bounds__(256) phi_csr_scatter_kernel(const uint32_t *__restrict__ triples, int64_t n, int64_t n_ids,
                                                              const int32_t *__restrict__ off, int32_t *__restrict__ cur,
                                                              int32_t *__restrict__ idx)
{
    GRID_STRIDE(i, n) {
        const uint32_t id = (uint32_t)triples[3 * i];
        if ((int64_t)id >= n_ids) continue;
        idx[off[id] + atomicAdd(&cur[id], 1)] = (int32_t)i;
    }
}
__global__ void __launch_bounds__(256) phi_csr_sort_kernel(const int32_t *__restrict__ off, int64_t n_ids, int32_t *__restrict__ idx)
{
    GRID_STRIDE(s, n_ids) {
        const int32_t lo = off[s], n = off[s + 1] - lo;
        for (int32_t a = 1; a < n; a++) {             // insertion sort: lists hold a few entries (at most walks x repeats)
            const int32_t v = idx[lo + a];
            int32_t b = a - 1;
            while (b >= 0 && idx[lo + b] > v) { idx[lo + b + 1] = idx[lo + b]; b--; }
            idx[lo + b + 1] = v;
        }
    }
}

void phi_launch_csr_count(hipStream_t st, const uint32_t *triples, int64_t n, int64_t n_ids, int32_t *cnt, uint32_t *err)
{
    if (n > 0) hipLaunchKernelGGL(phi_csr_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, triples, n, n_ids, cnt, err);
}
void phi_launch_csr_scatter(hipStream_t st, const uint32_t *triples, int64_t n, int64_t n_ids, const int32_t *off, int32_t *cur,
                            int32_t *idx)
{
    if (n > 0) hipLaunchKernelGGL(phi_csr_scatter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, triples, n, n_ids, off, cur, idx);
}
void phi_launch_csr_sort(hipStream_t st, const int32_t *off, int64_t n_ids, int32_t *idx)
{
    if (n_ids > 0) hipLaunchKernelGGL(phi_csr_sort_kernel, dim3(grid_for(n_ids, 256)), dim3(256), 0, st, off, n_ids, idx);
}

// ------------------------------------------------------------------------- ordered compaction
// flags[n] (0/1) -> ascending list of the flagged indices.  2048 items per workgroup.
#define CMP_ITEMS 8
__global__ void __launch_bounds__(256) phi_flag_count_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                             int32_t *__restrict__ block_cnt)
{
    __shared__ int s_w[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * CMP_ITEMS;
    int c = 0;
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; j++)
        if (base + j < n) c += flags[base + j] != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(256) phi_flag_write_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                             const int64_t *__restrict__ block_off,
                                                             int32_t *__restrict__ out)
{
    __shared__ int s_w[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * CMP_ITEMS;
    uint32_t f = 0;
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; j++)
        if (base + j < n && flags[base + j]) f |= 1u << j;
    const int c = __popc(f);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int v = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_w[wid] = v;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wid; i++) woff += s_w[i];
    int64_t o = block_off[blockIdx.x] + woff + v - c;
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; j++)
        if (f & (1u << j)) out[o++] = (int32_t)(base + j);
}

int64_t phi_compact_num_blocks(int64_t n) { return (n + 256 * CMP_ITEMS - 1) / (256 * CMP_ITEMS); }

void phi_launch_flag_count(hipStream_t st, const uint8_t *flags, int64_t n, int32_t *block_cnt)
{
    const int64_t nb = phi_compact_num_blocks(n);
    if (nb > 0) hipLaunchKernelGGL(phi_flag_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, flags, n, block_cnt);
}
void phi_launch_flag_write(hipStream_t st, const uint8_t *flags, int64_t n, const int64_t *block_off, int32_t *out)
{
    const int64_t nb = phi_compact_num_blocks(n);
    if (nb > 0)
        hipLaunchKernelGGL(phi_flag_write_kernel, dim3((unsigned)nb), dim3(256), 0, st, flags, n, block_off, out);
}

// ------------------------------------------------------------------------- match
// compute_anchors: a walk minimiser is an anchor iff its hash is in the read spectrum.
__global__ void __launch_bounds__(256) phi_match_flags_kernel(const uint32_t *__restrict__ rec_slot, int64_t n_rec,
                                                              const uint32_t *__restrict__ u_uid,
                                                              const uint8_t *__restrict__ hit,
                                                              uint8_t *__restrict__ flags)
{
    GRID_STRIDE(i, n_rec) flags[i] = hit[u_uid[rec_slot[i]]];
}

void phi_launch_match_flags(hipStream_t st, const uint32_t *rec_slot, int64_t n_rec, const uint32_t *u_uid,
                            const uint8_t *hit, uint8_t *flags)
{
    if (n_rec > 0)
        hipLaunchKernelGGL(phi_match_flags_kernel, dim3(grid_for(n_rec, 256)), dim3(256), 0, st, rec_slot, n_rec,
                           u_uid, hit, flags);
}

// ------------------------------------------------------------------------- filter groups
__device__ __forceinline__ uint64_t group_key(uint32_t slot, const int32_t *__restrict__ walk_vtx, phi_ent_t e0,
                                              phi_ent_t e1, uint64_t seed)
{
    uint64_t h = seed ^ ((uint64_t)slot * 0x9E3779B97F4A7C15ull);
    h = phi_fmix64(h ^ (uint64_t)(e1 - e0 + 1));
    for (phi_ent_t e = e0; e <= e1; e++) h = phi_fmix64(h ^ (uint64_t)(uint32_t)walk_vtx[e]) + 0x632BE59BD9B4E019ull;
    return h == PHI_EMPTY_KEY ? 0 : h;
}

__device__ __forceinline__ bool same_group(const PhiFilterArgs &A, int32_t ra, int32_t rb)
{
    if (A.rec_slot[ra] != A.rec_slot[rb]) return false;
    const phi_ent_t a0 = A.rec_e0[ra], a1 = A.rec_e1[ra], b0 = A.rec_e0[rb], b1 = A.rec_e1[rb];
    if (a1 - a0 != b1 - b0) return false;
    for (uint32_t d = 0; d <= a1 - a0; d++)
        if (A.walk_vtx[a0 + d] != A.walk_vtx[b0 + d]) return false;
    return true;
}

// pass 1: claim a table slot per fingerprint; representative = smallest record index
__global__ void __launch_bounds__(256) phi_group_insert_kernel(PhiFilterArgs A, int64_t n_matched)
{
    GRID_STRIDE(j, n_matched) {
        const int32_t r = A.m_rec[j];
        const uint64_t key = group_key(A.rec_slot[r], A.walk_vtx, A.rec_e0[r], A.rec_e1[r], A.seed);
        uint64_t gs = key & A.g_mask;
        int probes = 0;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long *)&A.g_keys[gs], PHI_EMPTY_KEY, key);
            if (prev == PHI_EMPTY_KEY || prev == key) break;
            gs = (gs + 1) & A.g_mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(A.err, PHI_KERR_TABLE_FULL); break; }
        }
        atomicMin((uint32_t *)&A.g_rep[gs], (uint32_t)r);
        A.m_group[j] = (int32_t)gs;
    }
}

// pass 2: verify against the representative, count the group (the map's .first, :686-689)
__global__ void __launch_bounds__(256) phi_group_count_kernel(PhiFilterArgs A, int64_t n_matched)
{
    GRID_STRIDE(j, n_matched) {
        const int32_t r = A.m_rec[j];
        const int32_t gs = A.m_group[j];
        if (!same_group(A, r, A.g_rep[gs])) { atomicOr(A.err, PHI_KERR_FP_COLLISION); continue; }
        atomicAdd(&A.g_cnt[gs], (uint32_t)A.cls_mult[A.rec_cls[r]]);   // every walk entry of the class carries this anchor (:686-689)
    }
}

// pass 3: per minimiser, the largest group and whether any anchor spans >= 2 vertices
__global__ void __launch_bounds__(256) phi_group_max_kernel(PhiFilterArgs A, int64_t n_matched)
{
    GRID_STRIDE(j, n_matched) {
        const int32_t r = A.m_rec[j];
        const uint32_t id = A.u_uid[A.rec_slot[r]];          // (per dense id, not per slot of the 8x table: 30x smaller arrays)
        atomicMax(&A.slot_maxcnt[id], A.g_cnt[A.m_group[j]]);
        if (A.rec_e1[r] > A.rec_e0[r]) A.slot_multi[id] = 1;
    }
}

// pass 4: filtered / in-model counters over the distinct minimisers (ILP_index.cpp:698, :711, :822/:868)
__global__ void __launch_bounds__(256) phi_slot_count_kernel(PhiFilterArgs A, int64_t n_ids)
{
    int n_filtered = 0, n_model = 0;
    GRID_STRIDE(s, n_ids) {
        const uint32_t c = A.slot_maxcnt[s];
        if (c == 0) continue;                               // no anchor for this minimiser
        if ((float)c >= A.limit) n_filtered++;
        else if (A.slot_multi[s]) n_model++;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        n_filtered += __shfl_xor(n_filtered, d, 64);
        n_model += __shfl_xor(n_model, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_filtered) atomicAdd(&A.counters[0], (unsigned long long)n_filtered);
        if (n_model) atomicAdd(&A.counters[1], (unsigned long long)n_model);
    }
}

// pass 5: kept anchors (:704-709); dp anchors additionally span >= 2 vertices (:795/:846)
__global__ void __launch_bounds__(256) phi_kept_flags_kernel(PhiFilterArgs A, int64_t n_matched,
                                                             uint8_t *__restrict__ kept, uint8_t *__restrict__ dp)
{
    GRID_STRIDE(j, n_matched) {
        const int32_t r = A.m_rec[j];
        const bool keep = !((float)A.slot_maxcnt[A.u_uid[A.rec_slot[r]]] >= A.limit);
        kept[j] = keep;
        dp[j] = keep && A.rec_e1[r] > A.rec_e0[r];
    }
}

void phi_launch_group_insert(hipStream_t st, const PhiFilterArgs &A, int64_t n)
{
    if (n > 0) hipLaunchKernelGGL(phi_group_insert_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, A, n);
}
void phi_launch_group_count(hipStream_t st, const PhiFilterArgs &A, int64_t n)
{
    if (n > 0) hipLaunchKernelGGL(phi_group_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, A, n);
}
void phi_launch_group_max(hipStream_t st, const PhiFilterArgs &A, int64_t n)
{
    if (n > 0) hipLaunchKernelGGL(phi_group_max_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, A, n);
}
void phi_launch_slot_count(hipStream_t st, const PhiFilterArgs &A, int64_t u_cap)
{
    if (u_cap > 0) hipLaunchKernelGGL(phi_slot_count_kernel, dim3(grid_for(u_cap, 256)), dim3(256), 0, st, A, u_cap);
}
void phi_launch_kept_flags(hipStream_t st, const PhiFilterArgs &A, int64_t n, uint8_t *kept, uint8_t *dp)
{
    if (n > 0) hipLaunchKernelGGL(phi_kept_flags_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, A, n, kept, dp);
}

// ------------------------------------------------------------------------- gathers for the DP
// out[j] = src[idx[j]]
__global__ void phi_gather_i32_kernel(const int32_t *__restrict__ src, const uint32_t *__restrict__ idx, int64_t n,
                                      int32_t *__restrict__ out)
{
    GRID_STRIDE(j, n) out[j] = src[idx[j]];
}
void phi_launch_gather_i32(hipStream_t st, const int32_t *src, const uint32_t *idx, int64_t n, int32_t *out)
{
    if (n > 0) hipLaunchKernelGGL(phi_gather_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, src, idx, n, out);
}

__global__ void phi_gather_u64_kernel(const uint64_t *__restrict__ src, const int32_t *__restrict__ idx, int64_t n,
                                      uint64_t *__restrict__ out)
{
    GRID_STRIDE(j, n) out[j] = src[idx[j]];
}
void phi_launch_gather_u64(hipStream_t st, const uint64_t *src, const int32_t *idx, int64_t n, uint64_t *out)
{
    if (n > 0) hipLaunchKernelGGL(phi_gather_u64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, src, idx, n, out);
}

// CSR over walk entries of the dp anchors, which arrive sorted by their last entry e1:
// g_off[e] = first anchor with e1 >= e.
__global__ void phi_entry_csr_kernel(const phi_ent_t *__restrict__ a_e1, int64_t n_a, int64_t n_entries,
                                     int64_t *__restrict__ g_off)
{
    GRID_STRIDE(j, n_a + 1) {
        const int64_t lo = j == 0 ? 0 : (int64_t)a_e1[j - 1] + 1;
        const int64_t hi = j == n_a ? n_entries : (int64_t)a_e1[j];
        for (int64_t e = lo; e <= hi; e++) g_off[e] = j;
    }
}
void phi_launch_entry_csr(hipStream_t st, const phi_ent_t *a_e1, int64_t n_a, int64_t n_entries, int64_t *g_off)
{
    hipLaunchKernelGGL(phi_entry_csr_kernel, dim3(grid_for(n_a + 1, 256)), dim3(256), 0, st, a_e1, n_a, n_entries,
                       g_off);
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_anchors_kernel() {}
void phi_warm_anchors(hipStream_t st) { hipLaunchKernelGGL(phi_warm_anchors_kernel, dim3(1), dim3(64), 0, st); }
