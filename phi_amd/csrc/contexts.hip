// contexts.hip -- graph-side sketch de-duplication.
//
// The reference sketches every haplotype walk on its own (src/ILP_index.cpp:559-573 calling
// index_kmers :359-445): 49-200 walks that share almost all of their sequence are hashed 49-200
// times.  Here every DISTINCT walk context is sketched once.
//
// Which minimiser records index_kmers emits for the windows that START inside walk entry e (vertex
// v = walk_vtx[e]) is a function of
//   * the last base of the entry before e (the window starting there is the predecessor of v's first
//     window, and a record is emitted iff the window's hash differs from its predecessor's, :413),
//   * the bases of v, and
//   * the next w + k - 2 bases of the walk (a window spans w + k - 1 bases), i.e. the following entries
//     e+1 .. e+n up to that many bases, or up to the end of the walk.
// Entries (of any walk) with the same (left base, v, following vertices) form a CLASS: one representative
// per class is laid out in a flat "class space" -- [left base] v's bases [next w+k-2 bases] -- and sketched
// by the same kernels that sketch reads (sketch.hip); the record of the left base's own window is dropped.
// A class record holds (hash, position relative to v's first base, first / last entry offset of the
// vertices under the k-mer); the records of walk h are the records of the classes of its entries, in entry
// order, shifted by the entry's base offset -- never materialised per walk except on demand
// (phi_walk_minimizers) and for the anchors the filter keeps (phi_solve).
//
// Class identity is exact: an open-addressed table keyed by a seeded 64-bit fingerprint, every entry then
// VERIFIED against its class representative vertex by vertex; a fingerprint collision raises
// PHI_KERR_FP_COLLISION and the caller reseeds (same scheme as the anchor groups of anchors.hip).
#include <hip/hip_runtime.h>
#include "phi_dev.h"
#include "phi_kernels.h"

static inline unsigned grid_for(int64_t n, int tpb)
{
    int64_t nb = (n + tpb - 1) / tpb;
    if (nb > 256 * 32) nb = 256 * 32;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// walk of entry e: walk_off[h] <= e < walk_off[h + 1]
__device__ __forceinline__ int32_t walk_of(const int64_t *__restrict__ walk_off, int32_t n_walks, int64_t e)
{
    int32_t lo = 0, hi = n_walks;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (walk_off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------- vertex lengths
__global__ void __launch_bounds__(256) phi_vlen_kernel(const int64_t *__restrict__ seq_off, int64_t n_vtx, int32_t *__restrict__ vlen)
{
    GRID_STRIDE(v, n_vtx) vlen[v] = (int32_t)(seq_off[v + 1] - seq_off[v]);
}
void phi_launch_vlen(hipStream_t st, const int64_t *seq_off, int64_t n_vtx, int32_t *vlen)
{
    if (n_vtx > 0) hipLaunchKernelGGL(phi_vlen_kernel, dim3(grid_for(n_vtx, 256)), dim3(256), 0, st, seq_off, n_vtx, vlen);
}

// ------------------------------------------------------------------------- per-walk sums over the entries
// out[h] += sum over the entries e of walk h of val(e): val = vlen[walk_vtx[e]] (bases of a walk), or
// cnt[ent_cls[e]] (its minimisers, "Number of Minimizers" ILP_index.cpp:563).  One atomic per wave unless a
// wave straddles two walks.
template <bool BY_CLASS>
__global__ void __launch_bounds__(256) phi_walk_sum_kernel(const int32_t *__restrict__ key, const int32_t *__restrict__ val_of,
                                                           const int32_t *__restrict__ cls_rec_off, const int64_t *__restrict__ walk_off,
                                                           int32_t n_walks, int64_t n_entries, unsigned long long *__restrict__ out)
{
    // every wave takes one contiguous slice of the entries (a walk is a contiguous range: a slice meets one
    // or two of them), sums in registers and adds once per walk it met: thousands of atomics on a few dozen
    // addresses would serialise at ~12 ns each
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t per = ((n_entries + n_waves - 1) / n_waves + 63) & ~(int64_t)63;
    const int64_t lo = gw * per, hi = lo + per < n_entries ? lo + per : n_entries;
    if (lo >= hi) return;
    int32_t h = walk_of(walk_off, n_walks, lo);
    int64_t h_end = walk_off[h + 1];
    long long acc = 0;
    for (int64_t e0 = lo; e0 < hi; e0 += 64) {
        const int64_t e = e0 + lane;
        long long v = 0;
        if (e < hi) {
            const int32_t x = key[e];
            v = BY_CLASS ? (cls_rec_off[x + 1] - cls_rec_off[x]) : val_of[x];
        }
        if (e0 + 64 <= h_end) acc += v;                      // the whole row lies in walk h
        else {
            // the row crosses into the next walk(s): flush walk by walk
            int64_t row_lo = e0;
            for (;;) {
                const long long mine = (e >= row_lo && e < h_end) ? v : 0;
                acc += mine;
                if (h_end >= e0 + 64 || h_end >= hi) break;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
                if (lane == 0 && acc) atomicAdd(&out[h], (unsigned long long)acc);
                acc = 0;
                row_lo = h_end;
                h++;
                h_end = walk_off[h + 1];
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0 && acc) atomicAdd(&out[h], (unsigned long long)acc);
}
static inline unsigned walk_sum_grid(int64_t n_entries)
{
    int64_t nb = (n_entries + 256 * 64 - 1) / (256 * 64);       // >= 64 rows per wave
    if (nb > 1024) nb = 1024;
    return (unsigned)(nb < 1 ? 1 : nb);
}
void phi_launch_walk_bases(hipStream_t st, const int32_t *walk_vtx, const int32_t *vlen, const int64_t *walk_off, int32_t n_walks,
                           int64_t n_entries, unsigned long long *out)
{
    if (n_entries > 0)
        hipLaunchKernelGGL(phi_walk_sum_kernel<false>, dim3(walk_sum_grid(n_entries)), dim3(256), 0, st, walk_vtx, vlen, nullptr,
                           walk_off, n_walks, n_entries, out);
}
void phi_launch_walk_rec_counts(hipStream_t st, const int32_t *ent_cls, const int32_t *cls_rec_off, const int64_t *walk_off,
                                int32_t n_walks, int64_t n_entries, unsigned long long *out)
{
    if (n_entries > 0)
        hipLaunchKernelGGL(phi_walk_sum_kernel<true>, dim3(walk_sum_grid(n_entries)), dim3(256), 0, st, ent_cls, nullptr, cls_rec_off,
                           walk_off, n_walks, n_entries, out);
}

// ------------------------------------------------------------------------- the context of an entry
struct EntCtx { int32_t left; int32_t n; int32_t tail; };     // left: byte before the entry or 256 at a walk's start

__device__ __forceinline__ EntCtx entry_context(const PhiClassArgs &A, int64_t e, int64_t w_lo, int64_t w_hi)
{
    EntCtx c;
    c.left = 256;
    if (e > w_lo) {
        const int32_t pv = A.walk_vtx[e - 1];
        const int64_t pe = A.seq_off[pv + 1];
        c.left = pe > A.seq_off[pv] ? (int32_t)A.seq[pe - 1] : 257;    // 257: empty segment (rejected by the walk-entry pass)
    }
    c.n = 0; c.tail = 0;
    while (c.tail < A.tail_need && e + c.n + 1 < w_hi) { c.n++; c.tail += A.vlen[A.walk_vtx[e + c.n]]; }
    return c;
}

__device__ __forceinline__ uint64_t context_key(const PhiClassArgs &A, int64_t e, const EntCtx &c)
{
    uint64_t h = A.seed ^ ((uint64_t)(uint32_t)c.left * 0x9E3779B97F4A7C15ull);
    h = phi_fmix64(h ^ (uint64_t)(uint32_t)c.n);
    for (int32_t j = 0; j <= c.n; j++) h = phi_fmix64(h ^ (uint64_t)(uint32_t)A.walk_vtx[e + j]) + 0x632BE59BD9B4E019ull;
    return h == PHI_EMPTY_KEY ? 0 : h;
}

// pass 1: every entry claims the table slot of its fingerprint; representative = smallest entry index
// (the same on every rank).  Most entries find their key in place: look before the atomics.
__global__ void __launch_bounds__(256) phi_class_insert_kernel(PhiClassArgs A)
{
    GRID_STRIDE(e, A.n_entries) {
        const int32_t h = walk_of(A.walk_off, A.n_walks, e);
        const EntCtx c = entry_context(A, e, A.walk_off[h], A.walk_off[h + 1]);
        const uint64_t key = context_key(A, e, c);
        uint64_t s = key & A.t_mask;
        int probes = 0;
        for (;;) {
            unsigned long long prev = __builtin_nontemporal_load((const unsigned long long *)&A.t_keys[s]);
            if (prev == PHI_EMPTY_KEY) prev = atomicCAS((unsigned long long *)&A.t_keys[s], PHI_EMPTY_KEY, key);
            if (prev == PHI_EMPTY_KEY || prev == key) break;
            s = (s + 1) & A.t_mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(A.err, PHI_KERR_TABLE_FULL); break; }
        }
        if (__builtin_nontemporal_load(&A.t_rep[s]) > (uint32_t)e) atomicMin(&A.t_rep[s], (uint32_t)e);
        A.ent_slot[e] = (uint32_t)s;
    }
}

// pass 2: verify every entry against its representative (vertex by vertex), count the class, flag the
// representatives
__global__ void __launch_bounds__(256) phi_class_verify_kernel(PhiClassArgs A, uint8_t *__restrict__ is_rep)
{
    GRID_STRIDE(e, A.n_entries) {
        const uint32_t s = A.ent_slot[e];
        const int64_t r = (int64_t)A.t_rep[s];
        is_rep[e] = r == e;
        if (r != e) {
            const int32_t h = walk_of(A.walk_off, A.n_walks, e), hr = walk_of(A.walk_off, A.n_walks, r);
            const int64_t w_hi = A.walk_off[h + 1], r_lo = A.walk_off[hr], r_hi = A.walk_off[hr + 1];
            const EntCtx c = entry_context(A, e, A.walk_off[h], w_hi);
            int32_t rleft = 256;
            if (r > r_lo) {
                const int32_t pv = A.walk_vtx[r - 1];
                const int64_t pe = A.seq_off[pv + 1];
                rleft = pe > A.seq_off[pv] ? (int32_t)A.seq[pe - 1] : 257;
            }
            bool same = rleft == c.left && r + c.n < r_hi;
            // a context cut short by the end of its walk equals only one cut short at the same place
            if (same && c.tail < A.tail_need && r + c.n + 1 != r_hi) same = false;
            for (int32_t j = 0; same && j <= c.n; j++) same = A.walk_vtx[r + j] == A.walk_vtx[e + j];
            if (!same) { atomicOr(A.err, PHI_KERR_FP_COLLISION); continue; }
        }
        atomicAdd(&A.t_mult[s], 1u);
    }
}

// classes in the order of their representatives: slot -> class id, class multiplicity
__global__ void __launch_bounds__(256) phi_class_ids_kernel(const phi_ent_t *__restrict__ cls_rep, int64_t n_cls,
                                                            const uint32_t *__restrict__ ent_slot, const uint32_t *__restrict__ t_mult,
                                                            uint32_t *__restrict__ t_cid, int32_t *__restrict__ cls_mult)
{
    GRID_STRIDE(c, n_cls) {
        const uint32_t s = ent_slot[cls_rep[c]];
        t_cid[s] = (uint32_t)c;
        cls_mult[c] = (int32_t)t_mult[s];
    }
}
// ent_cls may alias ent_slot
__global__ void __launch_bounds__(256) phi_entry_class_kernel(const uint32_t *ent_slot, int64_t n_entries,
                                                              const uint32_t *__restrict__ t_cid, int32_t *ent_cls)
{
    GRID_STRIDE(e, n_entries) ent_cls[e] = (int32_t)t_cid[ent_slot[e]];
}

// bases of every class in class space: [left base] + the vertex + up to tail_need following bases
__global__ void __launch_bounds__(256) phi_class_len_kernel(PhiClassArgs A, const phi_ent_t *__restrict__ cls_rep, int64_t n_cls,
                                                            int32_t *__restrict__ cls_len, uint8_t *__restrict__ cls_left)
{
    GRID_STRIDE(c, n_cls) {
        const int64_t r = cls_rep[c];
        const int32_t h = walk_of(A.walk_off, A.n_walks, r);
        const EntCtx x = entry_context(A, r, A.walk_off[h], A.walk_off[h + 1]);
        const int32_t left = x.left != 256;
        cls_left[c] = (uint8_t)left;
        cls_len[c] = left + A.vlen[A.walk_vtx[r]] + (x.tail < A.tail_need ? x.tail : A.tail_need);
    }
}

void phi_launch_class_insert(hipStream_t st, const PhiClassArgs &A)
{
    if (A.n_entries > 0) hipLaunchKernelGGL(phi_class_insert_kernel, dim3(grid_for(A.n_entries, 256)), dim3(256), 0, st, A);
}
void phi_launch_class_verify(hipStream_t st, const PhiClassArgs &A, uint8_t *is_rep)
{
    if (A.n_entries > 0) hipLaunchKernelGGL(phi_class_verify_kernel, dim3(grid_for(A.n_entries, 256)), dim3(256), 0, st, A, is_rep);
}
void phi_launch_class_ids(hipStream_t st, const phi_ent_t *cls_rep, int64_t n_cls, const uint32_t *ent_slot, int64_t n_entries,
                          const uint32_t *t_mult, uint32_t *t_cid, int32_t *cls_mult, int32_t *ent_cls)
{
    if (n_cls > 0)
        hipLaunchKernelGGL(phi_class_ids_kernel, dim3(grid_for(n_cls, 256)), dim3(256), 0, st, cls_rep, n_cls, ent_slot, t_mult, t_cid, cls_mult);
    if (n_entries > 0)
        hipLaunchKernelGGL(phi_entry_class_kernel, dim3(grid_for(n_entries, 256)), dim3(256), 0, st, ent_slot, n_entries, t_cid, ent_cls);
}
void phi_launch_class_len(hipStream_t st, const PhiClassArgs &A, const phi_ent_t *cls_rep, int64_t n_cls, int32_t *cls_len, uint8_t *cls_left)
{
    if (n_cls > 0) hipLaunchKernelGGL(phi_class_len_kernel, dim3(grid_for(n_cls, 256)), dim3(256), 0, st, A, cls_rep, n_cls, cls_len, cls_left);
}

// ------------------------------------------------------------------------- class space -> packed words
// lane -> 32 bases of class space.  cls_base[c] = first base of class c (cls_base[n_cls] = total).
__global__ void __launch_bounds__(256) phi_pack_classes_kernel(const uint8_t *__restrict__ seq, const int64_t *__restrict__ seq_off,
                                                               const int32_t *__restrict__ walk_vtx, const int32_t *__restrict__ vlen,
                                                               const phi_ent_t *__restrict__ cls_rep, const uint8_t *__restrict__ cls_left,
                                                               const int64_t *__restrict__ cls_base, int64_t n_cls,
                                                               uint64_t *__restrict__ words, int64_t n_words, uint32_t *__restrict__ badbits,
                                                               uint8_t *__restrict__ ascii, unsigned long long *__restrict__ n_bad)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words + 4) return;
    if (wi >= n_words) {
        if (wi < n_words + 2) words[wi] = 0;
        if (badbits) badbits[wi] = 0;
        return;
    }
    const int64_t total = cls_base[n_cls];
    const int64_t b0 = wi * 32;
    uint64_t word = 0;
    uint32_t bad = 0;
    if (b0 < total) {
        int64_t c = phi_locate_in(cls_base, n_cls, b0);
        int64_t c_end = cls_base[c + 1];
        // cursor inside class c: entry ent (from the representative on), whose first base is class offset ent_o
        int64_t ent = -1, ent_o = 0, ent_end = 0;
        const uint8_t *src = nullptr;
        bool fresh = true;
        for (int j = 0; j < 32; j++) {
            const int64_t b = b0 + j;
            uint32_t ch = 'A';
            if (b < total) {
                while (b >= c_end) { c++; c_end = cls_base[c + 1]; fresh = true; }
                const int64_t o = b - cls_base[c];
                const int left = cls_left[c];
                if (left && o == 0) {
                    const int32_t pv = walk_vtx[cls_rep[c] - 1];
                    ch = seq[seq_off[pv + 1] - 1];
                } else {
                    if (fresh) {
                        ent = cls_rep[c]; ent_o = left;
                        const int32_t v = walk_vtx[ent];
                        ent_end = ent_o + vlen[v];
                        src = seq + seq_off[v] - ent_o;
                        fresh = false;
                    }
                    while (o >= ent_end) {
                        ent++; ent_o = ent_end;
                        const int32_t v = walk_vtx[ent];
                        ent_end = ent_o + vlen[v];
                        src = seq + seq_off[v] - ent_o;
                    }
                    ch = src[o];
                }
                bad |= (uint32_t)(!phi_is_acgt(ch)) << j;
                if (ascii) ascii[b] = (uint8_t)ch;
            }
            word = (word << 2) | phi_code(ch);
        }
    }
    words[wi] = word;
    if (badbits) badbits[wi] = bad;
    if (bad) atomicAdd(n_bad, (unsigned long long)__popc(bad));
}

void phi_launch_pack_classes(hipStream_t st, const uint8_t *seq, const int64_t *seq_off, const int32_t *walk_vtx, const int32_t *vlen,
                             const phi_ent_t *cls_rep, const uint8_t *cls_left, const int64_t *cls_base, int64_t n_cls, uint64_t *words,
                             int64_t n_words, uint32_t *badbits, uint8_t *ascii, unsigned long long *n_bad)
{
    if (n_words <= 0) return;
    const int64_t nb = (n_words + 4 + 255) / 256;
    hipLaunchKernelGGL(phi_pack_classes_kernel, dim3((unsigned)nb), dim3(256), 0, st, seq, seq_off, walk_vtx, vlen, cls_rep, cls_left,
                       cls_base, n_cls, words, n_words, badbits, ascii, n_bad);
}

// ------------------------------------------------------------------------- raw records -> class records
// Raw record i = (hash, flat position in class space), in window order.  The first record of a class
// with a left base belongs to the left base's own window (the first window of a sequence is always
// emitted): dropped.  Every other record gets its class, its position relative to the vertex's first base and
// the entries (of the representative) that own the first / last base of its k-mer (the anchor's vertex
// list, ILP_index.cpp:419-438).
__global__ void __launch_bounds__(256) phi_class_rec_kernel(const int64_t *__restrict__ raw_pos, int64_t n_raw,
                                                            const int64_t *__restrict__ cls_base, int64_t n_cls,
                                                            const phi_ent_t *__restrict__ cls_rep, const uint8_t *__restrict__ cls_left,
                                                            const int32_t *__restrict__ walk_vtx, const int32_t *__restrict__ vlen,
                                                            int32_t k, uint8_t *__restrict__ keep, int32_t *__restrict__ r_cls,
                                                            int32_t *__restrict__ r_rel, phi_ent_t *__restrict__ r_e0, phi_ent_t *__restrict__ r_e1)
{
    GRID_STRIDE(i, n_raw) {
        const int64_t p = raw_pos[i];
        const int64_t c = phi_locate_in(cls_base, n_cls, p);
        const int64_t cb = cls_base[c];
        const int left = cls_left[c];
        const bool first = i == 0 || raw_pos[i - 1] < cb;
        const bool kept = !(left && first);
        keep[i] = kept;
        const int32_t rel = (int32_t)(p - cb) - left;            // >= 0 for every kept record (its window starts inside the vertex)
        int64_t e = cls_rep[c];
        if (!kept) {
            // the left base's own window: dropped, and its k-mer may reach past the bases the class's entries own (the
            // walk's last vertex shorter than k - 1: the entries below would be looked for beyond the end of the walks --
            // found by a fuzz case, k = 4 on a last vertex of 3 bases, once the allocation order had changed)
            r_cls[i] = (int32_t)c; r_rel[i] = rel; r_e0[i] = (phi_ent_t)e; r_e1[i] = (phi_ent_t)e;
            continue;
        }
        int32_t cum = 0;
        int32_t rl = rel < 0 ? 0 : rel;
        while (cum + vlen[walk_vtx[e]] <= rl) { cum += vlen[walk_vtx[e]]; e++; }
        const int64_t e0 = e;
        const int32_t last = rl + k - 1;
        while (cum + vlen[walk_vtx[e]] <= last) { cum += vlen[walk_vtx[e]]; e++; }
        r_cls[i] = (int32_t)c;
        r_rel[i] = rel;
        r_e0[i] = (phi_ent_t)e0;
        r_e1[i] = (phi_ent_t)e;
    }
}
void phi_launch_class_rec(hipStream_t st, const int64_t *raw_pos, int64_t n_raw, const int64_t *cls_base, int64_t n_cls,
                          const phi_ent_t *cls_rep, const uint8_t *cls_left, const int32_t *walk_vtx, const int32_t *vlen, int32_t k,
                          uint8_t *keep, int32_t *r_cls, int32_t *r_rel, phi_ent_t *r_e0, phi_ent_t *r_e1)
{
    if (n_raw > 0)
        hipLaunchKernelGGL(phi_class_rec_kernel, dim3(grid_for(n_raw, 256)), dim3(256), 0, st, raw_pos, n_raw, cls_base, n_cls, cls_rep,
                           cls_left, walk_vtx, vlen, k, keep, r_cls, r_rel, r_e0, r_e1);
}

// out[j] = src[idx[j]] for the kept raw records
__global__ void __launch_bounds__(256) phi_class_rec_gather_kernel(const int32_t *__restrict__ idx, int64_t n, const uint64_t *__restrict__ raw_hash,
                                                                   const int32_t *__restrict__ r_cls, const int32_t *__restrict__ r_rel,
                                                                   const phi_ent_t *__restrict__ r_e0, const phi_ent_t *__restrict__ r_e1,
                                                                   uint64_t *__restrict__ o_hash, int32_t *__restrict__ o_cls,
                                                                   int32_t *__restrict__ o_rel, phi_ent_t *__restrict__ o_e0, phi_ent_t *__restrict__ o_e1)
{
    GRID_STRIDE(j, n) {
        const int32_t i = idx[j];
        o_hash[j] = raw_hash[i]; o_cls[j] = r_cls[i]; o_rel[j] = r_rel[i]; o_e0[j] = r_e0[i]; o_e1[j] = r_e1[i];
    }
}
void phi_launch_class_rec_gather(hipStream_t st, const int32_t *idx, int64_t n, const uint64_t *raw_hash, const int32_t *r_cls,
                                 const int32_t *r_rel, const phi_ent_t *r_e0, const phi_ent_t *r_e1, uint64_t *o_hash, int32_t *o_cls,
                                 int32_t *o_rel, phi_ent_t *o_e0, phi_ent_t *o_e1)
{
    if (n > 0)
        hipLaunchKernelGGL(phi_class_rec_gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, idx, n, raw_hash, r_cls, r_rel, r_e0, r_e1,
                           o_hash, o_cls, o_rel, o_e0, o_e1);
}

// off[c] = first record of class c in the class-ordered record list (off[n_cls] = n_rec)
__global__ void __launch_bounds__(256) phi_class_rec_off_kernel(const int32_t *__restrict__ rec_cls, int64_t n_rec, int64_t n_cls,
                                                                int32_t *__restrict__ off)
{
    GRID_STRIDE(c, n_cls + 1) {
        int64_t lo = 0, hi = n_rec;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rec_cls[mid] < c) lo = mid + 1; else hi = mid;
        }
        off[c] = (int32_t)lo;
    }
}
void phi_launch_class_rec_off(hipStream_t st, const int32_t *rec_cls, int64_t n_rec, int64_t n_cls, int32_t *off)
{
    hipLaunchKernelGGL(phi_class_rec_off_kernel, dim3(grid_for(n_cls + 1, 256)), dim3(256), 0, st, rec_cls, n_rec, n_cls, off);
}

// ------------------------------------------------------------------------- expansion
// Lists over (entry, record of the entry's class) pairs in entry order, for the entries [e_lo, e_hi):
//   sel == nullptr : every record         (phi_walk_minimizers)
//   sel != nullptr : records with sel[r]  (the anchors the filter keeps, phi_solve)
// Pass 1 counts per block of EXP_ITEMS * 256 entries, the caller scans the block counts, pass 2 writes.
#define EXP_ITEMS 8
template <bool WRITE, int KIND>       // KIND 0: (hash, position) of a walk's minimisers; 1: (id, e0, e1) triples + class record
__global__ void __launch_bounds__(256) phi_expand_kernel(PhiExpandArgs A)
{
    __shared__ long long s_w[4];
    const int64_t base = A.e_lo + ((int64_t)blockIdx.x * 256 + threadIdx.x) * EXP_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < EXP_ITEMS; j++) {
        const int64_t e = base + j;
        if (e >= A.e_hi) break;
        const int32_t c = A.ent_cls[e];
        const int32_t lo = A.cls_rec_off[c], hi = A.cls_rec_off[c + 1];
        if (A.sel_cnt) cnt += A.sel_cnt[c];
        else cnt += hi - lo;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long long v = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_w[wid] = v;
    __syncthreads();
    if (!WRITE) {
        if (threadIdx.x == 0) A.block_cnt[blockIdx.x] = (int32_t)(s_w[0] + s_w[1] + s_w[2] + s_w[3]);
        return;
    }
    long long woff = 0;
    for (int i = 0; i < wid; i++) woff += s_w[i];
    int64_t o = A.block_off[blockIdx.x] + woff + v - cnt;
    for (int j = 0; j < EXP_ITEMS; j++) {
        const int64_t e = base + j;
        if (e >= A.e_hi) break;
        const int32_t c = A.ent_cls[e];
        const int32_t lo = A.cls_rec_off[c], hi = A.cls_rec_off[c + 1];
        if (A.sel_cnt && A.sel_cnt[c] == 0) continue;
        const int64_t rep = A.cls_rep[c];
        for (int32_t r = lo; r < hi; r++) {
            if (A.sel && !A.sel[r]) continue;
            if (KIND == 0) {
                A.out_hash[o] = A.rec_hash[r];
                A.out_pos[o] = A.ent_base[e - A.e_lo] + A.rec_rel[r];
            } else {
                A.out_tri[3 * o + 0] = A.u_uid[A.rec_slot[r]];
                A.out_tri[3 * o + 1] = (phi_ent_t)(e + ((int64_t)A.rec_e0[r] - rep));
                A.out_tri[3 * o + 2] = (phi_ent_t)(e + ((int64_t)A.rec_e1[r] - rep));
            }
            o++;
        }
    }
}
int64_t phi_expand_num_blocks(int64_t n_entries) { return (n_entries + 256 * EXP_ITEMS - 1) / (256 * EXP_ITEMS); }
void phi_launch_expand_count(hipStream_t st, const PhiExpandArgs &A)
{
    const int64_t nb = phi_expand_num_blocks(A.e_hi - A.e_lo);
    if (nb > 0) hipLaunchKernelGGL((phi_expand_kernel<false, 0>), dim3((unsigned)nb), dim3(256), 0, st, A);
}
void phi_launch_expand_write(hipStream_t st, const PhiExpandArgs &A, int kind)
{
    const int64_t nb = phi_expand_num_blocks(A.e_hi - A.e_lo);
    if (nb <= 0) return;
    if (kind == 0) hipLaunchKernelGGL((phi_expand_kernel<true, 0>), dim3((unsigned)nb), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((phi_expand_kernel<true, 1>), dim3((unsigned)nb), dim3(256), 0, st, A);
}

// sel_cnt[c] = number of selected records of class c
__global__ void __launch_bounds__(256) phi_class_sel_count_kernel(const uint8_t *__restrict__ sel, const int32_t *__restrict__ cls_rec_off,
                                                                  int64_t n_cls, int32_t *__restrict__ sel_cnt)
{
    GRID_STRIDE(c, n_cls) {
        int32_t n = 0;
        for (int32_t r = cls_rec_off[c]; r < cls_rec_off[c + 1]; r++) n += sel[r] != 0;
        sel_cnt[c] = n;
    }
}
void phi_launch_class_sel_count(hipStream_t st, const uint8_t *sel, const int32_t *cls_rec_off, int64_t n_cls, int32_t *sel_cnt)
{
    if (n_cls > 0) hipLaunchKernelGGL(phi_class_sel_count_kernel, dim3(grid_for(n_cls, 256)), dim3(256), 0, st, sel, cls_rec_off, n_cls, sel_cnt);
}

// ---- the anchors of the model, expanded from the SELECTED records only.
// The generic expansion above walks every record of an entry's class and, per selected record, gathers its slot, the slot's
// dense id and its two entry offsets from four arrays, then stores 12 bytes at a per-lane address: 62 ms for config 5's
// 5.3 * 10^8 anchors (100 GB/s).  Here the selected records are first packed per class -- sel_off[c] .. sel_off[c + 1] into
// sel_tri: (dense id, first entry - representative, last entry - representative), 4.5 M records = 54 MB, resident in the
// Infinity Cache -- so that an entry costs two adjacent loads of sel_off and one 12-byte load per anchor, and a block's
// anchors are staged in LDS and leave as whole cache lines.
__global__ void __launch_bounds__(256) phi_class_sel_tri_kernel(const uint8_t *__restrict__ sel, const int32_t *__restrict__ cls_rec_off, int64_t n_cls,
                                                                const int32_t *__restrict__ sel_off, const phi_ent_t *__restrict__ cls_rep,
                                                                const uint32_t *__restrict__ rec_slot, const uint32_t *__restrict__ u_uid,
                                                                const phi_ent_t *__restrict__ rec_e0, const phi_ent_t *__restrict__ rec_e1,
                                                                int32_t *__restrict__ sel_tri)
{
    GRID_STRIDE(c, n_cls) {
        int64_t o = sel_off[c];
        if (sel_off[c + 1] == o) continue;
        const int64_t rep = cls_rep[c];
        for (int32_t r = cls_rec_off[c]; r < cls_rec_off[c + 1]; r++) {
            if (!sel[r]) continue;
            sel_tri[3 * o + 0] = (int32_t)u_uid[rec_slot[r]];
            sel_tri[3 * o + 1] = (int32_t)((int64_t)rec_e0[r] - rep);
            sel_tri[3 * o + 2] = (int32_t)((int64_t)rec_e1[r] - rep);
            o++;
        }
    }
}
void phi_launch_class_sel_tri(hipStream_t st, const uint8_t *sel, const int32_t *cls_rec_off, int64_t n_cls, const int32_t *sel_off, const phi_ent_t *cls_rep,
                              const uint32_t *rec_slot, const uint32_t *u_uid, const phi_ent_t *rec_e0, const phi_ent_t *rec_e1, int32_t *sel_tri)
{
    if (n_cls > 0)
        hipLaunchKernelGGL(phi_class_sel_tri_kernel, dim3(grid_for(n_cls, 256)), dim3(256), 0, st, sel, cls_rec_off, n_cls, sel_off, cls_rep, rec_slot, u_uid,
                           rec_e0, rec_e1, sel_tri);
}

#define EXP_STAGE 1536          // anchors of a block staged in LDS (18 KB: eight workgroups per CU); a block with more writes them directly
// What the DP needs of every anchor beside the triple -- its last entry and its span in edges --, the anchors per walk and the
// two things that would make the device path unusable (an anchor inside one vertex is no dp anchor; a span of PHI_RCAP edges or
// more) are made here as the anchors are written, not by a second pass over 6 GB of triples (phi_anchor_prep_kernel: 17 ms at
// config 5).  prep[0] += anchors with e1 <= e0, prep[1] |= 2 for a span >= PHI_RCAP; walk_cnt[h] += anchors whose first entry
// lies in walk h.  (That the list is sorted by last entry is checked by phi_sorted_u32_kernel on the 4-byte array.)
struct PhiExpandPrep { phi_ent_t *a_e1; uint8_t *a_span; const int64_t *walk_off; int32_t n_walks; unsigned long long *walk_cnt, *prep; };
__global__ void __launch_bounds__(256) phi_expand_tri_kernel(const int32_t *__restrict__ ent_cls, int64_t e_lo, int64_t e_hi,
                                                             const int32_t *__restrict__ sel_off, const int32_t *__restrict__ sel_tri,
                                                             const int64_t *__restrict__ block_off, uint32_t *__restrict__ out_tri, PhiExpandPrep P)
{
    __shared__ int s_w[4];
    __shared__ int s_h[2];                                        // walk of the block's first entry; anchors outside it
    __shared__ long long s_wr[2];
    __shared__ uint32_t s_out[EXP_STAGE * 3];
    const int64_t base = e_lo + ((int64_t)blockIdx.x * 256 + threadIdx.x) * EXP_ITEMS;
    int32_t lo[EXP_ITEMS], nn[EXP_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < EXP_ITEMS; j++) {
        const int64_t e = base + j;
        lo[j] = 0; nn[j] = 0;
        if (e < e_hi) {
            const int32_t c = ent_cls[e];
            lo[j] = sel_off[c];
            nn[j] = sel_off[c + 1] - lo[j];
        }
        cnt += nn[j];
    }
    if (threadIdx.x == 0 && P.a_e1) {
        int l = 0, h = P.n_walks;                                 // last walk with walk_off <= the block's first entry
        while (h - l > 1) { const int mid = (l + h) >> 1; if (P.walk_off[mid] <= base) l = mid; else h = mid; }
        s_h[0] = l; s_h[1] = 0;
        s_wr[0] = P.walk_off[l]; s_wr[1] = P.walk_off[l + 1];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int v = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_w[wid] = v;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wid; i++) woff += s_w[i];
    const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    int o = woff + v - cnt;                                       // this thread's first anchor inside the block
    const int64_t gbase = block_off[blockIdx.x];
    const bool staged = total <= EXP_STAGE;
    uint32_t *dst = staged ? s_out : out_tri + 3 * gbase;
    const int64_t w_lo = P.a_e1 ? s_wr[0] : 0, w_hi = P.a_e1 ? s_wr[1] : 0;
    int n_flat = 0, n_out = 0;                                    // anchors inside one vertex; anchors that begin in another walk
    uint32_t bad = 0;
#pragma unroll
    for (int j = 0; j < EXP_ITEMS; j++) {
        const uint32_t e = (uint32_t)(base + j);                  // (entries are below 2^32: phi_ent_t)
        for (int32_t r = lo[j], re = lo[j] + nn[j]; r < re; r++, o++) {
            const int32_t id = sel_tri[3 * (int64_t)r], d0 = sel_tri[3 * (int64_t)r + 1], d1 = sel_tri[3 * (int64_t)r + 2];
            const uint32_t e0 = e + (uint32_t)d0, e1 = e + (uint32_t)d1;
            dst[3 * (int64_t)o + 0] = (uint32_t)id;
            dst[3 * (int64_t)o + 1] = e0;
            dst[3 * (int64_t)o + 2] = e1;
            if (P.a_e1) {
                P.a_e1[gbase + o] = e1;
                P.a_span[gbase + o] = (uint8_t)(e1 > e0 ? e1 - e0 : 0);
                n_flat += e1 <= e0;
                if (e1 > e0 && e1 - e0 >= PHI_RCAP) bad |= 2u;
                if ((int64_t)e0 < w_lo || (int64_t)e0 >= w_hi) {  // (a block that straddles two walks: this anchor on its own)
                    int l = 0, h = P.n_walks;
                    while (h - l > 1) { const int mid = (l + h) >> 1; if (P.walk_off[mid] <= (int64_t)e0) l = mid; else h = mid; }
                    atomicAdd(&P.walk_cnt[l], 1ull);
                    n_out++;
                }
            }
        }
    }
    if (P.a_e1) {
        if (bad) atomicOr(&P.prep[1], (unsigned long long)bad);
        if (n_flat) atomicAdd(&P.prep[0], (unsigned long long)n_flat);
        if (n_out) atomicAdd(&s_h[1], n_out);
        __syncthreads();
        if (threadIdx.x == 0 && total - s_h[1] > 0) atomicAdd(&P.walk_cnt[s_h[0]], (unsigned long long)(total - s_h[1]));
    }
    if (!staged) return;
    __syncthreads();
    uint32_t *g = out_tri + 3 * gbase;
    for (int i = threadIdx.x; i < 3 * total; i += 256) g[i] = s_out[i];
}
// out |= 1 when a[i] < a[i - 1] somewhere
__global__ void __launch_bounds__(256) phi_sorted_u32_kernel(const uint32_t *__restrict__ a, int64_t n, unsigned long long *__restrict__ out)
{
    bool badv = false;
    GRID_STRIDE(i, n) if (i > 0 && a[i] < a[i - 1]) badv = true;
    if (__ballot(badv) && (threadIdx.x & 63) == 0) atomicOr(out, 1ull);
}
void phi_launch_expand_tri(hipStream_t st, const int32_t *ent_cls, int64_t e_lo, int64_t e_hi, const int32_t *sel_off, const int32_t *sel_tri,
                           const int64_t *block_off, uint32_t *out_tri, phi_ent_t *a_e1, uint8_t *a_span, const int64_t *walk_off, int32_t n_walks,
                           unsigned long long *walk_cnt, unsigned long long *prep, int64_t n_anchors)
{
    const int64_t nb = phi_expand_num_blocks(e_hi - e_lo);
    if (nb <= 0) return;
    PhiExpandPrep P{a_e1, a_span, walk_off, n_walks, walk_cnt, prep};
    hipLaunchKernelGGL(phi_expand_tri_kernel, dim3((unsigned)nb), dim3(256), 0, st, ent_cls, e_lo, e_hi, sel_off, sel_tri, block_off, out_tri, P);
    if (a_e1 && n_anchors > 1) hipLaunchKernelGGL(phi_sorted_u32_kernel, dim3(grid_for(n_anchors, 256)), dim3(256), 0, st, a_e1, n_anchors, prep + 1);
}

// sel[r] = 1 for the listed records (list of indices into the class records)
__global__ void __launch_bounds__(256) phi_mark_list_kernel(const int32_t *__restrict__ list, const int32_t *__restrict__ through, int64_t n,
                                                            uint8_t *__restrict__ sel)
{
    GRID_STRIDE(j, n) sel[through ? through[list[j]] : list[j]] = 1;
}
void phi_launch_mark_list(hipStream_t st, const int32_t *list, const int32_t *through, int64_t n, uint8_t *sel)
{
    if (n > 0) hipLaunchKernelGGL(phi_mark_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, list, through, n, sel);
}

// ------------------------------------------------------------------------- -d1 sharing histogram
// Number of walks every distinct walk minimiser occurs in (ILP_index.cpp:565-604): one launch per walk over
// its entries; the first (minimiser, walk) pair bumps the minimiser's walk count.
__global__ void __launch_bounds__(256) phi_share_count_cls_kernel(const int32_t *__restrict__ ent_cls, int64_t e_lo, int64_t e_hi,
                                                                  const int32_t *__restrict__ cls_rec_off, const uint32_t *__restrict__ rec_slot,
                                                                  int32_t walk, int32_t *__restrict__ last_walk, int32_t *__restrict__ n_walks_of)
{
    for (int64_t e = e_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_hi; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = ent_cls[e];
        for (int32_t r = cls_rec_off[c]; r < cls_rec_off[c + 1]; r++) {
            const uint32_t s = rec_slot[r];
            if (atomicExch(&last_walk[s], walk) != walk) atomicAdd(&n_walks_of[s], 1);
        }
    }
}
void phi_launch_share_count_cls(hipStream_t st, const int32_t *ent_cls, int64_t e_lo, int64_t e_hi, const int32_t *cls_rec_off,
                                const uint32_t *rec_slot, int32_t walk, int32_t *last_walk, int32_t *n_walks_of)
{
    if (e_hi > e_lo)
        hipLaunchKernelGGL(phi_share_count_cls_kernel, dim3(grid_for(e_hi - e_lo, 256)), dim3(256), 0, st, ent_cls, e_lo, e_hi, cls_rec_off,
                           rec_slot, walk, last_walk, n_walks_of);
}

// lens[i] = vlen[walk_vtx[e_lo + i]] : the caller scans them into the base offsets of one walk's entries
__global__ void __launch_bounds__(256) phi_entry_len_range_kernel(const int32_t *__restrict__ walk_vtx, const int32_t *__restrict__ vlen,
                                                                  int64_t e_lo, int64_t n, int32_t *__restrict__ lens)
{
    GRID_STRIDE(i, n) lens[i] = vlen[walk_vtx[e_lo + i]];
}
void phi_launch_entry_len_range(hipStream_t st, const int32_t *walk_vtx, const int32_t *vlen, int64_t e_lo, int64_t n, int32_t *lens)
{
    if (n > 0) hipLaunchKernelGGL(phi_entry_len_range_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, walk_vtx, vlen, e_lo, n, lens);
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_contexts_kernel() {}
void phi_warm_contexts(hipStream_t st) { hipLaunchKernelGGL(phi_warm_contexts_kernel, dim3(1), dim3(64), 0, st); }
