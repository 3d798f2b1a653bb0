// sketch.hip -- packing and (w,k)-minimiser sketch kernels for gfx950.
//
// Replaces the per-position std::string loops of the reference:
//   ILP_index::compute_hashes  src/ILP_index.cpp:447-493  (reads)
//   ILP_index::index_kmers     src/ILP_index.cpp:359-445  (haplotype walks, minus the vertex map)
// Semantics reproduced bit for bit: upper-casing, canonical k-mer =
// min(fwd, revcomp) in string order, window minimum with the rightmost tie, one record per
// window whose minimum hashes differently from the previous window's (prev_hash starts at
// UINT64_MAX per sequence), MurmurHash3_x64_128 seed 0 folded h1^h2.
//
// One flat base space holds all sequences of a batch back to back; a bitmap marks sequence
// starts.  A window is live iff no sequence starts inside it.  Every WAVE owns PHI_WCH consecutive
// window positions and never waits for another wave (no workgroup barrier):
//   phase 0  stage the chunk's packed words and bitmaps into LDS (the only global reads)
//   phase 1  canonical k-mers of the chunk (rolling 2-bit arithmetic)      -> LDS
//   phase 2  sliding-window minima, Q+1 consecutive windows per lane from a
//            suffix-min / core / prefix-min split (w+Q LDS reads per lane)
//   phase 3  candidate windows (minimum changed, or first window of a sequence), compacted
//            into LDS with a wave scan so that the hash runs on dense lanes only
//   phase 4  murmur3 of each candidate, hash-change test against its predecessor
//   phase 5  ballot/prefix-sum compaction of the emitted records and, by mode,
//            count | ordered write | table probe + coalesced log of the read hashes that are not walk minimisers
//
// Bases outside ACGTacgt (the reference keeps them as bytes: N sorts between G and T and is its
// own complement, ILP_index.cpp:350-353) cannot live in 2 bits.  The pack kernels set one bit per
// such base; the 2-bit path skips every window that, together with its predecessor, touches a
// marked base, and an exact byte-wise kernel (phi_sketch_bytes_kernel, launched after the 2-bit
// kernel; it leaves at once when the batch holds no such base) handles exactly those windows.
// With `allslow` the byte-wise kernel handles every window (ordered write of walks that contain
// such bases).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include "phi_dev.h"
#include "phi_kernels.h"
#ifndef PHI_ABL
#define PHI_ABL 0      // ablation / occupancy experiments: 0 = the product
#endif

#define TPB PHI_TPB
#define Q 8                     // windows per lane

int64_t phi_sketch_num_blocks(int64_t n_bases);

// ---------------------------------------------------------------------------------- packing

// 32 ASCII bases per lane -> one packed word (+ one 32-bit mask of the bases outside ACGTacgt).
static __device__ __forceinline__ void pack_ascii_word(int64_t wi, const uint8_t *__restrict__ bases, int64_t n,
                                                       uint64_t *__restrict__ words, int64_t n_words,
                                                       uint32_t *__restrict__ badbits,
                                                       unsigned long long *__restrict__ n_bad)
{
    if (wi >= n_words + 4) return;
    if (wi >= n_words) {                                    // zero padding: 2 words, 4 mask words
        if (wi < n_words + 2) words[wi] = 0;
        if (badbits) badbits[wi] = 0;
        return;
    }
    const int64_t b0 = wi * 32;
    uint64_t word = 0;
    uint32_t bad = 0;
    if (b0 + 32 <= n && ((uintptr_t)(bases + b0) & 15) == 0) {
        const uint4 *p = reinterpret_cast<const uint4 *>(bases + b0);
        uint4 v[2] = {p[0], p[1]};
        const uint32_t *u = reinterpret_cast<const uint32_t *>(v);
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t c = (u[q] >> (8 * j)) & 0xFFu;
                bad |= (uint32_t)(!phi_is_acgt(c)) << (4 * q + j);
                word = (word << 2) | phi_code(c);
            }
        }
    } else {
        for (int j = 0; j < 32; j++) {
            uint32_t c = 'A';
            if (b0 + j < n) { c = bases[b0 + j]; bad |= (uint32_t)(!phi_is_acgt(c)) << j; }
            word = (word << 2) | phi_code(c);
        }
    }
    words[wi] = word;
    if (badbits) badbits[wi] = bad;
    if (bad) atomicAdd(n_bad, (unsigned long long)__popc(bad));
}

#ifndef PHI_SKETCH_POOLED_TU
__global__ void __launch_bounds__(256) phi_pack_ascii_kernel(const uint8_t *__restrict__ bases, int64_t n,
                                                             uint64_t *__restrict__ words, int64_t n_words,
                                                             uint32_t *__restrict__ badbits,
                                                             unsigned long long *__restrict__ n_bad)
{
    pack_ascii_word((int64_t)blockIdx.x * blockDim.x + threadIdx.x, bases, n, words, n_words, badbits, n_bad);
}

// starts bitmap: bit (p & 63) of word p >> 6 set iff a sequence starts at base p.
__global__ void phi_mark_starts_kernel(const int64_t *__restrict__ seq_off, int64_t n_seq,
                                       unsigned long long *__restrict__ starts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seq) return;
    const int64_t p = seq_off[i];
    if (seq_off[i + 1] > p)                 // empty sequences own no base
        atomicOr(&starts[p >> 6], 1ull << (p & 63));
}

#endif

// Same bitmap, one whole word per lane: no memset, no atomics.  Word j covers bases [64j, 64j+64).
static __device__ __forceinline__ void start_bitmap_word(int64_t j, const int64_t *__restrict__ seq_off, int64_t n_seq,
                                                         unsigned long long *__restrict__ starts, int64_t n_sw, int64_t total = 0)
{
    if (j >= n_sw) return;
    const int64_t lo_b = j * 64, hi_b = lo_b + 64;
    // first sequence with seq_off >= lo_b (n_seq if none).  Reads are of similar lengths, so the answer
    // lies close to lo_b / mean length: gallop away from that guess, then bisect the bracket (3-4
    // dependent loads instead of log2(n_seq))
    int64_t lo = 0, hi = n_seq;
    if (total > 0 && n_seq > 64) {
        int64_t g = (int64_t)((double)lo_b / (double)total * (double)n_seq);
        g = g < 0 ? 0 : (g > n_seq - 1 ? n_seq - 1 : g);
        if (seq_off[g] >= lo_b) {                      // answer <= g
            hi = g;
            int64_t step = 1;
            while (hi - step >= 0) {
                if (seq_off[hi - step] < lo_b) { lo = hi - step + 1; break; }
                hi -= step;
                step <<= 1;
            }
        } else {                                       // answer > g
            lo = g + 1;
            int64_t step = 1;
            while (lo + step < n_seq) {
                if (seq_off[lo + step] >= lo_b) { hi = lo + step; break; }
                lo += step + 1;
                step <<= 1;
            }
        }
    }
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (seq_off[mid] < lo_b) lo = mid + 1; else hi = mid;
    }
    unsigned long long word = 0;
    for (int64_t r = lo; r < n_seq; r++) {
        const int64_t p = seq_off[r];
        if (p >= hi_b) break;
        if (seq_off[r + 1] > p) word |= 1ull << (p & 63);   // empty sequences own no base
    }
    starts[j] = word;
}

// one launch that forgets all reads: zero hit vector, zero striped counters (only when two resets follow each other with
// no read launch in between: otherwise the waves of the next read launch do it, see clean_finish)
#ifndef PHI_SKETCH_POOLED_TU
__global__ void __launch_bounds__(256) phi_reset_reads_kernel(uint64_t *__restrict__ hit_words, int64_t n_hit_words,
                                                              uint64_t *__restrict__ stripes, int64_t n_stripe_words)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t; i < n_hit_words; i += stride) hit_words[i] = 0;
    for (int64_t i = t; i < n_stripe_words; i += stride) stripes[i] = 0;
}

#endif

// ---------------------------------------------------------------------------------- helpers

// Lanes of one wave exchange data through LDS without a workgroup barrier: LDS instructions of a
// wave execute in issue order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// first set bit at local index >= from and < limit in an LDS bitmap, or INT_MAX
__device__ __forceinline__ int next_start_lds(const unsigned long long *s_bits, int from, int limit)
{
    if (from >= limit) return INT_MAX;
    int wi = from >> 6;
    const int wl = (limit - 1) >> 6;
    unsigned long long word = s_bits[wi] & (~0ull << (from & 63));
    for (;;) {
        if (word) {
            const int p = (wi << 6) + (__ffsll((long long)word) - 1);
            return p < limit ? p : INT_MAX;
        }
        if (++wi > wl) return INT_MAX;
        word = s_bits[wi];
    }
}

// 32 bases starting at local base lb of the staged words, left-aligned
__device__ __forceinline__ uint64_t lds_extract64(const uint64_t *s_words, int lb)
{
    const int wi = lb >> 5, s = (lb & 31) * 2;                 // s <= 62: no special case for s == 0
    return (s_words[wi] << s) | ((s_words[wi + 1] >> 1) >> (63 - s));
}

struct MinEnt { uint64_t v; int i; };
// b lies to the right of a: ties go right (the reference's deque pops on >=, ILP_index.cpp:397)
__device__ __forceinline__ MinEnt take_right(MinEnt a, MinEnt b) { return (b.v <= a.v) ? b : a; }

// Walk-minimiser table probe of one emitted read hash.  A hash found in the walk table is recorded by its hit flag
// alone; the others -- NOVEL hashes: sequencing errors, alleles no walk carries -- only feed |Sp_R| and the log counters
// that derive from it (ILP_index.cpp:622-641, 738-743, 883).  They are not entered into a set here (round 3: an atomicCAS
// per novel hash into a 64-MB table, a second dependent round trip behind the probe and ~100 bytes of write traffic per
// 8-byte key: 45 % of the launch on long noisy reads): the wave appends them to its LOG, coalesced, and the set is made
// from the log once, when somebody asks for |Sp_R| (phi_abi.hip sp_flush).
// returns true when h is NOT a walk minimiser
struct ProbeArgs { const uint64_t *u_kv; uint64_t u_mask; uint8_t *hit; uint32_t *err; };
__device__ __forceinline__ bool probe_table(const ProbeArgs &A, uint64_t h)
{
    if (h == PHI_EMPTY_KEY) { atomicOr(A.err, PHI_KERR_SENTINEL); return false; }
    uint64_t su = h & A.u_mask;
    const ulonglong2 *kv = reinterpret_cast<const ulonglong2 *>(A.u_kv);
    ulonglong2 e0 = kv[su];                            // key and dense id in one round trip
    // (both halves made live here: left alone the compiler loads the key, and the id in a second, dependent load
    //  inside the branch of a match -- 85 % of the probes)
    asm volatile("" : "+v"(e0.x), "+v"(e0.y));
    // walk-minimiser table: lookup, mark the minimiser as hit
    if (e0.x == h) { A.hit[(uint32_t)e0.y] = 1; return false; }
    if (e0.x != PHI_EMPTY_KEY) {
        for (int probes = 1; probes <= PHI_MAX_PROBE; probes++) {
            su = (su + 1) & A.u_mask;
            const ulonglong2 e = kv[su];
            if (e.x == h) { A.hit[(uint32_t)e.y] = 1; return false; }
            if (e.x == PHI_EMPTY_KEY) break;
        }
    }
    return true;
}

__device__ __forceinline__ bool probe_table(const PhiSketchArgs &A, uint64_t h)
{
    const ProbeArgs P{A.u_kv, A.u_mask, A.hit, A.err};
    return probe_table(P, h);
}

// Novel hashes a wave's log has no room for (a chunk that emits more than 1.5x what random sequence does; the byte-wise
// routine's windows): one atomic per round for all of them, then a coalesced store into the generation's overflow list.
// A full list raises PHI_KERR_TABLE_FULL: phi_add_reads grows it and replays the batch.  Wave-uniform call.
// returns how many the round sent there
struct OverflowArgs { uint64_t *list; unsigned long long *count; int64_t cap; uint32_t *err; };
__device__ __forceinline__ int overflow_novel(const OverflowArgs &O, bool ov, uint64_t h, int lane)
{
    const unsigned long long ob = __ballot(ov);
    if (!ob) return 0;
    const int first = __ffsll((long long)ob) - 1, n = __popcll(ob);
    unsigned long long base = 0;
    if (lane == first) base = atomicAdd(O.count, (unsigned long long)n);
    base = __shfl(base, first, 64);
    if (ov) {
        const unsigned long long idx = base + (unsigned long long)__popcll(ob & ((1ull << lane) - 1));
        if ((int64_t)idx < O.cap) O.list[idx] = h;
        else atomicOr(O.err, PHI_KERR_TABLE_FULL);
    }
    return n;
}
// The kernel's arguments where the launch put them (the kernarg segment; the one struct parameter sits at its start), through
// a pointer the compiler cannot see through: what is read this way is loaded (scalar loads) where it is used instead of
// living in scalar registers from the start of the kernel -- the read kernel's loop over a wave's chunks runs with every
// scalar register taken, and each value too many costs v_writelane / v_readlane pairs in every turn.
typedef const __attribute__((address_space(4))) PhiSketchArgs *KArgs;
__device__ __forceinline__ KArgs kargs_now()
{
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// ---------------------------------------------------------------------------------- byte-wise routine
// Exact restatement on ASCII for windows the 2-bit path cannot take (ILP_index.cpp:330-357, 388-414).

struct KRef { int64_t pos; int rc; };        // a k-mer: start base and strand

__device__ __forceinline__ uint32_t up_byte(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
__device__ __forceinline__ uint32_t comp_byte(uint32_t c)     // c is upper case; others map to themselves
{
    return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
}
__device__ __forceinline__ uint32_t kbyte(const uint8_t *__restrict__ s, KRef r, int x, int k)
{
    return r.rc ? comp_byte(up_byte(s[r.pos + k - 1 - x])) : up_byte(s[r.pos + x]);
}
__device__ __forceinline__ int kcmp(const uint8_t *__restrict__ s, KRef a, KRef b, int k)
{
    for (int x = 0; x < k; x++) {
        const uint32_t ca = kbyte(s, a, x, k), cb = kbyte(s, b, x, k);
        if (ca != cb) return ca < cb ? -1 : 1;
    }
    return 0;
}
__device__ __forceinline__ KRef canon_bytes(const uint8_t *__restrict__ s, int64_t i, int k)
{
    const KRef f{i, 0}, r{i, 1};
    return kcmp(s, r, f, k) < 0 ? r : f;                    // std::min(fwd, rev)
}
__device__ KRef window_best_bytes(const uint8_t *__restrict__ s, int64_t a, int k, int w)
{
    KRef best = canon_bytes(s, a, k);
    for (int j = 1; j < w; j++) {
        const KRef c = canon_bytes(s, a + j, k);
        if (kcmp(s, c, best, k) <= 0) best = c;             // ties go right
    }
    return best;
}
template <bool LONGK>   // LONGK: k may exceed 32 (eight lanes; kept out of the kernels that never see such k)
__device__ uint64_t khash_bytes(const uint8_t *__restrict__ s, KRef r, int k)
{
    if (LONGK && k > PHI_MAX_K_PACKED) {
        uint64_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < 8; g++) {                               // (constant indices: the lanes stay in registers)
            uint64_t v = 0;
            for (int x = 8 * g; x < 8 * g + 8 && x < k; x++) v |= (uint64_t)kbyte(s, r, x, k) << (8 * (x & 7));
            e[g] = v;
        }
        return phi_murmur_lanes8(e, k);
    }
    uint64_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int x = 0; x < k; x++) {
        const uint64_t b = (uint64_t)kbyte(s, r, x, k) << (8 * (x & 7));
        if (x < 8) e0 |= b; else if (x < 16) e1 |= b; else if (x < 24) e2 |= b; else e3 |= b;
    }
    return phi_murmur_lanes(e0, e1, e2, e3, k);
}

__device__ __forceinline__ bool range_has_bit(const unsigned long long *s_bits, int lo, int hi)   // [lo, hi]
{
    return next_start_lds(s_bits, lo, hi + 1) != INT_MAX;
}

// Windows la = 1..WCH of this wave's chunk that need the byte-wise path, in position order.
template <int MODE, bool LONGK = false>
__device__ __forceinline__ void slow_windows(const PhiSketchArgs &A, int64_t c0, int64_t chunk, int lane, int k, int w,
                                          const unsigned long long *s_bits, const unsigned long long *s_bad,
                                          bool allslow, int64_t out_base, int &n_emit, int &n_nov)
{
    const int64_t N = A.n_bases;
    const int span = w + k - 1;
    for (int r = 0; r < PHI_WCH / 64; r++) {
        const int la = 1 + r * 64 + lane;
        const int64_t a = c0 - 1 + la;
        const int lp = la + 63;                                     // local bit of base a
        bool todo = (a + span <= N) && !range_has_bit(s_bits, lp + 1, lp + span - 1);
        if (todo && !allslow) todo = range_has_bit(s_bad, lp - 1, lp + span - 1);
        bool emit = false;
        uint64_t h = 0;
        int64_t pos = 0;
        if (todo) {
            const bool first = (s_bits[lp >> 6] >> (lp & 63)) & 1ull;
            const KRef best = window_best_bytes(A.ascii, a, k, w);
            h = khash_bytes<LONGK>(A.ascii, best, k);
            pos = best.pos;
            if (first) emit = h != PHI_EMPTY_KEY;                   // prev_hash = UINT64_MAX (:383, :455)
            else {
                const KRef prev = window_best_bytes(A.ascii, a - 1, k, w);
                emit = !(prev.pos == best.pos && prev.rc == best.rc) && khash_bytes<LONGK>(A.ascii, prev, k) != h;
            }
        }
        const unsigned long long bal = __ballot(emit);
        bool novel = false;
        if (emit) {
            const int rank = n_emit + __popcll(bal & ((1ull << lane) - 1));
            if (MODE == PHI_MODE_WRITE) {
                A.out_hash[out_base + rank] = h;
                A.out_pos[out_base + rank] = pos;
            } else if (MODE == PHI_MODE_PROBE) {
                novel = probe_table(A, h);
            }
        }
        if (MODE == PHI_MODE_PROBE) {
            const OverflowArgs O{A.ov_list, A.ov_count, A.ov_cap, A.err};
            n_nov += overflow_novel(O, novel, h, lane);               // (the byte-wise routine's novel hashes: straight to the overflow list)
        }
        n_emit += __popcll(bal);
    }
    (void)chunk;
}

// ---------------------------------------------------------------------------------- sketch

#define WCH PHI_WCH
#define SWW 32          // staged packed words per wave:  (WCH + w + k + 62) / 32 + 1 <= 29
#define SBW 16          // staged bitmap words per wave:  (WCH + 64 + w + k) / 64 + 2 <= 16
#define SM(l) s_mp[(l) + ((l) >> 3)]   // one pad word per 8 entries: lane t reads entries 8t+i
                                        // = u64 index 9t+i: conflict-free for ds_read_b64

// k-mer slots of one wave: every lane rolls P = ceil((WCH + w) / 64) consecutive k-mers and stores all
// of them (the lanes past WCH + w write k-mers nobody reads), so the roll needs no per-store check
// (lanes whose first k-mer lies past WCH + w store nothing: the region ends with the last storing lane's P slots)
__host__ __device__ static inline int phi_wave_mp_u64(int w)
{
    const int M = WCH + w, P = (M + 63) / 64;
    const int slots = ((M - 1) / P + 1) * P;
    return ((slots + 8) * 9) / 8 + 8;
}
// items of one chunk: the window before its first candidate, its candidates, and one more per stretch of
// windows the 2-bit path leaves to the byte-wise one (a stretch is longer than a window's span: the first
// window after it carries its own predecessor, see phase 3); + 4 trash slots
__host__ __device__ static inline int phi_wave_items(int w, int k) { return 1 + WCH + WCH / (w + k + 1) + 2; }
// An item is 32 bits when it carries the minimiser's position (ordered write of the walks), 16 bits otherwise
// (slot 0 .. 512, "first window of its sequence", "only its hash is needed"): with the k-mer region above that is
// 6.5 KB of LDS per wave for the read kernel -- six workgroups per CU instead of five, which is worth 10 % (the
// kernel waits on its probes: DESIGN.md 4.1).
__host__ __device__ static inline int phi_wave_region_u64(int w, int k, bool pos)
{
    const int item_bytes = pos ? 4 : 2;
    return phi_wave_mp_u64(w) + SWW + 2 * SBW + ((phi_wave_items(w, k) + 4) * item_bytes + 7) / 8
#if PHI_ABL == 4
           + 256          // (occupancy experiment: fewer workgroups per CU)
#endif
        ;
}

// minimum of two k-mer values below 2^62 (k <= 31) in one instruction: bit patterns with the two top
// bits clear are non-negative finite doubles (never NaN or infinity), and IEEE order of non-negative
// doubles is the unsigned order of their bit patterns; f64 denormals are kept by the kernels' mode
// register (.amdhsa_float_denorm_mode_16_64 3), so the selected operand comes back unchanged
__device__ __forceinline__ uint64_t min_u62(uint64_t a, uint64_t b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (uint64_t)__double_as_longlong(r);
}

// the value one lane below (lane 0 gets `first`): DPP wave_shr:1, no LDS round trip
__device__ __forceinline__ uint64_t wave_prev_u64(uint64_t v, uint64_t first, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    const uint64_t p = ((uint64_t)hi << 32) | lo;
    return lane == 0 ? first : p;
}

// inclusive prefix sum over the 64 lanes on the VALU (DPP row shifts + row broadcasts; __shfl_up
// would be six ds_bpermute round trips)
__device__ __forceinline__ int wave_scan_inclusive(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2, 3
    return v;
}

// The byte-wise path of one workgroup: chunks blk, blk + n_blk, ... (see phi_sketch_bytes_kernel).
template <int MODE>
static __device__ __forceinline__ void bytes_role(const PhiSketchArgs &A, const unsigned long long *batch_bad,
                                                  int64_t blk, int64_t n_blk, unsigned long long *s_all)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (!A.allslow && batch_bad && *batch_bad == 0) return;
    const int64_t N = A.n_bases;
    const int64_t n_chunks = (N + WCH - 1) / WCH;
    const int64_t n_sw = N / 64 + 2;
    unsigned long long *s_bits = s_all + (size_t)wid * 2 * SBW, *s_bad = s_bits + SBW;
    for (int64_t chunk = blk * (TPB / 64) + wid; chunk < n_chunks; chunk += n_blk * (TPB / 64)) {
        const int64_t c0 = chunk * WCH;
        unsigned long long my_bad = 0;
        wave_sync();
        if (lane < SBW) {
            const int64_t wi = (c0 >> 6) - 1 + lane;
            s_bits[lane] = (wi >= 0 && wi < n_sw) ? A.starts[wi] : 0;
        } else if (lane < 2 * SBW) {
            const int64_t wi = (c0 >> 6) - 1 + (lane - SBW);
            if (A.badbits) my_bad = (wi >= 0 && wi < n_sw) ? A.badbits[wi] : 0;
            s_bad[lane - SBW] = my_bad;
        }
        const bool chunk_bad = __ballot(my_bad != 0) != 0ull;
        wave_sync();
        int n_emit = 0, n_nov = 0;
        if (A.allslow || chunk_bad) {
            const int64_t out_base = (MODE == PHI_MODE_WRITE) ? A.block_off[chunk] : 0;
            if (A.k > PHI_MAX_K_PACKED) slow_windows<MODE, true>(A, c0, chunk, lane, A.k, A.w, s_bits, s_bad, A.allslow != 0, out_base, n_emit, n_nov);
            else slow_windows<MODE>(A, c0, chunk, lane, A.k, A.w, s_bits, s_bad, A.allslow != 0, out_base, n_emit, n_nov);
        }
        if (MODE == PHI_MODE_COUNT) {
            if (lane == 0) A.block_cnt[chunk] = n_emit;
        }
    }
}


// ---- read batches: what used to be a preparation launch, done by the waves of the sketch launch

// This wave's share of emptying the buffers of the previous generation of reads (phi_reset_reads swaps the
// context's double buffers; the generation after this one will fill them again): its hit vector and its striped
// counters -- stores into buffers nothing in this launch reads.  (Until round 3 also the slots that generation had
// filled in a spectrum set, from a log of slots: the set is now made from the log of novel hashes when it is asked
// for, and nothing of it lives across a reset.)
template <class AT>
__device__ __forceinline__ void clean_finish(const AT &A, int64_t gw, int64_t n_waves, int lane)
{
    if (gw == 0 && lane == 0 && A.ov_zero) *A.ov_zero = 0;              // the overflow counter of the generation after this one
    if (A.ipc_mb && A.ipc_need) {
        // a group of processes: the hit vector about to be zeroed may still be read by a peer until this rank's gather
        // ipc_need has ended (phi_ipc.hip) -- it has, long ago, unless a peer lags: then wait (bounded: the gather itself gives up)
        // (relaxed polls: an acquire load would invalidate the XCD's L2 with every poll; the stores below depend on the loop's exit)
        while (__hip_atomic_load(A.ipc_mb + PHI_MB_GATHERED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < A.ipc_need) __builtin_amdgcn_s_sleep(32);
    }
    // (wave-uniform guards: most waves of a large launch have nothing to empty and skip on scalar compares)
    if (gw * 64 < A.q_n_hit_words)
        for (int64_t i = gw * 64 + lane; i < A.q_n_hit_words; i += n_waves * 64) A.q_hit_words[i] = 0;
    if (gw * 64 < A.q_n_stripe_words)
        for (int64_t i = gw * 64 + lane; i < A.q_n_stripe_words; i += n_waves * 64) A.q_stripes[i] = 0;
}

// Read starts of a chunk from the read offsets.  The chunk's bitmap covers bases [c0-64, c0-64+64*SBW); the first
// read starting at or after its first base is found by probing 64 consecutive offsets around position / mean
// read length (one load when reads are of similar lengths), then 64-ary narrowing.
struct StartProbe { int64_t base; int64_t v; int64_t nx; };
__device__ __forceinline__ StartProbe start_probe_issue(const PhiSketchArgs &A, int64_t c0, int lane)
{
    const int64_t lo_b = c0 - 64 > 0 ? c0 - 64 : 0;
    // (a guess: any value is safe, the probe is checked for bracketing the chunk.  Position x reads per base in 0.32
    //  fixed point: wave-uniform integer arithmetic instead of a double division per lane)
    int64_t g = (int64_t)(((unsigned long long)lo_b * A.reads_per_base_q32) >> 32) - 24;
    g = g < 0 ? 0 : (g > A.n_reads - 63 ? (A.n_reads - 63 > 0 ? A.n_reads - 63 : 0) : g);
    StartProbe p;
    p.base = g;
    const int64_t idx = g + lane;
    p.v = idx <= A.n_reads ? A.read_off[idx] : INT64_MAX;              // read_off[n_reads] = n_bases closes the array
    p.nx = idx + 1 <= A.n_reads ? A.read_off[idx + 1] : INT64_MAX;
    return p;
}
__device__ __forceinline__ void set_start_bit(unsigned long long *s_bits, int64_t p, int64_t lo_b)
{
    const int bit = (int)(p - lo_b);
    atomicOr(reinterpret_cast<unsigned int *>(s_bits) + (bit >> 5), 1u << (bit & 31));
}
__device__ __forceinline__ void start_bits_from_offsets(const PhiSketchArgs &A, int64_t c0, int lane, StartProbe pr,
                                                        unsigned long long *s_bits)
{
    const int64_t lo_b = c0 - 64, hi_b = lo_b + 64 * SBW;
    const unsigned long long ge = __ballot(pr.v >= lo_b), lt = __ballot(pr.v < hi_b);
    if ((pr.base == 0 || !(ge & 1ull)) && !(lt >> 63)) {
        // the 64 probed reads bracket the chunk (reads of similar lengths: nearly always): their offsets are all it takes
        if (pr.base + lane < A.n_reads && pr.v >= lo_b && pr.v < hi_b && pr.nx > pr.v) set_start_bit(s_bits, pr.v, lo_b);
        return;
    }
    {
        // reads of uneven lengths (long reads): the guess by mean length was off; correct it by what the probe saw --
        // the distance in bases from its middle offset, over the mean length -- and probe 64 reads there
        int64_t vm = __shfl(pr.v, 32, 64);
        if (vm == INT64_MAX) vm = A.n_bases;
        int64_t nb_all = A.n_bases, nr_all = A.n_reads;
        asm volatile("" : "+s"(nb_all), "+s"(nr_all));    // (not hoisted out of the loop over a wave's chunks: a division kept in registers through every turn)
        const double mean = (double)nb_all / (double)nr_all;
        int64_t g = pr.base + 32 + (int64_t)((double)(lo_b - vm) / mean) - 24;
        g = g < 0 ? 0 : (g > A.n_reads - 63 ? (A.n_reads - 63 > 0 ? A.n_reads - 63 : 0) : g);
        const int64_t idx = g + lane;
        const int64_t v2 = idx <= A.n_reads ? A.read_off[idx] : INT64_MAX;
        const int64_t nx2 = idx + 1 <= A.n_reads ? A.read_off[idx + 1] : INT64_MAX;
        const unsigned long long ge2 = __ballot(v2 >= lo_b), lt2 = __ballot(v2 < hi_b);
        if ((g == 0 || !(ge2 & 1ull)) && !(lt2 >> 63)) {
            if (idx < A.n_reads && v2 >= lo_b && v2 < hi_b && nx2 > v2) set_start_bit(s_bits, v2, lo_b);
            return;
        }
    }
    // first index r0 in [0, n_reads] with read_off[r0] >= lo_b  (read_off[n_reads] = n_bases > lo_b): 64-ary narrowing
    int64_t lo = 0, hi = A.n_reads;
    int64_t base = pr.base, stride = 1, v = pr.v;
    for (int round = 0; round < 64 && lo < hi; round++) {
        if (round) {
            stride = (hi - lo + 63) / 64;
            base = lo;
            const int64_t idx = base + lane * stride;
            v = idx <= A.n_reads ? A.read_off[idx] : INT64_MAX;
        }
        const unsigned long long g2 = __ballot(v >= lo_b);
        const int f = g2 ? __ffsll((long long)g2) - 1 : 64;              // the true lanes are a suffix
        if (f == 0) { hi = base < hi ? base : hi; }
        else {
            const int64_t below = base + (int64_t)(f - 1) * stride;      // read_off[below] < lo_b
            lo = below + 1 > lo ? below + 1 : lo;
            if (f < 64) { const int64_t at = base + (int64_t)f * stride; hi = at < hi ? at : hi; }
        }
    }
    // every read from r0 = lo on that starts below hi_b and owns a base sets its bit
    for (int64_t r = lo + lane;; r += 64) {
        int64_t p = INT64_MAX, nx = INT64_MAX;
        if (r < A.n_reads) { p = A.read_off[r]; nx = A.read_off[r + 1]; }
        if (p >= lo_b && p < hi_b && nx > p) set_start_bit(s_bits, p, lo_b);
        if (__shfl(p, 63, 64) >= hi_b) break;                             // offsets are monotone: nothing further starts here
    }
}

// Reads of ONE length (short-read sets as sequencers write them): no offsets array at all -- read r starts at r * len.  The
// starts inside a chunk's bitmap are the multiples of len in its base range: one division per lane (in double precision,
// corrected), lane j sets the j-th of them.  ~25 instructions and no memory access instead of a probe of 64 offsets.
__device__ __forceinline__ void start_bits_uniform(const PhiSketchArgs &A, int64_t c0, int lane, unsigned long long *s_bits)
{
    const int64_t org = c0 - 64, hi_b = org + 64 * SBW;
    const int64_t from = org > 0 ? org : 0;
    const uint32_t L = (uint32_t)A.uniform_len;
    if (A.n_bases <= 0xFFFFFFFFll) {
        // batches below 4 Gbases (all but whole-genome sets in one batch): the chunk's first base is the same for the whole
        // wave, so the division runs on the SCALAR unit -- a multiply by floor(2^32 / L), one correction -- and a lane adds its
        // multiple of L: no vector instruction but the last few
        const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)from);
        uint32_t q = __umulhi(x, A.inv_len_q32);               // floor(x / L) or one less
        uint32_t r = x - q * L;
        if (r >= L) { q++; r -= L; }
        const uint32_t first = x + (r ? L - r : 0u);             // the first multiple of L at or after x (may wrap past 2^32: then it is past the batch)
        const uint64_t p = (uint64_t)first + (uint64_t)(uint32_t)lane * (uint64_t)L;
        if (first >= x && (int64_t)p < hi_b && (int64_t)p < A.n_bases) set_start_bit(s_bits, (int64_t)p, org);
        return;
    }
    int64_t q = (int64_t)((double)from * A.inv_len);
    int64_t r = from - q * (int64_t)L;
    if (r < 0) { q--; r += L; }
    if (r >= (int64_t)L) { q++; r -= L; }
    const int64_t p = (q + (r != 0) + lane) * (int64_t)L;       // the (lane)-th multiple of L at or after `from`
    if (p < hi_b && p < A.n_bases) set_start_bit(s_bits, p, org);  // (64 lanes cover the range: L >= 32 > 1024 / 63)
}

// phase 0 of a read chunk: lane -> the 16 bases c0 - 32 + 16 lane .. + 15 as four words ('A' beyond the batch)
__device__ __forceinline__ uint4 load_bases16(const uint8_t *__restrict__ ascii, int64_t N, int64_t c0, int lane)
{
    const int64_t b = c0 - 32 + 16 * (int64_t)lane;
    if (b >= 0 && b + 16 <= N && (((uintptr_t)ascii + (uintptr_t)b) & 15) == 0) return *reinterpret_cast<const uint4 *>(ascii + b);
    uint32_t x[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
    if (b + 16 > 0 && b < N) {
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (b + j >= 0 && b + j < N) x[j >> 2] = (x[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | ((uint32_t)ascii[b + j] << (8 * (j & 3)));
    }
    return make_uint4(x[0], x[1], x[2], x[3]);
}

// WIDE: w > Q (windows of one lane overlap in a common core); otherwise brute force per window.
// KT/WT: compile-time k and w of the specialised instance (0 = take them from the arguments).
#ifndef PHI_SKETCH_POOLED_TU
template <int MODE, bool WIDE, int KT, int WT>
__global__ void __launch_bounds__(TPB, MODE == PHI_MODE_PROBE ? 6 : 1) phi_sketch_kernel(PhiSketchArgs A)   // (reads: six waves per SIMD, at most 80 VGPRs)
{
    constexpr bool FUSED = MODE == PHI_MODE_PROBE;      // read batches come as ASCII + read offsets (see phase 0)
    constexpr bool NEED_POS = MODE == PHI_MODE_WRITE;   // only the ordered write stores positions (ILP_index.cpp:423)
    constexpr bool FMIN = !NEED_POS && KT > 0 && KT <= 31;   // values < 2^62: minima by v_min_f64
    extern __shared__ uint64_t s_dyn[];

    // (wid stays a vector register: as a scalar -- readfirstlane -- the values derived from it overflow the SGPR file,
    //  +9 % VALU instructions of v_writelane / v_readlane traffic; reading the output phase's arguments late, through
    //  a laundered kernarg pointer, frees the SGPRs but pushes four VGPRs into scratch at the 80-register bound: -35 %)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k = KT > 0 ? KT : A.k, w = WT ? WT : A.w;       // (KT = -1: the instantiation for k > 32, see below)
    const int64_t N = A.n_bases;
    const int64_t chunk = (int64_t)blockIdx.x * (TPB / 64) + wid;
    const int64_t c0 = chunk * WCH;                       // first window start of this chunk
    if (FUSED && A.ipc_mb && chunk == 0 && lane == 0)      // (a group of processes: see PhiSketchArgs)
        __hip_atomic_store(A.ipc_mb + PHI_MB_SCORED, A.ipc_scored, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (c0 >= N) {                                        // wave-uniform
        // read batches, first launch after a reset: this wave's share of the buffers the previous generation of
        // reads filled (stores into buffers nothing in this launch reads: no ordering needed)
        if (FUSED && A.q_clean) clean_finish(A, chunk, (int64_t)gridDim.x * (TPB / 64), lane);
        return;
    }
    const uint64_t kmask = phi_kmask(k);
    const int M = WCH + w;                                // canonical values m[l], l -> k-mer c0-1+l
    const int span = w + k - 1;                           // bases under one window
    const bool have_bad = FUSED || A.badbits != nullptr;

    using MetaT = typename std::conditional<NEED_POS, uint32_t, uint16_t>::type;
    constexpr uint32_t ITEM_FIRST = NEED_POS ? 1u << 31 : 1u << 15;        // the first window of its sequence
    constexpr uint32_t ITEM_NOEMIT = NEED_POS ? 1u << 30 : 1u << 14;       // only its hash is needed (the window before a candidate)
    uint64_t *s_mp = s_dyn + (size_t)wid * phi_wave_region_u64(w, k, NEED_POS);   // k-mers; later the window minima
    uint64_t *s_words = s_mp + phi_wave_mp_u64(w);
    unsigned long long *s_bits = (unsigned long long *)(s_words + SWW);
    unsigned long long *s_bad = s_bits + SBW;
    MetaT *s_meta = (MetaT *)(s_bad + SBW);               // WCH + 1 items + 8 trash slots
    uint64_t *s_q = s_mp + lane * (Q + 1);                // SM(lane * Q + x) == s_q[x + (x >> 3)]

    // (names the shared phases use; the last four only matter in the pooled kernel)
    constexpr bool POOL = false;
    bool chunk_bad = false;
    int ncand = 0, n_emit = 0, n_log = 0, n_nov_slow = 0;
    int64_t out_base = 0;
    uint4 xn = make_uint4(0, 0, 0, 0);
    const int ci = 0, mine = 1;
    const int64_t stride = 0;
#include "sketch_phases.inc"

#if PHI_ABL == 15
    { asm volatile("" :: "v"(ncand)); return; }
#endif
    // ---- phases 4 + 5: items on dense lanes, 64 per round: murmur3 of the item's minimum, the
    //      hash-change test against the item before it (the lane below; lane 0 takes the last lane of
    //      the round before), ordered compaction, output
#if PHI_ABL == 2
    ncand = 0;
#endif
    if (ncand > 0) {
        uint64_t carry = PHI_EMPTY_KEY;
        for (int r0 = 0; r0 <= ncand; r0 += 64) {
            const int t = r0 + lane;
            const bool valid = t <= ncand;
            uint32_t meta = 0;
            uint64_t h = 0;
            if (valid) {
                meta = s_meta[t];
                h = phi_kmer_hash(SM((int)(meta & 0x3FFu)), k);
            }
            const uint64_t hp = wave_prev_u64(h, carry, lane);
            carry = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), 63) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, 63);
            const bool emit = valid && !(meta & ITEM_NOEMIT) && ((meta & ITEM_FIRST) || h != hp);
            const unsigned long long bal = __ballot(emit);
            bool novel = false;
            if (emit) {
                const int rank = n_emit + __popcll(bal & ((1ull << lane) - 1));
                if (MODE == PHI_MODE_WRITE) {
                    A.out_hash[out_base + rank] = h;
                    A.out_pos[out_base + rank] = c0 - 1 + (int64_t)((meta >> 10) & 0x3FFu);
                } else if (MODE == PHI_MODE_PROBE) {
                    novel = probe_table(A, h);
                }
            }
            if (MODE == PHI_MODE_PROBE) {
                // the round's novel hashes, appended to the chunk's log in lane order: one coalesced store
                const unsigned long long ib = __ballot(novel);
                const int pos = n_log + __popcll(ib & ((1ull << lane) - 1));
                const int cap = 1 << A.nov_shift;
                if (novel && pos < cap) A.nov_log[((A.log_base + chunk) << A.nov_shift) + pos] = h;
                n_log += __popcll(ib);
                if (n_log > cap) {                            // (wave-uniform, rare) past the chunk's log: the overflow list
                    const OverflowArgs O{A.ov_list, A.ov_count, A.ov_cap, A.err};
                    overflow_novel(O, novel && pos >= cap, h, lane);
                }
            }
            n_emit += __popcll(bal);
        }
    }

    if (FUSED && chunk_bad) {
        // (rare) windows over a base outside ACGTacgt, or right after one: the exact byte-wise routine, by the wave
        // that owns the chunk (its novel hashes go to the overflow list)
        int n_emit_slow = 0;
        slow_windows<MODE>(A, c0, chunk, lane, k, w, s_bits, s_bad, false, 0, n_emit_slow, n_nov_slow);
        n_emit += n_emit_slow;
    }
    if (FUSED && A.q_clean) clean_finish(A, chunk, (int64_t)gridDim.x * (TPB / 64), lane);
    if (MODE == PHI_MODE_COUNT) {
        if (lane == 0) A.block_cnt[chunk] = n_emit;
    } else if (MODE == PHI_MODE_PROBE) {
        // one atomic per wave for the number of novel hashes logged (n_log counts them: wave-uniform; an upper bound of what
        // the set will grow by) and of emitted records
        if (lane == 0) {
            const int cap = 1 << A.nov_shift;
            A.nov_cnt[A.log_base + chunk] = (uint16_t)(n_log < cap ? n_log : cap);
            const int n_nov_wave = n_log + n_nov_slow;
            const int stripe = (int)(chunk & (PHI_STRIPES - 1)) * 8;
            if (n_nov_wave && A.n_logged) atomicAdd(A.n_logged + stripe, (unsigned long long)n_nov_wave);
            if (n_emit && A.n_emitted) atomicAdd(A.n_emitted + stripe, (unsigned long long)n_emit);
        }
    }
}

#else
// The read kernel (PHI_MODE_PROBE) for batches of 12 Mbases and more, k <= 32 (phi_launch_sketch); compiled in
// sketch_pooled.hip only.  Wave g of A.wave_stride takes the chunks g, g + stride, g + 2 stride, ... and hashes their items
// in rounds of 64 FULL lanes -- the items a round leaves over (fewer than 64) wait in one register pair per lane for the
// next chunk's.  A chunk of short reads yields ~40 items: hashed chunk by chunk (phi_sketch_kernel) the rounds run at 61 %
// of their lanes, and hash + probe are 38 % of that kernel's instructions.  Phases 0 - 3b of a chunk are
// those of phi_sketch_kernel (sketch_phases.inc).
template <bool WIDE, int KT, int WT>
__global__ void __launch_bounds__(TPB, 6) phi_sketch_pool_kernel(PhiSketchArgs A)   // (six waves per SIMD, at most 80 VGPRs)
{
    constexpr int MODE = PHI_MODE_PROBE;
    constexpr bool FUSED = true, NEED_POS = false, POOL = true;      // (names of the shared phases)
    constexpr bool FMIN = KT > 0 && KT <= 31;            // values < 2^62: minima by v_min_f64
    static_assert(KT >= 0, "k <= 32");
    extern __shared__ uint64_t s_dyn[];

    const int k = KT > 0 ? KT : A.k, w = WT ? WT : A.w;
    const int64_t N = A.n_bases;
    const int64_t n_chunks_all = (N + WCH - 1) / WCH;
    const int64_t stride = (int64_t)A.wave_stride;
    int mine;                                             // chunks of this wave (wave-uniform)
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int64_t gw = (int64_t)blockIdx.x * (TPB / 64) + wid;      // this wave's job
        if (A.ipc_mb && gw == 0 && lane == 0)             // (a group of processes: see PhiSketchArgs)
            __hip_atomic_store(A.ipc_mb + PHI_MB_SCORED, A.ipc_scored, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (gw >= n_chunks_all || gw >= stride) {         // wave-uniform
            // read batches, first launch after a reset: this wave's share of the buffers the previous generation of
            // reads filled (stores into buffers nothing in this launch reads: no ordering needed)
            if (A.q_clean) clean_finish(A, gw, (int64_t)gridDim.x * (TPB / 64), lane);
            return;
        }
        mine = __builtin_amdgcn_readfirstlane((int)((n_chunks_all - gw + stride - 1) / stride));
    }
    const uint64_t kmask = phi_kmask(k);
    const int M = WCH + w;                                // canonical values m[l], l -> k-mer c0-1+l
    const int span = w + k - 1;                           // bases under one window
    const bool have_bad = true;

    using MetaT = uint16_t;
    constexpr uint32_t ITEM_FIRST = 1u << 15;            // the first window of its sequence
    constexpr uint32_t ITEM_NOEMIT = 1u << 14;           // only its hash is needed (the window before a candidate)
    int n_emit = 0, n_log = 0, n_nov_slow = 0;
    // the items waiting for a full round -- lane j < pend holds item j: its minimum and its flags (for k <= 31 in
    // the two bits a value leaves free) -- and the hash of the last item hashed
    uint64_t pv = 0;
    uint32_t pf = 0;
    int pend = 0;
    uint64_t carry = PHI_EMPTY_KEY;

    // the 16 bases of this lane in the NEXT chunk, loaded a turn ahead
    uint4 xn = load_bases16(A.ascii, N, ((int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)) * WCH, (int)(threadIdx.x & 63));
    // (one more turn after the wave's last chunk, for the round of the items still waiting -- so that the code of a round
    //  exists once)
    for (int ci = 0; ci <= mine; ci++) {
    // (what depends on the lane is worked out again in every turn -- the compiler would otherwise keep all of it, addresses,
    //  masks, the chunk's 64-bit position, in registers across the loop: 106 VGPRs in scratch at the 80-register bound;
    //  for the same reason this file is compiled with machine LICM off, see phi_launch_sketch_pooled)
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    // (the wave's index as a scalar: the chunk's position and the wave's LDS region are scalar arithmetic)
    const int lane = (int)(tid & 63u), wid = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int64_t gw = (int64_t)blockIdx.x * (TPB / 64) + wid;
    const int64_t chunk = gw + ci * stride;
    const int64_t c0 = chunk * WCH;                       // first window start of this chunk
    int ncand = 0;
    bool chunk_bad = false;
    int64_t out_base = 0;                                 // (a name of the shared phases: the ordered write of the walks)
    (void)out_base;
    uint64_t *s_mp = s_dyn + (size_t)wid * phi_wave_region_u64(w, k, NEED_POS);   // k-mers; later the window minima
    uint64_t *s_words = s_mp + phi_wave_mp_u64(w);
    unsigned long long *s_bits = (unsigned long long *)(s_words + SWW);
    unsigned long long *s_bad = s_bits + SBW;
    MetaT *s_meta = (MetaT *)(s_bad + SBW);               // WCH + 1 items + 8 trash slots
    uint64_t *s_q = s_mp + lane * (Q + 1);                // SM(lane * Q + x) == s_q[x + (x >> 3)]
    if (ci) wave_sync();                                  // the rounds of the chunk before have read its items
    if (ci < mine) {
#include "sketch_phases.inc"

#if PHI_ABL == 15
    { asm volatile("" :: "v"(ncand)); return; }
#endif
#if PHI_ABL == 2
    ncand = 0;
#endif
    }   // phases 0 - 3b of a chunk
    // ---- phases 4 + 5: items on dense lanes, 64 per round: murmur3 of the item's minimum, the hash-change test against
    //      the item before it (the lane below; lane 0 takes the last lane of the round before), table probes
    {
        // items of this chunk: 0 .. ncand (none when it has no candidate, none in the last turn).  Full rounds while there
        // are 64 items between the waiting ones and these; what is left joins (or becomes) the waiting ones; the last
        // turn's round takes the waiting ones as they are
        const int n_items = ncand > 0 ? ncand + 1 : 0;
        const bool flush = ci == mine;
        int taken = 0;
        while (pend + (n_items - taken) >= 64 || (flush && pend > 0)) {
            uint64_t v = pv;
            uint32_t f = pf;
            if (lane >= pend && !flush) {
                const uint32_t meta = s_meta[taken + lane - pend];
                v = SM((int)(meta & 0x3FFu));
                f = meta >> 14;                           // ITEM_FIRST, ITEM_NOEMIT: bits 15, 14 of a 16-bit item
                if (FMIN) v |= (uint64_t)f << 62;
            }
            const bool valid = !flush || lane < pend;
            taken += 64 - pend;
            pend = 0;
            // one round: hash, hash-change test against the item before, probes, slot log
            const uint64_t h = phi_kmer_hash(FMIN ? (v & 0x3FFFFFFFFFFFFFFFull) : v, k);
            const uint32_t flg = FMIN ? (uint32_t)(v >> 62) : f;              // bit 1: first window of its sequence, bit 0: only its hash is needed
            const uint64_t hp = wave_prev_u64(h, carry, lane);
            carry = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), 63) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, 63);
            const bool emit = valid && !(flg & 1u) && ((flg & 2u) || h != hp);
            const unsigned long long bal = __ballot(emit);
            bool novel = false;
            const KArgs R = kargs_now();
#if PHI_ABL == 5
            asm volatile("" :: "v"(h));                   // (experiment: hash, no probe)
#else
            if (emit) {
                const ProbeArgs P{R->u_kv, R->u_mask, R->hit, R->err};
                novel = probe_table(P, h);
            }
#endif
            // the round's novel hashes, appended to the wave's log in lane order (coalesced): the log entries of the wave's
            // chunks, one after the other -- entry pos of the wave is entry pos % cap of its chunk number pos / cap
            const unsigned long long ib = __ballot(novel);
            const int pos = n_log + __popcll(ib & ((1ull << lane) - 1));
            const int sh = R->nov_shift, room = mine << sh;
            if (novel && pos < room) R->nov_log[((R->log_base + gw + (int64_t)(pos >> sh) * stride) << sh) + (pos & ((1 << sh) - 1))] = h;
            n_log += __popcll(ib);
            if (n_log > room) {                           // (wave-uniform, rare) past the wave's log: the overflow list
                const OverflowArgs O{R->ov_list, R->ov_count, R->ov_cap, R->err};
                overflow_novel(O, novel && pos >= room, h, lane);
            }
            n_emit += __popcll(bal);
        }
        const int rem = flush ? 0 : n_items - taken;      // < 64 - pend
        if (lane >= pend && lane < pend + rem) {
            const uint32_t meta = s_meta[taken + lane - pend];
            pv = SM((int)(meta & 0x3FFu));
            pf = meta >> 14;
            if (FMIN) pv |= (uint64_t)pf << 62;
        }
        pend += rem;
    }

    if (chunk_bad) {
        // (rare) windows over a base outside ACGTacgt, or right after one: the exact byte-wise routine, by the wave
        // that owns the chunk (its novel hashes go to the overflow list)
        int n_emit_slow = 0;
        slow_windows<MODE>(A, c0, chunk, lane, k, w, s_bits, s_bad, false, 0, n_emit_slow, n_nov_slow);
        n_emit += n_emit_slow;
    }
    }   // chunks of this wave
    uint32_t tid_end = threadIdx.x;
    asm volatile("" : "+v"(tid_end));                     // (worked out again: not kept from the wave's start in two registers)
    const int lane = (int)(tid_end & 63u);
    const int64_t gw = (int64_t)blockIdx.x * (TPB / 64) + __builtin_amdgcn_readfirstlane((int)(tid_end >> 6));
    {
        // (the arguments of this part are loaded now, not kept in scalar registers through the loop)
        const KArgs R = kargs_now();
        if (R->q_clean) clean_finish(*R, gw, (int64_t)gridDim.x * (TPB / 64), lane);
        {
            // entries of the log: 1 << nov_shift per chunk of this wave, filled in order
            const int sh = R->nov_shift, cap = 1 << sh;
            for (int j = lane; j < mine; j += 64) {
                const int left = n_log - (j << sh);
                R->nov_cnt[R->log_base + gw + j * stride] = (uint16_t)(left < 0 ? 0 : left < cap ? left : cap);
            }
        }
        // one atomic per wave for the number of novel hashes logged (n_log counts them: wave-uniform) and emitted records
        const int n_nov_wave = n_log + n_nov_slow;
        if (lane == 0) {
            const int stripe = (int)(gw & (PHI_STRIPES - 1)) * 8;
            if (n_nov_wave && R->n_logged) atomicAdd(R->n_logged + stripe, (unsigned long long)n_nov_wave);
            if (n_emit && R->n_emitted) atomicAdd(R->n_emitted + stripe, (unsigned long long)n_emit);
        }
    }
}

#endif

#ifdef PHI_SKETCH_POOLED_TU
// sketch_pooled.hip: this file again, compiled for the POOLED instances of the read kernel alone, with machine LICM off --
// their loop over a wave's chunks runs at the 80-register bound of six waves per SIMD, and constants and addresses hoisted
// out of it stay in registers through every turn (106 VGPRs in scratch).  The one-chunk instances keep the default.
void phi_launch_sketch_pooled(hipStream_t st, unsigned nb, size_t lds, const PhiSketchArgs &A, hipEvent_t t0, hipEvent_t t1)
{
    if (A.k == 31 && A.w == 25) hipExtLaunchKernelGGL((phi_sketch_pool_kernel<true, 31, 25>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
    else if (A.w > Q) hipExtLaunchKernelGGL((phi_sketch_pool_kernel<true, 0, 0>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
    else hipExtLaunchKernelGGL((phi_sketch_pool_kernel<false, 0, 0>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
}
#else
void phi_launch_sketch_pooled(hipStream_t st, unsigned nb, size_t lds, const PhiSketchArgs &A, hipEvent_t t0, hipEvent_t t1);   // sketch_pooled.hip

// Exact byte-wise kernel: every wave walks chunks in a grid-stride loop.
//   allslow = 0 (reads): launched after the 2-bit kernel with a small grid; leaves at once when the
//                        batch holds no base outside ACGT (*batch_bad == 0), else handles the windows
//                        the 2-bit kernel skipped;
//   allslow = 1        : every window (count / ordered write of sequences with such bases).
template <int MODE>
__global__ void __launch_bounds__(TPB) phi_sketch_bytes_kernel(PhiSketchArgs A, const unsigned long long *batch_bad)
{
    __shared__ unsigned long long s_all[(TPB / 64) * 2 * SBW];
    bytes_role<MODE>(A, batch_bad, blockIdx.x, gridDim.x, s_all);
}

// single-workgroup exclusive scan of per-block counts (off[n] = total).  Tiles of 4096 counts, four consecutive per thread:
// the loads are coalesced whatever n is -- a chromosome-scale graph scans 1.3 M block counts here, six times per solve
// (a thread summing its own contiguous share of the array, as before, read with a stride of 5 KB between lanes: 4.4 ms a call).
__global__ void __launch_bounds__(1024) phi_scan_counts_kernel(const int32_t *__restrict__ cnt, int64_t n,
                                                               int64_t *__restrict__ off)
{
    __shared__ int64_t s_w[16];
    __shared__ int64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 4096) {
        const int64_t i0 = base + 4 * (int64_t)tid;
        int32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = i0 + j < n ? cnt[i0 + j] : 0;
        const int64_t c = (int64_t)v[0] + v[1] + v[2] + v[3];
        int64_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_w[wid] = inc;
        const int64_t carry = s_carry;
        __syncthreads();
        int64_t run = carry + inc - c;
        for (int x = 0; x < wid; x++) run += s_w[x];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + j < n) off[i0 + j] = run;
            run += v[j];
        }
        if (tid == 1023) s_carry = run;                   // (the last thread's running sum = everything up to the tile's end)
        __syncthreads();
    }
    if (tid == 0) off[n] = s_carry;
}

// ---------------------------------------------------------------------------------- launchers

void phi_launch_pack_ascii(hipStream_t st, const uint8_t *bases, int64_t n, uint64_t *words, int64_t n_words,
                           uint32_t *badbits, unsigned long long *n_bad)
{
    if (n_words <= 0) return;
    const int64_t nb = (n_words + 4 + 255) / 256;
    hipLaunchKernelGGL(phi_pack_ascii_kernel, dim3((unsigned)nb), dim3(256), 0, st, bases, n, words, n_words, badbits,
                       n_bad);
}

void phi_launch_mark_starts(hipStream_t st, const int64_t *seq_off, int64_t n_seq, unsigned long long *starts)
{
    if (n_seq <= 0) return;
    const int64_t nb = (n_seq + 255) / 256;
    hipLaunchKernelGGL(phi_mark_starts_kernel, dim3((unsigned)nb), dim3(256), 0, st, seq_off, n_seq, starts);
}

void phi_launch_sketch_bytes(hipStream_t st, int mode, const PhiSketchArgs &A, const unsigned long long *batch_bad)
{
    const int64_t nchunks = phi_sketch_num_blocks(A.n_bases);
    if (nchunks <= 0) return;
    int64_t nb = (nchunks + TPB / 64 - 1) / (TPB / 64);
    if (!A.allslow && nb > 1024) nb = 1024;           // usually leaves at once: keep the launch small
    // (count / ordered write of sequences with bases outside ACGT; the read kernel does such windows itself)
    if (mode == PHI_MODE_COUNT)
        hipLaunchKernelGGL(phi_sketch_bytes_kernel<PHI_MODE_COUNT>, dim3((unsigned)nb), dim3(TPB), 0, st, A, batch_bad);
    else
        hipLaunchKernelGGL(phi_sketch_bytes_kernel<PHI_MODE_WRITE>, dim3((unsigned)nb), dim3(TPB), 0, st, A, batch_bad);
}

void phi_launch_reset_reads(hipStream_t st, uint64_t *hit_words, int64_t n_hit_words, uint64_t *stripes, int64_t n_stripe_words)
{
    int64_t n = n_hit_words > n_stripe_words ? n_hit_words : n_stripe_words;
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(phi_reset_reads_kernel, dim3((unsigned)nb), dim3(256), 0, st, hit_words, n_hit_words, stripes, n_stripe_words);
}

// number of per-wave chunks (= entries of block_cnt / block_off)
int64_t phi_sketch_num_blocks(int64_t n_bases) { return n_bases <= 0 ? 0 : (n_bases + WCH - 1) / WCH; }

// t0 / t1 (optional): events that take the kernel's own start / end timestamps (hipExtLaunchKernelGGL:
// no extra marker packets on the stream, unlike hipEventRecord around the launch)
template <int MODE>
static void launch_sketch_mode(hipStream_t st, unsigned nb, size_t lds, const PhiSketchArgs &A, hipEvent_t t0, hipEvent_t t1, bool pooled)
{
    // the reference's defaults (options.cpp:7-8) get a fully unrolled instance, except for the
    // ordered write whose position tracking would push it past 128 VGPRs
    if constexpr (MODE == PHI_MODE_PROBE) {
        if (pooled) { phi_launch_sketch_pooled(st, nb, lds, A, t0, t1); return; }      // (k <= 32: see phi_launch_sketch)
    }
    if constexpr (MODE != PHI_MODE_WRITE) {
        if (A.k == 31 && A.w == 25) {
            hipExtLaunchKernelGGL((phi_sketch_kernel<MODE, true, 31, 25>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
            return;
        }
    }
    if constexpr (MODE == PHI_MODE_PROBE) {
        if (A.k > PHI_MAX_K_PACKED) {                      // reads with k > 32: staging, read starts and bookkeeping of the fused kernel, every window byte-wise
            hipExtLaunchKernelGGL((phi_sketch_kernel<MODE, true, -1, 0>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
            return;
        }
    }
    if (A.w > Q) hipExtLaunchKernelGGL((phi_sketch_kernel<MODE, true, 0, 0>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
    else hipExtLaunchKernelGGL((phi_sketch_kernel<MODE, false, 0, 0>), dim3(nb), dim3(TPB), lds, st, t0, t1, 0, A);
}

void phi_launch_sketch(hipStream_t st, int mode, const PhiSketchArgs &A0, hipEvent_t t0, hipEvent_t t1)
{
    const int64_t nchunks = phi_sketch_num_blocks(A0.n_bases);
    if (nchunks <= 0) return;
    PhiSketchArgs A = A0;
    // Reads (k <= 32), batches of 12 Mbases and more: wave g of `njobs` takes the chunks g, g + njobs, ... and hashes their
    // items in full rounds (POOL in the kernel).  Four times as many waves as the machine holds at once (256 CUs x 4 SIMDs
    // x 6 waves of this kernel): the SIMDs serve their oldest wave first, so waves of equal shares end far apart (the first
    // in half the time of the last) and it takes waves in waiting to fill the slots they leave -- measured at C3 / C5s:
    // 6 144 waves 417 / -, 12 288 424 / 439, 18 432 433 / 443, 24 576 447 / 451, 36 864 439 / 442 Gbases/s.  Fewer than
    // three chunks per wave are not worth a loop: smaller batches run one chunk per wave as before.
    int64_t njobs = nchunks;
    bool pooled = false;
    if (mode == PHI_MODE_PROBE && A.k <= PHI_MAX_K_PACKED) {
        int64_t slots = 4 * 256 * 4 * 6;
        int64_t min_chunks = 4 * 6144;
        if (const char *e = getenv("PHI_SKETCH_WAVES")) slots = atoll(e) > 0 ? atoll(e) : slots;
        if (const char *e = getenv("PHI_SKETCH_POOL_MIN")) min_chunks = atoll(e);
        if (nchunks >= min_chunks) {
            int64_t m = (nchunks + slots - 1) / slots;          // chunks per wave
            if (m < 3) m = 3;
            njobs = (nchunks + m - 1) / m;
            if (njobs > 0x7FFFFFFF) njobs = 0x7FFFFFFF;
            A.wave_stride = (int32_t)njobs;
            pooled = true;
        }
    }
    const unsigned nb = (unsigned)((njobs + TPB / 64 - 1) / (TPB / 64));
    const size_t lds = (size_t)phi_wave_region_u64(A.w, A.k, mode == PHI_MODE_WRITE) * 8 * (TPB / 64);
    if (mode == PHI_MODE_COUNT) launch_sketch_mode<PHI_MODE_COUNT>(st, nb, lds, A, t0, t1, false);
    else if (mode == PHI_MODE_WRITE) launch_sketch_mode<PHI_MODE_WRITE>(st, nb, lds, A, t0, t1, false);
    else launch_sketch_mode<PHI_MODE_PROBE>(st, nb, lds, A, t0, t1, pooled);
}

void phi_launch_scan_counts(hipStream_t st, const int32_t *cnt, int64_t n, int64_t *off)
{
    hipLaunchKernelGGL(phi_scan_counts_kernel, dim3(1), dim3(1024), 0, st, cnt, n, off);
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_sketch_kernel() {}
void phi_warm_sketch(hipStream_t st) { hipLaunchKernelGGL(phi_warm_sketch_kernel, dim3(1), dim3(64), 0, st); }
#endif
