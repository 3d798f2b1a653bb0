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
// starts.  A window is live iff no sequence starts inside it.  Each workgroup owns PHI_CH
// consecutive window positions:
//   phase 1  canonical k-mers of the chunk (rolling 2-bit arithmetic)      -> LDS
//   phase 2  sliding-window minima, Q+1 consecutive windows per lane from a
//            suffix-min / core / prefix-min split (w+Q LDS reads per lane)
//   phase 3  candidate windows (minimum changed, or first window of a sequence), compacted
//            into LDS with a wave scan so that the hash runs on dense lanes only
//   phase 4  murmur3 of each candidate, hash-change test against its predecessor
//   phase 5  ballot/prefix-sum compaction of the emitted records and, by mode,
//            count | ordered write | open-addressed spectrum insert + table probe
#include <hip/hip_runtime.h>
#include "phi_dev.h"
#include "phi_kernels.h"

#define TPB PHI_TPB
#define CH PHI_CH
#define Q (CH / TPB)

// ---------------------------------------------------------------------------------- packing

// 32 ASCII bases per lane -> one packed word.  n_bad counts bytes outside ACGTacgt.
__global__ void __launch_bounds__(256) phi_pack_ascii_kernel(const uint8_t *__restrict__ bases, int64_t n,
                                                             uint64_t *__restrict__ words, int64_t n_words,
                                                             unsigned long long *__restrict__ n_bad)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const int64_t b0 = wi * 32;
    uint64_t word = 0;
    int bad = 0;
    if (b0 + 32 <= n && ((uintptr_t)(bases + b0) & 15) == 0) {
        const uint4 *p = reinterpret_cast<const uint4 *>(bases + b0);
        uint4 v[2] = {p[0], p[1]};
        const uint32_t *u = reinterpret_cast<const uint32_t *>(v);
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t c = (u[q] >> (8 * j)) & 0xFFu;
                bad += !phi_is_acgt(c);
                word = (word << 2) | phi_code(c);
            }
        }
    } else {
        for (int j = 0; j < 32; j++) {
            uint32_t c = 'A';
            if (b0 + j < n) { c = bases[b0 + j]; bad += !phi_is_acgt(c); }
            word = (word << 2) | phi_code(c);
        }
    }
    words[wi] = word;
    if (bad) atomicAdd(n_bad, (unsigned long long)bad);
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

// Walk sequences gathered straight into packed words: lane -> 32 bases of the flat walk space.
// ebase[e] = flat base offset of walk entry e (monotone, ebase[n_entries] = total bases).
__global__ void __launch_bounds__(256) phi_pack_walks_kernel(const uint8_t *__restrict__ seq_concat,
                                                             const int64_t *__restrict__ seq_off,
                                                             const int32_t *__restrict__ walk_vtx,
                                                             const int64_t *__restrict__ ebase, int64_t n_entries,
                                                             uint64_t *__restrict__ words, int64_t n_words,
                                                             unsigned long long *__restrict__ n_bad)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const int64_t total = ebase[n_entries];
    const int64_t b0 = wi * 32;
    uint64_t word = 0;
    int bad = 0;
    if (b0 < total) {
        // last entry e with ebase[e] <= b0
        int64_t lo = 0, hi = n_entries;           // invariant: ebase[lo] <= b0 < ebase[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (ebase[mid] <= b0) lo = mid; else hi = mid;
        }
        int64_t e = lo;
        int64_t eend = ebase[e + 1];
        const uint8_t *src = seq_concat + seq_off[walk_vtx[e]] - ebase[e];
        for (int j = 0; j < 32; j++) {
            const int64_t b = b0 + j;
            uint32_t c = 'A';
            if (b < total) {
                while (b >= eend) {               // skip to the entry that owns base b
                    e++;
                    eend = ebase[e + 1];
                    src = seq_concat + seq_off[walk_vtx[e]] - ebase[e];
                }
                c = src[b];
                bad += !phi_is_acgt(c);
            }
            word = (word << 2) | phi_code(c);
        }
    }
    words[wi] = word;
    if (bad) atomicAdd(n_bad, (unsigned long long)bad);
}

// ---------------------------------------------------------------------------------- helpers

// exclusive prefix sum of one int per lane over the workgroup; *total = sum.  s_w: [TPB/64+1].
__device__ __forceinline__ int block_excl_scan(int x, int *total, int *s_w)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int v = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    if (lane == 63) s_w[wid] = v;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < TPB / 64; i++) {
        const int s = s_w[i];
        if (i < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + v - x;
}

// position of the first start bit in [from, limit), or INT64_MAX
__device__ __forceinline__ int64_t next_start(const unsigned long long *__restrict__ starts, int64_t from,
                                              int64_t limit)
{
    if (from >= limit) return INT64_MAX;
    int64_t wi = from >> 6;
    const int64_t wl = (limit - 1) >> 6;
    unsigned long long word = starts[wi] & (~0ull << (from & 63));
    for (;;) {
        if (word) {
            const int64_t p = (wi << 6) + (__ffsll((long long)word) - 1);
            return p < limit ? p : INT64_MAX;
        }
        if (++wi > wl) return INT64_MAX;
        word = starts[wi];
    }
}

struct MinEnt { uint64_t v; int i; };
// b lies to the right of a: ties go right (the reference's deque pops on >=, ILP_index.cpp:397)
__device__ __forceinline__ MinEnt take_right(MinEnt a, MinEnt b) { return (b.v <= a.v) ? b : a; }

// ---------------------------------------------------------------------------------- sketch

template <int MODE>
__global__ void __launch_bounds__(TPB) phi_sketch_kernel(PhiSketchArgs A)
{
    __shared__ uint64_t s_m[CH + PHI_MAX_W + 8];   // canonical k-mers, later candidate values
    __shared__ uint64_t s_hash[CH];
    __shared__ uint32_t s_meta[CH];
    __shared__ uint64_t s_prev;
    __shared__ uint64_t s_hprev;
    __shared__ int s_w[TPB / 64 + 1];

    const int tid = threadIdx.x;
    const int k = A.k, w = A.w;
    const int64_t N = A.n_bases;
    const int64_t c0 = (int64_t)blockIdx.x * CH;          // first window start of this chunk
    const uint64_t kmask = phi_kmask(k);
    const int M = CH + w;                                 // canonical values m[l], l -> k-mer c0-1+l

    // ---- phase 1: canonical k-mers
    {
        const int P = (M + TPB - 1) / TPB;
        const int l0 = tid * P;
        const int64_t j0 = c0 - 1 + l0;
        if (l0 < M) {
            const int l1 = min(l0 + P, M);
            uint64_t F = 0, R = 0, nxt = 0;
            bool live = false;
            for (int l = l0; l < l1; l++) {
                const int64_t j = j0 + (l - l0);
                uint64_t m = ~0ull;
                if (j >= 0 && j + k <= N) {
                    if (!live) {
                        F = phi_extract64(A.words, j) >> (64 - 2 * k);
                        R = phi_revcomp(F, k);
                        nxt = phi_extract64(A.words, j + k);   // bases j+k .. j+k+31
                        live = true;
                    } else {
                        const uint64_t b = nxt >> 62;
                        nxt <<= 2;
                        F = ((F << 2) | b) & kmask;
                        R = (R >> 2) | ((3 - b) << (2 * k - 2));
                    }
                    m = F < R ? F : R;
                }
                s_m[l] = m;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: minima of windows la = tid*Q .. tid*Q+Q  (window la = m[la .. la+w))
    uint64_t wv[Q + 1];
    int wp[Q + 1];
    {
        const int base = tid * Q;
        if (w > Q) {
            MinEnt L[Q + 1];                      // L[i] = min of m[base+i .. base+Q), ties right
            L[Q].v = 0; L[Q].i = -1;
#pragma unroll
            for (int i = Q - 1; i >= 0; i--) {
                MinEnt e; e.v = s_m[base + i]; e.i = base + i;
                L[i] = (i == Q - 1) ? e : take_right(e, L[i + 1]);
            }
            MinEnt core; core.v = s_m[base + Q]; core.i = base + Q;
            for (int x = base + Q + 1; x < base + w; x++) {
                MinEnt e; e.v = s_m[x]; e.i = x;
                core = take_right(core, e);
            }
            MinEnt Rr; Rr.v = 0; Rr.i = -1;       // min of m[base+w .. base+w+i)
#pragma unroll
            for (int i = 0; i <= Q; i++) {
                MinEnt t = (i < Q) ? take_right(L[i], core) : core;
                if (i > 0) {
                    MinEnt e; e.v = s_m[base + w + i - 1]; e.i = base + w + i - 1;
                    Rr = (i == 1) ? e : take_right(Rr, e);
                    t = take_right(t, Rr);
                }
                wv[i] = t.v; wp[i] = t.i;
            }
        } else {
#pragma unroll
            for (int i = 0; i <= Q; i++) {
                MinEnt t; t.v = s_m[base + i]; t.i = base + i;
                for (int x = 1; x < w; x++) {
                    MinEnt e; e.v = s_m[base + i + x]; e.i = base + i + x;
                    t = take_right(t, e);
                }
                wv[i] = t.v; wp[i] = t.i;
            }
        }
    }

    // ---- phase 3: candidate windows of this lane: outputs i = 1..Q, window start a = c0-1+tid*Q+i
    uint32_t cflag = 0, fflag = 0;
    {
        const int64_t a0 = c0 - 1 + (int64_t)tid * Q;
        const int span = w + k - 1;                       // bases under one window
        int64_t ns = next_start(A.starts, a0 + 2, min(N, a0 + Q + span + 1));
#pragma unroll
        for (int i = 1; i <= Q; i++) {
            const int64_t a = a0 + i;
            if (ns <= a) ns = next_start(A.starts, a + 1, min(N, a0 + Q + span + 1));
            const bool valid = (a + span <= N) && (ns > a + span - 1);
            if (valid) {
                const bool first = (A.starts[a >> 6] >> (a & 63)) & 1ull;
                if (first || wv[i] != wv[i - 1]) {
                    cflag |= 1u << i;
                    if (first) fflag |= 1u << i;
                }
            }
        }
    }
    int ncand;
    const int coff = block_excl_scan(__popc(cflag), &ncand, s_w);   // barriers: s_m reads are done
    {
        int c = coff;
#pragma unroll
        for (int i = 1; i <= Q; i++) {
            if (cflag & (1u << i)) {
                s_m[c] = wv[i];
                s_meta[c] = (uint32_t)(tid * Q + i) | ((uint32_t)wp[i] << 12) | ((fflag >> i) & 1u) << 31;
                if (c == 0) s_prev = wv[i - 1];
                c++;
            }
        }
    }
    __syncthreads();

    // ---- phase 4: hash candidates on dense lanes
    for (int c = tid; c < ncand; c += TPB) s_hash[c] = phi_kmer_hash(s_m[c], k);
    if (tid == TPB - 1 && ncand > 0) s_hprev = phi_kmer_hash(s_prev, k);
    __syncthreads();

    // ---- phase 5: hash-change test, ordered compaction, output
    int64_t out_base = 0;
    if (MODE == PHI_MODE_WRITE) out_base = A.block_off[blockIdx.x];
    int n_emit = 0, n_new = 0;
    for (int r0 = 0; r0 < ncand; r0 += TPB) {
        const int c = r0 + tid;
        bool emit = false;
        uint64_t h = 0;
        uint32_t meta = 0;
        if (c < ncand) {
            h = s_hash[c];
            meta = s_meta[c];
            const uint64_t hp = (meta >> 31) ? PHI_EMPTY_KEY : (c == 0 ? s_hprev : s_hash[c - 1]);
            emit = h != hp;
        }
        const unsigned long long bal = __ballot(emit);
        const int lane = tid & 63, wid = tid >> 6;
        if (lane == 0) s_w[wid] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < TPB / 64; i++) {
            const int s = s_w[i];
            if (i < wid) woff += s;
            tot += s;
        }
        __syncthreads();
        if (emit) {
            const int rank = n_emit + woff + __popcll(bal & ((1ull << lane) - 1));
            if (MODE == PHI_MODE_WRITE) {
                A.out_hash[out_base + rank] = h;
                A.out_pos[out_base + rank] = c0 - 1 + (int64_t)((meta >> 12) & 0x7FFFFu);
            } else if (MODE == PHI_MODE_PROBE) {
                if (h == PHI_EMPTY_KEY) {
                    atomicOr(A.err, PHI_KERR_SENTINEL);
                } else {
                    // read spectrum: open-addressed insert (ILP_index.cpp:622-635 keeps a set)
                    uint64_t slot = h & A.sp_mask;
                    int probes = 0;
                    for (;;) {
                        const unsigned long long prev =
                            atomicCAS((unsigned long long *)&A.sp_keys[slot], PHI_EMPTY_KEY, h);
                        if (prev == PHI_EMPTY_KEY) { n_new++; break; }
                        if (prev == h) break;
                        slot = (slot + 1) & A.sp_mask;
                        if (++probes > PHI_MAX_PROBE) { atomicOr(A.err, PHI_KERR_TABLE_FULL); break; }
                    }
                    // walk-minimiser table: lookup, mark the minimiser as hit
                    slot = h & A.u_mask;
                    for (probes = 0; probes <= PHI_MAX_PROBE; probes++) {
                        const uint64_t key = A.u_keys[slot];
                        if (key == h) { A.hit[A.u_uid[slot]] = 1; break; }
                        if (key == PHI_EMPTY_KEY) break;
                        slot = (slot + 1) & A.u_mask;
                    }
                }
            }
        }
        n_emit += tot;
    }
    if (MODE == PHI_MODE_COUNT) {
        if (tid == 0) A.block_cnt[blockIdx.x] = n_emit;
    } else if (MODE == PHI_MODE_PROBE) {
        // one atomic per workgroup for the number of new spectrum entries and emitted records
        int tot_new;
        block_excl_scan(n_new, &tot_new, s_w);
        if (tid == 0) {
            if (tot_new) atomicAdd(A.sp_count, (unsigned long long)tot_new);
            if (n_emit) atomicAdd(A.n_emitted, (unsigned long long)n_emit);
        }
    }
}

// single-workgroup exclusive scan of the per-chunk counts (launch-bound, tiny)
__global__ void __launch_bounds__(1024) phi_scan_counts_kernel(const int32_t *__restrict__ cnt, int64_t n,
                                                               int64_t *__restrict__ off)
{
    __shared__ int64_t s_part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = min(n, tid * per), hi = min(n, lo + per);
    int64_t s = 0;
    for (int64_t i = lo; i < hi; i++) s += cnt[i];
    s_part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int i = 0; i < 1024; i++) { const int64_t t = s_part[i]; s_part[i] = run; run += t; }
        off[n] = run;
    }
    __syncthreads();
    int64_t run = s_part[tid];
    for (int64_t i = lo; i < hi; i++) { off[i] = run; run += cnt[i]; }
}

// ---------------------------------------------------------------------------------- launchers

void phi_launch_pack_ascii(hipStream_t st, const uint8_t *bases, int64_t n, uint64_t *words, int64_t n_words,
                           unsigned long long *n_bad)
{
    if (n_words <= 0) return;
    const int64_t nb = (n_words + 255) / 256;
    hipLaunchKernelGGL(phi_pack_ascii_kernel, dim3((unsigned)nb), dim3(256), 0, st, bases, n, words, n_words, n_bad);
}

void phi_launch_mark_starts(hipStream_t st, const int64_t *seq_off, int64_t n_seq, unsigned long long *starts)
{
    if (n_seq <= 0) return;
    const int64_t nb = (n_seq + 255) / 256;
    hipLaunchKernelGGL(phi_mark_starts_kernel, dim3((unsigned)nb), dim3(256), 0, st, seq_off, n_seq, starts);
}

void phi_launch_pack_walks(hipStream_t st, const uint8_t *seq_concat, const int64_t *seq_off,
                           const int32_t *walk_vtx, const int64_t *ebase, int64_t n_entries, uint64_t *words,
                           int64_t n_words, unsigned long long *n_bad)
{
    if (n_words <= 0) return;
    const int64_t nb = (n_words + 255) / 256;
    hipLaunchKernelGGL(phi_pack_walks_kernel, dim3((unsigned)nb), dim3(256), 0, st, seq_concat, seq_off, walk_vtx,
                       ebase, n_entries, words, n_words, n_bad);
}

int64_t phi_sketch_num_blocks(int64_t n_bases) { return n_bases <= 0 ? 0 : (n_bases + CH - 1) / CH; }

void phi_launch_sketch(hipStream_t st, int mode, const PhiSketchArgs &A)
{
    const int64_t nb = phi_sketch_num_blocks(A.n_bases);
    if (nb <= 0) return;
    if (mode == PHI_MODE_COUNT)
        hipLaunchKernelGGL(phi_sketch_kernel<PHI_MODE_COUNT>, dim3((unsigned)nb), dim3(TPB), 0, st, A);
    else if (mode == PHI_MODE_WRITE)
        hipLaunchKernelGGL(phi_sketch_kernel<PHI_MODE_WRITE>, dim3((unsigned)nb), dim3(TPB), 0, st, A);
    else
        hipLaunchKernelGGL(phi_sketch_kernel<PHI_MODE_PROBE>, dim3((unsigned)nb), dim3(TPB), 0, st, A);
}

void phi_launch_scan_counts(hipStream_t st, const int32_t *cnt, int64_t n, int64_t *off)
{
    hipLaunchKernelGGL(phi_scan_counts_kernel, dim3(1), dim3(1024), 0, st, cnt, n, off);
}
