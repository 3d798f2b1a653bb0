// table.hip -- open-addressed hash tables in HBM (linear probing, 64-bit keys, atomicCAS).
//
// Replaces the reference's std::map<uint64_t,int32_t> Sp_R (src/ILP_index.cpp:616-635) and the
// per-minimiser std::map::find of compute_anchors (:495-526).  Roles are swapped with respect to
// the reference so that reads can stream and shard across GPUs: the table is built once from
// the walk minimisers (the "index"), reads probe it; a second set holds the distinct read
// hashes only to report |Sp_R| (:641).
#include <hip/hip_runtime.h>
#include "phi_dev.h"
#include "phi_kernels.h"

__global__ void phi_fill_u64_kernel(uint64_t *p, int64_t n, uint64_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}
__global__ void phi_fill_u32_kernel(uint32_t *p, int64_t n, uint32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

static inline unsigned grid_for(int64_t n, int tpb)
{
    int64_t nb = (n + tpb - 1) / tpb;
    if (nb > 256 * 16) nb = 256 * 16;        // grid-stride beyond 16 workgroups per CU
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

__global__ void __launch_bounds__(256) phi_table_pairs_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ uid,
                                                              int64_t cap, ulonglong2 *__restrict__ kv)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        kv[i] = make_ulonglong2(key, key == PHI_EMPTY_KEY ? 0ull : (unsigned long long)uid[i]);
    }
}

void phi_launch_table_pairs(hipStream_t st, const uint64_t *keys, const uint32_t *uid, int64_t cap, uint64_t *kv)
{
    if (cap > 0)
        hipLaunchKernelGGL(phi_table_pairs_kernel, dim3(grid_for(cap, 256)), dim3(256), 0, st, keys, uid, cap, (ulonglong2 *)kv);
}

void phi_launch_fill_u64(hipStream_t st, uint64_t *p, int64_t n, uint64_t v)
{
    if (n > 0) hipLaunchKernelGGL(phi_fill_u64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, n, v);
}
__global__ void phi_iota_i64_kernel(int64_t *p, int64_t n, int64_t step)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = i * step;
}
void phi_launch_iota_i64(hipStream_t st, int64_t *p, int64_t n, int64_t step)
{
    if (n > 0) hipLaunchKernelGGL(phi_iota_i64_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, p, n, step);
}

void phi_launch_fill_u32(hipStream_t st, uint32_t *p, int64_t n, uint32_t v)
{
    if (n > 0) hipLaunchKernelGGL(phi_fill_u32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, n, v);
}

__global__ void __launch_bounds__(256) phi_table_build_kernel(const uint64_t *__restrict__ rec_hash, int64_t n_rec,
                                                              uint64_t *__restrict__ u_keys,
                                                              uint32_t *__restrict__ u_rep, uint64_t u_mask,
                                                              uint32_t *__restrict__ rec_slot,
                                                              uint32_t *__restrict__ err)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = rec_hash[i];
        if (h == PHI_EMPTY_KEY) { atomicOr(err, PHI_KERR_SENTINEL); rec_slot[i] = 0; continue; }
        uint64_t slot = h & u_mask;
        int probes = 0;
        for (;;) {
            // the walks of a pangenome share most minimisers: look before the atomic (a key, once
            // written, never changes)
            unsigned long long prev = __builtin_nontemporal_load((const unsigned long long *)&u_keys[slot]);
            if (prev == PHI_EMPTY_KEY) prev = atomicCAS((unsigned long long *)&u_keys[slot], PHI_EMPTY_KEY, h);
            if (prev == PHI_EMPTY_KEY || prev == h) break;
            slot = (slot + 1) & u_mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(err, PHI_KERR_TABLE_FULL); break; }
        }
        // smallest record index: same on every rank.  The stored value only falls, so a record that
        // sees a smaller one already there has nothing to add
        if (__builtin_nontemporal_load(&u_rep[slot]) > (uint32_t)i) atomicMin(&u_rep[slot], (uint32_t)i);
        rec_slot[i] = (uint32_t)slot;
    }
}

void phi_launch_table_build(hipStream_t st, const uint64_t *rec_hash, int64_t n_rec, uint64_t *u_keys,
                            uint32_t *u_rep, uint64_t u_mask, uint32_t *rec_slot, uint32_t *err)
{
    if (n_rec > 0)
        hipLaunchKernelGGL(phi_table_build_kernel, dim3(grid_for(n_rec, 256)), dim3(256), 0, st, rec_hash, n_rec,
                           u_keys, u_rep, u_mask, rec_slot, err);
}

// flags[i] = record i is the representative (first record) of its hash
__global__ void __launch_bounds__(256) phi_rep_flags_kernel(const uint32_t *__restrict__ rec_slot, int64_t n_rec,
                                                            const uint32_t *__restrict__ u_rep,
                                                            uint8_t *__restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (int64_t)gridDim.x * blockDim.x)
        flags[i] = u_rep[rec_slot[i]] == (uint32_t)i;
}
// dense minimiser ids: uid = rank of the representative record in position order, identical on
// every rank because the records are
__global__ void __launch_bounds__(256) phi_slot_uid_kernel(const int32_t *__restrict__ rep_list, int64_t n_unique,
                                                           const uint32_t *__restrict__ rec_slot,
                                                           uint32_t *__restrict__ u_uid)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_unique; j += (int64_t)gridDim.x * blockDim.x)
        u_uid[rec_slot[rep_list[j]]] = (uint32_t)j;
}
void phi_launch_rep_flags(hipStream_t st, const uint32_t *rec_slot, int64_t n_rec, const uint32_t *u_rep,
                          uint8_t *flags)
{
    if (n_rec > 0)
        hipLaunchKernelGGL(phi_rep_flags_kernel, dim3(grid_for(n_rec, 256)), dim3(256), 0, st, rec_slot, n_rec, u_rep,
                           flags);
}
void phi_launch_slot_uid(hipStream_t st, const int32_t *rep_list, int64_t n_unique, const uint32_t *rec_slot,
                         uint32_t *u_uid)
{
    if (n_unique > 0)
        hipLaunchKernelGGL(phi_slot_uid_kernel, dim3(grid_for(n_unique, 256)), dim3(256), 0, st, rep_list, n_unique,
                           rec_slot, u_uid);
}

// The table phi_table_build makes is sized by the RECORDS (every walk repeats most minimisers):
// 2^26 slots for ~10^6 distinct keys at C2, 768 MB that every read probe walks into at random.
// Once the distinct keys are known they are re-inserted into a table sized by THEM (load <= 0.25,
// 32 MB at C2: cache resident) and every record looks its new slot up.
__global__ void __launch_bounds__(256) phi_table_compact_insert_kernel(const int32_t *__restrict__ rep_list, int64_t n_unique,
                                                                       const uint64_t *__restrict__ rec_hash,
                                                                       uint64_t *__restrict__ keys, uint32_t *__restrict__ uid,
                                                                       uint64_t mask, uint32_t *__restrict__ err)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_unique; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = rec_hash[rep_list[i]];
        uint64_t slot = h & mask;
        int probes = 0;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long *)&keys[slot], PHI_EMPTY_KEY, h);
            if (prev == PHI_EMPTY_KEY) break;                  // the keys are distinct
            slot = (slot + 1) & mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(err, PHI_KERR_TABLE_FULL); break; }
        }
        uid[slot] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(256) phi_table_lookup_kernel(const uint64_t *__restrict__ rec_hash, int64_t n_rec,
                                                               const uint64_t *__restrict__ keys, uint64_t mask,
                                                               uint32_t *__restrict__ rec_slot, uint32_t *__restrict__ err)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = rec_hash[i];
        uint64_t slot = h & mask;
        int probes = 0;
        while (keys[slot] != h) {
            slot = (slot + 1) & mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(err, PHI_KERR_TABLE_FULL); break; }
        }
        rec_slot[i] = (uint32_t)slot;
    }
}

void phi_launch_table_compact(hipStream_t st, const int32_t *rep_list, int64_t n_unique, const uint64_t *rec_hash, int64_t n_rec,
                              uint64_t *keys, uint32_t *uid, uint64_t mask, uint32_t *rec_slot, uint32_t *err)
{
    if (n_unique > 0)
        hipLaunchKernelGGL(phi_table_compact_insert_kernel, dim3(grid_for(n_unique, 256)), dim3(256), 0, st, rep_list, n_unique,
                           rec_hash, keys, uid, mask, err);
    if (n_rec > 0)
        hipLaunchKernelGGL(phi_table_lookup_kernel, dim3(grid_for(n_rec, 256)), dim3(256), 0, st, rec_hash, n_rec, keys, mask,
                           rec_slot, err);
}

// hist[c] += 1 for every slot of the table that holds a key and occurs in c walks
__global__ void __launch_bounds__(256) phi_share_hist_kernel(const uint64_t *__restrict__ keys, int64_t cap,
                                                             const int32_t *__restrict__ n_walks_of,
                                                             unsigned long long *__restrict__ hist)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (int64_t)gridDim.x * blockDim.x)
        if (keys[s] != PHI_EMPTY_KEY) atomicAdd(&hist[n_walks_of[s]], 1ull);
}

void phi_launch_share_hist(hipStream_t st, const uint64_t *keys, int64_t cap, const int32_t *n_walks_of, unsigned long long *hist)
{
    if (cap > 0)
        hipLaunchKernelGGL(phi_share_hist_kernel, dim3(grid_for(cap, 256)), dim3(256), 0, st, keys, cap, n_walks_of, hist);
}

// u_keys != nullptr: a hash found in the walk-minimiser table sets its hit flag instead (the set holds
// only hashes absent from the table, see probe_tables in sketch.hip)
__global__ void __launch_bounds__(256) phi_spectrum_insert_kernel(const uint64_t *__restrict__ hashes, int64_t n,
                                                                  uint64_t *__restrict__ sp_keys, uint64_t sp_mask,
                                                                  unsigned long long *__restrict__ sp_count,
                                                                  const uint64_t *__restrict__ u_keys, uint64_t u_mask,
                                                                  const uint32_t *__restrict__ u_uid, uint8_t *__restrict__ hit,
                                                                  uint32_t *__restrict__ err)
{
    int n_new = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = hashes[i];
        if (h == PHI_EMPTY_KEY) continue;
        if (u_keys) {
            bool found = false;
            uint64_t su = h & u_mask;
            for (int probes = 0; probes <= PHI_MAX_PROBE; probes++) {
                const uint64_t key = u_keys[su];
                if (key == h) { hit[u_uid[su]] = 1; found = true; break; }
                if (key == PHI_EMPTY_KEY) break;
                su = (su + 1) & u_mask;
            }
            if (found) continue;
        }
        uint64_t slot = h & sp_mask;
        int probes = 0;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long *)&sp_keys[slot], PHI_EMPTY_KEY, h);
            if (prev == PHI_EMPTY_KEY) { n_new++; break; }
            if (prev == h) break;
            slot = (slot + 1) & sp_mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(err, PHI_KERR_TABLE_FULL); break; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_new += __shfl_xor(n_new, d, 64);
    if ((threadIdx.x & 63) == 0 && n_new)
        atomicAdd(sp_count + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) & (PHI_STRIPES - 1)) * 8, (unsigned long long)n_new);
}

void phi_launch_spectrum_insert(hipStream_t st, const uint64_t *hashes, int64_t n, uint64_t *sp_keys,
                                uint64_t sp_mask, unsigned long long *sp_count, const uint64_t *u_keys, uint64_t u_mask,
                                const uint32_t *u_uid, uint8_t *hit, uint32_t *err)
{
    if (n > 0)
        hipLaunchKernelGGL(phi_spectrum_insert_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, hashes, n, sp_keys,
                           sp_mask, sp_count, u_keys, u_mask, u_uid, hit, err);
}

// The read-spectrum set from the log of novel hashes (sketch.hip): thread -> entry (chunk, i); every lane of a wave
// that holds a logged hash inserts it, the wave adds its new entries to a striped counter.  All lanes of the launch are
// inserts (in the sketch kernel an insert was a lane's second dependent round trip behind its table probe).
__global__ void __launch_bounds__(256) phi_spectrum_flush_kernel(const uint64_t *__restrict__ nov_log, const uint16_t *__restrict__ nov_cnt,
                                                                 int64_t c_lo, int64_t n_ent, int32_t shift,
                                                                 uint64_t *__restrict__ sp_keys, uint64_t sp_mask,
                                                                 unsigned long long *__restrict__ sp_count, uint32_t *__restrict__ err)
{
    int n_new = 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_ent; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = c_lo + (t >> shift);
        const int i = (int)(t & ((1 << shift) - 1));
        if (i >= (int)nov_cnt[c]) continue;
        const uint64_t h = __builtin_nontemporal_load(&nov_log[(c << shift) + i]);
        uint64_t slot = h & sp_mask;
        int probes = 0;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long *)&sp_keys[slot], PHI_EMPTY_KEY, h);
            if (prev == PHI_EMPTY_KEY) { n_new++; break; }
            if (prev == h) break;
            slot = (slot + 1) & sp_mask;
            if (++probes > PHI_MAX_PROBE) { atomicOr(err, PHI_KERR_TABLE_FULL); break; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_new += __shfl_xor(n_new, d, 64);
    if ((threadIdx.x & 63) == 0 && n_new)
        atomicAdd(sp_count + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) & (PHI_STRIPES - 1)) * 8, (unsigned long long)n_new);
}

void phi_launch_spectrum_flush(hipStream_t st, const uint64_t *nov_log, const uint16_t *nov_cnt, int64_t c_lo, int64_t c_hi, int32_t shift,
                               uint64_t *sp_keys, uint64_t sp_mask, unsigned long long *sp_count, uint32_t *err)
{
    const int64_t n_ent = (c_hi - c_lo) << shift;
    if (n_ent > 0)
        hipLaunchKernelGGL(phi_spectrum_flush_kernel, dim3(grid_for(n_ent, 256)), dim3(256), 0, st, nov_log, nov_cnt, c_lo, n_ent, shift,
                           sp_keys, sp_mask, sp_count, err);
}

// number of set hit flags (bytes != 0), added to *n_out
__global__ void __launch_bounds__(256) phi_count_flags_kernel(const uint8_t *__restrict__ flags, int64_t n,
                                                              unsigned long long *__restrict__ n_out)
{
    int cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        cnt += flags[i] != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_out, (unsigned long long)cnt);
}

void phi_launch_count_flags(hipStream_t st, const uint8_t *flags, int64_t n, unsigned long long *n_out)
{
    if (n > 0) hipLaunchKernelGGL(phi_count_flags_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, flags, n, n_out);
}

__global__ void __launch_bounds__(256) phi_spectrum_export_kernel(const uint64_t *__restrict__ sp_keys, int64_t cap,
                                                                  uint64_t *__restrict__ out,
                                                                  unsigned long long *__restrict__ n_out)
{
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < cap; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const uint64_t key = i < cap ? sp_keys[i] : PHI_EMPTY_KEY;
        const bool occ = key != PHI_EMPTY_KEY;
        const unsigned long long bal = __ballot(occ);
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == 0 && bal) base = atomicAdd(n_out, (unsigned long long)__popcll(bal));
        base = __shfl(base, 0, 64);
        if (occ) out[base + __popcll(bal & ((1ull << lane) - 1))] = key;
    }
}

void phi_launch_spectrum_export(hipStream_t st, const uint64_t *sp_keys, int64_t cap, uint64_t *out,
                                unsigned long long *n_out)
{
    if (cap > 0)
        hipLaunchKernelGGL(phi_spectrum_export_kernel, dim3(grid_for(cap, 256)), dim3(256), 0, st, sp_keys, cap, out,
                           n_out);
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_table_kernel() {}
void phi_warm_table(hipStream_t st) { hipLaunchKernelGGL(phi_warm_table_kernel, dim3(1), dim3(64), 0, st); }
