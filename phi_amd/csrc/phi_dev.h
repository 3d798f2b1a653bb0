// phi_dev.h -- k-mer and hash primitives shared by the gfx950 kernels.
//
// Layout of packed sequences in HBM: uint64 words, 32 bases per word, 2 bits per base
// (A=0 C=1 G=2 T=3: ASCII order, so unsigned integer order of a left-aligned k-mer equals the
// std::string order the reference uses, ILP_index.cpp:394), base j of a word at bits
// [62-2j, 64-2j): a 64-bit load returns 32 bases with the first base most significant.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PHI_HD __host__ __device__ __forceinline__
#else
#define PHI_HD static inline
#endif

#define PHI_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull   // also the reference's initial prev_hash (:383, :455)

// 2-bit code of an ASCII base; valid iff the upper-cased byte is one of ACGT.
PHI_HD uint32_t phi_code(uint32_t c) { return ((c >> 1) ^ (c >> 2)) & 3u; }
PHI_HD bool phi_is_acgt(uint32_t c)
{
    c &= 0xDFu;
    return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

// 32 bases starting at base i, left-aligned (bases beyond the buffer read as the padding word).
PHI_HD uint64_t phi_extract64(const uint64_t *words, int64_t i)
{
    const int64_t wi = i >> 5;
    const int s = (int)(i & 31) * 2;
    const uint64_t hi = words[wi];
    if (s == 0) return hi;
    return (hi << s) | (words[wi + 1] >> (64 - s));
}

// last index e in [0, n) with off[e] <= g, for a monotone off[0..n] with off[0] <= g < off[n].  The items are
// of similar sizes, so the answer lies near g / mean size: gallop from that guess, then bisect the bracket
// (3-5 dependent loads instead of log2(n)).
PHI_HD int64_t phi_locate_in(const int64_t *off, int64_t n, int64_t g)
{
    int64_t lo = 0, hi = n;                           // invariant: off[lo] <= g < off[hi]
    const int64_t total = off[n];
    int64_t q = total > 0 ? (int64_t)((double)g / (double)total * (double)n) : 0;
    q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
    if (off[q] <= g) {
        lo = q;
        for (int64_t step = 1;; step <<= 1) {
            if (lo + step >= n) break;
            if (off[lo + step] > g) { hi = lo + step; break; }
            lo += step;
        }
    } else {
        hi = q;
        for (int64_t step = 1;; step <<= 1) {
            if (hi - step <= 0) break;
            if (off[hi - step] <= g) { lo = hi - step; break; }
            hi -= step;
        }
    }
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

PHI_HD uint64_t phi_kmask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }

// reverse complement of a right-aligned k-mer value
PHI_HD uint64_t phi_revcomp(uint64_t f, int k)
{
    uint64_t x = ~f;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
}

PHI_HD uint64_t phi_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// low 64 bits of a * c from three 32-bit multiplies (measured on gfx950: v_mul_lo_u32 issues at the rate of
// v_add_u32, v_mad_u64_u32 at 2/3 of it -- profiles/r02d_ubench_valu_rates.txt -- so the count of instructions,
// not their kind, is what matters):
// one full 32x32 -> 64 product and the low halves of the two cross products
PHI_HD uint64_t phi_mul64(uint64_t a, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32), clo = (uint32_t)c, chi = (uint32_t)(c >> 32);
    const uint64_t t = (uint64_t)alo * clo;
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32) + alo * chi + ahi * clo;
    // opaque to the optimiser: it would fold the shifts of a following rotate into further multiplies
    asm("" : "+v"(lo), "+v"(hi));
    return ((uint64_t)hi << 32) | lo;
#else
    return a * c;
#endif
}

// h * 5 + c by shift and add (a 64-bit multiply-add would be three multiply instructions)
PHI_HD uint64_t phi_x5_plus(uint64_t h, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t q = h << 2;
    asm("" : "+v"(q));
    return q + h + c;
#else
    return h * 5 + c;
#endif
}

PHI_HD uint64_t phi_fmix64(uint64_t k)
{
    k ^= k >> 33; k = phi_mul64(k, 0xff51afd7ed558ccdull);
    k ^= k >> 33; k = phi_mul64(k, 0xc4ceb9fe1a85ec53ull);
    k ^= k >> 33;
    return k;
}

// 4 bases (8 bits, first base in the top two bits) -> 4 ASCII bytes, first base in byte 0.
PHI_HD uint32_t phi_ascii4(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t = __umul24(x, 0x100401u);                            // x | x << 10 | x << 20, full rate
    asm("" : "+v"(t));                                              // (keeps it a 24-bit multiply)
#else
    uint32_t t = x | (x << 10);
    t = t | (x << 20);
#endif
    const uint32_t sel = ((t << 4) | (x >> 6)) & 0x03030303u;      // byte j = code of base j
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0x54474341u, 0x54474341u, sel);    // "ACGT" byte lookup
#else
    uint32_t r = 0;
    for (int j = 0; j < 4; j++) r |= ((0x54474341u >> (8 * ((sel >> (8 * j)) & 3))) & 0xFFu) << (8 * j);
    return r;
#endif
}

// 8 bases (16 bits, first base in the top two bits) -> 8 ASCII bytes as a little-endian u64.
PHI_HD uint64_t phi_ascii8(uint32_t x16)
{
    return (uint64_t)phi_ascii4((x16 >> 8) & 0xFFu) | ((uint64_t)phi_ascii4(x16 & 0xFFu) << 32);
}

// hash128_to_64 (ILP_index.cpp:10-18) of k <= 32 bytes given as four little-endian 8-byte lanes
// (bytes beyond k are zero): MurmurHash3_x64_128 (MurmurHash3.cpp:255-332), seed 0, h1 ^ h2.
PHI_HD uint64_t phi_murmur_lanes(uint64_t e0, uint64_t e1, uint64_t e2, uint64_t e3, int k)
{
    const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    uint64_t h1 = 0, h2 = 0;
    const int nblocks = k >> 4;
    uint64_t t1, t2;                                               // tail lanes
    if (nblocks >= 1) {
        uint64_t k1 = e0, k2 = e1;
        k1 = phi_mul64(k1, c1); k1 = phi_rotl64(k1, 31); k1 = phi_mul64(k1, c2); h1 ^= k1;
        h1 = phi_rotl64(h1, 27); h1 += h2; h1 = phi_x5_plus(h1, 0x52dce729);
        k2 = phi_mul64(k2, c2); k2 = phi_rotl64(k2, 33); k2 = phi_mul64(k2, c1); h2 ^= k2;
        h2 = phi_rotl64(h2, 31); h2 += h1; h2 = phi_x5_plus(h2, 0x38495ab5);
        t1 = e2; t2 = e3;
    } else {
        t1 = e0; t2 = e1;
    }
    if (nblocks == 2) {                                            // k == 32: second full block
        uint64_t k1 = t1, k2 = t2;
        k1 = phi_mul64(k1, c1); k1 = phi_rotl64(k1, 31); k1 = phi_mul64(k1, c2); h1 ^= k1;
        h1 = phi_rotl64(h1, 27); h1 += h2; h1 = phi_x5_plus(h1, 0x52dce729);
        k2 = phi_mul64(k2, c2); k2 = phi_rotl64(k2, 33); k2 = phi_mul64(k2, c1); h2 ^= k2;
        h2 = phi_rotl64(h2, 31); h2 += h1; h2 = phi_x5_plus(h2, 0x38495ab5);
    } else {
        const int rem = k & 15;
        if (rem > 8) { uint64_t k2 = t2; k2 = phi_mul64(k2, c2); k2 = phi_rotl64(k2, 33); k2 = phi_mul64(k2, c1); h2 ^= k2; }
        if (rem > 0) { uint64_t k1 = t1; k1 = phi_mul64(k1, c1); k1 = phi_rotl64(k1, 31); k1 = phi_mul64(k1, c2); h1 ^= k1; }
    }
    h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
    h1 += h2; h2 += h1;
    h1 = phi_fmix64(h1); h2 = phi_fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;
}

// The same hash for k <= 64 bytes given as eight little-endian 8-byte lanes (bytes beyond k are zero): k-mers longer
// than 32 bases, which only the exact byte-wise path handles (MurmurHash3.cpp:255-332: 16-byte blocks, then the tail).
PHI_HD uint64_t phi_murmur_lanes8(const uint64_t *e, int k)
{
    const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    uint64_t h1 = 0, h2 = 0;
    const int nblocks = k >> 4;
    for (int b = 0; b < nblocks; b++) {
        uint64_t k1 = e[2 * b], k2 = e[2 * b + 1];
        k1 = phi_mul64(k1, c1); k1 = phi_rotl64(k1, 31); k1 = phi_mul64(k1, c2); h1 ^= k1;
        h1 = phi_rotl64(h1, 27); h1 += h2; h1 = phi_x5_plus(h1, 0x52dce729);
        k2 = phi_mul64(k2, c2); k2 = phi_rotl64(k2, 33); k2 = phi_mul64(k2, c1); h2 ^= k2;
        h2 = phi_rotl64(h2, 31); h2 += h1; h2 = phi_x5_plus(h2, 0x38495ab5);
    }
    const int rem = k & 15;
    if (rem > 8) { uint64_t k2 = e[2 * nblocks + 1]; k2 = phi_mul64(k2, c2); k2 = phi_rotl64(k2, 33); k2 = phi_mul64(k2, c1); h2 ^= k2; }
    if (rem > 0) { uint64_t k1 = e[2 * nblocks]; k1 = phi_mul64(k1, c1); k1 = phi_rotl64(k1, 31); k1 = phi_mul64(k1, c2); h1 ^= k1; }
    h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
    h1 += h2; h2 += h1;
    h1 = phi_fmix64(h1); h2 = phi_fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;
}

// The same hash of the k ASCII bytes spelled by a right-aligned 2-bit k-mer value.  1 <= k <= 32.
PHI_HD uint64_t phi_kmer_hash(uint64_t val, int k)
{
    const uint64_t L = val << (64 - 2 * k);                        // base i at bits [62-2i]
    uint64_t e[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        uint64_t a = phi_ascii8((uint32_t)(L >> (48 - 16 * g)) & 0xFFFFu);
        const int nb = k - 8 * g;                                   // bytes of this lane in use
        if (nb <= 0) a = 0;
        else if (nb < 8) a &= (1ull << (8 * nb)) - 1;
        e[g] = a;
    }
    return phi_murmur_lanes(e[0], e[1], e[2], e[3], k);
}
