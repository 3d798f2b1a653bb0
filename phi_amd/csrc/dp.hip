// dp.hip -- max-plus dynamic program over the haplotype-expanded DAG (replaces model.optimize()).
//
// The reference builds a MIQP/MILP on the expanded graph (src/ILP_index.cpp:1160-1409) and hands
// it to Gurobi (:1418).  With the anchor weights fixed, the same objective -- maximise
//     sum of weights of anchors whose hap-j edges are all traversed  -  2*(R/2) * #w-node uses
// over unit s->e flows -- is a longest-path problem whose state is
// (vertex v, walk h, q = number of consecutive hap-h edges just taken): an anchor of walk h that
// spans s edges and ends at v pays when q >= s.  Spans are <= k-1 <= 31, so run lengths >= 31
// are one class.
//
// Transitions (ILP_index.cpp:1203-1304):
//   stay      (u,h,q)   -> (next_h(u), h, q+1)                               cost 0
//   recombine (u,h',*)  -> (v,h,0) for an edge u->v with next_h'(u) != v     cost 2*(R/2)
//   start     (first(h), h, 0) = 0  (:1165-1195);   end at (last(h), h)       (:1388-1401)
// A walk that ends at u cannot leave u (its sink row is in - e = 0 and sum(e) = 1).
//
// Mapping: ONE workgroup, lane <-> walk, the topologically ordered vertices ("steps") are walked
// sequentially -- the chain is inherently serial, so the kernel is built to make one step cheap:
//   * the host lays the graph out as a STEP STREAM (32-byte record per step: flags, the live
//     in-edges as (steps back, out-edge index)) plus a 64-bit active-walk mask per step and wave;
//     the stream is staged through LDS in chunks, one chunk ahead;
//   * each lane keeps the scores of its last 31 run lengths as a DIFFERENCE ring in LDS
//     (slot-major, conflict-free): an anchor ending here adds +1 to a prefix of run lengths =
//     two ds_add; the run that turns 31 is folded into the scalar class L; only steps where a
//     recombination can leave the vertex (or the walk ends) pay the 31-term prefix-max;
//   * the best leaving states of recent vertices live in an LDS ring indexed by step, so a
//     recombination entry is two LDS look-ups; older ones fall back to the HBM copy;
//   * per-entry data (out-edge index + spans of the weight-1 anchors ending there, packed in one
//     64-bit word by phi_dp_words_kernel) sits in a per-lane LDS ring refilled in bulk every WD/2
//     steps: the step loop itself issues no global load, so no step waits on HBM latency (a
//     register prefetch "four entries ahead" made every step wait for the load it had just
//     issued -- vmcnt counts in order -- and cost 0.67 us per step).
#include <hip/hip_runtime.h>
#include "phi_kernels.h"

#define NEG (-(1 << 28))
#define CHK PHI_DP_CHUNK
#define RING PHI_DP_RING

// ------------------------------------------------------------------ per-run entry words
// word[e] = out-edge index (8 bits) | up to 11 spans of weight-1 anchors ending at e (5 bits each)
//           | overflow flag (bit 63: more than 11, the DP then walks the CSR for this entry)
__global__ void __launch_bounds__(256) phi_dp_words_kernel(const uint8_t *__restrict__ e_out,
                                                           const int64_t *__restrict__ g_off,
                                                           const uint8_t *__restrict__ g_span,
                                                           const uint8_t *__restrict__ a_weight, int64_t n_entries,
                                                           uint64_t *__restrict__ word)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries;
         e += (int64_t)gridDim.x * blockDim.x) {
        uint64_t wv = e_out[e];
        int n = 0;
        for (int64_t g = g_off[e]; g < g_off[e + 1]; g++) {
            if (!a_weight[g]) continue;
            if (n == 11) { wv |= 1ull << 63; break; }
            wv |= (uint64_t)(g_span[g] & 31) << (8 + 5 * n);
            n++;
        }
        word[e] = wv;
    }
}

void phi_launch_dp_words(hipStream_t st, const uint8_t *e_out, const int64_t *g_off, const uint8_t *g_span,
                         const uint8_t *a_weight, int64_t n_entries, uint64_t *word)
{
    if (n_entries <= 0) return;
    int64_t nb = (n_entries + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(phi_dp_words_kernel, dim3((unsigned)nb), dim3(256), 0, st, e_out, g_off, g_span, a_weight,
                       n_entries, word);
}

// ------------------------------------------------------------------ the DP
__device__ __forceinline__ unsigned long long pack_vh(int32_t val, int32_t h)
{
    // larger value wins, then the smaller walk id
    return ((unsigned long long)(uint32_t)(val - NEG) << 32) | (uint32_t)(0x7FFFFFFF - h);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(k, d, 64);
        k = o > k ? o : k;
    }
    return k;
}

template <int NW>   // waves in the workgroup
__global__ void __launch_bounds__(NW * 64) phi_dp_kernel(PhiDpArgs A)
{
    constexpr int NT = NW * 64;
    __shared__ int32_t s_rec[2][CHK][8];
    __shared__ unsigned long long s_mask[2][CHK][NW];
    __shared__ int32_t s_d[32][NT];                 // difference ring, slot-major: bank = lane
    __shared__ int32_t r_t1v[RING], r_t1h[RING], r_t1n[RING], r_t2v[RING], r_t2h[RING];
    __shared__ unsigned long long s_red[NW > 1 ? NW : 1];
    __shared__ int32_t s_oidx[NT];
    constexpr int WD = NW == 1 ? 64 : NW == 2 ? 32 : NW == 4 ? 16 : 4;   // per-lane ring of entry words
    __shared__ uint64_t s_w[WD][NT];

    const int h = threadIdx.x;
    const int lane = h & 63, wid = h >> 6;
    const bool has_walk = h < A.n_walks;
    const int64_t eb = has_walk ? A.walk_off[h] : 0;
    const int64_t ee = has_walk ? A.walk_off[h + 1] : 0;
    int64_t e = eb;                                  // next entry of this lane
    int64_t wl = eb;                                 // first entry not yet in the word ring
    int32_t T = NEG, L = NEG, entL = 0;

    // bring entries [wl, min(ee, e + WD)) into the ring: at most WD/2 per call (a lane consumes at
    // most one entry per step and this runs every WD/2 steps); all loads are issued before the
    // first ring write, so one call costs one memory latency
    auto refill = [&]() {
        const int64_t hi = min(ee, e + WD);
        uint64_t tmp[WD / 2];
#pragma unroll
        for (int j = 0; j < WD / 2; j++) {
            const int64_t x = wl + j;
            tmp[j] = A.word[has_walk ? min(x, ee - 1) : 0];
        }
#pragma unroll
        for (int j = 0; j < WD / 2; j++) {
            const int64_t x = wl + j;
            if (x < hi) s_w[x & (WD - 1)][h] = tmp[j];
        }
        wl = max(wl, min(hi, wl + WD / 2));
    };
    refill();
    refill();

    const int32_t n_steps = A.n_vtx;
    const int n_chunks = (n_steps + CHK - 1) / CHK;

    // stage chunk c of the step stream into LDS buffer c&1
    auto stage = [&](int c) {
        const int b = c & 1;
        const int32_t s0 = c * CHK;
        const int32_t ns = min(CHK, n_steps - s0);
        const int4 *src = reinterpret_cast<const int4 *>(A.st_rec + (int64_t)s0 * 8);
        int4 *dst = reinterpret_cast<int4 *>(&s_rec[b][0][0]);
        for (int i = h; i < ns * 2; i += NT) dst[i] = src[i];
        const unsigned long long *ms = A.st_mask + (int64_t)s0 * NW;
        unsigned long long *md = &s_mask[b][0][0];
        for (int i = h; i < ns * NW; i += NT) md[i] = ms[i];
    };
    stage(0);
    __syncthreads();

    for (int c = 0; c < n_chunks; c++) {
        const int b = c & 1;
        if (c + 1 < n_chunks) stage(c + 1);          // lands while this chunk is processed
        const int32_t s0 = c * CHK;
        const int32_t ns = min(CHK, n_steps - s0);
        for (int i = 0; i < ns; i++) {
            if ((i & (WD / 2 - 1)) == 0 && (c | i)) refill();
            const int32_t step = s0 + i;
            const int32_t flags = s_rec[b][i][0];
            const bool active = has_walk && ((s_mask[b][i][wid] >> lane) & 1ull);

            // ---- recombination entry into this vertex (uniform over the workgroup)
            int32_t E = NEG, Eh = -1, Esrc = -1;
            if (flags & PHI_DP_NEED_ENTRY) {
                const int n_in = (flags >> 8) & 0xFF;
                for (int j = 0; j < n_in; j++) {
                    const int32_t p = (j < 3) ? s_rec[b][i][2 + j] : A.in_packed[s_rec[b][i][1] + j - 3];
                    const int32_t back = (int32_t)((uint32_t)p >> 8), oj = p & 0xFF;
                    const int32_t src = step - back;
                    int32_t t1v, t1h, t1n, t2v, t2h;
                    if (back < RING) {
                        const int sl = src & (RING - 1);
                        t1v = r_t1v[sl]; t1h = r_t1h[sl]; t1n = r_t1n[sl]; t2v = r_t2v[sl]; t2h = r_t2h[sl];
                    } else {
                        const int32_t *g = A.tops + (int64_t)src * 5;
                        t1v = g[0]; t1h = g[1]; t1n = g[2]; t2v = g[3]; t2h = g[4];
                    }
                    const bool cont = t1n == oj;             // top1 continues along this edge: use top2
                    const int32_t val = cont ? t2v : t1v, hh = cont ? t2h : t1h;
                    if (hh < 0) continue;
                    if (val > E || (val == E && (hh < Eh || (hh == Eh && src < Esrc)))) { E = val; Eh = hh; Esrc = src; }
                }
                if (Eh >= 0) E -= A.cost;
                if (h == 0) { A.ent_src[step] = Esrc; A.ent_h[step] = Eh; }
            }

            int32_t dmax = NEG;
            int32_t oidx = 255;
            if (active) {
                const int32_t t = (int32_t)(e - eb);
                const uint64_t word = s_w[e & (WD - 1)][h];
                oidx = (int32_t)(word & 0xFF);
                if (t == 0) {
                    // walk start s_{first(h),h}: run length 0 scores 0, nothing older exists
#pragma unroll
                    for (int sidx = 0; sidx < 32; sidx++) s_d[sidx][h] = 0;
                    s_d[2][h] = NEG;                          // oldest (age 30) as an absolute value
                    s_d[0][h] = -NEG;                         // newest: 0 - NEG
                    T = 0; L = NEG; entL = 0;
                } else {
                    // the run that reaches length 31 leaves the ring and joins class L
                    const int so = (t + 1) & 31;
                    const int32_t v_old = s_d[so][h];
                    if (v_old > L) { L = v_old; entL = t - 31; }
                    atomicAdd(&s_d[(t + 2) & 31][h], v_old);  // next oldest becomes absolute
                    const int32_t Enew = (Eh >= 0) ? E : NEG;
                    s_d[t & 31][h] = Enew - T;
                    T = Enew;
                }
                // anchors of this walk ending here: +1 for every run length >= span
                uint64_t gw = word >> 8;
                if (word >> 63) {
                    for (int64_t g = A.g_off[e]; g < A.g_off[e + 1]; g++) {
                        if (!A.a_weight[g]) continue;
                        const int sp = A.g_span[g];
                        L += 1;
                        if (sp <= 30) { atomicAdd(&s_d[(t + 2) & 31][h], 1); atomicAdd(&s_d[(t - sp + 1) & 31][h], -1); }
                    }
                } else {
                    while (gw & 0x7FFFFFFFFFFFFFull) {
                        const int sp = (int)(gw & 31);
                        gw >>= 5;
                        if (sp == 0) continue;
                        L += 1;
                        if (sp <= 30) { atomicAdd(&s_d[(t + 2) & 31][h], 1); atomicAdd(&s_d[(t - sp + 1) & 31][h], -1); }
                    }
                }
                if ((flags & PHI_DP_NEED_TOPS) || e == ee - 1) {
                    // best run length: class L first (oldest), then ages 30..0; ties keep the older run
                    int32_t best = L, qb = 31, run = 0;
#pragma unroll
                    for (int a = 30; a >= 0; a--) {
                        run += s_d[(t - a) & 31][h];
                        if (run > best) { best = run; qb = a; }
                    }
                    if (best > NEG / 2) {
                        dmax = best;
                        A.dmax[e] = best; A.qbest[e] = (uint8_t)qb; A.lent[e] = entL;
                    } else {
                        A.dmax[e] = NEG; A.qbest[e] = 0; A.lent[e] = 0;
                    }
                }
                e++;
            }

            // ---- best states leaving this vertex, by out-edge: top1 overall, top2 the best on
            //      another out-edge.  Walks that end here do not leave.
            if (flags & PHI_DP_NEED_TOPS) {
                const bool leaving = active && oidx != 255 && dmax > NEG / 2;
                s_oidx[h] = oidx;
                const unsigned long long key = leaving ? pack_vh(dmax, h) : 0ull;
                unsigned long long k1 = wave_max_u64(key);
                if (NW > 1) {
                    if (lane == 0) s_red[wid] = k1;
                    __syncthreads();
                    k1 = s_red[0];
#pragma unroll
                    for (int x = 1; x < NW; x++) k1 = s_red[x] > k1 ? s_red[x] : k1;
                    __syncthreads();
                }
                int32_t t1v = NEG, t1h = -1, t1n = -1, t2v = NEG, t2h = -1;
                if (k1) {
                    t1v = (int32_t)(uint32_t)(k1 >> 32) + NEG;
                    t1h = 0x7FFFFFFF - (int32_t)(uint32_t)k1;
                    t1n = s_oidx[t1h];
                }
                unsigned long long k2 = wave_max_u64((leaving && oidx != t1n) ? key : 0ull);
                if (NW > 1) {
                    if (lane == 0) s_red[wid] = k2;
                    __syncthreads();
                    k2 = s_red[0];
#pragma unroll
                    for (int x = 1; x < NW; x++) k2 = s_red[x] > k2 ? s_red[x] : k2;
                }
                if (k2) {
                    t2v = (int32_t)(uint32_t)(k2 >> 32) + NEG;
                    t2h = 0x7FFFFFFF - (int32_t)(uint32_t)k2;
                }
                if (h == 0) {
                    const int sl = step & (RING - 1);
                    r_t1v[sl] = t1v; r_t1h[sl] = t1h; r_t1n[sl] = t1n; r_t2v[sl] = t2v; r_t2h[sl] = t2h;
                    int32_t *g = A.tops + (int64_t)step * 5;
                    g[0] = t1v; g[1] = t1h; g[2] = t1n; g[3] = t2v; g[4] = t2h;
                }
                if (NW > 1) __syncthreads();             // ring entry visible to the other waves
            }
        }
        __threadfence_block();
        __syncthreads();                                 // chunk c+1 staged; HBM tops visible
    }
}

void phi_launch_dp(hipStream_t st, const PhiDpArgs &A)
{
    const int nw = (A.n_walks + 63) / 64;
    if (nw <= 1) hipLaunchKernelGGL(phi_dp_kernel<1>, dim3(1), dim3(64), 0, st, A);
    else if (nw <= 2) hipLaunchKernelGGL(phi_dp_kernel<2>, dim3(1), dim3(128), 0, st, A);
    else if (nw <= 4) hipLaunchKernelGGL(phi_dp_kernel<4>, dim3(1), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(phi_dp_kernel<8>, dim3(1), dim3(512), 0, st, A);     // n_walks <= PHI_DP_MAX_WALKS
}

int phi_dp_num_waves(int n_walks)
{
    const int nw = (n_walks + 63) / 64;
    return nw <= 1 ? 1 : nw <= 2 ? 2 : nw <= 4 ? 4 : 8;
}
