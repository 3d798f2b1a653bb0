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
#include <stdlib.h>
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

// max over the 64 lanes on the VALU: DPP rotations inside each row of 16 lanes, then the four row
// results through v_readlane.  (__shfl_xor is ds_bpermute: six LDS round trips per reduction.)
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));     // row_ror:1
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));     // row_ror:2
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));     // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));     // row_ror:8
    const int32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// tops of a step packed in 16 bytes: x = top1 value, y = top2 value,
// z = (top1 walk + 1) | (top1 out-edge + 1) << 10 | (top2 walk + 1) << 20
__device__ __forceinline__ int4 pack_tops(int32_t t1v, int32_t t1h, int32_t t1n, int32_t t2v, int32_t t2h)
{
    return make_int4(t1v, t2v, (t1h + 1) | ((t1n + 1) << 10) | ((t2h + 1) << 20), 0);
}

template <int NW>   // waves in the workgroup
__global__ void __launch_bounds__(NW * 64) phi_dp_kernel(PhiDpArgs A)
{
    constexpr int NT = NW * 64;
    // Sixteen waves (513..1022 walks: the workgroup is the largest the hardware runs, a walk id has 10 bits in the packed tops):
    // the difference ring alone is 128 of the 160 KB of LDS, so the entry words come straight from HBM (no per-lane ring), the
    // step stream is staged in shorter chunks and fewer recent steps keep their leaving states in LDS.  Slower per step; the
    // same transitions, tie-breaks and outputs.
    constexpr bool WORDS_IN_LDS = NW < 16;
    constexpr int WD = NW == 1 ? 64 : NW == 2 ? 32 : NW == 4 ? 16 : 4;   // per-lane ring of entry words
    constexpr int WP = WD / 2;                                           // refill period in steps
    constexpr int CK = NW == 16 ? 32 : CHK;
    constexpr int RG = NW == 16 ? 256 : RING;
    __shared__ int4 s_rec[2][CK][2];
    __shared__ unsigned long long s_mask[2][CK][NW];
    __shared__ int32_t s_d[32][NT];                 // difference ring, slot-major: bank = lane
    __shared__ int4 s_top[RG];                      // packed tops of recent steps
    __shared__ unsigned long long s_red[NW > 1 ? NW : 1];
    __shared__ int32_t s_oidx[NT];
    __shared__ uint64_t s_w[WORDS_IN_LDS ? WD : 1][WORDS_IN_LDS ? NT : 1];

    const int h = threadIdx.x;
    const int lane = h & 63, wid = h >> 6;
    const bool has_walk = h < A.n_walks;
    const int64_t eb = has_walk ? A.walk_off[h] : 0;
    const int64_t ee = has_walk ? A.walk_off[h + 1] : 0;
    int64_t e = eb;                                  // next entry of this lane
    int64_t wl = eb;                                 // first entry not yet in the word ring
    int32_t T = NEG, L = NEG, entL = 0;

    // Word ring refill, split in two so that no step waits on HBM: `issue` starts the loads of
    // entries [wl, wl + WP) into registers, `land` (WP/2 steps later) writes those below
    // min(ee, e_at_issue + WD) into the ring.  A lane consumes at most one entry per step, so the
    // ring always holds the entries of the next WP/2 steps at least.
    uint64_t wtmp[WP];
    int64_t whi = eb;
    auto issue = [&]() {
        if (!WORDS_IN_LDS) return;
        whi = min(ee, e + WD);
#pragma unroll
        for (int j = 0; j < WP; j++) wtmp[j] = A.word[has_walk ? min(wl + j, ee - 1) : 0];
    };
    auto land = [&]() {
        if (!WORDS_IN_LDS) return;
#pragma unroll
        for (int j = 0; j < WP; j++) {
            const int64_t x = wl + j;
            if (x < whi) s_w[x & (WD - 1)][h] = wtmp[j];
        }
        wl = max(wl, min(whi, wl + WP));
    };
    issue(); land();
    issue(); land();
    // per-lane values of the NEXT entry, read right after the previous one was processed
    uint64_t nword = !has_walk ? 0 : WORDS_IN_LDS ? s_w[e & (WD - 1)][h] : A.word[e];
    int32_t nold = 0;

    const int32_t n_steps = A.n_vtx;
    const int n_chunks = (n_steps + CK - 1) / CK;

    // stage chunk c of the step stream into LDS buffer c&1
    auto stage = [&](int c) {
        const int b = c & 1;
        const int32_t s0 = c * CK;
        const int32_t ns = min(CK, n_steps - s0);
        const int4 *src = reinterpret_cast<const int4 *>(A.st_rec + (int64_t)s0 * 8);
        int4 *dst = &s_rec[b][0][0];
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
        const int32_t s0 = c * CK;
        const int32_t ns = min(CK, n_steps - s0);
        // the step record travels one step ahead in registers
        int4 ra = s_rec[b][0][0], rb = s_rec[b][0][1];
        unsigned long long mk = s_mask[b][0][wid];
        for (int i = 0; i < ns; i++) {
            const int inx = min(i + 1, CK - 1);
            const int4 na = s_rec[b][inx][0], nb = s_rec[b][inx][1];
            const unsigned long long nm = s_mask[b][inx][wid];
            if ((c | i) && (i & (WP - 1)) == 0) issue();
            if ((c | i) && (i & (WP - 1)) == WP / 2) land();
            const int32_t step = s0 + i;
            const int32_t flags = ra.x;
            const bool active = has_walk && ((mk >> lane) & 1ull);

            // ---- recombination entry into this vertex (uniform over the workgroup)
            int32_t E = NEG, Eh = -1, Esrc = -1;
            if (flags & PHI_DP_NEED_ENTRY) {
                const int n_in = (flags >> 8) & 0xFF;
                auto consider = [&](const int4 q, int32_t oj, int32_t src) {
                    const int32_t t1h = (q.z & 0x3FF) - 1, t1n = ((q.z >> 10) & 0x3FF) - 1, t2h = ((q.z >> 20) & 0x3FF) - 1;
                    const bool cont = t1n == oj;             // top1 continues along this edge: use top2
                    const int32_t val = cont ? q.y : q.x, hh = cont ? t2h : t1h;
                    if (hh < 0) return;
                    if (val > E || (val == E && (hh < Eh || (hh == Eh && src < Esrc)))) { E = val; Eh = hh; Esrc = src; }
                };
                const uint32_t b0 = (uint32_t)ra.z >> 8, b1 = (uint32_t)ra.w >> 8, b2 = (uint32_t)rb.x >> 8;
                if (n_in <= 3 && (b0 | b1 | b2) < RG) {
                    // usual case: all sources in the LDS ring, the three look-ups in flight together
                    const int4 q0 = s_top[(step - b0) & (RG - 1)];
                    const int4 q1 = s_top[(step - b1) & (RG - 1)];
                    const int4 q2 = s_top[(step - b2) & (RG - 1)];
                    consider(q0, ra.z & 0xFF, step - (int32_t)b0);
                    if (n_in > 1) consider(q1, ra.w & 0xFF, step - (int32_t)b1);
                    if (n_in > 2) consider(q2, rb.x & 0xFF, step - (int32_t)b2);
                } else {
                    for (int j = 0; j < n_in; j++) {
                        const int32_t p = j == 0 ? ra.z : j == 1 ? ra.w : j == 2 ? rb.x : A.in_packed[ra.y + j - 3];
                        const int32_t back = (int32_t)((uint32_t)p >> 8);
                        const int32_t src = step - back;
                        const int4 q = back < RG ? s_top[src & (RG - 1)]
                                                   : reinterpret_cast<const int4 *>(A.tops)[src];
                        consider(q, p & 0xFF, src);
                    }
                }
                if (Eh >= 0) E -= A.cost;
                if (h == 0) { A.ent_src[step] = Esrc; A.ent_h[step] = Eh; }
            }

            int32_t dmax = NEG;
            int32_t oidx = 255;
            if (active) {
                const int32_t t = (int32_t)(e - eb);
                const uint64_t word = nword;
                const int s2 = (t + 2) & 31;
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
                    const int32_t v_old = nold;               // = s_d[(t + 1) & 31][h]
                    if (v_old > L) { L = v_old; entL = t - 31; }
                    atomicAdd(&s_d[s2][h], v_old);            // next oldest becomes absolute
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
                        if (sp <= 30) { atomicAdd(&s_d[s2][h], 1); atomicAdd(&s_d[(t - sp + 1) & 31][h], -1); }
                    }
                } else {
                    while (gw & 0x7FFFFFFFFFFFFFull) {
                        const int sp = (int)(gw & 31);
                        gw >>= 5;
                        if (sp == 0) continue;
                        L += 1;
                        if (sp <= 30) { atomicAdd(&s_d[s2][h], 1); atomicAdd(&s_d[(t - sp + 1) & 31][h], -1); }
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
                        A.dmax[e] = best; A.bstart[e] = qb < 31 ? t - qb : entL;
                    } else {
                        A.dmax[e] = NEG; A.bstart[e] = 0;
                    }
                }
                e++;
                // the next entry's word and expiring slot: in flight until this lane is active again
                nword = WORDS_IN_LDS ? s_w[e & (WD - 1)][h] : A.word[min(e, ee - 1)];
                nold = s_d[s2][h];
            }

            // ---- best states leaving this vertex, by out-edge: top1 overall, top2 the best on
            //      another out-edge.  Walks that end here do not leave.
            if (flags & PHI_DP_NEED_TOPS) {
                const bool leaving = active && oidx != 255 && dmax > NEG / 2;
                int32_t t1v = NEG, t1h = -1, t1n = -1, t2v = NEG, t2h = -1;
                if (NW == 1) {
                    const int32_t m1 = wave_max_i32(leaving ? dmax : NEG);
                    if (m1 > NEG / 2) {
                        const int l1 = __ffsll((long long)__ballot(leaving && dmax == m1)) - 1;   // lowest walk id
                        t1v = m1; t1h = l1;
                        t1n = __builtin_amdgcn_readlane(oidx, l1);
                        const bool other = leaving && oidx != t1n;
                        const int32_t m2 = wave_max_i32(other ? dmax : NEG);
                        if (m2 > NEG / 2) { t2v = m2; t2h = __ffsll((long long)__ballot(other && dmax == m2)) - 1; }
                    }
                } else {
                    s_oidx[h] = oidx;
                    const int32_t m1 = wave_max_i32(leaving ? dmax : NEG);
                    unsigned long long k1 = 0;
                    if (m1 > NEG / 2) k1 = pack_vh(m1, wid * 64 + __ffsll((long long)__ballot(leaving && dmax == m1)) - 1);
                    if (lane == 0) s_red[wid] = k1;
                    __syncthreads();
                    k1 = s_red[0];
#pragma unroll
                    for (int x = 1; x < NW; x++) k1 = s_red[x] > k1 ? s_red[x] : k1;
                    if (k1) {
                        t1v = (int32_t)(uint32_t)(k1 >> 32) + NEG;
                        t1h = 0x7FFFFFFF - (int32_t)(uint32_t)k1;
                        t1n = s_oidx[t1h];
                    }
                    __syncthreads();
                    const bool other = leaving && oidx != t1n;
                    const int32_t m2 = wave_max_i32(other ? dmax : NEG);
                    unsigned long long k2 = 0;
                    if (m2 > NEG / 2) k2 = pack_vh(m2, wid * 64 + __ffsll((long long)__ballot(other && dmax == m2)) - 1);
                    if (lane == 0) s_red[wid] = k2;
                    __syncthreads();
                    k2 = s_red[0];
#pragma unroll
                    for (int x = 1; x < NW; x++) k2 = s_red[x] > k2 ? s_red[x] : k2;
                    if (k2) {
                        t2v = (int32_t)(uint32_t)(k2 >> 32) + NEG;
                        t2h = 0x7FFFFFFF - (int32_t)(uint32_t)k2;
                    }
                }
                if (h == 0) {
                    const int4 q = pack_tops(t1v, t1h, t1n, t2v, t2h);
                    s_top[step & (RG - 1)] = q;
                    reinterpret_cast<int4 *>(A.tops)[step] = q;
                }
                if (NW > 1) __syncthreads();             // ring entry visible to the other waves
            }
            ra = na; rb = nb; mk = nm;
        }
        __threadfence_block();
        __syncthreads();                                 // chunk c+1 staged; HBM tops visible
    }
}

int phi_dp_num_waves(int n_walks)
{
    const int nw = (n_walks + 63) / 64;
    const int forced = getenv("PHI_DP_WAVES") ? atoi(getenv("PHI_DP_WAVES")) : 0;      // (tests: the sixteen-wave instance on few walks)
    if (forced == 16) return 16;
    return nw <= 1 ? 1 : nw <= 2 ? 2 : nw <= 4 ? 4 : nw <= 8 ? 8 : 16;
}

void phi_launch_dp(hipStream_t st, const PhiDpArgs &A)
{
    const int nw = phi_dp_num_waves(A.n_walks);          // (the step stream's masks are laid out for this many waves)
    if (nw == 1) hipLaunchKernelGGL(phi_dp_kernel<1>, dim3(1), dim3(64), 0, st, A);
    else if (nw == 2) hipLaunchKernelGGL(phi_dp_kernel<2>, dim3(1), dim3(128), 0, st, A);
    else if (nw == 4) hipLaunchKernelGGL(phi_dp_kernel<4>, dim3(1), dim3(256), 0, st, A);
    else if (nw == 8) hipLaunchKernelGGL(phi_dp_kernel<8>, dim3(1), dim3(512), 0, st, A);
    else hipLaunchKernelGGL(phi_dp_kernel<16>, dim3(1), dim3(1024), 0, st, A);   // n_walks <= PHI_DP_MAX_WALKS
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_dp_kernel() {}
void phi_warm_dp(hipStream_t st) { hipLaunchKernelGGL(phi_warm_dp_kernel, dim3(1), dim3(64), 0, st); }
