// solve_dev.hip -- the bookkeeping of phi_solve on the anchors where they are: in HBM.
//
// The exact solve (phi_solve.hip) replaces ILP_index.cpp:776-1418 (model construction + model.optimize()) by DP runs
// under an optimality certificate.  Between the runs it needs, of the model's anchors: the DP's per-anchor arrays,
// the minimisers that repeat along a walk, the weights of a relaxation (anchors of a set S of minimisers count 0),
// and -- for the path a run returns -- how often every minimiser is covered.  At chromosome scale the model has
// 5 * 10^8 anchors (6 GB of triples): shipping them to the host and looping there cost 5 of the solve's 5.7 s.
// These kernels do the same work on the device copy; the host sees counters and short lists.  The branch-and-bound
// proper (clusters of one minimiser's anchors, rarely reached) still runs on a host copy, fetched when it is needed.
//
// Anchor = triple (minimiser id, first entry e0, last entry e1), sorted by e1 (= walk order, then position).
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

__device__ __forceinline__ int32_t walk_of_entry(const int64_t *__restrict__ walk_off, int32_t n_walks, int64_t e)
{
    int lo = 0, hi = n_walks;                        // last h with walk_off[h] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (walk_off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}
// one atomicAdd per wave for a counter most lanes bump
__device__ __forceinline__ void wave_count(bool pred, unsigned long long *ctr)
{
    const unsigned long long b = __ballot(pred);
    if (b && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)b) - 1)) atomicAdd(ctr, (unsigned long long)__popcll(b));
}

// ---- the DP's per-anchor arrays, the anchors per walk, and what would make the device path unusable
// out[0] = anchors inside one vertex (e1 <= e0: not dp anchors), out[1] |= 1 unsorted, |= 2 span >= PHI_RCAP
__global__ void __launch_bounds__(256) phi_anchor_prep_kernel(const uint32_t *__restrict__ tri, int64_t n, const int64_t *__restrict__ walk_off,
                                                              int32_t n_walks, phi_ent_t *__restrict__ a_e1, uint8_t *__restrict__ a_span,
                                                              unsigned long long *__restrict__ walk_cnt, unsigned long long *__restrict__ out)
{
    GRID_STRIDE(i0, (n + 63) & ~(int64_t)63) {
        const bool in = i0 < n;
        const int64_t i = in ? i0 : n - 1;
        const phi_ent_t e0 = tri[i * 3 + 1], e1 = tri[i * 3 + 2];
        wave_count(in && e1 <= e0, &out[0]);
        uint32_t bad = 0;
        if (in && i > 0 && e1 < tri[(i - 1) * 3 + 2]) bad |= 1u;
        if (in && e1 > e0 && e1 - e0 >= PHI_RCAP) bad |= 2u;
        if (bad) atomicOr(&out[1], (unsigned long long)bad);
        if (in) { a_e1[i] = e1; a_span[i] = (uint8_t)(e1 > e0 ? e1 - e0 : 0); }
        // anchors come in walk order: a wave is almost always inside one walk
        const int32_t h = walk_of_entry(walk_off, n_walks, e0);
        const int32_t h0 = __builtin_amdgcn_readfirstlane(h);
        if (__ballot(in && h != h0) == 0) wave_count(in, &walk_cnt[h0]);
        else if (in) atomicAdd(&walk_cnt[h], 1ull);
    }
}
void phi_launch_anchor_prep(hipStream_t st, const uint32_t *tri, int64_t n, const int64_t *walk_off, int32_t n_walks, phi_ent_t *a_e1, uint8_t *a_span,
                            unsigned long long *walk_cnt, unsigned long long *out)
{
    if (n > 0) hipLaunchKernelGGL(phi_anchor_prep_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, tri, n, walk_off, n_walks, a_e1, a_span, walk_cnt, out);
}

// ---- the score range (phi_solve.hip): sum over vertices of the most anchors that end there on one walk
__global__ void __launch_bounds__(256) phi_vertex_most_kernel(const int64_t *__restrict__ g_off, int64_t n_entries, const int32_t *__restrict__ walk_vtx,
                                                              int32_t *__restrict__ vmax)
{
    GRID_STRIDE(e, n_entries) {
        const int64_t cnt = g_off[e + 1] - g_off[e];
        if (cnt > 0) atomicMax(&vmax[walk_vtx[e]], (int32_t)(cnt > INT32_MAX ? INT32_MAX : cnt));
    }
}
__global__ void __launch_bounds__(256) phi_sum_i32_kernel(const int32_t *__restrict__ v, int64_t n, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0;
    GRID_STRIDE(i, n) s += (unsigned long long)v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
void phi_launch_vertex_most(hipStream_t st, const int64_t *g_off, int64_t n_entries, const int32_t *walk_vtx, int32_t *vmax)
{
    if (n_entries > 0) hipLaunchKernelGGL(phi_vertex_most_kernel, dim3(grid_for(n_entries, 256)), dim3(256), 0, st, g_off, n_entries, walk_vtx, vmax);
}
void phi_launch_sum_i32(hipStream_t st, const int32_t *v, int64_t n, unsigned long long *out)
{
    if (n > 0) hipLaunchKernelGGL(phi_sum_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, v, n, out);
}

// ---- minimisers with two anchors on one walk.  Without the minimiser -> anchors map (which costs 0.16 s to build at chromosome scale and is only needed by
// the branch and bound proper): the anchors come in walk order, so ONE launch per walk over that walk's anchors --
// last[id] holds the last walk in which minimiser id was met; meeting it again in the same launch is a repeat.
__global__ void __launch_bounds__(256) phi_repeat_walk_kernel(const uint32_t *__restrict__ tri, int64_t lo, int64_t hi, int32_t walk,
                                                              int32_t *__restrict__ last, uint8_t *__restrict__ flags)
{
    for (int64_t a = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < hi; a += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t id = tri[a * 3];
        if (atomicExch(&last[id], walk) == walk) flags[id] = 1;
    }
}
void phi_launch_repeat_walk(hipStream_t st, const uint32_t *tri, int64_t lo, int64_t hi, int32_t walk, int32_t *last, uint8_t *flags)
{
    if (hi > lo) hipLaunchKernelGGL(phi_repeat_walk_kernel, dim3(grid_for(hi - lo, 256)), dim3(256), 0, st, tri, lo, hi, walk, last, flags);
}
// weights of a relaxation from a flag per minimiser (in_s[id] != 0: its anchors count 0), over all anchors
__global__ void __launch_bounds__(256) phi_weights_kernel(const uint32_t *__restrict__ tri, int64_t n, const uint8_t *__restrict__ in_s, uint8_t *__restrict__ wgt)
{
    GRID_STRIDE(a, n) wgt[a] = in_s[tri[a * 3]] ? 0 : 1;
}
void phi_launch_weights(hipStream_t st, const uint32_t *tri, int64_t n, const uint8_t *in_s, uint8_t *wgt)
{
    if (n > 0) hipLaunchKernelGGL(phi_weights_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, tri, n, in_s, wgt);
}

// ---- what a path covers.  segs = (first entry, last entry) of the path's stretches along single walks; the anchors
// traversed by a stretch are those with  es <= e0, e1 <= ee, i.e. indices [g_off[es], g_off[ee + 1]) with e0 >= es.
// COUNT: cov_all / cov_w[minimiser] += 1 (all / weight-1 anchors); ctr[0] = weight-1 anchors covered, ctr[1] =
//        minimisers covered, ctr[2] = length of `twice` = minimisers whose weight-1 anchors are covered at least twice.
// CLEAR: the same anchors again, their counters back to zero.
template <bool CLEAR>
__global__ void __launch_bounds__(256) phi_path_cover_kernel(const phi_ent_t *__restrict__ segs, int32_t n_seg, const int64_t *__restrict__ g_off,
                                                             const uint32_t *__restrict__ tri, const uint8_t *__restrict__ wgt,
                                                             int32_t *__restrict__ cov_all, int32_t *__restrict__ cov_w,
                                                             unsigned long long *__restrict__ ctr, uint32_t *__restrict__ twice, int64_t twice_cap)
{
    for (int32_t q = blockIdx.y; q < n_seg; q += gridDim.y) {
        const phi_ent_t es = segs[2 * q], ee = segs[2 * q + 1];
        const int64_t lo = g_off[es], hi = g_off[(int64_t)ee + 1];
        const int64_t span = (hi - lo + 63) & ~(int64_t)63;
        for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < span; i0 += (int64_t)gridDim.x * blockDim.x) {
            const int64_t a = lo + i0;
            const bool in = a < hi && tri[a * 3 + 1] >= es;
            const uint32_t s = in ? tri[a * 3] : 0;
            if (CLEAR) {
                if (in) { cov_all[s] = 0; cov_w[s] = 0; }
                continue;
            }
            const bool first = in && atomicAdd(&cov_all[s], 1) == 0;
            const bool w = in && wgt[a] != 0;
            const int32_t before = w ? atomicAdd(&cov_w[s], 1) : -1;
            wave_count(w, &ctr[0]);
            wave_count(first, &ctr[1]);
            if (before == 1) {
                const unsigned long long at = atomicAdd(&ctr[2], 1ull);
                if ((int64_t)at < twice_cap) twice[at] = s;
            }
        }
    }
}
void phi_launch_path_cover(hipStream_t st, bool clear, const phi_ent_t *segs, int32_t n_seg, const int64_t *g_off, const uint32_t *tri, const uint8_t *wgt,
                           int32_t *cov_all, int32_t *cov_w, unsigned long long *ctr, uint32_t *twice, int64_t twice_cap)
{
    if (n_seg <= 0) return;
    // (a path of few stretches covers millions of anchors, one of thousands of stretches a few each)
    const dim3 grid(n_seg <= 8 ? 512 : n_seg <= 256 ? 32 : 2, (unsigned)(n_seg < 16384 ? n_seg : 16384));
    if (clear) hipLaunchKernelGGL(phi_path_cover_kernel<true>, grid, dim3(256), 0, st, segs, n_seg, g_off, tri, wgt, cov_all, cov_w, ctr, twice, twice_cap);
    else hipLaunchKernelGGL(phi_path_cover_kernel<false>, grid, dim3(256), 0, st, segs, n_seg, g_off, tri, wgt, cov_all, cov_w, ctr, twice, twice_cap);
}

// the minimisers of `slots` the path does not cover at all -> out[0 .. ctr[3])
__global__ void __launch_bounds__(256) phi_uncovered_slots_kernel(const uint32_t *__restrict__ slots, int64_t n, const int32_t *__restrict__ cov_all,
                                                                  unsigned long long *__restrict__ ctr, uint32_t *__restrict__ out)
{
    GRID_STRIDE(i, n) {
        const uint32_t s = slots[i];
        if (cov_all[s] == 0) out[atomicAdd(&ctr[3], 1ull)] = s;      // (out holds n entries)
    }
}
void phi_launch_uncovered_slots(hipStream_t st, const uint32_t *slots, int64_t n, const int32_t *cov_all, unsigned long long *ctr, uint32_t *out)
{
    if (n > 0) hipLaunchKernelGGL(phi_uncovered_slots_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, slots, n, cov_all, ctr, out);
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_solve_dev_kernel() {}
void phi_warm_solve_dev(hipStream_t st) { hipLaunchKernelGGL(phi_warm_solve_dev_kernel, dim3(1), dim3(64), 0, st); }
