// dp.hip -- max-plus dynamic program over the haplotype-expanded DAG (replaces model.optimize()).
//
// The reference builds a MIQP/MILP on the expanded graph (src/ILP_index.cpp:1160-1409) and hands
// it to Gurobi (:1418).  With the anchor weights fixed, the same objective -- maximise
//     sum of weights of anchors whose hap-j edges are all traversed  -  2*(R/2) * #w-node uses
// over unit s->e flows -- is a longest-path problem whose state is
// (vertex v, haplotype h, q = number of consecutive hap-h edges just taken, capped at 31):
// an anchor of walk h that spans s edges and ends at v pays when q >= s.
//
// Transitions (ILP_index.cpp:1203-1304):
//   stay      (u,h,q)   -> (next_h(u), h, min(q+1,31))                       cost 0
//   recombine (u,h',*)  -> (v,h,0) for an edge u->v with next_h'(u) != v     cost 2*(R/2)
//   start     (first(h), h, 0) = 0  (:1165-1195);   end at (last(h), h)       (:1388-1401)
// A walk that ends at u cannot leave u (its sink row is in - e = 0 and sum(e) = 1).
//
// Mapping: ONE workgroup, lane <-> walk; the chain of topologically ordered vertices is walked
// sequentially, each lane keeps its 32 run-length scores in registers.  The kernel is latency
// bound by construction (10^5..10^7 dependent steps of a few hundred cycles); it touches
// O(sum |walk|) bytes once.
#include <hip/hip_runtime.h>
#include "phi_kernels.h"

#define NEG (-(1 << 29))

__device__ __forceinline__ unsigned long long pack_vh(int32_t val, int32_t h)
{
    // larger value wins, then the smaller walk id
    return ((unsigned long long)(uint32_t)(val - NEG) << 32) | (uint32_t)(0x7FFFFFFF - h);
}

template <int NW>   // waves in the workgroup
__global__ void __launch_bounds__(NW * 64) phi_dp_kernel(PhiDpArgs A)
{
    __shared__ unsigned long long s_red[NW > 1 ? NW : 1];
    __shared__ int32_t s_next[NW * 64];

    const int h = threadIdx.x;
    const int lane = h & 63, wid = h >> 6;
    const bool has_walk = h < A.n_walks;
    const int64_t eb = has_walk ? A.walk_off[h] : 0;
    const int64_t ee = has_walk ? A.walk_off[h + 1] : 0;
    int64_t e = eb;                                             // next entry of this lane
    int32_t nextv = (e < ee) ? A.walk_vtx[e] : -1;
    int32_t nextv2 = (e + 1 < ee) ? A.walk_vtx[e + 1] : -1;

    int32_t vec[PHI_RCAP];
#pragma unroll
    for (int q = 0; q < PHI_RCAP; q++) vec[q] = NEG;
    int32_t lent = 0;

    for (int32_t step = 0; step < A.n_vtx; step++) {
        const int32_t v = A.topo[step];
        const bool active = has_walk && nextv == v;

        // recombination entry into v: best state leaving an in-neighbour along another route
        int32_t E = NEG, Eu = -1, Eh = -1;
        for (int64_t x = A.in_off[v]; x < A.in_off[v + 1]; x++) {
            const int32_t u = A.in_src[x];
            const bool cont = A.top1n[u] == v;                  // top1 continues along u->v: use top2
            const int32_t val = cont ? A.top2v[u] : A.top1v[u];
            const int32_t hh = cont ? A.top2h[u] : A.top1h[u];
            if (hh < 0) continue;
            if (val > E || (val == E && (hh < Eh || (hh == Eh && u < Eu)))) { E = val; Eu = u; Eh = hh; }
        }
        if (Eh >= 0) E -= A.cost;
        if (h == 0) { A.ent_v[v] = E; A.ent_u[v] = Eu; A.ent_h[v] = Eh; }

        int32_t dmax = NEG;
        if (active) {
            if (e == eb) {                                      // walk start: s_{first(h),h}
#pragma unroll
                for (int q = 1; q < PHI_RCAP; q++) vec[q] = NEG;
                vec[0] = 0;
            } else {
                if (vec[PHI_RCAP - 2] > vec[PHI_RCAP - 1]) {    // capped run restarts from run length 30
                    vec[PHI_RCAP - 1] = vec[PHI_RCAP - 2];
                    lent = (int32_t)(e - eb) - (PHI_RCAP - 1);
                }
#pragma unroll
                for (int q = PHI_RCAP - 2; q >= 1; q--) vec[q] = vec[q - 1];
                vec[0] = (Eh >= 0) ? E : NEG;
            }
            // anchors of this walk ending here
            const int64_t g0 = A.g_off[e], g1 = A.g_off[e + 1];
            for (int64_t g = g0; g < g1; g++) {
                const int32_t wgt = A.a_weight[g];
                const int32_t s = A.g_span[g];
#pragma unroll
                for (int q = 1; q < PHI_RCAP; q++) vec[q] += (q >= s) ? wgt : 0;
            }
            int32_t qb = 0;
#pragma unroll
            for (int q = 0; q < PHI_RCAP; q++)
                if (vec[q] >= dmax && vec[q] > NEG / 2) { dmax = vec[q]; qb = q; }
            A.dmax[e] = dmax;
            A.qbest[e] = (uint8_t)qb;
            A.lent[e] = lent;
            e++;
            nextv = nextv2;
            nextv2 = (e + 1 < ee) ? A.walk_vtx[e + 1] : -1;
        }

        // best states leaving v, grouped by the next vertex of their walk: top1 overall, top2 the
        // best whose next vertex differs from top1's.  Walks that end at v do not leave it.
        const bool leaving = active && nextv >= 0 && dmax > NEG / 2;
        s_next[h] = nextv;
        unsigned long long key = leaving ? pack_vh(dmax, h) : 0ull;
        unsigned long long k1 = key;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(k1, d, 64); k1 = o > k1 ? o : k1; }
        if (NW > 1) {
            if (lane == 0) s_red[wid] = k1;
            __syncthreads();
            k1 = s_red[0];
#pragma unroll
            for (int i = 1; i < NW; i++) k1 = s_red[i] > k1 ? s_red[i] : k1;
        }
        __syncthreads();                                        // s_next visible
        int32_t t1v = NEG, t1h = -1, t1n = -1, t2v = NEG, t2h = -1;
        if (k1) {
            t1v = (int32_t)(uint32_t)(k1 >> 32) + NEG;
            t1h = 0x7FFFFFFF - (int32_t)(uint32_t)k1;
            t1n = s_next[t1h];
        }
        unsigned long long k2 = (leaving && nextv != t1n) ? key : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(k2, d, 64); k2 = o > k2 ? o : k2; }
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) s_red[wid] = k2;
            __syncthreads();
            k2 = s_red[0];
#pragma unroll
            for (int i = 1; i < NW; i++) k2 = s_red[i] > k2 ? s_red[i] : k2;
        }
        if (k2) {
            t2v = (int32_t)(uint32_t)(k2 >> 32) + NEG;
            t2h = 0x7FFFFFFF - (int32_t)(uint32_t)k2;
        }
        if (h == 0) {
            A.top1v[v] = t1v; A.top1h[v] = t1h; A.top1n[v] = t1n;
            A.top2v[v] = t2v; A.top2h[v] = t2h;
        }
        __threadfence_block();
        __syncthreads();                                        // tops of v visible to later steps
    }
}

void phi_launch_dp(hipStream_t st, const PhiDpArgs &A)
{
    const int nw = (A.n_walks + 63) / 64;
    if (nw <= 1) hipLaunchKernelGGL(phi_dp_kernel<1>, dim3(1), dim3(64), 0, st, A);
    else if (nw <= 2) hipLaunchKernelGGL(phi_dp_kernel<2>, dim3(1), dim3(128), 0, st, A);
    else if (nw <= 4) hipLaunchKernelGGL(phi_dp_kernel<4>, dim3(1), dim3(256), 0, st, A);
    else if (nw <= 8) hipLaunchKernelGGL(phi_dp_kernel<8>, dim3(1), dim3(512), 0, st, A);
    else hipLaunchKernelGGL(phi_dp_kernel<16>, dim3(1), dim3(1024), 0, st, A);
}
