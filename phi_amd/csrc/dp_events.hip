// dp_events.hip -- the max-plus DP of dp.hip, event driven: only the vertices where something can
// happen are walked serially, everything else is prefix sums made by wide kernels.
//
// dp.hip spends one serial step on EVERY vertex because each lane shifts a 31-slot run-length
// window per walk entry.  Here a run of walk h that began at entry s (entered by a recombination
// of value E_s, or the walk start with E = 0) is kept as that pair only.  With
//     End(x) = number of weight-1 anchors of the walk with last entry  <= x
//     SB(s)  = number of weight-1 anchors of the walk with first entry <  s
// its score at entry x is  E_s + #{anchors inside [s, x]}, which is
//     E_s - SB(s) + End(x)                      once x - s >= 31  (no anchor spans more than 31 edges)
//     E_s + G_x[x - s]                          before that, G_x[a] = #{anchors inside [x - a, x]}
// End, SB and the 31 small counts G_x are functions of the anchor weights only: three wide kernels
// make them per DP run (phi_dp_event_fill_kernel packs them into a 48-byte record per EVENT).  An
// event is a walk entry on a vertex where the serial part has work: a recombination can enter
// (ENTRY), leave (TOPS), or the walk starts / ends.  On the synthetic MHC graph that is one vertex
// in four; chain vertices cost nothing.
//
// Per lane (= walk) the live runs form a deque in LDS.  A new run (s, E) is dropped when an older
// run already has E' >= E: the older run contains every anchor the younger one does, and ties go
// to the older run as in dp.hip.  When it is kept it evicts, from the back, the runs whose key
// E' - SB(s') is smaller than its own: they lead by fewer anchors than they trail in value and
// can never catch up.  Runs older than 30 entries are folded into one scalar (the best key).  A
// query (TOPS / walk end) is then max(best key + End, max over the few young runs of E_s + G[age]).
//
// Same transitions, same tie-breaks and the same outputs as dp.hip (which stays as the kernel for
// more than 128 walks): tests/test_gpu_parity.py runs both on the same inputs.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "phi_kernels.h"

#define NEG (-(1 << 28))
#define NEGK (-(1 << 30))
#define CHK PHI_DP_CHUNK
#define RING PHI_DP_RING
enum { DP_SEQ = 0, DP_ROW = 1, DP_PATH = 2 };   // modes of the event kernels (see phi_dp_events_pc_kernel)

// ------------------------------------------------------------------ static: which entries are events
__global__ void __launch_bounds__(256) phi_event_flags_kernel(const int32_t *__restrict__ walk_vtx, int64_t n_entries,
                                                              const int32_t *__restrict__ cvtx, uint8_t *__restrict__ flags)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (int64_t)gridDim.x * blockDim.x)
        flags[e] = cvtx[walk_vtx[e]] >= 0;
}

// ev_off[h] = first event of walk h (ev_e ascending, walk_off[n_walks] = n_entries)
__global__ void phi_event_off_kernel(const phi_ent_t *__restrict__ ev_e, int64_t n_ev, const int64_t *__restrict__ walk_off,
                                     int32_t n_walks, int64_t *__restrict__ ev_off)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_walks) return;
    const int64_t key = walk_off[h];
    int64_t lo = 0, hi = n_ev;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)ev_e[mid] < key) lo = mid + 1; else hi = mid;
    }
    ev_off[h] = lo;
}

void phi_launch_event_flags(hipStream_t st, const int32_t *walk_vtx, int64_t n_entries, const int32_t *cvtx, uint8_t *flags)
{
    if (n_entries <= 0) return;
    int64_t nb = (n_entries + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(phi_event_flags_kernel, dim3((unsigned)nb), dim3(256), 0, st, walk_vtx, n_entries, cvtx, flags);
}

void phi_launch_event_off(hipStream_t st, const phi_ent_t *ev_e, int64_t n_ev, const int64_t *walk_off, int32_t n_walks,
                          int64_t *ev_off)
{
    hipLaunchKernelGGL(phi_event_off_kernel, dim3((unsigned)((n_walks + 1 + 63) / 64)), dim3(64), 0, st, ev_e, n_ev,
                       walk_off, n_walks, ev_off);
}

// ------------------------------------------------------------------ per run: counts, prefix sums, event records
// exclusive prefix sums of int32 counts, three phases (1024 items per workgroup)
template <class T>
__global__ void __launch_bounds__(256) phi_scan_blocksum_kernel(const T *__restrict__ cnt, int64_t n,
                                                                int32_t *__restrict__ blk)
{
    __shared__ int s_w[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j < n) c += cnt[base + j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// (off may be cnt itself: a thread reads its four counts before it writes its four sums, and nobody else's)
template <class T>
__global__ void __launch_bounds__(256) phi_scan_apply_kernel(const T *cnt, int64_t n,
                                                             const int64_t *__restrict__ blk_off, int32_t *off)
{
    __shared__ int s_w[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    int v[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j] = (base + j < n) ? cnt[base + j] : 0; c += v[j]; }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wid; i++) woff += s_w[i];
    int run = (int)blk_off[blockIdx.x] + woff + inc - c;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) off[base + j] = run;
        run += v[j];
    }
    if (base <= n && n < base + 4) off[n] = run - 0;   // total (the items past n are zero)
}

// 64-bit variant (flat base offsets of the walk entries)
__global__ void __launch_bounds__(256) phi_scan_blocksum64_kernel(const int32_t *__restrict__ cnt, int64_t n,
                                                                  int64_t *__restrict__ blk)
{
    __shared__ long long s_w[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    long long c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j < n) c += cnt[base + j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// single-workgroup exclusive scan of 64-bit block sums (a few thousand items)
__global__ void __launch_bounds__(1024) phi_scan_sums64_kernel(const int64_t *__restrict__ v, int64_t n, int64_t *__restrict__ off)
{
    __shared__ long long s_part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = min(n, tid * per), hi = min(n, lo + per);
    long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += v[i];
    s_part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; i++) { const long long t = s_part[i]; s_part[i] = run; run += t; }
        off[n] = run;
    }
    __syncthreads();
    long long run = s_part[tid];
    for (int64_t i = lo; i < hi; i++) { off[i] = run; run += v[i]; }
}

__global__ void __launch_bounds__(256) phi_scan_apply64_kernel(const int32_t *__restrict__ cnt, int64_t n,
                                                               const int64_t *__restrict__ blk_off, int64_t *__restrict__ off)
{
    __shared__ long long s_w[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    long long v[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j] = (base + j < n) ? cnt[base + j] : 0; c += v[j]; }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long long inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    long long woff = 0;
    for (int i = 0; i < wid; i++) woff += s_w[i];
    long long run = blk_off[blockIdx.x] + woff + inc - c;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (base + j < n) off[base + j] = run;
        run += v[j];
    }
    if (base <= n && n < base + 4) off[n] = run;
}

int64_t phi_scan_i32_num_blocks(int64_t n) { return (n + 1 + 1023) / 1024; }

void phi_launch_scan_sums_i64(hipStream_t st, const int64_t *v, int64_t n, int64_t *off)
{
    hipLaunchKernelGGL(phi_scan_sums64_kernel, dim3(1), dim3(1024), 0, st, v, n, off);
}

void phi_launch_scan_i64(hipStream_t st, const int32_t *cnt, int64_t n, int64_t *off, int64_t *blk, int64_t *blk_off)
{
    const int64_t nb = phi_scan_i32_num_blocks(n);
    hipLaunchKernelGGL(phi_scan_blocksum64_kernel, dim3((unsigned)nb), dim3(256), 0, st, cnt, n, blk);
    phi_launch_scan_sums_i64(st, blk, nb, blk_off);
    hipLaunchKernelGGL(phi_scan_apply64_kernel, dim3((unsigned)nb), dim3(256), 0, st, cnt, n, blk_off, off);
}

// off[0..n] = exclusive prefix sums of cnt[0..n); blk / blk_off: scratch of phi_scan_i32_num_blocks(n) (+1) items
void phi_launch_scan_i32(hipStream_t st, const int32_t *cnt, int64_t n, int32_t *off, int32_t *blk, int64_t *blk_off)
{
    const int64_t nb = phi_scan_i32_num_blocks(n);
    hipLaunchKernelGGL(phi_scan_blocksum_kernel<int32_t>, dim3((unsigned)nb), dim3(256), 0, st, cnt, n, blk);
    phi_launch_scan_counts(st, blk, nb, blk_off);
    hipLaunchKernelGGL(phi_scan_apply_kernel<int32_t>, dim3((unsigned)nb), dim3(256), 0, st, cnt, n, blk_off, off);
}
// the same over bytes (the anchor weights of a DP run)
void phi_launch_scan_u8(hipStream_t st, const uint8_t *cnt, int64_t n, int32_t *off, int32_t *blk, int64_t *blk_off)
{
    const int64_t nb = phi_scan_i32_num_blocks(n);
    hipLaunchKernelGGL(phi_scan_blocksum_kernel<uint8_t>, dim3((unsigned)nb), dim3(256), 0, st, cnt, n, blk);
    phi_launch_scan_counts(st, blk, nb, blk_off);
    hipLaunchKernelGGL(phi_scan_apply_kernel<uint8_t>, dim3((unsigned)nb), dim3(256), 0, st, cnt, n, blk_off, off);
}

// One 48-byte record per event:
//   int4  A = { compact step | overflow << 31, entry (phi_ent_t), End (inclusive), SB }
//         End / SB = weight-1 anchors of the walk that end at or before / begin before the entry.  With the anchors sorted by
//         last entry and W = prefix sums of their weights (wpre), "end before entry x" is W[g_off[x]]; an anchor that
//         begins before e and ends at or after it ends within the next 30 entries (span <= 31): a short range to look at.
//         (Until round 3 both came from per-ENTRY counters -- two arrays of 4 bytes per walk entry zeroed, filled by an
//          atomic per anchor and prefix-summed in every DP run: 35 of a run's 204 ms at chromosome scale.)
//   32 B  G : byte a (1..30) = #{weight-1 anchors of the walk inside [entry - a, entry]}, byte 0 = 0,
//             byte 31 = out-edge index of the entry (255: the walk ends here)
// overflow: more than 255 anchors end inside the window (the DP then counts from the CSR).
// The anchors an event looks at -- those ending on the 30 entries up to it (G, End) and on the 31 from it (SB) -- are ONE range
// of the list sorted by last entry, and the ranges of a block's 256 consecutive events overlap almost entirely: the block loads
// the union once, coalesced, into LDS (last entry; span | weight << 7) and every event counts from there.  (Round 3: every
// event walked its ranges in global memory, three dependent loads per anchor and 24 anchors per event at config 5: 30 ms per
// DP run.)  A block whose union is longer than the stage (events far apart along chain vertices) reads global memory as before.
// (3 072 entries = 15 KB: ten workgroups per CU; with 6 144 -- five -- the kernel took 19-21 instead of 15.6 ms at config 5, with 1 024
//  too many blocks fell back to global memory: 18.8)
#define EVF_STAGE 3072
__global__ void __launch_bounds__(256) phi_dp_event_fill_kernel(const phi_ent_t *__restrict__ ev_e, int64_t n_ev,
                                                                const int32_t *__restrict__ walk_vtx,
                                                                const int32_t *__restrict__ cvtx,
                                                                const int64_t *__restrict__ walk_off, int32_t n_walks,
                                                                const uint8_t *__restrict__ e_out,
                                                                const int64_t *__restrict__ g_off,
                                                                const uint8_t *__restrict__ g_span,
                                                                const uint8_t *__restrict__ a_weight,
                                                                const phi_ent_t *__restrict__ a_e1,
                                                                const int32_t *__restrict__ wpre, int64_t n_entries,
                                                                uint4 *__restrict__ ev)
{
    __shared__ uint32_t s_e1[EVF_STAGE];
    __shared__ uint8_t s_sw[EVF_STAGE];
    __shared__ int64_t s_rng[2];
    const int64_t n_blocks = (n_ev + 255) / 256;
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t i = blk * 256 + threadIdx.x;
        const bool in = i < n_ev;
        const int64_t e = ev_e[in ? i : n_ev - 1];
        // walk of e: last h with walk_off[h] <= e.  Events are in entry order: nearly every wave lies inside one walk --
        // found once, for the wave's first event, on the scalar unit; the lanes search for themselves only when the wave
        // straddles walks (eight dependent loads per event otherwise, for 2.4 * 10^8 events per DP run at chromosome scale)
        int lo = 0;
        {
            const int64_t e1st = ((int64_t)__builtin_amdgcn_readfirstlane((int)(e >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)e);
            int hi = n_walks;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (walk_off[mid] <= e1st) lo = mid; else hi = mid;
            }
            if (__ballot(e >= walk_off[lo + 1]) != 0ull) {    // (wave-uniform)
                hi = n_walks;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (walk_off[mid] <= e) lo = mid; else hi = mid;
                }
            }
        }
        const int64_t eb = walk_off[lo];
        const int64_t x0 = e - 29 > eb ? e - 29 : eb;
        const int64_t xe = e + 31 < n_entries ? e + 31 : n_entries;
        // the dp anchors are sorted by their last entry: those ending on the entries x0 .. e are ONE range of the list (two
        // loads of the CSR instead of two per entry of the window), those ending on e .. e + 30 another
        const int64_t g_lo = g_off[x0], g_e = g_off[e], g_hi = g_off[e + 1], g_fe = g_off[xe];
        if (threadIdx.x == 0) s_rng[0] = g_lo;                 // (both bounds grow with the event)
        if (i == (n_ev < blk * 256 + 256 ? n_ev : blk * 256 + 256) - 1) s_rng[1] = g_fe;
        __syncthreads();
        const int64_t gb = s_rng[0], ge = s_rng[1];
        const bool staged = ge - gb <= EVF_STAGE;
        if (staged)
            for (int64_t j = threadIdx.x; j < ge - gb; j += 256) {
                s_e1[j] = a_e1[gb + j];
                s_sw[j] = (uint8_t)(g_span[gb + j] | (a_weight[gb + j] ? 0x80 : 0));
            }
        __syncthreads();
        if (in) {
            unsigned long long w[4] = {0, 0, 0, 0};
            int total = 0;
            int32_t sb_extra = 0;
            auto count = [&](int a) {                          // age a run needs to contain this anchor
                if (a > 30) return;
                total++;
                const unsigned long long one = 1ull << (8 * (a & 7));
                if ((a >> 3) == 0) w[0] += one; else if ((a >> 3) == 1) w[1] += one; else if ((a >> 3) == 2) w[2] += one; else w[3] += one;
            };
            const uint32_t e32 = (uint32_t)e;
            if (staged) {
                for (int64_t g = g_lo - gb; g < g_hi - gb; g++) {
                    const uint32_t sw = s_sw[g];
                    if (sw & 0x80) count((int)(e32 - (s_e1[g] - (sw & 0x7F))));
                }
                for (int64_t g = g_e - gb; g < g_fe - gb; g++) {
                    const uint32_t sw = s_sw[g];
                    sb_extra += (sw & 0x80) && (int64_t)s_e1[g] - (int64_t)(sw & 0x7F) < e;
                }
            } else {
                for (int64_t g = g_lo; g < g_hi; g++)
                    if (a_weight[g]) count((int)(e - ((int64_t)a_e1[g] - g_span[g])));
                for (int64_t g = g_e; g < g_fe; g++) sb_extra += a_weight[g] && (int64_t)a_e1[g] - g_span[g] < e;
            }
            const bool ovf = total > 255;
            // byte-wise inclusive prefix sums across the four words
            unsigned long long carry = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                w[q] = w[q] * 0x0101010101010101ull + carry * 0x0101010101010101ull;
                carry = w[q] >> 56;
            }
            w[3] = (w[3] & 0x00FFFFFFFFFFFFFFull) | ((unsigned long long)e_out[e] << 56);
            // End and SB count from the walk's first entry (anchors of earlier walks have both begun and ended before it):
            // keys E - SB stay within the score range whatever the number of walks
            const int32_t base = wpre[g_off[eb]];
            uint4 A;
            A.x = (uint32_t)cvtx[walk_vtx[e]] | (ovf ? 0x80000000u : 0u);
            A.y = (uint32_t)e;
            A.z = (uint32_t)(wpre[g_hi] - base);
            A.w = (uint32_t)(wpre[g_e] - base + sb_extra);
            ev[i * 3 + 0] = A;
            ev[i * 3 + 1] = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
            ev[i * 3 + 2] = make_uint4((uint32_t)w[2], (uint32_t)(w[2] >> 32), (uint32_t)w[3], (uint32_t)(w[3] >> 32));
        }
        __syncthreads();                                       // (the stage is refilled for the next block of events)
    }
}

void phi_launch_dp_event_fill(hipStream_t st, const PhiDpEventArgs &A, const uint8_t *e_out, const int32_t *walk_vtx,
                              const int32_t *cvtx, const phi_ent_t *a_e1, const int32_t *wpre, int64_t n_entries)
{
    if (A.n_ev <= 0) return;
    int64_t nb = (A.n_ev + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(phi_dp_event_fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, A.ev_e, A.n_ev, walk_vtx, cvtx,
                       A.walk_off, A.n_walks, e_out, A.g_off, A.g_span, A.a_weight, a_e1, wpre, n_entries,
                       reinterpret_cast<uint4 *>(A.ev));
}

// ------------------------------------------------------------------ the DP
__device__ __forceinline__ unsigned long long ev_pack_vh(int32_t val, int32_t h)
{
    return ((unsigned long long)(uint32_t)(val - NEG) << 32) | (uint32_t)(0x7FFFFFFF - h);
}

__device__ __forceinline__ int32_t ev_wave_max_i32(int32_t v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));     // row_ror:1
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));     // row_ror:2
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));     // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));     // row_ror:8
    const int32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// two independent maxima in one pass: the DPP chains interleave, a lone wave issues them back to back
__device__ __forceinline__ void ev_wave_max2_i32(int32_t a, int32_t b, int32_t &ma, int32_t &mb)
{
    a = max(a, __builtin_amdgcn_update_dpp(a, a, 0x121, 0xF, 0xF, false)); b = max(b, __builtin_amdgcn_update_dpp(b, b, 0x121, 0xF, 0xF, false));
    a = max(a, __builtin_amdgcn_update_dpp(a, a, 0x122, 0xF, 0xF, false)); b = max(b, __builtin_amdgcn_update_dpp(b, b, 0x122, 0xF, 0xF, false));
    a = max(a, __builtin_amdgcn_update_dpp(a, a, 0x124, 0xF, 0xF, false)); b = max(b, __builtin_amdgcn_update_dpp(b, b, 0x124, 0xF, 0xF, false));
    a = max(a, __builtin_amdgcn_update_dpp(a, a, 0x128, 0xF, 0xF, false)); b = max(b, __builtin_amdgcn_update_dpp(b, b, 0x128, 0xF, 0xF, false));
    ma = max(max(__builtin_amdgcn_readlane(a, 0), __builtin_amdgcn_readlane(a, 16)), max(__builtin_amdgcn_readlane(a, 32), __builtin_amdgcn_readlane(a, 48)));
    mb = max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)), max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
}

// (score, walk) of a leaving state as one key: larger score first, then the lower walk id.  Scores lie
// above NEG = -2^28, so the key is below 2^38: a positive (denormal) double bit pattern, whose IEEE order
// is the integer order -- v_max_f64 reduces it, and the winner's walk comes out of the key (no ballot,
// no find-first, no branches).  0 = no leaving state.
__device__ __forceinline__ unsigned long long ev_key(int32_t val, int lane)
{
    return ((unsigned long long)(uint32_t)(val - NEG) << 8) | (unsigned long long)(255 - lane);
}
__device__ __forceinline__ unsigned long long ev_max_u53(unsigned long long a, unsigned long long b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (unsigned long long)__double_as_longlong(r);
}
#define EV_DPP64(x, ctrl, rm, bc)                                                                                             \
    (((unsigned long long)(uint32_t)__builtin_amdgcn_update_dpp((int)((x) >> 32), (int)((x) >> 32), ctrl, rm, 0xF, bc) << 32) | \
     (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(x), (int)(uint32_t)(x), ctrl, rm, 0xF, bc))
// wave maxima of two keys, the two DPP chains interleaved; the result is uniform (read from lane 63)
__device__ __forceinline__ void ev_wave_max2_key(unsigned long long a, unsigned long long b, unsigned long long &ma, unsigned long long &mb)
{
    a = ev_max_u53(a, EV_DPP64(a, 0x121, 0xF, false)); b = ev_max_u53(b, EV_DPP64(b, 0x121, 0xF, false));   // row_ror:1
    a = ev_max_u53(a, EV_DPP64(a, 0x122, 0xF, false)); b = ev_max_u53(b, EV_DPP64(b, 0x122, 0xF, false));   // row_ror:2
    a = ev_max_u53(a, EV_DPP64(a, 0x124, 0xF, false)); b = ev_max_u53(b, EV_DPP64(b, 0x124, 0xF, false));   // row_ror:4
    a = ev_max_u53(a, EV_DPP64(a, 0x128, 0xF, false)); b = ev_max_u53(b, EV_DPP64(b, 0x128, 0xF, false));   // row_ror:8
    a = ev_max_u53(a, EV_DPP64(a, 0x142, 0xA, false)); b = ev_max_u53(b, EV_DPP64(b, 0x142, 0xA, false));   // row_bcast:15 into rows 1, 3
    a = ev_max_u53(a, EV_DPP64(a, 0x143, 0xC, false)); b = ev_max_u53(b, EV_DPP64(b, 0x143, 0xC, false));   // row_bcast:31 into rows 2, 3
    ma = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(a >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a, 63);
    mb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63);
}

__device__ __forceinline__ int4 ev_pack_tops(int32_t t1v, int32_t t1h, int32_t t1n, int32_t t2v, int32_t t2h)
{
    return make_int4(t1v, t2v, (t1h + 1) | ((t1n + 1) << 10) | ((t2h + 1) << 20), 0);
}

// exact count for an event whose window overflowed the byte counters: weight-1 anchors inside [es, e]
__device__ int32_t ev_count_inside(const PhiDpEventArgs &A, int64_t es, int64_t e)
{
    int32_t n = 0;
    for (int64_t x = es; x <= e; x++)
        for (int64_t g = A.g_off[x]; g < A.g_off[x + 1]; g++)
            if (A.a_weight[g] && x - A.g_span[g] >= es) n++;
    return n;
}

// MODE: DP_SEQ (one workgroup, all steps) or DP_PATH (one workgroup per block of steps, from the block's entry
// vector A.blk_S; see the modes of phi_dp_events_pc_kernel below)
template <int NW, int MODE>   // waves in the workgroup (1, 2 or 4)
__global__ void __launch_bounds__(NW * 64) phi_dp_events_kernel(PhiDpEventArgs A)
{
    constexpr int NT = NW * 64;
    constexpr int D = NW == 1 ? 16 : NW == 2 ? 8 : 4;   // per-lane ring of event records
    constexpr int P = D / 2;                         // refill period in steps
    // four waves: 160 KB of LDS hold queues of 16 runs (31 can be alive in theory, 2-3 are in practice: a
    // deeper one raises PHI_KERR_DP_QUEUE and the caller reruns with dp.hip) and the tops of 512 steps
    constexpr int QD = NW <= 2 ? 32 : 16;
    constexpr int RNG = NW <= 2 ? RING : 512;
    __shared__ int4 s_rec[2][CHK][2];
    __shared__ int4 s_top[RNG];                     // packed tops of recent steps
    __shared__ uint4 s_ev[D][3][NT];
    __shared__ int32_t s_qs[QD][NT], s_qE[QD][NT], s_qK[QD][NT];   // live young runs: start, value, key
    const int32_t q_limit = A.q_limit > 0 && A.q_limit < QD ? A.q_limit : (QD == 32 ? 32 : QD);
    __shared__ unsigned long long s_red[NW > 1 ? NW : 1];
    __shared__ int32_t s_oidx[NT];

    const int h = threadIdx.x;
    const int lane = h & 63, wid = h >> 6;
    const bool has_walk = h < A.n_walks;
    const int64_t eb = has_walk ? A.walk_off[h] : 0;
    const int64_t ee = has_walk ? A.walk_off[h + 1] : 0;
    const int32_t sb = MODE == DP_PATH ? (int32_t)blockIdx.x : 0;
    const int32_t kbeg = MODE == DP_PATH ? A.blk_lo[sb] : 0, kend = MODE == DP_PATH ? A.blk_lo[sb + 1] : A.n_k;
    // events of this lane: [vb, ve) (DP_PATH: from the first one inside the block)
    const int64_t vb = !has_walk ? 0 : MODE == DP_PATH ? (int64_t)A.blk_ev[(int64_t)sb * A.lane_stride + h] : A.ev_off[h];
    const int64_t ve = has_walk ? A.ev_off[h + 1] : 0;
    int64_t vi = vb;                                 // next event
    int64_t wl = vb;                                 // first event not yet in the ring
    const uint4 *evg = reinterpret_cast<const uint4 *>(A.ev);

    // ring refill in two halves (see dp.hip): loads issued at step 0 mod P land P/2 steps later
    uint4 wtmp[P][3];
    int64_t whi = vb;
    auto issue = [&]() {
        whi = min(ve, vi + D);
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int64_t x = (ve > vb) ? min(wl + j, ve - 1) : 0;
            wtmp[j][0] = evg[x * 3 + 0]; wtmp[j][1] = evg[x * 3 + 1]; wtmp[j][2] = evg[x * 3 + 2];
        }
    };
    auto land = [&]() {
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int64_t x = wl + j;
            if (x < whi) {
                const int sl = (int)(x & (D - 1));
                s_ev[sl][0][h] = wtmp[j][0]; s_ev[sl][1][h] = wtmp[j][1]; s_ev[sl][2][h] = wtmp[j][2];
            }
        }
        wl = max(wl, min(whi, wl + P));
    };
    if (A.n_ev > 0) { issue(); land(); issue(); land(); }
    // the next event of this lane, in registers
    uint4 cA = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    if (vi < ve) cA = s_ev[vi & (D - 1)][0][h];

    int32_t qh = 0, qt = 0;                          // deque [qh, qt) of young runs
    int32_t M = NEGK, sL = 0, Emax = NEG;            // best key of the old runs, its start; best value ever entered
    if (MODE == DP_PATH) { sL = -1; if (has_walk) M = A.blk_S[(int64_t)sb * A.lane_stride + h]; }   // (-1: began before this block)

    const int32_t n_steps = kend - kbeg;
    const int n_chunks = (n_steps + CHK - 1) / CHK;
    auto stage = [&](int c) {
        const int b = c & 1;
        const int32_t s0 = kbeg + c * CHK;
        const int32_t ns = min(CHK, kend - s0);
        const int4 *src = reinterpret_cast<const int4 *>(A.k_rec + (int64_t)s0 * 8);
        int4 *dst = &s_rec[b][0][0];
        for (int i = h; i < ns * 2; i += NT) dst[i] = src[i];
    };
    stage(0);
    __syncthreads();

    for (int c = 0; c < n_chunks; c++) {
        const int b = c & 1;
        if (c + 1 < n_chunks) stage(c + 1);
        const int32_t s0 = kbeg + c * CHK;
        const int32_t ns = min(CHK, kend - s0);
        int4 ra = s_rec[b][0][0], rb = s_rec[b][0][1];
        for (int i = 0; i < ns; i++) {
            const int inx = min(i + 1, CHK - 1);
            const int4 na = s_rec[b][inx][0], nb = s_rec[b][inx][1];
            if ((c | i) && (i & (P - 1)) == 0) issue();
            if ((c | i) && (i & (P - 1)) == P / 2) {
                land();
                if (vi < ve) cA = s_ev[vi & (D - 1)][0][h];     // (a peek made before its record landed saw a stale slot)
            }
            const int32_t k = s0 + i;
            const int32_t flags = ra.x;
            const bool active = (int32_t)(cA.x & 0x7FFFFFFFu) == k;      // (no event: 0xFFFFFFFF, a step no block reaches)

            // ---- recombination entry into this vertex (uniform over the workgroup)
            int32_t E = NEG, Eh = -1, Esrc = -1;
            if (flags & PHI_DP_NEED_ENTRY) {
                const int n_in = (flags >> 8) & 0xFF;
                auto consider = [&](const int4 q, int32_t oj, int32_t src) {
                    const int32_t t1h = (q.z & 0x3FF) - 1, t1n = ((q.z >> 10) & 0x3FF) - 1, t2h = ((q.z >> 20) & 0x3FF) - 1;
                    const bool cont = t1n == oj;
                    const int32_t val = cont ? q.y : q.x, hh = cont ? t2h : t1h;
                    if (hh < 0) return;
                    if (val > E || (val == E && (hh < Eh || (hh == Eh && src < Esrc)))) { E = val; Eh = hh; Esrc = src; }
                };
                const uint32_t b0 = (uint32_t)ra.z >> 8, b1 = (uint32_t)ra.w >> 8, b2 = (uint32_t)rb.x >> 8;
                if (n_in <= 3 && (b0 | b1 | b2) < RNG) {
                    const int4 q0 = s_top[(k - b0) & (RNG - 1)];
                    const int4 q1 = s_top[(k - b1) & (RNG - 1)];
                    const int4 q2 = s_top[(k - b2) & (RNG - 1)];
                    consider(q0, ra.z & 0xFF, k - (int32_t)b0);
                    if (n_in > 1) consider(q1, ra.w & 0xFF, k - (int32_t)b1);
                    if (n_in > 2) consider(q2, rb.x & 0xFF, k - (int32_t)b2);
                } else {
                    for (int j = 0; j < n_in; j++) {
                        const int32_t p = j == 0 ? ra.z : j == 1 ? ra.w : j == 2 ? rb.x : A.k_in_packed[ra.y + j - 3];
                        const int32_t back = (int32_t)((uint32_t)p >> 8);
                        const int32_t src = k - back;
                        const int4 q = back < RNG ? s_top[src & (RNG - 1)] : reinterpret_cast<const int4 *>(A.tops)[src];
                        consider(q, p & 0xFF, src);
                    }
                }
                if (Eh >= 0) E -= A.cost;
                if (h == 0) { A.ent_src[k] = Esrc; A.ent_h[k] = Eh; }
            }

            int32_t dmax = NEG;
            int32_t oidx = 255;
            if (active) {
                const int64_t e = (int64_t)cA.y;
                const bool ovf = (cA.x >> 31) != 0;
                const int32_t t = (int32_t)(e - eb);
                const int32_t End = (int32_t)cA.z, SB = (int32_t)cA.w;
                const int sl = (int)(vi & (D - 1));
                const uint8_t *gb = reinterpret_cast<const uint8_t *>(&s_ev[sl][1][h]);
                oidx = reinterpret_cast<const uint8_t *>(&s_ev[sl][2][h])[15];
                // runs older than 30 entries: one scalar
                while (qh != qt && t - s_qs[qh & (QD - 1)][h] >= 31) {
                    const int32_t key = s_qK[qh & (QD - 1)][h];
                    if (key > M) { M = key; sL = s_qs[qh & (QD - 1)][h]; }
                    qh++;
                }
                // a run begins here: the walk start, or a recombination entry worth keeping
                int32_t newE = NEG;
                if (t == 0) { newE = 0; qh = qt = 0; M = NEGK; Emax = NEG; }
                else if ((flags & PHI_DP_NEED_ENTRY) && Eh >= 0) newE = E;
                if (newE > Emax) {
                    Emax = newE;
                    const int32_t key = newE - SB;
                    while (qh != qt && s_qK[(qt - 1) & (QD - 1)][h] < key) qt--;
                    if (qt - qh >= q_limit) { atomicOr(A.err, PHI_KERR_DP_QUEUE); qt--; }   // result void: the caller falls back
                    s_qs[qt & (QD - 1)][h] = t; s_qE[qt & (QD - 1)][h] = newE; s_qK[qt & (QD - 1)][h] = key;
                    qt++;
                }
                if ((flags & PHI_DP_NEED_TOPS) || e == ee - 1) {
                    // oldest first, strict improvement: ties keep the older run
                    int32_t best = M > NEGK / 2 ? M + End : NEG, bs = sL;
                    for (int32_t j = qh; j != qt; j++) {
                        const int32_t s = s_qs[j & (QD - 1)][h];
                        const int a = t - s;
                        int32_t inside;
                        if (!ovf) inside = a < 16 ? gb[a] : reinterpret_cast<const uint8_t *>(&s_ev[sl][2][h])[a - 16];
                        else inside = ev_count_inside(A, eb + s, e);
                        const int32_t val = s_qE[j & (QD - 1)][h] + inside;
                        if (val > best) { best = val; bs = s; }
                    }
                    if (best > NEG / 2) { dmax = best; A.dmax[e] = best; A.bstart[e] = bs; }
                    else { A.dmax[e] = NEG; A.bstart[e] = 0; }
                }
                vi++;
                cA = make_uint4(0xFFFFFFFFu, 0, 0, 0);
                if (vi < ve) cA = s_ev[vi & (D - 1)][0][h];
            }

            // ---- best states leaving this vertex (as in dp.hip)
            if (flags & PHI_DP_NEED_TOPS) {
                const bool leaving = active && oidx != 255 && dmax > NEG / 2;
                int32_t t1v = NEG, t1h = -1, t1n = -1, t2v = NEG, t2h = -1;
                if (NW == 1) {
                    const int32_t m1 = ev_wave_max_i32(leaving ? dmax : NEG);
                    if (m1 > NEG / 2) {
                        const int l1 = __ffsll((long long)__ballot(leaving && dmax == m1)) - 1;
                        t1v = m1; t1h = l1;
                        t1n = __builtin_amdgcn_readlane(oidx, l1);
                        const bool other = leaving && oidx != t1n;
                        const int32_t m2 = ev_wave_max_i32(other ? dmax : NEG);
                        if (m2 > NEG / 2) { t2v = m2; t2h = __ffsll((long long)__ballot(other && dmax == m2)) - 1; }
                    }
                } else {
                    s_oidx[h] = oidx;
                    const int32_t m1 = ev_wave_max_i32(leaving ? dmax : NEG);
                    unsigned long long k1 = 0;
                    if (m1 > NEG / 2) k1 = ev_pack_vh(m1, wid * 64 + __ffsll((long long)__ballot(leaving && dmax == m1)) - 1);
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
                    const int32_t m2 = ev_wave_max_i32(other ? dmax : NEG);
                    unsigned long long k2 = 0;
                    if (m2 > NEG / 2) k2 = ev_pack_vh(m2, wid * 64 + __ffsll((long long)__ballot(other && dmax == m2)) - 1);
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
                    const int4 q = ev_pack_tops(t1v, t1h, t1n, t2v, t2h);
                    s_top[k & (RNG - 1)] = q;
                    reinterpret_cast<int4 *>(A.tops)[k] = q;
                }
                if (NW > 1) __syncthreads();
            }
            ra = na; rb = nb;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (MODE == DP_PATH) {
        // what crosses the cut at the block's end: the best key of this walk's live runs (oldest first, ties keep the
        // older run) and where that run began
        int32_t best = M > NEGK / 2 ? M : NEGK, bs = sL;
        for (int32_t j = qh; j != qt; j++) {
            const int32_t key = s_qK[j & (QD - 1)][h];
            if (key > best) { best = key; bs = s_qs[j & (QD - 1)][h]; }
        }
        A.blk_keys_out[(int64_t)sb * A.lane_stride + h] = (has_walk && vb < ve) ? best : NEGK;
        A.blk_carry[(int64_t)sb * A.lane_stride + h] = bs;
    }
}

// ------------------------------------------------------------------ up to 64 walks: consumer + producers
// The serial wave (the consumer) touches LDS only.  NPW producer waves of the same workgroup keep it
// fed and carry its results out, one barrier every P steps:
//   * refill the per-lane event ring from HBM and stage the step records, a period ahead;
//   * write the (best score, run start) pairs of the events consumed in the previous period, the
//     packed tops and the entry choices of its steps back to HBM.
//
// BLOCKS OF STEPS IN PARALLEL.  The chain over the compact steps can be cut between steps k-1 and k when no
// recombination edge crosses the cut and, on every walk, no anchor spans it (phi_solve.hip picks such "clean"
// cuts): a run that crosses a clean cut scores its anchors left and right of it independently, so all that
// crosses is ONE number per walk -- the best key  E_s - SB(s)  of its live runs -- and the steps of a block are a
// max-plus linear map of that vector.  Three modes of the same kernel:
//   DP_SEQ   one workgroup walks all steps (the only mode for graphs without usable cuts)
//   DP_ROW   one workgroup per (block, entry walk j): the block's steps from the unit vector "key 0 on walk j"
//            (walk starts inside the block only in the extra row j = n_walks) -> row j of the block's transfer
//            matrix: the keys on all walks at the block's end, and the best value of a path ending inside
//   DP_PATH  one workgroup per block, from the TRUE entry vector (the host chains the matrices): writes what
//            DP_SEQ writes (best scores / run starts per event, tops, entries) plus, per walk, where the run
//            that carries the best key out of the block began -- the backtrack follows runs across blocks

// CK: steps of the step stream staged at a time (two chunks in LDS).  Blocks of at most 64 steps -- the class-lane blocks of a
// chromosome-scale graph are ~18 -- take CK = 32 and a ring of 64 tops: 26 instead of 35 KB of LDS, six workgroups per CU instead of four.
template <int NPW, int MODE, int P, int RINGT, int QD, int CK = CHK>
__global__ void __launch_bounds__(64 * (1 + NPW)) phi_dp_events_pc_kernel(PhiDpEventArgs A)
{
    constexpr int D = 2 * P;
    constexpr int PER_CHUNK = CK / P;
    static_assert(P % NPW == 0 && CK % P == 0, "ring geometry");
    __shared__ int4 s_rec[2 * CK][2];              // step records: ring of two chunks, index = (step - k0) mod 2*CHK
    __shared__ int4 s_top[RINGT];                   // packed tops of recent steps
    __shared__ int2 s_ent[2 * P];                   // (source step, walk) of the entries of the last two periods
    __shared__ uint4 s_ev[D][3][64];
    __shared__ int4 s_res[D][64];                   // (best score, run start, entry) of consumed events
    __shared__ int32_t s_qs[QD][64], s_qE[QD][64], s_qK[QD][64];
    __shared__ int32_t s_vi[2][64];              // the consumer's position at each barrier, double-buffered: written for period p + 1 while the producers still read period p's

    const int wave = threadIdx.x >> 6, h = threadIdx.x & 63;
    // the block of steps of this workgroup, and (DP_ROW) the lane the unit vector sits on.  DP_ROW may run on CLASS
    // LANES (A.lane_walk, more than 64 walks): lane l stands for all walks that do the same inside the block and is
    // played by the first of them; rows 0..63 are the class lanes, row 64 the walk starts.
    const bool cls = MODE == DP_ROW && A.lane_walk != nullptr;
    const unsigned n_rows = cls ? 65u : (unsigned)(A.n_walks + 1);
    int32_t sb = 0, row_j = -1;
    if (MODE == DP_ROW) { sb = (int32_t)(blockIdx.x / n_rows); row_j = (int32_t)(blockIdx.x % n_rows); }
    if (MODE == DP_PATH) sb = (int32_t)blockIdx.x;
    const bool start_row = MODE == DP_ROW && row_j == (int32_t)n_rows - 1;
    int32_t w = h;                                   // the walk behind this lane
    bool has_walk = h < A.n_walks;
    if (cls) { w = A.lane_walk[(int64_t)sb * 64 + h]; has_walk = w >= 0; if (!has_walk) w = 0; }
    const int32_t k0 = MODE == DP_SEQ ? 0 : A.blk_lo[sb];
    const int32_t k1 = MODE == DP_SEQ ? A.n_k : A.blk_lo[sb + 1];
    const int32_t n_steps = k1 - k0;
    const phi_ent_t eb = has_walk ? (phi_ent_t)A.walk_off[w] : 0;
    const phi_ent_t e_last = has_walk ? (phi_ent_t)(A.walk_off[w + 1] - 1) : 0xFFFFFFFFu;     // (0xFFFFFFFF: no entry, PHI_MAX_ENTRIES)
    const int32_t v0 = has_walk ? (int32_t)A.ev_off[w] : 0;     // events of this lane's walk: [v0, ve)
    const int32_t ve = has_walk ? (int32_t)A.ev_off[w + 1] : 0;
    const int32_t vb = MODE == DP_SEQ ? v0 : (has_walk ? A.blk_ev[(int64_t)sb * A.lane_stride + w] : 0);   // first event inside the block
    const int n_per = (n_steps + P - 1) / P;
    if (MODE == DP_ROW && !start_row) {
        const int32_t wj = cls ? A.lane_walk[(int64_t)sb * 64 + row_j] : row_j;
        if (wj < 0) return;                                      // no such class in this block (uniform)
        // a walk that has not begun before the block, or has no event left, carries nothing in: the row is empty
        const int32_t jb = A.blk_ev[(int64_t)sb * A.lane_stride + wj];
        if (jb <= (int32_t)A.ev_off[wj] || jb >= (int32_t)A.ev_off[wj + 1]) {            // uniform over the workgroup
            if (wave == 0) {
                A.row_out[(int64_t)blockIdx.x * 64 + h] = NEGK;
                if (h == 0) { A.rowend_out[blockIdx.x] = NEG; if (A.rownew_out) A.rownew_out[blockIdx.x] = NEGK; if (A.rowdiag_out) A.rowdiag_out[(int64_t)sb * 64 + row_j] = NEGK; }
            }
            return;
        }
    }

    if (wave > 0) {
        // ================================================================ producers
        const int pw = wave - 1;
        const uint4 *evg = reinterpret_cast<const uint4 *>(A.ev);
        auto stage = [&](int c) {
            const int32_t s0 = k0 + c * CK;
            const int32_t ns = min(CK, k1 - s0);
            const int4 *src = reinterpret_cast<const int4 *>(A.k_rec + (int64_t)s0 * 8);
            int4 *dst = &s_rec[(c & 1) * CK][0];
            for (int i = pw * 64 + h; i < ns * 2; i += 64 * NPW) dst[i] = src[i];
        };
        stage(0);
#pragma unroll
        for (int n = 0; n < D / NPW; n++) {
            const int32_t x = vb + pw + n * NPW;
            if (x < ve) {
                const uint4 a = evg[(int64_t)x * 3 + 0], g0 = evg[(int64_t)x * 3 + 1], g1 = evg[(int64_t)x * 3 + 2];
                s_ev[x & (D - 1)][0][h] = a; s_ev[x & (D - 1)][1][h] = g0; s_ev[x & (D - 1)][2][h] = g1;
            }
        }
        int32_t wl = min(ve, vb + D);                // first event not in the ring (the same in every producer wave)
        int32_t vprev = vb;                          // consumer position one barrier ago
        __syncthreads();                             // B_0
        for (int p = 0;; p++) {
            const int32_t vi = has_walk ? s_vi[p & 1][h] : 0;
            if (MODE != DP_ROW) {
                // 1. results of the events consumed during the previous period
#pragma unroll
                for (int n = 0; n < P / NPW; n++) {
                    const int32_t x = vprev + pw + n * NPW;
                    if (x < vi) {
                        const int4 r = s_res[x & (D - 1)][h];
                        A.dmax[(phi_ent_t)r.z] = r.x; A.bstart[(phi_ent_t)r.z] = r.y;
                    }
                }
                // 2. tops and entry choices of the previous period's steps (values of steps without
                //    TOPS / ENTRY are never read back)
                if (pw == 0 && p > 0 && h < P) {
                    const int32_t r = (p - 1) * P + h;
                    if (r < n_steps) {
                        const int32_t k = k0 + r;
                        reinterpret_cast<int4 *>(A.tops)[k] = s_top[k & (RINGT - 1)];
                        const int2 en = s_ent[r & (2 * P - 1)];
                        A.ent_src[k] = en.x; A.ent_h[k] = en.y;
                    }
                }
            }
            if (p == n_per) break;                   // after the last barrier: only the write-backs
            // 3. refill: events [wl, min(ve, vi + D)), at most P of them; all loads first
            const int32_t hi = min(ve, vi + D);
            uint4 t0[P / NPW], t1[P / NPW], t2[P / NPW];
#pragma unroll
            for (int n = 0; n < P / NPW; n++) {
                const int64_t xc = (ve > v0) ? min(wl + pw + n * NPW, ve - 1) : 0;
                t0[n] = evg[xc * 3 + 0]; t1[n] = evg[xc * 3 + 1]; t2[n] = evg[xc * 3 + 2];
            }
#pragma unroll
            for (int n = 0; n < P / NPW; n++) {
                const int32_t x = wl + pw + n * NPW;
                if (x < hi) { s_ev[x & (D - 1)][0][h] = t0[n]; s_ev[x & (D - 1)][1][h] = t1[n]; s_ev[x & (D - 1)][2][h] = t2[n]; }
            }
            wl = max(wl, min(hi, wl + P));
            // 4. the step records of the next chunk
            if (p % PER_CHUNK == 0 && (p / PER_CHUNK + 1) * CK < n_steps) stage(p / PER_CHUNK + 1);
            vprev = vi;
            __syncthreads();                         // B_{p+1}
        }
        return;
    }

    // ==================================================================== the consumer
    int32_t vi = vb;                                 // next event
    s_vi[0][h] = vi;
    __syncthreads();                                 // B_0
    uint4 cA = make_uint4(0xFFFFFFFFu, 0, 0, 0);     // the next event of this lane, in registers
    if (vi < ve) cA = s_ev[vi & (D - 1)][0][h];
    // live young runs: deque in LDS; the head (start, value) and the tail key also in registers
    int32_t qh = 0, qn = 0, hs = 0, hE = 0, tk = 0;
    int32_t M = NEGK, sL = MODE == DP_SEQ ? 0 : -1, Emax = NEG;   // best key of the old runs and its start (-1: began before this block); best value ever entered
    if (MODE == DP_ROW && h == row_j) M = 0;
    if (MODE == DP_PATH && has_walk) M = A.blk_S[(int64_t)sb * A.lane_stride + h];
    const bool starts_on = MODE != DP_ROW || start_row;   // walks may begin inside the block
    int32_t endbest = NEG;
    int32_t Kn = NEGK;                               // DP_ROW: best key of a run that began inside the block on this lane                           // DP_ROW: best value at a walk's last entry inside the block
    int4 ra = s_rec[0][0], rb = s_rec[0][1];
    int4 lastq = make_int4(NEG, NEG, 0, 0);          // tops of the latest TOPS step, forwarded in registers
    int32_t lastk = -1;

    // recombination entry into the vertex of one step record (uniform): value, walk and source step
    auto eval_entry = [&](const int4 xa, const int4 xb, int32_t k, int32_t &E, int32_t &Eh, int32_t &Esrc) {
        E = NEG; Eh = -1; Esrc = -1;
        const int32_t flags = xa.x;
        const int n_in = (flags >> 8) & 0xFF;
        auto consider = [&](const int4 q, int32_t oj, int32_t src) {
            const int32_t t1h = (q.z & 0x3FF) - 1, t1n = ((q.z >> 10) & 0x3FF) - 1, t2h = ((q.z >> 20) & 0x3FF) - 1;
            const bool cont = t1n == oj;
            const int32_t val = cont ? q.y : q.x, hh = cont ? t2h : t1h;
            if (hh < 0) return;
            if (val > E || (val == E && (hh < Eh || (hh == Eh && src < Esrc)))) { E = val; Eh = hh; Esrc = src; }
        };
        const uint32_t b0 = (uint32_t)xa.z >> 8, b1 = (uint32_t)xa.w >> 8, b2 = (uint32_t)xb.x >> 8;
        if (n_in == 1 && b0 < RINGT) {
            const int32_t src = k - (int32_t)b0;
            const int4 q = src == lastk ? lastq : s_top[src & (RINGT - 1)];
            const int32_t t1h = (q.z & 0x3FF) - 1, t1n = ((q.z >> 10) & 0x3FF) - 1, t2h = ((q.z >> 20) & 0x3FF) - 1;
            const bool cont = t1n == (xa.z & 0xFF);
            const int32_t hh = cont ? t2h : t1h;
            if (hh >= 0) { E = cont ? q.y : q.x; Eh = hh; Esrc = k - (int32_t)b0; }
        } else if (n_in <= 3 && (b0 | b1 | b2) < RINGT) {
            const int4 q0 = s_top[(k - b0) & (RINGT - 1)];
            const int4 q1 = s_top[(k - b1) & (RINGT - 1)];
            const int4 q2 = s_top[(k - b2) & (RINGT - 1)];
            consider(q0, xa.z & 0xFF, k - (int32_t)b0);
            if (n_in > 1) consider(q1, xa.w & 0xFF, k - (int32_t)b1);
            if (n_in > 2) consider(q2, xb.x & 0xFF, k - (int32_t)b2);
        } else {
            for (int j = 0; j < n_in; j++) {
                const int32_t pk = j == 0 ? xa.z : j == 1 ? xa.w : j == 2 ? xb.x : A.k_in_packed[xa.y + j - 3];
                const int32_t back = (int32_t)((uint32_t)pk >> 8);
                const int32_t src = k - back;
                // tops older than the ring come from HBM: the producers wrote them at least RINGT - 2P steps ago
                // (a block of steps is never longer than its ring, and no edge enters it from before)
                const int4 q = (back < RINGT || MODE == DP_ROW) ? s_top[src & (RINGT - 1)] : reinterpret_cast<const int4 *>(A.tops)[src];
                consider(q, pk & 0xFF, src);
            }
        }
        if (Eh >= 0) E -= A.cost;
        if (h == 0) s_ent[(k - k0) & (2 * P - 1)] = make_int2(Esrc, Eh);
    };

    for (int p = 0; p < n_per; p++) {
        const int32_t r_end = min(n_steps, (p + 1) * P);
        // The ring is guaranteed to hold this lane's next P events when a period begins, not one more: a lane that
        // consumed an event on every step of the last period peeked at a record the producers were still loading
        // (a stale slot: the lane then never saw its next event).  Peek again now that the barrier has passed.
        if (vi < ve) cA = s_ev[vi & (D - 1)][0][h];
        for (int32_t r = p * P; r < r_end;) {
            const int32_t k = k0 + r;
            // records of the next two steps (past the last step: stale records, never used)
            const int4 na = s_rec[(r + 1) & (2 * CK - 1)][0], nb = s_rec[(r + 1) & (2 * CK - 1)][1];
            const int4 n2a = s_rec[(r + 2) & (2 * CK - 1)][0], n2b = s_rec[(r + 2) & (2 * CK - 1)][1];
            // two alleles of one site (PHI_DP_PAIR: neither leaves states, no walk visits both): one iteration
            const bool pair = (ra.x & PHI_DP_PAIR) && r + 1 < r_end;
            const int32_t stepk = (int32_t)(cA.x & 0x7FFFFFFFu);    // (no event: 0xFFFFFFFF, a step no block reaches)
            const bool second = pair && stepk == k + 1;
            const bool active = stepk == k || second;
            const int32_t flags = second ? na.x : ra.x;

            // ---- recombination entry into this vertex (uniform per step)
            int32_t E = NEG, Eh = -1, Esrc = -1;
            if (ra.x & PHI_DP_NEED_ENTRY) eval_entry(ra, rb, k, E, Eh, Esrc);
            if (pair && (na.x & PHI_DP_NEED_ENTRY)) {
                int32_t E1, Eh1, Esrc1;
                eval_entry(na, nb, k + 1, E1, Eh1, Esrc1);
                if (second) { E = E1; Eh = Eh1; Esrc = Esrc1; }
            } else if (second) { E = NEG; Eh = -1; }

            int32_t dmax = NEG;
            int32_t oidx = 255;
            if (active) {
                const int sl = (int)(vi & (D - 1));
                const phi_ent_t e = cA.y;
                const bool ovf = (cA.x >> 31) != 0;
                const int32_t t = (int32_t)(e - eb);
                const int32_t End = (int32_t)cA.z, SB = (int32_t)cA.w;
                const uint8_t *gb = reinterpret_cast<const uint8_t *>(&s_ev[sl][1][h]);   // byte a of G at gb[(a & 15) + (a >> 4) * 1024]
                oidx = gb[15 + 1024];
                // runs older than 30 entries: one scalar
                while (qn > 0 && t - hs >= 31) {
                    const int32_t key = qn == 1 ? tk : s_qK[qh & (QD - 1)][h];
                    if (key > M) { M = key; sL = hs; }
                    qh++; qn--;
                    if (qn > 0) { hs = s_qs[qh & (QD - 1)][h]; hE = s_qE[qh & (QD - 1)][h]; }
                }
                // a run begins here: the walk start, or a recombination entry worth keeping
                int32_t newE = NEG;
                if (t == 0) { newE = starts_on ? 0 : NEG; qn = 0; M = NEGK; Emax = NEG; }
                else if ((flags & PHI_DP_NEED_ENTRY) && Eh >= 0) newE = E;
                if (newE > Emax) {
                    Emax = newE;
                    const int32_t key = newE - SB;
                    while (qn > 0 && tk < key) {
                        qn--;
                        if (qn > 0) tk = s_qK[(qh + qn - 1) & (QD - 1)][h];
                    }
                    if (QD < 32 && qn == QD) { atomicOr(A.err, PHI_KERR_DP_QUEUE); qn--; }   // result void: the caller takes DP_SEQ
                    const int slq = (qh + qn) & (QD - 1);
                    s_qs[slq][h] = t; s_qE[slq][h] = newE; s_qK[slq][h] = key;
                    if (qn == 0) { hs = t; hE = newE; }
                    tk = key;
                    qn++;
                    if (MODE == DP_ROW && key > Kn) Kn = key;
                }
                int32_t bs = 0;
                if ((flags & PHI_DP_NEED_TOPS) || e == e_last) {
                    // oldest first, strict improvement: ties keep the older run
                    int32_t best = M > NEGK / 2 ? M + End : NEG;
                    bs = sL;
                    for (int32_t j = 0; j < qn; j++) {
                        const int32_t s = j == 0 ? hs : s_qs[(qh + j) & (QD - 1)][h];
                        const int32_t Es = j == 0 ? hE : s_qE[(qh + j) & (QD - 1)][h];
                        const int a = t - s;
                        const int32_t inside = !ovf ? (int32_t)gb[(a & 15) + (a >> 4) * 1024] : ev_count_inside(A, (int64_t)eb + s, e);
                        const int32_t val = Es + inside;
                        if (val > best) { best = val; bs = s; }
                    }
                    if (best > NEG / 2) dmax = best; else bs = 0;
                    if (MODE == DP_ROW && e == e_last) endbest = max(endbest, dmax);
                }
                s_res[sl][h] = make_int4(dmax, bs, (int32_t)e, 0);
                vi++;
                cA = make_uint4(0xFFFFFFFFu, 0, 0, 0);
                if (vi < ve) cA = s_ev[vi & (D - 1)][0][h];
            }

            // ---- best states leaving this vertex (as in dp.hip); never for a pair
            if (!pair && (ra.x & PHI_DP_NEED_TOPS)) {
                const bool leaving = active && oidx != 255 && dmax > NEG / 2;
                int32_t t1v = NEG, t1h = -1, t1n = -1, t2v = NEG, t2h = -1;
                if (__ballot(leaving && oidx > 1) == 0ull) {
                    // at most two out-edges in use (a bubble): best walk per edge, two independent reductions
                    const bool on0 = leaving && oidx == 0, on1 = leaving && oidx == 1;
                    unsigned long long K0, K1;
                    ev_wave_max2_key(on0 ? ev_key(dmax, h) : 0ull, on1 ? ev_key(dmax, h) : 0ull, K0, K1);
                    // top1: larger value, then the lower walk id (= the larger key); top2: the best on the other edge
                    const bool first0 = K0 >= K1;
                    const unsigned long long Kf = first0 ? K0 : K1, Ks = first0 ? K1 : K0;
                    t1v = Kf ? (int32_t)(uint32_t)(Kf >> 8) + NEG : NEG;
                    t1h = Kf ? 255 - (int32_t)(Kf & 255) : -1;
                    t1n = Kf ? (first0 ? 0 : 1) : -1;
                    t2v = Ks ? (int32_t)(uint32_t)(Ks >> 8) + NEG : NEG;
                    t2h = Ks ? 255 - (int32_t)(Ks & 255) : -1;
                } else {
                    const int32_t m1 = ev_wave_max_i32(leaving ? dmax : NEG);
                    if (m1 > NEG / 2) {
                        const int l1 = __ffsll((long long)__ballot(leaving && dmax == m1)) - 1;
                        t1v = m1; t1h = l1;
                        t1n = __builtin_amdgcn_readlane(oidx, l1);
                        const bool other = leaving && oidx != t1n;
                        const int32_t m2 = ev_wave_max_i32(other ? dmax : NEG);
                        if (m2 > NEG / 2) { t2v = m2; t2h = __ffsll((long long)__ballot(other && dmax == m2)) - 1; }
                    }
                }
                lastq = ev_pack_tops(t1v, t1h, t1n, t2v, t2h);
                lastk = k;
                if (h == 0) s_top[k & (RINGT - 1)] = lastq;
            }
            if (pair) { ra = n2a; rb = n2b; r += 2; }
            else { ra = na; rb = nb; r += 1; }
        }
        s_vi[(p + 1) & 1][h] = vi;
        __syncthreads();                             // B_{p+1}
    }
    if (MODE != DP_SEQ) {
        // what crosses the cut at the block's end: the best key of this walk's live runs (oldest first, ties keep the
        // older run) and where that run began
        int32_t best = M > NEGK / 2 ? M : NEGK, bs = sL;
        for (int32_t j = 0; j < qn; j++) {
            const int32_t key = s_qK[(qh + j) & (QD - 1)][h];
            if (key > best) { best = key; bs = s_qs[(qh + j) & (QD - 1)][h]; }
        }
        if (MODE == DP_ROW) {
            // (class lanes: the row's own column goes to an array of its own and leaves NEGK behind -- the chain then takes the
            //  other classes' maximum over the whole row without looking for the diagonal)
            const bool on_diag = A.rowdiag_out && h == row_j;
            if (on_diag) A.rowdiag_out[(int64_t)sb * 64 + h] = has_walk ? best : NEGK;
            A.row_out[(int64_t)blockIdx.x * 64 + h] = has_walk && !on_diag ? best : NEGK;
            const int32_t eb_all = ev_wave_max_i32(endbest);
            if (h == 0) A.rowend_out[blockIdx.x] = eb_all;
            // what a walk of the unit lane's class other than the one that carries the key gets: the new runs of that lane
            if (A.rownew_out && h == row_j) A.rownew_out[blockIdx.x] = Kn;
            if (A.rownew_out && start_row && h == 0) A.rownew_out[blockIdx.x] = NEGK;
        } else {
            // (a walk with no event left when the block began carries nothing that is ever read: reported as "no run",
            //  as the row pass does)
            A.blk_keys_out[(int64_t)sb * A.lane_stride + h] = (has_walk && vb < ve) ? best : NEGK;
            A.blk_carry[(int64_t)sb * A.lane_stride + h] = bs;
        }
    }
}

// ------------------------------------------------------------------ where the chain may be cut (per solve)
// diff[e0 + 1] += 1, diff[e1 + 1] -= 1 for every dp anchor (first / last entry e0 < e1): the prefix sum at entry e
// is the number of anchors with e0 < e <= e1, i.e. that a cut right before e would split
__global__ void __launch_bounds__(256) phi_cut_cov_kernel(const phi_ent_t *__restrict__ a_e1, const uint8_t *__restrict__ a_span, int64_t n_a,
                                                          int32_t *__restrict__ diff)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_a; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e1 = a_e1[g];
        const int32_t sp = a_span[g];
        if (sp <= 0) continue;
        atomicAdd(&diff[e1 - sp + 1], 1);
        atomicAdd(&diff[e1 + 1], -1);
    }
}
// clean[e] = 1 iff no anchor is split by a cut before entry e (cov_excl = exclusive prefix sums of diff)
__global__ void __launch_bounds__(256) phi_cut_clean_kernel(const int32_t *__restrict__ cov_excl, int64_t n_entries, int32_t *__restrict__ clean)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= n_entries; e += (int64_t)gridDim.x * blockDim.x)
        clean[e] = e < n_entries && cov_excl[e + 1] == 0;
}
// The same flags without counting: a cut before e is split by an anchor iff some anchor ending at x >= e has span > x - e,
// and spans are below PHI_RCAP, so x <= e + 30.  A block takes a tile of entries, looks through the anchors ending on the
// tile and the 30 entries behind it (the dp anchors are sorted by last entry: g_off is their CSR) and marks in LDS the
// entries each one covers; what is left unmarked is clean.  (The counting version -- two atomics per anchor into a difference
// array, a prefix sum over all entries, a pass for the flags -- took 36 ms at config 5; nothing else read the counts.)
#define CUT_TILE 4096
__global__ void __launch_bounds__(256) phi_cut_clean_direct_kernel(const int64_t *__restrict__ g_off, const uint8_t *__restrict__ g_span, int64_t n_entries,
                                                                   int32_t *__restrict__ clean)
{
    __shared__ uint8_t s_dirty[CUT_TILE];
    const int64_t n_tiles = (n_entries + 1 + CUT_TILE - 1) / CUT_TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t e0 = t * CUT_TILE;
        for (int i = threadIdx.x; i < CUT_TILE / 4; i += 256) reinterpret_cast<uint32_t *>(s_dirty)[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < CUT_TILE + PHI_RCAP - 1; i += 256) {
            const int64_t x = e0 + i;
            if (x >= n_entries) break;
            const int64_t lo = g_off[x], hi = g_off[x + 1];
            int m = 0;
            for (int64_t g = lo; g < hi; g++) m = max(m, (int)g_span[g]);
            // the entries e with x - m < e <= x, inside the tile
            for (int64_t e = max(e0, x - m + 1); e <= x && e < e0 + CUT_TILE; e++) s_dirty[e - e0] = 1;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CUT_TILE; i += 256) {
            const int64_t e = e0 + i;
            if (e <= n_entries) clean[e] = e < n_entries && !s_dirty[i];
        }
        __syncthreads();
    }
}
void phi_launch_cut_clean_direct(hipStream_t st, const int64_t *g_off, const uint8_t *g_span, int64_t n_entries, int32_t *clean)
{
    int64_t nb = (n_entries + 1 + CUT_TILE - 1) / CUT_TILE;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(phi_cut_clean_direct_kernel, dim3((unsigned)nb), dim3(256), 0, st, g_off, g_span, n_entries, clean);
}

// Between two consecutive events of a walk (entries l < p on compact steps a < b) the walk only runs along chain
// vertices: a cut before any step in (a, b] is fine for this walk iff some entry in (l, p] is clean.  Where none is,
// the steps a+1 .. b are closed: stepdiff[a + 1] += 1, stepdiff[b + 1] -= 1.
__global__ void __launch_bounds__(256) phi_cut_events_kernel(const phi_ent_t *__restrict__ ev_e, int64_t n_ev, const int64_t *__restrict__ ev_off,
                                                             const int64_t *__restrict__ walk_off, int32_t n_walks,
                                                             const int32_t *__restrict__ walk_vtx, const int32_t *__restrict__ cvtx,
                                                             const int32_t *__restrict__ ncl_excl, int32_t *__restrict__ stepdiff)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ev; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = ev_e[i];
        int lo = 0, hi = n_walks;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (walk_off[mid] <= p) lo = mid; else hi = mid;
        }
        if (i == ev_off[lo]) continue;                         // the walk's first event: nothing before it
        const int64_t l = ev_e[i - 1];
        if (ncl_excl[p + 1] - ncl_excl[l + 1] > 0) continue;   // a clean entry in (l, p]
        const int32_t a = cvtx[walk_vtx[l]], b = cvtx[walk_vtx[p]];
        atomicAdd(&stepdiff[a + 1], 1);
        atomicAdd(&stepdiff[b + 1], -1);
    }
}
// blk_ev[b][h] = first event of walk h on a compact step >= blk_lo[b]
__global__ void __launch_bounds__(256) phi_blk_ev_kernel(const int32_t *__restrict__ blk_lo, int32_t n_blk, const phi_ent_t *__restrict__ ev_e,
                                                         const int64_t *__restrict__ ev_off, int32_t n_walks, const int32_t *__restrict__ walk_vtx,
                                                         const int32_t *__restrict__ cvtx, int32_t *__restrict__ blk_ev)
{
    const int b = blockIdx.x, h = threadIdx.x;             // blockDim.x = the lane stride: 64 or 256
    if (b >= n_blk) return;
    int32_t out = 0;
    if (h < n_walks) {
        const int32_t k0 = blk_lo[b];
        int64_t lo = ev_off[h], hi = ev_off[h + 1];
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cvtx[walk_vtx[ev_e[mid]]] < k0) lo = mid + 1; else hi = mid;
        }
        out = (int32_t)lo;
    }
    blk_ev[(int64_t)b * blockDim.x + h] = out;
}
void phi_launch_cut_cov(hipStream_t st, const phi_ent_t *a_e1, const uint8_t *a_span, int64_t n_a, int32_t *diff)
{
    if (n_a <= 0) return;
    int64_t nb = (n_a + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(phi_cut_cov_kernel, dim3((unsigned)nb), dim3(256), 0, st, a_e1, a_span, n_a, diff);
}
void phi_launch_cut_clean(hipStream_t st, const int32_t *cov_excl, int64_t n_entries, int32_t *clean)
{
    int64_t nb = (n_entries + 1 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(phi_cut_clean_kernel, dim3((unsigned)nb), dim3(256), 0, st, cov_excl, n_entries, clean);
}
void phi_launch_cut_events(hipStream_t st, const phi_ent_t *ev_e, int64_t n_ev, const int64_t *ev_off, const int64_t *walk_off, int32_t n_walks,
                           const int32_t *walk_vtx, const int32_t *cvtx, const int32_t *ncl_excl, int32_t *stepdiff)
{
    if (n_ev <= 0) return;
    int64_t nb = (n_ev + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(phi_cut_events_kernel, dim3((unsigned)nb), dim3(256), 0, st, ev_e, n_ev, ev_off, walk_off, n_walks, walk_vtx, cvtx,
                       ncl_excl, stepdiff);
}
void phi_launch_blk_ev(hipStream_t st, const int32_t *blk_lo, int32_t n_blk, const phi_ent_t *ev_e, const int64_t *ev_off, int32_t n_walks,
                       const int32_t *walk_vtx, const int32_t *cvtx, int32_t *blk_ev)
{
    if (n_blk > 0)
        hipLaunchKernelGGL(phi_blk_ev_kernel, dim3((unsigned)n_blk), dim3(n_walks <= 64 ? 64 : 256), 0, st, blk_lo, n_blk, ev_e, ev_off, n_walks,
                           walk_vtx, cvtx, blk_ev);
}


// ------------------------------------------------------------------ more than 64 walks: blocks on CLASS LANES
// Inside a short block of steps most walks do the same thing: they run through the same vertices over the same
// anchors (haplotypes that share the alleles of the block's few sites).  What a block does to a walk is a function of
// the walk's event records inside it -- steps, entry distances, anchor counts relative to the walk's count at the
// block's first event, out-edges -- so walks whose records agree form a CLASS, and the block's transfer matrix is
// needed per class, not per walk: <= 64 class lanes run on the one-wave consumer kernel (DP_ROW), whatever the number
// of walks.  A block with more than 64 classes raises PHI_KERR_DP_CLASSES (the caller keeps the whole chain).
//
// Per run and block: sig(walk) = 128-bit hash of (begun before the block, events left, and per event inside the
// block: step - k0, entry - first entry, walk start / walk end, End - c, SB - c, the window counts G[a] for
// a <= entry - first entry (older windows reach before the cut and are never read for a run that began inside),
// the out-edge).  With c = SB at a walk's first event of the block, walks j, r of one class have
// End_j - End_r = SB_j - SB_r = c_j - c_r throughout the block; d_j = c_j - c_r(class of j) is the walk's OFFSET
// from the walk r that plays its class lane.
__device__ __forceinline__ unsigned long long cls_mix(unsigned long long h, unsigned long long x, unsigned long long m)
{
    h = (h ^ x) * m;
    return h ^ (h >> 29);
}
__global__ void __launch_bounds__(256) phi_blk_classes_kernel(PhiBlkClassArgs G)
{
    __shared__ unsigned long long s_a[256], s_b[256];
    __shared__ int32_t s_idx[256], s_c0[256];
    __shared__ int32_t s_wcnt[4];
    const int b = blockIdx.x, h = threadIdx.x, lane = h & 63, wid = h >> 6;
    const int32_t LS = G.lane_stride;
    const bool has_walk = h < G.n_walks;
    unsigned long long ha = 0x243F6A8885A308D3ull, hb = 0x13198A2E03707344ull;
    int32_t c0 = 0;
    if (has_walk) {
        const uint4 *evg = reinterpret_cast<const uint4 *>(G.ev);
        const int32_t k0 = G.blk_lo[b];
        const int64_t v0 = G.ev_off[h], ve = G.ev_off[h + 1];
        const int64_t vb = G.blk_ev[(int64_t)b * LS + h];
        const int64_t vend = b + 1 < G.n_blk ? (int64_t)G.blk_ev[(int64_t)(b + 1) * LS + h] : ve;
        const int64_t eb = G.walk_off[h], ee = G.walk_off[h + 1];
        const unsigned long long head = (vb > v0 ? 1u : 0u) | (vb < ve ? 2u : 0u);
        ha = cls_mix(ha, head, 0x9E3779B97F4A7C15ull); hb = cls_mix(hb, head, 0xC2B2AE3D27D4EB4Full);
        int64_t e_first = 0;
        for (int64_t i = vb; i < vend; i++) {
            const uint4 A = evg[i * 3 + 0], B = evg[i * 3 + 1], C = evg[i * 3 + 2];
            const int64_t e = (int64_t)A.y;
            if (i == vb) { e_first = e; c0 = (int32_t)A.w; }
            const int64_t rel = e - e_first;
            unsigned long long w[4] = {(unsigned long long)B.x | ((unsigned long long)B.y << 32), (unsigned long long)B.z | ((unsigned long long)B.w << 32),
                                       (unsigned long long)C.x | ((unsigned long long)C.y << 32), (unsigned long long)C.z | ((unsigned long long)C.w << 32)};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t keep = rel + 1 - 8 * q;              // bytes 8q .. 8q+7 hold the counts of ages 8q .. 8q+7
                unsigned long long m = keep >= 8 ? ~0ull : keep <= 0 ? 0ull : ((1ull << (8 * keep)) - 1);
                if (q == 3) m |= 0xFF00000000000000ull;            // byte 31: the out-edge
                w[q] &= m;
            }
            const unsigned long long f0 = ((unsigned long long)(uint32_t)((int32_t)(A.x & 0x7FFFFFFFu) - k0) << 32) | (unsigned long long)(uint32_t)rel;
            const unsigned long long f1 = ((unsigned long long)(uint32_t)((int32_t)A.z - c0) << 32) | (unsigned long long)(uint32_t)((int32_t)A.w - c0);
            unsigned long long f2 = (e == eb ? 1u : 0u) | (e == ee - 1 ? 2u : 0u);
            if (A.x >> 31) f2 |= 4u | ((unsigned long long)(h + 1) << 8);   // counted from the walk's own anchors: a class of its own
            const unsigned long long f[7] = {f0, f1, f2, w[0], w[1], w[2], w[3]};
#pragma unroll
            for (int q = 0; q < 7; q++) { ha = cls_mix(ha, f[q], 0x9E3779B97F4A7C15ull); hb = cls_mix(hb, f[q], 0xC2B2AE3D27D4EB4Full); }
        }
    }
    s_a[h] = ha; s_b[h] = hb; s_c0[h] = c0;
    if (h < 64) G.lane_walk[(int64_t)b * 64 + h] = -1;
    __syncthreads();
    // the class of a walk is led by the first walk with its signature
    int32_t leader = h;
    if (has_walk)
        for (int j = 0; j < h; j++)
            if (s_a[j] == ha && s_b[j] == hb) { leader = j; break; }
    const bool is_leader = has_walk && leader == h;
    const unsigned long long bal = __ballot(is_leader);
    if (lane == 0) s_wcnt[wid] = __popcll(bal);
    __syncthreads();
    int32_t before = __popcll(bal & ((1ull << lane) - 1));
    for (int x = 0; x < wid; x++) before += s_wcnt[x];
    const int32_t total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    s_idx[h] = before;
    __syncthreads();
    if (h == 0) {
        G.blk_ncls[b] = total < 64 ? total : 64;
        if (total > 64) atomicOr(G.err, PHI_KERR_DP_CLASSES);
    }
    if (has_walk) {
        const int32_t idx = s_idx[leader];
        if (is_leader && idx < 64) G.lane_walk[(int64_t)b * 64 + idx] = h;
        G.walk_lane[(int64_t)b * LS + h] = idx < 64 ? idx : 0;
        G.coff[(int64_t)b * LS + h] = c0 - s_c0[leader];       // relative to the walk that plays the class lane

    }
}

// The chain over the blocks, on class lanes: S_{b+1}[x] for every walk x from S_b and the block's class rows.
// With l(x) the class lane of walk x in block b and d the offsets (keys of walk x are keys of its lane minus d[x]):
//   own run, and runs begun on x's lane by it       S[x] + row[l(x)][l(x)]
//   another walk j of x's class                     S[j] + d[j] + rownew[l(x)] - d[x]
//   a walk j of another class C                     S[j] + d[j] + row[C][l(x)] - d[x]
//   a walk that starts inside the block             row[64][l(x)] - d[x]
// Only the best S[j] + d[j] of a class matters (and, for the walks that reach it, whether another walk does too or
// else the second best).  One workgroup, one block per iteration, two barriers; thread (wave w, lane l) holds, in
// registers, column l of the rows of classes C = w, w + 4, ..: the "another class" term is a function of the class
// lane alone, so the four waves split the classes and leave four partial maxima per lane in LDS.  Rows and tables
// of block b + DEPTH are on their way while block b is chained.  All sums stay far inside int32: keys and offsets are
// bounded by the anchors of one walk (< 2^27, phi_solve.hip).
//
// The chain is a sequence of max-plus AFFINE maps of the walks' keys (S' = M_b (x) S (+) T_b: every term above is a maximum of
// sums, T_b the walk starts inside the block), so it can be cut: MODE 1 runs the blocks of one SEGMENT from a unit vector
// ("key 0 on walk u", walk starts switched off: column u of the segment's matrix) or from no key at all with the starts on
// (its constant term) -- segments x (walks + 1) independent workgroups --, phi_seg_chain_kernel chains the segments'
// matrices (one workgroup, segments x walks rows), and MODE 2 replays every segment from its true entry keys, in
// parallel, writing what MODE 0 (the whole chain by one workgroup) writes.  C5: 94 671 blocks in 64 segments.
// U: unit vectors one workgroup of MODE 1 carries through its segment AT ONCE.  The blocks' rows are the same for every unit
// vector of a segment (24 MB per segment at C5, read by 201 workgroups: 316 GB through the L2s per DP run), and what a
// workgroup spends per block is latency -- two barriers, LDS atomics, dependent reads --, not arithmetic: with U chains
// interleaved in one instruction stream a block's rows are loaded once for U of them and the barriers are shared.
template <int MODE, int U = 1>
__global__ void __launch_bounds__(256) phi_blk_chain_kernel(PhiBlkClassArgs G, const int32_t *__restrict__ rows, const int32_t *__restrict__ rownew,
                                                            const int32_t *__restrict__ rowdiag, int32_t *__restrict__ blk_S,
                                                            const int32_t *__restrict__ seg_lo, int32_t *__restrict__ seg_row,
                                                            const int32_t *__restrict__ seg_S)
{
    static_assert(MODE == 1 || U == 1, "several unit vectors at once: the segments' matrices only");
    constexpr int NR = 65 * 64;
    constexpr int DEPTH = 4;
    // "no value" is NEGK everywhere: NEGK + a key (|key| < 2^27) and NEGK + NEGK stay below NEGK / 2 and inside int32, so a
    // maximum of sums needs no test of its operands (the loop is the longest dependent chain of an iteration: one
    // workgroup walks 10^5 blocks, 1.1 us each, and what it costs is its instruction count -- C5: 112 ms per DP run)
    constexpr int32_t NONE = NEGK;
    __shared__ int32_t s_b1[2][U][64], s_b2[2][U][64], s_n1[2][U][64];  // per class: best S + d, second best, walks that reach the best
    __shared__ int32_t s_diag[2][64], s_start[2][64], s_new[2][64];
    __shared__ int32_t s_part[2][U][4][64];
    const int x = threadIdx.x, lane = x & 63, wid = __builtin_amdgcn_readfirstlane(x >> 6);
    const int32_t LS = G.lane_stride;
    const bool has_walk = x < G.n_walks;
    // the blocks this workgroup chains, the keys it starts from, whether walks may start inside
    const int32_t groups = MODE == 1 ? (G.n_walks + 1 + U - 1) / U : 1;
    const int32_t seg = MODE == 0 ? 0 : MODE == 1 ? (int32_t)blockIdx.x / groups : (int32_t)blockIdx.x;
    const int32_t unit0 = MODE == 1 ? ((int32_t)blockIdx.x % groups) * U : -1;
    const int32_t b_lo = MODE == 0 ? 0 : seg_lo[seg], nb = MODE == 0 ? G.n_blk : seg_lo[seg + 1];
    int32_t S[U];
    bool with_starts[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        with_starts[u] = MODE != 1 || unit0 + u == G.n_walks;           // (units past the last one carry nothing: all keys stay NEGK)
        S[u] = NEGK;
        if (MODE == 1 && x == unit0 + u) S[u] = 0;
        if (MODE == 2 && has_walk) S[u] = seg_S[(int64_t)seg * LS + x];
    }
    int32_t st_row[DEPTH][16], st_x[DEPTH], st_lx[DEPTH], st_dx[DEPTH];
    const int32_t *p_row = rows + wid * 64 + lane;   // + b * NR + 256 * i
    const int32_t *p_x = wid == 0 ? rows + 64 * 64 + lane : wid == 1 ? rownew + lane : rowdiag + lane;   // + b * (NR | 65 | 64): walk starts (wave 0), new-run keys (wave 1), the rows' own columns (wave 2)
    const int32_t x_stride = wid == 0 ? NR : wid == 1 ? 65 : 64;
    const int32_t xs = has_walk ? x : 0;
    auto issue = [&](auto J, int32_t b) {
        constexpr int j = decltype(J)::value;
        const int32_t *src = p_row + (int64_t)b * NR;
#pragma unroll
        for (int i = 0; i < 16; i++) st_row[j][i] = src[256 * i];
        st_x[j] = wid < 3 ? p_x[(int64_t)b * x_stride] : NEGK;
        st_lx[j] = G.walk_lane[(int64_t)b * LS + xs];
        st_dx[j] = G.coff[(int64_t)b * LS + xs];
    };
    auto step = [&](auto J, int32_t b) {
        constexpr int j = decltype(J)::value;
        const int p = b & 1;
        const int32_t lx = st_lx[j], dx = st_dx[j];
        if (wid == 0) s_start[p][lane] = st_x[j];                         // (lanes past the block's classes are never read)
        if (wid == 1) s_new[p][lane] = st_x[j];
        if (wid == 2) s_diag[p][lane] = st_x[j];
        if (MODE != 1 && has_walk) blk_S[(int64_t)b * LS + x] = S[0];
        bool live[U];
        int32_t v[U], b1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            live[u] = has_walk && S[u] > NEGK / 2;
            v[u] = S[u] + dx;
            if (live[u]) atomicMax(&s_b1[p][u][lx], v[u]);               // (reset in the previous iteration, a barrier ago)
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            b1[u] = s_b1[p][u][lx];
            if (live[u]) { if (v[u] == b1[u]) atomicAdd(&s_n1[p][u][lx], 1); else atomicMax(&s_b2[p][u][lx], v[u]); }
        }
        if (x < 64) {
#pragma unroll
            for (int u = 0; u < U; u++) { s_b1[p ^ 1][u][x] = NONE; s_b2[p ^ 1][u][x] = NONE; s_n1[p ^ 1][u][x] = 0; }   // for the next iteration (last read before this iteration's first barrier)
        }
        // partial maxima over this wave's classes, for class lane `lane`
        // (branch-free: the sixteen LDS reads go out together.  Classes past the block's count have no live walk: their
        //  maximum is NONE, and NONE plus whatever lies in their rows -- keys of an earlier solve at most -- stays "no
        //  value"; the row's own column holds NEGK, see the row pass)
#pragma unroll
        for (int u = 0; u < U; u++) {
            int32_t part = NONE;
            int32_t o16[16];
#pragma unroll
            for (int i = 0; i < 16; i++) o16[i] = s_b1[p][u][wid + 4 * i];
#pragma unroll
            for (int i = 0; i < 16; i++) part = max(part, o16[i] + st_row[j][i]);
            s_part[p][u][wid][lane] = part;
        }
        issue(J, min(b + DEPTH, nb - 1));                      // (always: the compiler can then count the loads in flight instead of draining them)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            int32_t best = NEGK;
            if (has_walk) {
                if (live[u]) { const int32_t r = s_diag[p][lx]; if (r > NEGK / 2) best = S[u] + r; }
                {
                    // the best of the other walks of the class
                    const int32_t o = (live[u] && v[u] == b1[u] && s_n1[p][u][lx] == 1) ? s_b2[p][u][lx] : b1[u];
                    const int32_t kn = s_new[p][lx];
                    if (o > NEGK / 2 && kn > NEGK / 2 && o + kn - dx > best) best = o + kn - dx;
                }
                const int32_t d = max(max(s_part[p][u][0][lx], s_part[p][u][1][lx]), max(s_part[p][u][2][lx], s_part[p][u][3][lx]));
                if (d > NEGK / 2 && d - dx > best) best = d - dx;
                if (with_starts[u]) { const int32_t r = s_start[p][lx]; if (r > NEGK / 2 && r - dx > best) best = r - dx; }
            }
            S[u] = best > NEGK / 2 ? best : NEGK;
        }
        // (no barrier here: the next iteration writes the other parity only, and what it reads of it was settled
        //  before this iteration's second barrier)
    };
    if (x < 64) {
#pragma unroll
        for (int u = 0; u < U; u++) { s_b1[b_lo & 1][u][x] = NONE; s_b2[b_lo & 1][u][x] = NONE; s_n1[b_lo & 1][u][x] = 0; }
    }
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    issue(I0{}, min(b_lo, nb - 1));
    issue(I1{}, min(b_lo + 1, nb - 1));
    issue(I2{}, min(b_lo + 2, nb - 1));
    issue(I3{}, min(b_lo + 3, nb - 1));
    __syncthreads();
    int32_t b = b_lo;
    for (; b + DEPTH <= nb; b += DEPTH) {
        step(I0{}, b);
        step(I1{}, b + 1);
        step(I2{}, b + 2);
        step(I3{}, b + 3);
    }
    if (b < nb) step(I0{}, b);
    if (b + 1 < nb) step(I1{}, b + 1);
    if (b + 2 < nb) step(I2{}, b + 2);
    if (MODE == 1 && has_walk) {
#pragma unroll
        for (int u = 0; u < U; u++)
            if (unit0 + u <= G.n_walks) seg_row[((int64_t)seg * (G.n_walks + 1) + unit0 + u) * LS + x] = S[u];      // what leaves the segment
    }
}

// The segments' matrices chained: seg_S[g] = the keys entering segment g.  seg_row[(g * (walks + 1) + u) * LS + x] = the key
// leaving segment g on walk x when key 0 entered on walk u (u = walks: when nothing entered -- the walk starts inside).
__global__ void __launch_bounds__(256) phi_seg_chain_kernel(int32_t n_seg, int32_t n_walks, int32_t LS, const int32_t *__restrict__ seg_row,
                                                            int32_t *__restrict__ seg_S)
{
    __shared__ int32_t s_S[256];
    const int x = threadIdx.x;
    int32_t S = NEGK;
    for (int32_t g = 0; g < n_seg; g++) {
        seg_S[(int64_t)g * LS + x] = S;
        s_S[x] = S;
        __syncthreads();
        const int32_t *R = seg_row + (int64_t)g * (n_walks + 1) * LS + x;
        int32_t best = R[(int64_t)n_walks * LS];
#pragma unroll 8
        for (int32_t u = 0; u < n_walks; u++) best = max(best, s_S[u] + R[(int64_t)u * LS]);   // (NEGK + anything stays "no value")
        S = best > NEGK / 2 ? best : NEGK;
        __syncthreads();
    }
}

// the two passes must agree on what leaves every block: keys_out[b] (DP_PATH) against S[b + 1] (the chain)
__global__ void __launch_bounds__(256) phi_blk_check_kernel(const int32_t *__restrict__ keys, const int32_t *__restrict__ S, int32_t n_blk,
                                                            int32_t LS, int32_t n_walks, int32_t *__restrict__ bad)
{
    const int64_t n = (int64_t)(n_blk - 1) * LS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t h = (int32_t)(i % LS);
        if (h >= n_walks) continue;
        const int32_t got = keys[i], want = S[i + LS];
        // (a walk that begins inside the next block or ended before it carries nothing that is ever read)
        if (got != want && !(got <= NEGK / 2 && want <= NEGK / 2)) { atomicAdd(&bad[0], 1); atomicMin(&bad[1], (int32_t)(i / LS)); }
    }
}

// backtrack across blocks: the last block before b_from in which the run carried on walk h began -> out = {block, start}
__global__ void __launch_bounds__(256) phi_carry_resolve_kernel(const int32_t *__restrict__ carry, int32_t LS, int32_t b_from, int32_t h,
                                                                int32_t *__restrict__ out)
{
    __shared__ int32_t s_best;
    for (int32_t hi = b_from - 1; hi >= 0; hi -= 256) {
        if (threadIdx.x == 0) s_best = -1;
        __syncthreads();
        const int32_t b = hi - (int32_t)threadIdx.x;
        const int32_t v = b >= 0 ? carry[(int64_t)b * LS + h] : -1;
        if (v >= 0) atomicMax(&s_best, b);
        __syncthreads();
        const int32_t bb = s_best;
        if (bb >= 0) {
            if (b == bb) { out[0] = bb; out[1] = v; }
            return;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = -1; out[1] = -1; }
}

void phi_launch_blk_classes(hipStream_t st, const PhiBlkClassArgs &G)
{
    if (G.n_blk > 0) hipLaunchKernelGGL(phi_blk_classes_kernel, dim3((unsigned)G.n_blk), dim3(256), 0, st, G);
}
void phi_launch_blk_chain(hipStream_t st, const PhiBlkClassArgs &G, const int32_t *rows, const int32_t *rownew, const int32_t *rowdiag, int32_t *blk_S)
{
    hipLaunchKernelGGL(phi_blk_chain_kernel<0>, dim3(1), dim3(256), 0, st, G, rows, rownew, rowdiag, blk_S, nullptr, nullptr, nullptr);
}
// the same chain cut into n_seg segments (d_seg_lo[n_seg + 1]: first block of each): unit rows, segment chain, replay
void phi_launch_blk_chain_segments(hipStream_t st, const PhiBlkClassArgs &G, const int32_t *rows, const int32_t *rownew, const int32_t *rowdiag, int32_t *blk_S,
                                   int32_t n_seg, const int32_t *d_seg_lo, int32_t *seg_row, int32_t *seg_S)
{
    const int units = getenv("PHI_DP_CHAIN_UNITS") ? atoi(getenv("PHI_DP_CHAIN_UNITS")) : 3;      // unit vectors per workgroup (tests: 1, 2)
    const unsigned nu = (unsigned)(G.n_walks + 1);
    if (units >= 4) hipLaunchKernelGGL((phi_blk_chain_kernel<1, 4>), dim3((unsigned)n_seg * ((nu + 3) / 4)), dim3(256), 0, st, G, rows, rownew, rowdiag, blk_S, d_seg_lo, seg_row, seg_S);
    else if (units == 3) hipLaunchKernelGGL((phi_blk_chain_kernel<1, 3>), dim3((unsigned)n_seg * ((nu + 2) / 3)), dim3(256), 0, st, G, rows, rownew, rowdiag, blk_S, d_seg_lo, seg_row, seg_S);
    else if (units == 2) hipLaunchKernelGGL((phi_blk_chain_kernel<1, 2>), dim3((unsigned)n_seg * ((nu + 1) / 2)), dim3(256), 0, st, G, rows, rownew, rowdiag, blk_S, d_seg_lo, seg_row, seg_S);
    else hipLaunchKernelGGL((phi_blk_chain_kernel<1, 1>), dim3((unsigned)n_seg * nu), dim3(256), 0, st, G, rows, rownew, rowdiag, blk_S, d_seg_lo, seg_row, seg_S);
    hipLaunchKernelGGL(phi_seg_chain_kernel, dim3(1), dim3(256), 0, st, n_seg, G.n_walks, G.lane_stride, seg_row, seg_S);
    hipLaunchKernelGGL(phi_blk_chain_kernel<2>, dim3((unsigned)n_seg), dim3(256), 0, st, G, rows, rownew, rowdiag, blk_S, d_seg_lo, seg_row, seg_S);
}
void phi_launch_blk_check(hipStream_t st, const int32_t *keys, const int32_t *S, int32_t n_blk, int32_t LS, int32_t n_walks, int32_t *bad)
{
    int64_t nb = ((int64_t)(n_blk - 1) * LS + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(phi_blk_check_kernel, dim3((unsigned)nb), dim3(256), 0, st, keys, S, n_blk, LS, n_walks, bad);
}
void phi_launch_carry_resolve(hipStream_t st, const int32_t *carry, int32_t LS, int32_t b_from, int32_t h, int32_t *out)
{
    hipLaunchKernelGGL(phi_carry_resolve_kernel, dim3(1), dim3(256), 0, st, carry, LS, b_from, h, out);
}
// DP_PATH on walk lanes for more than 64 walks (blocks of at most 512 steps)
void phi_launch_dp_block_paths_wide(hipStream_t st, const PhiDpEventArgs &A)
{
    if (A.n_walks <= 128) hipLaunchKernelGGL((phi_dp_events_kernel<2, DP_PATH>), dim3((unsigned)A.n_blk), dim3(128), 0, st, A);
    else hipLaunchKernelGGL((phi_dp_events_kernel<4, DP_PATH>), dim3((unsigned)A.n_blk), dim3(256), 0, st, A);
}

void phi_launch_dp_events(hipStream_t st, const PhiDpEventArgs &A)
{
    if (A.n_walks <= 64) hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_SEQ, 8, RING, 32>), dim3(1), dim3(64 * 3), 0, st, A);
    else if (A.n_walks <= 128) hipLaunchKernelGGL((phi_dp_events_kernel<2, DP_SEQ>), dim3(1), dim3(128), 0, st, A);
    else hipLaunchKernelGGL((phi_dp_events_kernel<4, DP_SEQ>), dim3(1), dim3(256), 0, st, A);    // n_walks <= PHI_DP_EVENT_MAX_WALKS
}

// blocks of steps in parallel (<= 64 walks): the rows of the blocks' transfer matrices, then (the host has chained
// them into the entry vector of every block) the blocks themselves.  A block is at most as long as its ring of tops
// (A.blk_ring: 1024 steps, two workgroups per CU; 2048 for graphs whose longest stretch without a cut needs it).
void phi_launch_dp_block_rows(hipStream_t st, const PhiDpEventArgs &A)
{
    const unsigned grid = (unsigned)A.n_blk * (A.lane_walk ? 65u : (unsigned)(A.n_walks + 1));
    // (blocks of at most 256 steps: a ring of 256 tops and queues of 8 runs make a task 51 KB of LDS, three per CU
    //  instead of two -- the consumer wave is latency-bound, so throughput follows the number of resident tasks;
    //  the second pass keeps this layout, the rows go one step further:)
    // (a period of two steps instead of four halves the event ring and the result ring: 35 KB, FOUR tasks per CU; the extra
    //  barriers cost the consumer less than the fourth task gains -- C2 rows 2.8 -> 2.3 ms with their copy to the host)
    if (A.blk_ring <= 256 && A.blk_max_len > 0 && A.blk_max_len <= 64 && !getenv("PHI_DP_ROWS_LARGE"))
        hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_ROW, 2, 64, 8, 32>), dim3(grid), dim3(64 * 3), 0, st, A);
    else if (A.blk_ring <= 256) hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_ROW, 2, 256, 8>), dim3(grid), dim3(64 * 3), 0, st, A);
    else if (A.blk_ring <= 1024) hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_ROW, 4, 1024, 16>), dim3(grid), dim3(64 * 3), 0, st, A);
    else hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_ROW, 4, 2048, 16>), dim3(grid), dim3(64 * 3), 0, st, A);
}
void phi_launch_dp_block_paths(hipStream_t st, const PhiDpEventArgs &A)
{
    if (A.blk_ring <= 256) hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_PATH, 4, 256, 8>), dim3((unsigned)A.n_blk), dim3(64 * 3), 0, st, A);
    else if (A.blk_ring <= 1024) hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_PATH, 4, 1024, 16>), dim3((unsigned)A.n_blk), dim3(64 * 3), 0, st, A);
    else hipLaunchKernelGGL((phi_dp_events_pc_kernel<2, DP_PATH, 4, 2048, 16>), dim3((unsigned)A.n_blk), dim3(64 * 3), 0, st, A);
}

// One empty launch loads this translation unit's code object onto the device: the HIP runtime does that lazily, at the
// first launch of any of its kernels (0.5-1.3 ms per unit, measured inside phi_set_graph / phi_solve before
// phi_ctx_create did it up front).
__global__ void phi_warm_dp_events_kernel() {}
void phi_warm_dp_events(hipStream_t st) { hipLaunchKernelGGL(phi_warm_dp_events_kernel, dim3(1), dim3(64), 0, st); }
