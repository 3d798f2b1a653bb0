// walk_text.hip -- the walks of a GFA resolved ON THE DEVICE from the text of its W-lines.
//
// The reference parses every W-line on one core (gfa-io.cpp:367-432: a walk is ">s17>s18<s19..." -- a step is a strand
// character and a segment name), then ILP_index::read_gfa copies the vertices into paths[h] (ILP_index.cpp:96-113).  At
// chromosome scale the walks ARE the file: config 5's GFA is 10.9 GB, 10.5 GB of it walk text, and resolving it was the
// largest host stage left (0.76 s on 16 threads, then 5.3 GB of walk entries to upload).  Here the text goes to HBM as it
// is (phi_walk_text_upload: pinned staging, while the host still enters the segment names) and four kernels resolve it
// (phi_walk_text_resolve):
//     tab      the first tab of every walk (optional tags follow the walk field: nothing behind it is a step)
//     count    steps ('>' / '<') per 4-KB tile; a '<' anywhere makes the graph IRREGULAR
//     scan     exclusive scan of the tile counts = where every tile writes, and the walk offsets
//     parse    every step's name -> vertex: <prefix><canonical decimal, at most 9 digits> -> num2id[number]; any other name
//              form, an unknown number or a name the table does not hold makes the graph IRREGULAR
// An irregular graph (reverse steps -- the reference flips such walks by majority strand, gfa-io.cpp:64-115 --, names that
// are not <prefix><number>, steps naming no segment -- the reference leaves those out --) is NOT resolved here: the caller
// falls back to the host reader (phi_graph_resolve_walks), whose rules are the reference's.  What this path accepts it
// resolves exactly as the host reader does (tests/test_gpu_walk_text.py: the reference's MHC_4 graph, tagged W-lines,
// mixed name forms, reversed walks).
#include <string.h>
#include <thread>
#include <vector>
#include "phi_ctx.h"
#include "phi_dev.h"

#define HIPCHK(call) do { int rc_ = phi_hip_check(c, (call), #call); if (rc_) return rc_; } while (0)
#define PHICHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

#define WT_TILE 4096            // bytes per tile; every walk's text starts on a tile boundary of the device buffer

namespace {

__device__ __forceinline__ bool is_step(unsigned c) { return c == '>' || c == '<'; }

// walk of a tile: tile0[w] <= t < tile0[w + 1]
__device__ __forceinline__ int walk_of_tile(const int64_t *__restrict__ tile0, int n_walks, int64_t t)
{
    int lo = 0, hi = n_walks;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile0[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(256) wt_tab_kernel(const uint8_t *__restrict__ text, const int64_t *__restrict__ tile0, int n_walks, int64_t n_tiles,
                                                     unsigned long long *__restrict__ walk_end)
{
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int w = walk_of_tile(tile0, n_walks, t);
        const int64_t base = t * WT_TILE + 16 * (int64_t)threadIdx.x;
        if ((unsigned long long)base >= walk_end[w]) continue;           // (racy read of a value that only shrinks: a stale one only costs a look)
        const uint4 v = *reinterpret_cast<const uint4 *>(text + base);
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
        int first = -1;
#pragma unroll
        for (int q = 3; q >= 0; q--) {
            const uint32_t y = x[q] ^ 0x09090909u;
            if ((y - 0x01010101u) & ~y & 0x80808080u)
                for (int j = 3; j >= 0; j--) if (((x[q] >> (8 * j)) & 0xFFu) == '\t') first = 4 * q + j;
        }
        if (first >= 0) atomicMin(&walk_end[w], (unsigned long long)(base + first));
    }
}

// steps of this thread's 16 bytes (below the walk's end), as a bit mask; *rev: a '<' among them
__device__ __forceinline__ uint32_t step_mask16(const uint8_t *__restrict__ text, int64_t base, int64_t end, bool *rev)
{
    if (base >= end) return 0;
    const uint4 v = *reinterpret_cast<const uint4 *>(text + base);
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned ch = (x[q] >> (8 * j)) & 0xFFu;
            if (is_step(ch) && base + 4 * q + j < end) { m |= 1u << (4 * q + j); *rev |= ch == '<'; }
        }
    return m;
}

__global__ void __launch_bounds__(256) wt_count_kernel(const uint8_t *__restrict__ text, const int64_t *__restrict__ tile0, int n_walks, int64_t n_tiles,
                                                       const unsigned long long *__restrict__ walk_end, int32_t *__restrict__ tile_cnt, uint32_t *__restrict__ irregular)
{
    __shared__ int s_cnt;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const int w = walk_of_tile(tile0, n_walks, t);
        bool rev = false;
        const uint32_t m = step_mask16(text, t * WT_TILE + 16 * (int64_t)threadIdx.x, (int64_t)walk_end[w], &rev);
        int n = __popc(m);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
        if ((threadIdx.x & 63) == 0 && n) atomicAdd(&s_cnt, n);
        if (rev) atomicOr(irregular, 1u);
        __syncthreads();
        if (threadIdx.x == 0) tile_cnt[t] = s_cnt;
        __syncthreads();
    }
}

struct WtParseArgs {
    const uint8_t *text; const int64_t *tile0; int n_walks; int64_t n_tiles;
    const unsigned long long *walk_end; const int64_t *tile_off;
    const uint8_t *prefix; int prefix_n;            // (device copy of the prefix)
    const int32_t *num2id; int64_t n_num; int32_t n_seg;
    int32_t *walk_vtx; uint32_t *irregular;
};

__global__ void __launch_bounds__(256) wt_parse_kernel(WtParseArgs A)
{
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int64_t t = blockIdx.x; t < A.n_tiles; t += gridDim.x) {
        const int w = walk_of_tile(A.tile0, A.n_walks, t);
        const int64_t end = (int64_t)A.walk_end[w];
        const int64_t base = t * WT_TILE + 16 * (int64_t)threadIdx.x;
        bool rev = false;
        uint32_t m = step_mask16(A.text, base, end, &rev);
        // rank of this thread's first step inside the tile: wave scan by shuffles, then the four waves' totals
        const int n = __popc(m);
        int inc = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
        if (lane == 63) s_w[wid] = inc;
        __syncthreads();
        int64_t rank = A.tile_off[t] + inc - n;
        for (int x = 0; x < wid; x++) rank += s_w[x];
        __syncthreads();
        while (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1;
            const uint8_t *p = A.text + base + j + 1;                  // the name: up to the next step or the walk's end
            const int64_t room = end - (base + j + 1);
            int32_t id = -1;
            bool ok = room > A.prefix_n;
            for (int i = 0; ok && i < A.prefix_n; i++) ok = p[i] == A.prefix[i];
            if (ok) {
                const uint8_t *d = p + A.prefix_n;
                const int64_t left = room - A.prefix_n;
                uint64_t num = 0;
                int nd = 0;
                while (nd < left && nd < 10 && (unsigned)(d[nd] - '0') <= 9u) { num = num * 10 + (unsigned)(d[nd] - '0'); nd++; }
                // the digits must be the whole name (the next byte starts a step, or the walk ends), at most nine, canonical
                ok = nd >= 1 && nd <= 9 && (nd == left || is_step(d[nd])) && (d[0] != '0' || nd == 1);
                if (ok && (int64_t)num < A.n_num) id = A.num2id[num];
            }
            if (id < 0 || id >= A.n_seg) { atomicOr(A.irregular, 2u); id = 0; }
            A.walk_vtx[rank++] = id;
        }
    }
}

__global__ void wt_ends_kernel(const int32_t *__restrict__ walk_vtx, const int64_t *__restrict__ walk_off, int n_walks, int32_t *__restrict__ ends)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_walks) return;
    const int64_t lo = walk_off[w], hi = walk_off[w + 1];
    ends[2 * w] = hi > lo ? walk_vtx[lo] : -1;
    ends[2 * w + 1] = hi > lo ? walk_vtx[hi - 1] : -1;
}

__global__ void wt_walk_off_kernel(const int64_t *__restrict__ tile0, const int64_t *__restrict__ tile_off, int n_walks, int64_t *__restrict__ walk_off)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w <= n_walks) walk_off[w] = tile_off[tile0[w]];
}

}  // namespace

extern "C" {

// The walk fields of the W-lines, as they stand in the file (host memory: a mapping will do), to the device.  Asynchronous on
// the context's second stream behind pinned staging: returns when the last piece has been handed to the copy engine.
int phi_walk_text_upload(phi_ctx *c, const phi_walk_text *walks, int32_t n_walks)
{
    if (!c || n_walks < 0 || (n_walks > 0 && !walks)) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    auto &W = c->wtext;
    W.ready = false;
    if (n_walks == 0) {                                // nothing to resolve: text of an earlier call is let go
        if (W.d_text.p) { (void)hipFree(W.d_text.p); W.d_text = DevBuf{}; }
        W.n_walks = 0; W.tile0.assign(1, 0); W.t_len.clear();
        W.ready = true;
        return PHI_OK;
    }
    W.n_walks = n_walks;
    W.tile0.assign((size_t)n_walks + 1, 0);
    W.t_len.assign((size_t)n_walks, 0);
    for (int32_t w = 0; w < n_walks; w++) {
        if (walks[w].n < 0 || (walks[w].n > 0 && !walks[w].text)) return phi_fail(c, PHI_ERR_INVALID, "phi_walk_text_upload: walk %d has no text", w);
        W.t_len[(size_t)w] = walks[w].n;
        W.tile0[(size_t)w + 1] = W.tile0[(size_t)w] + (walks[w].n + WT_TILE - 1) / WT_TILE;
    }
    const int64_t n_tiles = W.tile0[(size_t)n_walks];
    PHICHK(phi_dev_ensure(c, W.d_text, (size_t)n_tiles * WT_TILE + 256));
    // pinned staging, two buffers, four threads filling one while the other is on its way (as phi_set_graph's large uploads)
    constexpr size_t PIECE = (size_t)64 << 20;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc(&stage[i], PIECE, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    }
    auto release = [&]() {
        for (int i = 0; i < 2; i++) { if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); } if (stage[i]) (void)hipHostFree(stage[i]); }
    };
    if (e != hipSuccess) { release(); return phi_hip_check(c, e, "pinned staging buffers"); }
    int k = 0;
    int rc = PHI_OK;
    for (int32_t w = 0; w < n_walks && !rc; w++) {
        char *dst = W.d_text.as<char>() + (size_t)W.tile0[(size_t)w] * WT_TILE;
        const char *src = walks[w].text;
        const size_t bytes = (size_t)walks[w].n;
        for (size_t off = 0; off < bytes && !rc; off += PIECE, k ^= 1) {
            const size_t n = std::min(PIECE, bytes - off);
            rc = phi_hip_check(c, hipEventSynchronize(ev[k]), "hipEventSynchronize");      // (the copy that last read this buffer; a fresh event is complete)
            if (rc) break;
            {
                const int nt = n >= ((size_t)4 << 20) ? 4 : 1;
                std::vector<std::thread> th;
                char *d = static_cast<char *>(stage[k]);
                const char *s = src + off;
                for (int t = 1; t < nt; t++) th.emplace_back([=]() { memcpy(d + n * t / nt, s + n * t / nt, n * (t + 1) / nt - n * t / nt); });
                memcpy(d, s, n / nt);
                for (auto &x : th) x.join();
            }
            rc = phi_hip_check(c, hipMemcpyAsync(dst + off, stage[k], n, hipMemcpyHostToDevice, c->aux_stream), "hipMemcpyAsync");
            if (!rc) rc = phi_hip_check(c, hipEventRecord(ev[k], c->aux_stream), "hipEventRecord");
        }
    }
    if (!rc) rc = phi_hip_check(c, hipStreamSynchronize(c->aux_stream), "hipStreamSynchronize");
    release();
    if (rc) return rc;
    W.ready = true;
    return PHI_OK;
}

// The uploaded walks resolved into the context's walk entries.  walk_off_out[n_walks + 1].  *irregular != 0: nothing was
// resolved (bit 0: a reverse step, bit 1: a name this path does not resolve) -- fall back to the host reader.
int phi_walk_text_resolve(phi_ctx *c, const char *prefix, int32_t prefix_n, const int32_t *num2id, int64_t n_num, int32_t n_seg,
                          int64_t *walk_off_out, uint32_t *irregular)
{
    if (!c || !walk_off_out || !irregular || prefix_n < 0 || prefix_n > 16 || n_num < 0 || (n_num > 0 && !num2id)) return PHI_ERR_INVALID;
    auto &W = c->wtext;
    if (!W.ready) return phi_fail(c, PHI_ERR_STATE, "phi_walk_text_resolve before phi_walk_text_upload");
    HIPCHK(hipSetDevice(c->device));
    PhiStageTimer tm("walk text");
    *irregular = 0;
    c->walks_on_device = false;
    const int n_walks = W.n_walks;
    const int64_t n_tiles = W.tile0[(size_t)n_walks];
    if (n_tiles >= ((int64_t)1 << 31)) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 8 TB of walk text");
    DevBuf d_tile0, d_end, d_cnt, d_off, d_num2id, d_prefix, d_woff, d_flag;
    struct Guard { std::vector<DevBuf *> b; ~Guard() { for (DevBuf *x : b) if (x->p) (void)hipFree(x->p); } } guard{{&d_tile0, &d_end, &d_cnt, &d_off, &d_num2id, &d_prefix, &d_woff, &d_flag}};
    PHICHK(phi_dev_ensure(c, d_tile0, ((size_t)n_walks + 1) * 8));
    PHICHK(phi_dev_ensure(c, d_end, ((size_t)n_walks + 1) * 8));
    PHICHK(phi_dev_ensure(c, d_cnt, ((size_t)n_tiles + 1) * 4));
    PHICHK(phi_dev_ensure(c, d_off, ((size_t)n_tiles + 2) * 8));
    PHICHK(phi_dev_ensure(c, d_num2id, (size_t)std::max<int64_t>(n_num, 1) * 4));
    PHICHK(phi_dev_ensure(c, d_prefix, 64));
    PHICHK(phi_dev_ensure(c, d_woff, ((size_t)n_walks + 1) * 8));
    PHICHK(phi_dev_ensure(c, d_flag, 64));
    std::vector<unsigned long long> h_end((size_t)n_walks + 1, 0);
    for (int w = 0; w < n_walks; w++) h_end[(size_t)w] = (unsigned long long)(W.tile0[(size_t)w] * WT_TILE + W.t_len[(size_t)w]);
    HIPCHK(hipMemcpyAsync(d_tile0.p, W.tile0.data(), ((size_t)n_walks + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_end.p, h_end.data(), ((size_t)n_walks + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n_num) HIPCHK(hipMemcpyAsync(d_num2id.p, num2id, (size_t)n_num * 4, hipMemcpyHostToDevice, c->stream));
    char pre[64] = {0};
    if (prefix_n) memcpy(pre, prefix, (size_t)prefix_n);
    HIPCHK(hipMemcpyAsync(d_prefix.p, pre, 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_flag.p, 0, 64, c->stream));
    if (tm.on) { HIPCHK(hipStreamSynchronize(c->stream)); tm.lap("tables to the device"); }
    // (the tail of the last tile of every walk, and the bytes behind the buffer's end that a name's look-ahead may touch, hold
    //  whatever the allocation held: the kernels never take a byte at or behind a walk's end for a step or a digit)
    const unsigned nb = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_tiles, 256 * 32));
    if (n_tiles > 0) {
        hipLaunchKernelGGL(wt_tab_kernel, dim3(nb), dim3(256), 0, c->stream, W.d_text.as<uint8_t>(), d_tile0.as<int64_t>(), n_walks, n_tiles, d_end.as<unsigned long long>());
        hipLaunchKernelGGL(wt_count_kernel, dim3(nb), dim3(256), 0, c->stream, W.d_text.as<uint8_t>(), d_tile0.as<int64_t>(), n_walks, n_tiles,
                           d_end.as<unsigned long long>(), d_cnt.as<int32_t>(), d_flag.as<uint32_t>());
    }
    HIPCHK(hipMemsetAsync(d_cnt.as<int32_t>() + n_tiles, 0, 4, c->stream));
    PHICHK(phi_scan_counts_wide(c, d_cnt.as<int32_t>(), n_tiles + 1, d_off.as<int64_t>()));
    hipLaunchKernelGGL(wt_walk_off_kernel, dim3((unsigned)(n_walks / 256 + 1)), dim3(256), 0, c->stream, d_tile0.as<int64_t>(), d_off.as<int64_t>(), n_walks, d_woff.as<int64_t>());
    HIPCHK(hipMemcpyAsync(walk_off_out, d_woff.p, ((size_t)n_walks + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int64_t n_entries = walk_off_out[n_walks];
    tm.lap("tabs, steps per tile, scan");
    if (!flag && n_entries > PHI_MAX_ENTRIES) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 2^32 - 64 walk entries");
    if (!flag) {
        PHICHK(phi_dev_ensure(c, c->d_walk_vtx, (size_t)std::max<int64_t>(n_entries, 1) * 4));
        tm.lap("room for the entries");
        if (n_tiles > 0) {
            WtParseArgs A{W.d_text.as<uint8_t>(), d_tile0.as<int64_t>(), n_walks, n_tiles, d_end.as<unsigned long long>(), d_off.as<int64_t>(),
                          d_prefix.as<uint8_t>(), prefix_n, d_num2id.as<int32_t>(), n_num, n_seg, c->d_walk_vtx.as<int32_t>(), d_flag.as<uint32_t>()};
            hipLaunchKernelGGL(wt_parse_kernel, dim3(nb), dim3(256), 0, c->stream, A);
        }
        // the first and the last vertex of every walk, for phi_set_graph's host pass
        DevBuf d_ends;
        struct G2 { DevBuf &b; ~G2() { if (b.p) (void)hipFree(b.p); } } g2{d_ends};
        PHICHK(phi_dev_ensure(c, d_ends, (size_t)std::max(n_walks, 1) * 8));
        hipLaunchKernelGGL(wt_ends_kernel, dim3((unsigned)(n_walks / 256 + 1)), dim3(256), 0, c->stream, c->d_walk_vtx.as<int32_t>(), d_woff.as<int64_t>(), n_walks, d_ends.as<int32_t>());
        W.ends.assign((size_t)n_walks * 2, -1);
        if (n_walks) HIPCHK(hipMemcpyAsync(W.ends.data(), d_ends.p, (size_t)n_walks * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&flag, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipGetLastError());
        tm.lap("names -> vertices");
    }
    // The text has served.  Its buffer stays with the context (the next upload takes it again; phi_walk_text_upload(ctx, NULL, 0),
    // PHI_WALK_TEXT_FREE=1 or the context's end let it go): the driver hands freed device memory out again only once it has
    // cleared it, and with these 10 GB freed here phi_solve's large allocations waited 0.25 s at config 5 (solve 0.50 -> 0.75 s,
    // four runs of each: profiles/README.md r04i).
    if (getenv("PHI_WALK_TEXT_FREE") && W.d_text.p) { (void)hipFree(W.d_text.p); W.d_text = DevBuf{}; }
    W.ready = false;
    tm.lap("text let go");
    *irregular = flag;
    if (flag) return PHI_OK;
    c->walks_on_device = true;
    c->walks_on_device_n = n_entries;
    return PHI_OK;
}

// (tests) the walk entries phi_walk_text_resolve left on the device
int phi_walk_entries(phi_ctx *c, int32_t *out, int64_t cap, int64_t *n)
{
    if (!c || !n) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    *n = c->walks_on_device ? c->walks_on_device_n : c->n_entries;
    if (!out || cap < *n || *n == 0) return PHI_OK;
    HIPCHK(phi_copy_sync(c, out, c->d_walk_vtx.p, (size_t)*n * 4, hipMemcpyDeviceToHost));
    return PHI_OK;
}

}  // extern "C"

__global__ void phi_warm_walk_text_kernel() {}
void phi_warm_walk_text(hipStream_t st) { hipLaunchKernelGGL(phi_warm_walk_text_kernel, dim3(1), dim3(64), 0, st); }
