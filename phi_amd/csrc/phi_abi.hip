// phi_abi.hip -- C ABI of include/phi_amd.h: context, graph index build, read batches.
// Host-side orchestration only; all per-base work runs in the kernels of sketch.hip, table.hip,
// anchors.hip and dp.hip.  There is no CPU fallback: without a HIP device every entry point
// that needs one returns PHI_ERR_DEVICE.
#include <chrono>
#include <atomic>
#include <memory>
#include <deque>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <future>
#include "phi_ctx.h"
#include "phi_dev.h"

// scalar slots in d_scalars (8 bytes each)
enum { S_ERR = 0, S_NBAD = 1, S_BATCHBAD = 2, S_NEMIT = 3, S_FILTERED = 4, S_INMODEL = 5, S_EXPORT = 6, S_BATCHBAD2 = 7,
       S_OVCNT = 8 /* .. 10: three rotating counters of the overflow list, generation g uses g % 3 (phi_ctx.h) */, S_N = 11 };

int phi_fail(phi_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);            // phi_set_graph runs a GPU thread beside the caller's: both may fail
        c->last_error = buf;
    }
    return code;
}

int phi_hip_check(phi_ctx *c, hipError_t e, const char *what)
{
    if (e == hipSuccess) return PHI_OK;
    const int code = (e == hipErrorOutOfMemory) ? PHI_ERR_NOMEM : PHI_ERR_DEVICE;
    return phi_fail(c, code, "%s: %s", what, hipGetErrorString(e));
}

#define HIPCHK(call) do { int rc_ = phi_hip_check(c, (call), #call); if (rc_) return rc_; } while (0)
#define PHICHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// Large device buffers that are let go are kept for the next taker of about their size instead of going back to the driver:
// the driver clears device memory before it hands it out again (measured on this pool: ~28 us per MB once the freed memory is
// what the next hipMalloc gets -- 60 ms for the 2.1 GB of a DP run's prefix sums at config 5, 0.25 s of phi_solve when the 10 GB
// of walk text were freed before it, and every allocation of a process that starts right behind another one's end), and
// phi_set_graph's temporaries are the sizes phi_solve asks for next (4 bytes per walk entry, several times).  The pool is per
// device, emptied when a solve is done, when a context goes, and when an allocation fails.  PHI_DEVICE_POOL=0: off.
namespace {
struct DevPool { std::mutex mu; std::vector<DevBuf> bufs; };
DevPool g_pool[64];
// (PHI_DEVICE_POOL_MIN=bytes: tests pool everything, so that every buffer comes back with an earlier owner's contents)
const size_t POOL_MIN = getenv("PHI_DEVICE_POOL_MIN") ? (size_t)atoll(getenv("PHI_DEVICE_POOL_MIN")) : ((size_t)16 << 20);
thread_local bool t_pool_bypass = false;
bool pool_on()
{
    static const bool on = !(getenv("PHI_DEVICE_POOL") && atoi(getenv("PHI_DEVICE_POOL")) == 0);
    return on && !t_pool_bypass;
}
int cur_device() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : -1; }
bool pool_take(int dev, size_t want, DevBuf &b)
{
    if (dev < 0) return false;
    DevPool &P = g_pool[dev];
    std::lock_guard<std::mutex> lk(P.mu);
    int best = -1;
    for (int i = 0; i < (int)P.bufs.size(); i++)
        if (P.bufs[(size_t)i].cap >= want && P.bufs[(size_t)i].cap <= want + want / 8 && (best < 0 || P.bufs[(size_t)i].cap < P.bufs[(size_t)best].cap)) best = i;
    if (best < 0) return false;
    b = P.bufs[(size_t)best];
    P.bufs.erase(P.bufs.begin() + best);
    return true;
}
}  // namespace
void phi_pool_flush(int dev)
{
    if (dev < 0 || dev >= 64) return;
    std::vector<DevBuf> v;
    { std::lock_guard<std::mutex> lk(g_pool[dev].mu); v.swap(g_pool[dev].bufs); }
    for (DevBuf &x : v) if (x.p) (void)hipFree(x.p);
}
static void dev_free(DevBuf &b)
{
    if (!b.p) { b.cap = 0; return; }
    int dev = -1;
    if (b.cap >= POOL_MIN && pool_on()) {
        // (the device the buffer lives on -- not the calling thread's current one: a thread that has not chosen a device is on device 0)
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, b.p) == hipSuccess && at.device >= 0 && at.device < 64 && at.device == cur_device()) dev = at.device;
        else (void)hipGetLastError();
    }
    if (dev >= 0) {
        (void)hipDeviceSynchronize();                      // (what hipFree does before it lets memory go: nobody still works on it)
        std::lock_guard<std::mutex> lk(g_pool[dev].mu);
        g_pool[dev].bufs.push_back(b);
    } else {
        (void)hipFree(b.p);
    }
    b.p = nullptr; b.cap = 0;
}

int phi_dev_ensure(phi_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return PHI_OK;
    dev_free(b);
    size_t want = bytes < 256 ? 256 : bytes;
    if (want >= POOL_MIN && pool_on() && pool_take(c->device, want, b)) return PHI_OK;
    static const bool timing = getenv("PHI_TIMING_ALLOC") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&b.p, want);
    if (e == hipErrorOutOfMemory) {                        // what the pool holds may be what is missing
        (void)hipGetLastError();
        phi_pool_flush(c->device);
        e = hipMalloc(&b.p, want);
    }
    if (timing) {
        static std::atomic<long long> total_ns{0}, calls{0};
        const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        total_ns += ns; calls++;
        fprintf(stderr, "[phi alloc] %10zu bytes %8.1f us (total %lld calls, %.3f ms)\n", want, ns / 1e3, (long long)calls, total_ns / 1e6);
    }
    if (e != hipSuccess) { b.p = nullptr; return phi_fail(c, PHI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e)); }
    b.cap = want;
    return PHI_OK;
}

// (the pinned staging buffers serve phi_set_graph's uploads only: given back when it is done)
static void stage_release(phi_ctx *c)
{
    for (int i = 0; i < 2; i++) {
        if (c->stage_ev[i]) { (void)hipEventSynchronize(c->stage_ev[i]); (void)hipEventDestroy(c->stage_ev[i]); }
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        c->h_stage[i] = nullptr; c->stage_ev[i] = nullptr;
    }
}

// A large array from the caller's PAGEABLE memory (the walk entries of a chromosome-scale graph: 5 GB): the runtime stages such
// a copy through its own pinned buffers with one thread, 7 GB/s -- most of phi_set_graph at that size.  Here: two pinned
// buffers of the context, filled by four threads, each sent while the other is filled.
static int upload_staged(phi_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t st)
{
    constexpr size_t PIECE = (size_t)64 << 20;
    if (!c->h_stage[0]) {
        // both buffers and both events, or none: a context never holds half of them
        void *b[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipError_t e = hipSuccess;
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipHostMalloc(&b[i], PIECE, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            for (int i = 0; i < 2; i++) { if (b[i]) (void)hipHostFree(b[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
            return phi_hip_check(c, e, "pinned staging buffers");
        }
        for (int i = 0; i < 2; i++) { c->h_stage[i] = b[i]; c->stage_ev[i] = ev[i]; }
    }
    int k = 0;
    for (size_t off = 0; off < bytes; off += PIECE, k ^= 1) {
        const size_t n = std::min(PIECE, bytes - off);
        HIPCHK(hipEventSynchronize(c->stage_ev[k]));          // (the copy that last read this buffer; a fresh event is complete)
        {
            const int nt = 4;
            std::vector<std::thread> th;
            char *d = static_cast<char *>(c->h_stage[k]);
            const char *s = static_cast<const char *>(src) + off;
            for (int t = 1; t < nt; t++) th.emplace_back([=]() { memcpy(d + n * t / nt, s + n * t / nt, n * (t + 1) / nt - n * t / nt); });
            memcpy(d, s, n / nt);
            for (auto &x : th) x.join();
        }
        HIPCHK(hipMemcpyAsync(static_cast<char *>(dst) + off, c->h_stage[k], n, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(c->stage_ev[k], st));
    }
    return PHI_OK;
}

template <class T> static int upload(phi_ctx *c, DevBuf &b, const T *src, size_t n, hipStream_t st = nullptr)
{
    PHICHK(phi_dev_ensure(c, b, (n ? n : 1) * sizeof(T)));
    if (!st) st = c->stream;
    if (n * sizeof(T) >= ((size_t)256 << 20) && !getenv("PHI_NO_STAGED_UPLOAD")) {
        hipPointerAttribute_t at{};
        const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();                               // (an unregistered pointer is an error of that call, not of ours)
        if (!pinned) return upload_staged(c, b.p, src, n * sizeof(T), st);
    }
    if (n) HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
    return PHI_OK;
}

static uint64_t *scalar(phi_ctx *c, int i) { return c->d_scalars.as<uint64_t>() + i; }
static unsigned long long *logged_stripes(phi_ctx *c) { return c->d_stripes.as<unsigned long long>(); }
static unsigned long long *emit_stripes(phi_ctx *c) { return c->d_stripes.as<unsigned long long>() + PHI_STRIPES * 8; }
#define STRIPE_BYTES ((size_t)PHI_STRIPES * 8 * 8)

// off[0..n] = exclusive prefix sums of cnt[0..n): one workgroup for a few thousand items, the three-phase
// scan beyond (the per-chunk counts of 250 Mbases of walks are half a million items: 1 ms in one workgroup)
int phi_scan_counts_wide(phi_ctx *c, const int32_t *cnt, int64_t n, int64_t *off)
{
    if (n <= 8192) { phi_launch_scan_counts(c->stream, cnt, n, off); return PHI_OK; }
    const int64_t nb = phi_scan_i32_num_blocks(n);
    PHICHK(phi_dev_ensure(c, c->d_scan_blk64, (size_t)nb * 8));
    PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
    phi_launch_scan_i64(c->stream, cnt, n, off, c->d_scan_blk64.as<int64_t>(), c->d_scan_blkoff.as<int64_t>());
    return PHI_OK;
}

// Called by everything that observes the read state.  A reset leaves nothing pending (phi_reset_reads swaps the context's
// double buffers, see phi_ctx.h); in a group of processes the last gather of the hit vectors runs on a stream of its own
// (phi_ipc.hip): the context's stream waits for it here.
int phi_flush_reset(phi_ctx *c) { return c && c->ipc ? phi_ipc_wait_pending(c) : PHI_OK; }

static void swap_read_bufs(phi_ctx *c)
{
    if (c->hit_n == PHI_HIT_RING) {
        // a group of processes: the ended read set's vector stays readable for the peers (it is zeroed two read sets later)
        const DevBuf old = c->d_hit;
        c->d_hit = c->alt.hit; c->alt.hit = c->hit_extra[0]; c->hit_extra[0] = c->hit_extra[1]; c->hit_extra[1] = old;
        c->hit_idx = (c->hit_idx + 1) % PHI_HIT_RING;
    } else {
        std::swap(c->d_hit, c->alt.hit);
    }
    std::swap(c->d_stripes, c->alt.stripes);
}

static int sum_stripes(phi_ctx *c, const void *d, int n_sets, uint64_t *out)
{
    std::vector<uint64_t> h((size_t)n_sets * PHI_STRIPES * 8);
    int rc = phi_hip_check(c, phi_copy_sync(c, h.data(), d, (size_t)n_sets * STRIPE_BYTES, hipMemcpyDeviceToHost), "D2H counters");
    if (rc) return rc;
    for (int s_ = 0; s_ < n_sets; s_++) {
        uint64_t a = 0;
        for (int i = 0; i < PHI_STRIPES; i++) a += h[((size_t)s_ * PHI_STRIPES + i) * 8];
        out[s_] = a;
    }
    return PHI_OK;
}

int phi_read_counts(phi_ctx *c, uint64_t *n_logged, uint64_t *n_emitted)
{
    int frc = phi_flush_reset(c);
    if (frc) return frc;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return phi_fail(c, PHI_ERR_DEVICE, "stream synchronize failed");
    uint64_t v[2];
    int rc = sum_stripes(c, c->d_stripes.p, 2, v);
    if (rc) return rc;
    if (n_logged) *n_logged = v[0];
    if (n_emitted) *n_emitted = v[1];
    return PHI_OK;
}

static uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }

// a device buffer that grows and KEEPS its first `keep` bytes
static int dev_grow_keep(phi_ctx *c, DevBuf &b, size_t bytes, size_t keep)
{
    if (bytes <= b.cap && b.p) return PHI_OK;
    DevBuf nb;
    int rc = phi_dev_ensure(c, nb, bytes);
    if (rc) return rc;
    if (b.p && keep) {
        rc = phi_hip_check(c, phi_copy_sync(c, nb.p, b.p, std::min(keep, b.cap), hipMemcpyDeviceToDevice), "copy into the grown buffer");
        if (rc) { dev_free(nb); return rc; }
    } else if (b.p) {
        rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "stream synchronize");      // an earlier launch may still write the old buffer
        if (rc) { dev_free(nb); return rc; }
    }
    dev_free(b);
    b = nb;
    return PHI_OK;
}

static int sp_set_size(phi_ctx *c, uint64_t *n)
{
    *n = 0;
    if (c->sp_cap == 0 || c->sp_set_gen != c->sp_gen) return PHI_OK;
    return sum_stripes(c, c->d_sp_cnt.p, 1, n);
}

static int sp_clear(phi_ctx *c, uint64_t cap)
{
    if (cap != c->sp_cap || !c->d_sp_keys.p) {
        dev_free(c->d_sp_keys);
        c->sp_cap = 0;
        PHICHK(phi_dev_ensure(c, c->d_sp_keys, cap * 8));
        c->sp_cap = cap;
    }
    PHICHK(phi_dev_ensure(c, c->d_sp_cnt, STRIPE_BYTES));
    phi_launch_fill_u64(c->stream, c->d_sp_keys.as<uint64_t>(), (int64_t)cap, PHI_EMPTY_KEY);
    HIPCHK(hipMemsetAsync(c->d_sp_cnt.p, 0, STRIPE_BYTES, c->stream));
    return PHI_OK;
}

// room in the set for `more` further keys (load factor <= 0.5): a set that has to grow is listed, emptied at its new
// size and filled again
static int sp_reserve(phi_ctx *c, uint64_t in_set, uint64_t more)
{
    const uint64_t need = pow2_at_least(std::max<uint64_t>(1u << 16, 2 * (in_set + more)));
    if (c->sp_set_gen != c->sp_gen) {                         // the set of another generation of reads: nothing of it is kept
        PHICHK(sp_clear(c, (c->sp_cap >= need && c->sp_cap <= 4 * need) ? c->sp_cap : need));
        c->sp_set_gen = c->sp_gen;
        return PHI_OK;
    }
    if (need <= c->sp_cap) return PHI_OK;
    PHICHK(phi_dev_ensure(c, c->d_export, (size_t)std::max<uint64_t>(in_set, 1) * 8));
    HIPCHK(hipMemsetAsync(scalar(c, S_EXPORT), 0, 8, c->stream));
    phi_launch_spectrum_export(c->stream, c->d_sp_keys.as<uint64_t>(), (int64_t)c->sp_cap, c->d_export.as<uint64_t>(),
                               (unsigned long long *)scalar(c, S_EXPORT));
    HIPCHK(hipStreamSynchronize(c->stream));
    PHICHK(sp_clear(c, need));
    phi_launch_spectrum_insert(c->stream, c->d_export.as<uint64_t>(), (int64_t)in_set, c->d_sp_keys.as<uint64_t>(), c->sp_cap - 1,
                               c->d_sp_cnt.as<unsigned long long>(), nullptr, 0, nullptr, nullptr, (uint32_t *)scalar(c, S_ERR));
    return PHI_OK;
}

// The set made current: every novel hash this generation of reads has logged so far is entered (the log's chunks since
// the last flush, the overflow list's entries since then).  Waits for the stream.  *n_in_set (optional) = the set's size.
int phi_sp_flush(phi_ctx *c, uint64_t *n_in_set)
{
    PHICHK(phi_sync_check(c));                                // (also: an overflow list that ran full under an unwaited batch)
    c->async_batches = false;
    uint64_t logged = 0, in_set = 0;
    PHICHK(phi_read_counts(c, &logged, nullptr));
    unsigned long long ov_now = 0;
    HIPCHK(phi_copy_sync(c, &ov_now, scalar(c, S_OVCNT + (int)(c->sp_gen % 3)), 8, hipMemcpyDeviceToHost));
    if ((int64_t)ov_now > c->ov_cap) ov_now = (unsigned long long)c->ov_cap;
    const bool pending = c->log_chunks > c->log_done || (int64_t)ov_now > c->ov_done;
    PHICHK(sp_set_size(c, &in_set));
    if (pending) {
        const uint64_t more = logged > (uint64_t)c->logged_done ? logged - (uint64_t)c->logged_done : 0;
        PHICHK(sp_reserve(c, in_set, more));
        if (c->log_chunks > c->log_done)
            phi_launch_spectrum_flush(c->stream, c->d_novlog.as<uint64_t>(), c->d_novcnt.as<uint16_t>(), c->log_done, c->log_chunks, c->nov_shift,
                                      c->d_sp_keys.as<uint64_t>(), c->sp_cap - 1, c->d_sp_cnt.as<unsigned long long>(), (uint32_t *)scalar(c, S_ERR));
        if ((int64_t)ov_now > c->ov_done)
            phi_launch_spectrum_insert(c->stream, c->d_ovlist.as<uint64_t>() + c->ov_done, (int64_t)ov_now - c->ov_done, c->d_sp_keys.as<uint64_t>(),
                                       c->sp_cap - 1, c->d_sp_cnt.as<unsigned long long>(), nullptr, 0, nullptr, nullptr, (uint32_t *)scalar(c, S_ERR));
        HIPCHK(hipGetLastError());
        c->log_done = c->log_chunks; c->ov_done = (int64_t)ov_now; c->logged_done = (int64_t)logged;
        PHICHK(phi_sync_check(c));
        PHICHK(sp_set_size(c, &in_set));
    }
    if (n_in_set) *n_in_set = in_set;
    return PHI_OK;
}

// |Sp_R| = hit flags set (read hashes that are walk minimisers) + size of the set of the others
int phi_spectrum_count(phi_ctx *c, uint64_t *n_distinct)
{
    uint64_t in_set = 0, flagged = 0;
    PHICHK(phi_sp_flush(c, &in_set));
    if (c->n_unique > 0) {
        HIPCHK(hipMemsetAsync(scalar(c, S_EXPORT), 0, 8, c->stream));
        phi_launch_count_flags(c->stream, c->d_hit.as<uint8_t>(), c->n_unique, (unsigned long long *)scalar(c, S_EXPORT));
        HIPCHK(hipMemcpyAsync(&flagged, scalar(c, S_EXPORT), 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *n_distinct = in_set + flagged;
    return PHI_OK;
}

int phi_pin_ensure(phi_ctx *c, size_t bytes)
{
    if (c->pin_future.valid()) c->pin_future.wait();          // an allocation started by phi_set_graph
    if (bytes <= c->h_pin_cap) return PHI_OK;
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr; c->h_pin_cap = 0;
    c->h_kept = PhiAnchorSpan{}; c->h_dp = PhiAnchorSpan{}; c->anchors_host = false;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&c->h_pin, want, hipHostMallocDefault) != hipSuccess) { c->h_pin = nullptr; return phi_fail(c, PHI_ERR_NOMEM, "pinned host allocation of %zu bytes failed", want); }
    c->h_pin_cap = want;
    return PHI_OK;
}

// wait for the stream and translate the device error word
int phi_sync_check(phi_ctx *c)
{
    PHICHK(phi_flush_reset(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t s[S_N];
    HIPCHK(phi_copy_sync(c, s, c->d_scalars.p, sizeof s, hipMemcpyDeviceToHost));
    const uint32_t err = (uint32_t)s[S_ERR];
    if (err & PHI_KERR_TABLE_FULL) return phi_fail(c, PHI_ERR_OVERFLOW, "open-addressed table overflow (probe bound %d), or the overflow list of novel read hashes ran full under a batch handed over with phi_add_reads_device", PHI_MAX_PROBE);
    if (err & PHI_KERR_SENTINEL) return phi_fail(c, PHI_ERR_UNSUPPORTED, "a minimiser hashes to UINT64_MAX (table sentinel)");
    return PHI_OK;
}

extern "C" {

const char *phi_strerror(int status)
{
    switch (status) {
    case PHI_OK: return "ok";
    case PHI_ERR_INVALID: return "invalid argument";
    case PHI_ERR_NOMEM: return "out of memory";
    case PHI_ERR_DEVICE: return "HIP device error";
    case PHI_ERR_STATE: return "call order violated";
    case PHI_ERR_UNSUPPORTED: return "unsupported input";
    case PHI_ERR_WALK: return "walk does not follow the graph";
    case PHI_ERR_OVERFLOW: return "internal table overflow";
    default: return "unknown status";
    }
}

const char *phi_last_error(const phi_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int phi_ctx_create(int device_id, phi_ctx **out)
{
    if (!out) return PHI_ERR_INVALID;
    *out = nullptr;
    PhiStageTimer tm("ctx_create");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return PHI_ERR_DEVICE;
    tm.lap("runtime start (hipInit)");
    if (device_id < 0 || device_id >= n_dev) return PHI_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return PHI_ERR_DEVICE;
    phi_ctx *c = new (std::nothrow) phi_ctx();
    if (!c) return PHI_ERR_NOMEM;
    c->device = device_id;
    // What a context costs is the runtime's one-time work, all of it latency: a stream is a hardware queue (~20 ms each),
    // the kernels' code objects load at the first launch of their translation unit (~13 ms), and the first host copies of
    // every kind -- large through the copy engines' staging buffers, small through the runtime's own copy kernels -- set up
    // what they go through (~20 ms).  Two threads: this one makes the context's stream and loads the code objects, the other
    // makes the second stream (the host thread's copies inside phi_set_graph) and warms the copy paths on it.  Nothing
    // touches the null stream, whose queue would cost as much again (phi_copy_sync / phi_memset_sync in phi_ctx.h).
    std::future<hipError_t> aux = std::async(std::launch::async, [c, device_id]() {
        hipError_t e = hipSetDevice(device_id);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        void *d = nullptr;
        std::vector<char> h((size_t)1 << 20, 0);
        if (hipMalloc(&d, h.size()) == hipSuccess) {
            (void)hipMemcpyAsync(d, h.data(), h.size(), hipMemcpyHostToDevice, c->aux_stream);
            (void)hipMemcpyAsync(h.data(), d, h.size(), hipMemcpyDeviceToHost, c->aux_stream);
            (void)hipMemcpyAsync(d, h.data(), 64, hipMemcpyHostToDevice, c->aux_stream);
            (void)hipMemcpyAsync(h.data(), d, 64, hipMemcpyDeviceToHost, c->aux_stream);
            (void)hipMemsetAsync(d, 0, 4096, c->aux_stream);
            (void)hipStreamSynchronize(c->aux_stream);
            (void)hipFree(d);
        }
        return hipSuccess;
    });
    auto bail = [&](int code) {
        if (aux.valid()) (void)aux.get();
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
        dev_free(c->d_scalars); dev_free(c->d_stripes); dev_free(c->alt.stripes);
        delete c;
        return code;
    };
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(PHI_ERR_DEVICE);
    c->stream = c->own_stream;
    tm.lap("stream");
    if (phi_dev_ensure(c, c->d_scalars, S_N * 8) || hipMemsetAsync(c->d_scalars.p, 0, S_N * 8, c->stream) != hipSuccess ||
        phi_dev_ensure(c, c->d_stripes, 2 * STRIPE_BYTES) || hipMemsetAsync(c->d_stripes.p, 0, 2 * STRIPE_BYTES, c->stream) != hipSuccess ||
        phi_dev_ensure(c, c->alt.stripes, 2 * STRIPE_BYTES) || hipMemsetAsync(c->alt.stripes.p, 0, 2 * STRIPE_BYTES, c->stream) != hipSuccess)
        return bail(PHI_ERR_DEVICE);
    // the kernels' code objects are loaded lazily, per translation unit, at their first launch: do that here, once
    if (tm.on && getenv("PHI_TIMING_UNITS")) {
        // (diagnostics: what each unit's code object costs to load)
        (void)hipStreamSynchronize(c->stream); tm.lap("  scalars");
        phi_warm_sketch(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: sketch");
        phi_warm_table(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: table");
        phi_warm_anchors(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: anchors");
        phi_warm_contexts(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: contexts");
        phi_warm_dp(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: dp");
        phi_warm_dp_events(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: dp_events");
        phi_warm_solve_dev(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: solve_dev");
        phi_warm_reads_text(c->stream); (void)hipStreamSynchronize(c->stream); tm.lap("  code object: reads_text");
    }
    phi_warm_sketch(c->stream); phi_warm_table(c->stream); phi_warm_anchors(c->stream); phi_warm_contexts(c->stream);
    phi_warm_dp(c->stream); phi_warm_dp_events(c->stream); phi_warm_solve_dev(c->stream); phi_warm_reads_text(c->stream);
    phi_warm_walk_text(c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(PHI_ERR_DEVICE);
    tm.lap("scalars + code objects");
    if (aux.get() != hipSuccess) { c->aux_stream = nullptr; return bail(PHI_ERR_DEVICE); }
    c->last_error.clear();
    tm.lap("wait for the second stream + first host copies");
    *out = c;
    return PHI_OK;
}

void phi_ctx_destroy(phi_ctx *c)
{
    if (c && c->pin_future.valid()) c->pin_future.wait();
    if (c && c->dp_alloc_future.valid()) (void)c->dp_alloc_future.get();
    if (c && c->h_pin) { (void)hipSetDevice(c->device); (void)hipHostFree(c->h_pin); c->h_pin = nullptr; }
    if (c) { (void)hipSetDevice(c->device); stage_release(c); }
    if (!c) return;
    (void)phi_comm_destroy(c);
    (void)phi_ipc_destroy(c);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->d_anchors, &c->d_cov_all, &c->d_cov_w, &c->d_slots, &c->d_slots2, &c->d_segs, &c->d_ctr, &c->d_vmax, &c->d_lane_walk, &c->d_walk_lane, &c->d_coff, &c->d_blk_ncls, &c->d_rownew, &c->d_blk_bad, &c->d_seg_lo, &c->d_seg_row, &c->d_seg_S, &c->wtext.d_text, &c->d_sel_off, &c->d_sel_tri, &c->d_blk_lo, &c->d_blk_ev, &c->d_blk_S, &c->d_row_out, &c->d_rowend, &c->d_blk_keys, &c->d_blk_carry, &c->d_cov, &c->d_cov2, &c->d_stepdiff, &c->alt.hit, &c->hit_extra[0], &c->hit_extra[1], &c->alt.stripes, &c->d_sp_cnt, &c->d_novlog, &c->d_novcnt, &c->d_ovlist, &c->d_vlen, &c->d_ent_cls, &c->d_cls_rep, &c->d_cls_left, &c->d_cls_mult, &c->d_cls_base, &c->d_cls_rec_off, &c->d_rec_cls, &c->d_rec_rel, &c->d_u_replist, &c->d_adj_off, &c->d_adj, &c->d_topo_rank, &c->d_cnt_edge, &c->d_walk_err, &c->d_sa_cnt, &c->d_sa_cur, &c->d_sa_off, &c->d_sa_idx, &c->d_seq, &c->d_seq_off, &c->d_walk_vtx, &c->d_walk_off, &c->d_topo, &c->d_in_off,
                     &c->d_in_src, &c->d_e_out, &c->d_st_rec, &c->d_st_mask, &c->d_in_packed, &c->d_word, &c->d_wwords, &c->d_wbad,
                     &c->d_wascii, &c->d_wstarts, &c->d_rec_hash, &c->d_rec_pos, &c->d_rec_slot,
                     &c->d_rec_e0, &c->d_rec_e1, &c->d_u_keys, &c->d_u_rep, &c->d_u_uid, &c->d_u_kv, &c->d_in_s, &c->d_last_walk, &c->d_rowdiag, &c->d_wpre, &c->d_hit, &c->d_sp_keys, &c->d_rbases,
                     &c->d_roff, &c->d_roff_made, &c->d_peer_send, &c->d_export, &c->d_scalars, &c->d_stripes, &c->d_blk_cnt,
                     &c->d_blk_off, &c->d_flags, &c->d_flags2, &c->d_list, &c->d_list2, &c->d_list3, &c->d_walk_last, &c->d_m_rec, &c->d_m_group,
                     &c->d_g_keys, &c->d_g_rep, &c->d_g_cnt, &c->d_slot_maxcnt, &c->d_slot_multi, &c->d_a_e1,
                     &c->d_g_off, &c->d_g_span, &c->d_a_weight, &c->d_dmax, &c->d_bstart, &c->d_k_rec, &c->d_k_in, &c->d_cvtx, &c->d_ev_e, &c->d_ev_off, &c->d_ev,
                     &c->d_off_end, &c->d_off_start, &c->d_scan_blk, &c->d_scan_blk64, &c->d_scan_blkoff, &c->d_top,
                     &c->d_ent};
    phi_pool_flush(c->device);
    t_pool_bypass = true;                                  // (a context that goes gives its memory back to the driver)
    struct Bypass { ~Bypass() { t_pool_bypass = false; } } bypass_guard;
    for (DevBuf *b : all) dev_free(*b);
    for (auto &pr : c->prof_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->h_err) (void)hipHostFree(c->h_err);
    {
        auto &T = c->text;
        for (int i = 0; i < 2; i++) { dev_free(T.text[i]); dev_free(T.bases[i]); dev_free(T.roff[i]); }
        dev_free(T.tile_cnt); dev_free(T.ls); dev_free(T.pre); dev_free(T.blk); dev_free(T.sum);
        if (T.h_sum) (void)hipHostFree(T.h_sum);
        if (T.ev_copy) (void)hipEventDestroy(T.ev_copy);
    }
    (void)hipStreamDestroy(c->own_stream);
    (void)hipStreamDestroy(c->aux_stream);
    delete c;
}

int phi_set_stream(phi_ctx *c, void *hip_stream)
{
    if (!c) return PHI_ERR_INVALID;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PHI_OK;
}

int phi_set_params(phi_ctx *c, int32_t k, int32_t w, float threshold, int32_t recombination, uint32_t flags)
{
    if (!c) return PHI_ERR_INVALID;
    if (c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_set_params must precede phi_set_graph");
    if (k < 1 || k > PHI_MAX_K) return phi_fail(c, PHI_ERR_INVALID, "k=%d outside [1,%d]", k, PHI_MAX_K);
    if (w < 1 || w > PHI_MAX_W) return phi_fail(c, PHI_ERR_INVALID, "w=%d outside [1,%d]", w, PHI_MAX_W);
    if (recombination < 0) return phi_fail(c, PHI_ERR_INVALID, "recombination penalty must be >= 0");
    c->k = k; c->w = w; c->threshold = threshold; c->recombination = recombination; c->flags = flags;
    return PHI_OK;
}

// count pass -> scan -> ordered write of the minimiser records of one packed flat sequence
static int sketch_records(phi_ctx *c, const uint64_t *words, const unsigned long long *starts, int64_t n_bases,
                          int32_t k, int32_t w, const uint8_t *ascii_if_bad, DevBuf &out_hash, DevBuf &out_pos,
                          int64_t *n_out)
{
    *n_out = 0;
    const int64_t nb = phi_sketch_num_blocks(n_bases);
    if (nb == 0) return PHI_OK;
    PHICHK(phi_dev_ensure(c, c->d_blk_cnt, (size_t)nb * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_off, (size_t)(nb + 1) * 8));
    PhiSketchArgs A{};
    A.words = words; A.starts = starts; A.n_bases = n_bases; A.k = k; A.w = w;
    // sequences with bases outside ACGT take the exact byte-wise path for every window, so that the
    // ordered write stays in position order
    A.ascii = ascii_if_bad; A.allslow = ascii_if_bad ? 1 : 0; A.badbits = nullptr;
    A.block_cnt = c->d_blk_cnt.as<int32_t>();
    A.err = (uint32_t *)scalar(c, S_ERR);
    if (A.allslow) phi_launch_sketch_bytes(c->stream, PHI_MODE_COUNT, A, nullptr);
    else phi_launch_sketch(c->stream, PHI_MODE_COUNT, A);
    PHICHK(phi_scan_counts_wide(c, c->d_blk_cnt.as<int32_t>(), nb, c->d_blk_off.as<int64_t>()));
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->d_blk_off.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    PHICHK(phi_dev_ensure(c, out_hash, (size_t)std::max<int64_t>(total, 1) * 8));
    PHICHK(phi_dev_ensure(c, out_pos, (size_t)std::max<int64_t>(total, 1) * 8));
    A.block_off = c->d_blk_off.as<int64_t>();
    A.out_hash = out_hash.as<uint64_t>();
    A.out_pos = out_pos.as<int64_t>();
    if (A.allslow) phi_launch_sketch_bytes(c->stream, PHI_MODE_WRITE, A, nullptr);
    else phi_launch_sketch(c->stream, PHI_MODE_WRITE, A);
    HIPCHK(hipGetLastError());
    *n_out = total;
    return PHI_OK;
}


}  // extern "C" (helpers below are C++)

// Classes of walk entries with equal context, their sketch in class space and the class records
// (contexts.hip).  Leaves d_vlen, d_ent_cls, d_cls_*, d_rec_{hash,cls,rel,e0,e1}, n_cls, n_rec,
// h_walk_base / walk_bases.  Runs on the context's stream; called by the GPU thread of phi_set_graph.
static int build_classes(phi_ctx *c, int32_t n_vtx, int32_t n_walks, int64_t n_entries)
{
    PhiStageTimer tg("set_graph");
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ev0, ev1};
    HIPCHK(hipEventRecord(ev0, c->stream));
    PHICHK(phi_dev_ensure(c, c->d_vlen, (size_t)n_vtx * 4));
    phi_launch_vlen(c->stream, c->d_seq_off.as<int64_t>(), n_vtx, c->d_vlen.as<int32_t>());
    // bases of every walk (flat base offset of each walk: the positions phi_walk_minimizers reports are walk-relative)
    PHICHK(phi_dev_ensure(c, c->d_list, (size_t)(n_walks + 1) * 8));
    HIPCHK(hipMemsetAsync(c->d_list.p, 0, (size_t)(n_walks + 1) * 8, c->stream));
    phi_launch_walk_bases(c->stream, c->d_walk_vtx.as<int32_t>(), c->d_vlen.as<int32_t>(), c->d_walk_off.as<int64_t>(), n_walks, n_entries,
                          c->d_list.as<unsigned long long>());
    c->h_walk_base.assign(n_walks + 1, 0);
    HIPCHK(hipMemcpyAsync(c->h_walk_base.data() + 1, c->d_list.p, (size_t)n_walks * 8, hipMemcpyDeviceToHost, c->stream));

    tg.lap("[gpu thread]     events, vlen, walk bases");
    // ---- classes: table of context fingerprints, verified entry by entry
    PHICHK(phi_dev_ensure(c, c->d_ent_cls, (size_t)n_entries * 4));
    PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)n_entries));
    DevBuf t_keys, t_rep, t_mult;
    struct Guard { DevBuf &a, &b, &d; ~Guard() { dev_free(a); dev_free(b); dev_free(d); } } guard{t_keys, t_rep, t_mult};
    PhiClassArgs A{};
    A.walk_vtx = c->d_walk_vtx.as<int32_t>(); A.walk_off = c->d_walk_off.as<int64_t>(); A.n_walks = n_walks; A.n_entries = n_entries;
    A.vlen = c->d_vlen.as<int32_t>(); A.seq = c->d_seq.as<uint8_t>(); A.seq_off = c->d_seq_off.as<int64_t>();
    A.tail_need = c->w + c->k - 2;
    A.ent_slot = c->d_ent_cls.as<uint32_t>();
    A.err = (uint32_t *)scalar(c, S_ERR);
    // a pangenome has a few contexts per vertex; walks that share nothing have one per entry
    const uint64_t cap_max = pow2_at_least(std::max<uint64_t>(1024, 2 * (uint64_t)n_entries));
    uint64_t cap = std::min(cap_max, pow2_at_least(std::max<uint64_t>(1024, 4 * (uint64_t)n_vtx)));
    tg.lap("[gpu thread]     entry buffers");
    for (int attempt = 0;; attempt++) {
        PHICHK(phi_dev_ensure(c, t_keys, cap * 8));
        PHICHK(phi_dev_ensure(c, t_rep, cap * 4));
        PHICHK(phi_dev_ensure(c, t_mult, cap * 4));
        A.t_keys = t_keys.as<uint64_t>(); A.t_rep = t_rep.as<uint32_t>(); A.t_mult = t_mult.as<uint32_t>(); A.t_mask = cap - 1;
        A.seed = 0x13198A2E03707344ull + 0x9E3779B97F4A7C15ull * (uint64_t)attempt;
        phi_launch_fill_u64(c->stream, A.t_keys, (int64_t)cap, PHI_EMPTY_KEY);
        phi_launch_fill_u32(c->stream, A.t_rep, (int64_t)cap, 0xFFFFFFFFu);
        HIPCHK(hipMemsetAsync(A.t_mult, 0, cap * 4, c->stream));
        phi_launch_class_insert(c->stream, A);
        uint32_t err = 0;
        HIPCHK(hipMemcpyAsync(&err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (err & PHI_KERR_TABLE_FULL) {
            if (cap == cap_max) return phi_fail(c, PHI_ERR_OVERFLOW, "walk-context table overflow (internal error)");
            cap = std::min(cap_max, cap * 8);
            err &= ~PHI_KERR_TABLE_FULL;
            HIPCHK(phi_copy_sync(c, scalar(c, S_ERR), &err, 4, hipMemcpyHostToDevice));
            attempt--;
            continue;
        }
        phi_launch_class_verify(c->stream, A, c->d_flags.as<uint8_t>());
        HIPCHK(hipMemcpyAsync(&err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!(err & PHI_KERR_FP_COLLISION)) break;
        if (attempt >= 7) return phi_fail(c, PHI_ERR_DEVICE, "walk-context fingerprints collide under 8 seeds (internal error)");
        err &= ~PHI_KERR_FP_COLLISION;
        HIPCHK(phi_copy_sync(c, scalar(c, S_ERR), &err, 4, hipMemcpyHostToDevice));
    }
    tg.lap("[gpu thread]     table insert + verify");
    for (int32_t h = 0; h < n_walks; h++) c->h_walk_base[h + 1] += c->h_walk_base[h];
    c->walk_bases = c->h_walk_base[n_walks];
    // classes in the order of their representatives (smallest entry): the same on every rank
    PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), n_entries, c->d_cls_rep, &c->n_cls));
    const int64_t nc = c->n_cls;
    tg.lap("[gpu thread]     compact representatives");
    PHICHK(phi_dev_ensure(c, c->d_cls_mult, (size_t)nc * 4));
    PHICHK(phi_dev_ensure(c, c->d_cls_left, (size_t)nc));
    PHICHK(phi_dev_ensure(c, c->d_cls_base, (size_t)(nc + 1) * 8));
    PHICHK(phi_dev_ensure(c, c->d_cls_rec_off, (size_t)(nc + 1) * 4));
    // (the rep slot array t_rep is reused as slot -> class id)
    phi_launch_class_ids(c->stream, c->d_cls_rep.as<phi_ent_t>(), nc, A.ent_slot, n_entries, A.t_mult, A.t_rep, c->d_cls_mult.as<int32_t>(),
                         c->d_ent_cls.as<int32_t>());
    PHICHK(phi_dev_ensure(c, c->d_list3, (size_t)nc * 4));
    phi_launch_class_len(c->stream, A, c->d_cls_rep.as<phi_ent_t>(), nc, c->d_list3.as<int32_t>(), c->d_cls_left.as<uint8_t>());
    {
        const int64_t nb = phi_scan_i32_num_blocks(nc);
        PHICHK(phi_dev_ensure(c, c->d_scan_blk64, (size_t)nb * 8));
        PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
        phi_launch_scan_i64(c->stream, c->d_list3.as<int32_t>(), nc, c->d_cls_base.as<int64_t>(), c->d_scan_blk64.as<int64_t>(),
                            c->d_scan_blkoff.as<int64_t>());
    }
    int64_t run = 0;
    HIPCHK(hipMemcpyAsync(&run, c->d_cls_base.as<int64_t>() + nc, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->cls_bases = run;
    if (tg.on) fprintf(stderr, "[phi timing] set_graph: %lld entries in %lld classes, %lld bases of class space for %lld bases of walks\\n",
                       (long long)n_entries, (long long)nc, (long long)run, (long long)c->walk_bases);
    tg.lap("[gpu thread]   classes");

    // ---- class space: packed bases, start bitmap, sketch
    const int64_t n_words = (run + 31) / 32;
    PHICHK(phi_dev_ensure(c, c->d_wwords, (size_t)(n_words + 2) * 8));
    PHICHK(phi_dev_ensure(c, c->d_wbad, (size_t)(n_words + 6) * 4));
    auto pack = [&](uint8_t *ascii) {
        phi_launch_pack_classes(c->stream, c->d_seq.as<uint8_t>(), c->d_seq_off.as<int64_t>(), c->d_walk_vtx.as<int32_t>(), c->d_vlen.as<int32_t>(),
                                c->d_cls_rep.as<phi_ent_t>(), c->d_cls_left.as<uint8_t>(), c->d_cls_base.as<int64_t>(), nc,
                                c->d_wwords.as<uint64_t>(), n_words, c->d_wbad.as<uint32_t>(), ascii, (unsigned long long *)scalar(c, S_NBAD));
    };
    pack(nullptr);
    // bases outside ACGTacgt in the graph: keep a flat ASCII copy of class space for the byte-wise path
    const uint8_t *cls_ascii = nullptr;
    {
        uint64_t n_bad = 0;
        HIPCHK(hipMemcpyAsync(&n_bad, scalar(c, S_NBAD), 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (n_bad || c->k > PHI_MAX_K_PACKED) {                 // (k > 32: the byte-wise path for every window)
            PHICHK(phi_dev_ensure(c, c->d_wascii, (size_t)run + 64));
            HIPCHK(hipMemsetAsync(scalar(c, S_NBAD), 0, 8, c->stream));
            pack(c->d_wascii.as<uint8_t>());
            cls_ascii = c->d_wascii.as<uint8_t>();
        }
    }
    const size_t n_sw = (size_t)(run / 64 + 2);
    PHICHK(phi_dev_ensure(c, c->d_wstarts, n_sw * 8));
    HIPCHK(hipMemsetAsync(c->d_wstarts.p, 0, n_sw * 8, c->stream));
    phi_launch_mark_starts(c->stream, c->d_cls_base.as<int64_t>(), nc, c->d_wstarts.as<unsigned long long>());
    int64_t n_raw = 0;
    DevBuf raw_hash;
    struct Guard1 { DevBuf &a; ~Guard1() { dev_free(a); } } guard1{raw_hash};
    PHICHK(sketch_records(c, c->d_wwords.as<uint64_t>(), c->d_wstarts.as<unsigned long long>(), run, c->k, c->w, cls_ascii, raw_hash,
                          c->d_rec_pos, &n_raw));
    if (n_raw >= (int64_t)1 << 31) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 2^31 minimisers in the distinct walk contexts");
    tg.lap("[gpu thread]   class-space pack + sketch");

    // ---- raw records -> class records (the left base's own window dropped)
    const int64_t nr0 = std::max<int64_t>(n_raw, 1);
    DevBuf r_cls, r_rel, r_e0, r_e1;
    struct Guard4 { DevBuf &a, &b, &d, &e; ~Guard4() { dev_free(a); dev_free(b); dev_free(d); dev_free(e); } } guard4{r_cls, r_rel, r_e0, r_e1};
    PHICHK(phi_dev_ensure(c, r_cls, (size_t)nr0 * 4));
    PHICHK(phi_dev_ensure(c, r_rel, (size_t)nr0 * 4));
    PHICHK(phi_dev_ensure(c, r_e0, (size_t)nr0 * 4));
    PHICHK(phi_dev_ensure(c, r_e1, (size_t)nr0 * 4));
    PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)std::max<int64_t>(nr0, n_entries)));
    phi_launch_class_rec(c->stream, c->d_rec_pos.as<int64_t>(), n_raw, c->d_cls_base.as<int64_t>(), nc, c->d_cls_rep.as<phi_ent_t>(),
                         c->d_cls_left.as<uint8_t>(), c->d_walk_vtx.as<int32_t>(), c->d_vlen.as<int32_t>(), c->k, c->d_flags.as<uint8_t>(),
                         r_cls.as<int32_t>(), r_rel.as<int32_t>(), r_e0.as<phi_ent_t>(), r_e1.as<phi_ent_t>());
    PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), n_raw, c->d_list2, &c->n_rec));
    const int64_t nr = std::max<int64_t>(c->n_rec, 1);
    PHICHK(phi_dev_ensure(c, c->d_rec_hash, (size_t)nr * 8));
    PHICHK(phi_dev_ensure(c, c->d_rec_cls, (size_t)nr * 4));
    PHICHK(phi_dev_ensure(c, c->d_rec_rel, (size_t)nr * 4));
    PHICHK(phi_dev_ensure(c, c->d_rec_e0, (size_t)nr * 4));
    PHICHK(phi_dev_ensure(c, c->d_rec_e1, (size_t)nr * 4));
    phi_launch_class_rec_gather(c->stream, c->d_list2.as<int32_t>(), c->n_rec, raw_hash.as<uint64_t>(), r_cls.as<int32_t>(), r_rel.as<int32_t>(),
                                r_e0.as<phi_ent_t>(), r_e1.as<phi_ent_t>(), c->d_rec_hash.as<uint64_t>(), c->d_rec_cls.as<int32_t>(),
                                c->d_rec_rel.as<int32_t>(), c->d_rec_e0.as<phi_ent_t>(), c->d_rec_e1.as<phi_ent_t>());
    phi_launch_class_rec_off(c->stream, c->d_rec_cls.as<int32_t>(), c->n_rec, nc, c->d_cls_rec_off.as<int32_t>());
    HIPCHK(hipEventRecord(ev1, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                  // the temporaries above go out of scope
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    c->index_gpu_ms = ms;
    HIPCHK(hipGetLastError());
    return PHI_OK;
}

extern "C" {

int phi_set_graph(phi_ctx *c, int32_t n_vtx, const char *seq_concat, const int64_t *seq_off, const int64_t *adj_off,
                  const int32_t *adj, int32_t n_walks, const int64_t *walk_off, const int32_t *walk_vtx,
                  const int32_t *topo_rank)
{
    if (!c) return PHI_ERR_INVALID;
    // walk_vtx == NULL: the entries are on the device already, resolved there from the W-lines' text (phi_walk_text_resolve)
    const bool dev_walks = walk_vtx == nullptr;
    if (dev_walks && c && walk_off && n_walks > 0 && !(c->walks_on_device && c->walks_on_device_n == walk_off[n_walks] && (int32_t)(c->wtext.ends.size() / 2) == n_walks))
        return phi_fail(c, PHI_ERR_STATE, "phi_set_graph without walk_vtx: phi_walk_text_resolve must have resolved exactly these walks on this context");
    if (n_vtx <= 0 || n_walks <= 0 || !seq_concat || !seq_off || !adj_off || !walk_off || !topo_rank)
        return phi_fail(c, PHI_ERR_INVALID, "phi_set_graph: null pointer or empty graph");
    if (adj_off[n_vtx] > 0 && !adj) return phi_fail(c, PHI_ERR_INVALID, "phi_set_graph: adj is null");
    HIPCHK(hipSetDevice(c->device));
    if (c->ipc) return phi_fail(c, PHI_ERR_STATE, "phi_set_graph on a context in a group of processes: phi_ipc_destroy first (the peers have this context's hit vectors mapped)");
    c->have_graph = false;
    c->solved = false;

    PhiStageTimer tm("set_graph");
    // the solve downloads its kept anchors (12 bytes each, a fraction of the walk entries) into pinned
    // memory; pinning tens of MB takes 5-30 ms, so it happens on a thread of its own, from now on
    if (c->pin_future.valid()) c->pin_future.wait();
    {
        int64_t ne = walk_off[n_walks];                      // not validated yet: clamp
        ne = ne < 0 ? 0 : (ne > PHI_MAX_ENTRIES ? PHI_MAX_ENTRIES : ne);
        const size_t want = ((size_t)ne / 4 + 4096) * sizeof(PhiAnchorHost);
        // (only for graphs whose solve is likely to take the host copy of the anchors: a model of 2^16 anchors or more
        //  stays on the device, solve_dev.hip, and pinning tens of MB here holds up the other threads' HIP calls)
        if (want > c->h_pin_cap && ne / 4 < ((int64_t)1 << 16) && !getenv("PHI_PREPIN")) {
            c->h_kept = PhiAnchorSpan{}; c->h_dp = PhiAnchorSpan{}; c->anchors_host = false;
            c->pin_future = std::async(std::launch::async, [c, want]() {
                (void)hipSetDevice(c->device);
                if (c->h_pin) (void)hipHostFree(c->h_pin);
                c->h_pin = nullptr; c->h_pin_cap = 0;
                void *p = nullptr;
                if (hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess) { c->h_pin = p; c->h_pin_cap = want; }
            });
        }
    }
    // ---- validate and keep host copies
    if (seq_off[0] != 0 || adj_off[0] != 0 || walk_off[0] != 0) return phi_fail(c, PHI_ERR_INVALID, "offset arrays must start at 0");
    for (int32_t v = 0; v < n_vtx; v++)
        if (seq_off[v + 1] < seq_off[v] || adj_off[v + 1] < adj_off[v]) return phi_fail(c, PHI_ERR_INVALID, "offsets not monotone at vertex %d", v);
    for (int32_t h = 0; h < n_walks; h++)
        if (walk_off[h + 1] <= walk_off[h]) return phi_fail(c, PHI_ERR_INVALID, "walk %d is empty", h);
    const int64_t n_edges = adj_off[n_vtx], n_entries = walk_off[n_walks];
    // (the first and the last vertex of a walk are all the host pass looks at of the walk entries)
    auto walk_first = [&](int32_t h) -> int32_t { return dev_walks ? c->wtext.ends[(size_t)h * 2] : walk_vtx[walk_off[h]]; };
    auto walk_last = [&](int32_t h) -> int32_t { return dev_walks ? c->wtext.ends[(size_t)h * 2 + 1] : walk_vtx[walk_off[h + 1] - 1]; };
    if (n_entries > PHI_MAX_ENTRIES) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 2^32 - 64 walk entries");
    // The DP's per-entry buffers of a chromosome-scale graph (5 x 4-8 bytes per walk entry: 26 GB at 1.3 G entries) are
    // allocated now, on a thread of their own: the driver clears device memory as it hands it out (tens of GB/s), which
    // otherwise shows up as half a second at the start of phi_solve.  Joined before this call returns.
    if (c->dp_alloc_future.valid()) (void)c->dp_alloc_future.get();
    if (n_entries >= ((int64_t)1 << 24) && n_walks <= PHI_DP_EVENT_MAX_WALKS) {
        c->dp_alloc_future = std::async(std::launch::async, [c, n_entries]() -> int {
            if (hipSetDevice(c->device) != hipSuccess) return PHI_ERR_DEVICE;
            const size_t ne = (size_t)n_entries;
            PHICHK(phi_dev_ensure(c, c->d_g_off, (ne + 1) * 8));
            PHICHK(phi_dev_ensure(c, c->d_dmax, ne * 4));
            PHICHK(phi_dev_ensure(c, c->d_bstart, ne * 4));
            PHICHK(phi_dev_ensure(c, c->d_off_end, (ne + 3) * 4));
            PHICHK(phi_dev_ensure(c, c->d_off_start, (ne + 3) * 4));
            return PHI_OK;
        });
    }
    c->n_vtx = n_vtx; c->n_walks = n_walks; c->n_entries = n_entries;
    // The graph arrays go to the device while this thread validates them (copies from the caller's pageable arrays:
    // 52 MB and 3 ms at C2).  Only copies: no kernel indexes with them before the validation below has passed.
    // (joined by the GPU thread, or by the future's destructor on an early return)
    std::future<int> uploads = std::async(std::launch::async, [&]() -> int {
        HIPCHK(hipSetDevice(c->device));
        PHICHK(upload(c, c->d_seq, seq_concat, (size_t)seq_off[n_vtx]));
        PHICHK(upload(c, c->d_seq_off, seq_off, (size_t)n_vtx + 1));
        if (!dev_walks) PHICHK(upload(c, c->d_walk_vtx, walk_vtx, (size_t)n_entries));
        PHICHK(upload(c, c->d_walk_off, walk_off, (size_t)n_walks + 1));
        PHICHK(upload(c, c->d_adj_off, adj_off, (size_t)n_vtx + 1));
        if (n_edges == 0) PHICHK(phi_dev_ensure(c, c->d_adj, 4));
        else PHICHK(upload(c, c->d_adj, adj, (size_t)n_edges));
        PHICHK(upload(c, c->d_topo_rank, topo_rank, (size_t)n_vtx));
        return PHI_OK;
    });

    // topological order from the ranks; every edge must go forward (acyclic GFA, README.md:70-75).  All host threads: at chromosome
    // scale these are 8.4 M + 11 M random accesses that every kernel of the index build waits for.
    c->h_topo.assign(n_vtx, -1);
    std::vector<int64_t> indeg(n_vtx, 0);
    {
        PhiHostError verr;
        phi_parallel_chunks(n_vtx, 1 << 16, [&](int64_t lo, int64_t hi, int) {
            for (int64_t v = lo; v < hi && !verr.failed(); v++) {
                const int32_t r = topo_rank[v];
                int32_t none = -1;
                if (r < 0 || r >= n_vtx || !__atomic_compare_exchange_n(&c->h_topo[(size_t)r], &none, (int32_t)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                    verr.set(PHI_ERR_INVALID, "topo_rank is not a permutation (vertex %d): is the graph cyclic?", (int)v);
                    return;
                }
            }
        });
        if (verr.failed()) return phi_fail(c, verr.code, "%s", verr.msg.c_str());
        phi_parallel_chunks(n_vtx, 1 << 16, [&](int64_t lo, int64_t hi, int) {
            for (int64_t u = lo; u < hi && !verr.failed(); u++)
                for (int64_t x = adj_off[u]; x < adj_off[u + 1]; x++) {
                    const int32_t v = adj[x];
                    if (v < 0 || v >= n_vtx) { verr.set(PHI_ERR_INVALID, "edge target %d out of range", v); return; }
                    if (topo_rank[u] >= topo_rank[v]) { verr.set(PHI_ERR_INVALID, "edge %d->%d goes backwards in topo_rank: graph must be acyclic", (int)u, v); return; }
                    __atomic_fetch_add(&indeg[(size_t)v], 1, __ATOMIC_RELAXED);
                }
        });
        if (verr.failed()) return phi_fail(c, verr.code, "%s", verr.msg.c_str());
    }

    // (the walk entries are range-checked by the first kernel that reads them: phi_walk_edges_kernel, code 4 below)
    tm.lap("validate graph, copies");
    // ---- the GPU side of the index (uploads, entry offsets, walk sketch, minimiser table) runs on its
    //      own host thread while this one makes the pass over the walk entries below: neither needs
    //      the other's results
    if (n_walks > PHI_DP_MAX_WALKS) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than %d walks", PHI_DP_MAX_WALKS);
    c->dp_nw = phi_dp_num_waves(n_walks);
    const int nw64 = c->dp_nw;
    // host copy of the walk entries for the solve (38 MB at C2, 4 ms of page faults): a few threads of their own, joined
    // before this call returns.  NOT for a chromosome-scale graph (5.3 GB at 1.3 G entries, beside the caller's own copy):
    // what the solve looks up there -- a few entries per recombination of the backtrack, the stretches of the decoded path --
    // it reads from the device copy (phi_solve.hip walk_vtx_*); the branch and bound proper fetches the array if it ever starts.
    int64_t host_walks_max = (int64_t)1 << 26;
    if (const char *e = getenv("PHI_HOST_WALKS_MAX")) host_walks_max = atoll(e);                     // tests: no host copy at any size
    const bool keep_host_walks = n_entries <= host_walks_max;
    if (!c->h_walk_vtx.resize(keep_host_walks ? n_entries : 0)) return phi_fail(c, PHI_ERR_NOMEM, "host allocation failed");
    std::future<void> wv_copy = std::async(std::launch::async, [&]() {
        if (!keep_host_walks) return;
        int32_t *dst = c->h_walk_vtx.data();
        if (dev_walks) {                                       // (resolved on the device: the host copy comes from there)
            (void)hipSetDevice(c->device);
            if (n_entries && hipMemcpyAsync(dst, c->d_walk_vtx.p, (size_t)n_entries * 4, hipMemcpyDeviceToHost, c->aux_stream) == hipSuccess)
                (void)hipStreamSynchronize(c->aux_stream);
            return;
        }
        const int nt = 4;
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([=]() {
                const int64_t lo = n_entries * t / nt, hi = n_entries * (t + 1) / nt;
                if (!dev_walks) memcpy(dst + lo, walk_vtx + lo, (size_t)(hi - lo) * 4);
            });
        for (auto &x : th) x.join();
    });
    struct CopyJoiner { std::future<void> &f; ~CopyJoiner() { if (f.valid()) f.wait(); } } copy_joiner{wv_copy};
    std::vector<int32_t> cnt_edge(std::max<int64_t>(n_edges, 1), 0), cont_total(n_vtx, 0);
    // the every-vertex stream of dp.hip: beyond 256 walks, when asked for, and as the fallback of the
    // four-wave event kernel (129-256 walks) whose per-lane queues are shallower than the worst case
    const bool want_masks = !(n_walks <= PHI_DP_EVENT_SAFE_WALKS && !getenv("PHI_DP_DENSE"));
    // the walk-entry pass (out-edge of every entry, walks per edge, walks per vertex) runs on the GPU as soon
    // as the walks are uploaded; its edge counts come back through this promise, the rest stays on the device
    std::promise<int> edges_promise;
    std::future<int> edges_future = edges_promise.get_future();
    int32_t walk_err[4] = {0, 0, 0, 0};
    auto gpu_part = [&]() -> int {
        // whatever happens, the host thread waiting for the edge counts is released
        struct PromiseGuard { std::promise<int> &p; bool done = false; ~PromiseGuard() { if (!done) p.set_value(PHI_ERR_DEVICE); } } pg{edges_promise};
        HIPCHK(hipSetDevice(c->device));
        PhiStageTimer tg("set_graph");
        PHICHK(uploads.get());                                 // (the graph arrays: on their way since before the validation)
        {
            // the walk-entry pass: walks follow edges of forward vertices (ILP_index.cpp:104-107 exits on reverse
            // strand; an edge-less step would leave an anchor's edge variables unconstrained, :799-815)
            int rc = PHI_OK;
            auto pass = [&]() -> int {
                PHICHK(phi_dev_ensure(c, c->d_e_out, (size_t)n_entries));
                PHICHK(phi_dev_ensure(c, c->d_cnt_edge, cnt_edge.size() * 4));
                PHICHK(phi_dev_ensure(c, c->d_walk_err, 16));
                HIPCHK(hipMemsetAsync(c->d_cnt_edge.p, 0, cnt_edge.size() * 4, c->stream));
                HIPCHK(hipMemsetAsync(c->d_walk_err.p, 0, 16, c->stream));
                if (want_masks) {
                    PHICHK(phi_dev_ensure(c, c->d_st_mask, (size_t)n_vtx * nw64 * 8));
                    HIPCHK(hipMemsetAsync(c->d_st_mask.p, 0, (size_t)n_vtx * nw64 * 8, c->stream));
                }
                phi_launch_walk_edges(c->stream, c->d_walk_vtx.as<int32_t>(), c->d_walk_off.as<int64_t>(), n_walks, n_entries, n_vtx,
                                      c->d_adj_off.as<int64_t>(), c->d_adj.as<int32_t>(), c->d_seq_off.as<int64_t>(),
                                      c->d_topo_rank.as<int32_t>(), c->d_e_out.as<uint8_t>(), c->d_cnt_edge.as<int32_t>(),
                                      want_masks ? c->d_st_mask.as<unsigned long long>() : nullptr, nw64, c->d_walk_err.as<int32_t>());
                HIPCHK(hipMemcpyAsync(cnt_edge.data(), c->d_cnt_edge.p, cnt_edge.size() * 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipMemcpyAsync(walk_err, c->d_walk_err.p, 16, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                return PHI_OK;
            };
            rc = pass();
            edges_promise.set_value(rc);
            pg.done = true;
            if (rc) return rc;
            if (tg.on) tg.lap("[gpu thread] uploads + walk-entry pass");
        }
        if (walk_err[0]) return PHI_OK;                        // the main thread reports it; nothing below may index with such walks
        // ---- stage 1a on the GPU (ILP_index.cpp:559-573), de-duplicated: classes of walk entries with equal
        //      context, one sketch per class, the minimiser table from the class records (contexts.hip)
        HIPCHK(hipMemsetAsync(c->d_scalars.p, 0, S_N * 8, c->stream));
        HIPCHK(hipMemsetAsync(c->d_stripes.p, 0, 2 * STRIPE_BYTES, c->stream));
        PHICHK(build_classes(c, n_vtx, n_walks, n_entries));
        c->h_kept = PhiAnchorSpan{}; c->h_dp = PhiAnchorSpan{}; c->anchors_host = false;
        if (tg.on) (void)hipStreamSynchronize(c->stream);
        tg.lap("[gpu thread] classes + class sketch");
        const int64_t nr = std::max<int64_t>(c->n_rec, 1);
        PHICHK(phi_dev_ensure(c, c->d_rec_slot, (size_t)nr * 4));
        // The table is built over the class records (nearly all distinct) at twice their number, then
        // re-inserted at 8x the distinct keys (load ~12 %: read probes settle on the first slot).
        const uint64_t cap_full = pow2_at_least(std::max<uint64_t>(1024, 2 * (uint64_t)c->n_rec));
        const uint64_t UMULT = 8;
        PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)nr));
        {
            c->u_cap = cap_full;
            PHICHK(phi_dev_ensure(c, c->d_u_keys, c->u_cap * 8));
            PHICHK(phi_dev_ensure(c, c->d_u_rep, c->u_cap * 4));
            phi_launch_fill_u64(c->stream, c->d_u_keys.as<uint64_t>(), (int64_t)c->u_cap, PHI_EMPTY_KEY);
            phi_launch_fill_u32(c->stream, c->d_u_rep.as<uint32_t>(), (int64_t)c->u_cap, 0xFFFFFFFFu);
            phi_launch_table_build(c->stream, c->d_rec_hash.as<uint64_t>(), c->n_rec, c->d_u_keys.as<uint64_t>(),
                                   c->d_u_rep.as<uint32_t>(), c->u_cap - 1, c->d_rec_slot.as<uint32_t>(),
                                   (uint32_t *)scalar(c, S_ERR));
            // dense, rank-independent minimiser ids: rank of the first class record of each hash
            phi_launch_rep_flags(c->stream, c->d_rec_slot.as<uint32_t>(), c->n_rec, c->d_u_rep.as<uint32_t>(),
                                 c->d_flags.as<uint8_t>());
            PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), c->n_rec, c->d_u_replist, &c->n_unique));   // waits for the stream
        }
        {
            // wanted capacity: 8x the distinct keys; re-insert them (and look every record up again) when
            // the table is more than a factor two away from it
            const uint64_t want = pow2_at_least(std::max<uint64_t>(1024, UMULT * (uint64_t)c->n_unique));
            if (c->u_cap > 2 * want || 2 * c->u_cap < want) {
                DevBuf keys2, uid2;
                struct Guard { DevBuf &a, &b; ~Guard() { dev_free(a); dev_free(b); } } guard{keys2, uid2};   // error paths below
                PHICHK(phi_dev_ensure(c, keys2, want * 8));
                PHICHK(phi_dev_ensure(c, uid2, want * 4));
                phi_launch_fill_u64(c->stream, keys2.as<uint64_t>(), (int64_t)want, PHI_EMPTY_KEY);
                phi_launch_table_compact(c->stream, c->d_u_replist.as<int32_t>(), c->n_unique, c->d_rec_hash.as<uint64_t>(), c->n_rec,
                                         keys2.as<uint64_t>(), uid2.as<uint32_t>(), want - 1, c->d_rec_slot.as<uint32_t>(),
                                         (uint32_t *)scalar(c, S_ERR));
                HIPCHK(hipStreamSynchronize(c->stream));
                dev_free(c->d_u_keys); dev_free(c->d_u_uid); dev_free(c->d_u_rep);
                c->d_u_keys = keys2; c->d_u_uid = uid2;
                keys2 = DevBuf{}; uid2 = DevBuf{};           // ownership moved
                c->u_cap = want;
            } else {
                PHICHK(phi_dev_ensure(c, c->d_u_uid, c->u_cap * 4));
                phi_launch_slot_uid(c->stream, c->d_u_replist.as<int32_t>(), c->n_unique, c->d_rec_slot.as<uint32_t>(),
                                    c->d_u_uid.as<uint32_t>());
            }
        }
        PHICHK(phi_dev_ensure(c, c->d_u_kv, c->u_cap * 16));
        phi_launch_table_pairs(c->stream, c->d_u_keys.as<uint64_t>(), c->d_u_uid.as<uint32_t>(), (int64_t)c->u_cap,
                               c->d_u_kv.as<uint64_t>());
        // records of each walk ("Number of Minimizers", ILP_index.cpp:563) = sum over its entries of their class's records
        {
            PHICHK(phi_dev_ensure(c, c->d_list2, (size_t)(n_walks + 1) * 8));
            HIPCHK(hipMemsetAsync(c->d_list2.p, 0, (size_t)(n_walks + 1) * 8, c->stream));
            phi_launch_walk_rec_counts(c->stream, c->d_ent_cls.as<int32_t>(), c->d_cls_rec_off.as<int32_t>(), c->d_walk_off.as<int64_t>(),
                                       n_walks, n_entries, c->d_list2.as<unsigned long long>());
            c->h_n_minimizers.assign(n_walks, 0);
            HIPCHK(hipMemcpyAsync(c->h_n_minimizers.data(), c->d_list2.p, (size_t)n_walks * 8, hipMemcpyDeviceToHost, c->stream));
        }
        PHICHK(phi_dev_ensure(c, c->d_hit, (size_t)(c->n_unique / 8 + 1) * 8));
        HIPCHK(hipMemsetAsync(c->d_hit.p, 0, (size_t)(c->n_unique / 8 + 1) * 8, c->stream));
        PHICHK(phi_dev_ensure(c, c->alt.hit, (size_t)(c->n_unique / 8 + 1) * 8));
        HIPCHK(hipMemsetAsync(c->alt.hit.p, 0, (size_t)(c->n_unique / 8 + 1) * 8, c->stream));
        HIPCHK(hipMemsetAsync(c->alt.stripes.p, 0, 2 * STRIPE_BYTES, c->stream));
        HIPCHK(hipGetLastError());
        PHICHK(phi_sync_check(c));
        tg.lap("[gpu thread] walk sketch + table");
        return PHI_OK;
    };
    std::future<int> gpu_future = std::async(std::launch::async, gpu_part);
    // every early return below must first wait for that thread
    struct Joiner { std::future<int> &f; ~Joiner() { if (f.valid()) f.wait(); } } joiner{gpu_future};
    // ---- host copies of the graph, while the GPU thread uploads
    // (each array on a thread of its own: at chromosome scale they are 0.45 GB of first-touched pages, 130 ms one after the other)
    {
        std::thread t1([&]() { c->h_seq.assign(seq_concat, seq_concat + seq_off[n_vtx]); });
        std::thread t2([&]() { c->h_seq_off.assign(seq_off, seq_off + n_vtx + 1); });
        std::thread t3([&]() { c->h_adj_off.assign(adj_off, adj_off + n_vtx + 1); c->h_adj.assign(adj, adj + n_edges); });
        c->h_walk_off.assign(walk_off, walk_off + n_walks + 1);
        c->h_topo_rank.assign(topo_rank, topo_rank + n_vtx);
        t1.join(); t2.join(); t3.join();
    }
    tm.lap("host copies");
    // ---- one parallel pass over the walk entries (host threads over fixed chunks of entries):
    //   * walks follow edges of forward vertices (ILP_index.cpp:104-107 exits on reverse strand; an
    //     edge-less step would make the anchor's edge variables unconstrained, :799-815)
    //   * out-edge index of every entry, walks per edge (DP step stream, dp.hip)
    //   * mask of walks on every topological step, bases of every walk, host copy of the entries
    {
        const int erc = edges_future.get();
        if (erc) return erc;
        if (walk_err[0] == 4) return phi_fail(c, PHI_ERR_WALK, "walk %d holds vertex %d out of range", walk_err[1], walk_err[2]);
        if (walk_err[0] == 1) return phi_fail(c, PHI_ERR_UNSUPPORTED, "walk %d passes through empty segment %d", walk_err[1], walk_err[2]);
        if (walk_err[0] == 2) return phi_fail(c, PHI_ERR_WALK, "walk %d steps %d->%d without a graph edge", walk_err[1], walk_err[2], walk_err[3]);
        if (walk_err[0] == 3) return phi_fail(c, PHI_ERR_UNSUPPORTED, "vertex %d has more than 254 out-edges", walk_err[2]);
    }
    tm.lap("walk entries: pass on the GPU");
    bool start_interior = false, end_interior = false;
    for (int32_t h = 0; h < n_walks; h++) {
        if (indeg[walk_first(h)] > 0) start_interior = true;
        const int32_t last = walk_last(h);
        if (adj_off[last + 1] > adj_off[last]) end_interior = true;
    }
    if (start_interior && end_interior)
        return phi_fail(c, PHI_ERR_UNSUPPORTED, "walks both start and end at interior vertices: the reference model "
                        "admits flow leak/spawn artefacts there (ILP_index.cpp:1330) that are not emulated");
    for (int32_t u = 0; u < n_vtx; u++)
        for (int64_t x = adj_off[u]; x < adj_off[u + 1]; x++) cont_total[u] += cnt_edge[x];

    // ---- DP step stream (dp.hip): per step the live in-edges as (steps back, out-edge index).  All host threads: the records
    //      are 32 bytes per vertex (268 MB at chromosome scale, first touched by whoever writes them), the in-edges of a vertex
    //      are gathered with an atomic cursor and then sorted, so that the stream does not depend on who came first.
    std::unique_ptr<int32_t[]> st_rec_buf(new int32_t[(size_t)n_vtx * 8]);
    int32_t *const st_rec = st_rec_buf.get();
    const size_t st_rec_n = (size_t)n_vtx * 8;
    std::vector<int32_t> in_packed;
    {
        // live in-edges of v: (u, x) with some walk on u continuing along another edge than x
        std::vector<int32_t> live_cnt(n_vtx + 1, 0);
        std::vector<uint8_t> tops(n_vtx, 0);
        PhiHostError perr;
        const int64_t VCH = 1 << 16;
        phi_parallel_chunks(n_vtx, VCH, [&](int64_t lo, int64_t hi, int) {
            for (int64_t u = lo; u < hi; u++)
                for (int64_t x = adj_off[u]; x < adj_off[u + 1]; x++)
                    if (cont_total[u] - cnt_edge[x] > 0) {
                        const int64_t back = (int64_t)topo_rank[adj[x]] - topo_rank[u];
                        if (back >= (1 << 23)) { perr.set(PHI_ERR_UNSUPPORTED, "edge spans more than 2^23 topological steps"); return; }
                        __atomic_fetch_add(&live_cnt[(size_t)adj[x] + 1], 1, __ATOMIC_RELAXED);
                        tops[u] = 1;
                    }
        });
        if (perr.failed()) return phi_fail(c, perr.code, "%s", perr.msg.c_str());
        for (int32_t v = 0; v < n_vtx; v++) live_cnt[v + 1] += live_cnt[v];
        std::vector<int32_t> live(std::max<int32_t>(live_cnt[n_vtx], 1)), cur(live_cnt.begin(), live_cnt.end() - 1);
        phi_parallel_chunks(n_vtx, VCH, [&](int64_t lo, int64_t hi, int) {
            for (int64_t u = lo; u < hi; u++)
                for (int64_t x = adj_off[u]; x < adj_off[u + 1]; x++)
                    if (cont_total[u] - cnt_edge[x] > 0) {
                        const int64_t back = (int64_t)topo_rank[adj[x]] - topo_rank[u];
                        live[(size_t)__atomic_fetch_add(&cur[(size_t)adj[x]], 1, __ATOMIC_RELAXED)] = (int32_t)(back << 8) | (int32_t)(x - adj_off[u]);
                    }
        });
        // the in-edges beyond the third of a step go to in_packed: where, from the counts
        std::vector<int64_t> extra_off((size_t)n_vtx + 1, 0);
        for (int32_t s = 0; s < n_vtx; s++) {
            const int32_t v = c->h_topo[s];
            const int n_in = live_cnt[v + 1] - live_cnt[v];
            if (n_in > 255) return phi_fail(c, PHI_ERR_UNSUPPORTED, "vertex %d has more than 255 in-edges", v);
            extra_off[(size_t)s + 1] = extra_off[(size_t)s] + std::max(0, n_in - 3);
        }
        if (extra_off[(size_t)n_vtx] > INT32_MAX) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 2^31 recombination in-edges");
        in_packed.assign((size_t)extra_off[(size_t)n_vtx], 0);
        phi_parallel_chunks(n_vtx, VCH, [&](int64_t lo, int64_t hi, int) {
            for (int64_t s = lo; s < hi; s++) {
                const int32_t v = c->h_topo[(size_t)s];
                int32_t *r = &st_rec[(size_t)s * 8];
                const int n_in = live_cnt[v + 1] - live_cnt[v];
                int32_t *in = live.data() + live_cnt[v];
                if (n_in > 1) std::sort(in, in + n_in);
                r[0] = (n_in ? PHI_DP_NEED_ENTRY : 0) | (tops[v] ? PHI_DP_NEED_TOPS : 0) | (n_in << 8);
                r[1] = (int32_t)extra_off[(size_t)s];
                r[2] = r[3] = r[4] = 0;
                for (int j = 0; j < n_in; j++) {
                    if (j < 3) r[2 + j] = in[j];
                    else in_packed[(size_t)extra_off[(size_t)s] + (size_t)(j - 3)] = in[j];
                }
                r[5] = v; r[6] = 0; r[7] = 0;
            }
        });
    }

    tm.lap("  dense step records");
    // ---- compact step stream of the event-driven DP (dp_events.hip): only the vertices where a
    //      recombination can enter or leave, or a walk starts or ends; in-edges count compact steps back
    c->dp_events = n_walks <= PHI_DP_EVENT_MAX_WALKS && !getenv("PHI_DP_DENSE");
    c->n_k = 0; c->n_ev = 0;
    std::vector<int32_t> k_rec, k_in, cvtx;                    // (alive until the end of this call: their uploads are not waited for here)
    if (c->dp_events) {
        std::vector<uint8_t> lane_only(n_vtx, 0);
        for (int32_t h = 0; h < n_walks; h++) {
            lane_only[walk_first(h)] = 1;
            lane_only[walk_last(h)] = 1;
        }
        {
            // the compact steps, numbered in step order: counted per chunk of steps, then written, by all threads
            const int64_t SCH = 1 << 16, n_sch = ((int64_t)n_vtx + SCH - 1) / SCH;
            std::vector<int32_t> ch_cnt((size_t)n_sch + 1, 0);
            c->h_cstep.resize((size_t)n_vtx);
            auto keeps = [&](int64_t s_) { return (st_rec[(size_t)s_ * 8] & 3) || lane_only[(size_t)c->h_topo[(size_t)s_]]; };
            phi_parallel_chunks(n_vtx, SCH, [&](int64_t lo, int64_t hi, int) {
                int32_t n = 0;
                for (int64_t s_ = lo; s_ < hi; s_++) n += keeps(s_);
                ch_cnt[(size_t)(lo / SCH) + 1] = n;
            });
            for (int64_t i = 0; i < n_sch; i++) ch_cnt[(size_t)i + 1] += ch_cnt[(size_t)i];
            c->h_kstep.resize((size_t)ch_cnt[(size_t)n_sch]);
            phi_parallel_chunks(n_vtx, SCH, [&](int64_t lo, int64_t hi, int) {
                int32_t k_ = ch_cnt[(size_t)(lo / SCH)];
                for (int64_t s_ = lo; s_ < hi; s_++) {
                    if (keeps(s_)) { c->h_cstep[(size_t)s_] = k_; c->h_kstep[(size_t)k_++] = (int32_t)s_; }
                    else c->h_cstep[(size_t)s_] = -1;
                }
            });
        }
        c->n_k = (int32_t)c->h_kstep.size();
        k_rec.assign((size_t)c->n_k * 8, 0); k_in.clear(); cvtx.assign(n_vtx, 0);
        for (int32_t k = 0; k < c->n_k; k++) {
            const int32_t s = c->h_kstep[k];
            const int32_t *ro = &st_rec[(size_t)s * 8];
            int32_t *r = &k_rec[(size_t)k * 8];
            const int n_in = (ro[0] >> 8) & 0xFF;
            r[0] = ro[0] | (lane_only[c->h_topo[s]] ? PHI_DP_LANE_ONLY : 0);
            r[1] = (int32_t)k_in.size();
            for (int j = 0; j < n_in; j++) {
                const int32_t p = j < 3 ? ro[2 + j] : in_packed[ro[1] + j - 3];
                const int32_t kc = c->h_cstep[s - (int32_t)((uint32_t)p >> 8)];
                if (kc < 0) return phi_fail(c, PHI_ERR_DEVICE, "live in-edge from a vertex without leaving states (internal error)");
                const int32_t pc = ((k - kc) << 8) | (p & 0xFF);
                if (j < 3) r[2 + j] = pc; else k_in.push_back(pc);
            }
            r[5] = ro[5];
        }
        // two alleles of one site: consecutive in topological order, no edge between them (so no walk
        // visits both), neither leaves recombination states -> the consumer takes them in one iteration
        int64_t n_pairs = 0;
        for (int32_t k = 0; k + 1 < c->n_k; k++) {
            int32_t *r0 = &k_rec[(size_t)k * 8], *r1 = r0 + 8;
            if ((r0[0] | r1[0]) & PHI_DP_NEED_TOPS) continue;
            const int32_t s0 = c->h_kstep[k], s1 = c->h_kstep[k + 1];
            if (s1 != s0 + 1) continue;
            const int32_t v0 = c->h_topo[s0], v1 = c->h_topo[s1];
            bool edge = false;
            for (int64_t a = c->h_adj_off[v0]; a < c->h_adj_off[v0 + 1] && !edge; a++) edge = c->h_adj[a] == v1;
            if (edge) continue;
            r0[0] |= PHI_DP_PAIR;
            n_pairs++;
            k++;                                           // pairs do not overlap
        }
        if (tm.on) {
            int64_t n_tops = 0, n_entry = 0;
            for (int32_t k = 0; k < c->n_k; k++) { n_tops += (k_rec[(size_t)k * 8] & PHI_DP_NEED_TOPS) != 0; n_entry += (k_rec[(size_t)k * 8] & PHI_DP_NEED_ENTRY) != 0; }
            fprintf(stderr, "[phi timing] set_graph: %d compact steps: %lld with TOPS, %lld with ENTRY, %lld pairs\n", c->n_k, (long long)n_tops, (long long)n_entry, (long long)n_pairs);
        }
        phi_parallel_chunks(n_vtx, 1 << 16, [&](int64_t lo, int64_t hi, int) { for (int64_t v = lo; v < hi; v++) cvtx[(size_t)v] = c->h_cstep[(size_t)topo_rank[v]]; });
        tm.lap("  compact records");
        // where the chain of steps may be cut (dp_events.hip, blocks in parallel): not between the two steps of a pair,
        // and only where no recombination edge of this or a later step comes from before the cut
        {
            c->h_k_cut_ok.assign((size_t)c->n_k + 1, 1);
            int32_t min_src = INT32_MAX;                       // smallest source step of an in-edge of any step >= k
            for (int32_t k = c->n_k - 1; k >= 0; k--) {
                const int32_t *r = &k_rec[(size_t)k * 8];
                const int n_in = (r[0] >> 8) & 0xFF;
                for (int j = 0; j < n_in; j++) {
                    const int32_t pk = j < 3 ? r[2 + j] : k_in[(size_t)r[1] + j - 3];
                    min_src = std::min(min_src, k - (int32_t)((uint32_t)pk >> 8));
                }
                if (min_src < k) c->h_k_cut_ok[(size_t)k] = 0;
                if (k > 0 && (k_rec[(size_t)(k - 1) * 8] & PHI_DP_PAIR)) c->h_k_cut_ok[(size_t)k] = 0;
            }
            c->h_k_cut_ok[0] = 0; c->h_k_cut_ok[(size_t)c->n_k] = 0;
        }
        // (tried on a stream of this thread's own, so that these copies do not queue behind the GPU thread's kernels:
        //  0.8 ms slower -- the pageable copies of a second stream do not share the first one's staging)
        PHICHK(upload(c, c->d_k_rec, k_rec.data(), k_rec.size()));
        PHICHK(upload(c, c->d_k_in, k_in.data(), k_in.size()));
        PHICHK(upload(c, c->d_cvtx, cvtx.data(), cvtx.size()));
        // (no wait: a synchronisation here would also wait for whatever the GPU thread has queued on the stream; the
        //  vectors live until phi_sync_check at the end of this call)
    }
    tm.lap("DP step stream");
    // ---- the GPU side has been running meanwhile
    {
        const int grc = gpu_future.get();
        if (grc) return grc;
    }
    tm.lap("wait for the GPU thread");
    // ---- device copies of what the host pass made
    c->dp_dense_ready = want_masks;
    if (want_masks) {                                          // the every-vertex stream serves dp.hip only
        PHICHK(upload(c, c->d_st_rec, st_rec, st_rec_n));
        PHICHK(upload(c, c->d_in_packed, in_packed.data(), in_packed.size()));
    }
    // events of every walk: its entries on the compact steps
    if (c->dp_events) {
        PHICHK(phi_dev_ensure(c, c->d_flags, (size_t)n_entries));
        phi_launch_event_flags(c->stream, c->d_walk_vtx.as<int32_t>(), n_entries, c->d_cvtx.as<int32_t>(), c->d_flags.as<uint8_t>());
        PHICHK(phi_compact(c, c->d_flags.as<uint8_t>(), n_entries, c->d_ev_e, &c->n_ev));
        // (event indices are 32-bit signed in the block tables and on the DP lanes; events are the entries on vertices where a
        //  recombination can enter or leave or a walk begins or ends: 18 % of the entries of a chromosome-scale graph)
        if (c->n_ev >= (int64_t)1 << 31) return phi_fail(c, PHI_ERR_UNSUPPORTED, "more than 2^31 walk entries on vertices with recombination edges");
        PHICHK(phi_dev_ensure(c, c->d_ev_off, (size_t)(n_walks + 1) * 8));
        phi_launch_event_off(c->stream, c->d_ev_e.as<phi_ent_t>(), c->n_ev, c->d_walk_off.as<int64_t>(), n_walks,
                             c->d_ev_off.as<int64_t>());
    }
    HIPCHK(hipGetLastError());
    PHICHK(phi_sync_check(c));
    if (tm.on) fprintf(stderr, "[phi timing] set_graph: %d vertices, %d compact steps, %lld entries, %lld events\n", n_vtx, c->n_k, (long long)n_entries, (long long)c->n_ev);
    tm.lap("late uploads + event list");

    c->reads_bases = 0; c->reads_count = 0; c->spectrum_override = -1;
    c->sp_set_gen = -1; c->log_chunks = c->log_done = 0; c->logged_done = 0; c->ov_done = 0; c->ov_bound = 0; c->async_batches = false;
    c->walks_on_device = false;                                // (consumed: a later phi_set_graph brings its own walks)
    c->nov_shift = phi_nov_shift(c->w);
    if (const char *e = getenv("PHI_NOV_SHIFT")) c->nov_shift = std::max(0, std::min(9, atoi(e)));   // tests: chunk logs of a few entries, so that ordinary reads spill into the overflow list
    c->alt.needs_clean = false;
    c->next_flag_zeroed = false;                               // (set_graph zeroed all scalars, the overflow counters among them)
    stage_release(c);
    if (c->dp_alloc_future.valid()) {
        const int rc = c->dp_alloc_future.get();
        if (rc) return rc;
        tm.lap("wait for the DP buffers");
    }
    c->have_graph = true;
    return PHI_OK;
}

}  // extern "C"

extern "C" {

int phi_add_reads_device(phi_ctx *c, const void *d_bases, const void *d_read_off, int64_t n_reads, int64_t n_bases)
{
    const int rc = phi_add_reads_device_impl(c, d_bases, d_read_off, n_reads, n_bases, false);
    if (c && rc == PHI_OK) c->async_batches = true;            // nobody waits behind this batch: an overflow list that ran full shows at the next check
    return rc;
}

}  // extern "C"

// replay: the same batch again after the overflow list of novel hashes was grown (its first pass filled the list):
// everything a batch does is idempotent (hit flags, the log's entries: the replay rewrites the same ones) except the
// counts of emitted minimisers and of logged hashes, which are not repeated
int phi_add_reads_device_impl(phi_ctx *c, const void *d_bases, const void *d_read_off, int64_t n_reads, int64_t n_bases, bool replay)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_add_reads before phi_set_graph");
    if (n_reads < 0 || n_bases < 0 || (n_bases > 0 && !d_bases)) return phi_fail(c, PHI_ERR_INVALID, "phi_add_reads: bad arguments");
    // no offsets: reads of one length, n_bases / n_reads each
    int64_t uniform_len = 0;
    if (!d_read_off && n_reads > 0 && n_bases > 0) {
        if (n_bases % n_reads) return phi_fail(c, PHI_ERR_INVALID, "phi_add_reads_device without offsets: %lld bases are not %lld reads of one length", (long long)n_bases, (long long)n_reads);
        uniform_len = n_bases / n_reads;
        if (uniform_len < 32 || uniform_len > 0x7FFFFFFF) {
            // (shorter than the kernel's arithmetic covers, or absurdly long: make the offsets)
            PHICHK(phi_dev_ensure(c, c->d_roff_made, (size_t)(n_reads + 1) * 8));
            phi_launch_iota_i64(c->stream, c->d_roff_made.as<int64_t>(), n_reads + 1, uniform_len);
            d_read_off = c->d_roff_made.p;
            uniform_len = 0;
        }
    }
    if (n_reads == 0 || n_bases == 0) { if (!replay) c->reads_count += n_reads; return PHI_OK; }
    HIPCHK(hipSetDevice(c->device));
    c->solved = false;
    // this batch's part of the log of novel hashes: 1 << nov_shift entries per chunk, behind the chunks logged so far.
    // A log that has no room grows -- keeping what it holds -- up to 1 GB; beyond that what it holds is entered into
    // the set (sp_flush) and the log starts over.  (A batch larger than the log has it made as large as the batch.)
    const int64_t n_log_chunks = phi_sketch_num_blocks(n_bases);
    if (replay) c->log_chunks -= c->last_log_chunks;          // the same entries again
    {
        const size_t ent = (size_t)8 << c->nov_shift;
        size_t need = (size_t)(c->log_chunks + n_log_chunks) * ent;
        if (need > c->d_novlog.cap || (size_t)(c->log_chunks + n_log_chunks) * 2 > c->d_novcnt.cap) {
            size_t budget = (size_t)1 << 30;
            if (const char *e = getenv("PHI_NOVLOG_BUDGET")) budget = (size_t)std::max<long long>(atoll(e), 1);      // tests: a log that starts over
            if (c->log_chunks > 0 && need > budget) {
                PHICHK(phi_sp_flush(c, nullptr));
                c->log_chunks = c->log_done = 0;
                need = (size_t)n_log_chunks * ent;
                // (the overflow list starts over with the log: what it held is in the set)
                HIPCHK(hipMemsetAsync(scalar(c, S_OVCNT + (int)(c->sp_gen % 3)), 0, 8, c->stream));
                c->ov_done = 0; c->ov_bound = 0;
            }
            if (need > c->d_novlog.cap || (size_t)(c->log_chunks + n_log_chunks) * 2 > c->d_novcnt.cap) {
                const size_t want = std::max(need, std::min(2 * c->d_novlog.cap, budget));
                PHICHK(dev_grow_keep(c, c->d_novlog, want, (size_t)c->log_chunks * ent));
                PHICHK(dev_grow_keep(c, c->d_novcnt, want / ent * 2 + 64, (size_t)c->log_chunks * 2));
            }
        }
        // The overflow list: room for everything this batch can emit at 1.5x the density of random sequence (2 / (w + 1) per
        // base) -- reads full of N, or k > 32, send ALL their novel hashes there, and a batch handed over with
        // phi_add_reads_device has nobody behind it to grow the list and replay (phi_add_reads and the text path do:
        // replay_if_full).  Memory that is reserved, not touched: 0.9 bytes per base at w = 25.
        if (!replay) {
            c->ov_bound += (int64_t)((double)n_bases * std::min(1.0, 3.0 / (c->w + 1))) + 16;
            const char *e = getenv("PHI_OVLIST_CAP");                                                                     // tests: a list that runs full (provokes the replay)
            if (e && !c->d_ovlist.p) {
                c->ov_cap = std::max<long long>(atoll(e), 1);
                PHICHK(phi_dev_ensure(c, c->d_ovlist, (size_t)c->ov_cap * 8));
            } else if (!e && c->ov_bound > c->ov_cap) {
                const int64_t cap = std::max<int64_t>(std::max<int64_t>(c->ov_bound, 2 * c->ov_cap), 1 << 16);
                PHICHK(dev_grow_keep(c, c->d_ovlist, (size_t)cap * 8, (size_t)c->ov_cap * 8));
                c->ov_cap = cap;
            }
        }
    }
    // ONE launch per batch: every wave stages its chunk straight from the ASCII bases (2-bit pack, bases outside
    // ACGTacgt, read starts from the offsets), sketches, hashes and probes; windows touching a base outside ACGT
    // take the exact byte-wise routine inside the same wave
    PhiSketchArgs A{};
    A.ascii = (const uint8_t *)d_bases;
    A.read_off = (const int64_t *)d_read_off; A.n_reads = n_reads;
    A.uniform_len = (int32_t)uniform_len; A.inv_len = uniform_len ? 1.0 / (double)uniform_len : 0.0;
    A.inv_len_q32 = uniform_len ? (uint32_t)(((uint64_t)1 << 32) / (uint64_t)uniform_len) : 0u;
    {
        const double q = (double)n_reads / (double)n_bases * 4294967296.0;      // (a guess: clamped, never wrong to round)
        A.reads_per_base_q32 = q >= 2147483648.0 ? 0x80000000u : (uint32_t)q;
    }
    A.allslow = c->k > PHI_MAX_K_PACKED;                       // longer k-mers: the byte-wise routine for every window
    A.n_bases = n_bases; A.k = c->k; A.w = c->w;
    A.n_logged = replay ? nullptr : logged_stripes(c);
    A.n_emitted = replay ? nullptr : emit_stripes(c);
    A.u_kv = c->d_u_kv.as<uint64_t>(); A.u_mask = c->u_cap - 1;
    A.hit = c->d_hit.as<uint8_t>();
    A.err = (uint32_t *)scalar(c, S_ERR);
    A.nov_log = c->d_novlog.as<uint64_t>(); A.nov_cnt = c->d_novcnt.as<uint16_t>(); A.log_base = c->log_chunks; A.nov_shift = c->nov_shift;
    A.ov_list = c->d_ovlist.as<uint64_t>(); A.ov_cap = c->ov_cap;
    A.ov_count = (unsigned long long *)scalar(c, S_OVCNT + (int)(c->sp_gen % 3));
    c->log_chunks += n_log_chunks; c->last_log_chunks = n_log_chunks;
    if (c->alt.needs_clean) {
        // the first launch since the reset: its waves zero what the ended generation filled (the other half of the
        // double buffers) for the generation after this one, and that generation's overflow counter
        A.q_clean = 1;
        A.ov_zero = (unsigned long long *)scalar(c, S_OVCNT + (int)((c->sp_gen + 1) % 3));
        A.q_hit_words = c->alt.hit.as<uint64_t>(); A.q_n_hit_words = c->n_unique / 8 + 1;
        A.q_stripes = c->alt.stripes.as<uint64_t>(); A.q_n_stripe_words = 2 * PHI_STRIPES * 8;
        c->alt.needs_clean = false;
        c->next_flag_zeroed = true;
    }
    phi_ipc_launch_args(c, A);                                 // (a context in a group of processes: the flags of its exchanges)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (c->prof && c->prof_period > 0 && (c->prof_seq++ % c->prof_period) == 0) {
        if (c->prof_used == c->prof_events.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            c->prof_events.emplace_back(a, b);
        }
        t0 = c->prof_events[c->prof_used].first; t1 = c->prof_events[c->prof_used].second;
        c->prof_used++;
        c->prof_bases += n_bases;
    }
    phi_launch_sketch(c->stream, PHI_MODE_PROBE, A, t0, t1);
    HIPCHK(hipGetLastError());
    if (!replay) { c->reads_bases += n_bases; c->reads_count += n_reads; }
    return PHI_OK;
}


// After the stream has been waited for: if the overflow list of novel hashes ran full under the batch whose data still
// sits at (d_bases, d_off) -- chunks that emit far more than random sequence does; the reference's std::map has no such
// limit, ILP_index.cpp:622-635 -- grow the list (keeping what it holds: everything below its old capacity is valid) and
// replay that batch.  err = the device error word as read behind the batch.
static int replay_if_full(phi_ctx *c, uint32_t err, const void *d_bases, const void *d_off, int64_t n_reads, int64_t n_bases)
{
    for (int attempt = 0; attempt < 8 && (err & PHI_KERR_TABLE_FULL); attempt++) {
        unsigned long long *cnt = (unsigned long long *)scalar(c, S_OVCNT + (int)(c->sp_gen % 3));
        unsigned long long wanted = 0;
        HIPCHK(phi_copy_sync(c, &wanted, cnt, 8, hipMemcpyDeviceToHost));
        const unsigned long long valid = std::min<unsigned long long>(wanted, (unsigned long long)c->ov_cap);
        const int64_t cap = (int64_t)std::max<unsigned long long>(4ull * (unsigned long long)c->ov_cap, 2 * wanted);
        PHICHK(dev_grow_keep(c, c->d_ovlist, (size_t)cap * 8, (size_t)valid * 8));
        c->ov_cap = cap;
        HIPCHK(phi_copy_sync(c, cnt, &valid, 8, hipMemcpyHostToDevice));
        err &= ~PHI_KERR_TABLE_FULL;
        HIPCHK(phi_copy_sync(c, scalar(c, S_ERR), &err, 4, hipMemcpyHostToDevice));
        PHICHK(phi_add_reads_device_impl(c, d_bases, d_off, n_reads, n_bases, true));
        HIPCHK(phi_copy_sync(c, &err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost));
    }
    return PHI_OK;
}

static bool offsets_monotone(const int64_t *off, int64_t n)
{
    // branch-free, so that the compiler vectorises it: 33 000 offsets in a few microseconds
    auto span = [off](int64_t lo, int64_t hi) { int bad = 0; for (int64_t r = lo; r < hi; r++) bad |= off[r + 1] < off[r]; return bad; };
    if (n < (1 << 20)) return !span(0, n);
    std::atomic<int> bad{0};
    phi_parallel_chunks(n, (int64_t)1 << 20, [&](int64_t lo, int64_t hi, int) { if (span(lo, hi)) bad.store(1); });
    return !bad.load();
}

static bool offsets_uniform(const int64_t *off, int64_t n, int64_t len)
{
    auto span = [off, len](int64_t lo, int64_t hi) { int bad = 0; for (int64_t r = lo; r < hi; r++) bad |= off[r + 1] - off[r] != len; return bad; };
    if (n < (1 << 20)) return !span(0, n);
    std::atomic<int> bad{0};
    phi_parallel_chunks(n, (int64_t)1 << 20, [&](int64_t lo, int64_t hi, int) { if (span(lo, hi)) bad.store(1); });
    return !bad.load();
}

extern "C" {

int phi_add_reads(phi_ctx *c, const char *bases, const int64_t *read_off, int64_t n_reads)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_add_reads before phi_set_graph");
    if (n_reads < 0 || (n_reads > 0 && !read_off)) return phi_fail(c, PHI_ERR_INVALID, "phi_add_reads: bad arguments");
    if (n_reads == 0) return PHI_OK;
    if (read_off[0] != 0) return phi_fail(c, PHI_ERR_INVALID, "read_off must start at 0");
    if (!offsets_monotone(read_off, n_reads)) {
        for (int64_t r = 0; r < n_reads; r++)
            if (read_off[r + 1] < read_off[r]) return phi_fail(c, PHI_ERR_INVALID, "read_off not monotone at read %lld", (long long)r);
    }
    const int64_t n_bases = read_off[n_reads];
    if (n_bases > 0 && !bases) return phi_fail(c, PHI_ERR_INVALID, "phi_add_reads: bases is null");
    // reads of one length (>= 32): their offsets are r * length -- they need not cross the link, nor be read by the kernel
    const int64_t len0 = read_off[1];
    const bool uniform = n_bases > 0 && len0 >= 32 && len0 * n_reads == n_bases && offsets_uniform(read_off, n_reads, len0);
    HIPCHK(hipSetDevice(c->device));
    PhiStageTimer tm("add_reads");
    // (every call ends with a wait for the stream: nothing of an earlier batch still reads the staging buffers)
    if (c->async_batches) { PHICHK(phi_sync_check(c)); c->async_batches = false; }   // (a replay below must only ever concern THIS batch)
    PHICHK(phi_dev_ensure(c, c->d_roff, (size_t)(n_reads + 1) * 8));
    if (!c->h_err) HIPCHK(hipHostMalloc((void **)&c->h_err, 64, hipHostMallocDefault));
    tm.lap("buffers");
    // One staged copy, then the kernel, on one stream.  What was measured on the MI355X box before settling for it
    // (profiles/r03_h2d_experiments.txt; C2 = 5.2 MB per batch, the link moves 52-57 GB/s):
    //   * pieces of the batch copied on a second stream while the piece before is sketched: every copy-engine -> compute
    //     dependency (event record + stream wait) costs ~50 us, more than the kernel of a 1-MB piece: 300 us per batch
    //     with 1-MB pieces, 262 with 2-MB, against 177 for the single copy (C3: 2245 / 1810 / 1277 us);
    //   * the kernel reading pinned bases in place across the link (every base is loaded exactly once): shader reads of
    //     host memory reach 25-32 GB/s, half the copy engine's rate: 190 us (C3: 1673 us).
    PHICHK(phi_dev_ensure(c, c->d_rbases, (size_t)std::max<int64_t>(n_bases, 1)));
    if (!uniform) HIPCHK(hipMemcpyAsync(c->d_roff.p, read_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n_bases) HIPCHK(hipMemcpyAsync(c->d_rbases.p, bases, (size_t)n_bases, hipMemcpyHostToDevice, c->stream));
    PHICHK(phi_add_reads_device_impl(c, c->d_rbases.p, uniform ? nullptr : c->d_roff.p, n_reads, n_bases, false));
    // the error word behind the last kernel; host buffers are borrowed for the call only: one wait for everything
    HIPCHK(hipMemcpyAsync(c->h_err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    tm.lap("H2D + sketch + probe");
    if (*c->h_err & PHI_KERR_TABLE_FULL) {
        PHICHK(replay_if_full(c, *c->h_err, c->d_rbases.p, uniform ? nullptr : c->d_roff.p, n_reads, n_bases));
        tm.lap("overflow list grown, batch replayed");
    }
    return PHI_OK;
}

// ---- reads as raw text: the records are found on the device (reads_text.hip)

int phi_reads_text_begin(phi_ctx *c, int64_t max_chunk_bytes)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_reads_text_begin before phi_set_graph");
    if (max_chunk_bytes <= 0) return phi_fail(c, PHI_ERR_INVALID, "phi_reads_text_begin: bad chunk size");
    HIPCHK(hipSetDevice(c->device));
    PHICHK(phi_sync_check(c));                                // (batches handed over without a wait: their overflow shows here, not in a replay of ours)
    c->async_batches = false;
    auto &T = c->text;
    const uint32_t chunk = (uint32_t)std::min<int64_t>(std::max<int64_t>(max_chunk_bytes, 64), (int64_t)1 << 28);
    // the carry holds what a chunk leaves unfinished: a record at most (the longest reads are a few Mbases, twice that as FASTQ)
    const uint32_t carry = getenv("PHI_TEXT_CARRY") ? (uint32_t)std::max<long long>(64, atoll(getenv("PHI_TEXT_CARRY"))) & ~15u : std::max<uint32_t>(chunk / 2, 1u << 24);
    const uint32_t line_cap = (carry + chunk) / 16 + 1024;    // a line of a read file is 75 bytes on average, never 16 (then: irregular)
    for (int i = 0; i < 2; i++) {
        PHICHK(phi_dev_ensure(c, T.text[i], (size_t)carry + chunk + 64));
        PHICHK(phi_dev_ensure(c, T.bases[i], (size_t)carry + chunk + 64));
        PHICHK(phi_dev_ensure(c, T.roff[i], ((size_t)line_cap + 2) * 8));
    }
    PHICHK(phi_dev_ensure(c, T.tile_cnt, ((size_t)phi_text_num_tiles(0, carry + chunk + 64) + 2) * 4));
    PHICHK(phi_dev_ensure(c, T.ls, ((size_t)line_cap + 2) * 4));
    PHICHK(phi_dev_ensure(c, T.pre, ((size_t)line_cap + 2) * 8));
    PHICHK(phi_dev_ensure(c, T.blk, ((size_t)phi_text_scan_blocks(line_cap) + 1) * 8));
    PHICHK(phi_dev_ensure(c, T.sum, sizeof(PhiTextSummary)));
    if (!T.h_sum) HIPCHK(hipHostMalloc((void **)&T.h_sum, 64 + sizeof(PhiTextSummary), hipHostMallocDefault));
    if (!T.ev_copy) HIPCHK(hipEventCreateWithFlags(&T.ev_copy, hipEventDisableTiming));
    T.carry_cap = carry; T.chunk_cap = chunk; T.line_cap = line_cap;
    T.active = true; T.irregular = false; T.started = false; T.mode = 0;
    T.fed = 0; T.taken = 0; T.slot = 0; T.carry_len = 0; T.carry_at = carry; T.h_carry.clear();
    T.last_slot = -1; T.why = 0; T.first_bad = 0; T.detached = false; T.carry_stale = false;
    return PHI_OK;
}

}  // extern "C"

// the sketch of the chunk before may have filled the overflow list of novel hashes: its bases and offsets are still in their slot
static int text_replay_last(phi_ctx *c, uint32_t err)
{
    auto &T = c->text;
    if (T.last_slot < 0 || !(err & PHI_KERR_TABLE_FULL)) return PHI_OK;
    return replay_if_full(c, err, T.bases[T.last_slot].p, T.last_uniform ? nullptr : T.roff[T.last_slot].p, T.last_reads, T.last_bases);
}

// the carry's host copy, when the pieces came from device memory (parked text: nobody had their bytes on the host)
static int text_carry_to_host(phi_ctx *c)
{
    auto &T = c->text;
    if (!T.carry_stale) return PHI_OK;
    T.h_carry.resize(T.carry_len);
    if (T.carry_len) HIPCHK(phi_copy_sync(c, T.h_carry.data(), T.text[T.slot].as<uint8_t>() + T.carry_at, T.carry_len, hipMemcpyDeviceToHost));
    T.carry_stale = false;
    return PHI_OK;
}

// one piece of the stream: m bytes at p (host memory) -- or at d_src (device memory, p = NULL; first = the piece's first byte)
static int text_piece(phi_ctx *c, const char *p, uint32_t m, int32_t *irregular, const void *d_src = nullptr, char first = 0)
{
    auto &T = c->text;
    T.dbg_reads = 0; T.dbg_bases = 0;
    if (p) first = p[0];
    if (!T.started) {
        T.started = true;
        // the layout is decided by the first byte of the stream; text before the first header is the host reader's business
        if (first == '@') T.mode = 1;
        else if (first == '>') T.mode = 0;
        else { T.irregular = true; T.why = PHI_TEXT_IRREGULAR_LAYOUT; T.first_bad = 0; *irregular = 1; return PHI_OK; }
    }
    if (p) PHICHK(text_carry_to_host(c));
    if (T.detached) { T.h_carry.clear(); T.detached = false; }
    const int slot = T.slot ^ 1;
    const uint32_t C = T.carry_cap, start = C - T.carry_len, end = C + m;
    uint8_t *buf = T.text[slot].as<uint8_t>();
    // chunk i + 1 crosses the link while chunk i is sketched: the copy runs on aux_stream, everything else on `stream`
    HIPCHK(hipMemcpyAsync(buf + C, p ? (const void *)p : d_src, m, p ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->aux_stream));
    HIPCHK(hipEventRecord(T.ev_copy, c->aux_stream));
    if (T.carry_len)
        HIPCHK(hipMemcpyAsync(buf + start, T.text[T.slot].as<uint8_t>() + T.carry_at, T.carry_len, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, T.ev_copy, 0));
    HIPCHK(hipMemsetAsync(T.sum.p, 0, sizeof(PhiTextSummary), c->stream));
    HIPCHK(hipMemsetAsync(&T.sum.as<PhiTextSummary>()->first_bad, 0xFF, 4, c->stream));
    PhiTextArgs A{};
    A.buf = buf; A.start = start; A.end = end; A.mode = T.mode; A.line_cap = T.line_cap;
    A.tile_cnt = T.tile_cnt.as<uint32_t>(); A.ls = T.ls.as<uint32_t>(); A.pre = T.pre.as<uint64_t>(); A.blk = T.blk.as<uint64_t>();
    A.read_off = T.roff[slot].as<int64_t>(); A.bases = T.bases[slot].as<uint8_t>(); A.sum = T.sum.as<PhiTextSummary>();
    phi_launch_reads_text(c->stream, A);
    HIPCHK(hipGetLastError());
    uint32_t *h_err = (uint32_t *)((char *)T.h_sum + sizeof(PhiTextSummary));
    HIPCHK(hipMemcpyAsync(T.h_sum, T.sum.p, sizeof(PhiTextSummary), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                 // also ends the sketch of the chunk before, and the copy of this one
    PHICHK(text_replay_last(c, *h_err));
    T.last_slot = -1;
    T.dbg_reads = 0; T.dbg_bases = 0;
    const PhiTextSummary S = *T.h_sum;
    const uint32_t tail = end - S.cons_end;
    if (S.err || tail > C) {
        // nothing of this chunk is taken: the caller parses the pending bytes (phi_reads_text_end) and this chunk on the host
        T.irregular = true; T.why = S.err ? S.err : PHI_TEXT_IRREGULAR_LINES; T.first_bad = S.first_bad; *irregular = 1;
        return PHI_OK;
    }
    if (S.n_rec) {
        // records of one length (>= 32): the sketch kernel computes the read starts, no offsets
        const bool uni = !S.not_uniform && S.n_bases % S.n_rec == 0 && S.n_bases / S.n_rec >= 32;
        T.last_uniform = uni;
        PHICHK(phi_add_reads_device_impl(c, T.bases[slot].p, uni ? nullptr : T.roff[slot].p, (int64_t)S.n_rec, (int64_t)S.n_bases, false));
        T.last_slot = slot; T.last_reads = (int64_t)S.n_rec; T.last_bases = (int64_t)S.n_bases;
        T.dbg_slot = slot; T.dbg_reads = (int64_t)S.n_rec; T.dbg_bases = (int64_t)S.n_bases;
    }
    // the bytes not taken, on the host as well: the last `tail` bytes of (carry before + this chunk)
    if (!p) T.carry_stale = true;                            // (fetched from the device when somebody asks: text_carry_to_host)
    else if (tail <= m) T.h_carry.assign(p + m - tail, p + m);
    else {
        const size_t keep = tail - m;                        // of the carry before
        T.h_carry.erase(T.h_carry.begin(), T.h_carry.end() - (ptrdiff_t)keep);
        T.h_carry.insert(T.h_carry.end(), p, p + m);
    }
    T.taken += (int64_t)(S.cons_end - start);
    T.fed += m;
    T.carry_len = tail; T.carry_at = S.cons_end; T.slot = slot;
    return PHI_OK;
}

extern "C" {

int phi_add_reads_text(phi_ctx *c, const char *text, int64_t n_bytes, int32_t *irregular)
{
    if (!c || !irregular || n_bytes < 0 || (n_bytes > 0 && !text)) return PHI_ERR_INVALID;
    *irregular = 0;
    auto &T = c->text;
    if (!T.active) return phi_fail(c, PHI_ERR_STATE, "phi_add_reads_text before phi_reads_text_begin");
    if (T.irregular) return phi_fail(c, PHI_ERR_STATE, "phi_add_reads_text after an irregular chunk: finish the stream on the host reader");
    HIPCHK(hipSetDevice(c->device));
    for (int64_t at = 0; at < n_bytes; ) {
        const uint32_t m = (uint32_t)std::min<int64_t>(n_bytes - at, T.chunk_cap);
        PHICHK(text_piece(c, text + at, m, irregular));
        if (*irregular) {
            // what the device has not taken = the carry + the rest of this call's bytes: all of it is handed back by
            // phi_reads_text_end, the caller goes on with the bytes of the stream AFTER this call's
            T.h_carry.insert(T.h_carry.end(), text + at, text + n_bytes);
            break;
        }
        at += m;
    }
    return PHI_OK;
}

/* Text that arrives before the graph is there waits in device memory (include/phi_amd.h): a park is no part of any context's
 * state -- its own stream, its own buffers --, so that a reader thread fills it while phi_set_graph runs on another. */
struct phi_text_park {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    struct Piece { DevBuf d; int64_t n = 0; char first = 0; hipEvent_t ev = nullptr; };      // ev: the copy of the bytes has landed
    std::deque<Piece> pieces;                                  // (a deque: references stay valid while pieces are added)
    std::vector<DevBuf> spare;                                 // buffers of released pieces
    std::vector<void *> pinned;
};

int phi_text_park_create(int32_t device, phi_text_park **out)
{
    if (!out) return PHI_ERR_INVALID;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return PHI_ERR_DEVICE;
    phi_text_park *p = new phi_text_park();
    p->device = device;
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { delete p; return PHI_ERR_DEVICE; }
    *out = p;
    return PHI_OK;
}

int phi_text_park_pin(phi_text_park *p, void *host, size_t bytes)
{
    if (!p || !host) return PHI_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess || hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return PHI_ERR_DEVICE; }
    p->pinned.push_back(host);
    return PHI_OK;
}

// the copy is ISSUED when this returns; phi_text_park_wait says when the host buffer may be written again (whoever takes the
// piece -- phi_add_reads_text_parked, phi_text_park_fetch -- waits for it by itself)
int phi_text_park_add_async(phi_text_park *p, const char *text, int64_t n, int32_t *index)
{
    if (!p || !text || n <= 0 || !index) return PHI_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return PHI_ERR_DEVICE;
    phi_text_park::Piece pc;
    const size_t want = (size_t)n + 64;
    {
        // a buffer a released piece left behind, if one is large enough (pieces are of one size but the last)
        std::lock_guard<std::mutex> lk(p->mu);
        for (size_t i = 0; i < p->spare.size(); i++)
            if (p->spare[i].cap >= want) { pc.d = p->spare[i]; p->spare.erase(p->spare.begin() + (ptrdiff_t)i); break; }
    }
    if (!pc.d.p) {
        if (hipMalloc(&pc.d.p, want) != hipSuccess) { (void)hipGetLastError(); return PHI_ERR_NOMEM; }
        pc.d.cap = want;
    }
    pc.n = n; pc.first = text[0];
    if (hipEventCreateWithFlags(&pc.ev, hipEventDisableTiming) != hipSuccess ||
        hipMemcpyAsync(pc.d.p, text, (size_t)n, hipMemcpyHostToDevice, p->stream) != hipSuccess || hipEventRecord(pc.ev, p->stream) != hipSuccess) {
        (void)hipGetLastError(); (void)hipStreamSynchronize(p->stream);
        if (pc.ev) (void)hipEventDestroy(pc.ev);
        (void)hipFree(pc.d.p); return PHI_ERR_DEVICE;
    }
    std::lock_guard<std::mutex> lk(p->mu);
    p->pieces.push_back(pc);
    *index = (int32_t)p->pieces.size() - 1;
    return PHI_OK;
}

static phi_text_park::Piece *park_piece(phi_text_park *p, int32_t index)
{
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(p->mu);
    return index >= 0 && (size_t)index < p->pieces.size() && p->pieces[(size_t)index].d.p ? &p->pieces[(size_t)index] : nullptr;
}

int phi_text_park_wait(phi_text_park *p, int32_t index)
{
    phi_text_park::Piece *pc = park_piece(p, index);
    if (!pc) return PHI_ERR_INVALID;
    if (pc->ev && hipEventSynchronize(pc->ev) != hipSuccess) { (void)hipGetLastError(); return PHI_ERR_DEVICE; }
    return PHI_OK;
}

int phi_text_park_add(phi_text_park *p, const char *text, int64_t n, int32_t *index)
{
    const int rc = phi_text_park_add_async(p, text, n, index);
    return rc ? rc : phi_text_park_wait(p, *index);
}

int64_t phi_text_park_bytes(phi_text_park *p, int32_t index) { phi_text_park::Piece *pc = park_piece(p, index); return pc ? pc->n : -1; }

int phi_text_park_fetch(phi_text_park *p, int32_t index, char *out, int64_t cap)
{
    phi_text_park::Piece *pc = park_piece(p, index);
    if (!pc || !out || cap < pc->n) return PHI_ERR_INVALID;
    if (hipSetDevice(p->device) != hipSuccess) return PHI_ERR_DEVICE;
    if (pc->ev) (void)hipEventSynchronize(pc->ev);
    if (hipMemcpyAsync(out, pc->d.p, (size_t)pc->n, hipMemcpyDeviceToHost, p->stream) != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) { (void)hipGetLastError(); return PHI_ERR_DEVICE; }
    return PHI_OK;
}

int phi_text_park_release(phi_text_park *p, int32_t index)
{
    phi_text_park::Piece *pc = park_piece(p, index);
    if (!pc) return PHI_ERR_INVALID;
    // The buffer stays with the park (the next piece takes it; phi_text_park_destroy frees them all): a hipFree per piece would
    // wait for the device each time -- between the chunks of a stream whose sketches are meant to overlap -- and 165 of them at
    // the end of config 5's reads are time inside whatever runs next.
    if (pc->ev) { (void)hipEventSynchronize(pc->ev); (void)hipEventDestroy(pc->ev); pc->ev = nullptr; }
    std::lock_guard<std::mutex> lk(p->mu);
    p->spare.push_back(pc->d);
    pc->d = DevBuf{};
    return PHI_OK;
}

void phi_text_park_destroy(phi_text_park *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->stream);
    for (auto &pc : p->pieces) { if (pc.ev) (void)hipEventDestroy(pc.ev); if (pc.d.p) (void)hipFree(pc.d.p); }
    for (auto &d : p->spare) if (d.p) (void)hipFree(d.p);
    for (void *h : p->pinned) (void)hipHostUnregister(h);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int phi_add_reads_text_parked(phi_ctx *c, phi_text_park *p, int32_t index, int32_t *irregular)
{
    if (!c || !irregular) return PHI_ERR_INVALID;
    *irregular = 0;
    phi_text_park::Piece *pc = park_piece(p, index);
    if (!pc || p->device != c->device) return phi_fail(c, PHI_ERR_INVALID, "phi_add_reads_text_parked: no such piece on this context's device");
    auto &T = c->text;
    if (!T.active) return phi_fail(c, PHI_ERR_STATE, "phi_add_reads_text_parked before phi_reads_text_begin");
    if (T.irregular) return phi_fail(c, PHI_ERR_STATE, "phi_add_reads_text_parked after an irregular chunk: finish the stream on the host reader");
    HIPCHK(hipSetDevice(c->device));
    if (pc->ev) HIPCHK(hipEventSynchronize(pc->ev));          // (the piece's bytes have landed)
    for (int64_t at = 0; at < pc->n; ) {
        const uint32_t m = (uint32_t)std::min<int64_t>(pc->n - at, T.chunk_cap);
        PHICHK(text_piece(c, nullptr, m, irregular, pc->d.as<char>() + at, pc->first));      // (the first byte matters to the stream's first piece only)
        if (*irregular) {
            // as phi_add_reads_text: what the device has not taken = the carry + the rest of this piece, handed back by phi_reads_text_end
            PHICHK(text_carry_to_host(c));
            const size_t had = T.h_carry.size();
            T.h_carry.resize(had + (size_t)(pc->n - at));
            HIPCHK(phi_copy_sync(c, T.h_carry.data() + had, pc->d.as<char>() + at, (size_t)(pc->n - at), hipMemcpyDeviceToHost));
            break;
        }
        at += m;
    }
    return PHI_OK;
}

int phi_reads_text_last_batch(phi_ctx *c, char *bases, int64_t cap_bases, int64_t *off, int64_t cap_reads, int64_t *n_reads, int64_t *n_bases)
{
    if (!c || !n_reads || !n_bases) return PHI_ERR_INVALID;
    auto &T = c->text;
    *n_reads = T.dbg_reads; *n_bases = T.dbg_bases;
    if (T.dbg_reads == 0 || cap_reads < T.dbg_reads || cap_bases < T.dbg_bases || !off) return PHI_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (T.dbg_bases && bases) HIPCHK(phi_copy_sync(c, bases, T.bases[T.dbg_slot].p, (size_t)T.dbg_bases, hipMemcpyDeviceToHost));
    HIPCHK(phi_copy_sync(c, off, T.roff[T.dbg_slot].p, (size_t)(T.dbg_reads + 1) * 8, hipMemcpyDeviceToHost));
    return PHI_OK;
}

int phi_reads_text_detach_carry(phi_ctx *c, const char **bytes, int64_t *n)
{
    if (!c || !bytes || !n) return PHI_ERR_INVALID;
    auto &T = c->text;
    if (!T.active || T.irregular) return phi_fail(c, PHI_ERR_STATE, "phi_reads_text_detach_carry: no regular text stream is open");
    HIPCHK(hipSetDevice(c->device));
    PHICHK(text_carry_to_host(c));
    *bytes = T.h_carry.data(); *n = (int64_t)T.h_carry.size();
    T.carry_len = 0; T.detached = true;                       // (the host copy is dropped by the next piece: the pointer stays valid until then)
    return PHI_OK;
}

int phi_reads_text_end(phi_ctx *c, const char **pending, int64_t *n_pending, int64_t *n_taken)
{
    if (!c) return PHI_ERR_INVALID;
    auto &T = c->text;
    if (!T.active) return phi_fail(c, PHI_ERR_STATE, "phi_reads_text_end before phi_reads_text_begin");
    HIPCHK(hipSetDevice(c->device));
    uint32_t err = 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(phi_copy_sync(c, &err, scalar(c, S_ERR), 4, hipMemcpyDeviceToHost));
    PHICHK(text_replay_last(c, err));
    T.last_slot = -1;
    T.active = false;
    if (!T.irregular) PHICHK(text_carry_to_host(c));
    if (T.detached) { T.h_carry.clear(); T.detached = false; }      // (handed out already)
    if (pending) *pending = T.h_carry.data();
    if (n_pending) *n_pending = (int64_t)T.h_carry.size();
    if (n_taken) *n_taken = T.taken;
    return PHI_OK;
}

int phi_reset_reads(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_reset_reads before phi_set_graph");
    HIPCHK(hipSetDevice(c->device));
    // The buffers this generation filled go to the back (the next read launch zeroes them), the other half
    // comes to the front: no launch, no wait.  The log of novel hashes starts over; the set made from it belongs to the
    // ended generation (sp_set_gen) and is emptied when the new one first needs it.
    const bool front_dirty = c->alt.needs_clean;               // two resets with no read launch in between
    swap_read_bufs(c);
    c->alt.needs_clean = true;
    if (front_dirty) {
        PHICHK(phi_ipc_before_generation(c, c->sp_gen + 1));   // (a group of processes: the peers are done with what is about to be zeroed)
        // nothing zeroed the half that comes to the front: do it now
        phi_launch_reset_reads(c->stream, c->d_hit.as<uint64_t>(), c->d_hit.p ? c->n_unique / 8 + 1 : 0, c->d_stripes.as<uint64_t>(), 2 * PHI_STRIPES * 8);
        HIPCHK(hipGetLastError());
    }
    c->sp_gen++;
    if (!c->next_flag_zeroed) HIPCHK(hipMemsetAsync(scalar(c, S_OVCNT + (int)(c->sp_gen % 3)), 0, 8, c->stream));   // the new generation's overflow counter
    c->next_flag_zeroed = false;
    c->log_chunks = c->log_done = 0; c->logged_done = 0; c->ov_done = 0; c->ov_bound = 0; c->last_log_chunks = 0;
    c->reads_bases = 0; c->reads_count = 0; c->spectrum_override = -1;
    c->solved = false;
    return PHI_OK;
}

int phi_reads_stats(phi_ctx *c, int64_t *n_reads, int64_t *n_bases, int64_t *n_emitted, int64_t *n_distinct)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_reads_stats before phi_set_graph");
    HIPCHK(hipSetDevice(c->device));
    PHICHK(phi_sync_check(c));
    uint64_t nd = 0, ne = 0;
    PHICHK(phi_read_counts(c, nullptr, &ne));
    PHICHK(phi_spectrum_count(c, &nd));
    if (n_reads) *n_reads = c->reads_count;
    if (n_bases) *n_bases = c->reads_bases;
    if (n_emitted) *n_emitted = (int64_t)ne;
    if (n_distinct) *n_distinct = (int64_t)nd;
    return PHI_OK;
}

int phi_hits_buffer(phi_ctx *c, void **d_hits, int64_t *n)
{
    if (!c || !d_hits || !n) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_hits_buffer before phi_set_graph");
    HIPCHK(hipSetDevice(c->device));
    PHICHK(phi_flush_reset(c));
    *d_hits = c->d_hit.p;
    *n = c->n_unique;
    return PHI_OK;
}

int phi_spectrum_export(phi_ctx *c, void **d_hashes, int64_t *n)
{
    if (!c || !d_hashes || !n) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_spectrum_export before phi_set_graph");
    HIPCHK(hipSetDevice(c->device));
    *d_hashes = nullptr; *n = 0;
    uint64_t cnt = 0;
    PHICHK(phi_sp_flush(c, &cnt));                            // waits for the stream
    PHICHK(phi_dev_ensure(c, c->d_export, (size_t)std::max<uint64_t>(cnt, 1) * 8));      // (an empty list still has an address)
    if (cnt) {
        HIPCHK(hipMemsetAsync(scalar(c, S_EXPORT), 0, 8, c->stream));
        phi_launch_spectrum_export(c->stream, c->d_sp_keys.as<uint64_t>(), (int64_t)c->sp_cap, c->d_export.as<uint64_t>(),
                                   (unsigned long long *)scalar(c, S_EXPORT));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *d_hashes = c->d_export.p;
    *n = (int64_t)cnt;
    return PHI_OK;
}

int phi_spectrum_import(phi_ctx *c, const void *d_hashes, int64_t n)
{
    if (!c || n < 0 || (n > 0 && !d_hashes)) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_spectrum_import before phi_set_graph");
    if (n == 0) return PHI_OK;
    HIPCHK(hipSetDevice(c->device));
    if (d_hashes == c->d_export.p) return phi_fail(c, PHI_ERR_INVALID, "phi_spectrum_import: pass a copy, not the export buffer");
    uint64_t in_set = 0;
    PHICHK(phi_sp_flush(c, &in_set));                         // the set made current first: what is imported goes straight into it
    PHICHK(sp_reserve(c, in_set, (uint64_t)n));
    phi_launch_spectrum_insert(c->stream, (const uint64_t *)d_hashes, n, c->d_sp_keys.as<uint64_t>(), c->sp_cap - 1,
                               c->d_sp_cnt.as<unsigned long long>(), c->d_u_keys.as<uint64_t>(), c->u_cap - 1, c->d_u_uid.as<uint32_t>(),
                               c->d_hit.as<uint8_t>(), (uint32_t *)scalar(c, S_ERR));
    HIPCHK(hipGetLastError());
    c->solved = false;
    return PHI_OK;
}

int phi_spectrum_set_size(phi_ctx *c, int64_t global_size)
{
    if (!c || global_size < 0) return PHI_ERR_INVALID;
    c->spectrum_override = global_size;
    return PHI_OK;
}

int phi_set_solve_budget(phi_ctx *c, int64_t max_dp_runs)
{
    if (!c) return PHI_ERR_INVALID;
    c->solve_budget = max_dp_runs > 0x7FFFFFFF ? 0x7FFFFFFF : max_dp_runs;
    c->solved = false;
    return PHI_OK;
}

int phi_solve(phi_ctx *c, phi_result *out)
{
    if (!c || !out) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_solve before phi_set_graph");
    HIPCHK(hipSetDevice(c->device));
    const int rc = phi_solve_impl(c);
    phi_pool_flush(c->device);                              // (what the index build and this solve let go and nobody took again)
    if (rc) return rc;
    *out = c->result;
    return PHI_OK;
}

int phi_path_sequence(phi_ctx *c, char *buf, int64_t cap)
{
    if (!c || (!buf && cap > 0)) return PHI_ERR_INVALID;
    if (!c->solved) return phi_fail(c, PHI_ERR_STATE, "phi_path_sequence before phi_solve");
    if (cap < c->result.hap_len) return phi_fail(c, PHI_ERR_INVALID, "buffer too small: need %lld bytes", (long long)c->result.hap_len);
    // (where every vertex of the path lands, then the copies by all host threads: 5.8 M vertices and 176 MB at chromosome scale)
    const int64_t np = (int64_t)c->h_path_vtx.size();
    std::vector<int64_t> at((size_t)np + 1, 0);
    for (int64_t i = 0; i < np; i++) { const int32_t v = c->h_path_vtx[(size_t)i]; at[(size_t)i + 1] = at[(size_t)i] + (c->h_seq_off[v + 1] - c->h_seq_off[v]); }
    phi_parallel_chunks(np, 1 << 15, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; i++) {
            const int32_t v = c->h_path_vtx[(size_t)i];
            memcpy(buf + at[(size_t)i], c->h_seq.data() + c->h_seq_off[v], (size_t)(at[(size_t)i + 1] - at[(size_t)i]));      // original case, :1580
        }
    });
    return PHI_OK;
}

int phi_sketch(phi_ctx *c, const char *bases, const int64_t *seq_off, int64_t n_seq, int32_t k, int32_t w,
               uint64_t *out_hash, int64_t *out_pos, int32_t *out_seq, int64_t cap, int64_t *n_out)
{
    if (!c || !n_out || n_seq < 0 || (n_seq > 0 && !seq_off)) return PHI_ERR_INVALID;
    if (k < 1 || k > PHI_MAX_K || w < 1 || w > PHI_MAX_W) return phi_fail(c, PHI_ERR_INVALID, "k or w out of range");
    *n_out = 0;
    if (n_seq == 0) return PHI_OK;
    for (int64_t r = 0; r < n_seq; r++)
        if (seq_off[r + 1] < seq_off[r]) return phi_fail(c, PHI_ERR_INVALID, "seq_off not monotone");
    const int64_t n_bases = seq_off[n_seq] - seq_off[0];
    if (n_bases == 0) return PHI_OK;
    if (!bases) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    DevBuf dB, dO, dW, dS, dH, dP;
    int rc = PHI_OK;
    std::vector<int64_t> off(seq_off, seq_off + n_seq + 1);
    for (auto &o : off) o -= seq_off[0];
    do {
        if ((rc = phi_dev_ensure(c, dB, (size_t)n_bases + 64))) break;
        if ((rc = phi_dev_ensure(c, dO, (size_t)(n_seq + 1) * 8))) break;
        const int64_t n_words = (n_bases + 31) / 32;
        if ((rc = phi_dev_ensure(c, dW, (size_t)(n_words + 2) * 8))) break;
        const size_t n_sw = (size_t)(n_bases / 64 + 2);
        if ((rc = phi_dev_ensure(c, dS, n_sw * 8))) break;
        if ((rc = phi_hip_check(c, hipMemcpyAsync(dB.p, bases + seq_off[0], (size_t)n_bases, hipMemcpyHostToDevice, c->stream), "H2D bases"))) break;
        if ((rc = phi_hip_check(c, hipMemcpyAsync(dO.p, off.data(), (size_t)(n_seq + 1) * 8, hipMemcpyHostToDevice, c->stream), "H2D offsets"))) break;
        if ((rc = phi_hip_check(c, hipMemsetAsync(dW.as<uint64_t>() + n_words, 0, 16, c->stream), "memset"))) break;
        if ((rc = phi_hip_check(c, hipMemsetAsync(dS.p, 0, n_sw * 8, c->stream), "memset"))) break;
        if ((rc = phi_hip_check(c, hipMemsetAsync(scalar(c, S_NBAD), 0, 8, c->stream), "memset"))) break;
        phi_launch_mark_starts(c->stream, dO.as<int64_t>(), n_seq, dS.as<unsigned long long>());
        phi_launch_pack_ascii(c->stream, dB.as<uint8_t>(), n_bases, dW.as<uint64_t>(), n_words, nullptr,
                              (unsigned long long *)scalar(c, S_NBAD));
        if ((rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "sync"))) break;
        uint64_t n_bad = 0;
        if ((rc = phi_hip_check(c, phi_copy_sync(c, &n_bad, scalar(c, S_NBAD), 8, hipMemcpyDeviceToHost), "D2H"))) break;
        int64_t total = 0;
        if ((rc = sketch_records(c, dW.as<uint64_t>(), dS.as<unsigned long long>(), n_bases, k, w,
                                 (n_bad || k > PHI_MAX_K_PACKED) ? dB.as<uint8_t>() : nullptr, dH, dP, &total))) break;
        if ((rc = phi_sync_check(c))) break;
        *n_out = total;
        if (cap >= total && total > 0) {
            std::vector<int64_t> gpos((size_t)total);
            if (out_hash && (rc = phi_hip_check(c, phi_copy_sync(c, out_hash, dH.p, (size_t)total * 8, hipMemcpyDeviceToHost), "D2H hash"))) break;
            if ((rc = phi_hip_check(c, phi_copy_sync(c, gpos.data(), dP.p, (size_t)total * 8, hipMemcpyDeviceToHost), "D2H pos"))) break;
            int64_t s = 0;
            for (int64_t i = 0; i < total; i++) {
                while (off[s + 1] <= gpos[i]) s++;
                if (out_pos) out_pos[i] = gpos[i] - off[s];
                if (out_seq) out_seq[i] = (int32_t)s;
            }
        }
    } while (0);
    dev_free(dB); dev_free(dO); dev_free(dW); dev_free(dS); dev_free(dH); dev_free(dP);
    // a failed stand-alone sketch must not poison later calls on this context
    (void)phi_memset_sync(c, c->d_scalars.p, 0, 16);
    return rc;
}

int phi_walk_minimizers(phi_ctx *c, int32_t walk, uint64_t *out_hash, int64_t *out_pos, int64_t cap, int64_t *n_out)
{
    if (!c || !n_out) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_walk_minimizers before phi_set_graph");
    if (walk < 0 || walk >= c->n_walks) return phi_fail(c, PHI_ERR_INVALID, "walk out of range");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->h_n_minimizers[walk];
    *n_out = n;
    if (cap < n || n == 0) return PHI_OK;
    // the records of a walk are never stored: expand the classes of its entries, in entry order
    const int64_t e_lo = c->h_walk_off[walk], e_hi = c->h_walk_off[walk + 1], ne = e_hi - e_lo;
    DevBuf lens, base, oh, op;
    struct Guard { DevBuf &a, &b, &d, &e; ~Guard() { dev_free(a); dev_free(b); dev_free(d); dev_free(e); } } guard{lens, base, oh, op};
    PHICHK(phi_dev_ensure(c, lens, (size_t)ne * 4));
    PHICHK(phi_dev_ensure(c, base, (size_t)(ne + 1) * 8));
    PHICHK(phi_dev_ensure(c, oh, (size_t)n * 8));
    PHICHK(phi_dev_ensure(c, op, (size_t)n * 8));
    phi_launch_entry_len_range(c->stream, c->d_walk_vtx.as<int32_t>(), c->d_vlen.as<int32_t>(), e_lo, ne, lens.as<int32_t>());
    {
        const int64_t nb = phi_scan_i32_num_blocks(ne);
        PHICHK(phi_dev_ensure(c, c->d_scan_blk64, (size_t)nb * 8));
        PHICHK(phi_dev_ensure(c, c->d_scan_blkoff, (size_t)(nb + 1) * 8));
        phi_launch_scan_i64(c->stream, lens.as<int32_t>(), ne, base.as<int64_t>(), c->d_scan_blk64.as<int64_t>(), c->d_scan_blkoff.as<int64_t>());
    }
    PhiExpandArgs X{};
    X.ent_cls = c->d_ent_cls.as<int32_t>(); X.e_lo = e_lo; X.e_hi = e_hi;
    X.cls_rec_off = c->d_cls_rec_off.as<int32_t>(); X.cls_rep = c->d_cls_rep.as<phi_ent_t>();
    X.rec_hash = c->d_rec_hash.as<uint64_t>(); X.rec_rel = c->d_rec_rel.as<int32_t>(); X.ent_base = base.as<int64_t>();
    X.out_hash = oh.as<uint64_t>(); X.out_pos = op.as<int64_t>();
    const int64_t nb = phi_expand_num_blocks(ne);
    PHICHK(phi_dev_ensure(c, c->d_blk_cnt, (size_t)nb * 4));
    PHICHK(phi_dev_ensure(c, c->d_blk_off, (size_t)(nb + 1) * 8));
    X.block_cnt = c->d_blk_cnt.as<int32_t>(); X.block_off = c->d_blk_off.as<int64_t>();
    phi_launch_expand_count(c->stream, X);
    PHICHK(phi_scan_counts_wide(c, c->d_blk_cnt.as<int32_t>(), nb, c->d_blk_off.as<int64_t>()));
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->d_blk_off.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (total != n) return phi_fail(c, PHI_ERR_DEVICE, "walk %d expands to %lld records, counted %lld (internal error)", walk, (long long)total, (long long)n);
    phi_launch_expand_write(c->stream, X, 0);
    HIPCHK(hipGetLastError());
    if (out_hash) HIPCHK(hipMemcpyAsync(out_hash, oh.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (out_pos) HIPCHK(hipMemcpyAsync(out_pos, op.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PHI_OK;
}

int phi_index_stats(phi_ctx *c, phi_index_info *out)
{
    if (!c || !out) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_index_stats before phi_set_graph");
    out->n_entries = c->n_entries; out->walk_bases = c->walk_bases;
    out->n_classes = c->n_cls; out->class_bases = c->cls_bases; out->n_class_records = c->n_rec;
    out->n_walk_minimizers = 0;
    for (int64_t m : c->h_n_minimizers) out->n_walk_minimizers += m;
    out->n_distinct_minimizers = c->n_unique;
    out->sketch_gpu_ms = c->index_gpu_ms;
    return PHI_OK;
}

int phi_solve_stats(phi_ctx *c, phi_solve_info *out)
{
    if (!c || !out) return PHI_ERR_INVALID;
    if (!c->solved) return phi_fail(c, PHI_ERR_STATE, "phi_solve_stats before phi_solve");
    memset(out, 0, sizeof *out);
    out->n_dp_anchors = c->n_dp;
    out->n_events = c->dp_events ? c->n_ev : c->n_entries;
    out->n_steps = c->dp_events ? c->n_k : c->n_vtx;
    out->dp_mode = !c->dp_events ? 0 : !c->dp_blocks ? 1 : c->dp_cls ? 3 : 2;
    out->n_blocks = c->dp_events && c->dp_blocks ? c->n_blk : 0;
    if (out->dp_mode == 3 && c->n_blk > 0) {
        HIPCHK(hipSetDevice(c->device));
        std::vector<int32_t> n((size_t)c->n_blk);
        HIPCHK(phi_copy_sync(c, n.data(), c->d_blk_ncls.p, n.size() * 4, hipMemcpyDeviceToHost));
        int64_t sum = 0;
        for (int32_t v : n) { sum += v; out->max_classes = std::max(out->max_classes, v); }
        out->mean_classes = (double)sum / (double)n.size();
    }
    return PHI_OK;
}

int phi_walk_sharing(phi_ctx *c, int64_t *hist, int32_t cap, int64_t *n_distinct)
{
    if (!c || !hist) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_walk_sharing before phi_set_graph");
    if (cap < c->n_walks + 1) return phi_fail(c, PHI_ERR_INVALID, "phi_walk_sharing: hist needs n_walks + 1 entries");
    HIPCHK(hipSetDevice(c->device));
    DevBuf last, cnt, dh;
    int rc = PHI_OK;
    do {
        if ((rc = phi_dev_ensure(c, last, c->u_cap * 4))) break;
        if ((rc = phi_dev_ensure(c, cnt, c->u_cap * 4))) break;
        if ((rc = phi_dev_ensure(c, dh, (size_t)(c->n_walks + 1) * 8))) break;
        phi_launch_fill_u32(c->stream, last.as<uint32_t>(), (int64_t)c->u_cap, 0xFFFFFFFFu);
        if ((rc = phi_hip_check(c, hipMemsetAsync(cnt.p, 0, c->u_cap * 4, c->stream), "memset"))) break;
        if ((rc = phi_hip_check(c, hipMemsetAsync(dh.p, 0, (size_t)(c->n_walks + 1) * 8, c->stream), "memset"))) break;
        for (int32_t h = 0; h < c->n_walks; h++)
            phi_launch_share_count_cls(c->stream, c->d_ent_cls.as<int32_t>(), c->h_walk_off[h], c->h_walk_off[h + 1], c->d_cls_rec_off.as<int32_t>(),
                                       c->d_rec_slot.as<uint32_t>(), h, last.as<int32_t>(), cnt.as<int32_t>());
        phi_launch_share_hist(c->stream, c->d_u_keys.as<uint64_t>(), (int64_t)c->u_cap, cnt.as<int32_t>(),
                              dh.as<unsigned long long>());
        if ((rc = phi_hip_check(c, hipGetLastError(), "launch"))) break;
        if ((rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "synchronize"))) break;
        std::vector<unsigned long long> hh(c->n_walks + 1);
        if ((rc = phi_hip_check(c, phi_copy_sync(c, hh.data(), dh.p, hh.size() * 8, hipMemcpyDeviceToHost), "D2H"))) break;
        for (int32_t i = 0; i <= c->n_walks; i++) hist[i] = (int64_t)hh[i];
        if (n_distinct) *n_distinct = c->n_unique;
    } while (0);
    dev_free(last); dev_free(cnt); dev_free(dh);
    return rc;
}

int phi_kept_anchors(phi_ctx *c, uint64_t *out_hash, int32_t *out_walk, int32_t *out_t0, int32_t *out_t1, int64_t cap,
                     int64_t *n_out)
{
    if (!c || !n_out) return PHI_ERR_INVALID;
    if (!c->solved) return phi_fail(c, PHI_ERR_STATE, "phi_kept_anchors before phi_solve");
    const int64_t n = c->n_kept;
    if (cap >= n) PHICHK(phi_host_anchors(c));                 // (a large model is solved on the device copy alone)
    *n_out = n;
    if (cap < n) return PHI_OK;
    if (out_hash && n && (int64_t)c->h_kept_hash.size() != n) {
        // the hashes stayed on the device (phi_solve does not need them): fetch them now
        HIPCHK(hipSetDevice(c->device));
        c->h_kept_hash.resize(n);
        // hash of a dense minimiser id = hash of its first class record
        std::vector<uint64_t> id_hash((size_t)c->n_unique);
        PHICHK(phi_dev_ensure(c, c->d_list2, (size_t)std::max<int64_t>(c->n_unique, 1) * 8));
        phi_launch_gather_u64(c->stream, c->d_rec_hash.as<uint64_t>(), c->d_u_replist.as<int32_t>(), c->n_unique, c->d_list2.as<uint64_t>());
        HIPCHK(hipMemcpyAsync(id_hash.data(), c->d_list2.p, (size_t)c->n_unique * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int64_t i = 0; i < n; i++) c->h_kept_hash[i] = id_hash[c->h_kept[i].slot];
    }
    for (int64_t i = 0; i < n; i++) {
        const PhiAnchorHost &a = c->h_kept[i];
        const int32_t h = phi_entry_walk(c, a.e0);
        if (out_hash) out_hash[i] = c->h_kept_hash[i];
        if (out_walk) out_walk[i] = h;
        if (out_t0) out_t0[i] = (int32_t)((int64_t)a.e0 - c->h_walk_off[h]);
        if (out_t1) out_t1[i] = (int32_t)((int64_t)a.e1 - c->h_walk_off[h]);
    }
    return PHI_OK;
}

int phi_host_register(phi_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterPortable));   // every GPU of a --devices run copies from it
    return PHI_OK;
}

int phi_host_unregister(phi_ctx *c, void *p)
{
    if (!c || !p) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipHostUnregister(p));
    return PHI_OK;
}

int phi_device_synchronize(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipGetLastError());
    return PHI_OK;
}

int phi_prof_enable(phi_ctx *c, int on)
{
    if (!c) return PHI_ERR_INVALID;
    c->prof = on != 0;
    c->prof_period = on > 0 ? on : 1;                   // on = n: every n-th sketch launch is bracketed
    c->prof_seq = 0;
    if (c->prof) {
        // create the event pairs of the next launches now, not inside the region being timed
        HIPCHK(hipSetDevice(c->device));
        while (c->prof_events.size() < c->prof_used + 256) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            c->prof_events.emplace_back(a, b);
        }
    }
    return PHI_OK;
}

int phi_prof_read(phi_ctx *c, int64_t *n_launches, double *total_ms, int64_t *total_bases)
{
    if (!c) return PHI_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->prof_used; i++) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
        c->prof_ms_done += ms;
        c->prof_n_done++;
    }
    c->prof_used = 0;
    if (n_launches) *n_launches = c->prof_n_done;
    if (total_ms) *total_ms = c->prof_ms_done;
    if (total_bases) *total_bases = c->prof_bases;
    c->prof_n_done = 0; c->prof_ms_done = 0.0; c->prof_bases = 0;
    return PHI_OK;
}

}  // extern "C"
