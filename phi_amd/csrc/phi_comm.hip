// phi_comm.hip -- the job's one exchange step, inside the C ABI: RCCL over xGMI.
//
// The reference is one process (no counterpart).  Reads shard across GPUs; every rank
// holds the full walk-minimiser index, whose dense minimiser ids are the same on every rank, and scores
// its own reads.  Before phi_solve the ranks exchange ONCE:
//   1. ncclAllReduce(MAX, uint8) of the hit vector (one byte per distinct walk minimiser), in place;
//   2. the distinct read hashes that are NOT walk minimisers (those that are, are the hit flags just
//      reduced): ncclAllGather of the list sizes, ncclAllGather of the lists padded to the longest,
//      then every rank inserts the other ranks' lists -- so |Sp_R| (ILP_index.cpp:641) and the
//      filtered / retained counters (:738-743) are those of the whole read set on every rank.
// Both run on the context's stream.  Message sizes are MBs (C2-C4) to ~100 MB (C5): latency-bound on
// 7 x 153 GB/s xGMI links, so one fused call each, never split, never per batch.
//
// librccl is loaded on first use (dlopen), not at library load: a single-GPU run of the command line
// never pays for it (it is the largest shared object of the ROCm stack).
#include <dlfcn.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <rccl/rccl.h>
#include "phi_ctx.h"
#include "phi_dev.h"

#define HIPCHK(call) do { int rc_ = phi_hip_check(c, (call), #call); if (rc_) return rc_; } while (0)
#define PHICHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

namespace {

struct RcclApi {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

std::mutex g_api_mu;
RcclApi g_api;

// resolve the six entry points once per process; an error text is kept for the caller
const RcclApi *rccl_api(std::string *why)
{
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.dl) return &g_api;
    if (!g_api.why.empty()) { if (why) *why = g_api.why; return nullptr; }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *dl = nullptr;
    for (const char *n : names)
        if ((dl = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;   // (a host that already loaded RCCL -- torch -- shares that copy by its soname)
    if (!dl) { g_api.why = std::string("librccl not found: ") + dlerror(); if (why) *why = g_api.why; return nullptr; }
    RcclApi a;
    a.dl = dl;
    bool ok = true;
    auto sym = [&](const char *n) { void *p = dlsym(dl, n); ok = ok && p; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(dl); g_api.why = "librccl lacks an expected entry point"; if (why) *why = g_api.why; return nullptr; }
    g_api = a;
    return &g_api;
}

}  // namespace

struct PhiComm {
    const RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, n = 1;
    DevBuf d_sizes, d_send, d_recv;
};

static int nccl_fail(phi_ctx *c, const RcclApi *api, ncclResult_t r, const char *what)
{
    return phi_fail(c, PHI_ERR_DEVICE, "%s: %s", what, api->GetErrorString(r));
}
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nccl_fail(c, api, r_, #call); } while (0)

extern "C" {

int phi_comm_unique_id(void *id_out, size_t cap)
{
    if (!id_out || cap < PHI_COMM_ID_BYTES) return PHI_ERR_INVALID;
    static_assert(PHI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "PHI_COMM_ID_BYTES is the size of ncclUniqueId");
    const RcclApi *api = rccl_api(nullptr);
    if (!api) return PHI_ERR_DEVICE;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return PHI_ERR_DEVICE;
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return PHI_OK;
}

int phi_comm_init(phi_ctx *c, const void *id, int32_t rank, int32_t n_ranks)
{
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return PHI_ERR_INVALID;
    if (c->comm) return phi_fail(c, PHI_ERR_STATE, "phi_comm_init: this context already has a communicator");
    std::string why;
    const RcclApi *api = rccl_api(&why);
    if (!api) return phi_fail(c, PHI_ERR_DEVICE, "%s", why.c_str());
    HIPCHK(hipSetDevice(c->device));
    PhiComm *pc = new (std::nothrow) PhiComm();
    if (!pc) return phi_fail(c, PHI_ERR_NOMEM, "host allocation failed");
    pc->api = api; pc->rank = rank; pc->n = n_ranks;
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = api->CommInitRank(&pc->comm, n_ranks, uid, rank);     // collective: every rank calls it
    if (r != ncclSuccess) { delete pc; return nccl_fail(c, api, r, "ncclCommInitRank"); }
    c->comm = pc;
    return PHI_OK;
}

int phi_comm_destroy(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PhiComm *pc = c->comm;
    if (!pc) return PHI_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (pc->comm) (void)pc->api->CommDestroy(pc->comm);
    DevBuf *bufs[] = {&pc->d_sizes, &pc->d_send, &pc->d_recv};
    for (DevBuf *b : bufs) if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    delete pc;
    c->comm = nullptr;
    return PHI_OK;
}

int phi_comm_info(const phi_ctx *c, int32_t *rank, int32_t *n_ranks)
{
    if (!c) return PHI_ERR_INVALID;
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (n_ranks) *n_ranks = c->comm ? c->comm->n : 1;
    return PHI_OK;
}

// step 1 alone: what a job that sends its batches in a loop calls after its last batch
int phi_comm_allreduce_hits(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_comm_allreduce_hits before phi_set_graph");
    PhiComm *pc = c->comm;
    if (!pc) return phi_fail(c, PHI_ERR_STATE, "no communicator: call phi_comm_init first");
    const RcclApi *api = pc->api;
    HIPCHK(hipSetDevice(c->device));
    void *d_hit = nullptr;
    int64_t n = 0;
    PHICHK(phi_hits_buffer(c, &d_hit, &n));                    // flushes a pending reset
    if (n > 0) NCCLCHK(api->AllReduce(d_hit, d_hit, (size_t)n, ncclUint8, ncclMax, pc->comm, c->stream));
    c->solved = false;
    return PHI_OK;
}

int phi_comm_exchange(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PHICHK(phi_comm_allreduce_hits(c));
    PhiComm *pc = c->comm;
    const RcclApi *api = pc->api;
    // ---- step 2: union of the read hashes that are not walk minimisers
    void *d_mine = nullptr;
    int64_t n_mine = 0;
    PHICHK(phi_spectrum_export(c, &d_mine, &n_mine));          // waits for the stream
    PHICHK(phi_dev_ensure(c, pc->d_sizes, (size_t)(pc->n + 1) * 8));
    int64_t *d_sizes = pc->d_sizes.as<int64_t>();
    HIPCHK(hipMemcpyAsync(d_sizes + pc->n, &n_mine, 8, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(api->AllGather(d_sizes + pc->n, d_sizes, 1, ncclInt64, pc->comm, c->stream));
    std::vector<int64_t> sizes((size_t)pc->n);
    HIPCHK(hipMemcpyAsync(sizes.data(), d_sizes, (size_t)pc->n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t mx = 0;
    for (int64_t s : sizes) {
        if (s < 0) return phi_fail(c, PHI_ERR_DEVICE, "spectrum exchange: negative list size (internal error)");
        mx = std::max(mx, s);
    }
    if (sizes[(size_t)pc->rank] != n_mine) return phi_fail(c, PHI_ERR_DEVICE, "spectrum exchange: own list size came back changed (internal error)");
    if (mx == 0 || pc->n == 1) return PHI_OK;
    PHICHK(phi_dev_ensure(c, pc->d_send, (size_t)mx * 8));
    PHICHK(phi_dev_ensure(c, pc->d_recv, (size_t)mx * 8 * (size_t)pc->n));
    if (n_mine) HIPCHK(hipMemcpyAsync(pc->d_send.p, d_mine, (size_t)n_mine * 8, hipMemcpyDeviceToDevice, c->stream));
    if (mx > n_mine) HIPCHK(hipMemsetAsync(pc->d_send.as<uint64_t>() + n_mine, 0xFF, (size_t)(mx - n_mine) * 8, c->stream));
    NCCLCHK(api->AllGather(pc->d_send.p, pc->d_recv.p, (size_t)mx, ncclUint64, pc->comm, c->stream));
    for (int r = 0; r < pc->n; r++)
        if (r != pc->rank && sizes[(size_t)r] > 0)
            PHICHK(phi_spectrum_import(c, pc->d_recv.as<uint64_t>() + (size_t)r * (size_t)mx, sizes[(size_t)r]));
    HIPCHK(hipStreamSynchronize(c->stream));                   // the receive buffer may be reused by the next exchange
    return PHI_OK;
}


// ---- the same exchange for the contexts of ONE process (one host thread per GPU: `PHI --devices`), without RCCL.
//
// An 8-rank ncclAllReduce of a 0.5-3 MB vector is tens of microseconds of launch and proxy latency -- as long as a GPU
// takes to SCORE a whole read set of the MHC configurations (22 us at C2).  Contexts of one process can do better: every
// GPU's hit vector is an address the others can load from (hipDeviceEnablePeerAccess: xGMI), OR is idempotent and
// monotone, so ONE kernel per GPU ORs the peers' vectors into its own, in place -- a peer vector that is itself half
// way through its update only ever shows bits of the union.  Ordering comes from HIP events, which cross devices inside a
// process: "my scoring is done" before the gather (recorded, then waited for by every peer's stream), "my gather is done"
// after it (so that no GPU resets a vector a peer still reads).  The host threads meet twice per exchange, at a
// barrier of their own, only to know that the events they are about to wait for have been recorded.
struct PhiPeers {
    int n = 0;
    std::vector<phi_ctx *> ctx;
    std::vector<hipEvent_t> ready, done;
    std::vector<void *> hits;                        // current hit vector of every rank
    std::vector<void *> sp_list;                     // exported spectrum list of every rank
    std::vector<int64_t> sp_n, n_unique;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t phase = 0;
    int failed = 0;
    // all n threads meet; returns nonzero when some rank raised `fail` before or at this barrier
    // (a rank that returned early, died or hangs must not leave the others waiting for ever: the wait is timed, and a barrier
    //  that timed out -- like a failure any rank reports -- fails every later barrier of the group: make a new group)
    int barrier(int fail)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (fail) failed = fail;
        const uint64_t p = phase;
        if (++arrived == n) { arrived = 0; phase++; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return phase != p; })) { failed = PHI_ERR_STATE; arrived = 0; phase++; cv.notify_all(); }
        return failed;
    }
};

namespace {
#define PHI_MAX_PEERS 16
struct PeerPtrs { const unsigned long long *p[PHI_MAX_PEERS]; };
// out[i] |= OR over the peers' vectors, eight flags at a time
__global__ void __launch_bounds__(256) phi_or_gather_kernel(unsigned long long *__restrict__ mine, PeerPtrs peers, int n_peers, int64_t n_words)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long v = 0;
#pragma unroll 4
        for (int r = 0; r < n_peers; r++) v |= __builtin_nontemporal_load(peers.p[r] + i);
        if (v & ~mine[i]) mine[i] |= v;
    }
}
}  // namespace

int phi_peers_create(int32_t n_ranks, void **group)
{
    if (!group || n_ranks < 1 || n_ranks > PHI_MAX_PEERS) return PHI_ERR_INVALID;
    PhiPeers *g = new (std::nothrow) PhiPeers();
    if (!g) return PHI_ERR_NOMEM;
    g->n = n_ranks;
    g->ctx.assign((size_t)n_ranks, nullptr);
    g->ready.assign((size_t)n_ranks, nullptr); g->done.assign((size_t)n_ranks, nullptr);
    g->hits.assign((size_t)n_ranks, nullptr); g->sp_list.assign((size_t)n_ranks, nullptr);
    g->sp_n.assign((size_t)n_ranks, 0); g->n_unique.assign((size_t)n_ranks, 0);
    *group = g;
    return PHI_OK;
}

// every rank's thread calls it once, after phi_set_graph (collective: the threads meet inside)
int phi_peers_join(phi_ctx *c, void *group, int32_t rank)
{
    PhiPeers *g = (PhiPeers *)group;
    if (!c || !g || rank < 0 || rank >= g->n) return PHI_ERR_INVALID;
    int rc = PHI_OK;
    if (!c->have_graph) rc = phi_fail(c, PHI_ERR_STATE, "phi_peers_join before phi_set_graph");
    if (!rc && hipSetDevice(c->device) != hipSuccess) rc = PHI_ERR_DEVICE;
    if (!rc) {
        g->ctx[(size_t)rank] = c;
        g->n_unique[(size_t)rank] = c->n_unique;
        if (hipEventCreateWithFlags(&g->ready[(size_t)rank], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->done[(size_t)rank], hipEventDisableTiming) != hipSuccess) rc = phi_fail(c, PHI_ERR_DEVICE, "event creation failed");
    }
    if (g->barrier(rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "another rank could not join the peer group");
    // every GPU may load from every other (the same device twice -- two contexts on one GPU, as the tests run it -- needs nothing)
    for (int r = 0; r < g->n && !rc; r++) {
        const int dev = g->ctx[(size_t)r]->device;
        if (g->n_unique[(size_t)r] != c->n_unique) { rc = phi_fail(c, PHI_ERR_INVALID, "the ranks of a peer group hold different graphs"); break; }
        if (dev == c->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c->device, dev) != hipSuccess || !can) { rc = phi_fail(c, PHI_ERR_DEVICE, "GPU %d cannot address GPU %d", c->device, dev); break; }
        const hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = phi_fail(c, PHI_ERR_DEVICE, "hipDeviceEnablePeerAccess(%d): %s", dev, hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if (g->barrier(rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "another rank could not reach its peers");
    c->peers = g; c->peer_rank = rank;
    return PHI_OK;
}

// collective over the group's threads, once per read set: hit vectors ORed in place, then the lists of read hashes that are
// no walk minimisers imported from every peer (they are read where they lie).  Afterwards phi_solve gives the same result
// on every rank, as after phi_comm_exchange.  hits_only: step 1 alone (what a job times per read set).
static int peers_exchange(phi_ctx *c, bool hits_only)
{
    if (!c) return PHI_ERR_INVALID;
    PhiPeers *g = c->peers;
    if (!g) return phi_fail(c, PHI_ERR_STATE, "no peer group: call phi_peers_join first");
    const int me = c->peer_rank;
    int rc = PHI_OK;
    void *d_hit = nullptr;
    int64_t n = 0;
    if (hipSetDevice(c->device) != hipSuccess) rc = PHI_ERR_DEVICE;
    if (!rc) rc = phi_hits_buffer(c, &d_hit, &n);
    if (!rc) {
        g->hits[(size_t)me] = d_hit;
        rc = phi_hip_check(c, hipEventRecord(g->ready[(size_t)me], c->stream), "hipEventRecord");
    }
    if (g->barrier(rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "a peer failed before the exchange");
    // ---- step 1: OR of the hit vectors.  From here to the phase's last barrier nothing returns: whatever fails is carried in
    //      rc to the next barrier, which every rank reaches (a rank that left between two barriers would leave the others waiting)
    PeerPtrs pp{};
    int np = 0;
    for (int r = 0; r < g->n; r++) {
        if (r == me) continue;
        if (!rc) rc = phi_hip_check(c, hipStreamWaitEvent(c->stream, g->ready[(size_t)r], 0), "hipStreamWaitEvent");
        pp.p[np++] = (const unsigned long long *)g->hits[(size_t)r];
    }
    const int64_t n_words = n / 8 + 1;                         // (the vectors are allocated in whole words)
    if (!rc && np && n > 0) {
        const unsigned nb = (unsigned)std::min<int64_t>((n_words + 255) / 256, 2048);
        hipLaunchKernelGGL(phi_or_gather_kernel, dim3(nb), dim3(256), 0, c->stream, (unsigned long long *)d_hit, pp, np, n_words);
        rc = phi_hip_check(c, hipGetLastError(), "phi_or_gather_kernel");
    }
    c->solved = false;
    if (hits_only) {
        if (!rc) rc = phi_hip_check(c, hipEventRecord(g->done[(size_t)me], c->stream), "hipEventRecord");
        if (g->barrier(rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "a peer failed in the exchange");
        for (int r = 0; r < g->n; r++) if (r != me) HIPCHK(hipStreamWaitEvent(c->stream, g->done[(size_t)r], 0));   // nobody resets a vector a peer still reads (no barrier follows: returning is safe)
        return PHI_OK;
    }
    // ---- step 2: the union of the read hashes that are not walk minimisers
    void *d_mine = nullptr;
    int64_t n_mine = 0;
    if (!rc) rc = phi_spectrum_export(c, &d_mine, &n_mine);    // waits for the stream
    if (!rc && n_mine) {
        // a copy the peers read: importing THEIR lists may regrow this context's set, which goes through the export buffer
        rc = phi_dev_ensure(c, c->d_peer_send, (size_t)n_mine * 8);
        if (!rc) rc = phi_hip_check(c, hipMemcpyAsync(c->d_peer_send.p, d_mine, (size_t)n_mine * 8, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
        if (!rc) rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        d_mine = c->d_peer_send.p;
    }
    g->sp_list[(size_t)me] = d_mine; g->sp_n[(size_t)me] = rc ? 0 : n_mine;
    if (g->barrier(rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "a peer failed in the exchange");
    for (int r = 0; r < g->n && !rc; r++)
        if (r != me && g->sp_n[(size_t)r] > 0) rc = phi_spectrum_import(c, g->sp_list[(size_t)r], g->sp_n[(size_t)r]);
    if (!rc) rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");   // the peers' lists and vectors are free again
    if (g->barrier(rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "a peer failed in the exchange");
    return PHI_OK;
}
int phi_peers_allreduce_hits(phi_ctx *c) { return peers_exchange(c, true); }
int phi_peers_exchange(phi_ctx *c) { return peers_exchange(c, false); }

// after every rank's last collective, by one thread
int phi_peers_destroy(void *group)
{
    PhiPeers *g = (PhiPeers *)group;
    if (!g) return PHI_OK;
    for (int r = 0; r < g->n; r++) {
        if (g->ctx[(size_t)r]) { (void)hipSetDevice(g->ctx[(size_t)r]->device); g->ctx[(size_t)r]->peers = nullptr; }
        if (g->ready[(size_t)r]) (void)hipEventDestroy(g->ready[(size_t)r]);
        if (g->done[(size_t)r]) (void)hipEventDestroy(g->done[(size_t)r]);
    }
    delete g;
    return PHI_OK;
}

}  // extern "C"
