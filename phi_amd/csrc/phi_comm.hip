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

}  // extern "C"
