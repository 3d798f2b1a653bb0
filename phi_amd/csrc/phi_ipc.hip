// phi_ipc.hip -- the job's one exchange step between PROCESSES of one node without RCCL: peer-mapped hit vectors.
//
// The reference is one process (no counterpart).  Reads shard across GPUs: every rank holds the full
// walk-minimiser index and scores its own reads; before the solve the hit vectors (one byte per distinct walk minimiser)
// are ORed.  With one process per GPU (bench.py under torch.distributed.run, any MPI-style launcher) the RCCL all-reduce
// of phi_comm.hip costs 25-40 us of launch and proxy latency for a vector of 0.5-3 MB -- more than a GPU takes to SCORE a
// whole read set of the MHC configurations (17 us at C2), so a per-read-set exchange over RCCL caps 8-GPU scaling near 3x.
// Here every rank maps the other ranks' hit vectors into its own address space (hipIpcGetMemHandle / hipIpcOpenMemHandle:
// loads then travel over xGMI like loads of peer-enabled memory inside one process, phi_comm.hip phi_peers_*) and ONE
// kernel per rank and read set ORs them into its own vector.  Nothing on the host takes part in an exchange:
//
//   * ordering between the ranks is a FLAG per rank in device memory the peers have mapped too: the gather kernel first
//     publishes (step, which of my hit buffers holds it) with a system-scope release store -- its stream has waited for the
//     scoring of that read set --, then its workgroups spin on the peers' flags (system-scope acquire loads, a sleep
//     between two, a timeout that raises an error instead of hanging the GPU) and OR the peers' vectors with system-scope
//     loads.  OR is idempotent and monotone: a peer vector that is itself half way through its own gather only ever shows
//     bits of the union;
//   * the gather runs on a stream of its own, so the exchange of read set i overlaps the scoring of read set i + 1.  For
//     that the hit vector exists FOUR times in this mode (phi_reset_reads rotates instead of swapping; phi_ctx.h): the
//     vector of read set g is zeroed by the first scoring launch of read set g + 3, which waits (an event) for this
//     rank's gather g + 1 -- and that gather has seen every peer's flag g + 1, published after the peer's gather g had
//     read this rank's vector g.  No "done" message is needed;
//   * whatever looks at the hit vector (phi_solve, phi_reads_stats, phi_hits_buffer, ...) first makes the context's stream
//     wait for the last gather (phi_flush_reset).
//
// The ranks meet on the host only to set up and tear down (and once per job for the lists of novel read hashes, which
// only feed log counters): through a small POSIX shared-memory block whose name is the group's id -- 128 bytes made by
// one rank and handed to the others out of band, like phi_comm_unique_id.
//
// NOT measured across GPUs: no multi-GPU node has been available to any round.  tests/test_gpu_ipc.py runs two and three
// PROCESSES on one GPU (IPC handles work on the same device; every line but the xGMI hop itself).
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include "phi_ctx.h"
#include "phi_dev.h"

#define HIPCHK(call) do { int rc_ = phi_hip_check(c, (call), #call); if (rc_) return rc_; } while (0)
#define PHICHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

#define PHI_IPC_MAX 16
#define PHI_IPC_MAGIC 0x50484969u
#define PHI_IPC_FLAG_SLOTS 4                  // a peer is at most one step ahead: the flag of step s lives in slot s % 4

namespace {

struct IpcSlot {
    hipIpcMemHandle_t hit[PHI_HIT_RING];
    hipIpcMemHandle_t mbox;
    hipIpcMemHandle_t list;
    uint64_t list_version;                    // changes when the list buffer was allocated anew (a new handle)
    int64_t list_n;
    int64_t n_unique;
    int32_t device, pid;
};
struct IpcShm {
    uint32_t magic;
    int32_t n_ranks;
    std::atomic<int32_t> arrived;
    std::atomic<uint32_t> phase;
    std::atomic<int32_t> failed;
    IpcSlot slot[PHI_IPC_MAX];
};

struct IpcPeerArgs {
    const unsigned long long *flag[PHI_IPC_MAX - 1];             // the peer's PHI_IPC_FLAG_SLOTS flags
    const unsigned long long *hit[PHI_IPC_MAX - 1][PHI_HIT_RING];
};

// One launch per rank and read set (see the head of this file).  flag value = step << 2 | hit buffer of that step.
__global__ void __launch_bounds__(256) phi_ipc_gather_kernel(unsigned long long *__restrict__ mine, unsigned long long *my_flags,
                                                             unsigned long long my_value, IpcPeerArgs P, int n_peers, int64_t n_words,
                                                             unsigned long long step, unsigned long long timeout_ticks, uint32_t *err)
{
    __shared__ int s_idx[PHI_IPC_MAX];
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(my_flags + (step % PHI_IPC_FLAG_SLOTS), my_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if ((int)threadIdx.x < n_peers) {
        const unsigned long long *f = P.flag[threadIdx.x] + (step % PHI_IPC_FLAG_SLOTS);
        const unsigned long long t0 = wall_clock64();
        unsigned long long v;
        for (;;) {
            v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((v >> 2) == step) break;
            if (wall_clock64() - t0 > timeout_ticks) { s_fail = 1; break; }      // (every wave reaches an exit: a peer that never comes must not hang the GPU)
            __builtin_amdgcn_s_sleep(16);
        }
        s_idx[threadIdx.x] = (int)(v & 3ull);
    }
    __syncthreads();
    if (s_fail) {
        if (threadIdx.x == 0) atomicOr(err, 1u);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long v = 0;
        for (int r = 0; r < n_peers; r++) v |= __hip_atomic_load(P.hit[r][s_idx[r]] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v & ~mine[i]) mine[i] |= v;
    }
}

}  // namespace

struct PhiIpc {
    IpcShm *shm = nullptr;
    int rank = 0, n = 1;
    hipStream_t xstream = nullptr;
    hipEvent_t ev_scored = nullptr;
    hipEvent_t ev_gen[PHI_HIT_RING] = {nullptr, nullptr, nullptr, nullptr};     // behind the last gather of read-set generation g: ev_gen[g % 4]
    int64_t gen_of[PHI_HIT_RING] = {-1, -1, -1, -1};
    hipEvent_t ev_last = nullptr;                                                 // the last gather issued (what observers wait for)
    bool pending = false;
    uint64_t step = 0;
    unsigned long long *mbox = nullptr;       // my flags + (at word 8) the error word of the gather kernels
    void *peer_hit[PHI_IPC_MAX][PHI_HIT_RING] = {};
    void *peer_mbox[PHI_IPC_MAX] = {};
    void *peer_list[PHI_IPC_MAX] = {};
    uint64_t peer_list_version[PHI_IPC_MAX] = {};
    DevBuf d_list;                            // my list of novel read hashes, where the peers read it
    uint64_t list_version = 0;
    double timeout_s = 20.0;
};

namespace {

// all ranks meet (host): sense-reversing counter in the shared block; nonzero when some rank failed or did not come
int shm_barrier(PhiIpc *g, int fail, double timeout_s = 120.0)
{
    IpcShm *s = g->shm;
    if (fail) s->failed.store(fail);
    const uint32_t p = s->phase.load();
    if (s->arrived.fetch_add(1) + 1 == s->n_ranks) {
        s->arrived.store(0);
        s->phase.fetch_add(1);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (s->phase.load() == p) {
            if (++spins < 2000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { s->failed.store(PHI_ERR_STATE); break; }
        }
    }
    return s->failed.load();
}

void ipc_release(phi_ctx *c, PhiIpc *g)
{
    (void)hipSetDevice(c->device);
    if (g->xstream) (void)hipStreamSynchronize(g->xstream);
    for (int r = 0; r < PHI_IPC_MAX; r++) {
        for (int b = 0; b < PHI_HIT_RING; b++) if (g->peer_hit[r][b]) (void)hipIpcCloseMemHandle(g->peer_hit[r][b]);
        if (g->peer_mbox[r]) (void)hipIpcCloseMemHandle(g->peer_mbox[r]);
        if (g->peer_list[r]) (void)hipIpcCloseMemHandle(g->peer_list[r]);
    }
    if (g->ev_scored) (void)hipEventDestroy(g->ev_scored);
    for (int i = 0; i < PHI_HIT_RING; i++) if (g->ev_gen[i]) (void)hipEventDestroy(g->ev_gen[i]);
    if (g->xstream) (void)hipStreamDestroy(g->xstream);
    if (g->mbox) (void)hipFree(g->mbox);
    if (g->d_list.p) (void)hipFree(g->d_list.p);
    if (g->shm) (void)munmap(g->shm, sizeof(IpcShm));
    delete g;
}

}  // namespace

// (phi_abi.hip) everything that looks at the hit vector comes through phi_flush_reset: the context's stream waits for the last gather
int phi_ipc_wait_pending(phi_ctx *c)
{
    PhiIpc *g = c->ipc;
    if (!g || !g->pending) return PHI_OK;
    HIPCHK(hipStreamWaitEvent(c->stream, g->ev_last, 0));
    g->pending = false;
    return PHI_OK;
}

// (phi_abi.hip phi_reset_reads, entering read-set generation `gen`) the first scoring launch of this generation zeroes the
// hit buffer of generation gen - 3: this rank's gather of generation gen - 2 must have ended (see the head of this file)
int phi_ipc_before_generation(phi_ctx *c, int64_t gen)
{
    PhiIpc *g = c->ipc;
    if (!g) return PHI_OK;
    for (int64_t h = gen - 2; h >= gen - 3 && h >= 0; h--) {
        const int i = (int)(h % PHI_HIT_RING);
        if (g->gen_of[i] == h) { HIPCHK(hipStreamWaitEvent(c->stream, g->ev_gen[i], 0)); break; }   // (gathers end in order: the later one covers the earlier)
    }
    return PHI_OK;
}

extern "C" {

int phi_ipc_unique_id(void *id_out, size_t cap)
{
    if (!id_out || cap < PHI_COMM_ID_BYTES) return PHI_ERR_INVALID;
    static std::atomic<uint32_t> counter{0};
    char name[PHI_COMM_ID_BYTES];
    memset(name, 0, sizeof name);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(name, sizeof name, "/phi_ipc_%d_%u_%llx", (int)getpid(), counter.fetch_add(1), (unsigned long long)now);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return PHI_ERR_DEVICE;
    if (ftruncate(fd, (off_t)sizeof(IpcShm)) != 0) { close(fd); shm_unlink(name); return PHI_ERR_NOMEM; }
    void *p = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); return PHI_ERR_NOMEM; }
    IpcShm *s = new (p) IpcShm();
    s->n_ranks = 0; s->arrived.store(0); s->phase.store(0); s->failed.store(0);
    s->magic = PHI_IPC_MAGIC;
    munmap(p, sizeof(IpcShm));
    memcpy(id_out, name, PHI_COMM_ID_BYTES);
    return PHI_OK;
}

int phi_ipc_init(phi_ctx *c, const void *id, int32_t rank, int32_t n_ranks)
{
    if (!c || !id || n_ranks < 1 || n_ranks > PHI_IPC_MAX || rank < 0 || rank >= n_ranks) return PHI_ERR_INVALID;
    if (!c->have_graph) return phi_fail(c, PHI_ERR_STATE, "phi_ipc_init before phi_set_graph");
    if (c->ipc) return phi_fail(c, PHI_ERR_STATE, "phi_ipc_init: this context already belongs to a group");
    HIPCHK(hipSetDevice(c->device));
    char name[PHI_COMM_ID_BYTES + 1];
    memcpy(name, id, PHI_COMM_ID_BYTES); name[PHI_COMM_ID_BYTES] = 0;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return phi_fail(c, PHI_ERR_DEVICE, "phi_ipc_init: no shared block %s (is the id from phi_ipc_unique_id of a process on this host?)", name);
    void *p = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return phi_fail(c, PHI_ERR_NOMEM, "phi_ipc_init: mmap failed");
    PhiIpc *g = new (std::nothrow) PhiIpc();
    if (!g) { munmap(p, sizeof(IpcShm)); return phi_fail(c, PHI_ERR_NOMEM, "host allocation failed"); }
    g->shm = (IpcShm *)p; g->rank = rank; g->n = n_ranks;
    if (const char *e = getenv("PHI_IPC_TIMEOUT_S")) g->timeout_s = std::max(0.01, atof(e));
    int rc = PHI_OK;
    if (g->shm->magic != PHI_IPC_MAGIC) rc = phi_fail(c, PHI_ERR_INVALID, "phi_ipc_init: %s is not a group block", name);
    if (rank == 0) g->shm->n_ranks = n_ranks;
    // ---- what this rank offers: its four hit buffers (the ring of phi_reset_reads), its flags
    auto step = [&](hipError_t e, const char *what) { if (!rc && e != hipSuccess) rc = phi_hip_check(c, e, what); };
    if (!rc) rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    const size_t hit_bytes = (size_t)(c->n_unique / 8 + 1) * 8;
    if (!rc && c->hit_n != PHI_HIT_RING) {
        for (int i = 0; i < PHI_HIT_RING - 2 && !rc; i++) {
            rc = phi_dev_ensure(c, c->hit_extra[i], hit_bytes);
            if (!rc) step(phi_memset_sync(c, c->hit_extra[i].p, 0, hit_bytes), "hipMemset");
        }
        if (!rc) c->hit_n = PHI_HIT_RING;
    }
    // the flags: uncached device memory when the runtime shares such memory between processes, ordinary device memory otherwise
    // (the kernels read and write them with system-scope atomics either way)
    void *mb = nullptr;
    IpcSlot &me = g->shm->slot[rank];
    if (!rc) {
        if (hipExtMallocWithFlags(&mb, 256, hipDeviceMallocUncached) != hipSuccess || hipIpcGetMemHandle(&me.mbox, mb) != hipSuccess) {
            if (mb) (void)hipFree(mb);
            mb = nullptr;
            (void)hipGetLastError();
            step(hipMalloc(&mb, 256), "hipMalloc(flags)");
        }
    }
    g->mbox = (unsigned long long *)mb;
    if (!rc) step(phi_memset_sync(c, mb, 0, 256), "hipMemset");
    if (!rc) {
        // ring order: the current buffer, the one the next reset brings to the front, ... (phi_ctx.h)
        void *ring[PHI_HIT_RING] = {c->d_hit.p, c->alt.hit.p, c->hit_extra[0].p, c->hit_extra[1].p};
        for (int b = 0; b < PHI_HIT_RING; b++) step(hipIpcGetMemHandle(&me.hit[(c->hit_idx + b) % PHI_HIT_RING], ring[b]), "hipIpcGetMemHandle(hit vector)");
        step(hipIpcGetMemHandle(&me.mbox, mb), "hipIpcGetMemHandle(flags)");
        me.n_unique = c->n_unique; me.device = c->device; me.pid = (int32_t)getpid(); me.list_version = 0; me.list_n = 0;
    }
    step(hipStreamCreateWithFlags(&g->xstream, hipStreamNonBlocking), "hipStreamCreate");
    step(hipEventCreateWithFlags(&g->ev_scored, hipEventDisableTiming), "hipEventCreate");
    for (int i = 0; i < PHI_HIT_RING; i++) step(hipEventCreateWithFlags(&g->ev_gen[i], hipEventDisableTiming), "hipEventCreate");
    if (shm_barrier(g, rc)) { ipc_release(c, g); return rc ? rc : phi_fail(c, PHI_ERR_STATE, "another rank could not join the group"); }
    // ---- what the others offer
    for (int r = 0; r < n_ranks && !rc; r++) {
        if (r == rank) continue;
        const IpcSlot &o = g->shm->slot[r];
        if (o.n_unique != c->n_unique) { rc = phi_fail(c, PHI_ERR_INVALID, "the ranks of a group hold different graphs (%lld and %lld distinct walk minimisers)", (long long)c->n_unique, (long long)o.n_unique); break; }
        if (o.pid == (int32_t)getpid()) { rc = phi_fail(c, PHI_ERR_INVALID, "phi_ipc_*: ranks %d and %d are contexts of one process (use phi_peers_*)", rank, r); break; }
        for (int b = 0; b < PHI_HIT_RING && !rc; b++) step(hipIpcOpenMemHandle(&g->peer_hit[r][b], o.hit[b], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(hit vector)");
        step(hipIpcOpenMemHandle(&g->peer_mbox[r], o.mbox, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(flags)");
    }
    (void)hipGetLastError();
    if (shm_barrier(g, rc)) { ipc_release(c, g); return rc ? rc : phi_fail(c, PHI_ERR_STATE, "another rank could not map its peers"); }
    if (rank == 0) (void)shm_unlink(name);                   // every rank has it mapped: nothing is left behind if a rank dies from here on
    c->ipc = g;
    return PHI_OK;
}

int phi_ipc_info(const phi_ctx *c, int32_t *rank, int32_t *n_ranks)
{
    if (!c) return PHI_ERR_INVALID;
    if (rank) *rank = c->ipc ? c->ipc->rank : 0;
    if (n_ranks) *n_ranks = c->ipc ? c->ipc->n : 1;
    return PHI_OK;
}

// step 1 alone, asynchronous: the gather of this read set on the group's own stream (it overlaps the scoring of the next)
int phi_ipc_allreduce_hits(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PhiIpc *g = c->ipc;
    if (!g) return phi_fail(c, PHI_ERR_STATE, "no group: call phi_ipc_init first");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->n_unique;
    const int64_t n_words = n / 8 + 1;                         // (the vectors are allocated in whole words)
    g->step++;
    HIPCHK(hipEventRecord(g->ev_scored, c->stream));           // (behind this read set's scoring -- and behind an earlier gather an observer made the stream wait for)
    HIPCHK(hipStreamWaitEvent(g->xstream, g->ev_scored, 0));
    if (g->n > 1) {
        IpcPeerArgs P{};
        int np = 0;
        for (int r = 0; r < g->n; r++) {
            if (r == g->rank) continue;
            P.flag[np] = (const unsigned long long *)g->peer_mbox[r];
            for (int b = 0; b < PHI_HIT_RING; b++) P.hit[np][b] = (const unsigned long long *)g->peer_hit[r][b];
            np++;
        }
        const unsigned nb = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_words + 255) / 256, 1024));
        const unsigned long long ticks = (unsigned long long)(g->timeout_s * 1e8);      // wall_clock64: 100 MHz
        hipLaunchKernelGGL(phi_ipc_gather_kernel, dim3(nb), dim3(256), 0, g->xstream, c->d_hit.as<unsigned long long>(), g->mbox,
                           (unsigned long long)((g->step << 2) | (uint64_t)c->hit_idx), P, np, n_words, (unsigned long long)g->step, ticks,
                           (uint32_t *)(g->mbox + 8));
        HIPCHK(hipGetLastError());
    }
    const int i = (int)(c->sp_gen % PHI_HIT_RING);
    HIPCHK(hipEventRecord(g->ev_gen[i], g->xstream));
    g->gen_of[i] = c->sp_gen;
    g->ev_last = g->ev_gen[i];
    g->pending = true;
    c->solved = false;
    return PHI_OK;
}

// has a gather given up on a peer?  (waits for the group's stream)
int phi_ipc_check(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PhiIpc *g = c->ipc;
    if (!g) return PHI_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(g->xstream));
    uint32_t err = 0;
    HIPCHK(phi_copy_sync(c, &err, g->mbox + 8, 4, hipMemcpyDeviceToHost));
    if (err) return phi_fail(c, PHI_ERR_DEVICE, "phi_ipc: a peer's flag did not arrive within %.1f s (a rank died, or the ranks do not call the exchange equally often)", g->timeout_s);
    return PHI_OK;
}

// steps 1 + 2, once per job after the rank's last read batch: afterwards phi_solve gives the same result on every rank
int phi_ipc_exchange(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PHICHK(phi_ipc_allreduce_hits(c));
    PhiIpc *g = c->ipc;
    int rc = phi_ipc_check(c);
    // ---- step 2: the union of the read hashes that are not walk minimisers; every rank offers its list in a buffer the
    //      peers map (anew only when the buffer had to grow), the host barrier tells them it is there
    void *d_mine = nullptr;
    int64_t n_mine = 0;
    if (!rc) rc = phi_spectrum_export(c, &d_mine, &n_mine);    // waits for the stream
    IpcSlot &me = g->shm->slot[g->rank];
    if (!rc && n_mine) {
        if ((size_t)n_mine * 8 > g->d_list.cap) {
            if (g->d_list.p) { (void)hipFree(g->d_list.p); g->d_list = DevBuf{}; }
            rc = phi_dev_ensure(c, g->d_list, (size_t)n_mine * 8 * 2);
            if (!rc) rc = phi_hip_check(c, hipIpcGetMemHandle(&me.list, g->d_list.p), "hipIpcGetMemHandle(list)");
            if (!rc) me.list_version = ++g->list_version;
        }
        if (!rc) rc = phi_hip_check(c, hipMemcpyAsync(g->d_list.p, d_mine, (size_t)n_mine * 8, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
        if (!rc) rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    me.list_n = rc ? 0 : n_mine;
    if (shm_barrier(g, rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "a peer failed in the exchange");
    for (int r = 0; r < g->n && !rc; r++) {
        if (r == g->rank) continue;
        const IpcSlot &o = g->shm->slot[r];
        if (o.list_n <= 0) continue;
        if (o.list_version != g->peer_list_version[r]) {
            if (g->peer_list[r]) { (void)hipIpcCloseMemHandle(g->peer_list[r]); g->peer_list[r] = nullptr; }
            rc = phi_hip_check(c, hipIpcOpenMemHandle(&g->peer_list[r], o.list, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle(list)");
            if (!rc) g->peer_list_version[r] = o.list_version;
        }
        if (!rc) rc = phi_spectrum_import(c, g->peer_list[r], o.list_n);
    }
    if (!rc) rc = phi_hip_check(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");      // the peers' lists are free again
    if (shm_barrier(g, rc)) return rc ? rc : phi_fail(c, PHI_ERR_STATE, "a peer failed in the exchange");
    return PHI_OK;
}

// collective (the ranks meet once more so that nobody unmaps memory a peer's gather still reads); also done by phi_ctx_destroy
int phi_ipc_destroy(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PhiIpc *g = c->ipc;
    if (!g) return PHI_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(g->xstream);
    (void)hipStreamSynchronize(c->stream);
    (void)shm_barrier(g, 0, 10.0);
    c->ipc = nullptr;
    c->hit_n = 2;                                              // (the two extra hit buffers stay allocated, unused)
    ipc_release(c, g);
    return PHI_OK;
}

}  // extern "C"
