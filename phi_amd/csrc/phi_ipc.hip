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
//   * ordering is FLAGS in device memory, never a host call, a HIP event or an extra launch on the scoring stream (measured:
//     an event record + a stream wait between two streams cost ~5 us of host time each -- four of them made a 17-us step
//     host-bound at 27 us --, and a one-thread signal kernel between two scoring launches cost the stream 5 us).  Per read
//     set a rank issues TWO plain launches: the scoring kernel, and on the group's own stream the GATHER kernel.  The
//     first wave of every scoring launch says "the exchanges issued before me have their read sets scored" (the stream is
//     in order: the scoring launches before it have ended) in a local flag; the gather's workgroups wait for that flag,
//     publish (step, which of my hit buffers holds it) for the peers with a system-scope release store, spin on the
//     peers' flags (system-scope acquire loads, a sleep between two, a timeout that raises an error instead of hanging the
//     GPU) and OR the peers' vectors into this rank's with system-scope loads.  OR is idempotent and monotone: a peer
//     vector that is itself half way through its own gather only ever shows bits of the union;
//   * the gather of read set i thus runs beside the scoring of read set i + 1.  For that the hit vector exists FOUR times
//     in this mode (phi_reset_reads rotates instead of swapping; phi_ctx.h): the vector of read set g is zeroed by the
//     waves of the first scoring launch of read set g + 3, each of which first makes sure (one load) that this rank's
//     gather g + 1 has ended -- and that gather has seen every peer's flag g + 1, published after the peer's gather g had
//     read this rank's vector g.  No "done" message is needed, and a rank can run at most two read sets ahead of a peer;
//   * whatever looks at the hit vector (phi_solve, phi_reads_stats, phi_hits_buffer, ...) first puts a one-thread kernel on
//     the context's stream that says "scored" for the last exchange and waits for its gather's "done" flag (phi_flush_reset).
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

#define MB_ERR PHI_MB_ERR
#define MB_SCORED PHI_MB_SCORED
#define MB_GATHERED PHI_MB_GATHERED
#define MB_BLOCKS PHI_MB_BLOCKS

// Spin on a flag until it reaches `want`.  The polls are RELAXED loads at agent / system scope: they go to the coherence point
// without touching the caches.  No acquire follows: everything the gather reads behind a flag it reads with scoped atomic
// loads as well (issued after the poll that saw the flag has returned), so no cache needs invalidating.  An acquire LOAD in the
// loop would invalidate this XCD's L2 with every poll (gfx942 / gfx950: the L2s of the XCDs are not coherent with each other,
// agent-scope acquire = buffer_inv sc1) -- measured: 64 lanes polling that way beside the scoring kernel made it five times
// slower; one acquire fence per workgroup behind the loop still cost the scoring kernel 15 %.
__device__ __forceinline__ bool spin_until(const unsigned long long *f, unsigned long long want, unsigned long long timeout_ticks, bool system)
{
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long v = system ? __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                            : __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) return true;
        if (wall_clock64() - t0 > timeout_ticks) return false;      // (every wave reaches an exit: a flag that never comes must not hang the GPU)
        __builtin_amdgcn_s_sleep(32);
    }
}

// On the context's stream, before something looks at the hit vector: "everything up to exchange `step` is scored" (no
// scoring launch may follow that would say so), then the gather of that exchange has ended.  One thread.
__global__ void phi_ipc_wait_kernel(unsigned long long *mb, unsigned long long step, unsigned long long timeout_ticks)
{
    __hip_atomic_store(mb + MB_SCORED, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (!spin_until(mb + MB_GATHERED, step, timeout_ticks, false)) atomicOr((uint32_t *)(mb + MB_ERR), 4u);
}

// On the context's stream, ranks that share a GPU only (phi_ipc_launch_args): the gather `need` has ended.  One thread.
__global__ void phi_ipc_need_kernel(unsigned long long *mb, unsigned long long need, unsigned long long timeout_ticks)
{
    if (!spin_until(mb + MB_GATHERED, need, timeout_ticks, false)) atomicOr((uint32_t *)(mb + MB_ERR), 2u);
}

// One launch per rank and read set, on the group's own stream (see the head of this file).
__global__ void __launch_bounds__(256) phi_ipc_gather_kernel(unsigned long long *mine, unsigned long long *mb,
                                                             unsigned long long my_value, IpcPeerArgs P, int n_peers, int64_t n_words,
                                                             unsigned long long step, unsigned long long timeout_ticks)
{
    __shared__ int s_idx[PHI_IPC_MAX];
    __shared__ int s_fail;
    if (threadIdx.x == 0) {
        s_fail = 0;
        if (!spin_until(mb + MB_SCORED, step, timeout_ticks, false)) s_fail = 1;       // this rank's own scoring of the step
        if (blockIdx.x == 0) __hip_atomic_store(mb + (step % PHI_IPC_FLAG_SLOTS), my_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if ((int)threadIdx.x < n_peers && !s_fail) {
        const unsigned long long *f = P.flag[threadIdx.x] + (step % PHI_IPC_FLAG_SLOTS);
        // (a peer is at most one step ahead, and a slot is reused four steps later: the step in the slot is this one's or an older one's)
        if (!spin_until(f, step << 2, timeout_ticks, true)) s_fail = 1;
        s_idx[threadIdx.x] = (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & 3ull);
    }
    __syncthreads();
    if (!s_fail) {
        // four words of every peer per thread and turn: their loads (a round trip over xGMI each) are all in flight before the first is used
        const int64_t T = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n_words; i0 += 4 * T) {
            unsigned long long v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t i = i0 + u * T;
                if (i < n_words)
                    for (int r = 0; r < n_peers; r++) v[u] |= __hip_atomic_load(P.hit[r][s_idx[r]] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t i = i0 + u * T;
                // (this rank's own vector: written by the scoring kernel on any XCD, read by later kernels on any XCD -- at the coherence point too)
                if (i < n_words && v[u] && (v[u] & ~__hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
                    __hip_atomic_fetch_or(mine + i, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (threadIdx.x == 0) {
        atomicOr((uint32_t *)(mb + MB_ERR), 1u);
    }
    // the last workgroup to end says so (also after a timeout: whoever waits for this gather must get on)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(mb + MB_BLOCKS, 1ull) + 1 == (unsigned long long)gridDim.x * step)
            __hip_atomic_store(mb + MB_GATHERED, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

struct PhiIpc {
    IpcShm *shm = nullptr;
    int rank = 0, n = 1;
    hipStream_t xstream = nullptr;
    bool shared_device = false;               // some peer runs on this rank's GPU (tests, rehearsals): see phi_ipc_launch_args
    uint64_t step = 0, waited = 0;            // exchanges issued; the last one the context's stream has been made to wait for
    uint64_t step_of_gen[PHI_HIT_RING] = {0, 0, 0, 0};      // the last exchange issued in read-set generation gen_tag[i] (i = generation % 4)
    int64_t gen_tag[PHI_HIT_RING] = {-1, -1, -1, -1};
    unsigned gather_blocks = 1;               // the same in every launch (the workgroup counter of the flag block counts on)
    unsigned long long *mbox = nullptr;       // my flag block (MB_* above)
    void *peer_hit[PHI_IPC_MAX][PHI_HIT_RING] = {};
    void *peer_mbox[PHI_IPC_MAX] = {};
    void *peer_list[PHI_IPC_MAX] = {};
    uint64_t peer_list_version[PHI_IPC_MAX] = {};
    DevBuf d_list;                            // my list of novel read hashes, where the peers read it
    uint64_t list_version = 0;
    double timeout_s = 20.0;
    unsigned long long ticks() const { return (unsigned long long)(timeout_s * 1e8); }      // wall_clock64: 100 MHz
};

namespace {

// all ranks meet (host): sense-reversing counter in the shared block; nonzero when some rank failed or did not come
int shm_barrier(PhiIpc *g, int fail, double timeout_s = 120.0)
{
    IpcShm *s = g->shm;
    if (fail) s->failed.store(fail);
    const uint32_t p = s->phase.load();
    if (s->arrived.fetch_add(1) + 1 == g->n) {
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
    if (!g || g->waited == g->step) return PHI_OK;
    hipLaunchKernelGGL(phi_ipc_wait_kernel, dim3(1), dim3(1), 0, c->stream, g->mbox, (unsigned long long)g->step, g->ticks());
    HIPCHK(hipGetLastError());
    g->waited = g->step;
    return PHI_OK;
}

// (phi_abi.hip phi_reset_reads) two resets with no read launch in between: the hit vector that comes to the front is zeroed
// by a launch of its own, which has no wave to look at the flags -- everything issued so far is waited for instead
int phi_ipc_before_generation(phi_ctx *c, int64_t) { return phi_ipc_wait_pending(c); }

// (phi_abi.hip, every read launch of a context in a group) what the launch's waves publish and what they make sure of before
// they zero the hit vector of read set gen - 3: this rank's last exchange of a read set <= gen - 2 has been gathered
void phi_ipc_launch_args(phi_ctx *c, PhiSketchArgs &A)
{
    PhiIpc *g = c->ipc;
    if (!g) return;
    A.ipc_mb = g->mbox;
    A.ipc_scored = g->step;
    A.ipc_need = 0;
    for (int64_t h = c->sp_gen - 2; h >= c->sp_gen - 3 && h >= 0; h--)
        if (g->gen_tag[h % PHI_HIT_RING] == h) { A.ipc_need = g->step_of_gen[h % PHI_HIT_RING]; break; }
    if (g->shared_device && A.q_clean && A.ipc_need) {
        // Ranks that SHARE a GPU (tests, rehearsals): scoring waves that wait inside the kernel would hold the very wave slots
        // the lagging peer's kernels need.  One thread on the stream waits instead (a launch more per read set: this mode only).
        hipLaunchKernelGGL(phi_ipc_need_kernel, dim3(1), dim3(1), 0, c->stream, g->mbox, (unsigned long long)A.ipc_need, g->ticks());
        A.ipc_need = 0;
    }
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
    g->gather_blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((c->n_unique / 8 + 1) + 1023) / 1024, 64));   // (few: one lane of each spins beside the scoring)
    if (const char *e = getenv("PHI_IPC_BLOCKS")) g->gather_blocks = (unsigned)std::max(1, atoi(e));
    if (shm_barrier(g, rc)) { ipc_release(c, g); return rc ? rc : phi_fail(c, PHI_ERR_STATE, "another rank could not join the group"); }
    // ---- what the others offer
    for (int r = 0; r < n_ranks && !rc; r++) {
        if (r == rank) continue;
        const IpcSlot &o = g->shm->slot[r];
        if (o.n_unique != c->n_unique) { rc = phi_fail(c, PHI_ERR_INVALID, "the ranks of a group hold different graphs (%lld and %lld distinct walk minimisers)", (long long)c->n_unique, (long long)o.n_unique); break; }
        if (o.device == c->device) g->shared_device = true;
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
    const int64_t n_words = c->n_unique / 8 + 1;               // (the vectors are allocated in whole words)
    g->step++;
    g->step_of_gen[c->sp_gen % PHI_HIT_RING] = g->step; g->gen_tag[c->sp_gen % PHI_HIT_RING] = c->sp_gen;
    IpcPeerArgs P{};
    int np = 0;
    for (int r = 0; r < g->n; r++) {
        if (r == g->rank) continue;
        P.flag[np] = (const unsigned long long *)g->peer_mbox[r];
        for (int b = 0; b < PHI_HIT_RING; b++) P.hit[np][b] = (const unsigned long long *)g->peer_hit[r][b];
        np++;
    }
    hipLaunchKernelGGL(phi_ipc_gather_kernel, dim3(g->gather_blocks), dim3(256), 0, g->xstream, c->d_hit.as<unsigned long long>(), g->mbox,
                       (unsigned long long)((g->step << 2) | (uint64_t)c->hit_idx), P, np, n_words, (unsigned long long)g->step, g->ticks());
    HIPCHK(hipGetLastError());
    c->solved = false;
    return PHI_OK;
}

// "Everything issued so far may complete": the gather of an exchange starts once its read set is scored, which the first wave
// of the context's NEXT read launch tells it -- or this call (a one-thread kernel on the context's stream), or anything of the
// library that looks at the hit vector.  Needed only before waiting on the device by other means (hipDeviceSynchronize,
// torch.cuda.synchronize) behind a LAST exchange: without it that wait lasts until the gather's timeout.  Asynchronous.
int phi_ipc_flush(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    if (!c->ipc) return PHI_OK;
    HIPCHK(hipSetDevice(c->device));
    return phi_ipc_wait_pending(c);
}

// has a gather given up on a peer?  (waits for the group's stream)
int phi_ipc_check(phi_ctx *c)
{
    if (!c) return PHI_ERR_INVALID;
    PhiIpc *g = c->ipc;
    if (!g) return PHI_OK;
    HIPCHK(hipSetDevice(c->device));
    PHICHK(phi_ipc_wait_pending(c));                           // (says "scored" for the last exchange: no scoring launch may follow that would)
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(g->xstream));
    uint32_t err = 0;
    HIPCHK(phi_copy_sync(c, &err, g->mbox + MB_ERR, 4, hipMemcpyDeviceToHost));
    if (err) return phi_fail(c, PHI_ERR_DEVICE, "phi_ipc: a flag did not arrive within %.1f s (error bits %u: 1 = a peer's -- a rank died, or the ranks do not call the exchange equally often --, 2 / 4 = this rank's own gather)", g->timeout_s, err);
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
    (void)phi_ipc_wait_pending(c);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(g->xstream);
    (void)shm_barrier(g, 0, 10.0);
    c->ipc = nullptr;
    c->hit_n = 2;                                              // (the two extra hit buffers stay allocated, unused)
    ipc_release(c, g);
    return PHI_OK;
}

}  // extern "C"
