/*
 * phi_amd.h -- C ABI of the MI355X-native PHI hot path (libphi_amd.so).
 *
 * The reference (at-cg/PHI) has no FFI seam: its hot path is the C++ member function
 * ILP_index::ILP_function (src/ILP_index.cpp:528-1601) reading public fields that
 * ILP_index::read_gfa (src/ILP_index.cpp:20-155) fills.  This header is the seam a maintainer
 * binds instead (INTEGRATION.md shows the patch to src/main.cpp:114-140): plain pointers and
 * sizes, no C++ or torch types.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 (PHI_OK) or a negative phi_status; nothing calls exit();
 *   - host pointers are borrowed for the duration of the call and copied;
 *   - result buffers are owned by the context and stay valid until the next phi_solve /
 *     phi_ctx_destroy;
 *   - one context per process per GPU, calls serialised by the caller (the reference is not
 *     re-entrant either: src/main.cpp:136-140).
 *
 * Limits (PHI_ERR_UNSUPPORTED / PHI_ERR_INVALID beyond them): k <= 64 (k <= 32 on the fast 2-bit kernels; 33 .. 64 through
 * the exact byte-wise routine, and only while no k-mer covers more than 32 vertices), w <= 256, at most 1022 walks (one per lane of the largest workgroup: 513 and more run a slower instance of the dense DP kernel),
 * at most 2^32 - 64 walk entries (2.5 G solved: profiles/wide_entries.py), fewer than 2^31 minimisers of the distinct walk
 * contexts, anchors in the model and walk entries on vertices with recombination edges, at most
 * 254 out-edges and 255 recombination in-edges per vertex, no walk through a segment without
 * sequence, no graph whose walks both start and end at interior vertices.
 */
#ifndef PHI_AMD_H
#define PHI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phi_ctx phi_ctx;

typedef enum {
    PHI_OK = 0,
    PHI_ERR_INVALID = -1,      /* bad argument (null pointer, k/w out of range, offsets not monotone) */
    PHI_ERR_NOMEM = -2,        /* host or device allocation failed */
    PHI_ERR_DEVICE = -3,       /* HIP runtime error, kernel fault, no gfx950 device */
    PHI_ERR_STATE = -4,        /* call order violated (e.g. phi_solve before phi_set_graph) */
    PHI_ERR_UNSUPPORTED = -5,  /* input the path does not handle: see phi_last_error() */
    PHI_ERR_WALK = -6,         /* walk does not follow graph edges / reverse-strand vertex:
                                  the reference exit(1)s at ILP_index.cpp:104-107, :1568-1572 */
    PHI_ERR_OVERFLOW = -7      /* an internal table overflowed its capacity */
} phi_status;

const char *phi_strerror(int status);
/* Human-readable detail of the last failure on this context ("" if none). */
const char *phi_last_error(const phi_ctx *ctx);

/* ILP_index::ILP_index(gfa_t*) (ILP_index.cpp:4-6).  device_id = HIP device ordinal.  Also pays the one-time costs of the
 * HIP runtime on this device (code objects of the kernels, staging of the first host copy: ~10 ms) so that they do not
 * fall into the first phi_set_graph / phi_solve: create the context while the graph file is being parsed. */
int phi_ctx_create(int device_id, phi_ctx **out);
void phi_ctx_destroy(phi_ctx *ctx);

/* Run all work of this context on the caller's HIP stream (hipStream_t passed as void*);
 * NULL restores the context's private stream, which is non-blocking: it is NOT ordered against
 * the null stream.  A caller that shares buffers with this context (phi_hits_buffer,
 * phi_add_reads_device, phi_spectrum_export) passes the stream its own work runs on, or
 * synchronises the device. */
int phi_set_stream(phi_ctx *ctx, void *hip_stream);

/* Flags of phi_set_params. */
#define PHI_FLAG_QCLP 1u       /* -q1 (main.cpp:66): accepted; both programs have the same optimum */
#define PHI_FLAG_MIXED 2u      /* -m1 (main.cpp:62): accepted; integral optimum of equal value exists */

/* main.cpp:118-131: k_mer, window, threshold, recombination, is_qclp/is_mixed.
 * k in [1,64] (33 .. 64: exact but slow, see the limits above), w in [1,256].  Must precede phi_set_graph. */
int phi_set_params(phi_ctx *ctx, int32_t k, int32_t w, float threshold, int32_t recombination,
                   uint32_t flags);

/*
 * ILP_index::read_gfa's outputs (ILP_index.h:53-61) as flat arrays, then stage 1a of
 * ILP_function (:559-573): the graph goes to HBM, every walk is sketched on the GPU and the
 * minimiser table is built.
 *   seq_concat/seq_off[n_vtx+1]  node_seq, original case
 *   adj_off[n_vtx+1]/adj         adj_list (forward strand)
 *   walk_off[n_walks+1]/walk_vtx paths
 *   topo_rank[n_vtx]             top_order_map
 */
int phi_set_graph(phi_ctx *ctx, int32_t n_vtx, const char *seq_concat, const int64_t *seq_off,
                  const int64_t *adj_off, const int32_t *adj, int32_t n_walks,
                  const int64_t *walk_off, const int32_t *walk_vtx, const int32_t *topo_rank);

/*
 * Stage 1b/2a of ILP_function (:617-655) for one batch of reads: sketch, spectrum insert,
 * match against the walk minimiser table.  Streaming: may be called repeatedly.
 * bases = raw ASCII (upper/lower case; any byte), read r = bases[read_off[r], read_off[r+1]).
 */
int phi_add_reads(phi_ctx *ctx, const char *bases, const int64_t *read_off, int64_t n_reads);
/* Same, with both arrays already resident in this GPU's HBM (n_bases = read_off[n_reads]).  d_read_off may be NULL for reads of
 * ONE length (n_bases / n_reads each, as sequencers write short reads): the kernel then computes the read starts and reads no offsets. */
int phi_add_reads_device(phi_ctx *ctx, const void *d_bases, const void *d_read_off, int64_t n_reads,
                         int64_t n_bases);
/*
 * The same stage fed with the reads file's TEXT (FASTA or FASTQ, already inflated), in pieces of any size in file order:
 * the records are found on the device (kseq's rules, src/kseq.h:192-233, ILP_index.cpp:313-328, for the two regular
 * layouts: FASTQ with four lines per record; FASTA, wrapped or not) instead of by a byte-at-a-time state machine on one
 * host core, and piece i + 1 crosses the link while piece i is sketched.
 *   phi_reads_text_begin   start a stream; max_chunk_bytes sizes the device buffers (a longer piece is cut)
 *   phi_add_reads_text     the next bytes of the stream.  The device takes the whole records it finds and keeps the
 *                          unfinished rest for the next call.  *irregular = 1: the text is not laid out in one of the
 *                          two regular ways (carriage returns, a wrapped FASTQ record, text before the first header,
 *                          a record longer than the buffers ...): nothing more is taken, further calls fail with
 *                          PHI_ERR_STATE, and the caller finishes the stream on the exact host reader (phi_host.h)
 *   phi_reads_text_end     ends the stream and hands back the bytes handed over but NOT taken (valid until the next
 *                          phi_reads_text_begin) -- after an irregular piece: everything from the first byte not taken
 *                          to the end of that call's bytes; at the end of a regular file: the last record, whose end
 *                          only the end of the file shows -- to be parsed by the host reader and added with
 *                          phi_add_reads; *n_taken = bytes of the stream taken as whole records.
 * Together the device-side records and the host-parsed rest are exactly kseq's records of the file.
 */
int phi_reads_text_begin(phi_ctx *ctx, int64_t max_chunk_bytes);
int phi_add_reads_text(phi_ctx *ctx, const char *text, int64_t n_bytes, int32_t *irregular);
int phi_reads_text_end(phi_ctx *ctx, const char **pending, int64_t *n_pending, int64_t *n_taken);
/* Introspection for the parity tests: the records the device took from the LAST piece handed to phi_add_reads_text (their
 * bases back to back and their offsets, off[0] = 0), copied to the host.  Sizes only when the buffers are too small. */
/*
 * Reads text that arrives BEFORE the graph is there (the command line: the reads file is read while the GFA is parsed and the
 * index built -- seconds at chromosome scale, with the link and 288 GB of HBM idle).  A park holds pieces of the stream in
 * device memory; it belongs to no context (its own stream and buffers), so a reader thread may fill it while phi_set_graph
 * runs on another thread.  phi_add_reads_text_parked is phi_add_reads_text with the index-th piece as its bytes (the pieces in
 * stream order, mixed freely with phi_add_reads_text calls); irregular text is handed back through phi_reads_text_end as ever,
 * and the host reads the pieces still parked with phi_text_park_fetch.
 *   phi_text_park_create / _destroy   on a device
 *   phi_text_park_pin                 page-locks a host buffer the pieces come from (unpinned by _destroy)
 *   phi_text_park_add                 copies n bytes to the device; returns when they are there (the host buffer is free again)
 *   phi_text_park_add_async / _wait   the same in two halves: the copy is issued / the host buffer may be written again (the next
 *                                     chunk is read into another buffer meanwhile)
 *   phi_text_park_bytes / _fetch      a piece's size; its bytes back on the host
 *   phi_text_park_release             the piece's device memory is let go (after phi_add_reads_text_parked took it)
 */
typedef struct phi_text_park phi_text_park;
int phi_text_park_create(int32_t device, phi_text_park **out);
int phi_text_park_pin(phi_text_park *park, void *host, size_t bytes);
int phi_text_park_add(phi_text_park *park, const char *text, int64_t n, int32_t *index);
int phi_text_park_add_async(phi_text_park *park, const char *text, int64_t n, int32_t *index);
int phi_text_park_wait(phi_text_park *park, int32_t index);
int64_t phi_text_park_bytes(phi_text_park *park, int32_t index);
int phi_text_park_fetch(phi_text_park *park, int32_t index, char *out, int64_t cap);
int phi_text_park_release(phi_text_park *park, int32_t index);
void phi_text_park_destroy(phi_text_park *park);
int phi_add_reads_text_parked(phi_ctx *ctx, phi_text_park *park, int32_t index, int32_t *irregular);
int phi_reads_text_last_batch(phi_ctx *ctx, char *bases, int64_t cap_bases, int64_t *off, int64_t cap_reads, int64_t *n_reads,
                              int64_t *n_bases);
/* For a stream whose chunks go to SEVERAL contexts in turn (one per GPU): hands out the bytes this context holds
 * unfinished (valid until its next phi_add_reads_text) and forgets them, so that the caller can put them in front of the
 * next chunk on whichever context takes it.  Does not wait for the sketch of the records already taken. */
int phi_reads_text_detach_carry(phi_ctx *ctx, const char **bytes, int64_t *n);
/* Forget all reads seen so far (graph index is kept).  The clearing itself may be folded into the
 * next batch's first launch; every call on this context that observes the spectrum, the counters
 * or the hit vector sees the reads forgotten.  A hit-vector pointer obtained earlier from
 * phi_hits_buffer must not be read between this call and the next phi_add_reads*: fetch it again. */
int phi_reset_reads(phi_ctx *ctx);
/* Totals since the last reset (waits for the stream): reads, bases, emitted read minimisers
 * (with multiplicity) and distinct read hashes so far. */
int phi_reads_stats(phi_ctx *ctx, int64_t *n_reads, int64_t *n_bases, int64_t *n_emitted, int64_t *n_distinct);

/*
 * Multi-GPU exchange (no reference counterpart: the reference is one process).  Each rank holds
 * a shard of the reads; before phi_solve the caller all-reduces (MAX) the hit vector in place
 * and tells every rank the size of the union spectrum.
 *   phi_hits_buffer     device pointer to uint8 hit[n], n = number of distinct walk minimisers;
 *                       index = dense minimiser id (rank of the hash's first occurrence in walk
 *                       position order), identical on every rank for the same graph; valid until
 *                       the next phi_reset_reads / phi_set_graph on this context
 *   phi_spectrum_export this rank's distinct read hashes that are NOT walk minimisers (those that
 *                       are, are exactly the set hit flags): device pointer to uint64[n] (valid
 *                       until the next call on this context)
 *   phi_spectrum_import merge another rank's exported hashes (a device buffer the caller owns): a
 *                       hash that is a walk minimiser sets its hit flag, any other joins the local
 *                       set; after the hit all-reduce and the imports |Sp_R| (ILP_index.cpp:641)
 *                       = flags set + set size is the same on every rank
 *   phi_spectrum_set_size  alternatively, override |Sp_R| used in the log counters
 */
int phi_hits_buffer(phi_ctx *ctx, void **d_hits, int64_t *n);
int phi_spectrum_export(phi_ctx *ctx, void **d_hashes, int64_t *n);
int phi_spectrum_import(phi_ctx *ctx, const void *d_hashes, int64_t n);
int phi_spectrum_set_size(phi_ctx *ctx, int64_t global_size);

/*
 * The same exchange done by the library itself with RCCL over xGMI (librccl is loaded on first use).
 * One context per GPU -- one process per GPU, or one host thread per GPU in one process (the `PHI
 * --devices 0,1,..` mode) -- all holding the same graph and parameters, each fed its own shard of reads.
 *   phi_comm_unique_id      ncclGetUniqueId: 128 bytes made by one rank and handed to the others out of
 *                           band (a file, MPI, a torch.distributed store, shared memory between threads)
 *   phi_comm_init           ncclCommInitRank on the context's device: collective over the n_ranks contexts
 *   phi_comm_allreduce_hits step 1 alone: ncclAllReduce(MAX, uint8) of the hit vector in place, on the
 *                           context's stream, asynchronous (what a job times per read set)
 *   phi_comm_exchange       the whole exchange, once per job after the rank's last read batch: step 1, then
 *                           ncclAllGather of the sizes and of the padded lists of phi_spectrum_export, and
 *                           the import of every other rank's list.  Afterwards the hit vector, |Sp_R| and so
 *                           phi_solve's result are identical on every rank.
 *   phi_comm_destroy        also done by phi_ctx_destroy
 */
#define PHI_COMM_ID_BYTES 128
int phi_comm_unique_id(void *id_out, size_t cap);
int phi_comm_init(phi_ctx *ctx, const void *id, int32_t rank, int32_t n_ranks);
int phi_comm_info(const phi_ctx *ctx, int32_t *rank, int32_t *n_ranks);
int phi_comm_allreduce_hits(phi_ctx *ctx);
int phi_comm_exchange(phi_ctx *ctx);
int phi_comm_destroy(phi_ctx *ctx);

/*
 * The same exchange for the contexts of ONE process -- one host thread and one context per GPU, the `PHI --devices` mode --
 * without RCCL: every GPU ORs the peers' hit vectors into its own with one kernel that loads them across xGMI
 * (hipDeviceEnablePeerAccess), ordered by HIP events; the lists of the other read hashes are imported where they lie.
 * Meant for hit vectors of a few MB, where an 8-rank ncclAllReduce is latency (tens of microseconds: as long as one GPU
 * takes to score a whole MHC read set).  All calls but create / destroy are collective over the group's threads.
 *   phi_peers_create          a group for n_ranks contexts (at most 16)
 *   phi_peers_join            every rank's thread, after phi_set_graph: peer access, events
 *   phi_peers_allreduce_hits  step 1 alone, asynchronous on the contexts' streams
 *   phi_peers_exchange        steps 1 + 2: afterwards phi_solve gives the same result on every rank
 *   phi_peers_destroy         once, when no rank will call into the group again
 */
int phi_peers_create(int32_t n_ranks, void **group);
int phi_peers_join(phi_ctx *ctx, void *group, int32_t rank);
int phi_peers_allreduce_hits(phi_ctx *ctx);
int phi_peers_exchange(phi_ctx *ctx);
int phi_peers_destroy(void *group);

/*
 * The same exchange between PROCESSES of one node -- one process and one context per GPU, as bench.py runs under
 * torch.distributed.run -- without RCCL: every rank maps the other ranks' hit vectors (hipIpcGetMemHandle /
 * hipIpcOpenMemHandle; the loads travel over xGMI) and ORs them into its own with one kernel per read set.  The ranks
 * order themselves through flags in device memory; no host call takes part in an exchange, and the gather of read set i
 * runs on a stream of its own beside the scoring of read set i + 1 (the context keeps four hit vectors for that).  For hit
 * vectors of a few MB, where an 8-rank ncclAllReduce is all latency.  One exchange per read set (per phi_reset_reads), the
 * same sequence of calls on every rank.  HSA_ENABLE_IPC_MODE_LEGACY=0 where the host driver only shares memory by dmabuf.
 *   phi_ipc_unique_id        128 bytes (the name of a small shared-memory block) made by one rank, handed to the others out of band
 *   phi_ipc_init             collective, after phi_set_graph: handles published and mapped; phi_set_graph is refused from here on
 *   phi_ipc_allreduce_hits   step 1 alone, asynchronous: the gather starts once the read set is scored, which the context's NEXT
 *                            read launch tells it (no launch, event or host call in between); whatever observes the hit vector
 *                            through this library afterwards waits for it by itself
 *   phi_ipc_flush            behind a LAST exchange, before waiting on the device by other means (hipDeviceSynchronize,
 *                            torch.cuda.synchronize): lets the gather start without a further read launch.  Asynchronous
 *   phi_ipc_exchange         steps 1 + 2 (the lists of novel read hashes, through mapped buffers and a host barrier), once per job
 *   phi_ipc_check            waits for the gathers issued so far; PHI_ERR_DEVICE when one gave up on a peer (PHI_IPC_TIMEOUT_S, 20 s)
 *   phi_ipc_destroy          collective; also done by phi_ctx_destroy
 */
int phi_ipc_unique_id(void *id_out, size_t cap);
int phi_ipc_init(phi_ctx *ctx, const void *id, int32_t rank, int32_t n_ranks);
int phi_ipc_info(const phi_ctx *ctx, int32_t *rank, int32_t *n_ranks);
int phi_ipc_allreduce_hits(phi_ctx *ctx);
int phi_ipc_flush(phi_ctx *ctx);
int phi_ipc_exchange(phi_ctx *ctx);
int phi_ipc_check(phi_ctx *ctx);
int phi_ipc_destroy(phi_ctx *ctx);

typedef struct {
    /* ---- solve (ILP_index.cpp:776-1418) */
    int64_t objective;          /* max  #covered minimisers - 2*(R/2)*#recombinations          */
    int64_t upper_bound;        /* proven bound; optimal iff upper_bound == objective           */
    int32_t optimal;            /* 1 when proven optimal; 0 when the search ran out of its budget of DP runs
                                   (phi_set_solve_budget): the path is feasible, the bound proven */
    int32_t n_dp_runs;          /* DP launches used (1 = certificate closed at the root)        */
    int64_t n_covered;          /* minimisers with >=1 anchor fully traversed (sum of z_i)      */
    /* ---- decode (:1431-1525) */
    int64_t n_path;
    const int32_t *path_vtx;    /* [n_path] vertices in topological order                       */
    const int32_t *path_hap;    /* [n_path] haplotype label of each vertex                      */
    int32_t recombination_count;/* :1519 adjacent label changes                                 */
    int32_t n_switches;         /* w-node traversals (each costs 2*(R/2))                       */
    int64_t hap_len;            /* length of the inferred sequence                              */
    /* ---- log counters (:563, :641, :734, :738-743, :883) */
    int32_t n_walks;
    const int64_t *n_minimizers;/* [n_walks] "Number of Minimizers"                             */
    const int64_t *n_anchors;   /* [n_walks] "Number of Anchors" (after the filter)             */
    int64_t spectrum_size;      /* |Sp_R|                                                       */
    int64_t filtered;           /* minimisers dropped by the shared-anchor filter               */
    int64_t retained;           /* spectrum_size - filtered                                     */
    int64_t n_in_model;         /* minimisers with a z_i ("% Minimizers are in ILP")            */
} phi_result;

/* Budget of the exact search behind phi_solve, counted in DP runs (never in wall-clock time: the same
 * input gives the same result and the same `optimal` flag on every run).  The reference's
 * model.optimize() (ILP_index.cpp:1412-1418) sets no limit; max_dp_runs <= 0 means the same here.
 * Default 4096 (minutes at most on a 49-walk MHC graph, seconds on small ones).  Almost every input closes at the
 * root in 1-3 runs; what does not are graphs where short k-mers repeat all over (k <= 8). */
int phi_set_solve_budget(phi_ctx *ctx, int64_t max_dp_runs);

/* Stages 2b-3 of ILP_function (:670-1525): filter, exact solve, decode.  Replaces
 * model.optimize() (:1418) with a max-plus DP + optimality certificate. */
int phi_solve(phi_ctx *ctx, phi_result *out);

/* ILP_index.cpp:1577-1581: concatenated original-case node sequences of the path.
 * buf must hold result.hap_len bytes. */
int phi_path_sequence(phi_ctx *ctx, char *buf, int64_t cap);

/*
 * Introspection used by the parity tests (tests/ compare these with oracle/).
 * phi_sketch: stand-alone (w,k)-minimiser sketch of arbitrary sequences on the GPU
 * (compute_hashes / index_kmers minus the vertex map).  Records come back sorted by
 * (sequence, position).  Pass cap = 0 to query *n_out.
 */
int phi_sketch(phi_ctx *ctx, const char *bases, const int64_t *seq_off, int64_t n_seq, int32_t k,
               int32_t w, uint64_t *out_hash, int64_t *out_pos, int32_t *out_seq, int64_t cap,
               int64_t *n_out);
/*
 * What phi_set_graph built.  The reference sketches every walk on its own (ILP_index.cpp:559-573); here walk
 * entries with the same context (the base before, the vertex, the next w+k-2 bases of the walk) form a class
 * that is sketched once (graph-side de-duplication): n_classes classes laid out in class_bases bases of "class
 * space" stand for walk_bases bases of walks, n_class_records minimiser records for n_walk_minimizers.
 * sketch_gpu_ms = GPU time from the first class kernel to the last class record (HIP events on the stream).
 */
typedef struct {
    int64_t n_entries, walk_bases;
    int64_t n_classes, class_bases, n_class_records;
    int64_t n_walk_minimizers, n_distinct_minimizers;
    double sketch_gpu_ms;
} phi_index_info;
int phi_index_stats(phi_ctx *ctx, phi_index_info *out);

/*
 * How the last phi_solve ran its DP (no reference counterpart: the reference hands the model to Gurobi).
 * dp_mode: 0 = every vertex is a step (more than 256 walks, or the fallback of the event kernels), 1 = event
 * driven, one chain over the compact steps, 2 = blocks of steps in parallel on walk lanes (<= 64 walks),
 * 3 = blocks in parallel with their transfer rows on class lanes (65 .. 256 walks).  n_blocks = 0 unless 2 / 3;
 * max_classes / mean_classes: class lanes per block of the last DP run (mode 3).
 */
typedef struct {
    int64_t n_dp_anchors, n_events;
    int32_t n_steps, n_blocks, dp_mode, max_classes;
    double mean_classes;
} phi_solve_info;
int phi_solve_stats(phi_ctx *ctx, phi_solve_info *out);

/* Minimisers of walk h found by phi_set_graph, sorted by position. */
int phi_walk_minimizers(phi_ctx *ctx, int32_t walk, uint64_t *out_hash, int64_t *out_pos,
                        int64_t cap, int64_t *n_out);
/*
 * The reference's -d1 report (ILP_index.cpp:565-604): hist[c], c = 1..n_walks, = number of distinct
 * walk minimisers that occur in exactly c walks (hist[0] = 0); *n_distinct = their total.  hist has
 * cap >= n_walks + 1 entries.  Valid after phi_set_graph.
 */
int phi_walk_sharing(phi_ctx *ctx, int64_t *hist, int32_t cap, int64_t *n_distinct);

/* Kept anchors after the filter (valid after phi_solve): hash, walk, first/last walk index. */
int phi_kept_anchors(phi_ctx *ctx, uint64_t *out_hash, int32_t *out_walk, int32_t *out_t0,
                     int32_t *out_t1, int64_t cap, int64_t *n_out);

/* Pin / unpin caller memory on this context's device (hipHostRegister): for host buffers handed to
 * phi_add_reads again and again, e.g. the chunk buffers of a streaming reads reader (phi_host.h), so
 * that the device copy is a direct DMA.  No reference counterpart. */
int phi_host_register(phi_ctx *ctx, void *p, size_t bytes);
int phi_host_unregister(phi_ctx *ctx, void *p);

/* Timing of the dominant kernel (the sketch kernel), measured with HIP events on the stream
 * the kernel is launched on.  phi_prof_enable(n), n >= 1, starts bracketing every n-th sketch launch
 * (a bracketed launch costs the stream a few microseconds more than a plain one, so a throughput
 * measurement samples: n = 8); phi_prof_enable(0) stops.  phi_prof_read returns the number of
 * bracketed launches, their summed duration and the bases they covered. */
int phi_prof_enable(phi_ctx *ctx, int on);
int phi_prof_read(phi_ctx *ctx, int64_t *n_launches, double *total_ms, int64_t *total_bases);

/*
 * The walks resolved ON THE DEVICE from the text of the GFA's W-lines (the reference: gfa-io.cpp:367-432 on one core, then
 * ILP_index.cpp:96-113).  At chromosome scale the walk text is the file (10.5 of config 5's 10.9 GB).
 *   phi_walk_text_upload    the walk field of every W-line (">s17>s18...", optional tags may follow), as it stands in the (mapped)
 *                           file, to the device through pinned staging; returns when the last piece is on its way -- call it
 *                           on a thread of its own while the host still reads the S- and L-lines
 *   phi_walk_text_resolve   names <prefix><canonical decimal> -> num2id[number] (the direct index of the host reader's name table,
 *                           include/phi_host.h phi_graph_name_index); walk_off_out[n_walks + 1].  *irregular != 0 (a reverse step:
 *                           the reference flips such walks by majority strand; a name of another form, or naming no segment: the
 *                           reference leaves such steps out): nothing was resolved, fall back to the host reader
 *   phi_set_graph(..., walk_vtx = NULL, ...)  then takes the walk entries from where phi_walk_text_resolve left them
 *   phi_walk_entries        (tests) a host copy of the walk entries on the device
 */
typedef struct { const char *text; int64_t n; } phi_walk_text;
int phi_walk_text_upload(phi_ctx *ctx, const phi_walk_text *walks, int32_t n_walks);
int phi_walk_text_resolve(phi_ctx *ctx, const char *prefix, int32_t prefix_n, const int32_t *num2id, int64_t n_num, int32_t n_seg,
                          int64_t *walk_off_out, uint32_t *irregular);
int phi_walk_entries(phi_ctx *ctx, int32_t *out, int64_t cap, int64_t *n);

/* Wait for everything this process has put on the context's device, on every stream, and report the device's error
 * state: a fault raised by an earlier asynchronous launch surfaces here (diagnostics; no reference counterpart). */
int phi_device_synchronize(phi_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
