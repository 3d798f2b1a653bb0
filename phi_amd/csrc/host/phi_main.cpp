// phi_main.cpp -- the `PHI` command-line driver: same flags, stderr log lines and FASTA output as
// the reference's src/main.cpp + the logging of ILP_index::ILP_function, with the hot path behind
// the C ABI of include/phi_amd.h (HIP kernels on one MI355X).  Own implementation.
//
//   ./PHI -g <target.gfa> -r <reads.fa> -o <haplotype.fasta> [-k -w -R -q -m -T -t -d -N -c]
//         [--device N | --devices 0,1,..] [--dp-budget RUNS]
//   ./PHI -g <target.gfa> -r a.fq -o a.fa -r b.fq -o b.fa ...      several read sets against ONE graph: the graph is parsed and
//         indexed once (the reference's harness runs PHI once per sample x coverage on the same graph,
//         data/run_batch_4_miqp.py:31-46), every job prints the log of a run of its own and writes its own FASTA
//
// --devices: one context and one host thread per GPU; every GPU builds the full index, the chunks of the reads file are
// handed out in turn, the library's RCCL exchange (phi_comm_*) merges hit vectors and spectra once, and the first
// GPU solves and reports.  A read set smaller than --shard-min-bases per GPU uses fewer GPUs.
//
// Flag semantics (main.cpp:38-95): -q (IQP/ILP), -m (mixed/integer), -N (naive expanded graph)
// choose between formulations with the same optimum; they are accepted and mapped onto the one
// exact solver.  -t only sized OpenMP/Gurobi thread pools and is accepted and ignored.  The log
// lines scraped by the reference's evaluation scripts (data/postprocessing_2_MIQP.py:55-79) keep
// their exact formats.
//
// From process start to the closed FASTA what a small input pays is the HIP runtime, not the
// path: ~60-180 ms to start it, ~20 ms per hardware queue, and ~100 ms for the driver to take the process's GPU state
// down again when it ends.  So: the device context is made by a thread of its own while the graph is parsed; the reads
// file goes to the device as raw text (the records are found there: phi_add_reads_text); and the process that does
// the work can be a CHILD (PHI_DETACH=1) -- the parent returns the child's status the moment the FASTA is closed and the log
// written, the child's teardown (free of the arrays, the driver's cleanup: ~0.1 s at MHC size, ~1 s and 65 GB of HBM at
// chromosome scale) goes on behind it.  Off by default: when the reference's command returns its resources are free, and a
// harness that starts the next GPU job at once would find them still held.
// Exit status: 0; 1 on any error; 3 when --dp-budget was given and ran out before the path was proven optimal.
#include <errno.h>
#include <getopt.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <string>
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../../include/phi_amd.h"
#include "../../../include/phi_host.h"

#define PHI_VERSION "1.0-mi355x"

static double t0_real, cpu0;
static double realtime()
{
    struct timeval tp;
    gettimeofday(&tp, nullptr);
    return tp.tv_sec + tp.tv_usec * 1e-6;
}
static double cputime()
{
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec) - cpu0;
}
static long peakrss()
{
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_maxrss * 1024;
}
// "[M::func::<wall>*<cpu/wall>] " stamp of the reference (sys.cpp:92-117 users)
static void stamp(const char *func)
{
    const double w = realtime() - t0_real;
    fprintf(stderr, "[M::%s::%.3f*%.2f] ", func, w, cputime() / (w > 0 ? w : 1e-9));
}

// PHI_TIMING=1: the stages of the run, each with its begin and end on the process's clock (seconds since main was entered)
struct StageMarks {
    bool on = getenv("PHI_TIMING") != nullptr;
    std::mutex mu;
    struct M { std::string name; double b, e; };
    std::vector<M> marks;
    void add(const char *name, double b, double e)
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu);
        marks.push_back(M{name, b - t0_real, e - t0_real});
    }
    void print()
    {
        if (!on) return;
        std::sort(marks.begin(), marks.end(), [](const M &a, const M &b) { return a.b < b.b; });
        fprintf(stderr, "[phi timing] main: entered at epoch %.6f\n", t0_real);
        for (const M &m : marks) fprintf(stderr, "[phi timing] main: stage %-34s %8.3f -> %8.3f s  (%9.3f ms)\n", m.name.c_str(), m.b, m.e, (m.e - m.b) * 1e3);
    }
};
static StageMarks g_marks;
struct Stage {
    const char *name; double b;
    explicit Stage(const char *n) : name(n), b(realtime()) {}
    ~Stage() { g_marks.add(name, b, realtime()); }
};

static void usage(FILE *fp, int k, int w, int R, int q, int m, float T, int t, const char *g, const char *r, const char *o, int d)
{
    fprintf(fp, "Usage: PHI -g <target.gfa> -r <reads.fa> -o <haplotype.fasta> \n");
    fprintf(fp, "Options:\n");
    fprintf(fp, "    -k INT       K-mer size [%d]\n", k);
    fprintf(fp, "    -w INT       Minimizer window size [%d]\n", w);
    fprintf(fp, "    -R INT       Recombination penalty [%d]\n", R);
    fprintf(fp, "    -q INT       Mode QP/ILP (default IQP i.e q1, use q0 for ILP) [%d]\n", q);
    fprintf(fp, "    -m INT       Mixed/Interger programming (default Mixed i.e -m1, use -m0 for Integer) [%d]\n", m);
    fprintf(fp, "    -T FLOAT     Threshold for minimizer filtering [%.3f]\n", T);
    fprintf(fp, "    -t INT       Threads [%d]\n", t);
    fprintf(fp, "    -g INT       GFA file [%s]\n", g);
    fprintf(fp, "    -r INT       Read [%s]\n", r);
    fprintf(fp, "    -o INT       Output haplotype [%s]\n", o);
    fprintf(fp, "    -d bool      Debug mode [%d]\n", d);
}

struct Options {
    int k = 31, w = 25, n_threads = 4, recombination = 100, is_qclp = 1, is_naive = 0, is_mixed = 1, debug = 0;
    int device = 0, max_occ = 5000;
    std::vector<int> devices;                                 // --devices: one context (and host thread) per GPU
    long long dp_budget = -1;                                 // --dp-budget: DP runs of the exact search; not given: no limit, as model.optimize()
    long long shard_min_bases = 50000000;                     // --shard-min-bases: text bytes of reads a further GPU must be worth
    float threshold = 1.0f;
    std::string gfa_file, reads_file, hap_file;               // (reads_file / hap_file: the first job's, for the usage text)
    std::vector<std::string> reads_files, hap_files;          // one entry per job: -r a -o a.fa -r b -o b.fa ...
    int argc = 0;
    char **argv = nullptr;
    bool detached = false;
};

// ---- the queue of raw text chunks between the reader thread and the device thread(s)
struct Chunk { char *text = nullptr; int64_t n = 0; int32_t parked = -1; };      // parked >= 0: the bytes wait in device memory (phi_text_park_*), no host buffer
struct ChunkQueue {
    std::deque<Chunk> buf;                                    // (a deque: entries for parked pieces are added while others are in use)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> q_free, q_full;                           // buffer indices; a full entry with n == 0 ends the stream
    bool stop = false;
    int take_full()                                           // blocks; the end marker stays in the queue for the other takers
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q_full.empty(); });
        const int s = q_full.front();
        if (buf[(size_t)s].n > 0) q_full.pop_front();
        return s;
    }
    Chunk *at(int s) { std::lock_guard<std::mutex> lk(mu); return &buf[(size_t)s]; }      // (entries never move; the deque's index may, while one is added)
    void give_free(int s)
    {
        { std::lock_guard<std::mutex> lk(mu); if (buf[(size_t)s].text) q_free.push_back(s); }
        cv.notify_all();
    }
};

// the rest of the stream for the host reader: blocks straight from the queue (phi_reads_stream_open_blocks)
struct QueueBlocks { ChunkQueue *q; int held = -1; phi_text_park *const *park = nullptr; std::vector<char> fetched; };      // park: where the park's handle will stand once the reader thread has made it
static int64_t next_block_from_queue(void *user, const char **block)
{
    QueueBlocks *qb = (QueueBlocks *)user;
    if (qb->held >= 0) { qb->q->give_free(qb->held); qb->held = -1; }
    const int s = qb->q->take_full();
    const Chunk &c = *qb->q->at(s);
    if (c.n <= 0) return c.n;                                 // 0: the end; negative: the reader thread failed
    if (c.parked >= 0) {
        // a piece that went to device memory before the graph was there: its bytes come back for the host reader
        qb->fetched.resize((size_t)c.n);
        if (!qb->park || phi_text_park_fetch(*qb->park, c.parked, qb->fetched.data(), c.n) != PHI_OK) return -1;
        (void)phi_text_park_release(*qb->park, c.parked);
        *block = qb->fetched.data();
        return c.n;
    }
    qb->held = s;
    *block = c.text;
    return c.n;
}

static int run(const Options &o)
{
    const int k = o.k, w = o.w, recombination = o.recombination, is_qclp = o.is_qclp, is_mixed = o.is_mixed, debug = o.debug;
    const std::string &gfa_file = o.gfa_file;
    const int n_jobs = (int)o.reads_files.size();
    std::string reads_file = o.reads_files[0], hap_file = o.hap_files[0];      // the job at hand
    char err[512] = "";

    // The device context (HIP initialisation) and the reads file are prepared by two host threads
    // while this one parses the graph: the three are independent.
    std::vector<int> devices = o.devices;
    if (devices.empty()) devices.push_back(o.device);
    for (size_t i = 0; i < devices.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (devices[i] == devices[j] && !getenv("PHI_ALLOW_SAME_DEVICE")) { fprintf(stderr, "[E::main] --devices names GPU %d twice\n", devices[i]); return 1; }   // (the tests run two contexts on one GPU)
    // a read set is sharded only over as many GPUs as it can keep busy: every further GPU costs an exchange (~tens of
    // microseconds) and an index build, and one GPU scores 50 Mbases in a fifth of a millisecond
    if (devices.size() > 1) {
        struct stat st;
        long long bytes = (stat(reads_file.c_str(), &st) == 0 && S_ISREG(st.st_mode)) ? (long long)st.st_size : -1;
        if (bytes >= 0) {
            const size_t want = (size_t)std::max<long long>(1, (bytes + o.shard_min_bases - 1) / std::max<long long>(1, o.shard_min_bases));
            if (want < devices.size()) {
                fprintf(stderr, "[M::main] reads file of %lld bytes: using %zu of the %zu GPUs given (--shard-min-bases %lld per GPU)\n", bytes, want, devices.size(), o.shard_min_bases);
                devices.resize(want);
            }
        }
    }
    const int n_dev = (int)devices.size();
    std::vector<phi_ctx *> ctxs((size_t)n_dev, nullptr);
    const bool timing = g_marks.on;
    std::shared_future<int> f_ctx = std::async(std::launch::async, [&]() {
        Stage st("device context(s) [thread]");
        // one host thread per GPU (each context initialises its own device)
        std::vector<std::future<int>> fs;
        for (int i = 0; i < n_dev; i++)
            fs.push_back(std::async(std::launch::async, [&, i]() { return phi_ctx_create(devices[(size_t)i], &ctxs[(size_t)i]); }));
        int r = 0;
        for (auto &f : fs) { const int ri = f.get(); if (ri && !r) r = ri; }
        return r;
    }).share();

    // The reads file is streamed as TEXT: a host thread fills chunk buffers with the file's (inflated)
    // bytes while this thread parses the graph and builds the index; every chunk then goes to phi_add_reads_text, which
    // finds the records on the device -- chunk i + 1 crosses the link while chunk i is sketched, and no byte of a
    // regular file is looked at by a host core.  Host memory stays bounded: 2 + GPUs buffers.
    int64_t chunk_bytes = getenv("PHI_READ_CHUNK") ? std::max<int64_t>(256, atoll(getenv("PHI_READ_CHUNK"))) : ((int64_t)64 << 20);
    if (!getenv("PHI_READ_CHUNK")) {
        // small plain files: one chunk of the file's size (of the largest file, when there are several jobs)
        int64_t need = 0;
        for (const std::string &rf : o.reads_files) {
            struct stat st;
            int64_t want = chunk_bytes;
            if (stat(rf.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
                FILE *fp = fopen(rf.c_str(), "rb");
                unsigned char m2[2] = {0, 0};
                const bool gz = fp && fread(m2, 1, 2, fp) == 2 && m2[0] == 0x1f && m2[1] == 0x8b;
                if (fp) fclose(fp);
                if (!gz) want = std::min<int64_t>(chunk_bytes, ((int64_t)st.st_size + 4095) & ~(int64_t)4095);
            }
            need = std::max(need, want);
        }
        chunk_bytes = need;
    }
    ChunkQueue Q;
    const int N_CHUNK_BUF = 2 + n_dev;                        // one in flight per GPU, two with the reader
    Q.buf.resize((size_t)N_CHUNK_BUF);
    for (int i = 0; i < N_CHUNK_BUF; i++) {
        Q.buf[(size_t)i].text = (char *)malloc((size_t)chunk_bytes);
        if (!Q.buf[(size_t)i].text) { fprintf(stderr, "[E::%s] out of memory\n", __func__); return 1; }
        Q.q_free.push_back(i);
    }
    char rerr[512] = "";
    std::future<int> f_reads;
    // reads text parked in device memory until the index is built (the reader thread's side of it is in start_reads below)
    phi_text_park *park = nullptr;
    std::atomic<bool> graph_ready{false}, park_go{false};      // the GFA is parsed (parking may begin) / the index is built (it ends)
    bool park_on = false;                                     // (the reader thread's)
    std::atomic<bool> park_pinned{false};                     // the reader thread pinned the chunk buffers (main then does not)
    int64_t parked_bytes = 0;
    const int64_t park_limit = getenv("PHI_TEXT_PARK_MAX") ? atoll(getenv("PHI_TEXT_PARK_MAX")) : ((int64_t)96 << 30);
    if (devices.size() == 1 && !(getenv("PHI_TEXT_PARK") && atoi(getenv("PHI_TEXT_PARK")) == 0)) {
        struct stat st;
        const int64_t least = getenv("PHI_TEXT_PARK_MIN") ? atoll(getenv("PHI_TEXT_PARK_MIN")) : ((int64_t)256 << 20);
        park_on = stat(reads_file.c_str(), &st) == 0 && S_ISREG(st.st_mode) && (int64_t)st.st_size >= least;
    }
    auto start_reads = [&]() {
      f_reads = std::async(std::launch::async, [&, rf = reads_file]() {
        Stage st("reads file -> text chunks [thread]");
        phi_text_stream *ts = nullptr;
        int r = phi_text_stream_open(rf.c_str(), &ts, rerr, sizeof rerr);
        int slot = -1, fly_slot = -1;
        int32_t fly_idx = -1;
        for (;;) {
            if (slot < 0) {
                std::unique_lock<std::mutex> lk(Q.mu);
                Q.cv.wait(lk, [&] { return !Q.q_free.empty() || Q.stop; });
                if (Q.stop) break;
                slot = Q.q_free.front(); Q.q_free.pop_front();
            }
            int64_t n = 0;
            if (r == PHI_HOST_OK) {
                n = phi_text_stream_read(ts, Q.buf[(size_t)slot].text, chunk_bytes, rerr, sizeof rerr);
                if (n < 0) { r = (int)n; }
            } else n = r;
            // While the graph is still being read and indexed the link and the HBM are idle: the chunk goes to device memory
            // now (phi_text_park_*), the host buffer is free for the next one at once, and when the index is there the reads
            // stage finds the text where the records are found anyway.  (One GPU; large files; until the index is built.)
            if (n > 0 && park_on && !graph_ready.load() && !park_go.load()) {
                // the GFA is still being read (all host threads, all of the memory bandwidth): chunks stay in their host buffers
                // as long as two more are free (parking needs two: one is read into while the other's copy is on its way); with
                // fewer, wait with this chunk in hand for the GFA or for a taker
                std::unique_lock<std::mutex> lk(Q.mu);
                Q.cv.wait(lk, [&] { return park_go.load() || graph_ready.load() || Q.q_free.size() >= 2 || Q.stop; });
            }
            if (n > 0 && park_on && park_go.load() && !graph_ready.load() && parked_bytes + n <= park_limit) {
                bool ok = true;
                if (!park) {
                    ok = phi_text_park_create(devices[0], &park) == PHI_OK;
                    for (auto &b : Q.buf) if (ok && b.text && phi_text_park_pin(park, b.text, (size_t)chunk_bytes) != PHI_OK) ok = false;
                    if (ok) park_pinned = true;
                }
                int32_t idx = -1;
                if (ok && phi_text_park_add_async(park, Q.buf[(size_t)slot].text, n, &idx) == PHI_OK) {
                    parked_bytes += n;
                    {
                        std::lock_guard<std::mutex> lk(Q.mu);
                        Q.buf.push_back(Chunk{nullptr, n, idx});
                        Q.q_full.push_back((int)Q.buf.size() - 1);
                    }
                    Q.cv.notify_all();
                    // this buffer is the copy engine's until its copy has landed: the next chunk is read into another one
                    // meanwhile, and the buffer of the copy before -- done by now -- goes back to the free ones
                    if (fly_slot >= 0) { (void)phi_text_park_wait(park, fly_idx); Q.give_free(fly_slot); }
                    fly_slot = slot; fly_idx = idx;
                    slot = -1;
                    continue;
                }
                park_on = false;                              // no room or no device yet: the usual way from here on
            }
            if (fly_slot >= 0) { (void)phi_text_park_wait(park, fly_idx); Q.give_free(fly_slot); fly_slot = -1; }
            Q.buf[(size_t)slot].n = n;                        // 0 ends the stream, a negative value ends it as failed
            {
                std::lock_guard<std::mutex> lk(Q.mu);
                Q.q_full.push_back(slot);
            }
            Q.cv.notify_all();
            slot = -1;
            if (n <= 0) break;
        }
        if (fly_slot >= 0) { (void)phi_text_park_wait(park, fly_idx); Q.give_free(fly_slot); }
        if (ts) phi_text_stream_close(ts);
        return r;
      });
    };
    start_reads();
    auto stop_reads = [&]() {
        { std::lock_guard<std::mutex> lk(Q.mu); Q.stop = true; }
        Q.cv.notify_all();
        if (f_reads.valid()) f_reads.wait();
    };

    // ---- graph (main.cpp:101-115)
    // One GPU: the walks stay TEXT in the reader (include/phi_host.h phi_gfa_read_deferred) and go to HBM as soon as the reader
    // knows where they are -- while it still enters the segment names --, and the device resolves them (phi_walk_text_*): at
    // chromosome scale the walks are 96% of the file.  Text the device path does not take (reverse steps, names of another form:
    // walk_text.hip) is resolved by the host after all, with the reference's rules.  Small files (PHI_WALK_TEXT_MIN bytes of
    // walk text, default 1 GB) and multi-GPU runs (every GPU needs the walks) stay with the host.
    phi_graph *g = nullptr;
    struct WalkText { std::shared_future<int> *ctx_ready; std::vector<phi_ctx *> *ctxs; int64_t min_bytes; int64_t bytes = 0; int rc = 0; bool sent = false; } wt{&f_ctx, &ctxs, 0};
    // (1 GB: the host threads resolve 70 MB of walk text -- a 49-walk MHC graph -- in 10 ms, hidden behind the 0.1 s the HIP runtime
    //  takes to start, while the device path has to wait for that start before its first byte moves: measured at C2, 32 MB as
    //  the threshold cost every process 40 ms)
    wt.min_bytes = getenv("PHI_WALK_TEXT_MIN") ? atoll(getenv("PHI_WALK_TEXT_MIN")) : ((int64_t)1 << 30);
    const bool defer_walks = n_dev == 1 && !(getenv("PHI_WALKS") && !strcmp(getenv("PHI_WALKS"), "host"));
    auto gfa_failed = [&]() {
        if (err[0] == 'E') fprintf(stderr, "%s\n", err);            // walk error text of ILP_index.cpp:105
        else fprintf(stderr, "[E::%s] failed to load the GFA file\n", "main");
        if (err[0] && err[0] != 'E') fprintf(stderr, "[E::%s] %s\n", "main", err);
        f_ctx.wait(); stop_reads();
        return 1;
    };
    {
        Stage st("GFA read + parse");
        int r;
        if (defer_walks)
            r = phi_gfa_read_deferred(gfa_file.c_str(), &g, [](void *user, const phi_host_walk_text *walks, int32_t n) {
                    WalkText &t = *static_cast<WalkText *>(user);
                    for (int32_t i = 0; i < n; i++) t.bytes += walks[i].n;
                    if (t.bytes < t.min_bytes) return;
                    if (t.ctx_ready->get()) return;                              // (no device: main reports it)
                    static_assert(sizeof(phi_host_walk_text) == sizeof(phi_walk_text), "the two libraries' walk text records");
                    t.rc = phi_walk_text_upload((*t.ctxs)[0], reinterpret_cast<const phi_walk_text *>(walks), n);
                    t.sent = t.rc == 0;
                }, &wt, err, sizeof err);
        else
            r = phi_gfa_read(gfa_file.c_str(), &g, err, sizeof err);
        if (r != PHI_HOST_OK) return gfa_failed();
    }
    park_go = true;                                           // (the reads text may go to device memory from here on: see start_reads)
    Q.cv.notify_all();
    if (defer_walks) {
        Stage st("walks");
        bool on_device = false;
        if (wt.rc) { fprintf(stderr, "[E::%s] walk text to the device: %s: %s\n", "main", phi_strerror(wt.rc), phi_last_error(ctxs[0])); stop_reads(); return 1; }
        if (wt.sent) {
            const char *prefix = nullptr; int32_t prefix_n = 0; const int32_t *num2id = nullptr; int64_t n_num = 0;
            uint32_t irregular = 0;
            if (phi_graph_name_index(g, &prefix, &prefix_n, &num2id, &n_num) == PHI_HOST_OK) {
                std::vector<int64_t> woff((size_t)phi_graph_n_walks(g) + 1, 0);
                const int r = phi_walk_text_resolve(ctxs[0], prefix, prefix_n, num2id, n_num, phi_graph_n_vtx(g), woff.data(), &irregular);
                if (r) { fprintf(stderr, "[E::%s] walks on the device: %s: %s\n", "main", phi_strerror(r), phi_last_error(ctxs[0])); stop_reads(); return 1; }
                if (!irregular) { phi_graph_set_walk_off(g, woff.data()); on_device = true; }
            } else {
                (void)phi_walk_text_upload(ctxs[0], nullptr, 0);                 // (lets the text on the device go)
            }
            if (timing) fprintf(stderr, "[phi] walks: %lld bytes of text %s\n", (long long)wt.bytes, on_device ? "resolved on the device" : irregular ? "irregular for the device: host" : "names not <prefix><number>: host");
        }
        if (!on_device && phi_graph_resolve_walks(g, err, sizeof err) != PHI_HOST_OK) return gfa_failed();
    }
    stamp("main");
    fprintf(stderr, "Loaded graph from: %s\n", gfa_file.c_str());
    char hap_name[4096];
    if (phi_hap_name(gfa_file.c_str(), reads_file.c_str(), hap_name, sizeof hap_name) < 0) { fprintf(stderr, "[E::%s] output name too long\n", "main"); f_ctx.wait(); stop_reads(); return 1; }

    int rc;
    {
        Stage st("wait for the device context");
        rc = f_ctx.get();
    }
    if (rc) { fprintf(stderr, "[E::%s] no usable MI355X (HIP) device %d: %s\n", "main", devices[0], phi_strerror(rc)); stop_reads(); return 1; }
    phi_ctx *ctx = ctxs[0];                                   // the context that solves and reports
    auto die_on = [&](phi_ctx *cx, const char *what, int code) {
        fprintf(stderr, "[E::%s] %s: %s: %s\n", "main", what, phi_strerror(code), phi_last_error(cx));
        stop_reads();
        return 1;
    };
    auto die = [&](const char *what, int code) { return die_on(ctx, what, code); };
    // One phase of the job on every GPU, each on its own host thread.  All threads of a phase are joined before the next
    // begins, and a collective (RCCL) is a phase of its own that is entered only when the phase before returned 0 on every
    // GPU: a rank that failed can then never leave the others waiting inside ncclCommInitRank / ncclAllReduce.
    std::atomic<bool> failed{false};
    auto run_on_all = [&](const char *what, const std::function<int(int, phi_ctx *)> &fn) -> int {
        if (n_dev == 1) { const int r = fn(0, ctx); if (r) failed = true; return r ? die(what, r) : 0; }
        std::vector<std::future<int>> fs;
        for (int i = 0; i < n_dev; i++) fs.push_back(std::async(std::launch::async, [&, i]() { const int r = fn(i, ctxs[(size_t)i]); if (r) failed = true; return r; }));
        int bad = -1, brc = 0;
        for (int i = 0; i < n_dev; i++) { const int r = fs[(size_t)i].get(); if (r && bad < 0) { bad = i; brc = r; } }
        if (bad >= 0) {
            if (brc == PHI_ERR_WALK) fprintf(stderr, "Error: %s\n", phi_last_error(ctxs[(size_t)bad]));
            return die_on(ctxs[(size_t)bad], what, brc);
        }
        return 0;
    };

    const uint32_t flags = (is_qclp ? PHI_FLAG_QCLP : 0) | (is_mixed ? PHI_FLAG_MIXED : 0);
    // the read shards of a multi-GPU run are merged by the library's own RCCL exchange (phi_comm_*)
    unsigned char comm_id[PHI_COMM_ID_BYTES];

    // ---- stage 1a: walks (ILP_index.cpp:556-611) on every GPU, while the reads are still being read
    const int32_t n_walks = phi_graph_n_walks(g);
    {
        Stage st("phi_set_graph (index build)");
        if (run_on_all("graph", [&](int, phi_ctx *cx) -> int {
                int r = phi_set_params(cx, k, w, o.threshold, recombination, flags);
                // the reference's model.optimize() has no limit (ILP_index.cpp:1412-1418): none here unless --dp-budget asks for one
                if (!r) r = phi_set_solve_budget(cx, o.dp_budget >= 0 ? o.dp_budget : 0);
                if (!r) r = phi_set_graph(cx, phi_graph_n_vtx(g), phi_graph_seq_concat(g), phi_graph_seq_off(g), phi_graph_adj_off(g),
                                          phi_graph_adj(g), n_walks, phi_graph_walk_off(g), phi_graph_walk_vtx(g), phi_graph_topo_rank(g));
                if (r == PHI_ERR_WALK && n_dev == 1) fprintf(stderr, "Error: %s\n", phi_last_error(cx));
                return r;
            })) return 1;
    }
    graph_ready = true;                                       // (the reader thread stops parking chunks: they are taken as they come now)
    Q.cv.notify_all();
    // The exchange of a multi-GPU run: the library's RCCL all-reduce.  PHI_EXCHANGE=peers takes the peer-mapped OR-gather
    // instead (one kernel per GPU, no RCCL: made for hit vectors of a few MB, every MHC-sized graph) -- opt-in until a run on
    // two or more GPUs has compared the two bit for bit: its cross-GPU loads have only ever run between contexts on ONE GPU.
    bool use_peers = false;
    void *peer_group = nullptr;
    if (n_dev > 1) {
        phi_index_info info;
        if ((rc = phi_index_stats(ctx, &info))) return die("index", rc);
        if (const char *e = getenv("PHI_EXCHANGE")) use_peers = strcmp(e, "peers") == 0;
        if (use_peers && (rc = phi_peers_create(n_dev, &peer_group))) return die("peer group", rc);
        Stage st(use_peers ? "peer group (xGMI peer access)" : "RCCL communicator");
        if (!use_peers && (rc = phi_comm_unique_id(comm_id, sizeof comm_id))) { fprintf(stderr, "[E::main] RCCL is not available: %s\n", phi_strerror(rc)); stop_reads(); return 1; }
        if (run_on_all("communicator", [&](int i, phi_ctx *cx) -> int { return use_peers ? phi_peers_join(cx, peer_group, i) : phi_comm_init(cx, comm_id, i, n_dev); })) return 1;
        fprintf(stderr, "[M::main] %d GPUs; hit vector of %lld flags merged through %s\n", n_dev, (long long)info.n_distinct_minimizers, use_peers ? "peer-mapped memory (one OR-gather kernel per GPU)" : "RCCL all-reduce");
    }

    // ---- one job per read set (-r a -o a.fa -r b -o b.fa ...): the graph, its index and the communicator are made once
    int status = 0;
    std::atomic<bool> pinned{true};
    std::once_flag pin_once;
    bool registered = false;
    for (int job = 0; job < n_jobs; job++) {
    if (job > 0) {
        // the next read set: clocks, names, the chunk queue and the reader thread start over; the contexts forget the reads
        fflush(nullptr);
        const double now = realtime();
        cpu0 += cputime();
        t0_real = now;
        { std::lock_guard<std::mutex> lk(g_marks.mu); g_marks.marks.clear(); }
        reads_file = o.reads_files[(size_t)job]; hap_file = o.hap_files[(size_t)job];
        if (phi_hap_name(gfa_file.c_str(), reads_file.c_str(), hap_name, sizeof hap_name) < 0) { fprintf(stderr, "[E::%s] output name too long\n", "main"); return 1; }
        {
            std::lock_guard<std::mutex> lk(Q.mu);
            Q.q_free.clear(); Q.q_full.clear(); Q.stop = false;
            for (int i = 0; i < N_CHUNK_BUF; i++) { Q.buf[(size_t)i].n = 0; Q.q_free.push_back(i); }
        }
        rerr[0] = 0;
        start_reads();
        if (run_on_all("reset", [&](int, phi_ctx *cx) -> int { return phi_reset_reads(cx); })) return 1;
        stamp("main");
        fprintf(stderr, "Loaded graph from: %s\n", gfa_file.c_str());
    }
    // ---- reads (main.cpp:136-137) and stage 1b/2a (:615-655), chunk by chunk.  The chunks are taken in stream order,
    //      one GPU at a time (a chunk needs the unfinished rest of the one before); the sketch of a chunk runs on
    //      behind the turn.  Text that is not laid out regularly goes through the host reader from that byte on.
    std::atomic<int> n_chunks{0};
    std::mutex turn_mu;
    std::vector<char> carry;                                  // several GPUs: the bytes the last turn left unfinished
    bool stream_done = false;                                 // under turn_mu
    int64_t stream_fed = 0;                                   // under turn_mu: bytes of the stream taken from the queue so far
    int64_t host_parsed_bases = 0;
    // the host reader over `prefix` + the rest of the queue -> phi_add_reads on this GPU (under turn_mu)
    auto finish_on_host = [&](phi_ctx *cx, const char *prefix, int64_t n_prefix, bool rest_of_queue, int64_t stream_offset) -> int {
        Stage st("host reader (kseq state machine)");
        QueueBlocks qb{&Q, -1, &park, {}};
        phi_reads_stream *rs = nullptr;
        if (phi_reads_stream_open_blocks(prefix, n_prefix, rest_of_queue ? next_block_from_queue : nullptr, &qb, stream_offset, &rs, rerr, sizeof rerr) != PHI_HOST_OK) return PHI_ERR_INVALID;
        const int64_t cap_b = std::max<int64_t>((int64_t)1 << 20, std::min<int64_t>(chunk_bytes, (int64_t)64 << 20)), cap_r = cap_b / 32 + 1024;
        std::vector<char> hb((size_t)cap_b);
        std::vector<int64_t> ho((size_t)cap_r + 1);
        int r = PHI_OK;
        for (;;) {
            const int64_t n = phi_reads_stream_next(rs, hb.data(), cap_b, ho.data(), cap_r, rerr, sizeof rerr);
            if (n < 0) { r = PHI_ERR_INVALID; break; }
            if (n == 0) break;
            host_parsed_bases += ho[(size_t)n];
            if ((r = phi_add_reads(cx, hb.data(), ho.data(), n))) break;
        }
        phi_reads_stream_close(rs);
        if (qb.held >= 0) Q.give_free(qb.held);
        if (r == PHI_ERR_INVALID && rerr[0]) fprintf(stderr, "[E::main] %s\n", rerr);
        return r;
    };
    {
        Stage st("reads: text -> device, records, sketch");
        if (run_on_all("reads", [&](int, phi_ctx *cx) -> int {
                int r = phi_reads_text_begin(cx, chunk_bytes);
                bool open = r == PHI_OK;
                while (!r) {
                    std::unique_lock<std::mutex> turn(turn_mu);
                    if (stream_done || failed) break;
                    const int slot = Q.take_full();
                    Chunk &cb = *Q.at(slot);
                    if (cb.n <= 0) {
                        // the end of the stream (left in the queue for the other GPUs): what is still unfinished is the file's
                        // last record, whose end only the end of the file shows -- the host reader's
                        stream_done = true;
                        if (cb.n < 0) { r = PHI_ERR_INVALID; fprintf(stderr, "[E::main] %s\n", rerr); break; }
                        const char *pend = nullptr;
                        int64_t n_pend = 0;
                        r = phi_reads_text_end(cx, &pend, &n_pend, nullptr);
                        open = false;
                        if (!r && n_dev > 1) { pend = carry.data(); n_pend = (int64_t)carry.size(); }
                        if (!r && n_pend) r = finish_on_host(cx, pend, n_pend, false, stream_fed - n_pend);
                        break;
                    }
                    if (++n_chunks >= 2)
                        // a file of more than one chunk: pin the buffers, so that the device copy of every further
                        // chunk is a direct DMA (pinning takes milliseconds: not worth it for a single chunk)
                        std::call_once(pin_once, [&]() {
                            if (park_pinned.load()) return;    // (the reader thread pinned them when it began to park chunks)
                            registered = true;
                            std::lock_guard<std::mutex> lk(Q.mu);      // (the reader thread may be adding an entry for a parked chunk)
                            for (auto &b : Q.buf) if (b.text && phi_host_register(cx, b.text, (size_t)chunk_bytes) != PHI_OK) pinned = false;
                        });
                    stream_fed += cb.n;
                    int32_t irr_carry = 0, irr = 0;
                    if (n_dev > 1 && !carry.empty()) r = phi_add_reads_text(cx, carry.data(), (int64_t)carry.size(), &irr_carry);
                    if (!r && !irr_carry) {
                        if (cb.parked >= 0) {
                            r = phi_add_reads_text_parked(cx, park, cb.parked, &irr);
                            if (!r) (void)phi_text_park_release(park, cb.parked);
                        } else r = phi_add_reads_text(cx, cb.text, cb.n, &irr);
                    }
                    if (!r && (irr_carry || irr)) {
                        // not one of the two regular layouts: the exact state machine takes the stream from the first byte not taken
                        stream_done = true;
                        const char *pend = nullptr;
                        int64_t n_pend = 0;
                        r = phi_reads_text_end(cx, &pend, &n_pend, nullptr);
                        open = false;
                        std::vector<char> all;
                        if (!r && irr_carry) {                 // (the carry, fed as a piece of its own, was what did not fit: this chunk follows it)
                            all.assign(pend, pend + n_pend);
                            all.insert(all.end(), cb.text, cb.text + cb.n);
                            pend = all.data(); n_pend = (int64_t)all.size();
                        }
                        Q.give_free(slot);
                        if (!r) {
                            if (timing) fprintf(stderr, "[phi timing] main: the reads text is not regular FASTA / 4-line FASTQ: host reader from the first byte not taken\n");
                            r = finish_on_host(cx, pend, n_pend, true, stream_fed - n_pend);
                        }
                        break;
                    }
                    if (!r && n_dev > 1) {
                        const char *p = nullptr;
                        int64_t n = 0;
                        r = phi_reads_text_detach_carry(cx, &p, &n);
                        if (!r) carry.assign(p, p + n);
                    }
                    Q.give_free(slot);
                }
                if (open) { const int r2 = phi_reads_text_end(cx, nullptr, nullptr, nullptr); if (!r) r = r2; }
                return r;
            })) return 1;
    }
    if (f_reads.get() != PHI_HOST_OK) { fprintf(stderr, "[E::%s] %s\n", "main", rerr); return 1; }
    if (n_dev > 1) {
        Stage st(use_peers ? "exchange (peer-mapped)" : "exchange (RCCL)");
        if (run_on_all("exchange", [&](int, phi_ctx *cx) -> int { return use_peers ? phi_peers_exchange(cx) : phi_comm_exchange(cx); })) return 1;
    }
    int64_t total_reads = 0;
    for (phi_ctx *cx : ctxs) {
        int64_t nr = 0;
        if ((rc = phi_reads_stats(cx, &nr, nullptr, nullptr, nullptr))) return die_on(cx, "reads", rc);
        total_reads += nr;
    }
    stamp("ILP_function");
    fprintf(stderr, "Graph has %d vertices, %d walks and read has %d reads\n", phi_graph_n_vtx(g), n_walks, (int)total_reads);
    // ---- stages 2b-3 (:670-1525)
    phi_result res;
    {
        Stage st("phi_solve (filter, exact solve, decode)");
        if ((rc = phi_solve(ctx, &res))) return die("solve", rc);
    }
    const double t_report = realtime();

    fprintf(stderr, "Number of Minimizers\n");
    for (int32_t h = 0; h < n_walks; h++) fprintf(stderr, "%s : %d\n", phi_graph_hap_name(g, h), (int)res.n_minimizers[h]);
    if (debug) {                                              // ILP_index.cpp:591-604
        std::vector<int64_t> hist((size_t)n_walks + 1, 0);
        int64_t n_distinct = 0;
        if ((rc = phi_walk_sharing(ctx, hist.data(), n_walks + 1, &n_distinct))) return die("sharing histogram", rc);
        fprintf(stderr, "Shared fraction of unique kmers by haplotypes\n");
        for (int32_t i = 1; i <= n_walks; i++)
            fprintf(stderr, "[Haplotypes: %d, fraction of unique shared kmers: %.5f]\n", i, (float)hist[i] / (float)n_distinct);
    }
    stamp("ILP_function");
    fprintf(stderr, "Haplotypes sketched\n");
    stamp("ILP_function");
    fprintf(stderr, "Indexed reads with spectrum size: %d\n", (int)res.spectrum_size);
    fprintf(stderr, "Number of Anchors\n");
    for (int32_t h = 0; h < n_walks; h++) fprintf(stderr, "%s : %d\n", phi_graph_hap_name(g, h), (int)res.n_anchors[h]);
    stamp("ILP_function");
    fprintf(stderr, "Filtered/Retained Minimizers: %.2f/%.2f%%\n", (float)res.filtered / (float)res.spectrum_size * 100,
            (float)res.retained / (float)res.spectrum_size * 100);
    stamp("ILP_function");
    fprintf(stderr, "%s model started\n", is_qclp ? "QP" : "ILP");
    stamp("ILP_function");
    fprintf(stderr, "%.2f%% Minimizers are in ILP\n", (res.n_in_model * 100.0) / res.spectrum_size);
    stamp("ILP_function");
    fprintf(stderr, "Minimizer constraints added to the model\n");
    stamp("ILP_function");
    fprintf(stderr, "%s\n", is_mixed ? "Using Mixed Integer Programming" : "Using Integer Programming");
    stamp("ILP_function");
    fprintf(stderr, "Optimized expanded graph constructed\n");
    stamp("ILP_function");
    fprintf(stderr, "Model optimized\n");
    if (debug || !res.optimal)
        fprintf(stderr, "[M::%s] objective %lld (upper bound %lld, %s) after %d DP run(s); %lld minimisers covered, %d w-node(s)\n", "solve",
                (long long)res.objective, (long long)res.upper_bound, res.optimal ? "proven optimal" : "NOT proven optimal", res.n_dp_runs,
                (long long)res.n_covered, res.n_switches);

    // ---- recombination report (:1508-1550): segments in output coordinates
    fprintf(stderr, "Recombination count: %d\n", res.recombination_count);
    fprintf(stderr, "Recombined haplotypes: ");
    {
        const int64_t *so = phi_graph_seq_off(g);
        int64_t str_id = 0, prev_str_id = 0;
        int32_t prev_hap = res.n_path ? res.path_hap[0] : 0;
        for (int64_t i = 0; i < res.n_path; i++) {
            const int32_t v = res.path_vtx[i];
            if (i > 0 && res.path_hap[i] != prev_hap) {
                // the reference adds the vertex length before testing the label (:1515-1523)
                str_id += so[v + 1] - so[v];
                fprintf(stderr, ">(%s,[%lld,%lld])", phi_graph_hap_name(g, prev_hap), (long long)prev_str_id, (long long)(str_id - 1));
                prev_hap = res.path_hap[i];
                prev_str_id = str_id;
            } else {
                str_id += so[v + 1] - so[v];
            }
        }
        if (res.n_path) fprintf(stderr, ">(%s,[%lld,%lld])", phi_graph_hap_name(g, prev_hap), (long long)prev_str_id, (long long)(str_id - 1));
        fprintf(stderr, "\n");
    }
    g_marks.add("report (log lines)", t_report, realtime());

    // ---- FASTA (:1577-1598)
    {
        Stage st("FASTA write");
        std::unique_ptr<char[]> seq(new char[(size_t)(res.hap_len > 0 ? res.hap_len : 1)]);      // (not zero-filled: every byte is written)
        if ((rc = phi_path_sequence(ctx, seq.get(), res.hap_len))) return die("sequence", rc);
        if (phi_write_fasta(hap_file.c_str(), hap_name, seq.get(), res.hap_len) != PHI_HOST_OK) {
            fprintf(stderr, "[E::%s] cannot write %s\n", "main", hap_file.c_str());
            return 1;
        }
    }
    stamp("ILP_function");
    fprintf(stderr, "Haplotype of size: %d written to: %s\n", (int)res.hap_len, hap_file.c_str());

    fprintf(stderr, "[M::%s] PHI Version: %s\n", "main", PHI_VERSION);
    fprintf(stderr, "[M::%s] CMD:", "main");
    for (int i = 0; i < o.argc; ++i) fprintf(stderr, " %s", o.argv[i]);
    fprintf(stderr, "\n[M::%s] Real time: %.3f sec; CPU: %.3f sec; Peak RSS: %.3f GB\n", "main", realtime() - t0_real, cputime(),
            peakrss() / 1024.0 / 1024.0 / 1024.0);
    if (timing) {
        if (parked_bytes) fprintf(stderr, "[phi timing] main: %lld bytes of the reads text waited in device memory for the index\n", (long long)parked_bytes);
        fprintf(stderr, "[phi timing] main: %d text chunk(s) of up to %lld bytes%s on %d GPU(s); %lld bases through the host reader; FASTA closed at epoch %.6f%s\n",
                n_chunks.load(), (long long)chunk_bytes, n_chunks >= 2 && pinned ? ", pinned" : "", n_dev, (long long)host_parsed_bases, realtime(),
                o.detached ? "; teardown detached" : "");
        g_marks.print();
        // what the peak is made of: the mapped GFA file's own pages count as resident (RssFile / RssShmem), anonymous memory is the rest
        if (FILE *fp = fopen("/proc/self/status", "r")) {
            char line[256];
            long hwm = -1, anon = -1, file = -1, shm = -1;
            while (fgets(line, sizeof line, fp)) {
                if (!strncmp(line, "VmHWM:", 6)) hwm = atol(line + 6);
                else if (!strncmp(line, "RssAnon:", 8)) anon = atol(line + 8);
                else if (!strncmp(line, "RssFile:", 8)) file = atol(line + 8);
                else if (!strncmp(line, "RssShmem:", 9)) shm = atol(line + 9);
            }
            fclose(fp);
            fprintf(stderr, "[phi timing] main: resident now: anonymous %.3f GB, mapped files %.3f GB (the GFA among them); peak %.3f GB\n", anon / 1048576.0, (file + shm) / 1048576.0, hwm / 1048576.0);
        }
    }
    if (!res.optimal) {
        // the reference returns only what model.optimize() proved (ILP_index.cpp:1418); here that can only fall short when
        // --dp-budget set a limit: the path written is feasible and within the printed bound, the exit status says so
        fprintf(stderr, "[W::main] the path written is NOT proven optimal: the exact search used its budget of %d DP runs "
                        "(objective %lld, proven upper bound %lld); raise it with --dp-budget N (0 = no limit, the default)\n",
                res.n_dp_runs, (long long)res.objective, (long long)res.upper_bound);
        status = 3;
    }
    }   // jobs
    if (getenv("PHI_FULL_TEARDOWN")) {                        // (leak checks: give everything back in order)
        stop_reads();
        if (park) phi_text_park_destroy(park);                 // (unpins the chunk buffers it pinned)
        for (auto &cb : Q.buf) {
            if (registered && cb.text) (void)phi_host_unregister(ctx, cb.text);
            free(cb.text);
        }
        phi_graph_free(g);
        for (phi_ctx *cx : ctxs) phi_ctx_destroy(cx);
    }
    return status;
}

int main(int argc, char *argv[])
{
    Options o;
    int help = 0;
    static struct option long_options[] = {{"version", no_argument, 0, 300}, {"device", required_argument, 0, 301}, {"dp-budget", required_argument, 0, 302},
                                           {"devices", required_argument, 0, 303}, {"shard-min-bases", required_argument, 0, 304}, {0, 0, 0, 0}};
    int c;
    // main.cpp:38 declares -h with an argument; a bare -h falls into the usage branch either way
    while ((c = getopt_long(argc, argv, "x:d:c:l:s:m:R:q:T:N:h:k:w:t:g:r:o:DS", long_options, nullptr)) >= 0) {
        if (c == 'w') o.w = atoi(optarg);
        else if (c == 'k') o.k = atoi(optarg);
        else if (c == 't') o.n_threads = atoi(optarg);
        else if (c == 'm') o.is_mixed = atoi(optarg);
        else if (c == 'g') o.gfa_file = optarg;
        else if (c == 'R') o.recombination = atoi(optarg);
        else if (c == 'q') o.is_qclp = atoi(optarg);
        else if (c == 'N') o.is_naive = atoi(optarg);
        else if (c == 'T') o.threshold = (float)atof(optarg);
        else if (c == 'r') { o.reads_file = optarg; o.reads_files.push_back(optarg); }
        else if (c == 'o') { o.hap_file = optarg; o.hap_files.push_back(optarg); }
        else if (c == 'c') o.max_occ = atoi(optarg);
        else if (c == 'd') o.debug = atoi(optarg);
        else if (c == 'h' || c == '?') help = 1;
        else if (c == 300) { fprintf(stderr, "PHI version: %s\n", PHI_VERSION); return 0; }
        else if (c == 301) o.device = atoi(optarg);
        else if (c == 302) o.dp_budget = atoll(optarg);
        else if (c == 304) o.shard_min_bases = std::max<long long>(1, atoll(optarg));
        else if (c == 303) {                                   // --devices 0,1,2,...: shard the reads over these GPUs
            o.devices.clear();
            for (const char *p = optarg; *p;) {
                char *end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p || d < 0) { fprintf(stderr, "[E::main] --devices takes a comma-separated list of GPU ordinals\n"); return 1; }
                o.devices.push_back((int)d);
                p = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') { fprintf(stderr, "[E::main] --devices takes a comma-separated list of GPU ordinals\n"); return 1; }
            }
        }
    }
    if (argc < 2 || o.gfa_file.empty() || o.reads_file.empty() || o.hap_file.empty() || help) {
        usage(stderr, o.k, o.w, o.recombination, o.is_qclp, o.is_mixed, o.threshold, o.n_threads, o.gfa_file.c_str(), o.reads_file.c_str(), o.hap_file.c_str(), o.debug);
        return 1;
    }
    if (o.reads_files.size() != o.hap_files.size()) { fprintf(stderr, "[E::main] %zu -r but %zu -o: several read sets against one graph are given as -r a.fq -o a.fa -r b.fq -o b.fa ...\n", o.reads_files.size(), o.hap_files.size()); return 1; }
    o.argc = argc; o.argv = argv;
    t0_real = realtime();

    // The work is done by a child; this process returns the child's status as soon as the child reports it -- after the
    // FASTA is closed and the log written, before the teardown.  Not under a profiler or any other preloaded library that
    // may have started the GPU runtime in this process already (a runtime does not survive a fork), and not when asked.
    bool detach = false;                                      // (opt-in: see the head of this file)
    if (const char *e = getenv("PHI_DETACH")) detach = atoi(e) != 0;
    if (getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("HSA_TOOLS_LIB") || getenv("ROCPROF_OUTPUT_PATH")) detach = false;
    if (const char *pl = getenv("LD_PRELOAD"))
        for (const char *tool : {"rocprof", "roctracer", "roctx", "rocsys", "omnitrace", "rocpd"})
            if (strstr(pl, tool)) detach = false;
    int report_fd = -1;
    if (detach) {
        int pfd[2];
        if (pipe(pfd) == 0) {
            fflush(nullptr);
            const pid_t pid = fork();
            if (pid > 0) {
                close(pfd[1]);
                unsigned char st = 0;
                ssize_t r;
                do r = read(pfd[0], &st, 1); while (r < 0 && errno == EINTR);
                if (r == 1) _exit(st);
                int ws = 0;                                   // the pipe closed without a status: the child died
                while (waitpid(pid, &ws, 0) < 0 && errno == EINTR) {}
                _exit(WIFEXITED(ws) ? WEXITSTATUS(ws) : 128 + (WIFSIGNALED(ws) ? WTERMSIG(ws) : 0));
            } else if (pid == 0) {
                close(pfd[0]);
                report_fd = pfd[1];
                o.detached = true;
                // (the parent's death ends the child: a killed parent takes it along -- and so does the parent's ordinary exit
                //  right after the status arrived: the child is then inside its teardown, which the signal merely cuts short)
                (void)prctl(PR_SET_PDEATHSIG, SIGTERM);
            } else { close(pfd[0]); close(pfd[1]); }           // no fork: one process
        }
    }
    const int status = run(o);
    fflush(nullptr);
    if (report_fd >= 0) {
        const unsigned char st = (unsigned char)status;
        ssize_t r;
        do r = write(report_fd, &st, 1); while (r < 0 && errno == EINTR);
        // nothing more is written: let a pipe that captures the log see its end now, not when the teardown is over
        close(report_fd); close(0); close(1); close(2);
    }
    // the arrays, the contexts and the runtime are given back by the exit itself (PHI_FULL_TEARDOWN=1 frees them one by one first)
    _exit(status);
}
