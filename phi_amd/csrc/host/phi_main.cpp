// phi_main.cpp -- the `PHI` command-line driver: same flags, stderr log lines and FASTA output as
// the reference's src/main.cpp + the logging of ILP_index::ILP_function, with the hot path behind
// the C ABI of include/phi_amd.h (HIP kernels on one MI355X).  Own implementation.
//
//   ./PHI -g <target.gfa> -r <reads.fa> -o <haplotype.fasta> [-k -w -R -q -m -T -t -d -N -c]
//         [--device N | --devices 0,1,..] [--dp-budget RUNS]
//
// --devices: one context and one host thread per GPU; every GPU builds the full index, the read chunks are
// handed out through a work queue (the shards balance themselves), the library's RCCL exchange (phi_comm_*)
// merges hit vectors and spectra once, and the first GPU solves and reports.
//
// Flag semantics (main.cpp:38-95): -q (IQP/ILP), -m (mixed/integer), -N (naive expanded graph)
// choose between formulations with the same optimum; they are accepted and mapped onto the one
// exact solver.  -t only sized OpenMP/Gurobi thread pools and is accepted and ignored.  The log
// lines scraped by the reference's evaluation scripts (data/postprocessing_2_MIQP.py:55-79) keep
// their exact formats.
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <string>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <thread>
#include <vector>
#include "../../../include/phi_amd.h"
#include "../../../include/phi_host.h"

#define PHI_VERSION "1.0-mi355x"

static double t0_real;
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
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
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

int main(int argc, char *argv[])
{
    int k = 31, w = 25, n_threads = 4, recombination = 100, is_qclp = 1, is_naive = 0, is_mixed = 1, debug = 0, help = 0;
    int device = 0, max_occ = 5000;
    std::vector<int> devices;                                 // --devices: one context (and host thread) per GPU
    long long dp_budget = -1;                                 // --dp-budget: DP runs of the exact search (default: the library's)
    float threshold = 1.0f;
    std::string gfa_file, reads_file, hap_file;
    static struct option long_options[] = {{"version", no_argument, 0, 300}, {"device", required_argument, 0, 301}, {"dp-budget", required_argument, 0, 302}, {"devices", required_argument, 0, 303}, {0, 0, 0, 0}};
    int c;
    // main.cpp:38 declares -h with an argument; a bare -h falls into the usage branch either way
    while ((c = getopt_long(argc, argv, "x:d:c:l:s:m:R:q:T:N:h:k:w:t:g:r:o:DS", long_options, nullptr)) >= 0) {
        if (c == 'w') w = atoi(optarg);
        else if (c == 'k') k = atoi(optarg);
        else if (c == 't') n_threads = atoi(optarg);
        else if (c == 'm') is_mixed = atoi(optarg);
        else if (c == 'g') gfa_file = optarg;
        else if (c == 'R') recombination = atoi(optarg);
        else if (c == 'q') is_qclp = atoi(optarg);
        else if (c == 'N') is_naive = atoi(optarg);
        else if (c == 'T') threshold = (float)atof(optarg);
        else if (c == 'r') reads_file = optarg;
        else if (c == 'o') hap_file = optarg;
        else if (c == 'c') max_occ = atoi(optarg);
        else if (c == 'd') debug = atoi(optarg);
        else if (c == 'h' || c == '?') help = 1;
        else if (c == 300) { fprintf(stderr, "PHI version: %s\n", PHI_VERSION); return 0; }
        else if (c == 301) device = atoi(optarg);
        else if (c == 302) dp_budget = atoll(optarg);
        else if (c == 303) {                                   // --devices 0,1,2,...: shard the reads over these GPUs
            devices.clear();
            for (const char *p = optarg; *p;) {
                char *end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p || d < 0) { fprintf(stderr, "[E::main] --devices takes a comma-separated list of GPU ordinals\n"); return 1; }
                devices.push_back((int)d);
                p = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') { fprintf(stderr, "[E::main] --devices takes a comma-separated list of GPU ordinals\n"); return 1; }
            }
        }
    }
    (void)max_occ; (void)is_naive; (void)n_threads;
    if (argc < 2 || gfa_file.empty() || reads_file.empty() || hap_file.empty() || help) {
        usage(stderr, k, w, recombination, is_qclp, is_mixed, threshold, n_threads, gfa_file.c_str(), reads_file.c_str(), hap_file.c_str(), debug);
        return 1;
    }
    t0_real = realtime();
    char err[512] = "";

    // The device context (HIP initialisation) and the reads file are prepared by two host threads
    // while this one parses the graph: the three are independent.
    if (devices.empty()) devices.push_back(device);
    const int n_dev = (int)devices.size();
    for (int i = 0; i < n_dev; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) { fprintf(stderr, "[E::main] --devices names GPU %d twice\n", devices[i]); return 1; }
    std::vector<phi_ctx *> ctxs((size_t)n_dev, nullptr);
    const bool timing = getenv("PHI_TIMING") != nullptr;
    std::future<int> f_ctx = std::async(std::launch::async, [&]() {
        // one host thread per GPU (each context initialises its own device)
        std::vector<std::future<int>> fs;
        for (int i = 0; i < n_dev; i++)
            fs.push_back(std::async(std::launch::async, [&, i]() { return phi_ctx_create(devices[(size_t)i], &ctxs[(size_t)i]); }));
        int r = 0;
        for (auto &f : fs) { const int ri = f.get(); if (ri && !r) r = ri; }
        if (timing) fprintf(stderr, "[phi timing] main: %d device context(s) ready at %.3f s\n", n_dev, realtime() - t0_real);
        return r;
    });
    // Reads are streamed: a host thread parses the file chunk by chunk into three
    // buffers (pinned once the device context exists) while this thread parses the graph, builds the
    // index and then hands every finished chunk to phi_add_reads -- parse / inflate of chunk i+1
    // overlaps the device copy and the kernels of chunk i, and host memory stays bounded.
    struct Chunk { char *bases; int64_t *off; int64_t n_reads; };
    const int64_t chunk_bases = getenv("PHI_READ_CHUNK") ? std::max<int64_t>(1024, atoll(getenv("PHI_READ_CHUNK"))) : ((int64_t)64 << 20);
    const int64_t chunk_reads = chunk_bases / 64 + 1024;
    const int N_CHUNK_BUF = 2 + n_dev;                       // one in flight per GPU, two with the parser
    std::vector<Chunk> chunk_buf(N_CHUNK_BUF);
    for (auto &cb : chunk_buf) {
        cb.bases = (char *)malloc((size_t)chunk_bases);
        cb.off = (int64_t *)malloc((size_t)(chunk_reads + 1) * sizeof(int64_t));
        cb.n_reads = 0;
        if (!cb.bases || !cb.off) { fprintf(stderr, "[E::%s] out of memory\n", __func__); return 1; }
    }
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::deque<int> q_free, q_full;                          // buffer indices; a full entry with n_reads == 0 ends the stream
    for (int i = 0; i < N_CHUNK_BUF; i++) q_free.push_back(i);
    char rerr[512] = "";
    int64_t total_reads = 0;
    bool stop_reader = false;
    std::future<int> f_reads = std::async(std::launch::async, [&]() {
        phi_reads_stream *rs = nullptr;
        int r = phi_reads_stream_open(reads_file.c_str(), &rs, rerr, sizeof rerr);
        for (;;) {
            int slot;
            {
                std::unique_lock<std::mutex> lk(q_mu);
                q_cv.wait(lk, [&] { return !q_free.empty() || stop_reader; });
                if (stop_reader) break;
                slot = q_free.front(); q_free.pop_front();
            }
            int64_t n = 0;
            if (r == PHI_HOST_OK) {
                n = phi_reads_stream_next(rs, chunk_buf[slot].bases, chunk_bases, chunk_buf[slot].off, chunk_reads, rerr, sizeof rerr);
                if (n < 0) { r = (int)n; n = 0; }
            }
            chunk_buf[slot].n_reads = n;
            {
                std::lock_guard<std::mutex> lk(q_mu);
                q_full.push_back(slot);
            }
            q_cv.notify_all();
            if (n == 0) break;
        }
        if (rs) { total_reads = phi_reads_stream_reads(rs); phi_reads_stream_close(rs); }
        if (timing) fprintf(stderr, "[phi timing] main: reads parsed at %.3f s\n", realtime() - t0_real);
        return r;
    });
    auto stop_reads = [&]() {
        { std::lock_guard<std::mutex> lk(q_mu); stop_reader = true; }
        q_cv.notify_all();
        if (f_reads.valid()) f_reads.wait();
    };

    // ---- graph (main.cpp:101-115)
    phi_graph *g = nullptr;
    if (phi_gfa_read(gfa_file.c_str(), &g, err, sizeof err) != PHI_HOST_OK) {
        if (err[0] == 'E') fprintf(stderr, "%s\n", err);            // walk error text of ILP_index.cpp:105
        else fprintf(stderr, "[E::%s] failed to load the GFA file\n", __func__);
        if (err[0] && err[0] != 'E') fprintf(stderr, "[E::%s] %s\n", __func__, err);
        f_ctx.wait(); stop_reads();
        return 1;
    }
    stamp(__func__);
    fprintf(stderr, "Loaded graph from: %s\n", gfa_file.c_str());
    char hap_name[4096];
    if (phi_hap_name(gfa_file.c_str(), reads_file.c_str(), hap_name, sizeof hap_name) < 0) { fprintf(stderr, "[E::%s] output name too long\n", __func__); f_ctx.wait(); stop_reads(); return 1; }

    int rc = f_ctx.get();
    if (rc) { fprintf(stderr, "[E::%s] no usable MI355X (HIP) device %d: %s\n", __func__, devices[0], phi_strerror(rc)); stop_reads(); return 1; }
    phi_ctx *ctx = ctxs[0];                                   // the context that solves and reports
    auto die_on = [&](phi_ctx *cx, const char *what, int code) {
        fprintf(stderr, "[E::%s] %s: %s: %s\n", "main", what, phi_strerror(code), phi_last_error(cx));
        stop_reads();
        return 1;
    };
    auto die = [&](const char *what, int code) { return die_on(ctx, what, code); };
    // rc of the first device thread that failed, with its context
    auto run_on_all = [&](const char *what, const std::function<int(int, phi_ctx *)> &fn) -> int {
        if (n_dev == 1) { const int r = fn(0, ctx); return r ? die(what, r) : 0; }
        std::vector<std::future<int>> fs;
        for (int i = 0; i < n_dev; i++) fs.push_back(std::async(std::launch::async, [&, i]() { return fn(i, ctxs[(size_t)i]); }));
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
    if (n_dev > 1 && (rc = phi_comm_unique_id(comm_id, sizeof comm_id))) { fprintf(stderr, "[E::main] RCCL is not available: %s\n", phi_strerror(rc)); stop_reads(); return 1; }

    // ---- stage 1a: walks (ILP_index.cpp:556-611) on every GPU, while the reads are still being parsed
    const int32_t n_walks = phi_graph_n_walks(g);
    if (run_on_all("graph", [&](int i, phi_ctx *cx) -> int {
            int r = phi_set_params(cx, k, w, threshold, recombination, flags);
            if (!r && dp_budget >= 0) r = phi_set_solve_budget(cx, dp_budget);
            if (!r) r = phi_set_graph(cx, phi_graph_n_vtx(g), phi_graph_seq_concat(g), phi_graph_seq_off(g), phi_graph_adj_off(g),
                                      phi_graph_adj(g), n_walks, phi_graph_walk_off(g), phi_graph_walk_vtx(g), phi_graph_topo_rank(g));
            if (r == PHI_ERR_WALK && n_dev == 1) fprintf(stderr, "Error: %s\n", phi_last_error(cx));
            if (!r && n_dev > 1) r = phi_comm_init(cx, comm_id, i, n_dev);
            return r;
        })) return 1;

    // ---- reads (main.cpp:136-137) and stage 1b/2a (:615-655), chunk by chunk: every GPU takes the next
    //      finished chunk (a work queue: the shards balance themselves), then the one exchange of the job
    std::atomic<int> n_chunks{0};
    std::atomic<bool> pinned{true};
    std::once_flag pin_once;
    if (run_on_all("reads", [&](int, phi_ctx *cx) -> int {
            for (;;) {
                int slot;
                {
                    std::unique_lock<std::mutex> lk(q_mu);
                    q_cv.wait(lk, [&] { return !q_full.empty(); });
                    slot = q_full.front();
                    if (chunk_buf[(size_t)slot].n_reads == 0) break;          // end of the stream: left in the queue for the other GPUs
                    q_full.pop_front();
                }
                const Chunk &cb = chunk_buf[(size_t)slot];
                if (++n_chunks >= 2)
                    // a file of more than one chunk: pin the buffers, so that the device copy of every further
                    // chunk is a direct DMA (pinning takes milliseconds: not worth it for a single chunk)
                    std::call_once(pin_once, [&]() {
                        for (auto &b : chunk_buf) if (phi_host_register(cx, b.bases, (size_t)chunk_bases) != PHI_OK) pinned = false;
                    });
                const int r = phi_add_reads(cx, cb.bases, cb.off, cb.n_reads);
                if (r) return r;
                {
                    std::lock_guard<std::mutex> lk(q_mu);
                    q_free.push_back(slot);
                }
                q_cv.notify_all();
            }
            return n_dev > 1 ? phi_comm_exchange(cx) : PHI_OK;
        })) return 1;
    if (f_reads.get() != PHI_HOST_OK) { fprintf(stderr, "[E::%s] %s\n", __func__, rerr); return 1; }
    stamp("ILP_function");
    fprintf(stderr, "Graph has %d vertices, %d walks and read has %d reads\n", phi_graph_n_vtx(g), n_walks, (int)total_reads);
    // ---- stages 2b-3 (:670-1525)
    phi_result res;
    if ((rc = phi_solve(ctx, &res))) return die("solve", rc);

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

    // ---- FASTA (:1577-1598)
    std::vector<char> seq((size_t)(res.hap_len > 0 ? res.hap_len : 1));
    if ((rc = phi_path_sequence(ctx, seq.data(), res.hap_len))) return die("sequence", rc);
    if (phi_write_fasta(hap_file.c_str(), hap_name, seq.data(), res.hap_len) != PHI_HOST_OK) {
        fprintf(stderr, "[E::%s] cannot write %s\n", __func__, hap_file.c_str());
        return 1;
    }
    stamp("ILP_function");
    fprintf(stderr, "Haplotype of size: %d written to: %s\n", (int)res.hap_len, hap_file.c_str());

    fprintf(stderr, "[M::%s] PHI Version: %s\n", __func__, PHI_VERSION);
    fprintf(stderr, "[M::%s] CMD:", __func__);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fprintf(stderr, "\n[M::%s] Real time: %.3f sec; CPU: %.3f sec; Peak RSS: %.3f GB\n", __func__, realtime() - t0_real, cputime(),
            peakrss() / 1024.0 / 1024.0 / 1024.0);
    if (timing) fprintf(stderr, "[phi timing] main: %d read chunk(s) of up to %lld bases%s on %d GPU(s)\n", n_chunks.load(), (long long)chunk_bases, n_chunks >= 2 && pinned ? ", pinned" : "", n_dev);
    for (auto &cb : chunk_buf) {
        if (n_chunks >= 2) (void)phi_host_unregister(ctx, cb.bases);
        free(cb.bases); free(cb.off);
    }
    phi_graph_free(g);
    for (phi_ctx *cx : ctxs) phi_ctx_destroy(cx);
    if (!res.optimal) {
        // the reference returns only what model.optimize() proved (ILP_index.cpp:1418): an unproven path is
        // written (it is feasible and within the printed bound) but the exit status says so
        fprintf(stderr, "[W::main] the path written is NOT proven optimal: the exact search used its budget of %d DP runs "
                        "(objective %lld, proven upper bound %lld); raise it with --dp-budget N (0 = no limit)\n",
                res.n_dp_runs, (long long)res.objective, (long long)res.upper_bound);
        return 3;
    }
    return 0;
}
