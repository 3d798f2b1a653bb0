/*
 * phi_oracle.c -- CPU restatement of PHI's hot path (stages 1-2: sketch, match, filter).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under phi_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker.  It is a scalar, string-based restatement in this repo's own words of
 * the algorithm in the reference (paths relative to /root/reference):
 *
 *   hash            src/MurmurHash3.cpp:60-63,81-90,255-332 ; src/ILP_index.cpp:10-18
 *   reverse strand  src/ILP_index.cpp:330-357
 *   read sketch     src/ILP_index.cpp:447-493   (compute_hashes)
 *   walk sketch     src/ILP_index.cpp:359-445   (index_kmers)
 *   read spectrum   src/ILP_index.cpp:615-638
 *   anchors         src/ILP_index.cpp:495-526, 643-655
 *   filter          src/ILP_index.cpp:670-743
 *   model counter   src/ILP_index.cpp:782-883   (which minimisers get a z_i)
 *
 * Pinning (see DESIGN.md "Oracle"): the murmur fold is checked against the reference's own
 * MurmurHash3.cpp compiled into oracle/_ref; the stage counters are checked against the
 * counters the reference produced on test/test.gfa+read.fa and test/MHC_4.gfa.gz+
 * CHM13_reads.fq.gz.  The solve stage (Gurobi) is "parity unpinned":
 * the solver is absent from /root/reference and from this image.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include <ctype.h>

/* ------------------------------------------------------------------ hash */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

/* MurmurHash3_x64_128(key,len,seed=0) folded h1^h2 (ILP_index.cpp:10-18). */
uint64_t orc_hash128_to_64(const void *key, int len)
{
    const uint8_t *data = (const uint8_t *)key;
    const int nblocks = len / 16;
    uint64_t h1 = 0, h2 = 0;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (int i = 0; i < nblocks; i++) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);      /* little-endian block loads */
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    int rem = len & 15;
    for (int i = rem - 1; i >= 8; i--) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; i--) k1 ^= (uint64_t)tail[i] << (8 * i);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 ^ h2;
}

/* ------------------------------------------------------------------ sketch */

/* reverse_strand (ILP_index.cpp:330-357): A<->T, C<->G on upper or lower case, result
 * upper case; every other byte is copied unchanged. */
static inline char comp_base(char c)
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return c;
    }
}

typedef struct { const char *s; int64_t pos; } dq_ent_t;

/*
 * Shared (w,k) minimiser scan of ILP_index.cpp:388-442 / :460-490.
 * seq is upper-cased into a private copy first (:369, :449).  Emits one record per window
 * whose minimum k-mer string hashes differently from the previous window's; out_pos gets
 * the start of that k-mer (deque front position, :423).  Returns the number of records
 * (records beyond cap are counted, not stored).
 */
int64_t orc_sketch(const char *seq_in, int64_t len, int k, int w,
                   uint64_t *out_hash, int64_t *out_pos, int64_t cap)
{
    if (len < (int64_t)w + k - 1) return 0;            /* :372, :453 */
    char *seq = (char *)malloc((size_t)len * 2);
    char *rc = seq + len;
    for (int64_t i = 0; i < len; i++) seq[i] = (char)toupper((unsigned char)seq_in[i]);
    for (int64_t i = 0; i < len; i++) rc[len - 1 - i] = comp_base(seq[i]);
    /* monotone deque over k-mer start positions, as a ring of w+1 entries */
    dq_ent_t *dq = (dq_ent_t *)malloc(sizeof(dq_ent_t) * (size_t)(w + 2));
    int head = 0, cnt = 0, ring = w + 2;
    uint64_t prev_hash = UINT64_MAX;
    int64_t prev_front = -1;
    uint64_t front_hash = 0;
    int64_t n = 0;
    for (int64_t i = 0; i + k <= len; i++) {
        const char *f = seq + i, *r = rc + (len - k - i);
        const char *m = memcmp(r, f, (size_t)k) < 0 ? r : f;   /* std::min(fwd, rev) */
        while (cnt > 0 && memcmp(dq[(head + cnt - 1) % ring].s, m, (size_t)k) >= 0) cnt--;
        dq[(head + cnt) % ring].s = m; dq[(head + cnt) % ring].pos = i; cnt++;
        if (cnt > 0 && dq[head].pos <= i - w) { head = (head + 1) % ring; cnt--; }
        if (i >= w - 1) {
            const dq_ent_t *best = &dq[head];
            if (best->pos != prev_front) {          /* same entry => same string => same hash */
                front_hash = orc_hash128_to_64(best->s, k);
                prev_front = best->pos;
            }
            if (front_hash != prev_hash) {
                prev_hash = front_hash;
                if (n < cap) {
                    if (out_hash) out_hash[n] = front_hash;
                    if (out_pos) out_pos[n] = best->pos;
                }
                n++;
            }
        }
    }
    free(dq); free(seq);
    return n;
}

/* Threads of the OpenMP loops below (the reference's -t, main.cpp:60; its loops: ILP_index.cpp:559, :617). */
void orc_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
int orc_max_threads(void) { return omp_get_max_threads(); }

/* compute_hashes over a batch of reads (ILP_index.cpp:617-621): returns the emitted minimisers. */
int64_t orc_sketch_reads(const char *reads_concat, const int64_t *read_off, int64_t n_reads, int k, int w)
{
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : total)
    for (int64_t r = 0; r < n_reads; r++)
        total += orc_sketch(reads_concat + read_off[r], read_off[r + 1] - read_off[r], k, w, 0, 0, 0);
    return total;
}

/* ------------------------------------------------------------------ stages 1-2 driver */

typedef struct {
    /* inputs kept by reference to caller memory only during orc_run */
    int32_t n_walks;
    /* per-walk outputs */
    int64_t *n_minimizers;      /* [n_walks] :563 */
    int64_t *n_anchors;         /* [n_walks] after the filter, :725-735 */
    /* spectrum */
    int64_t spectrum_size;      /* |Sp_R| :641 */
    uint64_t *spectrum;         /* sorted unique read hashes */
    int64_t filtered, retained; /* :719-721 */
    int64_t n_in_model;         /* count_kmer_matches :831/:877 */
    /* kept anchors, grouped by spectrum id r ascending, then walk h, then position */
    int64_t n_kept;
    int32_t *a_r;               /* spectrum id */
    int32_t *a_h;               /* walk */
    int32_t *a_t0, *a_t1;       /* first/last walk index of the vertices under the k-mer */
    int64_t *a_pos;             /* start base of the k-mer in the walk sequence */
    /* all walk minimisers before matching (for kernel parity): concatenated per walk */
    int64_t *m_off;             /* [n_walks+1] */
    uint64_t *m_hash;
    int64_t *m_pos;
    double stage_s[4];          /* wall seconds: walk sketch, read sketch + spectrum, anchors, filter */
} orc_result_t;

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static int64_t lower_bound_u64(const uint64_t *a, int64_t n, uint64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { int32_t r, h, t0, t1; int64_t pos; } anc_t;

/* vertex-list equality of two anchors == equality of the "v1_v2_..._" keys of :680-683 */
static int same_vertex_list(const anc_t *a, const anc_t *b, const int32_t *walk_vtx,
                            const int64_t *walk_off, const int64_t *node_len)
{
    const int32_t *va = walk_vtx + walk_off[a->h], *vb = walk_vtx + walk_off[b->h];
    int32_t i = a->t0, j = b->t0;
    for (;;) {
        while (i <= a->t1 && node_len[va[i]] == 0) i++;   /* empty segments own no base */
        while (j <= b->t1 && node_len[vb[j]] == 0) j++;
        if (i > a->t1 || j > b->t1) return i > a->t1 && j > b->t1;
        if (va[i] != vb[j]) return 0;
        i++; j++;
    }
}

static int32_t n_vertices(const anc_t *a, const int32_t *walk_vtx, const int64_t *walk_off,
                          const int64_t *node_len)
{
    const int32_t *v = walk_vtx + walk_off[a->h];
    int32_t n = 0;
    for (int32_t i = a->t0; i <= a->t1; i++) n += node_len[v[i]] > 0;
    return n;
}

/*
 * Flattened graph in, reads in, counters and kept anchors out.
 *   seq_concat/seq_off : node sequences, original case        (ILP_index.cpp:30-36)
 *   walk_off/walk_vtx  : paths[h]                              (:96-113)
 * Walks follow graph edges of a DAG, so the unique vertices under a k-mer, sorted by
 * topological rank (:419-438), are the consecutive walk entries t0..t1 that own a base.
 */
orc_result_t *orc_run(int32_t n_vtx, const char *seq_concat, const int64_t *seq_off,
                      int32_t n_walks, const int64_t *walk_off, const int32_t *walk_vtx,
                      const char *reads_concat, const int64_t *read_off, int64_t n_reads,
                      int k, int w, float threshold)
{
    orc_result_t *R = (orc_result_t *)calloc(1, sizeof(*R));
    R->n_walks = n_walks;
    R->n_minimizers = (int64_t *)calloc((size_t)n_walks, 8);
    R->n_anchors = (int64_t *)calloc((size_t)n_walks, 8);
    R->m_off = (int64_t *)calloc((size_t)n_walks + 1, 8);
    int64_t *node_len = (int64_t *)malloc(8 * (size_t)(n_vtx > 0 ? n_vtx : 1));
    for (int32_t v = 0; v < n_vtx; v++) node_len[v] = seq_off[v + 1] - seq_off[v];

    double t_stage = omp_get_wtime();
    /* ---- stage 1a: walks (index_kmers), one OpenMP task per walk as ILP_index.cpp:559 */
    int64_t m_n = 0;
    int64_t **wbase = (int64_t **)calloc((size_t)n_walks, sizeof(int64_t *));
    uint64_t **wh = (uint64_t **)calloc((size_t)n_walks, sizeof(uint64_t *));
    int64_t **wp = (int64_t **)calloc((size_t)n_walks, sizeof(int64_t *));
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t h = 0; h < n_walks; h++) {
        int64_t nv = walk_off[h + 1] - walk_off[h];
        const int32_t *wv = walk_vtx + walk_off[h];
        int64_t *base = (int64_t *)malloc(8 * (size_t)(nv + 1));
        base[0] = 0;
        for (int64_t i = 0; i < nv; i++) base[i + 1] = base[i] + node_len[wv[i]];
        wbase[h] = base;
        int64_t L = base[nv];
        char *hap = (char *)malloc((size_t)(L > 0 ? L : 1));
        for (int64_t i = 0; i < nv; i++)
            memcpy(hap + base[i], seq_concat + seq_off[wv[i]], (size_t)node_len[wv[i]]);
        int64_t n = orc_sketch(hap, L, k, w, 0, 0, 0);
        wh[h] = (uint64_t *)malloc(8 * (size_t)(n + 1));
        wp[h] = (int64_t *)malloc(8 * (size_t)(n + 1));
        orc_sketch(hap, L, k, w, wh[h], wp[h], n);
        R->n_minimizers[h] = n;
        free(hap);
    }
    for (int32_t h = 0; h < n_walks; h++) { m_n += R->n_minimizers[h]; R->m_off[h + 1] = m_n; }
    R->m_hash = (uint64_t *)malloc(8 * (size_t)(m_n + 1));
    R->m_pos = (int64_t *)malloc(8 * (size_t)(m_n + 1));
    for (int32_t h = 0; h < n_walks; h++) {
        memcpy(R->m_hash + R->m_off[h], wh[h], 8 * (size_t)R->n_minimizers[h]);
        memcpy(R->m_pos + R->m_off[h], wp[h], 8 * (size_t)R->n_minimizers[h]);
        free(wh[h]); free(wp[h]);
    }
    free(wh); free(wp);

    R->stage_s[0] = omp_get_wtime() - t_stage; t_stage = omp_get_wtime();
    /* ---- stage 1b: reads (compute_hashes, OpenMP over reads as :617) + spectrum (:615-638) */
    int64_t *r_off = (int64_t *)calloc((size_t)n_reads + 1, 8);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_reads; r++)
        r_off[r + 1] = orc_sketch(reads_concat + read_off[r], read_off[r + 1] - read_off[r], k, w, 0, 0, 0);
    for (int64_t r = 0; r < n_reads; r++) r_off[r + 1] += r_off[r];
    int64_t sp_n = r_off[n_reads];
    uint64_t *sp = (uint64_t *)malloc(8 * (size_t)(sp_n + 1));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_reads; r++)
        orc_sketch(reads_concat + read_off[r], read_off[r + 1] - read_off[r], k, w, sp + r_off[r], 0, r_off[r + 1] - r_off[r]);
    free(r_off);
    qsort(sp, (size_t)sp_n, 8, cmp_u64);
    int64_t u = 0;
    for (int64_t i = 0; i < sp_n; i++) if (i == 0 || sp[i] != sp[i - 1]) sp[u++] = sp[i];
    R->spectrum = sp; R->spectrum_size = u;

    R->stage_s[1] = omp_get_wtime() - t_stage; t_stage = omp_get_wtime();
    /* ---- stage 2a: anchors (compute_anchors, :495-526 + :645-655).  Walks in parallel (the reference: minimisers
     *      of one walk in parallel, :499); Anchor_hits[r][h][k] order = by spectrum id, then walk, then position:
     *      a stable counting sort by r of the walks' lists taken in walk order. */
    anc_t **wa = (anc_t **)calloc((size_t)n_walks, sizeof(anc_t *));
    int64_t *wa_n = (int64_t *)calloc((size_t)n_walks + 1, 8);
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t h = 0; h < n_walks; h++) {
        const int64_t *base = wbase[h];
        int64_t cap = 1 << 12, n = 0, t = 0;
        anc_t *a = (anc_t *)malloc(sizeof(anc_t) * (size_t)cap);
        for (int64_t i = R->m_off[h]; i < R->m_off[h + 1]; i++) {
            uint64_t hash = R->m_hash[i];
            int64_t id = lower_bound_u64(sp, u, hash);
            if (id >= u || sp[id] != hash) continue;
            int64_t p = R->m_pos[i];
            /* positions are non-decreasing along a walk: advance t to the owner of base p */
            while (!(base[t] <= p && p < base[t + 1])) t++;
            int64_t t1 = t;
            while (!(base[t1] <= p + k - 1 && p + k - 1 < base[t1 + 1])) t1++;
            if (n == cap) { cap *= 2; a = (anc_t *)realloc(a, sizeof(anc_t) * (size_t)cap); }
            a[n].r = (int32_t)id; a[n].h = h; a[n].t0 = (int32_t)t; a[n].t1 = (int32_t)t1; a[n].pos = p;
            n++;
        }
        wa[h] = a; wa_n[h] = n;
    }
    int64_t a_n = 0;
    for (int32_t h = 0; h < n_walks; h++) a_n += wa_n[h];
    /* g_off[r] .. g_off[r + 1]: the anchors of spectrum id r */
    int64_t *g_off = (int64_t *)calloc((size_t)u + 2, 8);
    for (int32_t h = 0; h < n_walks; h++)
        for (int64_t i = 0; i < wa_n[h]; i++) g_off[wa[h][i].r + 1]++;
    for (int64_t r = 0; r < u; r++) g_off[r + 1] += g_off[r];
    anc_t *anc = (anc_t *)malloc(sizeof(anc_t) * (size_t)(a_n + 1));
#pragma omp parallel
    {
        /* every thread places the ids of its own range, scanning the walks in order: stable */
        int nt = omp_get_num_threads(), me = omp_get_thread_num();
        int64_t r_lo = u * me / nt, r_hi = u * (me + 1) / nt;
        int64_t *cur = (int64_t *)malloc(8 * (size_t)(r_hi - r_lo + 1));
        for (int64_t r = r_lo; r < r_hi; r++) cur[r - r_lo] = g_off[r];
        for (int32_t h = 0; h < n_walks; h++)
            for (int64_t i = 0; i < wa_n[h]; i++) {
                int64_t r = wa[h][i].r;
                if (r >= r_lo && r < r_hi) anc[cur[r - r_lo]++] = wa[h][i];
            }
        free(cur);
    }
    for (int32_t h = 0; h < n_walks; h++) free(wa[h]);
    free(wa); free(wa_n);

    R->stage_s[2] = omp_get_wtime() - t_stage; t_stage = omp_get_wtime();
    /* ---- stage 2b: filter (:670-722), spectrum ids in parallel as :674: per id "dropped" and "has an anchor over
     *      two vertices or more"; the kept anchors are then copied out in id order */
    R->a_r = (int32_t *)malloc(4 * (size_t)(a_n + 1)); R->a_h = (int32_t *)malloc(4 * (size_t)(a_n + 1));
    R->a_t0 = (int32_t *)malloc(4 * (size_t)(a_n + 1)); R->a_t1 = (int32_t *)malloc(4 * (size_t)(a_n + 1));
    R->a_pos = (int64_t *)malloc(8 * (size_t)(a_n + 1));
    int64_t kept = 0, filtered = 0, in_model = 0;
    const float limit = threshold * (float)(uint32_t)n_walks;     /* threshold * num_walks, :698 */
    uint8_t *verdict = (uint8_t *)calloc((size_t)u + 1, 1);      /* bit 0: dropped, bit 1: in the model */
#pragma omp parallel
    {
        int32_t *grp = 0; int64_t grp_cap = 0;
#pragma omp for schedule(dynamic, 1024) reduction(+ : filtered, in_model)
        for (int64_t r = 0; r < u; r++) {
            int64_t s = g_off[r], e = g_off[r + 1], m = e - s;
            if (m == 0) continue;
            if (m > grp_cap) { grp_cap = m * 2; grp = (int32_t *)realloc(grp, 4 * (size_t)grp_cap); }
            /* group by vertex list, quadratic in the (small) anchors-per-minimiser count */
            int drop = 0;
            for (int64_t i = 0; i < m; i++) grp[i] = -1;
            for (int64_t i = 0; i < m && !drop; i++) {
                if (grp[i] >= 0) continue;
                int32_t c = 0;
                for (int64_t j = i; j < m; j++)
                    if (grp[j] < 0 && same_vertex_list(&anc[s + i], &anc[s + j], walk_vtx, walk_off, node_len)) {
                        grp[j] = (int32_t)i; c++;
                    }
                if ((float)c >= limit) drop = 1;
            }
            if (drop) { filtered++; verdict[r] = 1; continue; }
            int has_multi = 0;
            for (int64_t i = s; i < e && !has_multi; i++)
                if (n_vertices(&anc[i], walk_vtx, walk_off, node_len) >= 2) has_multi = 1;      /* :795/:846 */
            in_model += has_multi;                                                               /* :822/:868 */
            verdict[r] = (uint8_t)(has_multi << 1);
        }
        free(grp);
    }
    for (int64_t r = 0; r < u; r++) {
        if (verdict[r] & 1) continue;
        for (int64_t i = g_off[r]; i < g_off[r + 1]; i++) {
            R->a_r[kept] = anc[i].r; R->a_h[kept] = anc[i].h;
            R->a_t0[kept] = anc[i].t0; R->a_t1[kept] = anc[i].t1; R->a_pos[kept] = anc[i].pos;
            kept++;
            R->n_anchors[anc[i].h]++;
        }
    }
    free(verdict); free(g_off);
    R->stage_s[3] = omp_get_wtime() - t_stage;
    R->n_kept = kept;
    R->filtered = filtered;
    R->retained = u - filtered;           /* ids without anchors count as retained, :721 */
    R->n_in_model = in_model;

    free(anc); free(node_len);
    for (int32_t h = 0; h < n_walks; h++) free(wbase[h]);
    free(wbase);
    return R;
}

void orc_free(orc_result_t *R)
{
    if (!R) return;
    free(R->n_minimizers); free(R->n_anchors); free(R->spectrum);
    free(R->a_r); free(R->a_h); free(R->a_t0); free(R->a_t1); free(R->a_pos);
    free(R->m_off); free(R->m_hash); free(R->m_pos);
    free(R);
}

/* plain getters so the ctypes side never depends on struct layout */
double orc_stage_seconds(const orc_result_t *R, int i) { return (i >= 0 && i < 4) ? R->stage_s[i] : 0.0; }
int64_t orc_spectrum_size(const orc_result_t *R) { return R->spectrum_size; }
const uint64_t *orc_spectrum(const orc_result_t *R) { return R->spectrum; }
int64_t orc_filtered(const orc_result_t *R) { return R->filtered; }
int64_t orc_retained(const orc_result_t *R) { return R->retained; }
int64_t orc_n_in_model(const orc_result_t *R) { return R->n_in_model; }
const int64_t *orc_n_minimizers(const orc_result_t *R) { return R->n_minimizers; }
const int64_t *orc_n_anchors(const orc_result_t *R) { return R->n_anchors; }
int64_t orc_n_kept(const orc_result_t *R) { return R->n_kept; }
const int32_t *orc_a_r(const orc_result_t *R) { return R->a_r; }
const int32_t *orc_a_h(const orc_result_t *R) { return R->a_h; }
const int32_t *orc_a_t0(const orc_result_t *R) { return R->a_t0; }
const int32_t *orc_a_t1(const orc_result_t *R) { return R->a_t1; }
const int64_t *orc_a_pos(const orc_result_t *R) { return R->a_pos; }
const int64_t *orc_m_off(const orc_result_t *R) { return R->m_off; }
const uint64_t *orc_m_hash(const orc_result_t *R) { return R->m_hash; }
const int64_t *orc_m_pos(const orc_result_t *R) { return R->m_pos; }
